#!/bin/bash
set -u
exec < /dev/null
O=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $O
{
for b in x3_test x3_test_tall; do
  echo "== $b X3_NP=2 (accuracy check on, small volumes) "
  X3_NP=2 timeout 120 tools/dev/$b 1 2>&1 | grep -i "kind=0 Ci=\(16\|32\) Co=8\|FAIL\|err" | head -6 | cut -c1-200
  echo "== $b X3_NP=2 timings"
  X3_NOCHECK=1 X3_NORES=1 X3_NP=2 timeout 60 tools/dev/$b 3 2>&1 | grep "time" | grep -v "kind=3" | cut -c1-110
done
} | tee $O/r3c45_x3_tall.txt
exit 0
