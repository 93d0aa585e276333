#!/bin/bash
set -u
exec < /dev/null
O=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $O
{
for np in 2 3; do
  for ym in 0 1; do
    echo "== X3_NP=$np ymax=$ym"
    if [ $ym = 1 ]; then export X3_YMAX=1; else unset X3_YMAX; fi
    X3_NOCHECK=1 X3_NORES=1 X3_NP=$np timeout 60 tools/dev/x3_test 3 2>&1 | grep "time" | grep -v "kind=3" | cut -c1-110
  done
done
} | tee $O/r3c41_x3_ymax_cost.txt
exit 0
