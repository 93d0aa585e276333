#!/bin/bash
# what would a pre-split activation format buy?  the same kernel with the split arithmetic removed (wrong results, timing only)
set -u
exec < /dev/null
O=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $O
{
for b in x3_test x3_test_fake; do
  echo "== $b (trace build)"
  X3_TRACE=1 X3_NOCHECK=1 X3_NORES=1 X3_NP=3 timeout 60 tools/dev/${b}_trace 3 2>&1 | grep "trace" | cut -c1-260
  echo "== $b"
  X3_NOCHECK=1 X3_NORES=1 X3_NP=3 timeout 60 tools/dev/$b 3 2>&1 | grep "time" | cut -c1-110
done
} | tee $O/r3c20_x3_fake_split.txt
exit 0
