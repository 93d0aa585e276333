#!/bin/bash
# round 3, GPU visit 4: phase ablation of the split-operand conv kernel, both arithmetic forms (tools/dev/x3_test, X3_ABLATION build)
set -u
exec < /dev/null
O=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $O
{
echo "# X3_DBG bits: 1 no MFMAs, 2 no split / ring stores, 4 no input loads, 8 no output stores, 16 no B-fragment reads"
for np in 3 2; do
  for dbg in 0 1 16 17 2 4 6 8 14 31; do
    echo "== X3_NP=$np X3_DBG=$dbg"
    X3_NOCHECK=1 X3_NORES=1 X3_NP=$np X3_DBG=$dbg timeout 60 tools/dev/x3_test 3 2>&1 | grep "time" | cut -c1-110
  done
done
} | tee $O/r3c4_x3_ablation.txt
exit 0
