#!/bin/bash
set -u
exec < /dev/null
O=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $O
for g in 1024 512 256; do
  echo "== RCMVS_BN_RED_GRID=$g"
  RCMVS_BN_RED_GRID=$g timeout 120 python tools/dev/bn_time.py 2>&1 | grep -v amdgpu.ids
done | tee $O/r3c36_bn_grid_unrolled.txt
exit 0
