#!/bin/bash
set -u
exec < /dev/null
O=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $O
for gx in 512 256 128; do
  echo "== RCMVS_WGRAD_GX=$gx"
  RCMVS_WGRAD_GX=$gx timeout 120 python tools/dev/wgrad_time.py 2>&1 | grep -v amdgpu.ids
done | tee $O/r3c33_wgrad_gx16.txt
exit 0
