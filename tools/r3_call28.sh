#!/bin/bash
set -u
exec < /dev/null
O=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $O
for gx in 2048 1024 512 256; do
  echo "== RCMVS_WGRAD_GX=$gx"
  RCMVS_WGRAD_GX=$gx timeout 120 python tools/dev/wgrad_time.py 2>&1 | grep -v amdgpu.ids
done | tee $O/r3c29_wgrad_gx.txt
exit 0
