#!/bin/bash
set -u
exec < /dev/null
O=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $O
timeout 120 python tools/dev/bn_time.py 2>&1 | grep -v amdgpu.ids | tee $O/r3c46_bn_time.txt
timeout 900 python -m pytest tests/test_gpu_train.py -q -x -p no:cacheprovider -k "conv_bn or fused_batchnorm or train_step_then or sync_batchnorm" 2>&1 | tail -2
timeout 600 python bench.py --workload train_step 2>&1 | tail -1 | cut -c1-300
exit 0
