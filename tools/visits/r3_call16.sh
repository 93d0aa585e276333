#!/bin/bash
set -u
exec < /dev/null
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_train.py -q -s -k "gradients or conditioned" -p no:cacheprovider 2>&1 | grep -v "^$\|Warning\|warn" | tail -40 | cut -c1-330 | tee $O/r3c16_pytest_train.log
exit 0
