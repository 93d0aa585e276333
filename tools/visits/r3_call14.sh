#!/bin/bash
set -u
exec < /dev/null
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $O
timeout 600 python tools/dev/k1_grad_probe.py 2>&1 | grep -v "^/opt\|Warning\|warn" | tail -8 | cut -c1-330 | tee $O/r3c14_k1_grad_probe.txt
exit 0
