#!/bin/bash
set -u
exec < /dev/null
O=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $O
for m in 0 1 2 3 0; do MODE=$m timeout 200 python tools/dev/two_stream_x3_bisect.py 2>&1 | grep "MODE="; done | tee $O/r3c59_x3_bisect.txt
exit 0
