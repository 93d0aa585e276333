#!/bin/bash
set -u
exec < /dev/null
O=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $O
for k in elementwise matmul fill; do FOREIGN=$k timeout 200 python tools/dev/two_stream_foreign.py 2>&1 | grep "foreign work"; done | tee $O/r3c62_foreign.txt
exit 0
