#!/bin/bash
set -u
exec < /dev/null
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $O
echo "== copy hunt"
timeout 300 python tools/dev/copy_hunt.py 2>&1 | grep -v Warning | head -70 | cut -c1-260 | tee $O/r3c19_copy_hunt.txt
echo "== graph probe"
timeout 300 python tools/dev/graph_probe.py 2>&1 | tail -5 | tee $O/r3c19_graph_probe.txt
exit 0
