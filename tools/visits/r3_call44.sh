#!/bin/bash
set -u
exec < /dev/null
O=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $O
timeout 1800 python -m pytest tests -q -m gpu -p no:cacheprovider 2>&1 | tail -12 | tee $O/r3c44_pytest.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
exit 0
