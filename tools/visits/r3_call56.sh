#!/bin/bash
set -u
exec < /dev/null
O=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $O
for rep in 1 2; do
  echo "== run $rep, 10 rounds"
  ROUNDS=10 timeout 300 python tools/dev/two_stream_check.py 2>&1 | grep "two streams" | sed 's/worst |diff| over 16 scenes: //' | cut -c1-120
done | tee $O/r3c56_repro.txt
exit 0
