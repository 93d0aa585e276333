#!/bin/bash
set -u
exec < /dev/null
O=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $O
{
for rep in 1 2; do
  n=$(ROUNDS=10 timeout 300 python tools/dev/two_stream_check.py 2>&1 | grep "two streams" | grep -vc "'depth': 0.0, 'photometric_confidence': 0.0")
  echo "X3_ZERO_LDS build, default arithmetic: $n of 10 rounds corrupted"
done
MODE=0 timeout 200 python tools/dev/two_stream_x3_bisect.py 2>&1 | grep "MODE="
} | tee $O/r3c63_zero_lds.txt
exit 0
