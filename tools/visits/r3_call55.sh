#!/bin/bash
set -u
exec < /dev/null
O=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $O
for tag in none f p k1 cr dh none; do
  n=$(RCMVS_DEBUG_SYNC=$tag ROUNDS=8 timeout 200 python tools/dev/two_stream_check.py 2>&1 | grep "two streams" | grep -vc "'depth': 0.0, 'photometric_confidence': 0.0")
  echo "RCMVS_DEBUG_SYNC=$tag: $n of 8 rounds corrupted"
done | tee $O/r3c57_sync_bisect.txt
exit 0
