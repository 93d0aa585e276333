#!/bin/bash
# visit 80: stage 3's planes kernel reads stale values of stage 2's depth -- agent-scope loads in the reader / agent-scope release fence in the writer
mkdir -p gpurun_out; L=gpurun_out/r3c80.log; : > $L
for v in "" sc1 fence both; do
  echo "== variant '${v:-default}'" >> $L
  env SCENES=40 ROUNDS=6 CAPTURE_NAMES=hypothesis_planes ${v:+RCMVS_LIB=tools/dev/_variants/lib_$v.so} timeout 300 python tools/dev/two_stream_firstbad.py 2>&1 | grep "scenes differ somewhere\|first differing" | sed 's/(stage 3 starts.*//' | sed 's/round [0-9]* scene [0-9]* (stream [01]): //' | sort | uniq -c | sort -rn | head -12 >> $L
done
cat $L
