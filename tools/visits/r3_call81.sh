#!/bin/bash
# visit 81: repeat of visit 80 (default / agent-scope loads in the planes kernel, alternating), final outputs captured too
mkdir -p gpurun_out; L=gpurun_out/r3c81.log; : > $L
for v in "" sc1 "" sc1 "" sc1; do
  echo "== variant '${v:-default}'" >> $L
  env SCENES=40 ROUNDS=6 CAPTURE_NAMES=hypothesis_planes,depth_head ${v:+RCMVS_LIB=tools/dev/_variants/lib_$v.so} timeout 300 python tools/dev/two_stream_firstbad.py 2>&1 | grep "scenes differ somewhere\|first differing" | sed 's/(stage 3 starts.*//' | sed 's/round [0-9]* scene [0-9]* (stream [01]): //' | sed 's/round [0-9]*: //' | sort | uniq -c | sort -rn | head -12 >> $L
done
cat $L
