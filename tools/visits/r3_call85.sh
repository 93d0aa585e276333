#!/bin/bash
# visit 85: is the stale sector in the CU's vector L1 or in the XCD's L2?  planes kernel with plain / workgroup-scope (sc0) loads, alternating
mkdir -p gpurun_out; L=gpurun_out/r3c85.log; : > $L
for v in plain wg plain wg plain wg; do
  echo "== variant '$v'" >> $L
  env SCENES=40 ROUNDS=6 CAPTURE_NAMES=hypothesis_planes,depth_head RCMVS_LIB=tools/dev/_variants/lib_$v.so timeout 300 python tools/dev/two_stream_firstbad.py 2>&1 | grep "scenes differ somewhere\|first differing" | sed 's/(stage 3 starts.*//' | sed 's/round [0-9]* scene [0-9]* (stream [01]): //' | sed 's/round [0-9]*: //' | sort | uniq -c | sort -rn | head -8 >> $L
done
cat $L
