#!/bin/bash
# visit 77: first differing op of a corrupted two-stream scene, light capture (outputs of stage 3 only)
mkdir -p gpurun_out; L=gpurun_out/r3c77.log; : > $L
env SCENES=40 ROUNDS=8 timeout 300 python tools/dev/two_stream_firstbad.py 2>&1 | grep -v "Warning\|amdgpu.ids" | tail -80 >> $L
cat $L
