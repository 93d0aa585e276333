#!/bin/bash
# visit 79: the wrong elements of stage 3's planes tensor -- whose values are they (earlier scene at the same address?), and are they cache-line shaped?
mkdir -p gpurun_out; L=gpurun_out/r3c79.log; : > $L
env SCENES=40 ROUNDS=10 CAPTURE_NAMES=hypothesis_planes timeout 300 python tools/dev/two_stream_firstbad.py 2>&1 | grep -v "Warning\|amdgpu.ids" | tail -90 >> $L
cat $L
