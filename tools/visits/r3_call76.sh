#!/bin/bash
# visit 76: first differing op of a corrupted two-stream scene (asynchronous capture of every ops-layer call)
mkdir -p gpurun_out; L=gpurun_out/r3c76.log; : > $L
env SCENES=16 ROUNDS=4 timeout 300 python tools/dev/two_stream_firstbad.py 2>&1 | grep -v "Warning\|amdgpu.ids" | tail -80 >> $L
cat $L
