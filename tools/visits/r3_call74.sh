#!/bin/bash
# visit 74: per-round facts of the two-stream corruption (rounds 1 and 6 of the probe are clean every time): streams, speed, address ranges
mkdir -p gpurun_out; L=gpurun_out/r3c74.log; : > $L
env SCENES=40 LOGADDR=1 MINBYTES=16000000 timeout 120 python tools/dev/two_stream_depth.py 2>&1 | grep -v "Warning\|amdgpu.ids" >> $L
cat $L
