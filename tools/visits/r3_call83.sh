#!/bin/bash
# visit 83: soak of the two-stream mode with the adopted planes-kernel loads (graph replay check fixed; full outputs compared)
mkdir -p gpurun_out; L=gpurun_out/r3c83.log; : > $L
STEPS=1200 timeout 300 python tools/dev/two_stream_graph_probe.py 2>&1 | grep -v "Warning\|amdgpu.ids" | tail -3 >> $L
env SCENES=200 ROUNDS=8 timeout 300 python tools/dev/two_stream_depth.py 2>&1 | grep -v "Warning\|amdgpu.ids" | tail -1 | cut -c1-200 >> $L
ROUNDS=6 timeout 300 python tools/dev/two_stream_check.py 2>&1 | grep -v "Warning\|amdgpu.ids" | tail -8 >> $L
cat $L
