#!/bin/bash
mkdir -p gpurun_out
python tools/dev/layer_times.py 2>&1 | grep -v "Warning\|amdgpu.ids" | tee gpurun_out/r6_layer_times_${1:-a}.txt
