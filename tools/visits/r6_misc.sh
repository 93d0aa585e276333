#!/bin/bash
mkdir -p gpurun_out
python "$@" 2>&1 | grep -v "Warning\|amdgpu.ids" | tee gpurun_out/r6_misc.txt
