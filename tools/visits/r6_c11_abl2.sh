#!/bin/bash
mkdir -p gpurun_out
{
python tools/dev/c11_time2.py
for m in "$@"; do RCMVS_LIB=tools/dev/_variants/lib_c11_$m.so python tools/dev/c11_time2.py; done
} 2>&1 | grep -v "Warning\|amdgpu.ids" | tee gpurun_out/r6_c11_abl2.txt
