#!/bin/bash
mkdir -p gpurun_out
{
python tools/dev/c11_time.py
for m in "$@"; do RCMVS_LIB=tools/dev/_variants/lib_c11_$m.so python tools/dev/c11_time.py; done
for z in 2 4 8 16; do ZC=$z python tools/dev/c11_time.py; done
} 2>&1 | grep -v "Warning\|amdgpu.ids" | tee gpurun_out/r6_c11_abl.txt
