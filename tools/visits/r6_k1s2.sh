#!/bin/bash
mkdir -p gpurun_out
{ python tools/dev/k1_s2_time.py; for l in "$@"; do RCMVS_LIB=tools/dev/_variants/$l python tools/dev/k1_s2_time.py; done; python tools/dev/k1_s2_time.py; } 2>&1 | grep -v "Warning\|amdgpu.ids" | tee gpurun_out/r6_k1s2.txt
