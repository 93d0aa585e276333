#!/bin/bash
# Round 4, K1 lab visits (profiles/r4_k1_walls.txt).  In the build container first:
#   python tools/dev/k1_lab/gen_data.py
#   hipcc --offload-arch=gfx950 -O3 -std=c++17 -o tools/dev/_bin/k1_lab tools/dev/k1_lab/lab.hip
#   hipcc --offload-arch=gfx950 -O3 -std=c++17 -o tools/dev/_bin/valu_rate tools/dev/k1_lab/valu_rate.hip
# then, one gpurun call each:  tools/dev/_bin/k1_lab <variants>   (0-3 product debug variants, 20-36 the lab's)  |  tools/dev/_bin/valu_rate
mkdir -p gpurun_out
timeout 200 tools/dev/_bin/k1_lab "${@:-0 20 21 22 23 24 3 0}" 2>&1 | tee gpurun_out/r4_lab.log
