#!/bin/bash
# Round 5, K1 lab visits (profiles/r5_k1_window.txt).  In the build container first:
#   python tools/dev/k1_lab/gen_data.py; K1_GAIN=1 python tools/dev/k1_lab/gen_data.py
#   hipcc --offload-arch=gfx950 -O3 -std=c++17 -o tools/dev/_bin/k1_lab tools/dev/k1_lab/lab.hip
# usage: r5_k1_lab.sh <tag> <variants...>      -> gpurun_out/r5_lab_<tag>.log (bench scene, then its smooth-head twin)
mkdir -p gpurun_out
tag=$1; shift
{
  echo "== bench scene (prob.weight x20): variants $*"
  timeout 300 tools/dev/_bin/k1_lab "$@"
  echo "== smooth-head scene (prob.weight x1): variants $*"
  K1_LAB_DATA=tools/dev/k1_lab/data_smooth timeout 300 tools/dev/_bin/k1_lab "$@"
} 2>&1 | tee gpurun_out/r5_lab_$tag.log
