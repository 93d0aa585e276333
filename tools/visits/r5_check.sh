#!/bin/bash
# Round 5: K1-related GPU tests + one bench line.  usage: r5_check.sh <tag> [pytest -k expression]
mkdir -p gpurun_out
tag=$1; kexpr=${2:-"warp_variance or margin or two_scenes or cascade_vs_reference_golden"}
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -s -k "$kexpr" 2>&1 | tail -40 | tee gpurun_out/r5_check_$tag.log
timeout 600 python bench.py --steps 300 --warmup 20 > gpurun_out/r5_bench_$tag.json 2> gpurun_out/r5_bench_$tag.err
tail -c 600 gpurun_out/r5_bench_$tag.err
python - <<PY
import json
b = json.load(open("gpurun_out/r5_bench_$tag.json"))
print("value", b["value"], "ms", b["ms_per_step"], "roofline", b["roofline"]["frac"], b["roofline"]["per_stage_us"], "smooth", b["roofline"].get("smooth_scene"))
print("parity", b.get("parity")); print("two_procs", b.get("two_procs_per_gpu")); print("train", b.get("train_step")); print("conv", {k: b["roofline_conv"][k] for k in list(b["roofline_conv"])[:6]})
PY
