#!/bin/bash
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_parity.py -q -m gpu -s -k "warp_variance_timed or resize_rgb" 2>&1 | grep "K1 stage\|passed\|failed\|Error" | tee gpurun_out/r6_t2.log
timeout 600 python bench.py --steps 300 --no-cpu-baseline --no-train-step --no-side-pass > gpurun_out/r6_t2_bench.json 2>/dev/null
python - <<PY
import json
b = json.load(open("gpurun_out/r6_t2_bench.json"))
print("value", b["value"], "K1 frac", b["roofline"]["frac"], b["roofline"]["per_stage_us"], "smooth", b["roofline"]["smooth_scene"]["frac"], b["roofline"]["smooth_scene"]["per_stage_us"])
PY
