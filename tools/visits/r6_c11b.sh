#!/bin/bash
# Round 6: fused conv11 + prob: tests, then same-box A/B of bench.py with RCMVS_CONV11_PROB = 0 / 1 / 2.  usage: r6_c11b.sh <tag>
mkdir -p gpurun_out
tag=${1:-a}
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "conv11_prob or cascade_vs_reference_golden or margin or depth_head" 2>&1 | tail -4 | tee gpurun_out/r6_c11b_tests_$tag.log
for v in 0 1 2 0 1 2; do
    RCMVS_CONV11_PROB=$v timeout 600 python bench.py --steps 300 --warmup 20 --no-cpu-baseline --no-train-step --no-side-pass > gpurun_out/r6_c11b_bench_${v}_$tag.json 2> gpurun_out/r6_c11b_bench_$tag.err
    python - <<PY
import json
b = json.load(open("gpurun_out/r6_c11b_bench_${v}_$tag.json"))
print("CONV11_PROB=$v value", b["value"], "ms", b["ms_per_step"], "K1", b["roofline"]["per_stage_us"], "conv us", b["roofline_conv"]["us_per_scene"])
PY
done
