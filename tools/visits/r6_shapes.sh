#!/bin/bash
# Round 6: the K1 / guard GPU tests, then bench.py at the three shapes (default = BASELINE configs[1]; dtu_eval; tanks).  usage: r6_shapes.sh <tag>
mkdir -p gpurun_out
tag=${1:-a}
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_wrappers.py -x -q -m gpu -k "warp_variance or second_stream or cascade_vs_reference_golden" 2>&1 | tail -5 | tee gpurun_out/r6_shapes_tests_$tag.log
for shape in dtu_bench dtu_eval tanks; do
    timeout 900 python bench.py --shape $shape > gpurun_out/r6_bench_${shape}_$tag.json 2> gpurun_out/r6_bench_${shape}_$tag.err
    tail -c 300 gpurun_out/r6_bench_${shape}_$tag.err
    python - <<PY
import json
b = json.load(open("gpurun_out/r6_bench_${shape}_$tag.json"))
print("$shape", "value", b["value"], "ms", b["ms_per_step"], "K1 frac", b["roofline"]["frac"], b["roofline"]["per_stage_us"], "scene frac", b["roofline_scene"]["frac"], "parity", (b.get("parity") or {}).get("depth_l1_over_range"))
print("   conv", b["roofline_conv"]["us_per_scene"], b["roofline_conv"]["frac_hbm"], "rounds", b["timed_rounds"])
PY
done
