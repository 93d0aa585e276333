#!/bin/bash
set -u
exec < /dev/null
O=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $O
for mode in 0 2 1 0 2; do
  echo "== RCMVS_FP16_PAIR=$mode"
  RCMVS_FP16_PAIR=$mode timeout 600 python bench.py --steps 300 --warmup 10 --no-cpu-baseline --no-train-step 2>&1 | tail -1 > $O/r3c40_bench_$mode.json
  python - <<PY
import json
d = json.load(open("gpurun_out/r3c40_bench_$mode.json"))
print(d["ms_per_step"], "K1", d["roofline"]["per_stage_us"], "conv", d["roofline_conv"]["us_per_scene"], d["roofline_conv"]["largest_layers_us_tflops"])
PY
done | tee $O/r3c40_fp16_pair_pipeline.txt
exit 0
