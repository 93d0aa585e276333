#!/bin/bash
set -u
exec < /dev/null
O=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $O
echo "== pytest fp16 pair"
timeout 900 python -m pytest tests/test_gpu_parity.py -q -x -s -p no:cacheprovider -k "fp16_pair or x3h" 2>&1 | grep -E "fp16 pair|passed|failed|Error|assert" | tee $O/r3c42_pytest.log
for mode in 0 1 0 1; do
  echo "== RCMVS_FP16_PAIR=$mode"
  RCMVS_FP16_PAIR=$mode timeout 600 python bench.py --steps 300 --warmup 10 --no-train-step --cpu-scenes 3 2>&1 | tail -1 > $O/r3c42_bench_$mode.json
  python - <<PY
import json
d = json.load(open("gpurun_out/r3c42_bench_$mode.json"))
print(d["ms_per_step"], "K1", d["roofline"]["per_stage_us"], "conv", d["roofline_conv"]["us_per_scene"], "parity", {k: d["parity"][k] for k in ("depth_l1_over_range", "depth_max_abs_mm")}, "smooth", d["parity"]["smooth_head"])
PY
done | tee $O/r3c42_fp16_pair_pipeline.txt
exit 0
