#!/bin/bash
set -u
exec < /dev/null
O=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $O
for S in 2 1 2; do
  echo "== --streams $S"
  timeout 600 python bench.py --steps 600 --warmup 20 --no-cpu-baseline --no-train-step --streams $S 2>&1 | tail -1 > $O/r3c49_bench_s$S.json
  python - <<PY
import json
d = json.load(open("gpurun_out/r3c49_bench_s$S.json"))
print(d["value"], d["ms_per_step"], "K1 frac", d["roofline"]["frac"], d["roofline"].get("frac_single_stream"), d["roofline"]["per_stage_us"], d.get("single_stream"), "conv us", d["roofline_conv"]["us_per_scene"])
PY
done | tee $O/r3c49_streams.txt
timeout 600 python -m pytest tests/test_gpu_parity.py -q -x -p no:cacheprovider -k scene_pipeline 2>&1 | tail -2
exit 0
