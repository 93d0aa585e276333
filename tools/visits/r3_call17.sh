#!/bin/bash
# round 3, GPU visit 17+: producer variants: trace, harness timings, x3 parity, bench
set -u
exec < /dev/null
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out
tag=${1:-r3c17}
mkdir -p $O
{
for np in 3 2; do
  echo "== X3_NP=$np (trace build)"
  X3_TRACE=1 X3_NOCHECK=1 X3_NORES=1 X3_NP=$np timeout 60 tools/dev/x3_test_trace 3 2>&1 | grep "trace" | cut -c1-260
  echo "== X3_NP=$np"
  X3_NOCHECK=1 X3_NORES=1 X3_NP=$np timeout 60 tools/dev/x3_test 3 2>&1 | grep "time" | cut -c1-110
  X3_NOCHECK=1 X3_NP=$np timeout 60 tools/dev/x3_test 4 2>&1 | grep "time" | cut -c1-110
done
} | tee $O/${tag}_x3_times.txt
echo "== pytest subset"
timeout 1200 python -m pytest tests/test_gpu_parity.py -q -x -p no:cacheprovider -k "x3 or cascade_vs_reference_golden or costreg or conv3d_vs_oracle or deconv3d_vs_oracle or conv3d_golden" 2>&1 | tail -5 | tee $O/${tag}_pytest.log
echo "== bench"
timeout 600 python bench.py --steps 300 --warmup 10 --no-cpu-baseline --no-train-step 2>&1 | tail -1 | tee $O/${tag}_bench.json | cut -c1-330
python - <<PY
import json
d = json.load(open("gpurun_out/${tag}_bench.json"))
print(d["ms_per_step"], "roofline", d["roofline"]["frac"], d["roofline"]["per_stage_us"], "conv", d["roofline_conv"]["us_per_scene"], d["roofline_conv"]["largest_layers_us_tflops"])
PY
exit 0
