#!/bin/bash
# round 3, GPU visit 12: static VMEM counts per producer tick: trace, harness timings, parity, bench A/B
set -u
exec < /dev/null
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $O
{
for np in 3 2; do
  echo "== X3_NP=$np (trace build)"
  X3_TRACE=1 X3_NOCHECK=1 X3_NORES=1 X3_NP=$np timeout 60 tools/dev/x3_test_trace 3 2>&1 | grep "trace" | cut -c1-260
  echo "== X3_NP=$np"
  X3_NOCHECK=1 X3_NORES=1 X3_NP=$np timeout 60 tools/dev/x3_test 3 2>&1 | grep "time" | cut -c1-110
  X3_NOCHECK=1 X3_NP=$np timeout 60 tools/dev/x3_test 4 2>&1 | grep "time" | cut -c1-110
done
} | tee $O/r3c12_x3_times.txt
echo "== pytest subset"
timeout 1200 python -m pytest tests/test_gpu_parity.py -q -x -p no:cacheprovider -k "x3 or cascade_vs_reference_golden or costreg or conv3d_vs_oracle or deconv3d_vs_oracle or conv3d_golden" 2>&1 | tail -5 | tee $O/r3c12_pytest.log
echo "== bench, bf16 triple"
RCMVS_FP16_PAIR=0 timeout 600 python bench.py --steps 300 --warmup 10 --no-cpu-baseline 2>&1 | tail -1 | tee $O/r3c12_bench_x3.json | cut -c1-330
echo "== bench, fp16 pair"
timeout 600 python bench.py --steps 300 --warmup 10 --no-cpu-baseline 2>&1 | tail -1 | tee $O/r3c12_bench_x3h.json | cut -c1-330
python - <<'PY'
import json
for n in ("x3", "x3h"):
    try:
        d = json.load(open(f"gpurun_out/r3c12_bench_{n}.json"))
        print(n, d["ms_per_step"], "roofline", d["roofline"]["frac"], d["roofline"]["per_stage_us"], "conv", d["roofline_conv"]["us_per_scene"], d["roofline_conv"]["largest_layers_us_tflops"])
    except Exception as e:
        print(n, "no json", e)
PY
exit 0
