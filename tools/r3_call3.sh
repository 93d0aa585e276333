#!/bin/bash
# round 3, GPU visit 3: fp16-pair conv path (parity + A/B), K1 schedule variants
set -u
exec < /dev/null
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $O
echo "== pytest subset"
timeout 1200 python -m pytest tests/test_gpu_parity.py -q -x -p no:cacheprovider -s -k "x3 or variants_agree or cascade_vs_reference_golden or depth_head or costreg or conv3d_vs_oracle or deconv3d_vs_oracle" 2>&1 | grep -v "^$" | tail -45 | cut -c1-220 | tee $O/r3c3_pytest.log
echo "== K1 variants"
timeout 300 python tools/k1_ablate.py 0 4 5 6 0 2>&1 | tail -20 | tee $O/r3c3_k1_variants.txt
echo "== bench, bf16 triple"
RCMVS_FP16_PAIR=0 timeout 600 python bench.py --steps 300 --warmup 10 --no-cpu-baseline 2>&1 | tail -1 | tee $O/r3c3_bench_x3.json | cut -c1-330
echo "== bench, fp16 pair"
timeout 600 python bench.py --steps 300 --warmup 10 --no-cpu-baseline 2>&1 | tail -1 | tee $O/r3c3_bench_x3h.json | cut -c1-330
python - <<'PY'
import json
for n in ("x3", "x3h"):
    try:
        d = json.load(open(f"gpurun_out/r3c3_bench_{n}.json"))
        print(n, d["ms_per_step"], "roofline", d["roofline"]["frac"], d["roofline"]["per_stage_us"], "conv", d["roofline_conv"]["us_per_scene"], d["roofline_conv"]["largest_layers_us_tflops"], "parity", d.get("parity"))
    except Exception as e:
        print(n, "no json", e)
PY
echo "== rocprof"
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/r3c3_prof -o r3c3 -- python $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 3 --no-cpu-baseline > $O/r3c3_prof.log 2>&1 )
f=$(find $O/r3c3_prof -name "*kernel_stats.csv" 2>/dev/null | head -n 1)
if [ -n "$f" ] && [ -f "$f" ]; then cp "$f" $O/r3c3_kernel_stats.csv; head -n 40 "$f" | cut -c1-150; fi
find $O/r3c3_prof -name "*kernel_trace.csv" -delete 2>/dev/null
exit 0
