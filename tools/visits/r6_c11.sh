#!/bin/bash
# Round 6: the fused conv11 + prob pass: GPU tests, same-box A/B of bench.py (RCMVS_CONV11_PROB=0 / 1), kernel stats of the fused run.  usage: r6_c11.sh <tag>
mkdir -p gpurun_out
tag=${1:-a}
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -s -k "conv11_prob or cascade_vs_reference_golden or margin or depth_head" 2>&1 | grep -v "^$" | tail -30 | tee gpurun_out/r6_c11_tests_$tag.log
for v in 0 1 0 1; do
    RCMVS_CONV11_PROB=$v timeout 600 python bench.py --steps 300 --warmup 20 --no-cpu-baseline --no-train-step --no-side-pass > gpurun_out/r6_c11_bench_${v}_$tag.json 2> gpurun_out/r6_c11_bench_$tag.err
    python - <<PY
import json
b = json.load(open("gpurun_out/r6_c11_bench_${v}_$tag.json"))
print("CONV11_PROB=$v value", b["value"], "ms", b["ms_per_step"], "K1", b["roofline"]["per_stage_us"], "conv us", b["roofline_conv"]["us_per_scene"])
PY
done
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/r6_c11_prof_$tag -o c11 -- python $GRAFT_REPO_ROOT/bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-train-step --no-side-pass > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
f=$(find gpurun_out/r6_c11_prof_$tag -name "*kernel_stats.csv" | head -1)
python - <<PY
import csv
rows = list(csv.DictReader(open("$f")))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
for r in rows[:40]:
    n = r["Name"]
    print(f'{n[:70]:70s} calls {r["Calls"]:>5s} avg {float(r["AverageNs"])/1e3:8.1f} us  {100*float(r["TotalDurationNs"])/tot:5.2f} %')
PY
cp $f gpurun_out/r6_c11_kernel_stats_$tag.csv
