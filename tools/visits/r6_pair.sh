#!/bin/bash
# Round 6: FeatureNet's conv1.1 -> conv1.2 in one launch: tests, same-box A/B of bench.py (RCMVS_CONV_PAIR=0 / 1), kernel time.
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -s -k "conv2d_pair or feature_net or cascade_vs_reference_golden or margin" 2>&1 | grep "conv2d pair\|passed\|failed\|cascade_c2" | tee gpurun_out/r6_pair_tests.log
for v in 0 1 0 1; do
    RCMVS_CONV_PAIR=$v timeout 600 python bench.py --steps 300 --warmup 20 --no-cpu-baseline --no-train-step --no-side-pass > gpurun_out/r6_pair_bench_$v.json 2>/dev/null
    python - <<PY
import json
b = json.load(open("gpurun_out/r6_pair_bench_$v.json"))
print("CONV_PAIR=$v value", b["value"], "ms", b["ms_per_step"])
PY
done
R=$GRAFT_REPO_ROOT
( cd /tmp && export TMPDIR=/tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/pair_prof -o pair -- python $R/bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-train-step --no-side-pass > /dev/null 2>&1 )
f=$(find gpurun_out/pair_prof -name "*kernel_stats.csv" | head -1)
grep "conv2d_pair_kernel\|conv3d_x3_kernel<16, 16, 3, 3>" $f | cut -c1-60,150-260
rm -rf gpurun_out/pair_prof
