#!/bin/bash
# Round 6: FeatureNet's conv0.0 -> conv0.1 in one launch: tests, same-box A/B of bench.py (RCMVS_CONV_STEM=0 / 1), kernel time.
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -s -k "conv2d_stem or feature_net or cascade_vs_reference_golden or margin or first_layer" 2>&1 | grep "conv2d stem\|passed\|failed\|cascade_c2\|Error" | tee gpurun_out/r6_stem_tests.log
for v in 0 1 0 1; do
    RCMVS_CONV_STEM=$v timeout 600 python bench.py --steps 300 --warmup 20 --no-cpu-baseline --no-train-step --no-side-pass > gpurun_out/r6_stem_bench_$v.json 2>/dev/null
    python - <<PY
import json
b = json.load(open("gpurun_out/r6_stem_bench_$v.json"))
print("CONV_STEM=$v value", b["value"], "ms", b["ms_per_step"], "parity", b.get("parity"))
PY
done 2>&1 | tee -a gpurun_out/r6_stem_tests.log
R=$GRAFT_REPO_ROOT
( cd /tmp && export TMPDIR=/tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/stem_prof -o stem -- python $R/bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-train-step --no-side-pass > /dev/null 2>&1 )
f=$(find gpurun_out/stem_prof -name "*kernel_stats.csv" | head -1)
cp $f gpurun_out/r6_stem_kernel_stats.csv
grep "conv2d_stem_kernel\|conv2d_pair_kernel\|conv3d_z8" $f | cut -c1-60,150-300 | tee -a gpurun_out/r6_stem_tests.log
rm -rf gpurun_out/stem_prof
