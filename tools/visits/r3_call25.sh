#!/bin/bash
set -u
exec < /dev/null
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out
tag=${1:-r3c25}
mkdir -p $O
echo "== pytest train + render + parity subset"
timeout 1500 python -m pytest tests/test_gpu_train.py tests/test_gpu_render.py -q -x -p no:cacheprovider 2>&1 | tail -4 | tee $O/${tag}_pytest.log
echo "== bench train_step"
timeout 600 python bench.py --workload train_step 2>&1 | tail -1 | tee $O/${tag}_bench_train_step.json | cut -c1-400
echo "== rocprof (train step)"
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/${tag}_prof_train -o ${tag}t -- python $GRAFT_REPO_ROOT/bench.py --workload train_step --steps 5 --warmup 2 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/${tag}_prof_train.log 2>&1 )
f=$(find gpurun_out/${tag}_prof_train -name "*kernel_stats.csv" 2>/dev/null | head -n 1)
if [ -n "$f" ] && [ -f "$f" ]; then cp "$f" gpurun_out/${tag}_train_kernel_stats.csv; python - <<PY
import csv
rows=list(csv.DictReader(open("gpurun_out/${tag}_train_kernel_stats.csv")))
it=7
print("kernel ms/iter", sum(int(r['TotalDurationNs']) for r in rows)/it/1e6, "launches/iter", sum(int(r['Calls']) for r in rows)/it)
for key in ("pack","Fill","bn_","wgrad","warp_variance_bwd","copyBuffer","direct_copy","CUDAFunctor_add"):
    sel=[r for r in rows if key in r['Name']]
    print(f"  {key:20s} {sum(int(r['Calls']) for r in sel)/it:8.1f} launches {sum(int(r['TotalDurationNs']) for r in sel)/it/1e6:7.2f} ms")
PY
fi
find gpurun_out/${tag}_prof_train -name "*kernel_trace.csv" -delete 2>/dev/null
exit 0
