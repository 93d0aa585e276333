#!/bin/bash
# round 2 closing GPU visit: full parity suite, default bench line, training-step bench line, rocprof kernel stats
set -u
exec < /dev/null
tag=${1:-r2final}
mkdir -p gpurun_out
export TMPDIR=/tmp
echo "== pytest -m gpu"
timeout 1500 python -m pytest tests -q -m gpu -p no:cacheprovider 2>&1 | tail -8 | tee gpurun_out/${tag}_pytest.log
echo "== bench (default)"
timeout 600 python bench.py 2>&1 | tail -1 | tee gpurun_out/${tag}_bench.json | cut -c1-400
echo "== bench train_step"
timeout 600 python bench.py --workload train_step 2>&1 | tail -1 | tee gpurun_out/${tag}_bench_train_step.json | cut -c1-600
echo "== rocprof"
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/${tag}_prof -o ${tag} -- python $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 3 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/${tag}_prof.log 2>&1 )
f=$(find gpurun_out/${tag}_prof -name "*kernel_stats.csv" 2>/dev/null | head -n 1)
if [ -n "$f" ] && [ -f "$f" ]; then head -n 12 "$f" | cut -c1-160; fi
find gpurun_out/${tag}_prof -name "*kernel_trace.csv" -delete 2>/dev/null
exit 0
