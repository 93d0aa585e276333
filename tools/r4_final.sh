#!/bin/bash
# round-4 full GPU visit: full parity suite, smoke, default bench line, training-step bench line, rocprof kernel stats
set -u
exec < /dev/null
tag=${1:-r4run1}
mkdir -p gpurun_out
export TMPDIR=/tmp
echo "== pytest -m gpu"
timeout 1800 python -m pytest tests -q -m gpu -p no:cacheprovider 2>&1 | tail -8 | tee gpurun_out/${tag}_pytest.log
echo "== smoke"
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2 | tee gpurun_out/${tag}_smoke.log
echo "== bench (default)"
timeout 900 python bench.py 2>&1 | tail -1 | tee gpurun_out/${tag}_bench.json | cut -c1-400
echo "== bench train_step"
timeout 600 python bench.py --workload train_step 2>&1 | tail -1 | tee gpurun_out/${tag}_bench_train_step.json | cut -c1-600
echo "== rocprof (cascade)"
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/${tag}_prof -o ${tag} -- python $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-train-step --no-side-pass > $GRAFT_REPO_ROOT/gpurun_out/${tag}_prof.log 2>&1 )
f=$(find gpurun_out/${tag}_prof -name "*kernel_stats.csv" 2>/dev/null | head -n 1)
if [ -n "$f" ] && [ -f "$f" ]; then cp "$f" gpurun_out/${tag}_kernel_stats.csv; head -n 14 "$f" | cut -c1-160; fi
find gpurun_out/${tag}_prof -name "*kernel_trace.csv" -delete 2>/dev/null
echo "== rocprof (train step)"
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/${tag}_prof_train -o ${tag}t -- python $GRAFT_REPO_ROOT/bench.py --workload train_step --steps 5 --warmup 2 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/${tag}_prof_train.log 2>&1 )
f=$(find gpurun_out/${tag}_prof_train -name "*kernel_stats.csv" 2>/dev/null | head -n 1)
if [ -n "$f" ] && [ -f "$f" ]; then cp "$f" gpurun_out/${tag}_train_kernel_stats.csv; head -n 8 "$f" | cut -c1-160; fi
find gpurun_out/${tag}_prof_train -name "*kernel_trace.csv" -delete 2>/dev/null
exit 0
