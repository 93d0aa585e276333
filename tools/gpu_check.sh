#!/bin/bash
# One GPU-box visit: parity tests, smoke, bench, rocprofv3 kernel stats.  Logs -> gpurun_out/.
set -u
exec < /dev/null          # nothing below may wait on stdin (an empty file list once turned `head` into a 7-minute hang)
mkdir -p gpurun_out
export TMPDIR=/tmp
echo "== rocm-smi" ; rocm-smi --showproductname 2>/dev/null | head -8
python - <<PY
import os
print("cpu_count", os.cpu_count(), "affinity", len(os.sched_getaffinity(0)))
try: print("cpu.max", open("/sys/fs/cgroup/cpu.max").read().strip())
except Exception as e: print("no cpu.max", e)
PY
echo "== pytest -m gpu"
timeout 900 python -m pytest tests -q -m gpu -s -p no:cacheprovider 2>&1 | tail -60 | tee gpurun_out/pytest_gpu.log
echo "== smoke"
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -5 | tee gpurun_out/smoke.log
echo "== bench"
timeout 600 python bench.py --steps 30 --warmup 5 2>&1 | tail -3 | tee gpurun_out/bench.log
echo "== train bench"
timeout 300 python tools/train_bench.py 2>&1 | tail -4 | tee gpurun_out/train_bench.log
echo "== rocprof"
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof -o r1 -- python $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 3 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/prof_bench.log 2>&1 )
ls -R gpurun_out/prof 2>/dev/null | head -n 20
f=$(find gpurun_out/prof -name "*kernel_stats.csv" 2>/dev/null | head -n 1)
if [ -n "$f" ] && [ -f "$f" ]; then head -n 40 "$f"; fi
find gpurun_out/prof -name "*kernel_trace.csv" -delete 2>/dev/null
exit 0
