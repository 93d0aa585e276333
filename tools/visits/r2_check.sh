#!/bin/bash
# round 2 GPU visit: parity tests, bench, optional rocprof.  usage: tools/r2_check.sh [tag] [prof]
set -u
exec < /dev/null
tag=${1:-r2}
mkdir -p gpurun_out
export TMPDIR=/tmp
echo "== pytest -m gpu"
timeout 1200 python -m pytest tests -q -m gpu -x -p no:cacheprovider 2>&1 | tail -15 | tee gpurun_out/${tag}_pytest.log
echo "== bench"
timeout 600 python bench.py --steps 30 --warmup 5 2>&1 | tail -2 | tee gpurun_out/${tag}_bench.json
if [ "${2:-}" = "prof" ]; then
  echo "== rocprof"
  ( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/${tag}_prof -o ${tag} -- python $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 3 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/${tag}_prof.log 2>&1 )
  f=$(find gpurun_out/${tag}_prof -name "*kernel_stats.csv" 2>/dev/null | head -n 1)
  if [ -n "$f" ] && [ -f "$f" ]; then head -n 45 "$f" | cut -c1-200; fi
  find gpurun_out/${tag}_prof -name "*kernel_trace.csv" -delete 2>/dev/null
fi
exit 0
