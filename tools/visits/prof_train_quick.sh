#!/bin/bash
# one short rocprofv3 kernel-trace pass over the training-step bench + the bench line itself: bash tools/visits/prof_train_quick.sh <tag> [pytest -k expression]
set -u
exec < /dev/null
tag=${1:-tquick}
export TMPDIR=/tmp
mkdir -p gpurun_out
if [ -n "${2:-}" ]; then timeout 900 python -m pytest tests -q -m gpu -p no:cacheprovider -k "$2" 2>&1 | tail -4; fi
timeout 600 python bench.py --workload train_step --no-cpu-baseline 2>&1 | tail -1 | tee gpurun_out/${tag}_bench_train_step.json | cut -c1-330
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/${tag}_prof -o ${tag} -- python $GRAFT_REPO_ROOT/bench.py --workload train_step --steps 5 --warmup 2 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/${tag}_prof.log 2>&1 )
f=$(find gpurun_out/${tag}_prof -name "*kernel_stats.csv" 2>/dev/null | head -n 1)
if [ -n "$f" ] && [ -f "$f" ]; then cp "$f" gpurun_out/${tag}_train_kernel_stats.csv; fi
find gpurun_out/${tag}_prof -name "*kernel_trace.csv" -delete 2>/dev/null
python - <<P
import csv
rows=list(csv.DictReader(open('gpurun_out/${tag}_train_kernel_stats.csv')))
tot=sum(float(r['TotalDurationNs']) for r in rows)
print('kernel time total ms', tot/1e6, 'launches', sum(int(r['Calls']) for r in rows))
for r in rows[:40]:
    n=int(r['Calls']); t=float(r['TotalDurationNs'])
    print(f"{r['Name'][:100]:100s} {n:6d} {t/n/1e3:8.1f} us {100*t/tot:5.1f} %")
P
exit 0
