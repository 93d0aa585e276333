#!/bin/bash
# one short rocprofv3 kernel-trace pass over the default bench (per-kernel durations of the shipped pipeline): bash tools/visits/prof_quick.sh <tag>
set -u
exec < /dev/null
tag=${1:-quick}
export TMPDIR=/tmp
mkdir -p gpurun_out
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/${tag}_prof -o ${tag} -- python $GRAFT_REPO_ROOT/bench.py --steps 100 --warmup 5 --no-cpu-baseline --no-train-step --no-side-pass > $GRAFT_REPO_ROOT/gpurun_out/${tag}_prof.log 2>&1 )
f=$(find gpurun_out/${tag}_prof -name "*kernel_stats.csv" 2>/dev/null | head -n 1)
if [ -n "$f" ] && [ -f "$f" ]; then cp "$f" gpurun_out/${tag}_kernel_stats.csv; fi
t=$(find gpurun_out/${tag}_prof -name "*kernel_trace.csv" 2>/dev/null | head -n 1)
python - "$t" "${2:-deep}" <<P
import csv, sys, collections
# per (kernel, grid) average duration of the kernels whose name contains argv[2]
acc = collections.defaultdict(lambda: [0, 0.0])
for r in csv.DictReader(open(sys.argv[1])):
    if sys.argv[2] in r['Kernel_Name']:
        k = (r['Kernel_Name'][:60], r.get('Grid_Size_X', r.get('Grid_Size', '?')))
        acc[k][0] += 1; acc[k][1] += float(r['End_Timestamp']) - float(r['Start_Timestamp'])
for k, (n, t) in sorted(acc.items()):
    print(f"{k[0]:60s} grid {k[1]:>8s} {n:6d} {t / n / 1e3:8.1f} us")
P
find gpurun_out/${tag}_prof -name "*kernel_trace.csv" -delete 2>/dev/null
tail -1 gpurun_out/${tag}_prof.log | cut -c1-200
python - <<P
import csv
rows=list(csv.DictReader(open('gpurun_out/${tag}_kernel_stats.csv')))
for r in rows[:45]:
    n=int(r['Calls']); t=float(r['TotalDurationNs'])
    print(f"{r['Name'][:100]:100s} {n:6d} {t/n/1e3:8.1f} us")
P
exit 0
