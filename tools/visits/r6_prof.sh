#!/bin/bash
# Round 6: rocprofv3 kernel stats of the default bench command (short), top kernels printed.  usage: r6_prof.sh <tag> [env assignments]
mkdir -p gpurun_out
tag=${1:-a}; shift
R=$GRAFT_REPO_ROOT
( cd /tmp && export TMPDIR=/tmp && timeout 600 env "$@" rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${tag}_prof -o ${tag} -- python $R/bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-train-step --no-side-pass > $R/gpurun_out/${tag}_prof.log 2>&1 )
f=$(find gpurun_out/${tag}_prof -name "*kernel_stats.csv" | head -1)
cp $f gpurun_out/${tag}_kernel_stats.csv
python - <<PY
import csv
rows = list(csv.DictReader(open("gpurun_out/${tag}_kernel_stats.csv")))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
scenes = 110
print(f"kernel time per scene: {tot / scenes / 1e3:.1f} us")
for r in rows[:36]:
    print(f'{r["Name"][:84]:84s} calls {r["Calls"]:>5s} avg {float(r["AverageNs"])/1e3:8.1f} us  {100*float(r["TotalDurationNs"])/tot:5.2f} %  {float(r["TotalDurationNs"])/scenes/1e3:7.1f} us/scene')
PY
rm -rf gpurun_out/${tag}_prof
