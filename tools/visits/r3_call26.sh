#!/bin/bash
set -u
exec < /dev/null
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out
tag=${1:-r3c26}
mkdir -p $O
echo "== pytest subset"
timeout 1200 python -m pytest tests/test_gpu_parity.py -q -x -p no:cacheprovider -k "mfma or costreg or cascade or golden or conv3d_vs_oracle or deconv3d_vs_oracle" 2>&1 | tail -4 | tee $O/${tag}_pytest.log
echo "== bench"
timeout 600 python bench.py --steps 300 --warmup 10 --no-cpu-baseline --no-train-step 2>&1 | tail -1 | tee $O/${tag}_bench.json | cut -c1-330
echo "== rocprof (cascade)"
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/${tag}_prof -o ${tag} -- python $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-train-step > $GRAFT_REPO_ROOT/gpurun_out/${tag}_prof.log 2>&1 )
f=$(find gpurun_out/${tag}_prof -name "*kernel_stats.csv" 2>/dev/null | head -n 1)
if [ -n "$f" ] && [ -f "$f" ]; then cp "$f" gpurun_out/${tag}_kernel_stats.csv; python - <<PY
import csv
rows=list(csv.DictReader(open("gpurun_out/${tag}_kernel_stats.csv")))
print("total us/scene", sum(int(r['TotalDurationNs']) for r in rows)/23/1e3)
for r in rows[:${2:-30}]:
    print(f"{r['Name'][:100]:100s} {int(r['Calls'])/23:5.1f} {float(r['AverageNs'])/1e3:8.1f} {int(r['TotalDurationNs'])/23/1e3:7.1f}")
PY
fi
find gpurun_out/${tag}_prof -name "*kernel_trace.csv" -delete 2>/dev/null
exit 0
