#!/bin/bash
# rocprofv3 kernel stats of the default bench for each library given ("product" or a path): per-kernel average us, filtered by a regex ($1)
mkdir -p gpurun_out
pat=$1; shift
R=$GRAFT_REPO_ROOT
for lib in "$@"; do
  ( cd /tmp && export TMPDIR=/tmp && ( [ $lib != product ] && export RCMVS_LIB=$R/$lib; timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/ks_prof -o ks -- python $R/bench.py --steps 100 --warmup 5 --no-cpu-baseline --no-train-step --no-side-pass > /dev/null 2>&1 ) )
  f=$(find gpurun_out/ks_prof -name "*kernel_stats.csv" | head -1)
  echo "== $lib"
  python - "$f" "$pat" <<'PY'
import csv, sys, re
rows = list(csv.DictReader(open(sys.argv[1])))
tot = 0.0
for r in rows:
    n = int(r['Calls']); t = float(r['TotalDurationNs'])
    if n >= 100: tot += t / 105.0
    if re.search(sys.argv[2], r['Name']):
        print(f"  {re.sub(r'[(].*', '', r['Name'])[:70]:70s} calls {n:6d} avg {t / n / 1e3:7.1f} us  per scene {t / 105e3:7.1f}")
print(f"  kernels per scene (105 scenes): {tot / 1e3:.1f} us")
PY
  cp $f gpurun_out/r6_kstats_$(basename $lib .so).csv
  rm -rf gpurun_out/ks_prof
done
