#!/bin/bash
# Round 6 closing visit: all GPU tests, smoke(), the default bench line, rocprofv3 kernel stats of the same command, PMC passes over K1.
# usage: r6_final.sh <tag> [skip-tests]     -> gpurun_out/<tag>_*
set -u
exec < /dev/null
tag=${1:-r6final}
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
if [ "${2:-}" != "skip-tests" ]; then
  echo "== pytest -m gpu"
  timeout 1800 python -m pytest tests -q -m gpu -p no:cacheprovider 2>&1 | tail -8 | tee gpurun_out/${tag}_pytest.log
  echo "== smoke"
  timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2 | tee gpurun_out/${tag}_smoke.log
fi
echo "== bench (default)"
timeout 900 python bench.py 2>&1 | tail -1 > gpurun_out/${tag}_bench.json; cut -c1-300 gpurun_out/${tag}_bench.json
echo "== bench (the driver's form)"
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 --no-train-step 2>&1 | tail -1 > gpurun_out/${tag}_bench_driver_form.json; cut -c1-200 gpurun_out/${tag}_bench_driver_form.json
echo "== bench at the reference's other shapes"
for shape in dtu_eval tanks; do
  timeout 900 python bench.py --shape $shape 2>&1 | tail -1 > gpurun_out/${tag}_bench_$shape.json; cut -c1-200 gpurun_out/${tag}_bench_$shape.json
done
echo "== rocprof kernel stats"
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${tag}_prof -o ${tag} -- python $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-train-step --no-side-pass > $R/gpurun_out/${tag}_prof.log 2>&1 )
f=$(find gpurun_out/${tag}_prof -name "*kernel_stats.csv" 2>/dev/null | head -n 1)
if [ -n "$f" ] && [ -f "$f" ]; then cp "$f" gpurun_out/${tag}_kernel_stats.csv; head -n 16 "$f" | cut -c1-170; fi
find gpurun_out/${tag}_prof -name "*kernel_trace.csv" -delete 2>/dev/null
echo "== PMC passes over K1 in the pipeline (separate passes, --kernel-trace only)"
OUT=$R/gpurun_out/${tag}_pmc
rm -rf $OUT; mkdir -p $OUT
cd /tmp
for pass in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_VALU SQ_BUSY_CYCLES" "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR GRBM_GUI_ACTIVE"; do
  t=$(echo $pass | cut -d' ' -f1)
  timeout 300 rocprofv3 --kernel-trace --pmc $pass --output-format csv -d $OUT -o p_$t -- python $R/bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-train-step --no-side-pass > $OUT/log_$t.txt 2>&1
done
cd $R
python - <<PY | tee gpurun_out/${tag}_k1_pmc.json
import csv, glob, collections, json
agg = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(collections.Counter)
for f in sorted(glob.glob("gpurun_out/${tag}_pmc/*counter_collection.csv")):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "warp_variance" not in k and "conv11_prob" not in k: continue
        name = k.split("(")[0].replace("void rcmvs::", "").strip()
        agg[name][r["Counter_Name"]] += float(r["Counter_Value"]); n[name][r["Counter_Name"]] += 1
per = {}
for name in agg:
    d = {c: agg[name][c] / max(1, n[name][c]) for c in agg[name]}
    out = {"launches": n[name].get("FETCH_SIZE", 0), "fetch_bytes": 2.0 * 1024.0 * d.get("FETCH_SIZE", 0.0), "write_bytes": 1024.0 * d.get("WRITE_SIZE", 0.0)}
    for c, v in d.items():
        if c not in ("FETCH_SIZE", "WRITE_SIZE"): out[c] = v
    if d.get("SQ_WAVE_CYCLES"): out["valu_busy_of_wave_cycles"] = d.get("SQ_ACTIVE_INST_VALU", 0) / d["SQ_WAVE_CYCLES"]; out["wait_any_of_wave_cycles"] = d.get("SQ_WAIT_ANY", 0) / d["SQ_WAVE_CYCLES"]
    if d.get("SQ_LDS_ACTIVE"): out["lds_conflict_of_active"] = d.get("SQ_LDS_BANK_CONFLICT", 0) / d["SQ_LDS_ACTIVE"]
    per[name] = out
tot = sum(v["fetch_bytes"] + v["write_bytes"] for k, v in per.items() if "warp_variance" in k)
print(json.dumps({"source": "round 6 (tools/visits/r6_final.sh): rocprofv3 --kernel-trace --pmc <group>, separate passes over python bench.py --steps 6 --warmup 2 "
                            "(the K1 launches of the pipeline incl. the probe passes: both probability heads); FETCH_SIZE doubled (MI355X_MICROARCH: gfx950 reports half the "
                            "bytes of wide coalesced reads), counters in KB, averages per launch", "per_kernel": per, "bytes_per_scene": tot,
                  "algorithmic_bytes_per_scene": 457441280}, indent=1))
PY
rm -rf gpurun_out/${tag}_pmc gpurun_out/${tag}_prof        # (raw counter / trace files: tens of MB; the summaries above are what is kept)
exit 0
