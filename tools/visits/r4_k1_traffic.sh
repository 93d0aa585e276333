#!/bin/bash
# Round 4: HBM traffic of K1 (fused warp + variance) IN THE PIPELINE, from rocprofv3 PMC passes over `python bench.py --steps 6` (separate
# --pmc passes, --kernel-trace only; FETCH_SIZE doubled: MI355X_MICROARCH "HBM", gfx950 tallies 128-byte requests at 64 B; counters in KB).
set -u
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/k1traffic
rm -rf $OUT; mkdir -p $OUT
cd /tmp
for pass in FETCH_SIZE WRITE_SIZE; do
  timeout 300 rocprofv3 --kernel-trace --pmc $pass --output-format csv -d $OUT -o p_$pass -- python $GRAFT_REPO_ROOT/bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-train-step --no-side-pass > $OUT/log_$pass.txt 2>&1
done
cd $GRAFT_REPO_ROOT
python - <<'PY' | tee gpurun_out/r4_k1_traffic.json
import csv, glob, collections, json
agg = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(collections.Counter)
for f in sorted(glob.glob("gpurun_out/k1traffic/*counter_collection.csv")):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "warp_variance_tp_kernel" not in k: continue
        C = k.split("<")[1].split(",")[0].strip()
        agg[C][r["Counter_Name"]] += float(r["Counter_Value"]); n[C][r["Counter_Name"]] += 1
per = {}
for C in agg:
    per[C] = {"launches": n[C]["FETCH_SIZE"], "fetch_bytes": 2.0 * 1024.0 * agg[C]["FETCH_SIZE"] / max(1, n[C]["FETCH_SIZE"]),
              "write_bytes": 1024.0 * agg[C]["WRITE_SIZE"] / max(1, n[C]["WRITE_SIZE"])}
tot = sum(v["fetch_bytes"] + v["write_bytes"] for v in per.values())
print(json.dumps({"source": "round 4 (tools/visits/r4_k1_traffic.sh): rocprofv3 --kernel-trace --pmc FETCH_SIZE and --pmc WRITE_SIZE, separate passes over "
                            "`python bench.py --steps 6 --warmup 2` (the K1 launches of the pipeline, real plane tables); FETCH_SIZE doubled (MI355X_MICROARCH: gfx950 reports "
                            "half the bytes of wide coalesced reads), counters in KB", "per_stage_C": per, "bytes_per_scene": tot, "algorithmic_bytes_per_scene": 457441280}, indent=1))
PY
