#!/bin/bash
# PMC counters for the K1 variants (separate passes; kernel-trace only, no other trace domains).
# K1_VARIANTS="4 6" bash tools/k1_pmc.sh  profiles the LDS-staged variants (e.g. SQ_LDS_BANK_CONFLICT vs SQ_INSTS_LDS); default: production.
set -u
exec < /dev/null
export TMPDIR=/tmp
mkdir -p gpurun_out/pmc
cd /tmp
for pass in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_VALU" \
            "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAVES SQ_BUSY_CYCLES SQ_INST_CYCLES_VMEM SQ_LDS_BANK_CONFLICT" \
            "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum" \
            "FETCH_SIZE" "WRITE_SIZE" ; do
  tag=$(echo $pass | cut -d' ' -f1)
  timeout 300 rocprofv3 --kernel-trace --pmc $pass --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pmc -o k1_$tag -- python $GRAFT_REPO_ROOT/tools/k1_ablate.py ${K1_VARIANTS:-0} > $GRAFT_REPO_ROOT/gpurun_out/pmc/log_$tag.txt 2>&1
done
cd $GRAFT_REPO_ROOT
python - <<'PY'
import csv, glob, collections
for f in sorted(glob.glob("gpurun_out/pmc/*counter_collection.csv")):
    agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "warp_variance" not in k: continue
        name = k.split("(")[0].replace("void rcmvs::","")
        agg[name][r["Counter_Name"]] += float(r["Counter_Value"]); 
    print("==", f.split("/")[-1])
    for name in sorted(agg):
        print(name, {c: f"{v:.3g}" for c, v in agg[name].items()})
PY
