#!/bin/bash
# PMC passes over the marching prob conv (tools/dev/prob_ab.py, production dispatch only): wave occupancy, wait states, VALU / SALU / SMEM / LDS activity
set -u
exec < /dev/null
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/probpmc
mkdir -p $OUT
cd /tmp
for pass in "GRBM_GUI_ACTIVE SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_BUSY_CU_CYCLES" \
            "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_WAVE_CYCLES" \
            "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_SALU SQ_INST_CYCLES_SMEM SQ_WAVE_CYCLES" \
            "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM" \
            "SQ_WAIT_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_THREAD_CYCLES_VALU SQ_IFETCH SQ_WAVE_CYCLES"; do
  tag=$(echo $pass | cut -d' ' -f1-2 | tr ' ' '_')
  PROB_ONLY_AUTO=1 timeout 200 rocprofv3 --kernel-trace --pmc $pass --output-format csv -d $OUT -o p_$tag -- python $GRAFT_REPO_ROOT/tools/dev/prob_ab.py > $OUT/log_$tag.txt 2>&1
  tail -1 $OUT/log_$tag.txt | cut -c1-200
done
cd $GRAFT_REPO_ROOT
python - <<'PY'
import csv, glob, collections
for f in sorted(glob.glob("gpurun_out/probpmc/*counter_collection.csv")):
    agg = collections.defaultdict(lambda: collections.defaultdict(float)); calls = collections.Counter()
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "prob_conv_march" not in k: continue
        key = "grid %s" % r.get("Grid_Size", "?")
        agg[key][r["Counter_Name"]] += float(r["Counter_Value"]); calls[key] += 1
    print("==", f.split("/")[-1])
    for name, c in agg.items():
        print(f"{name:20s} " + " ".join(f"{k}={v:.4g}" for k, v in sorted(c.items())))
PY
find gpurun_out/probpmc -name "*kernel_trace.csv" -delete 2>/dev/null
exit 0
