#!/bin/bash
# PMC passes over the standalone x3 conv harness (tools/dev/x3_test 3): matrix-pipe busy, wave wait states, LDS
set -u
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/x3pmc
mkdir -p $OUT
cd /tmp
for pass in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_MFMA SQ_WAVE_CYCLES GRBM_GUI_ACTIVE" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_WAVE_CYCLES" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD"; do
  tag=$(echo $pass | cut -d' ' -f1)
  X3_NORES=1 timeout 120 rocprofv3 --kernel-trace --pmc $pass --output-format csv -d $OUT -o x3_$tag -- $GRAFT_REPO_ROOT/tools/dev/x3_test 3 > $OUT/log_$tag.txt 2>&1
done
cd $GRAFT_REPO_ROOT
python - <<'PY'
import csv, glob, collections
for f in sorted(glob.glob("gpurun_out/x3pmc/*counter_collection.csv")):
    agg = collections.defaultdict(lambda: collections.defaultdict(float)); calls = collections.Counter()
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "conv3d_x3_kernel" not in k: continue
        name = k.split("(")[0].replace("void rcmvs::", "")
        agg[name][r["Counter_Name"]] += float(r["Counter_Value"])
    print("==", f.split("/")[-1])
    for name, c in agg.items():
        print(f"{name:40s} " + " ".join(f"{k}={v:.4g}" for k, v in sorted(c.items())))
PY
