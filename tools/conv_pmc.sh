#!/bin/bash
# MFMA / VALU utilisation counters for the kernels of one bench.py run (separate --pmc passes, kernel-trace only).
set -u
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/convpmc
mkdir -p $OUT
cd /tmp
rocprofv3 -L 2>/dev/null | grep -o "SQ_[A-Z_0-9]*MFMA[A-Z_0-9]*" | sort -u > $OUT/mfma_counters.txt
cat $OUT/mfma_counters.txt | tr '\n' ' '; echo
for pass in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_MFMA SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES GRBM_GUI_ACTIVE" ; do
  tag=$(echo $pass | cut -d' ' -f1)
  timeout 400 rocprofv3 --kernel-trace --pmc $pass --output-format csv -d $OUT -o conv_$tag -- python $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 > $OUT/log_$tag.txt 2>&1
  tail -2 $OUT/log_$tag.txt | cut -c1-300
done
cd $GRAFT_REPO_ROOT
python - <<'PY'
import csv, glob, collections
for f in sorted(glob.glob("gpurun_out/convpmc/*counter_collection.csv")):
    agg = collections.defaultdict(lambda: collections.defaultdict(float)); calls = collections.Counter()
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "rcmvs" not in k: continue
        name = k.split("(")[0].replace("void rcmvs::", "")
        agg[name][r["Counter_Name"]] += float(r["Counter_Value"])
    print("==", f.split("/")[-1])
    rows = []
    for name, c in agg.items():
        busy = c.get("SQ_BUSY_CU_CYCLES", 0.0)
        mf = c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0)
        rows.append((mf, name, c))
    for mf, name, c in sorted(rows, reverse=True):
        busy = c.get("SQ_BUSY_CU_CYCLES", 0.0); gui = c.get("GRBM_GUI_ACTIVE", 0.0)
        util_busy = 100.0 * mf / busy if busy else float("nan")          # MFMA-busy cycles per busy-CU cycle
        util_gui = 100.0 * mf / (gui * 256 * 4) if gui else float("nan")  # gfx94x MfmaUtil formula, 256 CUs x 4 SIMDs
        print(f"{name[:70]:70s} mfma_busy/cu_busy {util_busy:6.1f}%  MfmaUtil(gui) {util_gui:6.1f}%  " +
              " ".join(f"{k}={v:.3g}" for k, v in sorted(c.items())))
PY
