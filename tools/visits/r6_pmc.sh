#!/bin/bash
# Round 6 (the round-4 script, all rcmvs kernels of the closing kernel set): rocprofv3 PMC passes (one counter group per pass, --kernel-trace only) over the config-2 scene AS SHIPPED -- the CostRegNet layers
# on the fp16-pair form of the matrix-core kernels -- for the bench line's `roofline_conv` (matrix-pipe busy, LDS bank conflicts, VALU).
set -u
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/r6pmc
rm -rf $OUT; mkdir -p $OUT
cd /tmp
for pass in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_MFMA SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES GRBM_GUI_ACTIVE" \
            "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_WAVE_CYCLES GRBM_GUI_ACTIVE" \
            "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD GRBM_GUI_ACTIVE"; do
  tag=$(echo $pass | cut -d' ' -f1)
  timeout 300 rocprofv3 --kernel-trace --pmc $pass --output-format csv -d $OUT -o p_$tag -- python $GRAFT_REPO_ROOT/bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-train-step --no-side-pass > $OUT/log_$tag.txt 2>&1
  tail -1 $OUT/log_$tag.txt | cut -c1-120
done
cd $GRAFT_REPO_ROOT
python - <<'PY' | tee gpurun_out/r6_pmc.txt
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(float)); calls = collections.defaultdict(collections.Counter)
for f in sorted(glob.glob("gpurun_out/r6pmc/*counter_collection.csv")):
    seen = set()
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "rcmvs" not in k: continue
        name = k.split("(")[0].replace("void rcmvs::", "")
        agg[name][r["Counter_Name"]] += float(r["Counter_Value"])
        calls[name][r["Counter_Name"]] += 1
rows = []
for name, c in agg.items():
    gui = c.get("GRBM_GUI_ACTIVE", 0.0) / max(1, sum(1 for k in ("SQ_VALU_MFMA_BUSY_CYCLES", "SQ_WAIT_ANY", "SQ_LDS_BANK_CONFLICT") if k in c))   # counted once per pass
    rows.append((gui, name, c))
print("# kernel | launches | GRBM_GUI_ACTIVE per launch / 8 XCDs (cycles) | matrix pipe busy = SQ_VALU_MFMA_BUSY_CYCLES / (cycles x 1024 SIMDs) | VALU busy = SQ_ACTIVE_INST_VALU x 4 / (cycles x 1024)"
      " | waves waiting SQ_WAIT_ANY / SQ_WAVE_CYCLES | issue stalls SQ_WAIT_INST_ANY / SQ_WAVE_CYCLES | LDS conflict cycles / LDS active | INSTS per launch: MFMA VALU SALU LDS VMEM_RD")
for gui, name, c in sorted(rows, reverse=True)[:40]:
    n = calls[name].get("SQ_VALU_MFMA_BUSY_CYCLES") or calls[name].get("GRBM_GUI_ACTIVE", 1)
    cyc = gui / 8.0
    f = lambda a, b: (a / b) if b else float("nan")
    print(f"{name[:58]:58s} n={n:4d} cyc/launch={cyc / max(n, 1):9.0f}  mfma_busy={100 * f(c.get('SQ_VALU_MFMA_BUSY_CYCLES', 0), cyc * 1024):5.1f}%  valu_busy={100 * f(4 * c.get('SQ_ACTIVE_INST_VALU', 0), cyc * 1024):5.1f}%  "
          f"wait_any={100 * f(c.get('SQ_WAIT_ANY', 0), c.get('SQ_WAVE_CYCLES', 0) / 2):5.1f}%  wait_inst={100 * f(c.get('SQ_WAIT_INST_ANY', 0), c.get('SQ_WAVE_CYCLES', 0) / 2):5.1f}%  "
          f"lds_conflict={100 * f(c.get('SQ_LDS_BANK_CONFLICT', 0), c.get('SQ_LDS_IDX_ACTIVE', 0)):5.1f}%  "
          f"insts: mfma={f(c.get('SQ_INSTS_MFMA', 0), n):.3g} valu={f(c.get('SQ_INSTS_VALU', 0), n):.3g} salu={f(c.get('SQ_INSTS_SALU', 0), n):.3g} lds={f(c.get('SQ_INSTS_LDS', 0), n):.3g} vmem_rd={f(c.get('SQ_INSTS_VMEM_RD', 0), n):.3g}")
PY
rm -rf gpurun_out/r6pmc
