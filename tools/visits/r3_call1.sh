#!/bin/bash
# round 3, GPU visit 1: (1) LDS-DMA semantics probe, (2) K1 vector-L1 / texture-addresser counters on the production kernel,
# (3) baseline bench line on this box, (4) hunt for the stray device copies of the forward.  Logs -> gpurun_out/r3c1_*.
set -u
exec < /dev/null
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $O
echo "== dma probe"
timeout 120 tools/dev/dma_probe 2>&1 | tee $O/r3c1_dma_probe.txt
echo "== counters available"
( cd /tmp && timeout 120 rocprofv3 -L > $O/r3c1_counters_all.txt 2>&1 )
grep -o "\(TCP\|TA\|TD\|GRBM\|TCC\)_[A-Za-z0-9_]*" $O/r3c1_counters_all.txt | sort -u | tr '\n' ' ' | cut -c1-6000
echo
echo "== K1 pmc"
mkdir -p $O/r3c1_pmc
cd /tmp
for pass in "GRBM_GUI_ACTIVE GRBM_TA_BUSY TCP_GATE_EN1_sum TCP_GATE_EN2_sum" \
            "GRBM_GUI_ACTIVE TCP_TD_TCP_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_PENDING_STALL_CYCLES_sum" \
            "GRBM_GUI_ACTIVE TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_TA_TCP_STATE_READ_sum TCP_TOTAL_ACCESSES_sum" \
            "GRBM_GUI_ACTIVE TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum" \
            "GRBM_GUI_ACTIVE TA_BUFFER_LOAD_WAVEFRONTS_sum TA_BUFFER_WAVEFRONTS_sum TD_TD_BUSY_sum TD_TC_STALL_sum" \
            "GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM" \
            "GRBM_GUI_ACTIVE TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_LATENCY_sum" \
            "FETCH_SIZE" "WRITE_SIZE"; do
  tag=$(echo $pass | tr ' ' '_' | cut -c1-60)
  timeout 200 rocprofv3 --kernel-trace --pmc $pass --output-format csv -d $O/r3c1_pmc -o k1_$tag -- python $GRAFT_REPO_ROOT/tools/k1_ablate.py 0 > $O/r3c1_pmc/log_$tag.txt 2>&1
  tail -2 $O/r3c1_pmc/log_$tag.txt | cut -c1-300
done
cd $GRAFT_REPO_ROOT
python - <<'PY'
import csv, glob, collections
for f in sorted(glob.glob("gpurun_out/r3c1_pmc/*counter_collection.csv")):
    agg = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "warp_variance" not in k: continue
        name = k.split("(")[0].replace("void rcmvs::", "")
        agg[name][r["Counter_Name"]] += float(r["Counter_Value"])
    print("==", f.split("/")[-1])
    for name in sorted(agg):
        print(" ", name, {c: f"{v:.4g}" for c, v in agg[name].items()})
PY
find gpurun_out/r3c1_pmc -name "*kernel_trace.csv" -delete 2>/dev/null
echo "== bench"
timeout 600 python bench.py --steps 200 --warmup 10 --no-cpu-baseline 2>&1 | tail -1 | tee $O/r3c1_bench.json | cut -c1-700
echo "== copy hunt"
timeout 300 python tools/dev/copy_hunt.py 2>&1 | tail -60 | tee $O/r3c1_copy_hunt.txt | cut -c1-260
exit 0
