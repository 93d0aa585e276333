#!/bin/bash
# Round 6: discriminators for the two-stream hazard (tools/dev/two_stream_diag.py) + the baseline bench line of the box.
mkdir -p gpurun_out
L=gpurun_out/r6_two_streams.log
: > $L
run() { echo "=== $*" >> $L; ( timeout 300 env "$@" python tools/dev/two_stream_diag.py ${MODE:-probe} 2>&1 | grep -v Warning | tail -60 ) >> $L; }
run X=1
MODE=spacer run X=1
run RCMVS_FP16_PAIR=0
run AMD_SERIALIZE_KERNEL=3
run GPU_MAX_HW_QUEUES=1
cat $L | cut -c1-400
timeout 600 python bench.py --steps 300 --warmup 20 > gpurun_out/r6_bench_base.json 2> gpurun_out/r6_bench_base.err
tail -c 400 gpurun_out/r6_bench_base.err
python - <<PY
import json
b = json.load(open("gpurun_out/r6_bench_base.json"))
print("value", b["value"], "ms", b["ms_per_step"], "roofline", b["roofline"]["frac"], b["roofline"]["per_stage_us"])
PY
