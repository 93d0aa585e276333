#!/bin/bash
# Round 4, the ONE A/B visit the round-3 review allowed the two-stream hazard (<= 10 GPU-minutes): does a one-lane agent-scope acquire at
# the top of every inference kernel (-DRCMVS_L1_ACQUIRE, csrc/common.h) alone -- planes kernel back on PLAIN loads -- give 0 corrupted scenes
# in >= 5000 with two scenes in flight, and what does it cost?   In the build container first:  bash tools/dev/build_plain_planes_variant.sh
mkdir -p gpurun_out; L=gpurun_out/r4_second.log; : > $L
run() { echo "-- $*" >> $L; env "$@" timeout 170 python tools/dev/two_stream_depth.py 2>&1 | grep -v "Warning\|amdgpu.ids" | tail -1 | cut -c1-300 >> $L; }
run SCENES=40 ROUNDS=8 RCMVS_LIB=tools/dev/_variants/lib_plain.so                 # positive control: the pre-fix library must still corrupt
run SCENES=90 ROUNDS=60 RCMVS_LIB=tools/dev/_variants/lib_l1acq_plain.so          # 5400 scenes on the candidate
run SCENES=90 ROUNDS=30 RCMVS_LIB=tools/dev/_variants/lib_l1acq_plain.so          # + 2700 (a second process)
echo "== cost: bench.py with the product library, then with lib_l1acq.so in its place (value, ms_per_step)" >> $L
line() { timeout 200 python bench.py --steps 400 --warmup 10 --no-cpu-baseline --no-train-step --no-side-pass 2>/dev/null | tail -1 | python -c "import json,sys; b=json.load(sys.stdin); print(b['value'], b['ms_per_step'])" >> $L; }
line
cp rc_mvsnet_amd/librcmvs_hip.so /tmp/product.so && cp tools/dev/_variants/lib_l1acq.so rc_mvsnet_amd/librcmvs_hip.so && line && line
cp /tmp/product.so rc_mvsnet_amd/librcmvs_hip.so
line
cat $L
