#!/bin/bash
# SECOND prepared visit of the next round: the two checks that follow from "the stale sector sat in the reading CU's vector L1"
# (DESIGN.md section 8).  In the build container first:  bash tools/dev/build_plain_planes_variant.sh
# Expected if that reading is right: product 0, acq 0, l1acq_plain 0, plain > 0, sc0 > 0 corrupted rounds; the last lines price the general
# protection (a one-lane acquire at the top of every inference kernel) on the one-stream and the two-scenes-in-flight throughput.
mkdir -p gpurun_out; L=gpurun_out/r4_second.log; : > $L
run() { env "$@" timeout 100 python tools/dev/two_stream_depth.py 2>&1 | grep -v "Warning\|amdgpu.ids" | tail -1 | cut -c1-220 >> $L; }
for i in 1 2; do
  run SCENES=40 ROUNDS=8 RCMVS_LIB=tools/dev/_variants/lib_plain.so
  run SCENES=40 ROUNDS=8 RCMVS_LIB=tools/dev/_variants/lib_acq.so
  run SCENES=40 ROUNDS=8 RCMVS_LIB=tools/dev/_variants/lib_sc0.so
  run SCENES=40 ROUNDS=8 RCMVS_LIB=tools/dev/_variants/lib_l1acq_plain.so
  run SCENES=40 ROUNDS=8
done
echo "== reduced probe (which concurrent work is needed), pre-fix library" >> $L
RCMVS_LIB=tools/dev/_variants/lib_plain.so timeout 200 python tools/dev/two_stream_minimal.py 2>&1 | grep -v "Warning\|amdgpu.ids" >> $L
echo "== cost of the general protection: bench.py with the product library, then with lib_l1acq.so in its place" >> $L
line() { timeout 300 python bench.py --steps 300 --warmup 10 --no-cpu-baseline --no-train-step 2>/dev/null | tail -1 | python -c "import json,sys; b=json.load(sys.stdin); print(b['value'], b['ms_per_step'], b['two_scenes_in_flight'])" >> $L; }
line
cp rc_mvsnet_amd/librcmvs_hip.so /tmp/product.so && cp tools/dev/_variants/lib_l1acq.so rc_mvsnet_amd/librcmvs_hip.so && line
cp /tmp/product.so rc_mvsnet_amd/librcmvs_hip.so
cat $L
