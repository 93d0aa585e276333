#!/bin/bash
# SECOND prepared visit of the next round: the two checks that follow from "the stale sector sat in the reading CU's vector L1"
# (DESIGN.md section 8).  In the build container first:  bash tools/dev/build_plain_planes_variant.sh
# Expected if that reading is right: product 0, acq 0, plain > 0, sc0 > 0 corrupted rounds.
mkdir -p gpurun_out; L=gpurun_out/r4_second.log; : > $L
run() { env "$@" timeout 100 python tools/dev/two_stream_depth.py 2>&1 | grep -v "Warning\|amdgpu.ids" | tail -1 | cut -c1-220 >> $L; }
for i in 1 2; do
  run SCENES=40 ROUNDS=8 RCMVS_LIB=tools/dev/_variants/lib_plain.so
  run SCENES=40 ROUNDS=8 RCMVS_LIB=tools/dev/_variants/lib_acq.so
  run SCENES=40 ROUNDS=8 RCMVS_LIB=tools/dev/_variants/lib_sc0.so
  run SCENES=40 ROUNDS=8
done
echo "== reduced probe (which concurrent work is needed), pre-fix library" >> $L
RCMVS_LIB=tools/dev/_variants/lib_plain.so timeout 200 python tools/dev/two_stream_minimal.py 2>&1 | grep -v "Warning\|amdgpu.ids" >> $L
cat $L
