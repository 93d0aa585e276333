#!/bin/bash
# visit 86: same-visit A/B of the adopted fix on the UNINSTRUMENTED probe: plain loads in the planes kernel against the product library, alternating
mkdir -p gpurun_out; L=gpurun_out/r3c86.log; : > $L
run() { env "$@" timeout 100 python tools/dev/two_stream_depth.py 2>&1 | grep -v "Warning\|amdgpu.ids" | tail -1 | cut -c1-260 >> $L; }
for i in 1 2 3; do
  run SCENES=40 ROUNDS=8 RCMVS_LIB=tools/dev/_variants/lib_plain.so
  run SCENES=40 ROUNDS=8
done
cat $L
