#!/bin/bash
# visit 75: two-stream corruption with explicit event dependencies between consecutive library launches of a stream
mkdir -p gpurun_out; L=gpurun_out/r3c75.log; : > $L
run() { env "$@" timeout 120 python tools/dev/two_stream_depth.py 2>&1 | grep -v "Warning\|amdgpu.ids" | tail -1 >> $L; }
run SCENES=40 FENCE=2
run SCENES=40 FENCE=1
run SCENES=40 FENCE=2
run SCENES=40 FENCE=1
cat $L
