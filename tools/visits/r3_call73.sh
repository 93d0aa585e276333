#!/bin/bash
# visit 73: two-stream corruption with K1's non-temporal store of the variance volume replaced by a plain store
mkdir -p gpurun_out; L=gpurun_out/r3c73.log; : > $L
run() { env "$@" timeout 120 python tools/dev/two_stream_depth.py 2>&1 | grep -v Warning | tail -1 >> $L; }
run SCENES=40
run SCENES=40
cat $L
