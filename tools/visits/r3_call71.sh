#!/bin/bash
# visit 71: two-stream corruption -- address ranges of the volumes per stream, guard bands
mkdir -p gpurun_out; L=gpurun_out/r3c71.log; : > $L
run() { env "$@" timeout 120 python tools/dev/two_stream_depth.py 2>&1 | grep -v Warning | tail -6 >> $L; }
run SCENES=16 ROUNDS=4 LOGADDR=1 MINBYTES=16000000
run SCENES=40 PAD=8388608 MINBYTES=16000000
run SCENES=40 PAD=8388608 MINBYTES=1000000
cat $L
