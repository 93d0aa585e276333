#!/bin/bash
# visit 69: two-stream corruption -- memory reuse or host run-ahead?
mkdir -p gpurun_out; L=gpurun_out/r3c69.log; : > $L
run() { env "$@" timeout 120 python tools/dev/two_stream_depth.py 2>&1 | grep -v Warning | tail -1 >> $L; }
run SCENES=40
run SCENES=40 KEEPALIVE=2 DELAY=4
run SCENES=40 AHEAD=2
run SCENES=40 AHEAD=4
run SCENES=40 AHEAD=8
run SCENES=16 KEEPALIVE=1 MINBYTES=16000000
run SCENES=16 KEEPALIVE=1 MAXBYTES=16000000
cat $L
