#!/bin/bash
# visit 70: two-stream corruption -- address ranges of the volumes per stream, guard bands, and two PROCESSES on one GPU
mkdir -p gpurun_out; L=gpurun_out/r3c70.log; : > $L
run() { env "$@" timeout 120 python tools/dev/two_stream_depth.py 2>&1 | grep -v Warning | tail -4 >> $L; }
run SCENES=16 ROUNDS=4 LOGADDR=1 MINBYTES=16000000
run SCENES=40 PAD=8388608 MINBYTES=16000000
run SCENES=40 PAD=8388608 MINBYTES=1000000
S=$(python -c "import time; print(time.time() + 45)")
START=$S TAG=a timeout 200 python tools/dev/two_process_check.py > gpurun_out/r3c70_a.log 2>&1 &
PA=$!
START=$S TAG=b timeout 200 python tools/dev/two_process_check.py > gpurun_out/r3c70_b.log 2>&1 &
PB=$!
wait $PA $PB
tail -1 gpurun_out/r3c70_a.log >> $L; tail -1 gpurun_out/r3c70_b.log >> $L
cat $L
