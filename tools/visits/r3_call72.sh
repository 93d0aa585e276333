#!/bin/bash
# visit 72: guard-band canaries around every ops-layer tensor -- does any kernel write outside its output?
mkdir -p gpurun_out; L=gpurun_out/r3c72.log; : > $L
timeout 200 python tools/dev/oob_canary.py 2>&1 | grep -v "amdgpu.ids" | tail -60 >> $L
STREAMS=2 timeout 200 python tools/dev/oob_canary.py 2>&1 | grep -v "amdgpu.ids" | tail -40 >> $L
cat $L
