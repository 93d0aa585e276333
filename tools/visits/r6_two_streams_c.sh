#!/bin/bash
# Round 6: two-stream diagnostic, taps at stage 3 too; product library and plain-store variant.
mkdir -p gpurun_out
L=gpurun_out/r6_two_streams_c.log
: > $L
run() { echo "=== $*" >> $L; ( timeout 300 env "$@" python tools/dev/two_stream_diag.py probe 2>&1 | grep -v "Warning\|amdgpu.ids" | cut -c1-700 ) >> $L; }
run X=1
run RCMVS_LIB=tools/dev/_variants/lib_plain_store.so
grep -n "===\|differing tensors\|K1 blocks\|\[probe\]\|bound rows" $L | cut -c1-400
