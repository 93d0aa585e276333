#!/bin/bash
# Round 6: (1) two-stream diagnostic with the product library and with the plain-store variant (tools/dev/build_plain_store_variant.sh),
# (2) the K1 GPU tests, (3) which K1 kernel for which view count (tools/dev/k1_views_ab.py).
mkdir -p gpurun_out
L=gpurun_out/r6_two_streams_b.log
: > $L
run() { echo "=== $*" >> $L; ( timeout 300 env "$@" python tools/dev/two_stream_diag.py probe 2>&1 | grep -v "Warning\|amdgpu.ids" | cut -c1-900 | tail -40 ) >> $L; }
run X=1
run RCMVS_LIB=tools/dev/_variants/lib_plain_store.so
tail -n 60 $L
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "warp_variance" 2>&1 | tail -5 | tee gpurun_out/r6_k1_tests.log
timeout 900 python tools/dev/k1_views_ab.py 2>&1 | grep -v "Warning\|amdgpu.ids" | tee gpurun_out/r6_k1_views.txt
