#!/bin/bash
# conv0 variants (tools/dev/build_z8_variants.sh, built in the container into tools/dev/_v/): per-layer times, conv0 rows + total, depth checksum
mkdir -p gpurun_out
out=gpurun_out/r6_z8_${1:-a}.txt; shift
: > $out
for rep in 1 2; do
for lib in "$@"; do
    RCMVS_LIB=$lib python tools/dev/layer_times.py 2>&1 | grep -E "^lib|->  8 1x(48|32|8)x|total" | tee -a $out
done
done
