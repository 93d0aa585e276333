#!/bin/bash
# same-box A/B of library builds: bench.py (300 steps, no side passes) per library given as arguments ("product" or a path), two rounds; then the kernel stats of the LAST one
mkdir -p gpurun_out
tag=$1; shift
: > gpurun_out/r6_libs_$tag.log
for rep in 1 2; do
for lib in "$@"; do
    ( [ $lib != product ] && export RCMVS_LIB=$GRAFT_REPO_ROOT/$lib; timeout 600 python bench.py --steps 300 --warmup 20 --no-cpu-baseline --no-train-step --no-side-pass 2>/dev/null | python -c "
import json,sys
b=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('$lib', 'value', b['value'], 'ms', b['ms_per_step'], 'parity', (b.get('parity') or {}).get('depth_l1_over_range'))" ) | tee -a gpurun_out/r6_libs_$tag.log
done
done
