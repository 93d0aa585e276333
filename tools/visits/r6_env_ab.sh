#!/bin/bash
# same-box A/B of environment settings: bench.py (300 steps, no side passes) per setting ("none" or NAME=VAL[,NAME=VAL]), two rounds
mkdir -p gpurun_out
tag=$1; shift
: > gpurun_out/r6_env_ab_$tag.log
for rep in 1 2; do
for cfg in "$@"; do
    ( [ "$cfg" != none ] && export ${cfg//,/ }; timeout 600 python bench.py --steps 300 --warmup 20 --no-cpu-baseline --no-train-step --no-side-pass 2>/dev/null | python -c "
import json,sys
b=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('$cfg', 'value', b['value'], 'ms', b['ms_per_step'], 'parity', b.get('parity'))" ) | tee -a gpurun_out/r6_env_ab_$tag.log
done
done
