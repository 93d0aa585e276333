#!/bin/bash
# per-layer conv times (tools/dev/layer_times.py) under environment settings given as arguments "NAME=VAL[,NAME=VAL]" (or "none"), two repetitions each
mkdir -p gpurun_out
out=gpurun_out/r6_layers_env_${1:-a}.txt; shift
: > $out
for rep in 1 2; do
for cfg in "$@"; do
    echo "== $cfg" | tee -a $out
    ( [ "$cfg" != none ] && export ${cfg//,/ }; python tools/dev/layer_times.py 2>&1 | grep -E "^lib|->|total" | tee -a $out )
done
done
