#!/bin/bash
# visit 82: the planes kernel reads the previous depth at agent scope -- uninstrumented two-stream checks
mkdir -p gpurun_out; L=gpurun_out/r3c82.log; : > $L
run() { env "$@" timeout 200 python tools/dev/two_stream_depth.py 2>&1 | grep -v "Warning\|amdgpu.ids" | tail -1 >> $L; }
run SCENES=40 ROUNDS=16
run SCENES=40 ROUNDS=16
run SCENES=12 ROUNDS=16
timeout 200 python tools/dev/two_stream_graph_probe.py 2>&1 | grep -v "Warning\|amdgpu.ids" | tail -4 >> $L
timeout 300 python bench.py --streams 2 2>/dev/null | tail -1 > gpurun_out/r3c82_bench2.json
python - >> $L <<'PY'
import json
b = json.load(open("gpurun_out/r3c82_bench2.json"))
print("bench --streams 2:", b["value"], b["ms_per_step"], {k: b[k] for k in b if "identical" in k or "single" in k})
PY
cat $L
