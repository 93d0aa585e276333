#!/bin/bash
set -u
exec < /dev/null
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out
tag=${1:-r3c24}
mkdir -p $O
echo "== FeatureNet timing"
timeout 300 python tools/dev/fnet_time.py 2>&1 | grep -v amdgpu.ids | tee $O/${tag}_fnet_time.txt
echo "== pytest subset"
timeout 1200 python -m pytest tests/test_gpu_parity.py -q -x -p no:cacheprovider -k "conv2d or fpn or feature or cascade or golden" 2>&1 | tail -4 | tee $O/${tag}_pytest.log
echo "== bench"
timeout 600 python bench.py --steps 300 --warmup 10 --no-cpu-baseline --no-train-step 2>&1 | tail -1 | tee $O/${tag}_bench.json | cut -c1-330
python - <<PY
import json
d = json.load(open("gpurun_out/${tag}_bench.json"))
print(d["ms_per_step"], "roofline", d["roofline"]["frac"], d["roofline"]["per_stage_us"], "conv", d["roofline_conv"]["us_per_scene"])
PY
exit 0
