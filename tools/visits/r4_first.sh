#!/bin/bash
# FIRST visit of the next round (prepared at the end of round 3, when the GPU budget was spent): what the round left unmeasured.
#  1. K1's wave-specialised form (csrc/warp_variance.hip: warp_variance_ws_kernel, debug variants 4-7) against the production kernel
#     at the three config-2 stage shapes -- bit-identical already (tests), speed unknown
#  2. the two-scenes-in-flight side pass a few times over (bench.py prints it; profiles/r3_two_streams.txt has the history)
mkdir -p gpurun_out; L=gpurun_out/r4_first.log; : > $L
timeout 200 python tools/k1_ablate.py 0 4 5 6 7 0 2>&1 | grep -v amdgpu.ids >> $L
for i in 1 2 3; do timeout 300 python bench.py --steps 200 --warmup 10 --no-cpu-baseline --no-train-step 2>/dev/null | tail -1 | python -c "import json,sys; b=json.load(sys.stdin); print(b['value'], b['ms_per_step'], b['two_scenes_in_flight'])" >> $L; done
cat $L
