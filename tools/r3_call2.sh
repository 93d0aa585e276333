#!/bin/bash
# round 3, GPU visit 2: fused depth head (parity + timing), K1 schedule variants, side-stream FPN output convs A/B
set -u
exec < /dev/null
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $O
echo "== pytest subset"
timeout 900 python -m pytest tests/test_gpu_parity.py -q -x -p no:cacheprovider -k "depth_head or variants_agree or cascade_vs_reference_golden or cascade_batch_two" 2>&1 | tail -6 | tee $O/r3c2_pytest.log
echo "== depth head A/B"
timeout 300 python tools/dev/prob_ab.py 2>&1 | tail -10 | tee $O/r3c2_prob_ab.txt
echo "== K1 variants"
timeout 300 python tools/k1_ablate.py 0 4 5 6 0 2>&1 | tail -22 | tee $O/r3c2_k1_variants.txt
echo "== bench, single stream"
RCMVS_SIDE_STREAM=0 timeout 600 python bench.py --steps 300 --warmup 10 --no-cpu-baseline 2>&1 | tail -1 | tee $O/r3c2_bench_single.json | cut -c1-330
echo "== bench, side stream"
timeout 600 python bench.py --steps 300 --warmup 10 --no-cpu-baseline 2>&1 | tail -1 | tee $O/r3c2_bench_side.json | cut -c1-330
echo "== bench, single stream again"
RCMVS_SIDE_STREAM=0 timeout 600 python bench.py --steps 300 --warmup 10 --no-cpu-baseline 2>&1 | tail -1 | cut -c1-330
exit 0
