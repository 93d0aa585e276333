#!/bin/bash
# round 2, GPU visit 1: pipelined K1 variants on hardware (bit identity + timing sweeps), conv baseline
set -u
exec < /dev/null
mkdir -p gpurun_out
export TMPDIR=/tmp
echo "== PS tests"
RCMVS_TEST_PS=1 timeout 600 python -m pytest tests/test_gpu_parity.py -q -m gpu -k "pipelined or variants_agree" -p no:cacheprovider -x 2>&1 | tail -30 | tee gpurun_out/r2_ps_tests.log
echo "== ablate default"
timeout 300 python tools/k1_ablate.py 0 1 3 12 13 10 8 2>&1 | tee gpurun_out/r2_k1_ablate.log
for dkb in 2 4 8; do
  echo "== ablate DKB=$dkb"
  K1_PS_DKB=$dkb timeout 200 python tools/k1_ablate.py 12 13 10 8 2>&1 | tee -a gpurun_out/r2_k1_ablate.log
done
for ptex in 72 128 256; do
  echo "== ablate PTEX=$ptex"
  K1_PS_PTEX=$ptex timeout 200 python tools/k1_ablate.py 12 13 8 2>&1 | tee -a gpurun_out/r2_k1_ablate.log
done
echo "== pad16 v10"
K1_PS_PAD=16 timeout 200 python tools/k1_ablate.py 10 11 2>&1 | tee -a gpurun_out/r2_k1_ablate.log
echo "== conv bench"
timeout 300 python tools/conv_bench.py 2>&1 | tee gpurun_out/r2_conv_bench.log
exit 0
