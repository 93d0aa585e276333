#!/bin/bash
mkdir -p gpurun_out
for v in 0 1; do echo "RCMVS_CONV11_PROB=$v"; RCMVS_CONV11_PROB=$v timeout 600 python -m pytest tests/test_gpu_parity.py -q -m gpu -s -k "extreme_activation" 2>&1 | grep "gain\|passed\|failed"; done | tee gpurun_out/r6_t.log
