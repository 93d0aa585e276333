#!/bin/bash
set -u
exec < /dev/null
O=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $O
{
for b in x3_test x3_test_fewc1 x3_test_fewc2; do
  echo "== $b X3_NP=2 check"
  X3_NP=2 timeout 120 tools/dev/$b 1 2>&1 | grep -i "kind=0 Ci=\(16\|8\) Co=8" | head -4 | cut -c1-200
  echo "== $b X3_NP=2 timings"
  X3_NOCHECK=1 X3_NORES=1 X3_NP=2 timeout 60 tools/dev/$b 3 2>&1 | grep "time" | grep "kind=0 Ci=\(16\|8\) Co=8" | cut -c1-110
done
} | tee $O/r3c47_x3_fewc.txt
echo "== homography change: cascade tests + bench"
timeout 600 python -m pytest tests/test_gpu_parity.py -q -x -p no:cacheprovider -k "homography or cascade" 2>&1 | tail -2
timeout 600 python bench.py --steps 300 --warmup 10 --no-cpu-baseline --no-train-step 2>&1 | tail -1 | cut -c1-200
exit 0
