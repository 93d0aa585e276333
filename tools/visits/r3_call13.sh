#!/bin/bash
# round 3, GPU visit 13: bisect of the FeatureNet gradient outlier by loss term
set -u
exec < /dev/null
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $O
{
for l in both depth noref noref_var; do
  echo "== PROBE_LOSS=$l"
  PROBE_LOSS=$l timeout 300 python tools/dev/grad_probe.py 128 160 2>&1 | grep -v "^/opt\|Warning\|warn" | tail -5 | cut -c1-330
done
} | tee $O/r3c13_grad_probe.txt
exit 0
