#!/bin/bash
set -u
exec < /dev/null
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $O
{
for l in depth; do
  echo "== PROBE_LOSS=$l PROBE_FEATGRAD=1"
  PROBE_FEATGRAD=1 PROBE_LOSS=$l timeout 300 python tools/dev/grad_probe.py 128 160 2>&1 | grep -v "^/opt\|Warning\|warn\|return float" | tail -7 | cut -c1-330
done
} | tee $O/r3c15_grad_probe_feat.txt
exit 0
