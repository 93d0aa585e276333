#!/bin/bash
set -u
exec < /dev/null
O=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $O
{
echo "== default (fp16 pair)"; timeout 300 python tools/dev/two_stream_check.py 2>&1 | grep "two streams"
echo "== RCMVS_FP16_PAIR=0"; RCMVS_FP16_PAIR=0 timeout 300 python tools/dev/two_stream_check.py 2>&1 | grep "two streams"
echo "== HIP_LAUNCH_BLOCKING-free, AMD_SERIALIZE_KERNEL=3 (sanity: serialised)"; AMD_SERIALIZE_KERNEL=3 timeout 300 python tools/dev/two_stream_check.py 2>&1 | grep "two streams"
} | cut -c1-260 | tee $O/r3c53_two_stream_bisect.txt
exit 0
