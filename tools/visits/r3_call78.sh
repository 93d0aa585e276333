#!/bin/bash
# visit 78: the first differing op is K1 of stage 3 -- are its inputs (feature maps of the just-in-time pyramid output conv, planes) already wrong?
mkdir -p gpurun_out; L=gpurun_out/r3c78.log; : > $L
env SCENES=40 ROUNDS=8 CAPTURE_NAMES=warp_variance,fpn_out_fused,hypothesis_planes,conv2d,conv2d_s2d,compose_homography_stages timeout 300 python tools/dev/two_stream_firstbad.py 2>&1 | grep -v "Warning\|amdgpu.ids" | tail -60 >> $L
cat $L
