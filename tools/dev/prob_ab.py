"""Developer A/B: depth head (prob conv 8 -> 1 + softmax + regression + confidence) at the config-2 stage shapes, HIP-event timing:
the fused single launch (production) against the two-launch form (plane-marching prob conv + in-place softmax kernel)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from rc_mvsnet_amd import _lib, ops
_lib.load()
dev = "cuda:0"
w = ops.pack_conv3d_weight(torch.randn(1, 8, 3, 3, 3, device=dev) / 15)
tot = {}
for D, H, W in ((48, 128, 160), (32, 256, 320), (8, 512, 640)):
    # volumes of all three stages alternate, as in the pipeline, so that nothing is served from a warm Infinity Cache
    xs = [torch.randn(1, D, H, W, 8, device=dev) for _ in range(4)]
    planes = torch.stack((425.0 + 50 * torch.rand(1, H, W, device=dev), 1.0 + 5 * torch.rand(1, H, W, device=dev)), dim=-1).contiguous()
    for name, var in (("strip", 3), ("fused", 2), ("two-launch", 1)):
        for i in range(4): ops.depth_head(xs[i], w, planes, variant=var)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for i in range(20): ops.depth_head(xs[i % 4], w, planes, variant=var)
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / 20
        tot[name] = tot.get(name, 0.0) + us
        nbytes = D * H * W * 8 * 4 + 4 * H * W * 4
        print(f"depth head {D}x{H}x{W}: {name:10s} {us:7.1f} us  ({nbytes / us / 1e3:.0f} GB/s of x8 + planes + outputs)")
for name, us in tot.items():
    print(f"{name}: {us:.1f} us per scene = {(199.2e6 + 6.6e6) / us / 1e3:.0f} GB/s = {(199.2e6 + 6.6e6) / us / 1e3 / 8000:.3f} of 8 TB/s")
