"""Developer A/B: prob conv (8 -> 1) plane-marching kernel vs the tile kernel, config-2 stage shapes, HIP-event timing."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from rc_mvsnet_amd import _lib, ops
_lib.load()
dev = "cuda:0"
w = ops.pack_conv3d_weight(torch.randn(1, 8, 3, 3, 3, device=dev) / 15)
for D, H, W in ((48, 128, 160), (32, 256, 320), (8, 512, 640)):
    x = torch.randn(1, D, H, W, 8, device=dev)
    for name, cfg in (("marching", 0), ("tile", 2)):
        ops.force_direct_conv(cfg)
        for _ in range(3): ops.conv3d(x, w)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): ops.conv3d(x, w)
        e1.record(); torch.cuda.synchronize()
        ops.force_direct_conv(0)
        us = e0.elapsed_time(e1) * 1e3 / 20
        print(f"prob conv {D}x{H}x{W}: {name:9s} {us:7.1f} us  ({D*H*W*8*4/us/1e3:.0f} GB/s of input)")
