"""Developer timing of the BatchNorm reductions (bn_stats, bn_bwd_reduce) at config-3 layer shapes (HIP events)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from rc_mvsnet_amd import _lib, train_ops
_lib.load()
dev = "cuda:0"
tot = 0.0
for C, D, H, W in ((8, 48, 128, 160), (8, 32, 256, 320), (8, 8, 512, 640), (16, 16, 128, 160), (32, 8, 64, 80), (64, 4, 32, 40), (8, 4, 512, 640), (32, 4, 128, 160)):
    y = torch.randn(1, D, H, W, C, device=dev)
    dz = torch.randn_like(y)
    sums = torch.zeros(2 * C + 1, device=dev, dtype=torch.float64)
    one = torch.ones(C, device=dev)
    for name, fn in (("bn_stats", lambda: train_ops.bn_stats(y, sums)), ("bn_bwd_reduce", lambda: train_ops.bn_bwd_reduce(y, dz, one, one, one, one, sums, True))):
        for _ in range(3): fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): fn()
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / 20
        tot += us
        print(f"{name:14s} C={C:2d} {D}x{H}x{W}: {us:7.1f} us  ({y.numel() * 4 * (2 if 'bwd' in name else 1) / us / 1e3:6.0f} GB/s)")
print(f"sum {tot:.0f} us")
