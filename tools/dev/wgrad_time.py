"""Developer timing of the 3-D conv weight-gradient kernels at the config-3 layer shapes (HIP events, 10 launches each)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from rc_mvsnet_amd import _lib, train_ops
_lib.load()
dev = "cuda:0"
shapes = [(32, 8, 1, 48, 128, 160), (16, 8, 1, 32, 256, 320), (8, 8, 1, 8, 512, 640), (8, 16, 2, 32, 256, 320), (16, 16, 1, 16, 128, 160),
          (16, 32, 2, 16, 128, 160), (32, 32, 1, 8, 64, 80), (32, 64, 2, 8, 64, 80), (64, 64, 1, 4, 32, 40), (8, 1, 1, 32, 256, 320), (8, 1, 1, 8, 512, 640)]
tot = 0.0
for Ci, Co, st, D, H, W in shapes:
    x = torch.randn(1, D, H, W, Ci, device=dev)
    dy = torch.randn(1, (D - 1) // st + 1, (H - 1) // st + 1, (W - 1) // st + 1, Co, device=dev)
    for _ in range(2): train_ops.conv3d_wgrad(x, dy, st)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): train_ops.conv3d_wgrad(x, dy, st)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 100
    fl = 2.0 * 27 * Ci * Co * dy.numel() / Co
    tot += us
    print(f"wgrad {Ci:2d}->{Co:2d} s{st} {D}x{H}x{W}: {us:8.1f} us  {fl / us / 1e6:6.1f} TF")
print(f"sum {tot:.0f} us")
