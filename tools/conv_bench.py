#!/usr/bin/env python
"""Time the 3-D conv layers of the three config-2 CostRegNets (GPU box)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from rc_mvsnet_amd import ops
dev = "cuda:0"
def t(fn, R=10):
    for _ in range(2): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(R): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / R
for (C, D, h, w) in ((32, 48, 128, 160), (16, 32, 256, 320), (8, 8, 512, 640)):
    x = torch.randn(1, D, h, w, C, device=dev)
    wp = ops.pack_conv3d_weight(torch.randn(8, C, 3, 3, 3, device=dev) * 0.05)
    sc, sh = torch.rand(8, device=dev) + 0.5, torch.randn(8, device=dev) * 0.1
    fl = 2 * 27 * C * 8 * D * h * w
    res = []
    for cfg, name in ((0, "lds default"), (2, "lds ck16"), (4, "lds split"), (8, "lds nosplit"), (1, "direct")):
        ops.force_direct_conv(cfg)
        us = t(lambda: ops.conv3d(x, wp, sc, sh, relu=True))
        res.append(f"{name} {us:7.1f} us {fl / us / 1e6:5.1f} TF")
    ops.force_direct_conv(0)
    print(f"conv0 {C}->8 @ {D}x{h}x{w}: " + " | ".join(res))
    x8 = torch.randn(1, D, h, w, 8, device=dev)
    w1 = ops.pack_conv3d_weight(torch.randn(1, 8, 3, 3, 3, device=dev) * 0.05)
    x16 = torch.randn(1, D // 2, h // 2, w // 2, 16, device=dev)
    wt = ops.pack_conv3d_weight(torch.randn(16, 8, 3, 3, 3, device=dev) * 0.05, transposed=True)
    us = t(lambda: ops.deconv3d(x16, wt, sc, sh, x8, relu=True))
    print(f"deconv11 16->8 -> {D}x{h}x{w}: {us:8.1f} us  {2 * 27 * 16 * 8 * D * h * w / 8 / us / 1e6:6.1f} TF")

print("-- conv2 16->16 (stage L1 volumes): LDS/scalar-weight kernel vs MFMA kernel")
for (D, h, w) in ((24, 64, 80), (16, 128, 160), (4, 256, 320)):
    x = torch.randn(1, D, h, w, 16, device=dev)
    wp = ops.pack_conv3d_weight(torch.randn(16, 16, 3, 3, 3, device=dev) * 0.05)
    sc, sh = torch.rand(16, device=dev) + 0.5, torch.randn(16, device=dev) * 0.1
    fl = 2 * 27 * 16 * 16 * D * h * w
    res = []
    for cfg, name in ((0, "lds"), (4, "lds split"), (16, "mfma")):
        ops.force_direct_conv(cfg)
        us = t(lambda: ops.conv3d(x, wp, sc, sh, relu=True))
        res.append(f"{name} {us:7.1f} us {fl / us / 1e6:5.1f} TF")
    ops.force_direct_conv(0)
    print(f"conv2 16->16 @ {D}x{h}x{w}: " + " | ".join(res))
