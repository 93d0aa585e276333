"""Developer A/B: prob conv (8 -> 1) plane-marching kernel, config-2 stage shapes, HIP-event timing; z chunk sweep (impl bits 16-23)
and the tile kernel for reference."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from rc_mvsnet_amd import _lib, ops
_lib.load()
dev = "cuda:0"
w = ops.pack_conv3d_weight(torch.randn(1, 8, 3, 3, 3, device=dev) / 15)
tot = {}
for D, H, W in ((48, 128, 160), (32, 256, 320), (8, 512, 640)):
    x = torch.randn(1, D, H, W, 8, device=dev)
    ops.force_direct_conv(0)
    ref = ops.conv3d(x, w)
    variants = [("auto", 0)] + [(f"zc={z}", z << 16) for z in (2, 3, 4, 6, 8, 12, 16) if z <= D] + [("tile", 2)]
    if os.environ.get("PROB_ONLY_AUTO"): variants = variants[:1]
    for name, cfg in variants:
        ops.force_direct_conv(cfg)
        for _ in range(3): y = ops.conv3d(x, w)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): ops.conv3d(x, w)
        e1.record(); torch.cuda.synchronize()
        ops.force_direct_conv(0)
        us = e0.elapsed_time(e1) * 1e3 / 20
        tot[name] = tot.get(name, 0.0) + us
        err = float((y - ref).abs().max())
        print(f"prob conv {D}x{H}x{W}: {name:6s} {us:7.1f} us  ({D*H*W*8*4/us/1e3:.0f} GB/s of input)  max |y - auto| {err:.2e}")
print("per scene:", {k: round(v, 1) for k, v in tot.items() if k in ("auto", "tile", "zc=8")})
