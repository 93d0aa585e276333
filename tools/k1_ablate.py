#!/usr/bin/env python
"""Time the K1 (warp+variance) code variants at the three BASELINE config-2 stage shapes with HIP
events on the launch stream, and print GB/s of algorithmic traffic.  GPU box only."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from rc_mvsnet_amd import _lib, ops, synthetic

lib = _lib.load()
dev = "cuda:0"
V, H, W = int(os.environ.get('K1_V', '3')), 512, 640
names = {0: "production exact", 1: "production fma", 2: "reference-order", 3: "store only", 100: "torch zero_ (memset)"}
variants = [int(a) for a in sys.argv[1:]] or [0, 1, 2, 3, 100]
dv = synthetic.depth_values(1).to(dev)
tot = {}
for (C, D, sc, key) in ((32, 48, 4, "stage1"), (16, 32, 2, "stage2"), (8, 8, 1, "stage3")):
    h, w = H // sc, W // sc
    g = torch.Generator().manual_seed(C)
    feats = torch.randn(1, V, h, w, C, generator=g).to(dev)
    rot, trans = ops.compose_homography(synthetic.proj_matrices(1, V, H, W)[key].to(dev))
    if sc == 4:
        planes = ops.hypothesis_planes(None, dv, (H, W), sc, D, 4)
    else:   # realistic per-pixel ranges around a smooth depth map
        prev = (600.0 + 100.0 * torch.rand(1, h // 2, w // 2, generator=g)).to(dev)
        planes = ops.hypothesis_planes(prev, dv, (H, W), sc, D, float(sc))
    nbytes = 4 * ((V - 1) * C * h * w + C * h * w + D * h * w + C * D * h * w)
    outbuf = torch.empty((1, D, h, w, C), device=dev)
    for v in variants:
        if v == 100:
            run = lambda: outbuf.zero_()
        else:
            run = lambda v=v: ops.warp_variance(feats, rot, trans, planes, D, variant=v)
        for _ in range(3):
            run()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record()
        R = 20
        for _ in range(R):
            run()
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / R
        tot.setdefault(v, []).append(us)
        print(f"C={C:2d} D={D:2d} {h}x{w}  variant {v} ({names[v]:24s}): {us:8.1f} us  {nbytes / us / 1e3:8.1f} GB/s")
for v, us_list in tot.items():          # a variant listed twice is timed twice (drift check); the best pass of each stage counts
    n = len(us_list) // 3
    t = sum(min(us_list[s * n:(s + 1) * n]) for s in range(3))
    print(f"variant {v} ({names[v]}): total {t:.1f} us/scene -> {457441280 / t / 1e3:.0f} GB/s = {457441280 / t / 1e3 / 8000:.3f} of 8 TB/s")
