"""Developer ablation of the K1 backward (rcmvs_warp_variance_bwd) at the config-3 stage shapes (4 views): production, without
run-length merging, and without the atomic scatter (arithmetic floor).  Two plane models: 'noisy' = per-pixel depth ranges around a
random previous depth (what a random-init network produces: the bench), 'smooth' = around a smooth previous depth."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from rc_mvsnet_amd import _lib, ops, synthetic
_lib.load()
dev = "cuda:0"
V, H, W = 4, 512, 640
dv = synthetic.depth_values(1).to(dev)
names = {0: "production", 2: "no run-length merging", 1: "no scatter (floor)", 3: "no merging, no scatter"}
for (C, D, sc, key) in ((32, 48, 4, "stage1"), (16, 32, 2, "stage2"), (8, 8, 1, "stage3")):
    h, w = H // sc, W // sc
    g = torch.Generator().manual_seed(C)
    feats = torch.randn(1, V, h, w, C, generator=g).to(dev)
    gvar = torch.randn(1, D, h, w, C, generator=g).to(dev)
    rot, trans = ops.compose_homography(synthetic.proj_matrices(1, V, H, W)[key].to(dev))
    for model in ("noisy", "smooth"):
        if sc == 4:
            if model == "smooth":
                continue
            planes = ops.hypothesis_planes(None, dv, (H, W), sc, D, 4)
        else:
            if model == "noisy":
                prev = (500.0 + 300.0 * torch.rand(1, h // 2, w // 2, generator=g)).to(dev)
            else:
                yy, xx = torch.meshgrid(torch.linspace(0, 1, h // 2), torch.linspace(0, 1, w // 2), indexing="ij")
                prev = (600.0 + 80.0 * xx + 40.0 * yy)[None].to(dev)
            planes = ops.hypothesis_planes(prev, dv, (H, W), sc, D, float(sc))
        for v in (0, 2, 1):
            run = lambda: ops.warp_variance_bwd(feats, rot, trans, planes, gvar, None, variant=v)
            for _ in range(2):
                run()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize(); e0.record()
            for _ in range(5):
                run()
            e1.record(); torch.cuda.synchronize()
            print(f"C={C:2d} D={D:2d} {h}x{w} {model:6s} planes: {names[v]:24s} {e0.elapsed_time(e1) / 5:7.3f} ms (incl. the zero fill of the gradient)")
