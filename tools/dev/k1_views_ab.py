"""Developer check (GPU box), round 6: which K1 kernel for which view count.  For the stage shapes of the DTU bench scene (V = 3, 512x640),
training (V = 4), the reference's DTU evaluation (V = 5, 1184x1600, eval_rcmvsnet_dtu.py:49-51) and Tanks and Temples (V = 7, 1056x1920,
D = 64/32/8, eval_rcmvsnet_tanks.py:47,53-55): the two-phase gather kernel (variant 0), its FMA build (1), the plane-pipelined form (7) and the
LDS-window form (5), on a pixel-invariant plane table (stage 1), a rough per-pixel table (what random weights produce at stages 2 / 3) and a
smooth one.  Prints us per launch and the algorithmic-bytes rate ((V + D) h w C 4 bytes... the kernel's own traffic)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from rc_mvsnet_amd import _lib, ops, synthetic
_lib.load()
dev = "cuda:0"
REPS = int(os.environ.get("REPS", "20"))


def timed(fn):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(REPS):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1000.0 / REPS


CASES = [("bench V=3", 3, 512, 640, (48, 32, 8)), ("train V=4", 4, 512, 640, (48, 32, 8)), ("dtu_eval V=5", 5, 1184, 1600, (48, 32, 8)),
         ("tanks V=7", 7, 1056, 1920, (64, 32, 8))]
for name, V, H, W, nd in CASES:
    pm = synthetic.proj_matrices(1, V, H, W)
    dv = synthetic.depth_values(1).to(dev)
    for s, (C, scale, ratio) in enumerate(((32, 4, 4), (16, 2, 2), (8, 1, 1))):
        D = nd[s]
        h, w = H // scale, W // scale
        g = torch.Generator().manual_seed(7 * V + s)
        feats = torch.randn(1, V, h, w, C, generator=g).to(dev)
        rot, trans = ops.compose_homography(pm[f"stage{s + 1}"].to(dev))
        tables = {}
        if s == 0:
            tables["uniform"] = ops.hypothesis_planes(None, dv, (H, W), scale, D, ratio)
        else:
            lo, hi = float(dv[0, 0]), float(dv[0, -1])
            step = (hi - lo) / 192.0 * ratio
            yy, xx = torch.meshgrid(torch.arange(h, dtype=torch.float32), torch.arange(w, dtype=torch.float32), indexing="ij")
            smooth = (0.5 * (lo + hi) + 0.2 * (hi - lo) * torch.sin(xx / (0.13 * w)) * torch.cos(yy / (0.17 * h))) - 0.5 * D * step
            rough = lo + (hi - lo - D * step) * torch.rand(h, w, generator=g)
            for tn, d0 in (("smooth", smooth), ("rough", rough)):
                tables[tn] = torch.stack((d0, torch.full_like(d0, step)), dim=-1).unsqueeze(0).contiguous().to(dev)
        alg = (V * h * w * C + D * h * w * C) * 4 + h * w * 8
        for tn, planes in tables.items():
            ref = ops.warp_variance(feats, rot, trans, planes, D, variant=2)
            tol = 2e-6 * max(1.0, float(ref.abs().max()))
            line = f"{name} stage{s + 1} C={C} D={D} {h}x{w} {tn:8s}:"
            for var in (0, 1, 7, 5):
                if var == 7 and V - 1 not in (2, 3, 4, 6):
                    continue
                out = ops.warp_variance(feats, rot, trans, planes, D, variant=var)
                err = float((out - ref).abs().max())
                assert err <= tol, (name, s, var, err, tol)
                us = timed(lambda: ops.warp_variance(feats, rot, trans, planes, D, variant=var))
                line += f"  v{var} {us:7.1f} us ({alg / us * 1e-6:5.2f} TB/s)"
            if tn != "rough":
                _, blocks, onw = ops.warp_variance_win(feats, rot, trans, planes, D, variant=5)
                line += f"  window path {onw}/{blocks}"
            print(line, flush=True)
        del feats, ref, out
        torch.cuda.empty_cache()
