"""Developer check (GPU box): K1 at the stage-2 shape of config 2 (C = 16, D = 32, 256 x 320, 3 views), rough and smooth plane tables, variants 0 / 1 / 7; RCMVS_LIB selects a variant build."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from rc_mvsnet_amd import _lib, ops, synthetic
if os.environ.get("RCMVS_LIB"):
    _lib.LIB_PATH = os.path.abspath(os.environ["RCMVS_LIB"])
_lib.load()
dev = "cuda:0"
V, H, W, C, D, scale, ratio = 3, 512, 640, 16, 32, 2, 2
h, w = H // scale, W // scale
g = torch.Generator().manual_seed(5)
feats = torch.randn(1, V, h, w, C, generator=g).to(dev)
rot, trans = ops.compose_homography(synthetic.proj_matrices(1, V, H, W)["stage2"].to(dev))
dv = synthetic.depth_values(1)
lo, hi = float(dv[0, 0]), float(dv[0, -1])
step = (hi - lo) / 192.0 * ratio
yy, xx = torch.meshgrid(torch.arange(h, dtype=torch.float32), torch.arange(w, dtype=torch.float32), indexing="ij")
tables = {"smooth": (0.5 * (lo + hi) + 0.2 * (hi - lo) * torch.sin(xx / (0.13 * w)) * torch.cos(yy / (0.17 * h))) - 0.5 * D * step,
          "rough": lo + (hi - lo - D * step) * torch.rand(h, w, generator=g)}
line = os.environ.get("RCMVS_LIB", "product") + ":"
for tn, d0 in tables.items():
    planes = torch.stack((d0, torch.full_like(d0, step)), dim=-1).unsqueeze(0).contiguous().to(dev)
    ref = ops.warp_variance(feats, rot, trans, planes, D, variant=2)
    for var in (0, 1):
        out = ops.warp_variance(feats, rot, trans, planes, D, variant=var)
        assert float((out - ref).abs().max()) <= 2e-6 * max(1.0, float(ref.abs().max()))
        for _ in range(3):
            ops.warp_variance(feats, rot, trans, planes, D, variant=var)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(30):
            ops.warp_variance(feats, rot, trans, planes, D, variant=var)
        e1.record()
        torch.cuda.synchronize()
        line += f"  {tn} v{var} {e0.elapsed_time(e1) / 30 * 1e3:.1f} us"
print(line)
