"""Developer probe: relative Frobenius error against fp64 of the split-bf16 kernels and of the fp32 FMA-chain kernels, for input
distributions shaped like activations and like back-propagated gradients (tiny magnitudes, heavy tails, exact zeros)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import torch.nn.functional as F
from rc_mvsnet_amd import _lib, ops

_lib.load()
dev = "cuda:0"
g = torch.Generator().manual_seed(0)


def dist(name, shape):
    n = torch.randn(shape, generator=g)
    if name == "normal": return n
    if name == "lognormal x1e-6": return n * torch.exp(2.0 * torch.randn(shape, generator=g)) * 1e-6
    if name == "heavy x1e-9 + zeros": return n * torch.exp(4.0 * torch.randn(shape, generator=g)) * 1e-9 * (torch.rand(shape, generator=g) > 0.5)
    if name == "x1e-20": return n * 1e-20
    if name == "x1e-30": return n * 1e-30
    raise KeyError(name)


cases = [("s1", 16, 16), ("s1", 16, 8), ("s1", 32, 8), ("s1", 8, 8), ("s2", 8, 16), ("s2", 16, 32), ("t2", 16, 8), ("p1", 32, 32)]
for dn in ("normal", "lognormal x1e-6", "heavy x1e-9 + zeros", "x1e-20", "x1e-30"):
    for kind, ci, co in cases:
        B, D, H, W = (2, 1, 40, 70) if kind == "p1" else (1, 6, 20, 40)
        x = dist(dn, (B, ci, D, H, W))
        if kind == "t2":
            w = torch.randn(ci, co, 3, 3, 3, generator=g) / (ci * 27 / 8) ** 0.5
            ref = F.conv_transpose3d(x.double(), w.double(), padding=1, stride=2, output_padding=1)
            wp = ops.pack_conv3d_weight(w.to(dev), transposed=True)
            run = lambda: ops.deconv3d(x.permute(0, 2, 3, 4, 1).contiguous().to(dev), wp)
        else:
            st = 2 if kind == "s2" else 1
            w = torch.randn(co, ci, 3, 3, 3, generator=g) / (ci * 27) ** 0.5
            ref = F.conv3d(x.double(), w.double(), padding=1, stride=st)
            wp = ops.pack_conv3d_weight(w.to(dev))
            run = lambda: ops.conv3d(x.permute(0, 2, 3, 4, 1).contiguous().to(dev), wp, stride=st)
        out = {}
        for name, cfg in (("x3", 0), ("fp32", 64)):
            ops.force_direct_conv(cfg)
            out[name] = run().cpu().permute(0, 4, 1, 2, 3).double()
            ops.force_direct_conv(0)
        e = {k: float((v - ref).norm() / ref.norm()) for k, v in out.items()}
        print(f"{dn:22s} {kind} {ci:2d}->{co:2d}: rel Frobenius error x3 {e['x3']:.2e}  fp32 {e['fp32']:.2e}")
