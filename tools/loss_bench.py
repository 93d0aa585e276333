"""Self-supervised loss at BASELINE config 3 shape (4 views, 512x640, batch 1): UnsupLossMultiStage forward + backward on
the HIP path (ms, HIP events) -- kernels only, e.g. under rocprofv3.  The CPU baseline beside it is
``python bench.py --workload unsup_loss``."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rc_mvsnet_amd import _lib, losses, synthetic       # noqa: E402


def depths(B, H, W):
    out = {}
    for i, s in enumerate((4, 2, 1)):
        h, w = H // s, W // s
        yy, xx = torch.meshgrid(torch.linspace(0, 1, h), torch.linspace(0, 1, w), indexing="ij")
        d = 620.0 + 110.0 * torch.sin(4.0 * xx + i) * torch.cos(3.0 * yy) + torch.randn(h, w, generator=torch.Generator().manual_seed(i))
        out["stage%d" % (i + 1)] = d.unsqueeze(0).repeat(B, 1, 1)
    return out


def main():
    _lib.load()
    dev = "cuda:0"
    B, V, H, W = 1, 4, 512, 640
    imgs, cams, dep = synthetic.images(B, V, H, W, 0), synthetic.proj_matrices(B, V, H, W), depths(B, H, W)
    gi, gc = imgs.to(dev), {k: v.to(dev) for k, v in cams.items()}
    mod = losses.UnsupLossMultiStage()

    def step():
        inputs = {k: {"depth": d.to(dev).requires_grad_(True)} for k, d in dep.items()}
        total, _ = mod(inputs, gi, gc, dlossw=[0.5, 1.0, 2.0])
        total.backward()
        return total

    for _ in range(3):
        step()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    n = 20
    e0.record()
    for _ in range(n):
        total = step()
    e1.record()
    torch.cuda.synchronize()
    hip_ms = e0.elapsed_time(e1) / n
    print(f"unsup loss 3 stages fwd+bwd: HIP {hip_ms:.3f} ms (loss {float(total):.6f})")


if __name__ == "__main__":
    main()
