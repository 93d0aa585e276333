"""Developer A/B (GPU box): the last FPN level as the fused kernel (bit-identical to the two-kernel path) and as the folded kernel, at the
config-2 shape (3 views, 512x640)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from rc_mvsnet_amd import _lib, ops
_lib.load()
dev = "cuda:0"
g = torch.Generator().manual_seed(0)
lat, up = torch.randn(3, 512, 640, 8, generator=g).to(dev), torch.randn(3, 256, 320, 32, generator=g).to(dev)
w_in_t, b_in = (0.3 * torch.randn(32, 8, 1, 1, generator=g)).to(dev), (0.1 * torch.randn(32, generator=g)).to(dev)
w_out_t = (0.1 * torch.randn(8, 32, 3, 3, generator=g)).to(dev)
w_in, w_out = ops.pack_conv2d_weight(w_in_t), ops.pack_conv2d_weight(w_out_t)
tab = ops.pack_fpn_folded(w_in_t, b_in, w_out_t)
def t(fn, R=20):
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(R): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / R
a = ops.fpn_out_fused(lat, up, w_in, b_in, w_out)
b = ops.fpn_out_folded(lat, up, tab)
print("max |folded - fused| / max|fused| =", float((a - b).abs().max() / a.abs().max()))
for i in range(2):
    print(f"fused  {t(lambda: ops.fpn_out_fused(lat, up, w_in, b_in, w_out)):7.1f} us   folded {t(lambda: ops.fpn_out_folded(lat, up, tab)):7.1f} us")
