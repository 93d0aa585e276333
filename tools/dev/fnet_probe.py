"""Developer probe: FeatureNet train-mode forward + backward on the HIP path (split-bf16 kernels on / off) against the reference
op graph in fp64 (CPU), one view per call like the reference's loop (models/casmvsnet.py:50-53)."""
import copy, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from oracle import aten_graph
from rc_mvsnet_amd import _lib, ops, synthetic
from rc_mvsnet_amd.casmvsnet import CascadeMVSNet

_lib.load()
dev = "cuda:0"
H, W = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (128, 160)
m0 = CascadeMVSNet(ndepths=[16, 16, 8], depth_interals_ratio=[4, 2, 1])
m0.load_state_dict(synthetic.cascade_state_dict(0, prob_gain=1.0), strict=True)
f0 = m0.feature
g = torch.Generator().manual_seed(1)
img = torch.rand(1, 3, H, W, generator=g)
up = {k: torch.randn(1, c, H // s, W // s, generator=g) for k, c, s in (("stage1", 32, 4), ("stage2", 16, 2), ("stage3", 8, 1))}


def run(fwd, device, dtype):
    f = copy.deepcopy(f0).to(device=device, dtype=dtype).train()
    out = fwd(f, img.to(device=device, dtype=dtype))
    loss = sum((out[k] * up[k].to(device=device, dtype=dtype)).sum() for k in out)
    loss.backward()
    return {k: v.detach().double().cpu() for k, v in out.items()}, {n: p.grad.detach().double().cpu() for n, p in f.named_parameters()}


o64, g64 = run(aten_graph.feature_pyramid, "cpu", torch.float64)
res = {"ref fp32 (GPU)": run(aten_graph.feature_pyramid, dev, torch.float32), "hip x3": run(lambda f, x: f(x), dev, torch.float32)}
ops.force_direct_conv(64)
res["hip fp32 kernels"] = run(lambda f, x: f(x), dev, torch.float32)
ops.force_direct_conv(0)
for name, (o, gr) in res.items():
    print(name, "forward:", {k: f"{float((o[k] - o64[k]).norm() / o64[k].norm()):.1e}" for k in o})
    errs = {n: float((gr[n] - g64[n]).norm() / g64[n].norm()) for n in g64}
    print("   grads:", ", ".join(f"{n.replace('.conv.weight', '.w').replace('.bn.', '.')} {e:.1e}" for n, e in errs.items()))
