"""Developer probe: which path is closer to an fp64 evaluation of the reference op graph -- the HIP training path with the
split-bf16 kernels, the same path on the fp32 FMA-chain kernels only, or the reference graph in fp32 on the GPU?
Prints per-parameter relative Frobenius errors of the stage-1 / feature-pyramid gradients against fp64 (CPU)."""
import copy, sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from oracle import aten_graph
from rc_mvsnet_amd import _lib, ops, synthetic
from rc_mvsnet_amd.casmvsnet import CascadeMVSNet

H, W = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (128, 160)
dev = "cuda:0" if torch.cuda.is_available() else "cpu"
m0 = CascadeMVSNet(ndepths=[16, 16, 8], depth_interals_ratio=[4, 2, 1], grad_method="detach")
m0.load_state_dict(synthetic.cascade_state_dict(0, prob_gain=1.0), strict=True)
imgs, pm, dv = synthetic.cascade_inputs(1, 3, H, W, 0)


def run(model, forward, device, dtype):
    model = copy.deepcopy(model).to(device=device, dtype=dtype).train()
    i, d = imgs.to(device=device, dtype=dtype), dv.to(device=device, dtype=dtype)
    p = {k: v.to(device=device, dtype=dtype) for k, v in pm.items()}
    out, noref = forward(model, i, p, d)
    loss = ((out["stage1"]["depth"] - 600.0) ** 2).mean() / 1e4 + 1e-2 * (noref ** 2).mean()
    loss.backward()
    return float(loss), {n: q.grad.detach().double().cpu() for n, q in model.named_parameters()
                         if q.grad is not None and (n.startswith("cost_regularization.0") or n.startswith("feature"))}


t = time.time()
l64, g64 = run(m0, aten_graph.cascade_forward, "cpu", torch.float64)
print(f"fp64 reference graph on CPU: loss {l64:.9f} ({time.time() - t:.0f} s)")
res = {}
if dev != "cpu":
    _lib.load()
    res["ref fp32 (GPU)"] = run(m0, aten_graph.cascade_forward, dev, torch.float32)
    res["hip x3"] = run(m0, lambda m, *a: m(*a), dev, torch.float32)
    ops.force_direct_conv(64)
    res["hip fp32 kernels"] = run(m0, lambda m, *a: m(*a), dev, torch.float32)
    ops.force_direct_conv(0)
else:
    res["ref fp32 (CPU)"] = run(m0, aten_graph.cascade_forward, "cpu", torch.float32)
for name, (l, g) in res.items():
    errs = {n: float((g[n] - g64[n]).norm() / g64[n].norm().clamp_min(1e-300)) for n in g64}
    worst = sorted(errs, key=errs.get)[-3:]
    vals = sorted(errs.values())
    if os.environ.get("PROBE_ALL"):
        for n in g64:
            if n.startswith("feature"): print(f"      {name:18s} {n:40s} {errs[n]:.2e}  |g| {float(g64[n].norm()):.3e}")
    print(f"{name:18s}: loss err {abs(l - l64) / abs(l64):.2e}; grad err vs fp64 median {vals[len(vals) // 2]:.2e}; worst " + ", ".join(f"{n} {errs[n]:.2e}" for n in reversed(worst)))
