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


CAP = {}        # PROBE_FEATGRAD=1: the stage-1 feature maps and the gradient that arrives at them, per path


def _capture(tag):
    """Wrap the two places where the stage-1 feature maps enter the warp: oracle/aten_graph.depth_stage (list of (B,C,h,w) maps) and
    ops.WarpVarianceFn.apply ((B,V,h,w,C) tensor); their values and gradients go to CAP[tag]."""
    CAP[tag] = {}
    orig_ds, orig_fn = aten_graph.depth_stage, ops.WarpVarianceFn.apply

    def ds(model, feats, *a, **k):
        if "f" not in CAP[tag]:
            CAP[tag]["f"] = torch.stack([f.detach()[0].permute(1, 2, 0) for f in feats]).double().cpu()
            for v, f in enumerate(feats):
                f.register_hook(lambda g, v=v: CAP[tag].setdefault("g", {}).__setitem__(v, g.detach()[0].permute(1, 2, 0).double().cpu()))
        return orig_ds(model, feats, *a, **k)

    def fn(feats, *a):
        if "f" not in CAP[tag]:
            CAP[tag]["f"] = feats.detach()[0].double().cpu()
            feats.register_hook(lambda g: CAP[tag].__setitem__("g", {v: g.detach()[0, v].double().cpu() for v in range(g.shape[1])}))
        return orig_fn(feats, *a)

    # ... and where the variance volume enters the 3-D U-Net: aten_graph.unet3d (B,C,D,h,w) / CostRegNet.features_cl_train (B,D,h,w,C)
    from rc_mvsnet_amd.casmvsnet import CostRegNet
    orig_un, orig_ft = aten_graph.unet3d, CostRegNet.features_cl_train

    def un(cr, variance, *a, **k):
        if "v" not in CAP[tag]:
            CAP[tag]["v"] = variance.detach()[0].permute(1, 2, 3, 0).double().cpu()
            variance.register_hook(lambda g: CAP[tag].__setitem__("gv", g.detach()[0].permute(1, 2, 3, 0).double().cpu()))
        return orig_un(cr, variance, *a, **k)

    def ft(self, var):
        if "v" not in CAP[tag]:
            CAP[tag]["v"] = var.detach()[0].double().cpu()
            var.register_hook(lambda g: CAP[tag].__setitem__("gv", g.detach()[0].double().cpu()))
        return orig_ft(self, var)

    aten_graph.depth_stage, ops.WarpVarianceFn.apply, aten_graph.unet3d, CostRegNet.features_cl_train = ds, fn, un, ft
    return lambda: (setattr(aten_graph, "depth_stage", orig_ds), setattr(ops.WarpVarianceFn, "apply", orig_fn),
                    setattr(aten_graph, "unet3d", orig_un), setattr(CostRegNet, "features_cl_train", orig_ft))


def run(model, forward, device, dtype, tag=None):
    restore = _capture(tag) if (tag and os.environ.get("PROBE_FEATGRAD")) else None
    try:
        return _run(model, forward, device, dtype)
    finally:
        if restore: restore()


def _run(model, forward, device, dtype):
    model = copy.deepcopy(model).to(device=device, dtype=dtype).train()
    i, d = imgs.to(device=device, dtype=dtype), dv.to(device=device, dtype=dtype)
    p = {k: v.to(device=device, dtype=dtype) for k, v in pm.items()}
    out, noref = forward(model, i, p, d)
    # PROBE_LOSS = both (the fixture's loss) | depth | noref: which term's gradient path carries an error
    which = os.environ.get("PROBE_LOSS", "both")
    loss = 0.0
    if which in ("both", "depth"): loss = loss + ((out["stage1"]["depth"] - 600.0) ** 2).mean() / 1e4
    if which in ("both", "noref"): loss = loss + 1e-2 * (noref ** 2).mean()
    if which == "noref_var": loss = 1e-2 * (noref[:, -32:] ** 2).mean()          # the source-only variance channels alone
    loss.backward()
    return float(loss), {n: q.grad.detach().double().cpu() for n, q in model.named_parameters()
                         if q.grad is not None and (n.startswith("cost_regularization.0") or n.startswith("feature"))}


t = time.time()
l64, g64 = run(m0, aten_graph.cascade_forward, "cpu", torch.float64, tag="fp64")
print(f"fp64 reference graph on CPU: loss {l64:.9f} ({time.time() - t:.0f} s)")
res = {}
if dev != "cpu":
    _lib.load()
    res["ref fp32 (GPU)"] = run(m0, aten_graph.cascade_forward, dev, torch.float32, tag="ref32")
    res["hip x3"] = run(m0, lambda m, *a: m(*a), dev, torch.float32, tag="hip")
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

if CAP:
    rel = lambda a, b: float((a - b).norm() / b.norm().clamp_min(1e-300))
    for tag in ("ref32", "hip"):
        if tag in CAP and "g" in CAP[tag]:
            print(f"{tag:6s}: stage-1 feature maps vs fp64 {rel(CAP[tag]['f'], CAP['fp64']['f']):.2e}; gradient arriving at them, per view: "
                  + ", ".join(f"{rel(CAP[tag]['g'][v], CAP['fp64']['g'][v]):.2e}" for v in sorted(CAP['fp64']['g']))
                  + (f"; variance volume vs fp64 {rel(CAP[tag]['v'], CAP['fp64']['v']):.2e}, gradient arriving at it {rel(CAP[tag]['gv'], CAP['fp64']['gv']):.2e}"
                     if "gv" in CAP[tag] and "gv" in CAP["fp64"] else ""))

    # is the error at the variance gradient spread out (arithmetic) or concentrated in a few 3x3x3 neighbourhoods (a ReLU of conv0 whose
    # pre-activation sits within one rounding error of zero takes the other branch: each such element rewrites 27 voxels x C channels)?
    for tag in ("ref32", "hip"):
        if tag in CAP and "gv" in CAP[tag]:
            d = (CAP[tag]["gv"] - CAP["fp64"]["gv"]).pow(2).sum(-1)                    # squared error per voxel (D,h,w)
            tot = float(d.sum())
            flat = d.flatten().sort(descending=True).values
            big = d > 1e-6 * float(CAP["fp64"]["gv"].pow(2).sum(-1).max())
            idx = big.nonzero()
            span = [(int(idx[:, a].min()), int(idx[:, a].max())) for a in range(3)] if len(idx) else []
            rest = float(flat[27 * 8:].sum())
            print(f"{tag:6s}: variance-gradient error: {int(big.sum())} of {d.numel()} voxels above 1e-3 of the largest voxel gradient; the 216 worst voxels carry "
                  f"{100 * float(flat[:216].sum()) / tot:.1f} % of the squared error (bounding box of the large ones {span}); relative error without them "
                  f"{(rest / float(CAP['fp64']['gv'].pow(2).sum())) ** 0.5:.2e}")
