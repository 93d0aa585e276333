"""Per-tensor gradient mismatch of the HIP training path and of the reference op graph (oracle/aten_graph.py; both on the GPU) against the
reference-autograd golden (tests/golden/train_grads.npz)."""
import os, sys, warnings
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch
from conftest import load_golden
from rc_mvsnet_amd import _lib, synthetic
from rc_mvsnet_amd.casmvsnet import CascadeMVSNet
warnings.simplefilter("ignore")
_lib.load()
dev = "cuda:0"
g = load_golden("train_grads")
res = {}
from oracle import aten_graph
for mode in ("hip", "aten"):
    m = CascadeMVSNet(ndepths=[8, 8, 8], depth_interals_ratio=[4, 2, 1])
    m.load_state_dict(synthetic.cascade_state_dict(0, prob_gain=2.0), strict=True)
    m = m.to(dev).train()
    imgs, pm, dv = synthetic.cascade_inputs(1, 3, 64, 96, 0)
    fwd = (lambda mm, *a: mm(*a)) if mode == "hip" else aten_graph.cascade_forward
    outputs, noref = fwd(m, imgs.to(dev), {k: v.to(dev) for k, v in pm.items()}, dv.to(dev))
    loss = ((outputs["stage1"]["depth"] - 600.0) ** 2).mean() / 1e4 + 1e-2 * (noref ** 2).mean()
    loss.backward()
    params = dict(m.named_parameters())
    res[mode] = {k[5:]: params[k[5:]].grad.cpu() for k in g if k.startswith("grad:")}
    res[mode]["_noref"] = float((noref ** 2).mean())
    res[mode]["_d1"] = outputs["stage1"]["depth"].detach().cpu()
rel = lambda a, b: float((a.double() - b.double()).abs().max() / b.double().abs().max().clamp_min(1e-30))
print("noref mean sq: hip", res["hip"]["_noref"], "aten", res["aten"]["_noref"], "ref", float(g["noref_mean_sq"]))
print("depth1 max abs diff: hip", float((res["hip"]["_d1"] - g["depth1"]).abs().max()), "aten", float((res["aten"]["_d1"] - g["depth1"]).abs().max()))
for k in g:
    if k.startswith("grad:"):
        r = torch.as_tensor(g[k])
        print(f"{k[5:]:48s} hip vs ref {rel(res['hip'][k[5:]], r):.2e}   aten(gpu) vs ref {rel(res['aten'][k[5:]], r):.2e}   hip vs aten {rel(res['hip'][k[5:]], res['aten'][k[5:]]):.2e}")
