"""Developer check (GPU box), round 5: image gain 3000, fp16-pair vs bf16-triple conv forms, per cascade stage and per K1 dispatch
(production; exact two-phase kernel everywhere; production with only stage 1 / only stage 3 forced exact)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from rc_mvsnet_amd import _lib, ops, synthetic
from rc_mvsnet_amd.casmvsnet import CascadeMVSNet_eval
_lib.load()
dev = "cuda:0"
sd = synthetic.cascade_state_dict(0, prob_gain=1.0)
imgs, pm, dv = synthetic.cascade_inputs(1, 3, 64, 96, 0)
rng = float(dv[0, -1] - dv[0, 0])
orig = ops.warp_variance
def make(mode):
    def f(feats, r, t, p, D, variant=None, uniform_planes=False):
        C = feats.shape[-1]
        exact = mode == "exact" or (mode == "s1exact" and C == 32) or (mode == "s3exact" and C == 8)
        return orig(feats, r, t, p, D, variant=0) if exact else orig(feats, r, t, p, D, variant=variant, uniform_planes=uniform_planes)
    return f
gain = 3000.0
for mode in ("production", "exact", "s1exact", "s3exact"):
    ops.warp_variance = make(mode)
    outs = {}
    for pair in (True, False):
        m = CascadeMVSNet_eval(ndepths=[48, 32, 8], depth_interals_ratio=[4, 2, 1])
        m.load_state_dict(sd, strict=True)
        m = m.to(dev).eval()
        m.fp16_pair = pair
        with torch.no_grad():
            outs[pair] = m((imgs * gain).to(dev), {k: v.to(dev) for k, v in pm.items()}, dv.to(dev))
    ops.warp_variance = orig
    msg = []
    for key in ("stage1", "stage2", "stage3"):
        dd = (outs[True][key]["depth"] - outs[False][key]["depth"]).abs()
        msg.append(f"{key}: L1/range {float(dd.mean()) / rng:.2e}, off by > 0.05 mm {int((dd >= 0.05).sum())} of {dd.numel()}")
    print(f"[K1 {mode}] pair vs triple  " + "   ".join(msg))
