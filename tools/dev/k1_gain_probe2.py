"""Developer check (GPU box), round 5: stage-1 variance volumes of the two K1 forms at image gain 3000 and what the fp16-pair conv0 makes of them."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from rc_mvsnet_amd import _lib, ops, synthetic
from rc_mvsnet_amd.casmvsnet import CascadeMVSNet_eval
_lib.load()
dev = "cuda:0"
sd = synthetic.cascade_state_dict(0, prob_gain=1.0)
imgs, pm, dv = synthetic.cascade_inputs(1, 3, 64, 96, 0)
orig = ops.warp_variance
cap = {}
def wrap(tag, force):
    def f(feats, r, t, p, D, variant=None, uniform_planes=False):
        v = orig(feats, r, t, p, D, variant=0) if force else orig(feats, r, t, p, D, variant=variant, uniform_planes=uniform_planes)
        cap.setdefault(tag, []).append((feats.clone(), v.clone()))
        return v
    return f
for gain in (3000.0,):
    for k1 in ("production", "exact"):
        ops.warp_variance = wrap(k1, k1 == "exact")
        m = CascadeMVSNet_eval(ndepths=[48, 32, 8], depth_interals_ratio=[4, 2, 1])
        m.load_state_dict(sd, strict=True)
        m = m.to(dev).eval()
        m.fp16_pair = True
        with torch.no_grad():
            m((imgs * gain).to(dev), {k: v.to(dev) for k, v in pm.items()}, dv.to(dev))
    ops.warp_variance = orig
    for s in range(3):
        fp, vp = cap["production"][s]; fe, ve = cap["exact"][s]
        d = (vp - ve).abs()
        print(f"stage {s+1}: feats equal {bool(torch.equal(fp, fe))}  max f^2 {float(fp.abs().max())**2:.4e}  var max prod {float(vp.max()):.4e} exact {float(ve.max()):.4e}  min prod {float(vp.min()):.3e} exact {float(ve.min()):.3e}"
              f"  max|d| {float(d.max()):.3e}  mean|d| {float(d.mean()):.3e}  negatives prod {int((vp < 0).sum())} exact {int((ve < 0).sum())}  nonfinite {int((~torch.isfinite(vp)).sum())}")
