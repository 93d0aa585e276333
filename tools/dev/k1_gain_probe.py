"""Developer check (GPU box), round 5: test_cascade_fp16_pair_form_on_extreme_activation_ranges[3000] -- which arithmetic moves the depth map
at image gain 3000 (variances 1e7 x the usual, a peaked = chaotic head): the K1 form (exact two-phase kernel everywhere vs the production
dispatch with its FMA-contracted window / plane-pipelined forms) or the conv form (fp16 pair vs bf16 triple)?"""
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
for gain in (1.0, 300.0, 3000.0):
    outs = {}
    for k1 in ("production", "exact"):
        ops.warp_variance = orig if k1 == "production" else (lambda f, r, t, p, D, variant=None, uniform_planes=False: orig(f, r, t, p, D, variant=0))
        for pair in (True, False):
            m = CascadeMVSNet_eval(ndepths=[48, 32, 8], depth_interals_ratio=[4, 2, 1])
            m.load_state_dict(sd, strict=True)
            m = m.to(dev).eval()
            m.fp16_pair = pair
            with torch.no_grad():
                outs[(k1, pair)] = m((imgs * gain).to(dev), {k: v.to(dev) for k, v in pm.items()}, dv.to(dev))["depth"]
    ops.warp_variance = orig
    d = lambda a, b: float((outs[a] - outs[b]).abs().mean()) / rng
    print(f"gain {gain:g}: pair vs triple  K1 production {d(('production', True), ('production', False)):.2e}   K1 exact {d(('exact', True), ('exact', False)):.2e}   |   "
          f"K1 production vs exact  pair {d(('production', True), ('exact', True)):.2e}   triple {d(('production', False), ('exact', False)):.2e}")
