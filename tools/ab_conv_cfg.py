#!/usr/bin/env python
"""A/B a conv kernel selection (ops.force_direct_conv -> rcmvs_debug_conv3d_fwd impl bits) against the default inside one process (same box, same clocks):
python tools/ab_conv_cfg.py <cfg> [<cfg> ...]   -> ms per CascadeMVSNet_eval.forward at config 2 for cfg 0 and each cfg."""
import os, sys, time, warnings
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from rc_mvsnet_amd import _lib, ops, synthetic
from rc_mvsnet_amd.casmvsnet import CascadeMVSNet_eval
warnings.simplefilter("ignore")
lib = _lib.load()
dev = "cuda:0"
m = CascadeMVSNet_eval(ndepths=[48, 32, 8], depth_interals_ratio=[4, 2, 1])
m.load_state_dict(synthetic.cascade_state_dict(0)); m = m.to(dev).eval()
imgs, pm, dv = synthetic.cascade_inputs(1, 3, 512, 640, 0)
imgs, dv = imgs.to(dev), dv.to(dev); pm = {k: v.to(dev) for k, v in pm.items()}
cfgs = [0] + [int(a) for a in sys.argv[1:]]
res = {c: [] for c in cfgs}
with torch.no_grad():
    for rep in range(4):
        for c in cfgs:
            ops.force_direct_conv(c)
            for _ in range(3): m(imgs, pm, dv)
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(30): m(imgs, pm, dv)
            torch.cuda.synchronize(); res[c].append((time.perf_counter() - t0) / 30 * 1e3)
ops.force_direct_conv(0)
for c in cfgs: print(f"cfg {c:3d}: " + " ".join(f"{t:.3f}" for t in res[c]) + f"  -> best {min(res[c]):.3f} ms = {1e3 / min(res[c]):.1f} scenes/s")
