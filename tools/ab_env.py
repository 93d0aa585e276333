#!/usr/bin/env python
"""A/B an environment switch of the inference path inside one process: python tools/ab_env.py RCMVS_OVERLAP 1 0"""
import os, sys, time, warnings
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from rc_mvsnet_amd import _lib, synthetic
from rc_mvsnet_amd.casmvsnet import CascadeMVSNet_eval
warnings.simplefilter("ignore")
_lib.load()
dev = "cuda:0"
name, vals = sys.argv[1], sys.argv[2:]
m = CascadeMVSNet_eval(ndepths=[48, 32, 8], depth_interals_ratio=[4, 2, 1])
m.load_state_dict(synthetic.cascade_state_dict(0)); m = m.to(dev).eval()
imgs, pm, dv = synthetic.cascade_inputs(1, 3, 512, 640, 0)
imgs, dv = imgs.to(dev), dv.to(dev); pm = {k: v.to(dev) for k, v in pm.items()}
res = {v: [] for v in vals}
outs = {}
with torch.no_grad():
    for rep in range(4):
        for v in vals:
            os.environ[name] = v
            for mod in m.modules():                        # execution plans read the environment when they are built
                if getattr(mod, "_plan", None) is not None:
                    mod._plan = None
            for _ in range(3): o = m(imgs, pm, dv)
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(30): o = m(imgs, pm, dv)
            torch.cuda.synchronize(); res[v].append((time.perf_counter() - t0) / 30 * 1e3)
            outs[v] = o["depth"].clone()
for v in vals: print(f"{name}={v}: " + " ".join(f"{t:.3f}" for t in res[v]) + f"  -> best {min(res[v]):.3f} ms = {1e3 / min(res[v]):.1f} scenes/s")
print("outputs identical:", all(torch.equal(outs[vals[0]], outs[v]) for v in vals))
