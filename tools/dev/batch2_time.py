"""Developer check (GPU box): scenes/s of CascadeMVSNet_eval.forward at batch 1 and batch 2 (config-2 shape)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from rc_mvsnet_amd import _lib, synthetic
from rc_mvsnet_amd.casmvsnet import CascadeMVSNet_eval
_lib.load()
dev = "cuda:0"
m = CascadeMVSNet_eval(ndepths=[48, 32, 8], depth_interals_ratio=[4, 2, 1])
m.load_state_dict(synthetic.cascade_state_dict(0), strict=True)
m = m.to(dev).eval()
for B in (1, 2, 4):
    i, p, d = synthetic.cascade_inputs(B, 3, 512, 640, 0)
    scene = (i.to(dev), {k: v.to(dev) for k, v in p.items()}, d.to(dev))
    with torch.no_grad():
        for _ in range(10):
            m(*scene)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        N = 200 // B
        for _ in range(N):
            m(*scene)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
    print(f"B={B}: {N * B / dt:.1f} scenes/s, {dt / N * 1e3:.3f} ms per forward")
