"""Developer probe (GPU box): throughput of independent scenes issued round-robin on S HIP streams (one model replica per stream:
the fp16-pair bound buffer is per model) against the single-stream loop of bench.py.  Scenes are independent reference views."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from rc_mvsnet_amd import _lib, synthetic
from rc_mvsnet_amd.casmvsnet import CascadeMVSNet_eval
_lib.load()
dev = "cuda:0"
N = int(os.environ.get("STEPS", "300"))
scenes = []
for seed in range(4):
    imgs, pm, dv = synthetic.cascade_inputs(1, 3, 512, 640, seed)
    scenes.append((imgs.to(dev), {k: v.to(dev) for k, v in pm.items()}, dv.to(dev)))


def make():
    m = CascadeMVSNet_eval(ndepths=[48, 32, 8], depth_interals_ratio=[4, 2, 1])
    m.load_state_dict(synthetic.cascade_state_dict(0)); return m.to(dev).eval()


for S in (1, 2, 3):
    models = [make() for _ in range(S)]
    streams = [torch.cuda.Stream() for _ in range(S)]
    outs = [None] * S
    with torch.no_grad():
        for rep in range(2):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for i in range(N if rep else 12 * S):
                k = i % S
                with torch.cuda.stream(streams[k]):
                    outs[k] = models[k](*scenes[i % 4])
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / N
        ref = models[0](*scenes[0])["depth"]
        torch.cuda.synchronize()
    print(f"{S} stream(s): {dt * 1e3:.4f} ms/scene  ({1.0 / dt:.1f} ref-scenes/s)")
