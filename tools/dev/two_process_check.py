"""Developer check (GPU box): the one-stream loop in TWO PROCESSES on the same GPU at the same time (START = wall-clock time both begin).
Separate address spaces, same CUs / caches / LDS: corruption here = concurrency on the hardware, none = something inside one process."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from rc_mvsnet_amd import _lib, synthetic
from rc_mvsnet_amd.casmvsnet import CascadeMVSNet_eval
_lib.load()
dev = "cuda:0"
scenes = []
for seed in range(4):
    imgs, pm, dv = synthetic.cascade_inputs(1, 3, 512, 640, seed)
    scenes.append((imgs.to(dev), {k: v.to(dev) for k, v in pm.items()}, dv.to(dev)))
m = CascadeMVSNet_eval(ndepths=[48, 32, 8], depth_interals_ratio=[4, 2, 1])
m.load_state_dict(synthetic.cascade_state_dict(0)); m = m.to(dev).eval()
NS = int(os.environ.get("SCENES", "400"))
with torch.no_grad():
    want = [m(*s)["depth"].clone() for s in scenes]
    want2 = [m(*s)["depth"].clone() for s in scenes]
    torch.cuda.synchronize()
    alone = all(torch.equal(a, b) for a, b in zip(want, want2))
    while time.time() < float(os.environ["START"]): pass
    t0 = time.perf_counter()
    got = [m(*scenes[i % 4])["depth"] for i in range(NS)]
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    w = [i for i, o in enumerate(got) if not torch.equal(o, want[i % 4])]
print(f"process {os.environ.get('TAG', '?')}: reference repeatable while warming up: {alone}; {len(w)} of {NS} scenes differ ({w[:12]}...); {dt * 1e3 / NS:.3f} ms/scene; window {t0:.2f}..{t0 + dt:.2f}")
