"""Developer check (GPU box): two-stream corruption with EVERY tensor the ops layer allocates kept alive for the whole round (no block is
ever handed out twice): clean here = a memory-reuse hazard between the streams; still wrong = no reuse involved."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from rc_mvsnet_amd import _lib, ops, synthetic, casmvsnet
from rc_mvsnet_amd.casmvsnet import CascadeMVSNet_eval
from rc_mvsnet_amd.scene_pipeline import ScenePipeline
_lib.load()
dev = "cuda:0"
KEEP = []
NR = int(os.environ.get("ROUNDS", "8"))
real_empty, real_zeros = torch.empty, torch.zeros
DELAY = int(os.environ.get("DELAY", "0"))
if os.environ.get("KEEPALIVE", "1") in ("1", "2"):
    def empty(*a, **k):
        t = real_empty(*a, **k); KEEP.append(t); return t

    def zeros(*a, **k):
        t = real_zeros(*a, **k); KEEP.append(t); return t
    torch.empty, torch.zeros = empty, zeros
scenes = []
for seed in range(4):
    imgs, pm, dv = synthetic.cascade_inputs(1, 3, 512, 640, seed)
    scenes.append((imgs.to(dev), {k: v.to(dev) for k, v in pm.items()}, dv.to(dev)))


def make():
    m = CascadeMVSNet_eval(ndepths=[48, 32, 8], depth_interals_ratio=[4, 2, 1])
    m.load_state_dict(synthetic.cascade_state_dict(0)); return m.to(dev).eval()


import warnings; warnings.simplefilter("ignore")
with torch.no_grad():
    ref = make()
    want = [ref(*s)["depth"].clone() for s in scenes]
    torch.cuda.synchronize(); KEEP.clear()
    bad = 0
    for rnd in range(NR):
        pipe = ScenePipeline(make, 2, dev)
        got = []
        hist = []
        for i in range(12):
            got.append(pipe(*scenes[i % 4])[0]["depth"])
            if DELAY:                                   # KEEPALIVE=2: a scene's tensors are released DELAY scenes later (no hipMalloc in steady state)
                hist.append(list(KEEP)); KEEP.clear()
                if len(hist) > DELAY: hist.pop(0)
        pipe.synchronize()
        bad += any(not torch.equal(o, want[i % 4]) for i, o in enumerate(got))
        n = len(KEEP); KEEP.clear()
print(f"KEEPALIVE={os.environ.get('KEEPALIVE', '1')} ({n} tensors held per round): {bad} of {NR} rounds corrupted")
