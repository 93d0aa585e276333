"""Developer check (GPU box), round 5: does the LDS-window K1 kernel of stage 1 change the two-streams-in-one-process behaviour?
24 config-2 scenes alternately on two HIP streams (one replica each) against the one-stream outputs, with the stage-1 hint on and off."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
os.environ.setdefault("RCMVS_ALLOW_MULTI_STREAM", "1")      # (ops._stream() refuses a second stream otherwise: this script studies exactly that)
import torch
from rc_mvsnet_amd import _lib, ops, synthetic
from rc_mvsnet_amd.casmvsnet import CascadeMVSNet_eval
_lib.load()
dev = "cuda:0"

def make():
    m = CascadeMVSNet_eval(ndepths=[48, 32, 8], depth_interals_ratio=[4, 2, 1])
    m.load_state_dict(synthetic.cascade_state_dict(0), strict=True)
    return m.to(dev).eval()

scenes = []
for seed in range(4):
    i, p, d = synthetic.cascade_inputs(1, 3, 512, 640, seed)
    scenes.append((i.to(dev), {k: v.to(dev) for k, v in p.items()}, d.to(dev)))
keys = [("stage1", "depth"), ("stage2", "depth"), ("depth",), ("photometric_confidence",)]
pick = lambda o, k: o[k[0]] if len(k) == 1 else o[k[0]][k[1]]
for hint in (1, 0, 1, 0):
    ops.K1_UNIFORM_PLANES = hint
    for sync_after_warmup in (False, True):
        with torch.no_grad():
            one = make()
            want = [[pick(one(*s), k).clone() for k in keys] for s in scenes]
            torch.cuda.synchronize()
            models = [make(), make()]
            streams = [torch.cuda.Stream(), torch.cuda.Stream()]
            got = []
            for i in range(2 + 24):
                with torch.cuda.stream(streams[i % 2]):
                    o = models[i % 2](*scenes[i % 4])
                if i == 1 and sync_after_warmup:
                    torch.cuda.synchronize()
                if i >= 2:
                    got.append((i, o))
            torch.cuda.synchronize()
        bad = []
        for i, o in got:
            for k, w in zip(keys, want[i % 4]):
                if not torch.equal(pick(o, k), w):
                    d = (pick(o, k) - w).abs()
                    bad.append((i, "/".join(k), float(d.max()), float((d > 0).float().mean())))
                    break
        print(f"hint {hint} sync-after-warm-up {sync_after_warmup}: {len(bad)} of {len(got)} scenes differ; first differing tensor per scene: {bad[:6]}")
