"""Developer check (GPU box): full-size scenes on 2 streams (eager, and hipGraph replay) against the one-stream outputs, per output tensor."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from rc_mvsnet_amd import _lib, synthetic
from rc_mvsnet_amd.casmvsnet import CascadeMVSNet_eval
from rc_mvsnet_amd.scene_pipeline import ScenePipeline
_lib.load()
dev = "cuda:0"
scenes = []
for seed in range(4):
    imgs, pm, dv = synthetic.cascade_inputs(1, 3, 512, 640, seed)
    scenes.append((imgs.to(dev), {k: v.to(dev) for k, v in pm.items()}, dv.to(dev)))


def make():
    m = CascadeMVSNet_eval(ndepths=[48, 32, 8], depth_interals_ratio=[4, 2, 1])
    m.load_state_dict(synthetic.cascade_state_dict(0)); return m.to(dev).eval()


def diff(a, b):
    return {k: float((a[k] - b[k]).abs().max()) for k in ("depth", "photometric_confidence")} | {s: float((a[s]["depth"] - b[s]["depth"]).abs().max()) for s in ("stage1", "stage2", "stage3")}


with torch.no_grad():
    ref = make()
    want = [{k: (v.clone() if torch.is_tensor(v) else {kk: vv.clone() for kk, vv in v.items()}) for k, v in ref(*s).items()} for s in scenes]
    again = ref(*scenes[0]); torch.cuda.synchronize()
    print("one stream, same model twice:", diff(again, want[0]))
    other = make()(*scenes[0]); torch.cuda.synchronize()
    print("one stream, second replica  :", diff(other, want[0]))
    for rnd in range(int(os.environ.get("ROUNDS", "3"))):
        pipe = ScenePipeline(make, 2, dev)
        got = [pipe(*scenes[i % 4]) for i in range(16)]
        pipe.synchronize()
        worst = {}
        for i, (o, _) in enumerate(got):
            for k, v in diff(o, want[i % 4]).items(): worst[k] = max(worst.get(k, 0.0), v)
        print(f"two streams eager, round {rnd}: worst |diff| over 16 scenes:", worst)
