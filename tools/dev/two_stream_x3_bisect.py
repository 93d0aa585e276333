"""Developer check (GPU box): does the two-stream corruption follow the split-operand kernel of stage 3's conv0 (8 -> 8, XT map)?
MODE=0 unchanged; 1: that layer on the non-x3 kernels; 2: every 3-D conv on the non-x3 kernels; 3: only K1 stage 3 on its reference-order variant."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from rc_mvsnet_amd import _lib, ops, synthetic
from rc_mvsnet_amd.casmvsnet import CascadeMVSNet_eval
from rc_mvsnet_amd.scene_pipeline import ScenePipeline
_lib.load()
dev = "cuda:0"
MODE = int(os.environ.get("MODE", "0"))
real_conv, real_wv = ops.conv3d, ops.warp_variance


def conv(x, w, *a, **k):
    hit = (MODE == 2) or (MODE == 1 and x.shape[-1] == 8 and w.co == 8 and k.get("stride", 1) == 1 and x.shape[1] > 1)
    if not hit:
        return real_conv(x, w, *a, **k)
    k.pop("x_absmax", None); k.pop("y_absmax", None)
    ops._CONV_IMPL = 64
    try:
        return real_conv(x, w, *a, **k)
    finally:
        ops._CONV_IMPL = 0


def wv(feats, rot, trans, planes, D, variant=0):
    if MODE == 3 and feats.shape[-1] == 8:
        return real_wv(feats, rot, trans, planes, D, variant=2)
    return real_wv(feats, rot, trans, planes, D, variant=variant)


ops.conv3d, ops.warp_variance = conv, wv
os.environ["RCMVS_FP16_PAIR"] = "0"          # (the bound rows would be left unset by the rerouted layers)
scenes = []
for seed in range(4):
    imgs, pm, dv = synthetic.cascade_inputs(1, 3, 512, 640, seed)
    scenes.append((imgs.to(dev), {k: v.to(dev) for k, v in pm.items()}, dv.to(dev)))


def make():
    m = CascadeMVSNet_eval(ndepths=[48, 32, 8], depth_interals_ratio=[4, 2, 1])
    m.load_state_dict(synthetic.cascade_state_dict(0)); return m.to(dev).eval()


with torch.no_grad():
    ref = make()
    want = [ref(*s)["depth"].clone() for s in scenes]
    torch.cuda.synchronize()
    bad_rounds = 0
    for rnd in range(8):
        pipe = ScenePipeline(make, 2, dev)
        got = [pipe(*scenes[i % 4])[0]["depth"] for i in range(16)]
        pipe.synchronize()
        bad_rounds += any(not torch.equal(o, want[i % 4]) for i, o in enumerate(got))
print(f"MODE={MODE}: {bad_rounds} of 8 rounds corrupted")
