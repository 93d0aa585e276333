"""Developer check (GPU box): which intermediate of a scene first differs when two scenes run on two streams (tools/dev/two_stream_check.py)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from rc_mvsnet_amd import _lib, ops, synthetic, casmvsnet
from rc_mvsnet_amd.casmvsnet import CascadeMVSNet_eval
from rc_mvsnet_amd.scene_pipeline import ScenePipeline
_lib.load()
dev = "cuda:0"
scenes = []
for seed in range(4):
    imgs, pm, dv = synthetic.cascade_inputs(1, 3, 512, 640, seed)
    scenes.append((imgs.to(dev), {k: v.to(dev) for k, v in pm.items()}, dv.to(dev)))
LOG = None
real_wv, real_dh, real_fc, real_hp = ops.warp_variance, ops.depth_head, casmvsnet.CostRegNet.features_cl, ops.hypothesis_planes


def wv(feats, rot, trans, planes, D, *a, **k):
    out = real_wv(feats, rot, trans, planes, D, *a, **k)
    if LOG is not None: LOG.append(("feats", feats.clone())); LOG.append(("planes", planes.clone())); LOG.append(("var", out.clone()))
    return out


def fc(self, x, *a, **k):
    out = real_fc(self, x, *a, **k)
    if LOG is not None: LOG.append(("x8", out.clone()))
    return out


def dh(x8, w, planes, *a, **k):
    out = real_dh(x8, w, planes, *a, **k)
    if LOG is not None: LOG.append(("depth", out[0].clone())); LOG.append(("conf", out[1].clone()))
    return out


ops.warp_variance, ops.depth_head, casmvsnet.CostRegNet.features_cl = wv, dh, fc


def make():
    m = CascadeMVSNet_eval(ndepths=[48, 32, 8], depth_interals_ratio=[4, 2, 1])
    m.load_state_dict(synthetic.cascade_state_dict(0)); return m.to(dev).eval()


with torch.no_grad():
    ref = make()
    want = []
    for s in scenes:
        LOG = []; ref(*s); want.append(LOG)
    LOG = None
    torch.cuda.synchronize()
    for rnd in range(4):
        pipe = ScenePipeline(make, 2, dev)
        logs = []
        for i in range(12):
            LOG = []; pipe(*scenes[i % 4]); logs.append(LOG)
        LOG = None
        pipe.synchronize()
        bad = 0
        for i, lg in enumerate(logs):
            for (name, t), (n2, w) in zip(lg, want[i % 4]):
                d = float((t - w).abs().max())
                if d != 0.0:
                    idx = [j for j, (nm, _) in enumerate(lg) if nm == name]
                    print(f"round {rnd} scene {i}: first difference at '{name}' (entry {lg.index((name, t)) if False else ''}) max |d| = {d:.4g}, "
                          f"fraction of elements differing = {float(((t - w) != 0).float().mean()):.5f}, order so far: {[nm for nm, _ in lg[:lg.index(next(x for x in lg if x[1] is t)) + 1]][-4:]}")
                    bad += 1
                    break
        print(f"round {rnd}: {bad} of 12 scenes differ")
