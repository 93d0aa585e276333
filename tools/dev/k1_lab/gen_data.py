#!/usr/bin/env python
"""Writes the K1 lab's inputs (tools/dev/k1_lab/data/): the plane tables {d0, delta} per pixel and the homographies of the three
config-2 stages of the bench scene (seed 0, synthetic.cascade_state_dict(0)), taken from one CPU oracle forward.  The plane tables of
stages 2 / 3 are what decides the locality of K1's gathers (noisy depth maps of random weights), so the lab uses the real ones."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, ROOT)
from rc_mvsnet_amd import synthetic
from oracle import cascade, warp
GAIN = float(os.environ.get("K1_GAIN", "20"))        # 20 = the bench scene, 1 = its smooth-head twin
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "data" if GAIN == 20 else "data_smooth")
os.makedirs(OUT, exist_ok=True)
V = int(os.environ.get("K1_V", "3"))
imgs, proj, dv = synthetic.cascade_inputs(1, V, 512, 640, 0)
sd = synthetic.cascade_state_dict(0, prob_gain=GAIN)
with torch.no_grad():
    out, aux = cascade.forward_eval(imgs, proj, dv, sd, impl="aten", return_aux=True)
for s, sc in ((1, 4), (2, 2), (3, 1)):
    key = f"stage{s}"
    smp = aux[key]["samples"]
    h, w = 512 // sc, 640 // sc
    if smp.dim() == 2:
        smp = smp.reshape(1, -1, 1, 1).expand(1, smp.shape[1], h, w)
    planes = torch.stack((smp[0, 0], smp[0, 1] - smp[0, 0]), -1).contiguous().numpy().astype(np.float32)
    planes.tofile(os.path.join(OUT, f"planes_s{s}.bin"))
    rt = []
    for v in range(1, V):
        rot, trans = warp.compose_homography(proj[key][:, v], proj[key][:, 0])
        rt.append(torch.cat((rot.reshape(-1), trans.reshape(-1))).numpy())
    np.concatenate(rt).astype(np.float32).tofile(os.path.join(OUT, f"rt_s{s}.bin"))   # per view: 9 rot + 3 trans
    print(key, planes.shape, planes[..., 0].min(), planes[..., 0].max(), planes[..., 1].mean())
