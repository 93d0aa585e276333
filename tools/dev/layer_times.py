"""Developer check (GPU box): per-layer times of the 3-D convolutions of one config-2 scene (HIP events around every launch, ops.CONV_EVENTS),
in launch order, with the layer's ideal activation traffic and what that is of the HBM peak."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from rc_mvsnet_amd import _lib, ops, synthetic
from rc_mvsnet_amd.casmvsnet import CascadeMVSNet_eval
if os.environ.get("RCMVS_LIB"):          # a variant build (tools/dev/build_*_variants.sh)
    _lib.LIB_PATH = os.path.abspath(os.environ["RCMVS_LIB"])
_lib.load()
dev = "cuda:0"
V, H, W = int(os.environ.get("V", 3)), int(os.environ.get("H", 512)), int(os.environ.get("W", 640))
m = CascadeMVSNet_eval(ndepths=[48, 32, 8], depth_interals_ratio=[4, 2, 1])
m.load_state_dict(synthetic.cascade_state_dict(0), strict=True)
m = m.to(dev).eval()
i, p, d = synthetic.cascade_inputs(1, V, H, W, 0)
scene = (i.to(dev), {k: v.to(dev) for k, v in p.items()}, d.to(dev))
N = 20
with torch.no_grad():
    for _ in range(5):
        out = m(*scene)
    torch.cuda.synchronize()
    print("lib", os.environ.get("RCMVS_LIB", "product"), "depth checksum %.10e" % float(out["depth"].double().sum()))
    ev = []
    ops.CONV_EVENTS = ev
    for _ in range(N):
        m(*scene)
    torch.cuda.synchronize()
    ops.CONV_EVENTS = None
per = len(ev) // N
tot = 0.0
for j in range(per):
    ms = sorted(ev[j + per * r][0].elapsed_time(ev[j + per * r][1]) for r in range(N))
    kind, B, D, Hh, Ww, Ci, Co = ev[j][2]
    vin = B * D * Hh * Ww
    if kind == "t2":
        vout, res = 8 * vin, 1
    elif kind == "s2":
        vout, res = B * ((D - 1) // 2 + 1) * ((Hh - 1) // 2 + 1) * ((Ww - 1) // 2 + 1), 0
    else:
        vout, res = vin, 0
    byts = 4.0 * (vin * Ci + vout * Co * (1 + res))
    us = ms[N // 2] * 1e3
    tot += us
    print(f"{j:3d} {kind} {Ci:3d}->{Co:3d} {B}x{D}x{Hh}x{Ww:<4d} {us:7.1f} us  {byts / 1e6:7.1f} MB  {byts / us * 1e-6:5.2f} TB/s")
print(f"total {tot:.1f} us per scene over {per} launches")
