"""Developer check (GPU box): scenes on stream A while stream B runs FOREIGN work (torch elementwise / copy kernels on its own tensors, no
library kernel, no shared memory).  Corruption here would mean that a kernel of the scene path is unsafe next to ANY concurrent kernel."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from rc_mvsnet_amd import _lib, synthetic
from rc_mvsnet_amd.casmvsnet import CascadeMVSNet_eval
_lib.load()
dev = "cuda:0"
KIND = os.environ.get("FOREIGN", "elementwise")
scenes = []
for seed in range(4):
    imgs, pm, dv = synthetic.cascade_inputs(1, 3, 512, 640, seed)
    scenes.append((imgs.to(dev), {k: v.to(dev) for k, v in pm.items()}, dv.to(dev)))
m = CascadeMVSNet_eval(ndepths=[48, 32, 8], depth_interals_ratio=[4, 2, 1])
m.load_state_dict(synthetic.cascade_state_dict(0)); m = m.to(dev).eval()
with torch.no_grad():
    want = [m(*s)["depth"].clone() for s in scenes]
    torch.cuda.synchronize()
    sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
    a = torch.randn(64 * 1024 * 1024 // 4, device=dev); b = torch.empty_like(a)
    w = torch.randn(2048, 2048, device=dev)
    bad = 0
    for rnd in range(8):
        got = []
        for i in range(12):
            with torch.cuda.stream(sb):
                for _ in range(40):
                    if KIND == "elementwise": b.copy_(a); b.mul_(1.0001)
                    elif KIND == "matmul": w2 = w @ w
                    else: b.zero_()
            with torch.cuda.stream(sa):
                got.append(m(*scenes[i % 4])["depth"])
        torch.cuda.synchronize()
        bad += any(not torch.equal(o, want[i % 4]) for i, o in enumerate(got))
print(f"foreign work '{KIND}' on the other stream: {bad} of 8 rounds corrupted")
