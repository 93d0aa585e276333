"""Developer probe (GPU box): what do the gaps between the ~100 launches of a scene cost?  The same CascadeMVSNet_eval.forward, eager
(python -> ctypes -> hipLaunchKernel per kernel) against a captured hipGraph replayed per scene (torch.cuda.CUDAGraph)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from rc_mvsnet_amd import _lib, synthetic
from rc_mvsnet_amd.casmvsnet import CascadeMVSNet_eval
_lib.load()
dev = "cuda:0"
m = CascadeMVSNet_eval(ndepths=[48, 32, 8], depth_interals_ratio=[4, 2, 1])
m.load_state_dict(synthetic.cascade_state_dict(0)); m = m.to(dev).eval()
scenes = []
for seed in range(4):
    imgs, pm, dv = synthetic.cascade_inputs(1, 3, 512, 640, seed)
    scenes.append((imgs.to(dev), {k: v.to(dev) for k, v in pm.items()}, dv.to(dev)))
N = int(os.environ.get("STEPS", "300"))
with torch.no_grad():
    for i in range(12): out = m(*scenes[i % 4])
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(N): out = m(*scenes[i % 4])
    torch.cuda.synchronize()
    eager = (time.perf_counter() - t0) / N
    ref = [m(*s)["depth"].clone() for s in scenes]
    graphs, outs = [], []
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for s in scenes: m(*s)
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    for s in scenes:
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            o = m(*s)
        graphs.append(g); outs.append(o)
    for i in range(12): graphs[i % 4].replay()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(N): graphs[i % 4].replay()
    torch.cuda.synchronize()
    graph = (time.perf_counter() - t0) / N
    same = all(torch.equal(outs[k]["depth"], ref[k]) for k in range(4))
print(f"eager {eager * 1e3:.4f} ms/scene   hipGraph replay {graph * 1e3:.4f} ms/scene   ({eager / graph:.3f}x)   depth maps bit-identical: {same}")
