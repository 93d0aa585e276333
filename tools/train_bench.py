#!/usr/bin/env python
"""Time one training iteration (BASELINE config 3 shape: V=4, 512x640, D=48/32/8, 1024 rays x 128 samples) on the HIP
training path ("hip") and, for comparison, with the two network forwards swapped for the reference's op graph on
PyTorch-ROCm ("aten": oracle/aten_graph.py; the losses stay on the HIP kernels).  GPU box only."""
import os, sys, time, warnings
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from rc_mvsnet_amd import train_step as ts, _lib

_lib.load()
warnings.simplefilter("ignore")
dev = torch.device("cuda:0")
H, W = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (512, 640)
modes = sys.argv[3].split(",") if len(sys.argv) > 3 else ("hip", "aten")
for mode in modes:
    kw = {}
    if mode == "aten":
        from oracle import aten_graph
        S = 128
        def render(m, vf, pd, batch):
            Hh, Ww = batch["imgs"].shape[-2:]
            pix = torch.stack((torch.randint(0, Ww, (1024,), device=dev), torch.randint(0, Hh, (1024,), device=dev)))
            return aten_graph.render_forward(m, vf, pd, batch, (pix, torch.randn(1024, S, device=dev), torch.rand(512, S, device=dev)))
        kw = dict(cascade_fn=aten_graph.cascade_forward, render_fn=render)
    model, model_nerf, opt = ts.build(dev)
    imgs, proj, dv, batch = ts.synthetic_sample(dev, H=H, W=W, V=4)
    for _ in range(2):
        l = ts.train_step(model, model_nerf, opt, imgs, proj, dv, batch, **kw)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 3
    for _ in range(n):
        l = ts.train_step(model, model_nerf, opt, imgs, proj, dv, batch, **kw)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n
    print(f"{mode}: {dt * 1e3:8.1f} ms / iteration   loss {l['loss']:.4f}   peak mem {torch.cuda.max_memory_allocated() / 2**30:.1f} GiB", flush=True)
    del model, model_nerf, opt
    torch.cuda.empty_cache()
    torch.cuda.reset_peak_memory_stats()
