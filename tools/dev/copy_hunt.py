import sys, os, traceback
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from rc_mvsnet_amd import _lib, synthetic
from rc_mvsnet_amd.casmvsnet import CascadeMVSNet_eval
_lib.load()
dev = "cuda:0"
m = CascadeMVSNet_eval(ndepths=[48, 32, 8], depth_interals_ratio=[4, 2, 1])
m.load_state_dict(synthetic.cascade_state_dict(0)); m = m.to(dev).eval()
imgs, pm, dv = synthetic.cascade_inputs(1, 3, 512, 640, 0)
imgs, dv = imgs.to(dev), dv.to(dev); pm = {k: v.to(dev) for k, v in pm.items()}
with torch.no_grad():
    for _ in range(3): m(imgs, pm, dv)
    torch.cuda.synchronize()
    from torch.profiler import profile, ProfilerActivity
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=False) as prof:
        m(imgs, pm, dv)
        torch.cuda.synchronize()
    evs = [e for e in prof.events() if "emcpy" in e.name or "copy" in e.name.lower()]
    for e in evs[:60]:
        print(e.name, e.device_type, getattr(e, "cuda_time", None), e.input_shapes if hasattr(e, "input_shapes") else "")
    print(prof.key_averages().table(sort_by="cuda_time_total", row_limit=25))
