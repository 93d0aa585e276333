"""Hunt for stray device copies / ATen glue kernels in one CascadeMVSNet_eval.forward (GPU box): every aten op that launches a
device kernel or a DtoD copy is listed with its input shapes and the package line that called it."""
import sys, os, collections
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
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True, record_shapes=True) as prof:
        m(imgs, pm, dv)
        torch.cuda.synchronize()
agg = collections.OrderedDict()
for e in prof.events():
    if not e.name.startswith("aten::"): continue
    dt = getattr(e, "device_time_total", None) or getattr(e, "cuda_time_total", 0)
    if not dt: continue
    site = next((s for s in (e.stack or []) if "rc_mvsnet_amd" in s), "?")
    key = (e.name, str(e.input_shapes)[:90], site.strip()[-80:])
    a = agg.setdefault(key, [0, 0.0]); a[0] += 1; a[1] += dt
for (name, shapes, site), (n, dt) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:50]:
    print(f"{dt:8.1f} us x{n:3d} {name:28s} {shapes:90s} {site}")
print(prof.key_averages().table(sort_by="cuda_time_total", row_limit=14, max_name_column_width=60))
