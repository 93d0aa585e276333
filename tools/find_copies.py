import sys, warnings, torch
warnings.simplefilter("ignore")
sys.path.insert(0, ".")
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
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
        m(imgs, pm, dv); torch.cuda.synchronize()
for e in prof.events():
    n = e.name
    if ("copy_" in n or "Memcpy" in n or "clone" in n or "contiguous" in n) and e.cpu_parent is not None and e.cpu_parent.name.startswith("aten::") is False:
        pass
import collections
c = collections.Counter()
for e in prof.events():
    if e.name in ("aten::copy_", "aten::clone", "aten::contiguous", "aten::to", "aten::_to_copy", "aten::cat", "aten::select", "aten::index"):
        st = [s for s in (e.stack or []) if "rc_mvsnet_amd" in s or "rc_mvsnet_amd" in s]
        c[(e.name, st[0] if st else "?")] += 1
for k, v in c.most_common(30): print(v, k)
