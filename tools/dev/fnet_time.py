"""Developer probe (GPU box): HIP-event timings of FeatureNet.forward_cl (3 views 512x640, eval) and of its last FPN level alone."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from rc_mvsnet_amd import _lib, ops, synthetic
from rc_mvsnet_amd.casmvsnet import CascadeMVSNet_eval
_lib.load()
dev = "cuda:0"
m = CascadeMVSNet_eval(ndepths=[48, 32, 8], depth_interals_ratio=[4, 2, 1])
m.load_state_dict(synthetic.cascade_state_dict(0)); m = m.to(dev).eval()
f = m.feature
x = torch.rand(3, 3, 512, 640, device=dev)
p = f.hip_plan()


def timeit(fn, n=30):
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n


with torch.no_grad():
    print(f"FeatureNet.forward_cl, 3 views: {timeit(lambda: f.forward_cl(x)):8.1f} us")
    c0 = torch.randn(3, 512, 640, 8, device=dev)
    intra = torch.randn(3, 256, 320, 32, device=dev)
    print(f"fpn_out_fused (1x1 lateral + up-add + 3x3 out conv, 8 -> 32 -> 8): {timeit(lambda: ops.fpn_out_fused(c0, intra, p['inner2'][0], p['inner2'][1], p['out3'])):8.1f} us")
