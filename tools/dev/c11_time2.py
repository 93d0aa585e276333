"""Developer check (GPU box): per-block vs per-step cost of the fused conv11 + prob kernel: full-resolution plane 512x640 with 1, 2, 4, 8, 16 cell planes."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from rc_mvsnet_amd import _lib, ops
if os.environ.get("RCMVS_LIB"):
    _lib.LIB_PATH = os.path.abspath(os.environ["RCMVS_LIB"])
_lib.load()
from rc_mvsnet_amd.casmvsnet import CostRegNet
dev = "cuda:0"
net = CostRegNet(8, 8).to(dev).eval()
plan = net.hip_plan()
out = []
for Dt in (1, 2, 4, 8, 16):
    Ht, Wt = 256, 320
    t = torch.randn(1, Dt, Ht, Wt, 16, device=dev)
    r = torch.randn(1, 2 * Dt, 2 * Ht, 2 * Wt, 8, device=dev)
    tm, rm = ops.absmax(t), ops.absmax(r)
    f = lambda: ops.conv11_prob(t, tm, plan["conv11"][0], plan["conv11"][1], plan["conv11"][2], r, rm, plan["conv11_coef"], plan["prob"], zchunk=2 * Dt)
    for _ in range(3):
        f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        f()
    e1.record()
    torch.cuda.synchronize()
    out.append(f"Dt={Dt}: {e0.elapsed_time(e1) * 50:.1f} us")
print(os.environ.get("RCMVS_LIB", "product"), " ".join(out))
