"""Developer check (GPU box): the fused conv11 + prob kernel alone on the three stage shapes of a config-2 scene (random data), us per launch;
RCMVS_LIB selects a timing-ablation build (tools/dev/build_c11_variants.sh), ZC a z chunk."""
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
zc = int(os.environ.get("ZC", "0"))
out = []
for name, (Dt, Ht, Wt) in (("S1", (24, 64, 80)), ("S2", (16, 128, 160)), ("S3", (4, 256, 320))):
    t = torch.randn(1, Dt, Ht, Wt, 16, device=dev)
    r = torch.randn(1, 2 * Dt, 2 * Ht, 2 * Wt, 8, device=dev)
    tm, rm = ops.absmax(t), ops.absmax(r)
    f = lambda: ops.conv11_prob(t, tm, plan["conv11"][0], plan["conv11"][1], plan["conv11"][2], r, rm, plan["conv11_coef"], plan["prob"], zchunk=zc)
    for _ in range(3):
        f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        f()
    e1.record()
    torch.cuda.synchronize()
    out.append(f"{name} {e0.elapsed_time(e1) * 50:.1f} us")
print(os.environ.get("RCMVS_LIB", "product"), "zc", zc, " ".join(out))
