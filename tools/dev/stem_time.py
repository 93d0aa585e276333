"""Developer check (GPU box): FeatureNet's fused conv0 block (csrc/conv2d_stem.hip) alone on the three images of a DTU scene; RCMVS_LIB selects a variant
build (tools/dev/build_variant.sh).  Run it under rocprofv3 --kernel-trace --stats for kernel durations (the loop here is host-bound)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from rc_mvsnet_amd import _lib
if os.environ.get("RCMVS_LIB", "product") != "product":
    _lib.LIB_PATH = os.path.abspath(os.environ["RCMVS_LIB"])
from rc_mvsnet_amd import ops
g = torch.Generator().manual_seed(0)
x = torch.randn(3, 3, 512, 640, generator=g).cuda()
wa, wb = (torch.randn(8, 3, 3, 3, generator=g) / 5).cuda(), (torch.randn(8, 8, 3, 3, generator=g) / 8).cuda()
sa, sb, ha, hb = (torch.rand(8, generator=g).cuda() + 0.5 for _ in range(4))
pa, img = ops.pack_conv2d_weight(wa, pad_in_to=4), ops.pack_conv2d_stem(wb)
for _ in range(200):
    y = ops.conv2d_stem(x, pa, sa, ha, img, sb, hb)
torch.cuda.synchronize()
print(os.environ.get("RCMVS_LIB", "product"), "checksum %.6e" % float(y.double().sum()))
