"""Deep U-Net levels (conv5 32->64 s2, conv6 64->64, conv7 64->32 transposed) per stage shape of a DTU scene:
fp32-MFMA kernel (conv3d_mfma.hip) vs the fp16-pair tile kernel (conv3d_deep.hip).  python tools/dev/deep_ab.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from rc_mvsnet_amd import ops

dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)


def timeit(f, n=200):
    for _ in range(10):
        f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        f()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


tot = {"fp32": 0.0, "pair": 0.0}
for name, (D, H, W) in (("stage1", (12, 32, 40)), ("stage2", (8, 64, 80)), ("stage3", (2, 128, 160))):
    x4 = torch.randn(1, D, H, W, 32, generator=g).to(dev)
    layers = []
    w5 = torch.randn(64, 32, 3, 3, 3, generator=g).to(dev) / (32 * 27) ** 0.5
    w6 = torch.randn(64, 64, 3, 3, 3, generator=g).to(dev) / (64 * 27) ** 0.5
    w7 = torch.randn(64, 32, 3, 3, 3, generator=g).to(dev) / (64 * 27 / 8) ** 0.5
    p5, p6, p7 = ops.pack_conv3d_weight(w5), ops.pack_conv3d_weight(w6), ops.pack_conv3d_weight(w7, transposed=True)
    sc64, sh64 = torch.rand(64, device=dev) + 0.5, torch.randn(64, device=dev) * 0.1
    sc32, sh32 = torch.rand(32, device=dev) + 0.5, torch.randn(32, device=dev) * 0.1
    b = torch.zeros(4, ops.ABSMAX_FLOATS, device=dev)
    b[0] = ops.absmax(x4)
    y5 = ops.conv3d(x4, p5, sc64, sh64, stride=2, relu=True, x_absmax=b[0], y_absmax=b[1])
    y6 = ops.conv3d(y5, p6, sc64, sh64, relu=True, x_absmax=b[1], y_absmax=b[2])
    y7 = ops.deconv3d(y6, p7, sc32, sh32, residual=x4, relu=True, x_absmax=b[2], y_absmax=b[3])
    r5 = ops.conv3d(x4, p5, sc64, sh64, stride=2, relu=True)
    r6 = ops.conv3d(r5, p6, sc64, sh64, relu=True)
    r7 = ops.deconv3d(r6, p7, sc32, sh32, residual=x4, relu=True)
    err = float((y7 - r7).abs().max() / r7.abs().max())
    assert float(b[3].max()) == float(y7.abs().max()), (float(b[3].max()), float(y7.abs().max()))
    for lname, fp32, pair in (("conv5 32->64 s2", lambda: ops.conv3d(x4, p5, sc64, sh64, stride=2, relu=True),
                               lambda: ops.conv3d(x4, p5, sc64, sh64, stride=2, relu=True, x_absmax=b[0], y_absmax=b[1])),
                              ("conv6 64->64   ", lambda: ops.conv3d(y5, p6, sc64, sh64, relu=True),
                               lambda: ops.conv3d(y5, p6, sc64, sh64, relu=True, x_absmax=b[1], y_absmax=b[2])),
                              ("conv7 64->32 t2", lambda: ops.deconv3d(y6, p7, sc32, sh32, residual=x4, relu=True),
                               lambda: ops.deconv3d(y6, p7, sc32, sh32, residual=x4, relu=True, x_absmax=b[2], y_absmax=b[3]))):
        t32, tp = timeit(fp32), timeit(pair)
        tot["fp32"] += t32; tot["pair"] += tp
        print(f"{name} {lname} in {D}x{H}x{W}: fp32-MFMA {t32:6.1f} us   fp16-pair tile kernel {tp:6.1f} us")
    print(f"{name}: chain max relative difference {err:.2e}")
print("per scene (back-to-back launches, incl. launch overhead):", {k: round(v, 1) for k, v in tot.items()})
