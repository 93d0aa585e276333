import os, sys, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
os.environ["RCMVS_DEEP_DBG"] = os.environ.get("RCMVS_DEEP_DBG", "4")
import torch
from rc_mvsnet_amd import ops, _lib
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
lib = ctypes.CDLL(_lib.LIB_PATH)
for name, (D, H, W) in (("stage1", (12, 32, 40)), ("stage3", (2, 128, 160))):
    x4 = torch.randn(1, D, H, W, 32, generator=g).to(dev)
    w5 = torch.randn(64, 32, 3, 3, 3, generator=g).to(dev) / (32 * 27) ** 0.5
    w6 = torch.randn(64, 64, 3, 3, 3, generator=g).to(dev) / (64 * 27) ** 0.5
    w7 = torch.randn(64, 32, 3, 3, 3, generator=g).to(dev) / (64 * 27 / 8) ** 0.5
    p5, p6, p7 = ops.pack_conv3d_weight(w5), ops.pack_conv3d_weight(w6), ops.pack_conv3d_weight(w7, transposed=True)
    b = torch.zeros(4, ops.ABSMAX_FLOATS, device=dev)
    b[0] = ops.absmax(x4)
    y5 = ops.conv3d(x4, p5, stride=2, relu=True, x_absmax=b[0], y_absmax=b[1])
    y6 = ops.conv3d(y5, p6, relu=True, x_absmax=b[1], y_absmax=b[2])
    big = torch.randn(64 << 20, device=dev)
    def trace(f, label):
        for rep in range(3):
            big.mul_(1.0001)          # evict caches / instruction cache with other work
            torch.cuda.synchronize()
            f()
            torch.cuda.synchronize()
            out = (ctypes.c_longlong * 128)()
            assert lib.rcmvs_debug_deep_trace(out) == 0
            for wv in (0, 7):
                t = [out[wv * 16 + k] for k in range(7)]
                print(f"{name} {label} rep {rep} wave {wv}: " + " ".join(f"{(t[k] - t[0]) * 10:6d}ns" for k in range(1, 7)))
    trace(lambda: ops.conv3d(x4, p5, stride=2, relu=True, x_absmax=b[0], y_absmax=b[1]), "conv5 s2")
    trace(lambda: ops.conv3d(y5, p6, relu=True, x_absmax=b[1], y_absmax=b[2]), "conv6 s1")
    trace(lambda: ops.deconv3d(y6, p7, residual=x4, relu=True, x_absmax=b[2], y_absmax=b[3]), "conv7 t2")
