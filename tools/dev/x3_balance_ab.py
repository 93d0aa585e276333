"""Developer A/B (GPU box): every split-operand 3-D conv layer of the three config-2 cost regularisations, chunked item schedule
(RCMVS_X3_BALANCE=0) against the balanced step-range schedule (=1), fp16-pair form.  One child process per mode (the switch is read once)."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if len(sys.argv) > 1 and sys.argv[1] == "child":
    sys.path.insert(0, ROOT)
    import torch
    from rc_mvsnet_amd import _lib, ops
    _lib.load()
    dev = "cuda:0"
    def t(fn, R=20):
        for _ in range(3): fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize(); e0.record()
        for _ in range(R): fn()
        e1.record(); torch.cuda.synchronize()
        return e0.elapsed_time(e1) * 1e3 / R
    g = torch.Generator().manual_seed(0)
    for (stage, C0, D, h, w) in ((1, 32, 48, 128, 160), (2, 16, 32, 256, 320), (3, 8, 8, 512, 640)):
        layers = [("conv0", "s1", C0, 8, 1), ("conv1", "s2", 8, 16, 1), ("conv2", "s1", 16, 16, 2), ("conv3", "s2", 16, 32, 2), ("conv4", "s1", 32, 32, 4),
                  ("conv9", "t2", 32, 16, 4), ("conv11", "t2", 16, 8, 2)]
        for name, kind, ci, co, div in layers:
            d_, h_, w_ = max(D // div, 1), h // div, w // div
            x = torch.randn(1, d_, h_, w_, ci, generator=g).to(dev)
            bound = ops.absmax(x)
            sc, sh = (torch.rand(co, generator=g) + 0.5).to(dev), (0.1 * torch.randn(co, generator=g)).to(dev)
            if kind == "t2":
                wp = ops.pack_conv3d_weight((0.05 * torch.randn(ci, co, 3, 3, 3, generator=g)).to(dev), transposed=True)
                us = t(lambda: ops.deconv3d(x, wp, sc, sh, relu=True, x_absmax=bound))
            else:
                wp = ops.pack_conv3d_weight((0.05 * torch.randn(co, ci, 3, 3, 3, generator=g)).to(dev))
                us = t(lambda: ops.conv3d(x, wp, sc, sh, stride=2 if kind == "s2" else 1, relu=True, x_absmax=bound))
            print(f"S{stage} {name:6s} {kind} {ci:2d}->{co:2d} {d_}x{h_}x{w_}: {us:7.1f}")
    sys.exit(0)
res = {}
for mode in ("0", "1"):
    out = subprocess.run([sys.executable, __file__, "child"], env=dict(os.environ, RCMVS_X3_BALANCE=mode), capture_output=True, text=True)
    res[mode] = [ln for ln in out.stdout.splitlines() if ln.startswith("S")]
    if not res[mode]: print(out.stderr[-2000:])
tot0 = tot1 = 0.0
for a, b in zip(res["0"], res["1"]):
    ua, ub = float(a.split(":")[1]), float(b.split(":")[1])
    tot0 += ua; tot1 += ub
    print(f"{a.split(':')[0]:42s} chunked {ua:7.1f}  balanced {ub:7.1f}  {100 * (ub / ua - 1):+6.1f} %")
print(f"sum: chunked {tot0:.1f}  balanced {tot1:.1f}")
