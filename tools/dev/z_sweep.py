"""Developer check (GPU box): time of one z-streaming conv launch against the depth of the volume (ticks per block) -- slope = a tick, intercept = the launch's fixed cost.
Back-to-back launches between two events (includes the ~1.5 us kernel boundary); pair form with a bound."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from rc_mvsnet_amd import ops
g = torch.Generator().manual_seed(0)
for (ci, co, st, H, W) in ((16, 16, 1, 128, 160), (16, 8, 1, 256, 320), (8, 16, 2, 256, 320), (8, 8, 1, 512, 640)):
    w = (torch.randn(co, ci, 3, 3, 3, generator=g) / (ci * 27) ** 0.5).cuda()
    wp = ops.pack_conv3d_weight(w)
    for D in (1, 2, 4, 8, 16, 32):
        x = torch.randn(1, D, H, W, ci, generator=g).cuda()
        xmax = ops.absmax(x)
        for _ in range(5): y = ops.conv3d(x, wp, stride=st, x_absmax=xmax)
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        best = 1e9
        for rep in range(5):
            a.record()
            for _ in range(40): y = ops.conv3d(x, wp, stride=st, x_absmax=xmax)
            b.record(); torch.cuda.synchronize()
            best = min(best, a.elapsed_time(b) / 40 * 1e3)
        print(f"{ci:2d}->{co:2d} s{st} {D:2d}x{H}x{W}: {best:6.1f} us   in {x.numel() * 4 / 1e6:6.1f} MB out {y.numel() * 4 / 1e6:6.1f} MB")
