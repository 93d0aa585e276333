import sys, torch
sys.path.insert(0, "/root/repo")
from rc_mvsnet_amd import ops as hip, synthetic, _lib
_lib.load()
gpu = lambda t: t.cuda().contiguous()
for (C, D, h, w) in ((8, 12, 64, 96), (8, 4, 5, 3), (16, 7, 33, 50), (8, 12, 64, 160)):
    for B in (1, 2):
        g = torch.Generator().manual_seed(C + h)
        feats = gpu(torch.randn(B, 3, h, w, C, generator=g))
        rot, trans = hip.compose_homography(gpu(synthetic.proj_matrices(B, 3, h * 4, w * 4)["stage1"]))
        planes = gpu(torch.stack((300.0 + 600.0 * torch.rand(B, h, w, generator=g), 2.0 + 40.0 * torch.rand(B, h, w, generator=g)), dim=-1))
        vref = hip.warp_variance(feats, rot, trans, planes, D, variant=2)
        for var in (5, 6):
            v, blocks, on_window = hip.warp_variance_win(feats, rot, trans, planes, D, variant=var)
            d = (v - vref).abs()
            bad = (d > 1e-4).nonzero()
            print(f"C={C} D={D} {h}x{w} B={B} variant {var}: window {on_window}/{blocks} max {float(d.max()):.3e} bad {bad.shape[0]} nan {int(torch.isnan(v).sum())}")
            if bad.shape[0]:
                bb = bad[:, 0].unique().tolist(); kk = bad[:, 1].unique().tolist(); yy = bad[:, 2].unique().tolist(); xx = bad[:, 3].unique().tolist()
                print("   b", bb, "k", kk, "y", yy[:20], "x", xx[:40])
