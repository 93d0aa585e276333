"""GPU (MI355X) parity tests: every HIP entry point, called through the C ABI, against the CPU
oracle on the same seeded inputs, against the golden fixtures captured from the reference, and --
at BASELINE config-2 size -- through size-independent properties.

Tolerances (fp32 path, stated per north_star): end-to-end depth  mean|dd| / (d_max - d_min) <= 1e-4;
kernel-level max-abs errors relative to the tensor's magnitude as written in each test."""
import numpy as np
import os

import pytest
import torch

from conftest import load_golden, rel_err

pytestmark = pytest.mark.gpu

DEV = "cuda:0"


@pytest.fixture(scope="module")
def hip():
    from rc_mvsnet_amd import _lib, ops
    _lib.load()                       # raises if the library has not been built: no fallback
    assert torch.cuda.is_available()
    return ops


def gpu(t):
    return t.to(DEV).contiguous()


# ------------------------------------------------------------------------------------------ layout
@pytest.mark.parametrize("shape", [(3, 8, 5, 7), (2, 32, 16, 20), (1, 3, 9, 11), (1, 16, 4, 6, 10)])
def test_layout_roundtrip(hip, shape):
    x = torch.randn(*shape)
    cl = hip.to_channels_last(gpu(x))
    perm = (0, *range(2, len(shape)), 1)
    assert torch.equal(cl.cpu(), x.permute(*perm).contiguous())
    assert torch.equal(hip.to_channels_first(cl).cpu(), x)


# ------------------------------------------------------------------------------------------ geometry
def test_compose_homography(hip):
    from oracle import warp
    from rc_mvsnet_amd import synthetic
    pm = synthetic.proj_matrices(2, 5, 512, 640)["stage2"]
    rot, trans = hip.compose_homography(gpu(pm))
    for v in range(1, 5):
        r, t = warp.compose_homography(pm[:, v].double(), pm[:, 0].double())
        assert rel_err(rot[:, v - 1].cpu().reshape(2, 3, 3), r) < 1e-6
        assert rel_err(trans[:, v - 1].cpu(), t) < 1e-6
    # the cascade composes its three stages in one launch: bit-identical to the per-stage calls
    pms = synthetic.proj_matrices(2, 5, 512, 640)
    rots, transs = hip.compose_homography_stages([gpu(pms[f"stage{k}"]) for k in (1, 2, 3)])
    for k in (1, 2, 3):
        r, t = hip.compose_homography(gpu(pms[f"stage{k}"]))
        assert torch.equal(rots[k - 1], r) and torch.equal(transs[k - 1], t)


@pytest.mark.parametrize("name,scale", [("planes_s2", 2), ("planes_s3", 1), ("planes_s2odd", 2)])
def test_hypothesis_planes_vs_golden(hip, name, scale):
    from rc_mvsnet_amd import synthetic
    g = load_golden(name)
    H, W = [int(v) for v in g["full_hw"]]
    D = int(g["ndepth"])
    pl = hip.hypothesis_planes(gpu(g["prev"]), gpu(synthetic.depth_values(1)), (H, W), scale, D, int(g["ratio"])).cpu()
    k = torch.arange(D, dtype=torch.float32).reshape(1, D, 1, 1)
    samples = pl[..., 0].unsqueeze(1) + k * pl[..., 1].unsqueeze(1)
    assert float((samples - g["out"]).abs().max()) < 3e-4          # mm (1 ulp at 600 mm = 6e-5)


def test_hypothesis_planes_stage1_exact(hip):
    from rc_mvsnet_amd import synthetic
    g = load_golden("planes_s1")
    H, W = [int(v) for v in g["full_hw"]]
    pl = hip.hypothesis_planes(None, gpu(synthetic.depth_values(1)), (H, W), 4, 48, 4).cpu()
    k = torch.arange(48, dtype=torch.float32).reshape(1, 48, 1, 1)
    assert torch.equal(pl[..., 0].unsqueeze(1) + k * pl[..., 1].unsqueeze(1), g["out"])


# ------------------------------------------------------------------------------------------ K1
def _k1_case(B, V, C, D, h, w, seed):
    from rc_mvsnet_amd import synthetic
    g = torch.Generator().manual_seed(seed)
    feats = [torch.randn(B, C, h, w, generator=g) for _ in range(V)]
    pm = synthetic.proj_matrices(B, V, h * 4, w * 4)["stage1"]
    d0 = 425.0 + 100.0 * torch.rand(B, h, w, generator=g)
    dl = 2.0 + 8.0 * torch.rand(B, h, w, generator=g)
    return feats, pm, torch.stack((d0, dl), dim=-1)


@pytest.mark.parametrize("B,V,C,D,h,w,uniform", [(1, 3, 32, 8, 16, 20, False), (2, 3, 16, 16, 24, 40, False), (1, 5, 8, 8, 32, 48, False),
                                                 (1, 2, 32, 5, 9, 13, False), (1, 7, 8, 12, 10, 70, False),
                                                 # the production kernels, each against the oracle directly (what a hinted call runs: csrc/warp_variance.hip,
                                                 # k1_production_variant): V = 3, C = 8 with per-pixel planes -> the plane-pipelined form (csrc/k1_pp.h);
                                                 # pixel-invariant planes + the caller's hint -> the LDS-window form (csrc/k1_win.h); the plane-pipelined form at the
                                                 # view counts the reference's workflows use (4 training, 5 DTU evaluation, 7 Tanks and Temples), with and without the
                                                 # stage-1 hint; V = 5, C = 16 -> the FMA build of the two-phase kernel
                                                 (1, 3, 8, 8, 32, 48, False), (1, 3, 32, 8, 16, 20, True), (1, 3, 16, 12, 24, 40, True), (2, 3, 8, 8, 32, 48, True),
                                                 (1, 4, 8, 8, 20, 48, False), (1, 4, 16, 8, 20, 24, False), (1, 5, 16, 8, 24, 40, False), (1, 5, 8, 8, 24, 72, False),
                                                 (1, 7, 32, 8, 12, 24, False), (1, 7, 8, 16, 16, 64, False), (1, 7, 16, 5, 9, 33, False),
                                                 (1, 4, 32, 8, 16, 20, True), (1, 5, 32, 12, 16, 24, True), (1, 7, 32, 8, 12, 20, True), (1, 2, 16, 4, 12, 36, True),
                                                 (2, 5, 8, 4, 8, 70, True)])
def test_warp_variance_vs_oracle(hip, B, V, C, D, h, w, uniform):
    from oracle import warp
    feats, pm, planes = _k1_case(B, V, C, D, h, w, 3)
    if uniform:                 # stage 1 of the cascade: the same planes at every pixel (models/modules.py:549-566)
        planes = planes[:, :1, :1].expand(B, h, w, 2).contiguous()
    k = torch.arange(D, dtype=torch.float32).reshape(1, D, 1, 1)
    samples = planes[..., 0].unsqueeze(1) + k * planes[..., 1].unsqueeze(1)
    # the oracle composes the homography in fp32 LU; feed the kernel the SAME rot/trans so that the
    # comparison isolates the kernel (the fp64 composer is tested separately)
    rots, transs = zip(*[warp.compose_homography(pm[:, v], pm[:, 0]) for v in range(1, V)])
    rot = torch.stack([r.reshape(B, 9) for r in rots], dim=1)
    trans = torch.stack(transs, dim=1)
    ref = warp.variance_volume(feats, pm, samples)                       # (B,C,D,h,w)
    f_cl = torch.stack([f.permute(0, 2, 3, 1) for f in feats], dim=1)    # (B,V,h,w,C)
    var = hip.warp_variance(gpu(f_cl), gpu(rot), gpu(trans), gpu(planes), D, uniform_planes=uniform).cpu().permute(0, 4, 1, 2, 3)
    diff = (var - ref).abs()
    print(f"K1 max|d|={float(diff.max()):.3e} exact={float((diff == 0).float().mean()):.4f}")
    assert float(diff.max()) < 1e-5 * max(1.0, float(ref.abs().max()))
    # ... and the kernel the routing table names really ran: the hinted call equals that debug variant bit for bit
    nsrc = V - 1
    want = 0
    if nsrc == 2:
        want = 5 if uniform else (7 if C == 8 else 0)
    elif nsrc in (3, 6):
        want = 7
    elif nsrc == 4:
        want = 1 if (C == 16 and not uniform) else 7
    vv = hip.warp_variance(gpu(f_cl), gpu(rot), gpu(trans), gpu(planes), D, variant=want).cpu().permute(0, 4, 1, 2, 3)
    assert torch.equal(vv, var), want


def test_warp_variance_timed_entry(hip):
    """rcmvs_warp_variance_timed_fwd (bench.py's roofline probe): the hinted call with the kernel's own start / stop timestamps in two caller-owned
    events -- same results bit for bit, a positive duration that is no longer than event records around the same launch."""
    if DEV == "cpu":
        pytest.skip("events are not modelled on the kernel emulation")
    from rc_mvsnet_amd import synthetic
    feats = gpu(torch.randn(1, 3, 128, 160, 32))
    rot, trans = hip.compose_homography(gpu(synthetic.proj_matrices(1, 3, 512, 640)["stage1"]))
    planes = hip.hypothesis_planes(None, gpu(synthetic.depth_values(1)), (512, 640), 4, 48, 4)
    want = hip.warp_variance(feats, rot, trans, planes, 48, uniform_planes=True)
    rec = [torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)]
    torch.cuda.synchronize()
    rec[0].record()
    hip.warp_variance(feats, rot, trans, planes, 48, uniform_planes=True)
    rec[1].record()
    torch.cuda.synchronize()
    try:
        hip.K1_EVENTS = []
        got = hip.warp_variance(feats, rot, trans, planes, 48, uniform_planes=True)
        torch.cuda.synchronize()
        (e0, e1), = hip.K1_EVENTS
    finally:
        hip.K1_EVENTS = None
    own, around = e0.elapsed_time(e1) * 1e3, rec[0].elapsed_time(rec[1]) * 1e3
    print(f"K1 stage-1 launch: {own:.1f} us by its own timestamps, {around:.1f} us between event records around it")
    assert torch.equal(got, want) and 5.0 < own <= around + 1.0


def test_warp_variance_golden_fixture(hip):
    """homo_warping fixture from the reference (incl. out-of-bounds / negative / zero depths):
    with V=2 and a zero reference map, var = w^2/2 - (w/2)^2 = w^2/4  ->  |w| = 2*sqrt(var)."""
    for name in ("warp_a1", "warp_a2", "warp_b1"):
        g = load_golden(name)
        src, depth, out = g["src"], g["depth"], g["out"]
        B, C, h, w = src.shape
        D = depth.shape[1]
        pm = torch.stack((g["ref_proj"], g["src_proj"]), dim=1)
        rot, trans = hip.compose_homography(gpu(pm))
        f_cl = torch.stack((torch.zeros_like(src), src), dim=1).permute(0, 1, 3, 4, 2).contiguous()
        for k in range(D):      # arbitrary per-pixel depth: one plane at a time, d_0 = depth_k, delta = 0
            planes = torch.stack((depth[:, k], torch.zeros_like(depth[:, k])), dim=-1)
            var = hip.warp_variance(gpu(f_cl), rot, trans, gpu(planes), 1).cpu()[:, 0].permute(0, 3, 1, 2)
            assert float((2 * var.clamp(min=0).sqrt() - out[:, :, k].abs()).abs().max()) < 2e-4


def test_warp_variance_properties_full_size(hip):
    """config-2 stage shapes: (a) identical views -> zero variance; (b) var(2f) == 4 var(f) exactly;
    (c) finite everywhere."""
    from rc_mvsnet_amd import synthetic
    for (C, D, h, w) in ((32, 48, 128, 160), (16, 32, 256, 320), (8, 8, 512, 640)):
        g = torch.Generator().manual_seed(C)
        f = torch.randn(1, 1, h, w, C, generator=g)
        pm = gpu(synthetic.proj_matrices(1, 3, 512, 640)["stage1" if C == 32 else ("stage2" if C == 16 else "stage3")])
        rot, trans = hip.compose_homography(pm)
        dv = gpu(synthetic.depth_values(1))
        planes = hip.hypothesis_planes(None, dv, (512, 640), 512 // h, D, 4)
        feats = gpu(torch.cat((f, torch.randn(1, 2, h, w, C, generator=g)), dim=1))
        v1 = hip.warp_variance(feats, rot, trans, planes, D)
        v2 = hip.warp_variance(feats * 2, rot, trans, planes, D)
        assert torch.isfinite(v1).all()
        assert torch.equal(v2, v1 * 4)
        same = gpu(f.repeat(1, 3, 1, 1, 1))
        eye_pm = pm.clone()
        eye_pm[:, 1:] = eye_pm[:, :1]
        r2, t2 = hip.compose_homography(eye_pm)
        v0 = hip.warp_variance(same, r2, t2, planes, D)
        assert float(v0.abs().max()) < 1e-5


def test_warp_variance_variants_agree(hip):
    """K1 production kernel (two-phase, LDS tap table, custom exact divisions; variant 0) must be BIT-IDENTICAL to the
    reference-order kernel (variant 2: one full coordinate chain per lane, compiler IEEE division), for every code path:
    compile-time 2 / 4 / 6 source views, and the general multi-chunk path (1, 3 source views).  The FMA-contracted build
    (variant 1) stays within 2e-6 relative."""
    from rc_mvsnet_amd import synthetic
    for (C, D, h, w, V) in ((32, 16, 20, 37, 3), (16, 8, 33, 50, 4), (8, 24, 30, 70, 2), (8, 8, 20, 40, 7),
                            (16, 16, 18, 30, 5), (32, 8, 12, 20, 5), (8, 12, 64, 96, 3), (32, 8, 9, 11, 7), (16, 8, 14, 22, 7)):
        g = torch.Generator().manual_seed(C + V)
        feats = gpu(torch.randn(2, V, h, w, C, generator=g))
        pm = gpu(synthetic.proj_matrices(2, V, h * 4, w * 4)["stage1"])
        rot, trans = hip.compose_homography(pm)
        planes = gpu(torch.stack((425.0 + 100.0 * torch.rand(2, h, w, generator=g), 2.0 + 8.0 * torch.rand(2, h, w, generator=g)), dim=-1))
        vref = hip.warp_variance(feats, rot, trans, planes, D, variant=2)
        v0 = hip.warp_variance(feats, rot, trans, planes, D, variant=0)
        v1 = hip.warp_variance(feats, rot, trans, planes, D, variant=1)
        vp = hip.warp_variance(feats, rot, trans, planes, D)                  # what a production call runs (FMA-contracted for V = 3, C = 8)
        exact = float((v0 == vref).float().mean())
        err1 = float((v1 - vref).abs().max()) / max(1.0, float(vref.abs().max()))
        errp = float((vp - vref).abs().max()) / max(1.0, float(vref.abs().max()))
        print(f"K1 C={C} D={D} V={V}: two-phase kernel vs reference-order bit-identical {exact:.6f}; FMA build max rel {err1:.2e}; production call {errp:.2e}")
        assert torch.equal(v0, vref)
        assert err1 < 2e-6 and errp < 2e-6
        tol = 2e-6 * max(1.0, float(vref.abs().max()))
        if V - 1 in (2, 3, 4, 6):           # plane-pipelined gather form: 2, 3, 4 or 6 source views
            v7 = hip.warp_variance(feats, rot, trans, planes, D, variant=7)
            assert float((v7 - vref).abs().max()) <= tol, ("pp", C, V)
        if V == 3:
            for var in (5, 6):               # window form (two source views; rough planes: most tiles fall back one by one)
                v5, blocks, on_window = hip.warp_variance_win(feats, rot, trans, planes, D, variant=var)
                assert float((v5 - vref).abs().max()) <= tol, ("win", var, C, V)
                assert 0 < blocks and on_window <= blocks
        # the plain ABI entry is the exact kernel for every view count (version 104)
        B_, V_, h_, w_, C_ = feats.shape
        vplain = torch.empty_like(v0)
        from rc_mvsnet_amd import _lib, ops as _ops
        _lib.check(_lib.load().rcmvs_warp_variance_fwd(_ops._chk(feats, "f"), _ops._chk(rot, "r"), _ops._chk(trans, "t"), _ops._chk(planes, "p"),
                                                       _ops._chk(vplain, "v"), B_, V_, C_, D, h_, w_, _ops._stream()), "warp_variance_fwd")
        assert torch.equal(vplain, vref)
    with pytest.raises(Exception):
        hip.warp_variance(feats, rot, trans, planes, D, variant=9)


def test_warp_variance_window_form(hip):
    """K1 with the tiles' source windows staged in LDS (csrc/k1_win.h; what the uniform-planes hint of stage 1 launches): same sampling
    positions as the reference-order kernel, FMA-contracted blend -> <= 2e-6 of the value range; on the config-2 stage-1 shape with the
    cascade's own plane table (pixel-invariant) nearly every tile must take the window path, on rough plane tables tiles fall back one
    by one; the hint changes nothing for view counts the window form is not built for."""
    from rc_mvsnet_amd import synthetic
    # (a) config-2 stage shapes, stage-1 style planes
    for (C, D, h, w, stage) in ((32, 48, 128, 160, "stage1"), (16, 32, 256, 320, "stage2"), (8, 8, 512, 640, "stage3")):
        g = torch.Generator().manual_seed(C)
        feats = gpu(torch.randn(1, 3, h, w, C, generator=g))
        rot, trans = hip.compose_homography(gpu(synthetic.proj_matrices(1, 3, 512, 640)[stage]))
        planes = hip.hypothesis_planes(None, gpu(synthetic.depth_values(1)), (512, 640), 512 // h, D, 4 if C == 32 else 1)
        vref = hip.warp_variance(feats, rot, trans, planes, D, variant=2)
        tol = 2e-6 * max(1.0, float(vref.abs().max()))
        for var in (5, 6):
            v, blocks, on_window = hip.warp_variance_win(feats, rot, trans, planes, D, variant=var)
            err = float((v - vref).abs().max())
            print(f"K1 window form C={C} variant {var}: {on_window} of {blocks} tiles on the window path, max |d| {err:.2e} (tol {tol:.2e})")
            assert err <= tol
            if C == 32:
                assert on_window >= 0.9 * blocks
        assert torch.equal(hip.warp_variance(feats, rot, trans, planes, D, uniform_planes=True), hip.warp_variance(feats, rot, trans, planes, D, variant=5))
    # (b) rough plane tables, ragged sizes, two items
    for (C, D, h, w) in ((32, 16, 20, 37), (16, 7, 33, 50), (8, 12, 64, 96), (32, 5, 9, 13), (8, 4, 5, 3)):
        g = torch.Generator().manual_seed(C + h)
        feats = gpu(torch.randn(2, 3, h, w, C, generator=g))
        rot, trans = hip.compose_homography(gpu(synthetic.proj_matrices(2, 3, h * 4, w * 4)["stage1"]))
        planes = gpu(torch.stack((300.0 + 600.0 * torch.rand(2, h, w, generator=g), 2.0 + 40.0 * torch.rand(2, h, w, generator=g)), dim=-1))
        vref = hip.warp_variance(feats, rot, trans, planes, D, variant=2)
        for var in (5, 6):
            v, blocks, on_window = hip.warp_variance_win(feats, rot, trans, planes, D, variant=var)
            assert float((v - vref).abs().max()) <= 2e-6 * max(1.0, float(vref.abs().max())), (C, var)
            assert 0 < blocks and on_window <= blocks
    # (c) the hint is only a hint: other view counts run the kernel measured fastest for them (profiles/r6_k1_views.txt: the plane-pipelined form)
    feats = gpu(torch.randn(1, 5, 12, 20, 32))
    rot, trans = hip.compose_homography(gpu(synthetic.proj_matrices(1, 5, 48, 80)["stage1"]))
    planes = gpu(torch.stack((torch.full((1, 12, 20), 500.0), torch.full((1, 12, 20), 5.0)), dim=-1))
    assert torch.equal(hip.warp_variance(feats, rot, trans, planes, 8, uniform_planes=True), hip.warp_variance(feats, rot, trans, planes, 8, variant=7))
    with pytest.raises(Exception):
        hip.warp_variance(feats, rot, trans, planes, 8, variant=5)


@pytest.mark.parametrize("s2d", [False, True, 32])
@pytest.mark.parametrize("N,H,W", [(1, 8, 32), (2, 19, 47), (3, 40, 64), (1, 5, 3)])
def test_conv2d_tile_vs_fp64(hip, N, H, W, s2d):
    """FeatureNet's two 32 -> 16 3x3 layers on the tile kernel (csrc/conv2d_tile.hip): the FPN output conv out2 (plain conv, with the bound of the variance
    volume kept in the epilogue) and conv1.0 -- a 5x5 stride-2 Conv2d(8, 16) + BatchNorm + ReLU read through the space-to-depth view of its input
    (models/modules.py:374,437) -- against an fp64 evaluation, next to the planar split-bf16 kernel they ran on: exact tile multiples, ragged tiles, maps
    smaller than a tile."""
    from rc_mvsnet_amd.casmvsnet import FeatureNet
    g = torch.Generator().manual_seed(H * 3 + W + int(s2d))
    f = torch.nn.functional
    if s2d == 32:                                                         # 32 -> 32 (conv2.1 / conv2.2): the eight-wave form, BatchNorm + ReLU
        x = torch.randn(N, 32, H, W, generator=g) * torch.exp(0.5 * torch.randn(N, 32, H, W, generator=g))
        w = torch.randn(32, 32, 3, 3, generator=g) / (32 * 9) ** 0.5
        sc, sh = 0.5 + torch.rand(32, generator=g), 0.2 * torch.randn(32, generator=g)
        ref = torch.relu(f.conv2d(x.double(), w.double(), padding=1) * sc.double().view(1, -1, 1, 1) + sh.double().view(1, -1, 1, 1))
        xcl = gpu(x.permute(0, 2, 3, 1))
        got_t = hip.conv2d_tile(xcl, hip.pack_conv2d_tile(gpu(w)), gpu(sc), gpu(sh), relu=True)
        old = hip.conv3d(xcl.unsqueeze(1), hip.pack_conv3d_weight(gpu(FeatureNet._w3(w).contiguous())), gpu(sc), gpu(sh), relu=True).squeeze(1)
    elif s2d:
        x = torch.randn(N, 8, 2 * H, 2 * W, generator=g) * torch.exp(0.5 * torch.randn(N, 8, 2 * H, 2 * W, generator=g))
        w5 = torch.randn(16, 8, 5, 5, generator=g) / (8 * 25) ** 0.5
        sc, sh = 0.5 + torch.rand(16, generator=g), 0.2 * torch.randn(16, generator=g)
        ref = torch.relu(f.conv2d(x.double(), w5.double(), stride=2, padding=2) * sc.double().view(1, -1, 1, 1) + sh.double().view(1, -1, 1, 1))
        w = FeatureNet._w5s2(w5)
        xcl = gpu(x.permute(0, 2, 3, 1))
        got_t = hip.conv2d_tile(xcl, hip.pack_conv2d_tile(gpu(w)), gpu(sc), gpu(sh), relu=True, s2d=True)
        old = hip.conv2d_s2d(xcl, hip.pack_conv3d_weight(gpu(FeatureNet._w3(w).contiguous())), gpu(sc), gpu(sh), relu=True)
    else:
        x = torch.randn(N, 32, H, W, generator=g) * torch.exp(0.5 * torch.randn(N, 32, H, W, generator=g))
        w = torch.randn(16, 32, 3, 3, generator=g) / (32 * 9) ** 0.5
        ref = f.conv2d(x.double(), w.double(), padding=1)
        xcl = gpu(x.permute(0, 2, 3, 1))
        bound = torch.zeros(hip.ABSMAX_FLOATS, device=xcl.device)
        got_t = hip.conv2d_tile(xcl, hip.pack_conv2d_tile(gpu(w)), ysq_absmax=bound)
        assert float(bound.max()) == float(got_t.abs().max() ** 2)
        old = hip.conv3d(xcl.unsqueeze(1), hip.pack_conv3d_weight(gpu(FeatureNet._w3(w).contiguous()))).squeeze(1)
    assert tuple(got_t.shape) == (N, H, W, 32 if s2d == 32 else 16)
    got, two = got_t.cpu().permute(0, 3, 1, 2).double(), old.cpu().permute(0, 3, 1, 2).double()
    mag = float(ref.abs().max())
    e_t, e_o = float((got - ref).abs().max()), float((two - ref).abs().max())
    print(f"conv2d tile s2d={s2d} {N}x{H}x{W}: max error vs fp64 / max: tile kernel {e_t / mag:.2e}, planar kernel {e_o / mag:.2e}")
    assert e_t <= 2.0 * e_o + 2e-7 * mag and e_t < 2e-6 * mag


@pytest.mark.parametrize("N,H,W", [(2, 28, 60), (1, 37, 53), (3, 9, 11), (3, 64, 96), (3, 512, 640)])
def test_conv2d_stem_vs_fp64(hip, N, H, W):
    """FeatureNet's conv0 block (Conv2d(3, 8) -> Conv2d(8, 8), each conv + BatchNorm(eval) + ReLU, models/modules.py:372-373) in one launch from the
    planar images (csrc/conv2d_stem.hip, the 8-channel map between the layers in LDS) against an fp64 evaluation, next to the two launches it
    replaces (the first layer on the tile kernel, the second on the planar split-bf16 kernel): exact tile multiples, ragged tiles, maps smaller
    than a tile, the second layer's zero padding of the INTERMEDIATE map at the image border; more tiles than resident blocks (the blocks are
    persistent: 60 tiles on the emulation's 24 resident blocks, a DTU scene's 2 442 on the GPU's 1 024)."""
    if DEV == "cpu" and H > 100:
        pytest.skip("GPU only: minutes on the kernel emulation")
    g = torch.Generator().manual_seed(H * 5 + W)
    x = torch.randn(N, 3, H, W, generator=g) * torch.exp(0.5 * torch.randn(N, 3, H, W, generator=g))
    wa, wb = torch.randn(8, 3, 3, 3, generator=g) / 5.0, torch.randn(8, 8, 3, 3, generator=g) / 8.0
    sa, sb = (0.5 + torch.rand(8, generator=g) for _ in range(2))
    ha, hb = (0.2 * torch.randn(8, generator=g) for _ in range(2))
    f = torch.nn.functional
    mid = torch.relu(f.conv2d(x.double(), wa.double(), padding=1) * sa.double().view(1, -1, 1, 1) + ha.double().view(1, -1, 1, 1))
    ref = torch.relu(f.conv2d(mid, wb.double(), padding=1) * sb.double().view(1, -1, 1, 1) + hb.double().view(1, -1, 1, 1))
    pa = hip.pack_conv2d_weight(gpu(wa), pad_in_to=4)
    got = hip.conv2d_stem(gpu(x), pa, gpu(sa), gpu(ha), hip.pack_conv2d_stem(gpu(wb)), gpu(sb), gpu(hb)).cpu().permute(0, 3, 1, 2).double()
    w3 = lambda w: torch.cat((torch.zeros_like(w).unsqueeze(2), w.unsqueeze(2), torch.zeros_like(w).unsqueeze(2)), dim=2)
    t = hip.conv2d_rgb(gpu(x), pa, gpu(sa), gpu(ha), relu=True)
    two = hip.conv3d(t.unsqueeze(1), hip.pack_conv3d_weight(gpu(w3(wb))), gpu(sb), gpu(hb), relu=True).squeeze(1).cpu().permute(0, 3, 1, 2).double()
    mag = float(ref.abs().max())
    e_f, e_2 = float((got - ref).abs().max()), float((two - ref).abs().max())
    print(f"conv2d stem {N}x{H}x{W}: max error vs fp64 / max: fused {e_f / mag:.2e}, two launches {e_2 / mag:.2e}")
    assert e_f <= 2.0 * e_2 + 2e-7 * mag and e_f < 2e-6 * mag


@pytest.mark.parametrize("N,H,W", [(1, 8, 30), (2, 19, 47), (3, 32, 64), (1, 5, 3)])
def test_conv2d_pair_vs_fp64(hip, N, H, W):
    """FeatureNet's conv1.1 -> conv1.2 (two 16 -> 16 3x3 Conv2d blocks: conv + BatchNorm(eval) + ReLU, models/modules.py:372-379) in one launch
    (csrc/conv2d_pair.hip, the map between them in LDS) against an fp64 evaluation, next to the two launches of the planar split-bf16 kernel it
    replaces: ragged tiles, maps smaller than a tile, the second layer's zero padding of the INTERMEDIATE map at the image border."""
    g = torch.Generator().manual_seed(H * 7 + W)
    x = torch.randn(N, 16, H, W, generator=g) * torch.exp(0.5 * torch.randn(N, 16, H, W, generator=g))
    wa, wb = (torch.randn(16, 16, 3, 3, generator=g) / 12.0 for _ in range(2))
    sa, sb = (0.5 + torch.rand(16, generator=g) for _ in range(2))
    ha, hb = (0.2 * torch.randn(16, generator=g) for _ in range(2))
    f = torch.nn.functional
    mid = torch.relu(f.conv2d(x.double(), wa.double(), padding=1) * sa.double().view(1, -1, 1, 1) + ha.double().view(1, -1, 1, 1))
    ref = torch.relu(f.conv2d(mid, wb.double(), padding=1) * sb.double().view(1, -1, 1, 1) + hb.double().view(1, -1, 1, 1))
    xcl = gpu(x.permute(0, 2, 3, 1))
    img = hip.pack_conv2d_pair(gpu(wa), gpu(wb))
    got = hip.conv2d_pair(xcl, img, gpu(sa), gpu(ha), gpu(sb), gpu(hb)).cpu().permute(0, 3, 1, 2).double()
    # the two launches it replaces (each layer as a one-plane volume on the planar kernel)
    w3 = lambda w: torch.cat((torch.zeros_like(w).unsqueeze(2), w.unsqueeze(2), torch.zeros_like(w).unsqueeze(2)), dim=2)
    pa, pb = hip.pack_conv3d_weight(gpu(w3(wa))), hip.pack_conv3d_weight(gpu(w3(wb)))
    t = hip.conv3d(xcl.unsqueeze(1), pa, gpu(sa), gpu(ha), relu=True)
    two = hip.conv3d(t, pb, gpu(sb), gpu(hb), relu=True).squeeze(1).cpu().permute(0, 3, 1, 2).double()
    mag = float(ref.abs().max())
    e_f, e_2 = float((got - ref).abs().max()), float((two - ref).abs().max())
    print(f"conv2d pair {N}x{H}x{W}: max error vs fp64 / max: fused {e_f / mag:.2e}, two launches {e_2 / mag:.2e}")
    assert e_f <= 2.0 * e_2 + 2e-7 * mag and e_f < 2e-6 * mag


@pytest.mark.parametrize("N,H,W,h,w", [(3, 64, 96, 16, 24), (2, 64, 96, 32, 48), (1, 20, 28, 20, 28), (2, 37, 53, 9, 13), (1, 16, 16, 40, 24)])
def test_resize_rgb_cl_is_torch_bilinear(hip, N, H, W, h, w):
    """The train variant's small images (models/casmvsnet.py:60-62: F.interpolate(..., mode="bilinear", align_corners=False)) fused with the
    channels-last transpose: ATen's arithmetic, so equal to torch on the same device up to the last bit or two (integer and ragged scales,
    identity, up-sampling)."""
    g = torch.Generator().manual_seed(H + w)
    x = gpu(torch.randn(N, 3, H, W, generator=g))
    ref = torch.nn.functional.interpolate(x, (h, w), mode="bilinear", align_corners=False).permute(0, 2, 3, 1)
    got = hip.resize_rgb_cl(x, (h, w))
    assert got.shape == (N, h, w, 3)
    assert float((got - ref).abs().max()) <= 1e-6 * max(1.0, float(ref.abs().max()))


# ------------------------------------------------------------------------------------------ K2/K3
@pytest.mark.parametrize("Ci,Co,mode", [(8, 16, "s2"), (16, 16, "s1"), (16, 32, "s2"), (32, 32, "s1"), (32, 64, "s2"),
                                        (64, 64, "s1"), (64, 32, "t2"), (32, 16, "t2")])
@pytest.mark.parametrize("big", [False, True])
def test_conv3d_mfma_matches_direct(hip, Ci, Co, mode, big):
    """The MFMA implicit-GEMM kernels against the direct kernels (themselves checked against the
    oracle): both tile shapes (one / four n-tiles per wave), ragged last tile, all epilogue parts."""
    g = torch.Generator().manual_seed(Ci + Co)
    shape = (1, 18, 61, 64) if big else (2, 5, 7, 9)            # big: >= 4096 n-tiles for the conv modes
    if big and mode == "t2":
        shape = (1, 17, 62, 64)
    x = gpu(torch.randn(*shape, Ci, generator=g))
    w = torch.randn(Co, Ci, 3, 3, 3, generator=g) / (Ci * 27) ** 0.5
    if mode == "t2":
        w = w.permute(1, 0, 2, 3, 4).contiguous()
    wp = hip.pack_conv3d_weight(gpu(w), transposed=(mode == "t2"))
    scale, shift = gpu(0.5 + torch.rand(Co, generator=g)), gpu(0.1 * torch.randn(Co, generator=g))

    def run():
        if mode == "t2":
            y = hip.deconv3d(x, wp)
            res = torch.ones_like(y) * 0.25
            return y, hip.deconv3d(x, wp, scale, shift, res, relu=True)
        st = 2 if mode == "s2" else 1
        y = hip.conv3d(x, wp, stride=st)
        res = torch.ones_like(y) * 0.25
        return y, hip.conv3d(x, wp, scale, shift, res, stride=st, relu=True)

    try:
        hip.force_direct_conv(True)
        d_plain, d_full = run()
    finally:
        hip.force_direct_conv(False)
    m_plain, m_full = run()
    assert m_plain.shape == d_plain.shape
    assert rel_err(m_plain.cpu(), d_plain.cpu()) < 1e-5
    assert rel_err(m_full.cpu(), d_full.cpu()) < 1e-5


@pytest.mark.parametrize("Ci,Co,stride", [(8, 8, 1), (32, 8, 1), (16, 16, 1), (32, 32, 1), (64, 64, 1), (8, 1, 1),
                                          (8, 16, 2), (16, 32, 2), (32, 64, 2), (41, 8, 1)])
def test_conv3d_vs_oracle(hip, Ci, Co, stride):
    from oracle import conv3d as oc
    g = torch.Generator().manual_seed(Ci * 100 + Co)
    x = torch.randn(2, Ci, 6, 9, 12, generator=g)
    w = torch.randn(Co, Ci, 3, 3, 3, generator=g) / (Ci * 27) ** 0.5
    scale = 0.5 + torch.rand(Co, generator=g)
    shift = torch.randn(Co, generator=g) * 0.1
    ref = oc.conv3d(x, w, stride)
    wp = hip.pack_conv3d_weight(gpu(w))
    xcl = gpu(x.permute(0, 2, 3, 4, 1))
    y = hip.conv3d(xcl, wp, stride=stride).cpu().permute(0, 4, 1, 2, 3)
    assert rel_err(y, ref) < 2e-5
    res = torch.randn_like(ref)
    y2 = hip.conv3d(xcl, wp, gpu(scale), gpu(shift), gpu(res.permute(0, 2, 3, 4, 1)), stride=stride, relu=True)
    ref2 = torch.relu(ref * scale.view(1, -1, 1, 1, 1) + shift.view(1, -1, 1, 1, 1)) + res
    assert rel_err(y2.cpu().permute(0, 4, 1, 2, 3), ref2) < 2e-5


@pytest.mark.parametrize("Ci,Co", [(8, 8), (16, 8), (32, 8), (16, 16), (32, 32)])
@pytest.mark.parametrize("shape", [(2, 5, 11, 21), (1, 9, 37, 70), (1, 20, 8, 32)])
def test_conv3d_x3_vs_fp64(hip, Ci, Co, shape):
    """The split-bf16 MFMA kernels (csrc/conv3d_x3.hip: three bf16 pieces per fp32 operand, six MFMAs per product, fp32
    accumulation) must be as close to an fp64 convolution as the fp32 FMA-chain kernels are: wide-dynamic-range inputs,
    ragged tiles in x / y, z chunks, batch 2, full epilogue, and the flipped-tap (data gradient) weight image."""
    g = torch.Generator().manual_seed(Ci * 7 + Co + shape[1])
    B, D, H, W = shape
    x = torch.randn(B, Ci, D, H, W, generator=g) * torch.exp(torch.randn(B, Ci, D, H, W, generator=g))
    w = torch.randn(Co, Ci, 3, 3, 3, generator=g) / (Ci * 27) ** 0.5
    scale, shift = 0.5 + torch.rand(Co, generator=g), 0.1 * torch.randn(Co, generator=g)
    ref = torch.nn.functional.conv3d(x.double(), w.double(), padding=1)
    res = torch.randn(B, Co, D, H, W, generator=g)
    ref2 = torch.relu(ref * scale.double().view(1, -1, 1, 1, 1) + shift.double().view(1, -1, 1, 1, 1)) + res.double()
    wp = hip.pack_conv3d_weight(gpu(w))
    xcl = gpu(x.permute(0, 2, 3, 4, 1))
    rcl = gpu(res.permute(0, 2, 3, 4, 1))
    outs = {}
    for name, cfg in (("x3", 0), ("fp32", 64)):
        try:
            hip.force_direct_conv(cfg)
            outs[name] = (hip.conv3d(xcl, wp).cpu().permute(0, 4, 1, 2, 3).double(),
                          hip.conv3d(xcl, wp, gpu(scale), gpu(shift), rcl, relu=True).cpu().permute(0, 4, 1, 2, 3).double())
        finally:
            hip.force_direct_conv(0)
    e_x3 = float((outs["x3"][0] - ref).abs().max()), float((outs["x3"][1] - ref2).abs().max())
    e_32 = float((outs["fp32"][0] - ref).abs().max()), float((outs["fp32"][1] - ref2).abs().max())
    mag = float(ref.abs().max())
    assert e_x3[0] <= 2.0 * e_32[0] + 1e-7 * mag and e_x3[1] <= 2.0 * e_32[1] + 1e-7 * mag, (e_x3, e_32, mag)
    assert e_x3[0] < 3e-6 * mag
    if Ci == Co:      # adjoint of a stride-1 conv (training dgrad): weight (Ci, Co, 27) with flipped taps
        wt = torch.randn(Ci, Co, 3, 3, 3, generator=g) / (Ci * 27) ** 0.5
        refT = torch.nn.functional.conv_transpose3d(x.double(), wt.double(), padding=1)
        yT = hip.conv3d(xcl, hip.pack_conv3d_weight(gpu(wt), transposed=2)).cpu().permute(0, 4, 1, 2, 3).double()
        assert float((yT - refT).abs().max()) < 3e-6 * float(refT.abs().max())


@pytest.mark.parametrize("Ci,Co,kind", [(8, 16, "s2"), (16, 32, "s2"), (16, 8, "t2"), (32, 16, "t2")])
@pytest.mark.parametrize("shape", [(2, 5, 11, 21), (1, 9, 37, 70), (1, 16, 8, 32)])
def test_conv3d_x3_strided_vs_fp64(hip, Ci, Co, kind, shape):
    """Stride-2 and transposed stride-2 forms of the split-bf16 MFMA kernel against fp64 (odd sizes: the stride-2 output of an
    odd extent, the last input cell of the transposed conv reading a neighbour outside the volume), full epilogue."""
    g = torch.Generator().manual_seed(Ci * 7 + Co + shape[2])
    B, D, H, W = shape
    x = torch.randn(B, Ci, D, H, W, generator=g) * torch.exp(torch.randn(B, Ci, D, H, W, generator=g))
    scale, shift = 0.5 + torch.rand(Co, generator=g), 0.1 * torch.randn(Co, generator=g)
    if kind == "s2":
        w = torch.randn(Co, Ci, 3, 3, 3, generator=g) / (Ci * 27) ** 0.5
        ref = torch.nn.functional.conv3d(x.double(), w.double(), padding=1, stride=2)
        wp = hip.pack_conv3d_weight(gpu(w))
        run = lambda *a, **k: hip.conv3d(gpu(x.permute(0, 2, 3, 4, 1)), wp, *a, stride=2, **k)
    else:
        w = torch.randn(Ci, Co, 3, 3, 3, generator=g) / (Ci * 27 / 8) ** 0.5
        ref = torch.nn.functional.conv_transpose3d(x.double(), w.double(), padding=1, stride=2, output_padding=1)
        wp = hip.pack_conv3d_weight(gpu(w), transposed=True)
        run = lambda *a, **k: hip.deconv3d(gpu(x.permute(0, 2, 3, 4, 1)), wp, *a, **k)
    res = torch.randn(ref.shape, generator=g)
    ref2 = torch.relu(ref * scale.double().view(1, -1, 1, 1, 1) + shift.double().view(1, -1, 1, 1, 1)) + res.double()
    outs = {}
    for name, cfg in (("x3", 0), ("fp32", 64)):
        try:
            hip.force_direct_conv(cfg)
            outs[name] = (run().cpu().permute(0, 4, 1, 2, 3).double(),
                          run(gpu(scale), gpu(shift), gpu(res.permute(0, 2, 3, 4, 1)), relu=True).cpu().permute(0, 4, 1, 2, 3).double())
        finally:
            hip.force_direct_conv(0)
    assert outs["x3"][0].shape == ref.shape
    mag = float(ref.abs().max())
    for i, r in enumerate((ref, ref2)):
        e_x3, e_32 = float((outs["x3"][i] - r).abs().max()), float((outs["fp32"][i] - r).abs().max())
        assert e_x3 <= 2.0 * e_32 + 1e-7 * mag and e_x3 < 3e-6 * mag, (i, e_x3, e_32, mag)


def test_conv3d_x3h_marching_forms_of_the_layers_that_moved_to_the_tile_kernel(hip):
    """conv3 / conv4 / conv9 (16->32 stride 2, 32->32, 32->16 transposed) run on the tile-owning kernel of csrc/conv3d_deep.hip since the end of
    round 4; RCMVS_DEEP3 / 4 / 9 = 0 send them back to the plane-marching fp16-pair kernels of csrc/conv3d_x3.hip.  The switches are read once per
    process, so the same parity cases run once more in a child process with the three switches off."""
    import subprocess, sys
    if DEV == "cpu":
        pytest.skip("GPU only: a child pytest process on the kernel emulation would take minutes")
    env = dict(os.environ, RCMVS_DEEP3="0", RCMVS_DEEP4="0", RCMVS_DEEP9="0", RCMVS_Z8_CONV2="0", RCMVS_ZS2="0")      # (+ conv2, 16 -> 16, and conv1, 8 -> 16 stride 2, which run on the z-streaming kernels of csrc/conv3d_z8.hip / conv3d_zs2.hip since round 6)
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.abspath(__file__), "-q", "-m", "gpu", "-p", "no:cacheprovider", "-k",
                        "test_conv3d_x3h_vs_fp64 and (32-32-s1 or 32-16-t2 or 16-32-s2 or 16-16-s1 or 8-16-s2)"], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and " passed" in r.stdout and "failed" not in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]


@pytest.mark.parametrize("Ci,Co,kind", [(8, 8, "s1"), (16, 8, "s1"), (32, 8, "s1"), (16, 16, "s1"), (32, 32, "s1"), (8, 16, "s2"), (16, 32, "s2"),
                                        (16, 8, "t2"), (32, 16, "t2"), (32, 64, "s2"), (64, 64, "s1"), (64, 32, "t2")])
@pytest.mark.parametrize("shape", [(2, 5, 11, 21), (1, 9, 37, 70), (1, 1, 13, 9)])
def test_conv3d_x3h_vs_fp64(hip, Ci, Co, kind, shape):
    """The fp16-pair form of the matrix-core kernels (csrc/conv3d_x3.hip, NP = 2, and the deep-level tile kernels of csrc/conv3d_deep.hip
    -- 32->64 stride 2, 64->64, 64->32 transposed, 32->32; the one-plane shape drops their out-of-volume taps: two fp16 pieces per operand after a power-of-two
    pre-scale taken from the caller's bound of max|x|, three MFMAs per product) against an fp64 convolution: as close as the fp32
    FMA-chain kernels are, on log-normal inputs (three decades of dynamic range), ragged tiles, z chunks, batch 2, the full
    epilogue.  The bound it returns (y_absmax) is max|y| exactly; a bound that is 16x too loose changes nothing measurable; and
    the result does not depend on the magnitude of the tensor (inputs scaled by 2^40: exactly the scaled output)."""
    deep = (Ci, Co, kind) in ((32, 64, "s2"), (64, 64, "s1"), (64, 32, "t2"), (32, 32, "s1"), (32, 16, "t2"), (16, 32, "s2"))      # csrc/conv3d_deep.hip
    if DEV == "cpu" and (shape[2] > 30 or (Ci, Co, kind) not in ((8, 8, "s1"), (16, 8, "s1"), (8, 16, "s2"), (16, 8, "t2"), (32, 64, "s2"), (64, 64, "s1"), (64, 32, "t2"), (32, 32, "s1"), (32, 16, "t2"), (16, 32, "s2"))
                         or (deep and Ci == 64 and shape[0] == 2)) \
            and os.environ.get("RCMVS_EMU_FULL", "0") != "1":
        pytest.skip("up to a minute on the kernel emulation: RCMVS_EMU_FULL=1 (always run on the GPU)")
    g = torch.Generator().manual_seed(Ci * 7 + Co + shape[2])
    B, D, H, W = shape
    x = torch.randn(B, Ci, D, H, W, generator=g) * torch.exp(torch.randn(B, Ci, D, H, W, generator=g))
    scale, shift = 0.5 + torch.rand(Co, generator=g), 0.1 * torch.randn(Co, generator=g)
    if kind == "t2":
        w = torch.randn(Ci, Co, 3, 3, 3, generator=g) / (Ci * 27 / 8) ** 0.5
        ref = torch.nn.functional.conv_transpose3d(x.double(), w.double(), padding=1, stride=2, output_padding=1)
        wp = hip.pack_conv3d_weight(gpu(w), transposed=True)
        run = lambda xin, *a, **k: hip.deconv3d(xin, wp, *a, **k)
    else:
        st = 2 if kind == "s2" else 1
        w = torch.randn(Co, Ci, 3, 3, 3, generator=g) / (Ci * 27) ** 0.5
        ref = torch.nn.functional.conv3d(x.double(), w.double(), padding=1, stride=st)
        wp = hip.pack_conv3d_weight(gpu(w))
        run = lambda xin, *a, **k: hip.conv3d(xin, wp, *a, stride=st, **k)
    res = torch.randn(ref.shape, generator=g)
    ref2 = torch.relu(ref * scale.double().view(1, -1, 1, 1, 1) + shift.double().view(1, -1, 1, 1, 1)) + res.double()
    xcl, rcl = gpu(x.permute(0, 2, 3, 4, 1)), gpu(res.permute(0, 2, 3, 4, 1))
    xmax = hip.absmax(xcl)
    assert float(xmax.max()) == float(xcl.abs().max()) and float(hip.absmax(xcl, square=True).max()) == float(xcl.abs().max() ** 2)
    back = lambda t: t.cpu().permute(0, 4, 1, 2, 3).double()
    ymax = torch.zeros(hip.ABSMAX_FLOATS, device=xcl.device)
    y_h = back(run(xcl, x_absmax=xmax))
    y2_t = run(xcl, gpu(scale), gpu(shift), rcl, relu=True, x_absmax=xmax, y_absmax=ymax)
    y2_h = back(y2_t)
    try:
        hip.force_direct_conv(64)                   # the fp32 FMA-chain kernels
        y_32, y2_32 = back(run(xcl)), back(run(xcl, gpu(scale), gpu(shift), rcl, relu=True))
    finally:
        hip.force_direct_conv(0)
    y_x3 = back(run(xcl))                            # the exact three-piece bf16 form
    mag = float(ref.abs().max())
    for got, f32, r in ((y_h, y_32, ref), (y2_h, y2_32, ref2)):
        e_h, e_32 = float((got - r).abs().max()), float((f32 - r).abs().max())
        assert e_h <= 2.0 * e_32 + 1e-7 * mag and e_h < 3e-6 * mag, (e_h, e_32, mag)
    fro = lambda a: float((a - ref).norm() / ref.norm())
    print(f"x3h {Ci}->{Co} {kind} {shape}: relative Frobenius error vs fp64: fp16 pair {fro(y_h):.2e}, bf16 triple {fro(y_x3):.2e}, fp32 FMA chain {fro(y_32):.2e}")
    assert fro(y_h) <= 2.0 * fro(y_32) + 2e-8
    assert float(ymax.max()) == float(y2_t.abs().max())
    y_loose = back(run(xcl, x_absmax=xmax * 16))
    assert float((y_loose - ref).abs().max()) <= 2.0 * float((y_32 - ref).abs().max()) + 1e-7 * mag
    big = 2.0 ** 40
    y_big = back(run(xcl * big, x_absmax=xmax * big))
    assert torch.equal(y_big, y_h * big)


@pytest.mark.parametrize("shape,zchunk", [((4, 6, 14), 0), ((4, 6, 14), 4), ((2, 13, 31), 0), ((1, 3, 5), 0), ((5, 7, 16), 2), ((6, 20, 30), 6), ((12, 32, 40), 0)])
def test_conv11_prob_vs_fp64(hip, shape, zchunk):
    """conv11 + prob in one pass (csrc/conv11_prob.hip: x = conv0 + relu(bn(deconv(t))), logits = prob(x), models/modules.py:497-500; the
    8-channel volume between the two layers never reaches memory) against an fp64 evaluation of the two layers, next to the two-launch form
    it replaces (fp16-pair transposed conv + the fp32 prob conv) and the all-fp32 kernels: ragged tiles (the 12 x 28 tile does not divide the
    sizes), z chunks with their halo steps, one-plane-pair volumes, log-normal inputs.  The derived bound of the intermediate volume
    (CostRegNet._conv11_bound_coef) really is a bound."""
    if DEV == "cpu" and shape[1] > 16 and os.environ.get("RCMVS_EMU_FULL", "0") != "1":
        pytest.skip("a minute on the kernel emulation: RCMVS_EMU_FULL=1 (always run on the GPU)")
    from rc_mvsnet_amd.casmvsnet import CostRegNet
    Dt, Ht, Wt = shape
    g = torch.Generator().manual_seed(Dt * 100 + Wt)
    net = CostRegNet(8, 8)
    with torch.no_grad():
        for prm in net.parameters():
            prm.copy_(torch.randn(prm.shape, generator=g) * (0.2 if prm.dim() > 1 else 0.5))
        net.conv11.bn.running_var.copy_(0.5 + torch.rand(8, generator=g))
        net.conv11.bn.running_mean.copy_(0.3 * torch.randn(8, generator=g))
        net.conv11.bn.weight.copy_(0.5 + torch.rand(8, generator=g))
    net = net.to(DEV).eval()
    t = torch.randn(1, 16, Dt, Ht, Wt, generator=g) * torch.exp(0.7 * torch.randn(1, 16, Dt, Ht, Wt, generator=g))
    r0 = torch.randn(1, 8, 2 * Dt, 2 * Ht, 2 * Wt, generator=g) * torch.exp(0.7 * torch.randn(1, 8, 2 * Dt, 2 * Ht, 2 * Wt, generator=g))
    with torch.no_grad():
        bn = net.conv11.bn
        sc = (bn.weight / torch.sqrt(bn.running_var + bn.eps)).cpu().double()
        sh = bn.bias.cpu().double() - bn.running_mean.cpu().double() * sc
        mid = torch.nn.functional.conv_transpose3d(t.double(), net.conv11.conv.weight.cpu().double(), padding=1, stride=2, output_padding=1)
        mid = torch.relu(mid * sc.view(1, -1, 1, 1, 1) + sh.view(1, -1, 1, 1, 1))
        x8 = mid + r0.double()
        ref = torch.nn.functional.conv3d(x8, net.prob.weight.cpu().double(), padding=1)[:, 0]
        plan = net.hip_plan()
        tcl, rcl = gpu(t.permute(0, 2, 3, 4, 1)), gpu(r0.permute(0, 2, 3, 4, 1))
        tmax, rmax = hip.absmax(tcl), hip.absmax(rcl)
        coef = plan["conv11_coef"].cpu().double()
        assert float(mid.abs().max()) <= float(coef[0]) * float(t.abs().max()) + float(coef[1])        # the derived bound holds
        got = hip.conv11_prob(tcl, tmax, plan["conv11"][0], plan["conv11"][1], plan["conv11"][2], rcl, rmax, plan["conv11_coef"], plan["prob"],
                              zchunk=zchunk).cpu().double()
        # the two launches it replaces: fp16-pair transposed conv, then the prob conv as an fp32 convolution
        x8_h = hip.deconv3d(tcl, *plan["conv11"], residual=rcl, relu=True, x_absmax=tmax)
        two = hip.conv3d(x8_h, plan["prob"]).cpu().double()[..., 0]
        try:
            hip.force_direct_conv(64)               # the fp32 FMA-chain kernels for both layers
            f32 = hip.conv3d(hip.deconv3d(tcl, *plan["conv11"], residual=rcl, relu=True), plan["prob"]).cpu().double()[..., 0]
        finally:
            hip.force_direct_conv(0)
        if Dt == 4:                 # eight planes: the head in the same launch = softmax / soft-argmin / confidence over the logits above, bit for bit
            planes = gpu(torch.stack((400.0 + 50.0 * torch.rand(1, 2 * Ht, 2 * Wt, generator=g), 1.0 + torch.rand(1, 2 * Ht, 2 * Wt, generator=g)), dim=-1))
            d1, c1 = hip.conv11_prob(tcl, tmax, plan["conv11"][0], plan["conv11"][1], plan["conv11"][2], rcl, rmax, plan["conv11_coef"], plan["prob"], planes=planes)
            d2, c2 = hip.softmax_head(gpu(got.float()), planes)
            assert torch.equal(d1, d2) and torch.equal(c1, c2)
    mag = float(ref.abs().max())
    e_f, e_2, e_32 = (float((a - ref).abs().max()) for a in (got, two, f32))
    print(f"conv11+prob {shape} zchunk {zchunk}: max error vs fp64 / max|logit|: fused {e_f / mag:.2e}, two launches {e_2 / mag:.2e}, fp32 kernels {e_32 / mag:.2e}")
    assert e_f <= 2.0 * e_32 + 1e-7 * mag and e_f < 3e-6 * mag, (e_f, e_32, mag)


@pytest.mark.parametrize("Ci,Co", [(8, 8), (16, 16), (32, 32), (32, 16), (64, 32)])
@pytest.mark.parametrize("shape", [(3, 11, 21), (2, 37, 70), (1, 8, 32)])
def test_conv3d_x3_planar_vs_fp64(hip, Ci, Co, shape):
    """A one-plane volume (the FeatureNet 3x3 layers run as 3-D convs with D = 1) goes to the planar form of the split-bf16
    kernel, which only holds the kd = 1 taps: the weight's other planes are random here and must not matter (they only ever see
    padding).  Against fp64, full epilogue, and the flipped-tap adjoint image (training data gradient)."""
    if DEV == "cpu" and Ci >= 32 and shape[1] > 30 and os.environ.get("RCMVS_EMU_FULL", "0") != "1":
        pytest.skip("half a minute on the kernel emulation: RCMVS_EMU_FULL=1 (always run on the GPU)")
    g = torch.Generator().manual_seed(Ci + shape[1])
    B, H, W = shape
    x = torch.randn(B, Ci, 1, H, W, generator=g) * torch.exp(torch.randn(B, Ci, 1, H, W, generator=g))
    w = torch.randn(Co, Ci, 3, 3, 3, generator=g) / (Ci * 9) ** 0.5
    scale, shift = 0.5 + torch.rand(Co, generator=g), 0.1 * torch.randn(Co, generator=g)
    res = torch.randn(B, Co, 1, H, W, generator=g)
    ref = torch.nn.functional.conv3d(x.double(), w.double(), padding=1)
    ref2 = torch.relu(ref * scale.double().view(1, -1, 1, 1, 1) + shift.double().view(1, -1, 1, 1, 1)) + res.double()
    wp = hip.pack_conv3d_weight(gpu(w))
    xcl, rcl = gpu(x.permute(0, 2, 3, 4, 1)), gpu(res.permute(0, 2, 3, 4, 1))
    outs = {}
    for name, cfg in (("x3", 0), ("fp32", 64)):
        try:
            hip.force_direct_conv(cfg)
            outs[name] = (hip.conv3d(xcl, wp).cpu().permute(0, 4, 1, 2, 3).double(),
                          hip.conv3d(xcl, wp, gpu(scale), gpu(shift), rcl, relu=True).cpu().permute(0, 4, 1, 2, 3).double())
        finally:
            hip.force_direct_conv(0)
    mag = float(ref.abs().max())
    for i, r in enumerate((ref, ref2)):
        e_x3, e_32 = float((outs["x3"][i] - r).abs().max()), float((outs["fp32"][i] - r).abs().max())
        assert e_x3 <= 2.0 * e_32 + 1e-7 * mag and e_x3 < 3e-6 * mag, (i, e_x3, e_32, mag)
    if Ci != Co:
        return
    wt = torch.randn(Ci, Co, 3, 3, 3, generator=g) / (Ci * 9) ** 0.5
    refT = torch.nn.functional.conv_transpose3d(x.double(), wt.double(), padding=1)
    yT = hip.conv3d(xcl, hip.pack_conv3d_weight(gpu(wt), transposed=2)).cpu().permute(0, 4, 1, 2, 3).double()
    assert float((yT - refT).abs().max()) < 3e-6 * float(refT.abs().max())


@pytest.mark.parametrize("Ci,Co,kind", [(8, 8, "s1"), (16, 8, "s1"), (32, 8, "s1"), (16, 16, "s1"), (8, 16, "s2"), (16, 32, "s2"), (16, 8, "t2"),
                                        (32, 32, "s1"), (32, 16, "t2"), (32, 16, "p1"), (64, 32, "p1"), (8, 8, "p1"), (16, 16, "p1"), (32, 32, "p1")])
def test_conv3d_x3_item_schedule_is_bit_exact(hip, Ci, Co, kind):
    """The split-bf16 kernel is persistent: one block walks several (batch, tile, z chunk) work items with the LDS ring running
    across item boundaries.  Whatever the block count (3: many items per block, round robin; 8 / 16: the XCD-contiguous order;
    default: one block per CU), every output voxel sees the same arithmetic: results must be bit-identical."""
    if DEV == "cpu" and (Ci, Co, kind) in ((32, 16, "t2"), (32, 32, "s1"), (64, 32, "p1"), (16, 16, "s1"), (32, 8, "s1")) and os.environ.get("RCMVS_EMU_FULL", "0") != "1":
        pytest.skip("a minute on the kernel emulation: RCMVS_EMU_FULL=1 (always run on the GPU)")
    g = torch.Generator().manual_seed(Ci + Co)
    B, D, H, W = (3, 1, 17, 35) if kind == "p1" else (2, 4, 9, 35)      # p1: one-plane volumes take the planar kernel
    x = gpu((torch.randn(B, D, H, W, Ci, generator=g) * torch.exp(torch.randn(B, D, H, W, Ci, generator=g))).contiguous())
    scale, shift = gpu(0.5 + torch.rand(Co, generator=g)), gpu(0.1 * torch.randn(Co, generator=g))
    if kind == "t2":
        wp = hip.pack_conv3d_weight(gpu(torch.randn(Ci, Co, 3, 3, 3, generator=g) / (Ci * 27 / 8) ** 0.5), transposed=True)
        res = gpu(torch.randn(B, 2 * D, 2 * H, 2 * W, Co, generator=g))
        run = lambda: hip.deconv3d(x, wp, scale, shift, res, relu=True)
    else:
        st = 2 if kind == "s2" else 1
        wp = hip.pack_conv3d_weight(gpu(torch.randn(Co, Ci, 3, 3, 3, generator=g) / (Ci * 27) ** 0.5))
        res = gpu(torch.randn(B, (D - 1) // st + 1, (H - 1) // st + 1, (W - 1) // st + 1, Co, generator=g))
        run = lambda: hip.conv3d(x, wp, scale, shift, res, stride=st, relu=True)
    base = run().cpu()
    assert torch.isfinite(base).all()
    for blocks in ((3, 8) if DEV == "cpu" else (3, 8, 16, 1)):      # (the emulated device has 6 CUs: 3 = several items per block, 8 = the XCD-contiguous order)
        try:
            hip.force_direct_conv(blocks << 8)          # bits 8-15 of the debug selector: cap on the x3 block count
            y = run().cpu()
        finally:
            hip.force_direct_conv(0)
        assert torch.equal(y, base), (blocks, float((y - base).abs().max()))


@pytest.mark.parametrize("Ci", [8, 16, 32])
@pytest.mark.parametrize("shape", [(1, 1, 9, 35), (1, 2, 17, 40), (2, 5, 9, 70), (1, 11, 20, 33)])
def test_conv0_stream_cuts_are_bit_exact(hip, Ci, shape):
    """conv0's z-streaming kernel (csrc/conv3d_z8.hip, fp16-pair form) cuts the flattened (tile, z) sequence into one range per block and
    never feeds the planes outside the volume (a whole tile is D ticks; the tick of plane D - 1 stores two output planes).  Whatever the
    block count -- 1: every tile whole; 3 / 7: cuts inside tiles, items that start or end at either face of the volume, one-plane items;
    default: one step per block on these sizes -- every output voxel sees the same arithmetic: results are bit-identical, and they are
    the fp64 convolution to the pair form's accuracy."""
    if DEV == "cpu" and (Ci == 32 or shape[1] > 2) and os.environ.get("RCMVS_EMU_FULL", "0") != "1":
        pytest.skip("a minute on the kernel emulation: RCMVS_EMU_FULL=1 (always run on the GPU)")
    B, D, H, W = shape
    g = torch.Generator().manual_seed(Ci + D)
    x = torch.randn(B, Ci, D, H, W, generator=g) * torch.exp(torch.randn(B, Ci, D, H, W, generator=g))
    w = torch.randn(8, Ci, 3, 3, 3, generator=g) / (Ci * 27) ** 0.5
    scale, shift = gpu(0.5 + torch.rand(8, generator=g)), gpu(0.1 * torch.randn(8, generator=g))
    ref = torch.relu(torch.nn.functional.conv3d(x.double(), w.double(), padding=1) * scale.cpu().double().view(1, -1, 1, 1, 1) + shift.cpu().double().view(1, -1, 1, 1, 1))
    xcl = gpu(x.permute(0, 2, 3, 4, 1))
    wp = hip.pack_conv3d_weight(gpu(w))
    xmax = hip.absmax(xcl)
    ymax = torch.zeros(hip.ABSMAX_FLOATS, device=xcl.device)
    base = hip.conv3d(xcl, wp, scale, shift, relu=True, x_absmax=xmax, y_absmax=ymax)
    assert float(ymax.max()) == float(base.abs().max())
    err = float((base.cpu().permute(0, 4, 1, 2, 3).double() - ref).abs().max())
    assert err < 3e-6 * float(ref.abs().max()), err
    for blocks in (1, 3, 7):
        try:
            hip.force_direct_conv(blocks << 8)          # bits 8-15 of the debug selector: cap on the block count
            y = hip.conv3d(xcl, wp, scale, shift, relu=True, x_absmax=xmax)
        finally:
            hip.force_direct_conv(0)
        assert torch.equal(y, base), (blocks, float((y - base).abs().max()))


@pytest.mark.parametrize("shape", [(1, 1, 9, 35), (1, 2, 17, 40), (2, 5, 9, 70), (1, 8, 20, 33), (1, 11, 33, 130)])
def test_conv1_stream_cuts_are_bit_exact(hip, shape):
    """conv1 (8 -> 16, stride 2) on its z-streaming kernel (csrc/conv3d_zs2.hip, fp16-pair form): a tick is an output plane = an input-plane pair, an item
    that starts inside a tile opens with a lead tick, planes outside the volume (z = -1; z = D for odd D) are never requested.  Whatever the block count
    -- 1: whole tiles; 3 / 7: cuts inside tiles; default -- every output voxel sees the same arithmetic: results are bit-identical, equal to the split
    kernel's (RCMVS_ZS2=0 is a child-process switch: here the fp64 convolution stands in, to the pair form's accuracy); even and odd depths, one-plane
    volumes, ragged tiles, batch 2."""
    if DEV == "cpu" and shape[1] * shape[3] > 300 and os.environ.get("RCMVS_EMU_FULL", "0") != "1":
        pytest.skip("a minute on the kernel emulation: RCMVS_EMU_FULL=1 (always run on the GPU)")
    B, D, H, W = shape
    g = torch.Generator().manual_seed(D * 31 + W)
    x = torch.randn(B, 8, D, H, W, generator=g) * torch.exp(torch.randn(B, 8, D, H, W, generator=g))
    w = torch.randn(16, 8, 3, 3, 3, generator=g) / (8 * 27) ** 0.5
    scale, shift = gpu(0.5 + torch.rand(16, generator=g)), gpu(0.1 * torch.randn(16, generator=g))
    ref = torch.relu(torch.nn.functional.conv3d(x.double(), w.double(), padding=1, stride=2) * scale.cpu().double().view(1, -1, 1, 1, 1) + shift.cpu().double().view(1, -1, 1, 1, 1))
    xcl = gpu(x.permute(0, 2, 3, 4, 1))
    wp = hip.pack_conv3d_weight(gpu(w))
    xmax = hip.absmax(xcl)
    ymax = torch.zeros(hip.ABSMAX_FLOATS, device=xcl.device)
    base = hip.conv3d(xcl, wp, scale, shift, stride=2, relu=True, x_absmax=xmax, y_absmax=ymax)
    assert tuple(base.shape) == (B, (D - 1) // 2 + 1, (H - 1) // 2 + 1, (W - 1) // 2 + 1, 16)
    assert float(ymax.max()) == float(base.abs().max())
    err = float((base.cpu().permute(0, 4, 1, 2, 3).double() - ref).abs().max())
    assert err < 3e-6 * float(ref.abs().max()), err
    for blocks in (1, 3, 7):
        try:
            hip.force_direct_conv(blocks << 8)          # bits 8-15 of the debug selector: cap on the block count
            y = hip.conv3d(xcl, wp, scale, shift, stride=2, relu=True, x_absmax=xmax)
        finally:
            hip.force_direct_conv(0)
        assert torch.equal(y, base), (blocks, float((y - base).abs().max()))


def test_prob_conv_z_chunk_does_not_change_the_result(hip):
    """The marching prob conv picks its z chunk per launch (grid fill); a block stages chunk + 2 planes, every output voxel still sums
    the same three planes in the same order: any chunk length (debug selector bits 16-23) must give bit-identical logits."""
    g = torch.Generator().manual_seed(21)
    x = gpu(torch.randn(1, 13, 20, 40, 8, generator=g))
    w = hip.pack_conv3d_weight(gpu(torch.randn(1, 8, 3, 3, 3, generator=g) / 15))
    base = hip.conv3d(x, w).cpu()
    for zc in (2, 3, 5, 13, 16):
        try:
            hip.force_direct_conv(zc << 16)
            y = hip.conv3d(x, w).cpu()
        finally:
            hip.force_direct_conv(0)
        assert torch.equal(y, base), zc


@pytest.mark.parametrize("shape", [(2, 11, 13, 45), (1, 20, 9, 70), (1, 3, 8, 32), (1, 8, 24, 33)])
def test_prob_conv_marching_kernel(hip, shape):
    """The 8 -> 1 prob conv runs on the plane-marching kernel (one staged plane feeds the three kd taps through rolling
    accumulators): against fp64, against the tile kernel it replaces, ragged tiles, z chunks (D > 8), batch 2, epilogue."""
    g = torch.Generator().manual_seed(shape[1])
    B, D, H, W = shape
    x = torch.randn(B, 8, D, H, W, generator=g) * torch.exp(torch.randn(B, 8, D, H, W, generator=g))
    w = torch.randn(1, 8, 3, 3, 3, generator=g) / 216 ** 0.5
    scale, shift = 0.5 + torch.rand(1, generator=g), 0.1 * torch.randn(1, generator=g)
    res = torch.randn(B, 1, D, H, W, generator=g)
    ref = torch.nn.functional.conv3d(x.double(), w.double(), padding=1)
    ref2 = torch.relu(ref * scale.double() + shift.double()) + res.double()
    wp = hip.pack_conv3d_weight(gpu(w))
    xcl, rcl = gpu(x.permute(0, 2, 3, 4, 1)), gpu(res.permute(0, 2, 3, 4, 1))
    outs = {}
    for name, cfg in (("march", 0), ("tile", 2)):
        try:
            hip.force_direct_conv(cfg)
            outs[name] = (hip.conv3d(xcl, wp).cpu().permute(0, 4, 1, 2, 3).double(),
                          hip.conv3d(xcl, wp, gpu(scale), gpu(shift), rcl, relu=True).cpu().permute(0, 4, 1, 2, 3).double())
        finally:
            hip.force_direct_conv(0)
    mag = float(ref.abs().max())
    for i, r in enumerate((ref, ref2)):
        e_m, e_t = float((outs["march"][i] - r).abs().max()), float((outs["tile"][i] - r).abs().max())
        assert e_m <= 2.0 * e_t + 1e-7 * mag and e_m < 3e-6 * mag, (i, e_m, e_t, mag)


@pytest.mark.parametrize("C,Co", [(8, 16), (16, 32)])
@pytest.mark.parametrize("shape", [(3, 22, 70), (1, 38, 36), (2, 8, 32)])
def test_conv2d_s2d_is_the_5x5_stride2_layer(hip, C, Co, shape):
    """rcmvs_conv2d_s2d_fwd: the FeatureNet 5x5 stride-2 layers as a 3x3 conv of the space-to-depth view of the input, read by the
    planar split-bf16 kernel straight from the un-rearranged map.  Against nn.Conv2d(k = 5, stride 2, pad 2) in fp64, with the
    folded-BN epilogue, and against the same kernel on the materialised view (bit-identical)."""
    from rc_mvsnet_amd.casmvsnet import FeatureNet
    from rc_mvsnet_amd._lib import RcmvsError
    N, H, W = shape
    g = torch.Generator().manual_seed(C + H)
    x = torch.randn(N, C, H, W, generator=g) * torch.exp(torch.randn(N, C, H, W, generator=g))
    w5 = torch.randn(Co, C, 5, 5, generator=g) / (25 * C) ** 0.5
    scale, shift = 0.5 + torch.rand(Co, generator=g), 0.1 * torch.randn(Co, generator=g)
    ref = torch.relu(torch.nn.functional.conv2d(x.double(), w5.double(), stride=2, padding=2) * scale.double().view(1, -1, 1, 1)
                     + shift.double().view(1, -1, 1, 1))
    wp = hip.pack_conv3d_weight(gpu(FeatureNet._w3(FeatureNet._w5s2(w5)).contiguous()))
    xcl = gpu(x.permute(0, 2, 3, 1))
    y = hip.conv2d_s2d(xcl, wp, gpu(scale), gpu(shift), relu=True)
    assert tuple(y.shape) == (N, H // 2, W // 2, Co)
    err = float((y.cpu().permute(0, 3, 1, 2).double() - ref).abs().max())
    assert err < 3e-6 * float(ref.abs().max()), err
    y2 = hip.conv3d(FeatureNet._s2d(xcl).contiguous().unsqueeze(1), wp, gpu(scale), gpu(shift), relu=True).squeeze(1)
    assert torch.equal(y, y2)
    with pytest.raises(RcmvsError):
        hip.conv2d_s2d(xcl[:, :H - 1].contiguous(), wp)            # odd height


@pytest.mark.parametrize("Ci", [8, 16, 32, 44])
def test_conv3d_lds_halo_kernel(hip, Ci):
    """Cout = 8 stride-1 layers run on the LDS-staged halo kernel: against the oracle (ragged tiles: sizes
    that are not multiples of the 2x8x16 tile, batch 2, full epilogue) and against the direct kernel."""
    from oracle import conv3d as oc
    g = torch.Generator().manual_seed(Ci)
    x = torch.randn(2, Ci, 5, 11, 21, generator=g)
    w = torch.randn(8, Ci, 3, 3, 3, generator=g) / (Ci * 27) ** 0.5
    scale, shift = 0.5 + torch.rand(8, generator=g), 0.1 * torch.randn(8, generator=g)
    ref = oc.conv3d(x, w)
    res = torch.randn_like(ref)
    ref2 = torch.relu(ref * scale.view(1, -1, 1, 1, 1) + shift.view(1, -1, 1, 1, 1)) + res
    wp = hip.pack_conv3d_weight(gpu(w))
    xcl = gpu(x.permute(0, 2, 3, 4, 1))
    y = hip.conv3d(xcl, wp)
    y2 = hip.conv3d(xcl, wp, gpu(scale), gpu(shift), gpu(res.permute(0, 2, 3, 4, 1)), relu=True)
    assert rel_err(y.cpu().permute(0, 4, 1, 2, 3), ref) < 2e-5
    assert rel_err(y2.cpu().permute(0, 4, 1, 2, 3), ref2) < 2e-5
    try:
        hip.force_direct_conv(True)
        yd = hip.conv3d(xcl, wp)
    finally:
        hip.force_direct_conv(False)
    assert rel_err(y.cpu(), yd.cpu()) < 1e-5


@pytest.mark.parametrize("Ci,Co", [(64, 32), (32, 16), (16, 8)])
def test_deconv3d_vs_oracle(hip, Ci, Co):
    from oracle import conv3d as oc
    g = torch.Generator().manual_seed(Ci + Co)
    x = torch.randn(2, Ci, 3, 5, 6, generator=g)
    w = torch.randn(Ci, Co, 3, 3, 3, generator=g) / (Ci * 27 / 8) ** 0.5
    ref = oc.conv_transpose3d(x, w)
    wp = hip.pack_conv3d_weight(gpu(w), transposed=True)
    y = hip.deconv3d(gpu(x.permute(0, 2, 3, 4, 1)), wp).cpu().permute(0, 4, 1, 2, 3)
    assert y.shape == ref.shape
    assert rel_err(y, ref) < 2e-5


def test_conv3d_golden_and_linearity(hip):
    g = load_golden("conv_raw")
    wp = hip.pack_conv3d_weight(gpu(g["w"]))
    xcl = gpu(g["x"].permute(0, 2, 3, 4, 1))
    y = hip.conv3d(xcl, wp, stride=2)
    assert rel_err(y.cpu().permute(0, 4, 1, 2, 3), g["y"]) < 2e-5
    wpt = hip.pack_conv3d_weight(gpu(g["wt"]), transposed=True)
    yt = hip.deconv3d(y, wpt)
    assert rel_err(yt.cpu().permute(0, 4, 1, 2, 3), g["yt"]) < 2e-5
    assert torch.equal(hip.conv3d(xcl * 2, wp, stride=2), y * 2)      # exact: power-of-two scaling


def test_costreg_vs_golden(hip):
    from rc_mvsnet_amd import synthetic
    from rc_mvsnet_amd.casmvsnet import CostRegNet
    g = load_golden("costreg_eval")
    sd = synthetic.cost_reg_state_dict(np.random.RandomState(3), "cr", 8)
    net = CostRegNet(8, 8)
    net.load_state_dict({k[3:]: v for k, v in sd.items()}, strict=True)
    net = net.to(DEV).eval()
    with torch.no_grad():
        out = net(gpu(g["x"]))
    assert out.shape == g["out"].shape
    assert rel_err(out.cpu(), g["out"]) < 5e-5


# ------------------------------------------------------------------------------------------ 2-D pyramid
@pytest.mark.parametrize("Ci,Co,K,stride", [(3, 8, 3, 1), (8, 8, 3, 1), (8, 16, 5, 2), (16, 16, 3, 1), (16, 32, 5, 2), (32, 32, 3, 1),
                                            (32, 32, 1, 1), (16, 32, 1, 1), (32, 16, 3, 1), (8, 32, 1, 1), (32, 8, 3, 1)])
def test_conv2d_vs_torch_cpu(hip, Ci, Co, K, stride):
    import torch.nn.functional as F
    g = torch.Generator().manual_seed(Ci * 10 + Co)
    x = torch.randn(2, Ci, 38, 52, generator=g)               # ragged tiles, batch 2
    w = torch.randn(Co, Ci, K, K, generator=g) / (Ci * K * K) ** 0.5
    scale, shift = 0.5 + torch.rand(Co, generator=g), 0.1 * torch.randn(Co, generator=g)
    ref = F.conv2d(x, w, None, stride, K // 2)
    Cip = 4 if Ci == 3 else Ci
    wp = hip.pack_conv2d_weight(gpu(w), pad_in_to=Cip)
    xcl = gpu(x) if Ci != 3 else None
    xin = hip.rgb_to_nhwc4(gpu(x)) if Ci == 3 else gpu(x.permute(0, 2, 3, 1))
    y = hip.conv2d(xin, wp, stride=stride)
    assert rel_err(y.cpu().permute(0, 3, 1, 2), ref) < 2e-5
    y2 = hip.conv2d(xin, wp, gpu(scale), gpu(shift), stride=stride, relu=True)
    assert rel_err(y2.cpu().permute(0, 3, 1, 2), torch.relu(ref * scale.view(1, -1, 1, 1) + shift.view(1, -1, 1, 1))) < 2e-5
    if stride == 1:                                           # FPN merge: nearest x2 up-sample add + bias
        up = torch.randn(2, Co, 19, 26, generator=g)
        y3 = hip.conv2d(xin, wp, None, gpu(shift), up_add=gpu(up.permute(0, 2, 3, 1)), stride=1)
        ref3 = F.interpolate(up, scale_factor=2, mode="nearest") + (ref + shift.view(1, -1, 1, 1))
        assert rel_err(y3.cpu().permute(0, 3, 1, 2), ref3) < 2e-5


def test_feature_net_vs_oracle(hip):
    from rc_mvsnet_amd import synthetic
    from rc_mvsnet_amd.casmvsnet import CascadeMVSNet_eval
    from oracle.feature_net import feature_net
    sd = synthetic.cascade_state_dict(0)
    m = CascadeMVSNet_eval()
    m.load_state_dict(sd, strict=True)
    m = m.to(DEV).eval()
    for (H, W) in ((64, 96), (512, 640)):
        imgs = synthetic.images(1, 3, H, W, 0)[0]
        with torch.no_grad():
            ref = feature_net(imgs, sd)
            out = m.feature.forward_cl(gpu(imgs))
        for k in ref:
            e = (out[k].cpu().permute(0, 3, 1, 2) - ref[k]).abs()
            print(f"FeatureNet {H}x{W} {k}: max err {float(e.max()):.3e} mean {float(e.mean()):.3e} (|ref| max {float(ref[k].abs().max()):.2f})")
            assert float(e.max()) < 5e-5 * max(1.0, float(ref[k].abs().max()))


def test_feature_net_fused_kernels_match_the_one_layer_launches(hip, monkeypatch):
    """FeatureNet with its fused / tile kernels (conv0.0 -> conv0.1, conv1.0 and out2, conv1.1 -> conv1.2, conv2.1 / conv2.2: csrc/conv2d_stem.hip, conv2d_tile.hip,
    conv2d_pair.hip) against the same network with every one of them switched off (RCMVS_CONV_STEM / _TILE / _PAIR = 0: the one-layer launches they replace), on image
    sizes that are ragged against every tile shape (14 x 30, 8 x 32, 8 x 30 pixels; sizes are multiples of 4 as the pyramid requires) and batches of 1 - 3 images:
    the three stage maps agree to fp32 rounding (the exact split-bf16 arithmetic on both sides; the first layer an fp32 FMA chain on both)."""
    from rc_mvsnet_amd import synthetic, casmvsnet
    from rc_mvsnet_amd.casmvsnet import CascadeMVSNet_eval
    m = CascadeMVSNet_eval()
    m.load_state_dict(synthetic.cascade_state_dict(0), strict=True)
    m = m.to(DEV).eval()
    sizes = ((1, 36, 52), (3, 64, 100), (2, 92, 124)) if DEV != "cpu" else ((1, 36, 52),)
    for (N, H, W) in sizes:
        imgs = gpu(synthetic.images(1, N, H, W, 3)[0])
        outs = []
        for on in (True, False):
            for name in ("CONV_STEM", "CONV_TILE", "CONV_PAIR"):
                monkeypatch.setattr(casmvsnet, name, on)
            m.feature._plan = None                          # (the switches are read when the plan is built)
            with torch.no_grad():
                outs.append({k: v.clone() for k, v in m.feature.forward_cl(imgs).items()})
        m.feature._plan = None
        assert ("stem" in m.feature.hip_plan()) == bool(casmvsnet.CONV_STEM)
        for k in outs[0]:
            a, b = outs[0][k], outs[1][k]
            err = float((a - b).abs().max()) / max(1.0, float(b.abs().max()))
            assert a.shape == b.shape and err < 2e-6, (N, H, W, k, err)
    m.feature._plan = None


def _golden_state(g, prefix):
    return {k[len(prefix):]: torch.as_tensor(np.asarray(v)) for k, v in g.items() if k.startswith(prefix)}


def _standalone_block(name):
    from rc_mvsnet_amd import casmvsnet as C
    return {"conv3d_s2": lambda: C.Conv3d(8, 16, stride=2, padding=1),
            "conv3d_norelu": lambda: C.Conv3d(16, 16, relu=False, padding=1),
            "deconv3d": lambda: C.Deconv3d(16, 8, stride=2, padding=1, output_padding=1),
            "conv2d_5x5s2": lambda: C.Conv2d(16, 32, 5, stride=2, padding=2),
            "conv2d_bias": lambda: C.Conv2d(8, 8, 3, 1, padding=1, bn=False),
            "conv2d_rgb": lambda: C.Conv2d(3, 8, 3, 1, padding=1),
            "conv2d_1x1": lambda: C.Conv2d(16, 32, 1, relu=False),
            "deconv2d": lambda: C.Deconv2d(32, 16, 3, stride=2, padding=1, output_padding=1)}[name]()


@pytest.mark.parametrize("name", ["conv3d_s2", "conv3d_norelu", "deconv3d", "conv2d_5x5s2", "conv2d_bias", "conv2d_rgb", "conv2d_1x1", "deconv2d"])
def test_standalone_blocks_vs_reference_golden(hip, name):
    """The building blocks of models/modules.py called on their own (eval mode): same constructor arguments, the reference's state
    dict, the reference's output (tests/golden/blocks.npz, made by importing the reference)."""
    g = load_golden("blocks")
    m = _standalone_block(name)
    m.load_state_dict(_golden_state(g, name + ":sd:"), strict=True)
    m = m.to(DEV).eval()
    with torch.no_grad():
        y = m(gpu(g[name + ":x"]))
    assert tuple(y.shape) == tuple(g[name + ":y"].shape)
    assert rel_err(y.cpu(), g[name + ":y"]) < 2e-5


def test_deconv2d_fuse_and_unet_pyramid_vs_reference_golden(hip):
    """DeConv2dFuse (models/modules.py:342-360) and the reference's non-default feature pyramid, FeatureNet(arch_mode='unet')
    (models/modules.py:363-464), 3 and 2 stages, against the imported reference."""
    from rc_mvsnet_amd import casmvsnet as C, synthetic
    g = load_golden("blocks")
    fuse = C.DeConv2dFuse(32, 16, 3)
    fuse.load_state_dict(_golden_state(g, "fuse:sd:"), strict=True)
    fuse = fuse.to(DEV).eval()
    with torch.no_grad():
        y = fuse(gpu(g["fuse:x_pre"]), gpu(g["fuse:x"]))
    assert rel_err(y.cpu(), g["fuse:y"]) < 2e-5
    img = synthetic.images(1, 2, 32, 48, int(g["image_seed"]))[0]
    for ns in (3, 2):
        net = C.FeatureNet(base_channels=8, num_stage=ns, arch_mode="unet")
        net.load_state_dict(synthetic.feature_unet_state_dict(int(g["unet_seed"]), 8, ns), strict=True)
        net = net.to(DEV).eval()
        assert net.out_channels == [32, 16, 8][:ns]
        with torch.no_grad():
            out = net(gpu(img))
        assert sorted(out) == [f"stage{k + 1}" for k in range(ns)]
        for k, v in out.items():
            assert tuple(v.shape) == tuple(g[f"unet{ns}:{k}"].shape)
            assert rel_err(v.cpu(), g[f"unet{ns}:{k}"]) < 5e-5, (ns, k)
        net.train()
        with pytest.raises(Exception, match="fpn"):
            net(gpu(img))


def test_cascade_on_the_unet_pyramid_vs_reference_golden(hip):
    """CascadeMVSNet_eval(arch_mode='unet') end to end against the imported reference (smooth probability head)."""
    from rc_mvsnet_amd import synthetic
    from rc_mvsnet_amd.casmvsnet import CascadeMVSNet_eval
    g = load_golden("blocks")
    m = CascadeMVSNet_eval(ndepths=[16, 8, 8], depth_interals_ratio=[4, 2, 1], arch_mode="unet")
    m.load_state_dict(synthetic.cascade_unet_state_dict(3), strict=True)
    m = m.to(DEV).eval()
    imgs, pm, dv = synthetic.cascade_inputs(1, 3, 64, 96, 2)
    with torch.no_grad():
        out = m(gpu(imgs), {k: gpu(v) for k, v in pm.items()}, gpu(dv))
    dd = (out["depth"].cpu() - g["cascade_unet:depth"]).abs()
    rng = float(dv.max() - dv.min())
    print(f"cascade on the unet pyramid: depth L1/range = {float(dd.mean()) / rng:.3e}, max {float(dd.max()):.3e} mm")
    assert float(dd.mean()) / rng < 1e-4
    stable = dd < 0.05
    assert float(stable.float().mean()) > 0.97
    assert float(((out["photometric_confidence"].cpu() - g["cascade_unet:conf"]).abs()[stable] > 1e-3).float().mean()) < 0.02


def test_execution_plans_follow_their_parameters(hip):
    """The packed-weight plans of CostRegNet / FeatureNet are validated per call against (address, version) of every parameter and
    buffer they were built from (read straight from the module dictionaries): an in-place update, load_state_dict, a REPLACED Parameter
    object and a replaced BatchNorm child all rebuild the plan; an untouched module reuses it."""
    from rc_mvsnet_amd import casmvsnet as C, synthetic
    torch.manual_seed(3)
    cr = C.CostRegNet(8, 8).to(DEV).eval()
    fn = C.FeatureNet(8, num_stage=3, arch_mode="fpn").to(DEV).eval()
    x = gpu(torch.randn(1, 8, 8, 16, 24))                    # (the shapes of costreg_eval.npz / test_feature_net_vs_oracle)
    img = gpu(synthetic.images(1, 1, 64, 96, 1)[0])
    with torch.no_grad():
        y0, f0 = cr(x), fn(img)
        p_cr, p_fn = cr.hip_plan(), fn.hip_plan()
        assert cr.hip_plan() is p_cr and fn.hip_plan() is p_fn and torch.equal(cr(x), y0)
        cr.conv3.conv.weight.mul_(1.5)                                   # in place: version bump
        assert cr.hip_plan() is not p_cr and not torch.equal(cr(x), y0)
        cr.conv3.conv.weight.div_(1.5)
        p_cr = cr.hip_plan()
        cr.conv5.conv.weight = torch.nn.Parameter(cr.conv5.conv.weight.detach() * 2.0)      # a new Parameter object
        assert cr.hip_plan() is not p_cr
        p_cr = cr.hip_plan()
        cr.conv0.bn = torch.nn.BatchNorm3d(8).to(DEV).eval()             # a replaced child module (what SyncBatchNorm conversion does)
        assert cr.hip_plan() is not p_cr
        sd = {k: v.clone() for k, v in fn.state_dict().items()}
        sd["out3.weight"] *= 0.5
        fn.load_state_dict(sd)                                           # copies in place: version bump
        assert fn.hip_plan() is not p_fn
        f1 = fn(img)
        assert torch.equal(f1["stage1"], f0["stage1"]) and not torch.equal(f1["stage3"], f0["stage3"])
        p_fn = fn.hip_plan()
        fn.conv1[1].bn.running_var.add_(0.25)                            # a buffer of the trunk
        assert fn.hip_plan() is not p_fn


def test_depthnet_on_its_own_vs_reference_golden(hip):
    """DepthNet_eval / DepthNet (models/casmvsnet.py:45-124, 234-311) called like the reference calls them -- a list of per-view maps,
    (B,V,2,4,4) projections, a (B,D,h,w) sample volume, a CostRegNet -- against the imported reference, eval mode (where the train
    variant's volume_feature_no_ref carries the reference's in-place pow_ quirk)."""
    from rc_mvsnet_amd import casmvsnet as C, synthetic
    g = load_golden("blocks")
    B, V, Cf, D, h, w = 1, 3, 8, 8, 16, 24
    feats = [gpu(g[f"depthnet:feat{v}"]) for v in range(V)]
    pm = gpu(synthetic.proj_matrices(B, V, h, w)["stage3"])
    dv = gpu(g["depthnet:depth_values"])
    imgs = gpu(synthetic.images(B, V, h, w, 4))
    cr = C.CostRegNet(Cf, 8)
    cr.load_state_dict({k[2:]: v for k, v in synthetic.cost_reg_state_dict(np.random.RandomState(12), "x", Cf, prob_gain=1.0).items()}, strict=True)
    cr = cr.to(DEV).eval()
    rng = float(dv.max() - dv.min())
    with torch.no_grad():
        o = C.DepthNet(False).eval()(feats, pm, dv, D, cr, imgs)
        t = C.DepthNet(True).eval()(feats, pm, dv, D, cr, imgs)
        assert sorted(o) == ["depth", "photometric_confidence"] and sorted(t) == ["depth", "photometric_confidence", "volume_feature_no_ref"]
        assert float((o["depth"].cpu() - g["depthnet:depth"]).abs().mean()) / rng < 1e-4
        assert float((o["photometric_confidence"].cpu() - g["depthnet:conf"]).abs().mean()) < 1e-4
        assert torch.equal(t["depth"], o["depth"])
        assert float((t["depth"].cpu() - g["depthnet:depth_t"]).abs().mean()) / rng < 1e-4
        assert rel_err(t["volume_feature_no_ref"].cpu(), g["depthnet:noref"]) < 2e-5
        with pytest.raises(Exception, match="uniformly spaced"):
            C.DepthNet(False).eval()(feats, pm, dv * dv / 500.0, D, cr, imgs)


def test_standalone_conv3d_block_trains_like_torch(hip):
    """Conv3d / Deconv3d called on their own in train mode: batch-statistics BatchNorm, autograd through the HIP kernels, running
    statistics updated -- against the same block in plain PyTorch (fp64) on the host."""
    from rc_mvsnet_amd import casmvsnet as C
    g = torch.Generator().manual_seed(11)
    for make, shape in ((lambda: C.Conv3d(8, 16, stride=2, padding=1), (2, 8, 8, 8, 8)),
                        (lambda: C.Deconv3d(16, 8, stride=2, padding=1, output_padding=1), (1, 16, 4, 4, 8))):
        m = make()
        ref = torch.nn.Sequential(type(m.conv)(m.conv.in_channels, m.conv.out_channels, 3, stride=m.conv.stride, padding=1, bias=False,
                                               **({"output_padding": 1} if isinstance(m.conv, torch.nn.ConvTranspose3d) else {})),
                                  torch.nn.BatchNorm3d(m.conv.out_channels, momentum=0.1)).double()
        ref[0].weight.data.copy_(m.conv.weight.data)
        x = torch.randn(*shape, generator=g)
        gy = torch.randn(*ref(x.double()).shape, generator=g)
        ref.zero_grad()
        ref[1].running_mean.zero_(); ref[1].running_var.fill_(1.0)
        xr = x.double().requires_grad_(True)
        yr = torch.relu(ref(xr))
        (yr * gy.double()).sum().backward()
        m = m.to(DEV).train()
        xg = gpu(x).requires_grad_(True)
        y = m(xg)
        (y * gpu(gy)).sum().backward()
        assert rel_err(y.detach().cpu(), yr.detach()) < 5e-5
        assert rel_err(xg.grad.cpu(), xr.grad) < 2e-4
        assert rel_err(m.conv.weight.grad.cpu(), ref[0].weight.grad) < 2e-4
        assert rel_err(m.bn.running_var.cpu(), ref[1].running_var) < 1e-5


# ------------------------------------------------------------------------------------------ K4
@pytest.mark.parametrize("D,h,w", [(8, 12, 16), (48, 16, 24), (32, 9, 130)])
def test_depth_head_vs_oracle(hip, D, h, w):
    from oracle import conv3d as oc, depth_head as od
    g = torch.Generator().manual_seed(D)
    x = torch.randn(2, 8, D, h, w, generator=g)
    wprob = torch.randn(1, 8, 3, 3, 3, generator=g) * 0.5
    planes = torch.stack((425.0 + 50 * torch.rand(2, h, w, generator=g), 1.0 + 5 * torch.rand(2, h, w, generator=g)), dim=-1)
    k = torch.arange(D, dtype=torch.float32).reshape(1, D, 1, 1)
    samples = planes[..., 0].unsqueeze(1) + k * planes[..., 1].unsqueeze(1)
    logits = oc.conv3d(x, wprob).squeeze(1)
    depth_ref, conf_ref, p_ref = od.depth_head(logits, samples)
    wp = hip.pack_conv3d_weight(gpu(wprob))
    depth, conf, prob = hip.depth_head(gpu(x.permute(0, 2, 3, 4, 1)), wp, gpu(planes), want_prob=True)
    assert rel_err(prob.cpu(), p_ref) < 2e-5
    assert float((depth.cpu() - depth_ref).abs().max()) < 2e-3                # mm, of ~600
    assert abs(float(prob.sum(1).mean()) - 1.0) < 1e-5
    fidx = od.depth_regression(p_ref, torch.arange(D, dtype=torch.float32))
    safe = (fidx - fidx.round()).abs() > 1e-3
    assert float((conf.cpu() - conf_ref).abs()[safe].max()) < 1e-4


@pytest.mark.parametrize("D,h,w", [(8, 12, 16), (8, 21, 70), (48, 16, 24), (32, 9, 130)])
def test_depth_head_forms_are_bit_identical(hip, D, h, w):
    """The depth head's own marching prob conv (buffer loads / stores, nothing predicated) gives the bits of the generic one, and the
    one-launch form of the last stage (D = 8: logits kept in registers, softmax / soft-argmin / confidence in the same thread) the
    bits of the two launches -- with and without the probability volume."""
    g = torch.Generator().manual_seed(D + h)
    x = gpu(torch.randn(2, D, h, w, 8, generator=g))
    wp = hip.pack_conv3d_weight(gpu(torch.randn(1, 8, 3, 3, 3, generator=g) * 0.5))
    planes = gpu(torch.stack((425.0 + 50 * torch.rand(2, h, w, generator=g), 1.0 + 5 * torch.rand(2, h, w, generator=g)), dim=-1))
    outs = {}
    try:
        for impl in (0, 1, 2, 3):
            hip.DEPTH_HEAD_IMPL = impl
            outs[impl] = [t.cpu() for t in hip.depth_head(x, wp, planes, want_prob=True)]
        hip.DEPTH_HEAD_IMPL = 0
        outs["noprob"] = [t.cpu() for t in hip.depth_head(x, wp, planes)]
    finally:
        hip.DEPTH_HEAD_IMPL = 0
    for key in (1, 2, 3):
        for a, b_ in zip(outs[0], outs[key]):
            assert torch.equal(a, b_), (key, float((a - b_).abs().max()))
    assert torch.equal(outs["noprob"][0], outs[0][0]) and torch.equal(outs["noprob"][1], outs[0][1])


@pytest.mark.parametrize("D,h,w,zc", [(8, 12, 16, 0), (8, 21, 70, 0), (48, 16, 24, 0), (32, 9, 130, 0), (32, 9, 45, 5), (16, 8, 33, 3), (7, 10, 31, 2)])
def test_depth_head_matrix_core_form_vs_fp32_form(hip, D, h, w, zc):
    """The prob conv on the matrix cores (csrc/prob_pair.hip: M = (kd, kw) partial sums, three rotated weight images, fp16 pairs after a
    power-of-two pre-scale from the caller's bound) against the exact fp32 head: logits to fp32-chain accuracy on log-normal inputs
    (three decades of range), so probabilities, depth and confidence agree to rounding; every z chunk length and start phase of the
    rotation, ragged tiles (widths that are not multiples of 14 or 32), batch 2, the one-launch form (D = 8) and the two-launch form,
    and a bound that is 8x too loose."""
    if DEV == "cpu" and h * w * D > 13000 and os.environ.get("RCMVS_EMU_FULL", "0") != "1":
        pytest.skip("tens of seconds on the kernel emulation: RCMVS_EMU_FULL=1 (always run on the GPU)")
    g = torch.Generator().manual_seed(D * 3 + w)
    x = gpu(torch.randn(2, D, h, w, 8, generator=g) * torch.exp(torch.randn(2, D, h, w, 8, generator=g)))
    wprob = torch.randn(1, 8, 3, 3, 3, generator=g) * 0.1
    wp = hip.pack_conv3d_weight(gpu(wprob))
    planes = gpu(torch.stack((425.0 + 50 * torch.rand(2, h, w, generator=g), 1.0 + 5 * torch.rand(2, h, w, generator=g)), dim=-1))
    bound = hip.absmax(x)
    d32, c32, p32 = hip.depth_head(x, wp, planes, want_prob=True)
    logits64 = torch.nn.functional.conv3d(x.cpu().permute(0, 4, 1, 2, 3).double(), wprob.double(), padding=1).squeeze(1)
    p64 = torch.softmax(logits64, dim=1)
    try:
        hip.DEPTH_HEAD_IMPL = zc << 8
        for bnd in (bound, bound * 8.0):
            dh, ch, ph = hip.depth_head(x, wp, planes, want_prob=True, x_absmax=bnd)
            e_h, e_32 = float((ph.cpu().double() - p64).abs().max()), float((p32.cpu().double() - p64).abs().max())
            assert e_h <= 2.0 * e_32 + 2e-6, (e_h, e_32)
            assert float((dh - d32).abs().max()) < 2e-3                          # mm, of ~600
            fidx = (p64 * torch.arange(D, dtype=torch.float64).view(1, D, 1, 1)).sum(1)
            safe = (fidx - fidx.round()).abs() > 1e-3
            assert float((ch.cpu() - c32.cpu()).abs()[safe].max()) < 1e-4
        if D == 8:
            d2, c2 = hip.depth_head(x, wp, planes, x_absmax=bound)             # without the probability volume
            dh, ch, _ = hip.depth_head(x, wp, planes, want_prob=True, x_absmax=bound)
            assert torch.equal(d2, dh) and torch.equal(c2, ch)
            hip.DEPTH_HEAD_IMPL = 1                                             # two launches on the same logits: the same bits
            d3, c3, p3 = hip.depth_head(x, wp, planes, want_prob=True, x_absmax=bound)
            assert torch.equal(d3, dh) and torch.equal(c3, ch)
    finally:
        hip.DEPTH_HEAD_IMPL = 0


def test_depth_head_golden(hip):
    g = load_golden("depth_head")
    # feed the golden logits through a 1-hot prob conv: x channel 0 = logits, centre tap weight 1
    logits = g["logits"]
    B, D, h, w = logits.shape
    x = torch.zeros(B, D, h, w, 8)
    x[..., 0] = logits
    wprob = torch.zeros(1, 8, 3, 3, 3)
    wprob[0, 0, 1, 1, 1] = 1.0
    s = g["samples"]
    planes = torch.stack((s[:, 0], s[:, 1] - s[:, 0]), dim=-1)
    depth, conf, prob = hip.depth_head(gpu(x), hip.pack_conv3d_weight(gpu(wprob)), gpu(planes), want_prob=True)
    assert rel_err(prob.cpu(), g["prob"]) < 1e-5
    safe = (g["fidx"] - g["fidx"].round()).abs() > 1e-3
    assert float((conf.cpu() - g["conf"]).abs()[safe].max()) < 1e-4


# ------------------------------------------------------------------------------------------ end to end
def _run_cascade(name, train_variant=False):
    from rc_mvsnet_amd import synthetic
    from rc_mvsnet_amd.casmvsnet import CascadeMVSNet_eval
    g = load_golden(name)
    H, W, V = int(g["H"]), int(g["W"]), int(g["V"])
    nd, ra = [int(v) for v in g["ndepths"]], [int(v) for v in g["ratios"]]
    sd = synthetic.cascade_state_dict(0, prob_gain=float(g["prob_gain"])) if "prob_gain" in g else synthetic.cascade_state_dict(0)
    if len(nd) == 1:
        sd = {k: v for k, v in sd.items() if not (k.startswith("feature.inner") or k.startswith("feature.out2")
                                                   or k.startswith("feature.out3") or k.startswith("cost_regularization.1")
                                                   or k.startswith("cost_regularization.2"))}
    m = CascadeMVSNet_eval(ndepths=nd, depth_interals_ratio=ra, cr_base_chs=[8] * len(nd))
    m.load_state_dict(sd, strict=True)
    m = m.to(DEV).eval()
    imgs, pm, dv = synthetic.cascade_inputs(1, V, H, W, 0)
    with torch.no_grad():
        out = m(gpu(imgs), {k: gpu(v) for k, v in pm.items()}, gpu(dv))
    return g, out, float(dv[0, -1] - dv[0, 0])


@pytest.mark.parametrize("name", ["cascade_c1", "cascade_small", "cascade_v5", "cascade_v7_d64", "cascade_c2"])
def test_cascade_vs_reference_golden(hip, name):
    g, out, rng = _run_cascade(name)
    err = float((out["depth"].cpu() - g["depth"]).abs().mean()) / rng
    mx = float((out["depth"].cpu() - g["depth"]).abs().max())
    print(f"{name}: depth L1/range = {err:.3e}  max|dd| = {mx:.3e} mm")
    assert err < 1e-4
    # The seeded weights give a sharply peaked, chaotic probability volume (prob.weight x20): a
    # 1-ulp change of a stage-1 logit can move a later stage's hypothesis range at isolated pixels.
    # Bound how many pixels do that, and require the confidence to agree on all the others.
    dd = (out["depth"].cpu() - g["depth"]).abs()
    stable = dd < 0.05                                           # mm
    print(f"{name}: unstable pixels = {1 - float(stable.float().mean()):.4f}")
    # config 2 sits at ~11 % because the reference's fp32 torch.inverse homography differs from the
    # fp64 composition by ~1e-5 px (test_cascade_c2_hot_path_isolated pins that down)
    assert float(stable.float().mean()) > (0.85 if name == "cascade_c2" else 0.97)
    cd = (out["photometric_confidence"].cpu() - g["conf"]).abs()
    assert float((cd[stable] > 1e-3).float().mean()) < (0.08 if name == "cascade_c2" else 0.02)
    assert set(out.keys()) >= {"depth", "photometric_confidence", "stage1"}


def test_cascade_c2_smooth_head_vs_reference_golden(hip):
    """BASELINE config 2 at full size against the imported reference (tests/golden/cascade_c2_smooth.npz, make_golden.py --full)
    with a well-conditioned, trained-like probability head (prob.weight x1): the soft-argmin then follows the logits smoothly, so
    the headline parity is not hostage to knife-edge pixels -- depth L1 an order of magnitude inside the 1e-4 tolerance and
    >= 99 % of the pixels within 0.05 mm, confidences agreeing on those."""
    g, out, rng = _run_cascade("cascade_c2_smooth")
    dd = (out["depth"].cpu() - g["depth"]).abs()
    err = float(dd.mean()) / rng
    stable = dd < 0.05
    cd = (out["photometric_confidence"].cpu() - g["conf"]).abs()
    print(f"cascade_c2_smooth: depth L1/range = {err:.3e}  max|dd| = {float(dd.max()):.3e} mm  stable = {float(stable.float().mean()):.5f}  "
          f"conf max diff on stable = {float(cd[stable].max()):.2e}")
    assert err < 1e-5
    assert float(stable.float().mean()) >= 0.99
    assert float((cd[stable] > 1e-3).float().mean()) < 0.01


@pytest.mark.parametrize("gain", [0.0, 1e-4, 1.0, 3e3])
def test_cascade_fp16_pair_form_on_extreme_activation_ranges(hip, gain):
    """The fp16-pair form of the cost regularisation rests on activation bounds (a power-of-two pre-scale per layer).  Ranges the seeded
    goldens do not reach: all-zero images (every bound from constants), tiny and huge image magnitudes (feature maps 1e-4 x / 3e3 x the
    usual: variances 1e-8 x / 1e7 x) -- the depth map must stay finite and agree with the exact bf16-triple form like on ordinary inputs
    (well-conditioned head; `model.fp16_pair` selects the form per module)."""
    from rc_mvsnet_amd import synthetic
    from rc_mvsnet_amd.casmvsnet import CascadeMVSNet_eval
    sd = synthetic.cascade_state_dict(0, prob_gain=1.0)
    imgs, pm, dv = synthetic.cascade_inputs(1, 3, 64, 96, 0)
    outs = {}
    for pair in (True, False):
        m = CascadeMVSNet_eval(ndepths=[48, 32, 8], depth_interals_ratio=[4, 2, 1])
        m.load_state_dict(sd, strict=True)
        m = m.to(DEV).eval()
        m.fp16_pair = pair
        with torch.no_grad():
            outs[pair] = m(gpu(imgs * gain), {k: gpu(v) for k, v in pm.items()}, gpu(dv))
    rng = float(dv[0, -1] - dv[0, 0])
    for key in ("depth", "photometric_confidence"):
        assert torch.isfinite(outs[True][key]).all() and torch.isfinite(outs[False][key]).all(), key
    dd = (outs[True]["depth"] - outs[False]["depth"]).abs()
    stable = dd < 0.05
    print(f"gain {gain:g}: pair vs exact depth L1/range = {float(dd.mean()) / rng:.2e} (stable pixels: {float(dd[stable].mean()) / rng:.2e}), "
          f"max {float(dd.max()):.2e} mm, pixels off by more than 0.05 mm: {int((~stable).sum())} of {dd.numel()}")
    # At gain 3e3 the network's logits are 1e7 x the usual: the softmax is a hard arg-max, and a stage-1 pixel (there are 384 of them at
    # this size) whose two best hypotheses tie flips by a plane with ANY 1e-7 change of the arithmetic -- then fans out 4 x per stage
    # into ~1.5 % of the final map.  Measured (tools/dev/k1_gain_probe.py): ONE such stage-1 pixel, with either K1 form; with round 5's
    # K1 forms its flip is 2.8 mm instead of 0.18 mm and crosses the 0.05 mm line at stage 3.  So: the ordinary gains must agree
    # everywhere; the extreme ones on 97 % of the pixels (two such ties), and to 1e-5 of the range there.
    if gain <= 1.0:
        assert float(dd.mean()) / rng < 1e-5 and float(stable.float().mean()) >= 0.99
    else:
        assert float(stable.float().mean()) >= 0.97 and float(dd[stable].mean()) / rng < 1e-5
        # ... and the WHOLE map stays bounded: a regression that moves many pixels by millimetres must not pass as "unstable pixels".  Measured
        # (round 6, with and without the fused conv11 + prob pass): 90 of 6 144 pixels, 5.07e-5 of the range over the whole map, largest flip
        # 21.7 mm = two stage-1 plane intervals (2 x 4 x 2.65 mm).  Ceilings: twice the measured L1, three stage-1 intervals.
        assert float(dd.mean()) / rng < 1e-4 and float(dd.max()) < 32.0


def test_cascade_config5_arithmetic_vs_reference_golden(hip):
    """BASELINE config 5's arithmetic (eval_rcmvsnet_tanks.py:47,53-55: 7 views, ndepths 64,32,8) at a small size against the imported
    reference, well-conditioned head: the six-source-view form of K1 inside a cascade (two view groups of three per plane), the 64-plane
    depth head (stage-1 depth map and confidence compared on their own) and the hand-over of the stage depth maps."""
    g, out, rng = _run_cascade("cascade_v7_d64_smooth")
    for key, ref in (("depth", g["depth"]), ("stage1", g["depth1"]), ("stage2", g["depth2"])):
        got = out[key]["depth"] if key != "depth" else out["depth"]
        dd = (got.cpu() - ref).abs()
        print(f"cascade_v7_d64_smooth {key}: depth L1/range = {float(dd.mean()) / rng:.3e}  max = {float(dd.max()):.3e} mm")
        assert float(dd.mean()) / rng < 1e-5 and float((dd < 0.05).float().mean()) >= 0.99, key
    c1 = (out["stage1"]["photometric_confidence"].cpu() - g["conf1"]).abs()
    cd = (out["photometric_confidence"].cpu() - g["conf"]).abs()
    assert float((c1 > 1e-3).float().mean()) < 0.01 and float((cd > 1e-3).float().mean()) < 0.01


@pytest.mark.parametrize("name,l1_tol", [("cascade_c1", 1e-4), ("cascade_c2_smooth", 1e-5)])
def test_cascade_fp16_pair_form_vs_reference_golden(hip, monkeypatch, name, l1_tol):
    """The cost regularisation on the two-piece fp16 form of the matrix-core kernels (RCMVS_FP16_PAIR=1: bounds of the activations from
    the feature maps and from the layers' own epilogues, half the MFMAs) against the reference golden, same tolerances as the exact
    three-piece form; and the two forms against each other on the well-conditioned head."""
    if DEV == "cpu" and name != "cascade_c1":
        pytest.skip("full size: GPU only")
    monkeypatch.setenv("RCMVS_FP16_PAIR", "1")
    g, out, rng = _run_cascade(name)
    dd = (out["depth"].cpu() - g["depth"]).abs()
    err = float(dd.mean()) / rng
    print(f"{name} (fp16 pair): depth L1/range = {err:.3e}  max|dd| = {float(dd.max()):.3e} mm  stable = {float((dd < 0.05).float().mean()):.5f}")
    assert err < l1_tol
    assert float((dd < 0.05).float().mean()) >= (0.99 if name == "cascade_c2_smooth" else 0.97)
    if name == "cascade_c2_smooth":
        monkeypatch.setenv("RCMVS_FP16_PAIR", "0")
        _, exact, _ = _run_cascade(name)
        d2 = (out["depth"] - exact["depth"]).abs()
        print(f"  fp16 pair vs exact triple: depth mean |d| = {float(d2.mean()):.3e} mm, max {float(d2.max()):.3e} mm")
        assert float(d2.mean()) / rng < 2e-6


C2_MARGIN_L1 = 7.5e-5          # depth L1 / range of cascade_c2 must stay below this (north_star's tolerance: 1e-4)
C2_MARGIN_STABLE = 0.88        # ... and this share of its pixels within 0.05 mm of the reference (measured 0.8906 in round 5, minus 1 %)


def test_cascade_c2_parity_margin_is_pinned(hip, monkeypatch):
    """The headline scene (prob.weight x20, a chaotic soft-argmin) sits at ~6.5e-5 of the depth range from the fp32
    reference, with 1e-4 allowed: every arithmetic shortcut spends from that margin.  This test is the ceiling: it fails when the
    default configuration passes 7.5e-5 or loses more than 1 % of its stable pixels, and prints what each shortcut costs (the same scene
    with one switch set back to the exact form at a time), so that the next change shows what it spent."""
    from rc_mvsnet_amd import casmvsnet

    def measure():
        g, out, rng = _run_cascade("cascade_c2")
        dd = (out["depth"].cpu() - g["depth"]).abs()
        return float(dd.mean()) / rng, float((dd < 0.05).float().mean())

    err, stable = measure()
    print(f"cascade_c2 default: depth L1 / range {err:.3e}, stable pixels {stable:.4f}  (ceilings {C2_MARGIN_L1:.1e} / {C2_MARGIN_STABLE})")
    for label, setter in (("RCMVS_FP16_PAIR=0 (exact bf16 triple in the cost regularisation)", lambda mp: mp.setenv("RCMVS_FP16_PAIR", "0")),
                          ("DEEP_PAIR off (fp32 MFMAs at the deep levels)", lambda mp: mp.setattr(casmvsnet, "DEEP_PAIR", False)),
                          ("HEAD_PAIR off (fp32 prob conv)", lambda mp: mp.setattr(casmvsnet, "HEAD_PAIR", False))):
        with monkeypatch.context() as mp:
            setter(mp)
            e1, s1 = measure()
        print(f"    {label}: {e1:.3e}, {s1:.4f}")
    assert err <= C2_MARGIN_L1
    assert stable >= C2_MARGIN_STABLE


def test_reference_fp32_homography_depends_on_the_backend(hip):
    """Why the product does not chase the reference's fp32 `torch.inverse` homography (models/modules.py:314-316) bit for bit:
    the reference's own value depends on where it runs.  On 300 random DTU-like rigs the fp32 composition is evaluated with
    LAPACK on the host (what the golden fixtures hold) and with PyTorch-ROCm's inverse on this GPU (what the reference would
    compute on this box); both land ~1e-5 pixel from the exact (fp64) homography and as far from EACH OTHER as from the truth,
    so no single fp32 operation order is "the reference".  The product's composer (fp64 inside the kernel, rounded once) is
    the exact homography to fp32 rounding."""
    from oracle import warp
    g = torch.Generator().manual_seed(300)
    n = 300
    # random rigs: intrinsics around the DTU ones, rotations up to ~0.3 rad, centres within ~150 mm
    K = torch.eye(3).repeat(n, 1, 1)
    K[:, 0, 0] = 360.0 + 20.0 * torch.rand(n, generator=g)
    K[:, 1, 1] = K[:, 0, 0] * (1.0 + 0.01 * torch.randn(n, generator=g))
    K[:, 0, 2], K[:, 1, 2] = 80.0 + 2.0 * torch.randn(n, generator=g), 64.0 + 2.0 * torch.randn(n, generator=g)

    def extrinsic(scale):
        w = scale * torch.randn(n, 3, generator=g)
        th = w.norm(dim=1, keepdim=True).clamp_min(1e-9)
        k = w / th
        Kx = torch.zeros(n, 3, 3)
        Kx[:, 0, 1], Kx[:, 0, 2], Kx[:, 1, 0], Kx[:, 1, 2], Kx[:, 2, 0], Kx[:, 2, 1] = -k[:, 2], k[:, 1], k[:, 2], -k[:, 0], -k[:, 1], k[:, 0]
        R = torch.eye(3) + torch.sin(th)[:, :, None] * Kx + (1 - torch.cos(th))[:, :, None] * (Kx @ Kx)
        E = torch.eye(4).repeat(n, 1, 1)
        E[:, :3, :3] = R
        E[:, :3, 3] = (-R @ (150.0 * torch.randn(n, 3, 1, generator=g))).squeeze(-1)
        return E

    proj = torch.zeros(n, 2, 2, 4, 4)
    for v in range(2):
        proj[:, v, 0] = extrinsic(0.02 if v == 0 else 0.15)
        proj[:, v, 1, :3, :3] = K
    rot_t, trans_t = warp.compose_homography(proj[:, 1].double(), proj[:, 0].double())           # exact to fp64 rounding
    rot_c, trans_c = warp.compose_homography(proj[:, 1], proj[:, 0])                               # fp32, LAPACK on the host
    pg = gpu(proj)
    rot_g, trans_g = warp.compose_homography(pg[:, 1], pg[:, 0])                                   # fp32, PyTorch-ROCm on this GPU
    rot_h, trans_h = hip.compose_homography(pg)                                                    # product

    def pixels(rot, trans):          # source pixel of the reference image corner at the far plane, in fp64
        rot, trans = rot.double().cpu().reshape(n, 3, 3), trans.double().cpu().reshape(n, 3)
        p = rot @ torch.tensor([159.0, 127.0, 1.0], dtype=torch.float64) * 935.0 + trans
        return p[:, :2] / p[:, 2:3]

    truth = pixels(rot_t, trans_t)
    e_cpu = (pixels(rot_c, trans_c) - truth).norm(dim=1)
    e_gpu = (pixels(rot_g, trans_g) - truth).norm(dim=1)
    e_hip = (pixels(rot_h, trans_h) - truth).norm(dim=1)
    d_ref = (pixels(rot_c, trans_c) - pixels(rot_g, trans_g)).norm(dim=1)
    print(f"pixel error vs exact homography (median / max over {n} rigs): host fp32 {float(e_cpu.median()):.2e} / {float(e_cpu.max()):.2e}, "
          f"GPU fp32 {float(e_gpu.median()):.2e} / {float(e_gpu.max()):.2e}, product {float(e_hip.median()):.2e} / {float(e_hip.max()):.2e}; "
          f"host fp32 vs GPU fp32 {float(d_ref.median()):.2e} / {float(d_ref.max()):.2e}")
    assert float(e_hip.max()) < 2e-4 and float(e_hip.median()) <= float(e_cpu.median())           # the product is the most exact of the three
    assert float(d_ref.median()) > 0.2 * float(e_cpu.median())                                    # the two "references" disagree at the level of their own error
    assert float((d_ref > 0).float().mean()) > 0.9


def test_cascade_c2_hot_path_isolated(hip):
    """BASELINE config 2 through the HIP hot path with the two non-hot-path inputs pinned to the
    reference's own values: (a) the CPU oracle's feature maps (no MIOpen in the loop) and (b) the
    reference's fp32 homographies (torch.inverse in fp32 is ~1e-5 px off the true homography and no
    fp32 operation order reproduces it -- tools/ notes in DESIGN.md -- so the product composes in
    fp64).  With both pinned, the deviation from the reference golden must drop to the level at
    which two CPU fp32 implementations differ (oracle 'spec' vs reference: L1/range 6.4e-6, 0.9 %
    of pixels off by > 0.05 mm); each pin is also reported separately."""
    from rc_mvsnet_amd import synthetic
    from rc_mvsnet_amd.casmvsnet import CascadeMVSNet_eval
    from oracle.feature_net import feature_net
    from oracle import warp
    g = load_golden("cascade_c2")
    sd = synthetic.cascade_state_dict(0)
    m = CascadeMVSNet_eval()
    m.load_state_dict(sd, strict=True)
    m = m.to(DEV).eval()
    imgs, pm, dv = synthetic.cascade_inputs(1, 3, 512, 640, 0)
    rng = float(dv[0, -1] - dv[0, 0])
    with torch.no_grad():
        per_view = [feature_net(imgs[:, v], sd) for v in range(3)]
        feats = {k: gpu(torch.cat([f[k] for f in per_view], dim=0)) for k in per_view[0]}
        homs = {}
        for k, p in pm.items():
            rots, transs = zip(*[warp.compose_homography(p[:, v], p[:, 0]) for v in range(1, 3)])
            homs[k] = (gpu(torch.stack([r.reshape(1, 9) for r in rots], dim=1)), gpu(torch.stack(transs, dim=1)))
        gi, gp, gd = gpu(imgs), {k: gpu(v) for k, v in pm.items()}, gpu(dv)
        res = {}
        for tag, kw in (("fp64-homography + MIOpen features", {}), ("exact features only", {"features": feats}),
                        ("reference homographies only", {"homographies": homs}),
                        ("both pinned", {"features": feats, "homographies": homs})):
            out = m._forward_hip(gi, gp, gd, **kw)
            dd = (out["depth"].cpu() - g["depth"]).abs()
            cd = (out["photometric_confidence"].cpu() - g["conf"]).abs()
            res[tag] = (float(dd.mean()) / rng, float((dd > 0.05).float().mean()), float((cd[dd < 0.05] > 1e-3).float().mean()))
            print(f"c2 [{tag}]: L1/range {res[tag][0]:.3e}  pixels off > 0.05 mm {res[tag][1]:.4f}  max {float(dd.max()):.2e} mm"
                  f"  confidence mismatches on stable pixels {res[tag][2]:.4f}")
    assert res["both pinned"][0] < 2e-5
    assert res["both pinned"][1] < 0.03
    assert res["both pinned"][2] < 0.03
    assert res["fp64-homography + MIOpen features"][0] < 1e-4


def test_train_variant_volume_feature(hip):
    from rc_mvsnet_amd import synthetic
    from rc_mvsnet_amd.casmvsnet import CascadeMVSNet
    g = load_golden("train_extras")
    H, W, V = int(g["H"]), int(g["W"]), int(g["V"])
    m = CascadeMVSNet(ndepths=[8, 8, 8])
    m.load_state_dict(synthetic.cascade_state_dict(0), strict=True)
    m = m.to(DEV).eval()
    imgs, pm, dv = synthetic.cascade_inputs(1, V, H, W, 0)
    with torch.no_grad():
        out, vf = m(gpu(imgs), {k: gpu(v) for k, v in pm.items()}, gpu(dv))
    assert vf.shape == g["vf_eval"].shape
    assert rel_err(vf.cpu(), g["vf_eval"]) < 1e-4


def test_no_silent_fallback_on_cpu_tensor(hip):
    """ops refuse CPU tensors instead of computing somewhere else."""
    from rc_mvsnet_amd._lib import RcmvsError
    with pytest.raises(RcmvsError):
        hip.to_channels_last(torch.zeros(1, 4, 2, 2))


def test_tanks_and_temples_shape(hip):
    """BASELINE config 5: 7 views, 1056x1920, D = (64, 32, 8) -- the general-V K1 path (6 source views in
    LDS-sized chunks), 1 GB variance volumes, D = 64 depth head.  No reference golden at this size (a CPU
    run takes minutes): size-independent properties only -- finite, depth inside the hypothesis range of
    each stage, confidence in [0, 1], deterministic."""
    import time
    from rc_mvsnet_amd import synthetic
    from rc_mvsnet_amd.casmvsnet import CascadeMVSNet_eval
    H, W, V = 1056, 1920, 7
    m = CascadeMVSNet_eval(ndepths=[64, 32, 8], depth_interals_ratio=[4, 2, 1])
    m.load_state_dict(synthetic.cascade_state_dict(0), strict=True)
    m = m.to(DEV).eval()
    imgs, pm, dv = synthetic.cascade_inputs(1, V, H, W, 0)
    gi, gp, gd = gpu(imgs), {k: gpu(v) for k, v in pm.items()}, gpu(dv)
    with torch.no_grad():
        out = m(gi, gp, gd)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        out2 = m(gi, gp, gd)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
    print(f"T&T-shape forward: {dt * 1e3:.1f} ms")
    assert out["depth"].shape == (1, H, W)
    for k in ("stage1", "stage2", "stage3"):
        assert torch.isfinite(out[k]["depth"]).all() and torch.isfinite(out[k]["photometric_confidence"]).all()
    d1 = out["stage1"]["depth"]
    assert float(d1.min()) >= 425.0 - 1e-2 and float(d1.max()) <= 425.0 + 2.65 * 191 + 1e-2
    c = out["photometric_confidence"]
    assert float(c.min()) >= 0.0 and float(c.max()) <= 1.0 + 1e-5
    assert torch.equal(out["depth"], out2["depth"])


@pytest.mark.parametrize("C,V,D,h,w,with_noref,smooth", [(8, 3, 4, 16, 20, True, False), (16, 3, 8, 24, 40, False, False), (32, 5, 6, 12, 24, True, False),
                                                         (8, 2, 3, 7, 9, True, False), (8, 3, 6, 16, 40, True, True), (16, 4, 5, 12, 36, False, True),
                                                         (32, 3, 9, 10, 33, True, True)])
def test_warp_variance_backward_vs_oracle_autograd(hip, C, V, D, h, w, with_noref, smooth):
    """K1 backward (rcmvs_warp_variance_bwd through ops.WarpVarianceFn) against torch autograd through the
    oracle's op-by-op restatement of homo_warping + variance (oracle/warp.py), float64 on the CPU.  smooth: hypothesis planes that
    vary slowly over the image (what stage 1 always has: long runs for the kernel's run-length merging), noisy over the right third."""
    from rc_mvsnet_amd import synthetic
    from oracle import warp as ow
    gen = torch.Generator().manual_seed(C + V)
    H, W = 4 * h, 4 * w
    proj = synthetic.proj_matrices(1, V, H, W)["stage1"]
    proj[:, :, 1, :2, :3] *= (w / (W / 4.0))                        # intrinsics of an h x w map
    feats = [torch.randn(1, C, h, w, generator=gen) for _ in range(V)]
    imgs = torch.rand(1, V, 3, h, w, generator=gen)
    depth = (450.0 + 80.0 * torch.rand(1, 1, h, w, generator=gen)) + 12.0 * torch.arange(D).view(1, D, 1, 1)
    if smooth:
        yy, xx = torch.meshgrid(torch.arange(h, dtype=torch.float32), torch.arange(w, dtype=torch.float32), indexing="ij")
        base = 500.0 + 0.8 * xx + 0.5 * yy + 3.0 * torch.sin(xx / 5.0)
        base[:, w - w // 3:] += 40.0 * torch.rand(h, w // 3, generator=gen)
        depth = base.view(1, 1, h, w) + 2.5 * torch.arange(D).view(1, D, 1, 1)
    planes = torch.stack((depth[:, 0], depth[:, 1] - depth[:, 0]), dim=-1).contiguous()
    depth = planes[..., 0].unsqueeze(1) + torch.arange(D, dtype=torch.float32).view(1, D, 1, 1) * planes[..., 1].unsqueeze(1)
    gvar = torch.randn(1, D, h, w, C, generator=gen)
    gnr = torch.randn(1, 3 * (V - 1) + C, D, h, w, generator=gen)
    # ---- oracle: autograd through the restated op graph, in float64 (sampling positions from the fp32 chain)
    f64 = [f.double().requires_grad_(True) for f in feats]
    rots = [ow.compose_homography(proj[:, v], proj[:, 0]) for v in range(1, V)]
    warped = []
    for v in range(1, V):
        ix, iy = ow.warp_coords(rots[v - 1][0], rots[v - 1][1], depth, h, w)
        warped.append(ow.bilinear_gather_zeros(f64[v], ix.double(), iy.double()))
    ref = f64[0].unsqueeze(2).expand(-1, -1, D, -1, -1)
    s = ref + sum(warped)
    q = ref ** 2 + sum(t ** 2 for t in warped)
    var = q / V - (s / V) ** 2
    loss = (var.permute(0, 2, 3, 4, 1) * gvar.double()).sum()
    if with_noref:
        sn = sum(warped)
        qn = sum(t ** 2 for t in warped)
        loss = loss + ((qn / V - (sn / V) ** 2) * gnr[:, -C:].double()).sum()
    loss.backward()
    ref_grads = torch.stack([f.grad[0].permute(1, 2, 0) for f in f64]).float()          # (V,h,w,C)
    # ---- HIP
    f_cl = torch.stack([f[0].permute(1, 2, 0) for f in feats]).unsqueeze(0).contiguous()
    f_gpu = gpu(f_cl).requires_grad_(True)
    rot = gpu(torch.stack([r[0].reshape(1, 9) for r in rots], dim=1))                 # the oracle's fp32 homographies
    trans = gpu(torch.stack([r[1].reshape(1, 3) for r in rots], dim=1))
    imgs_cl = gpu(imgs.permute(0, 1, 3, 4, 2).contiguous()) if with_noref else None
    out = hip.WarpVarianceFn.apply(f_gpu, rot, trans, gpu(planes), D, imgs_cl)
    if with_noref:
        var_g, noref_g = out
        (var_g * gpu(gvar)).sum().add((noref_g * gpu(gnr)).sum()).backward()
    else:
        (out * gpu(gvar)).sum().backward()
    got = f_gpu.grad[0].cpu()
    err = rel_err(got, ref_grads)
    print(f"K1 bwd C={C} V={V}: rel err {err:.2e}")
    assert err < 2e-5


def test_cascade_batch_two_equals_two_singles(hip, monkeypatch):
    """Batch handling end to end (eval-mode BN makes samples independent): a B = 2 forward must reproduce the two B = 1
    forwards -- FeatureNet over B*V images, per-sample homographies / plane tables, batched volumes.  (On the exact arithmetic form:
    batches always take it -- the fp16-pair form scales by a per-launch activation bound -- and the chaotic test head amplifies
    any rounding difference between the two forms to millimetres at isolated pixels.)"""
    monkeypatch.setenv("RCMVS_FP16_PAIR", "0")
    from rc_mvsnet_amd import synthetic
    from rc_mvsnet_amd.casmvsnet import CascadeMVSNet_eval
    m = CascadeMVSNet_eval(ndepths=[16, 8, 8], depth_interals_ratio=[4, 2, 1])
    m.load_state_dict(synthetic.cascade_state_dict(0), strict=True)
    m = m.to(DEV).eval()
    a = synthetic.cascade_inputs(1, 3, 64, 96, 0)
    b = synthetic.cascade_inputs(1, 3, 64, 96, 1)
    imgs = gpu(torch.cat((a[0], b[0])))
    pm = {k: gpu(torch.cat((a[1][k], b[1][k]))) for k in a[1]}
    dv = gpu(torch.cat((a[2], b[2])))
    with torch.no_grad():
        o2 = m(imgs, pm, dv)
        oa = m(gpu(a[0]), {k: gpu(v) for k, v in a[1].items()}, gpu(a[2]))
        ob = m(gpu(b[0]), {k: gpu(v) for k, v in b[1].items()}, gpu(b[2]))
    for key in ("depth", "photometric_confidence"):
        assert o2[key].shape[0] == 2
        assert float((o2[key][0] - oa[key][0]).abs().max()) < 1e-3, key
        assert float((o2[key][1] - ob[key][0]).abs().max()) < 1e-3, key


def test_dtu_eval_shape(hip):
    """The reference's DTU evaluation setting (eval_rcmvsnet_dtu.py:49-51: 5 views, 1600 x 1184, D = 48/32/8): compile-time
    4-source-view K1 path, 0.7 GB stage-1 volume.  Properties only (no golden at this size)."""
    import time
    from rc_mvsnet_amd import synthetic
    from rc_mvsnet_amd.casmvsnet import CascadeMVSNet_eval
    H, W, V = 1184, 1600, 5
    m = CascadeMVSNet_eval()
    m.load_state_dict(synthetic.cascade_state_dict(0), strict=True)
    m = m.to(DEV).eval()
    imgs, pm, dv = synthetic.cascade_inputs(1, V, H, W, 0)
    gi, gp, gd = gpu(imgs), {k: gpu(v) for k, v in pm.items()}, gpu(dv)
    with torch.no_grad():
        out = m(gi, gp, gd)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        out = m(gi, gp, gd)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
    print(f"DTU-eval-shape forward: {dt * 1e3:.1f} ms")
    assert out["depth"].shape == (1, H, W)
    assert torch.isfinite(out["depth"]).all() and torch.isfinite(out["photometric_confidence"]).all()
    c = out["photometric_confidence"]
    assert float(c.min()) >= 0.0 and float(c.max()) <= 1.0 + 1e-5


@pytest.mark.parametrize("N,H,W", [(2, 36, 44), (1, 16, 16), (3, 128, 160), (1, 18, 50)])
def test_fpn_out_fused_is_bit_identical(N, H, W):
    """rcmvs_fpn_out_fused (1x1 lateral + up-add + 3x3 output conv in one launch) against the two rcmvs_conv2d_fwd calls it
    replaces: same arithmetic order, so the maps must be equal bit for bit (ragged tiles, image borders, several images)."""
    from rc_mvsnet_amd import _lib, ops
    _lib.load()
    g = torch.Generator().manual_seed(N * 1000 + H)
    dev = DEV
    lat = torch.randn(N, H, W, 8, generator=g).to(dev)
    up = torch.randn(N, H // 2, W // 2, 32, generator=g).to(dev)
    w_in = ops.pack_conv2d_weight((0.3 * torch.randn(32, 8, 1, 1, generator=g)).to(dev))
    b_in = (0.1 * torch.randn(32, generator=g)).to(dev)
    w_out = ops.pack_conv2d_weight((0.1 * torch.randn(8, 32, 3, 3, generator=g)).to(dev))
    want = ops.conv2d(ops.conv2d(lat, w_in, None, b_in, up_add=up), w_out)
    got = ops.fpn_out_fused(lat, up, w_in, b_in, w_out)
    assert got.shape == want.shape and torch.equal(got, want)
    with pytest.raises(_lib.RcmvsError):
        ops.fpn_out_fused(lat[:, :-1], up, w_in, b_in, w_out)


@pytest.mark.parametrize("N,H,W", [(3, 64, 96), (1, 17, 33), (2, 16, 16)])
def test_first_layer_reads_the_planar_images_itself(hip, N, H, W):
    """rcmvs_conv2d_fwd with Ci = 3: FeatureNet's first layer staged straight from the (N,3,H,W) input equals the NHWC4 pass + the same
    layer, bit for bit (same kernel body, same arithmetic), ragged tiles and borders included."""
    g = torch.Generator().manual_seed(N + H)
    x = gpu(torch.randn(N, 3, H, W, generator=g))
    w = hip.pack_conv2d_weight(gpu(0.2 * torch.randn(8, 3, 3, 3, generator=g)), pad_in_to=4)
    sc, sh = gpu(torch.rand(8, generator=g) + 0.5), gpu(0.1 * torch.randn(8, generator=g))
    want = hip.conv2d(hip.rgb_to_nhwc4(x), w, sc, sh, relu=True)
    got = hip.conv2d_rgb(x, w, sc, sh, relu=True)
    assert torch.equal(got, want)
    with pytest.raises(Exception):
        hip.conv2d_rgb(x[:, :2].contiguous(), w, sc, sh)


@pytest.mark.parametrize("Ci,N,H,W,up,relu", [(16, 2, 10, 14, True, False), (32, 1, 6, 22, False, True), (16, 1, 4, 6, False, False), (32, 3, 8, 18, True, True)])
def test_conv1x1_matrix_core_form(hip, Ci, N, H, W, up, relu):
    """rcmvs_conv1x1_mfma_fwd -- FeatureNet's 1x1 layers (16 -> 32 lateral merge with the nearest x2 up-add, 32 -> 32 output conv) on the matrix
    cores with exact three-piece bf16 operands -- against fp64 and against the fp32 FMA-chain kernel: as accurate, ragged n-tiles, BatchNorm /
    bias / up-add / ReLU epilogue, the squared bound of its output."""
    g = torch.Generator().manual_seed(Ci + H)
    x = torch.randn(N, H, W, Ci, generator=g) * torch.exp(torch.randn(N, H, W, Ci, generator=g))
    w = torch.randn(32, Ci, 1, 1, generator=g) / Ci ** 0.5
    sc, sh = 0.5 + torch.rand(32, generator=g), 0.1 * torch.randn(32, generator=g)
    ua = torch.randn(N, H // 2, W // 2, 32, generator=g) if up else None
    want = torch.nn.functional.conv2d(x.permute(0, 3, 1, 2).double(), w.double()).permute(0, 2, 3, 1) * sc.double() + sh.double()
    if up:
        want = want + ua.double().repeat_interleave(2, dim=1).repeat_interleave(2, dim=2)
    if relu:
        want = want.clamp_min(0)
    wp = hip.pack_conv2d_weight(gpu(w))
    args = (gpu(x), wp, gpu(sc), gpu(sh))
    bound = torch.zeros(hip.ABSMAX_FLOATS, device=DEV)
    got = hip.conv1x1(*args, up_add=gpu(ua) if up else None, relu=relu, ysq_absmax=bound, mfma=True)
    ref32 = hip.conv1x1(*args, up_add=gpu(ua) if up else None, relu=relu)
    mag = float(want.abs().max())
    e_m, e_32 = float((got.cpu().double() - want).abs().max()), float((ref32.cpu().double() - want).abs().max())
    assert e_m <= 2.0 * e_32 + 1e-7 * mag and e_m < 3e-6 * mag, (e_m, e_32, mag)
    assert float(bound.max()) == float(got.abs().max() ** 2)
    with pytest.raises(Exception):
        hip.conv1x1(gpu(x[..., :8].contiguous()), hip.pack_conv2d_weight(gpu(w[:, :8].contiguous())), mfma=True)      # 8 -> 32 has no matrix-core form


def test_feature_output_convs_keep_the_variance_bound(hip):
    """FeatureNet's three output convs leave (max|f|)^2 of their maps -- the bound of the variance volume the fp16-pair cost regularisation
    needs -- in the bound vector they are handed, bit-equal to what rcmvs_absmax_fwd(square=1) computes in a pass over the map (the launch
    the cascade no longer needs), and the maps themselves are what the bound-less thunks produce (stage 1 goes through the planar
    matrix-core form of its 1x1 conv when a bound is wanted: equal to 1e-6)."""
    from rc_mvsnet_amd import synthetic
    from rc_mvsnet_amd.casmvsnet import FeatureNet
    net = FeatureNet(base_channels=8, num_stage=3, arch_mode="fpn")
    net.load_state_dict({k[len("feature."):]: v for k, v in synthetic.cascade_state_dict(0).items() if k.startswith("feature.")}, strict=True)
    net = net.to(DEV).eval()
    img = gpu(synthetic.images(1, 3, *((32, 48) if DEV == "cpu" else (64, 96)), 3)[0])        # (a quarter of the pixels on the kernel emulation)
    with torch.no_grad():
        plain = net.forward_cl(img)
        thunks = net.forward_cl(img, lazy=True)
        for key in ("stage1", "stage2", "stage3"):
            bound = torch.zeros(hip.ABSMAX_FLOATS, device=DEV)
            f, kept = thunks[key](bound)
            assert kept, key
            want = hip.absmax(f, square=True)
            assert float(bound.max()) == float(want.max()) > 0.0, key
            assert rel_err(f.cpu(), plain[key].cpu()) < 2e-6, key
            f2, kept2 = thunks[key]()
            assert not kept2 and torch.equal(f2, plain[key]), key


@pytest.mark.parametrize("N,H,W", [(2, 36, 44), (1, 16, 16), (3, 128, 160), (1, 18, 50), (1, 2, 2), (1, 4, 34)])
def test_fpn_out_folded_matches_the_unfused_path(N, H, W):
    """rcmvs_fpn_out_folded -- the last FPN level with the 1x1 lateral conv folded into the 3x3 output conv (one 3x3 conv 8 -> 8 on the
    lateral map, one 2x2 conv 32 -> 8 per output parity on the half-resolution map, the lateral bias through the taps inside the image) --
    against the same level in PyTorch fp64 (models/modules.py:448-462) and against the two-kernel path: ragged tiles, every border class
    (first / last row and column, 2-pixel images), several images."""
    from rc_mvsnet_amd import _lib, ops
    if DEV == "cpu" and N * H * W > 10000 and os.environ.get("RCMVS_EMU_FULL", "0") != "1":
        pytest.skip("tens of seconds on the kernel emulation: RCMVS_EMU_FULL=1 (always run on the GPU)")
    _lib.load()
    g = torch.Generator().manual_seed(N * 1000 + H + W)
    lat = torch.randn(N, H, W, 8, generator=g)
    up = torch.randn(N, H // 2, W // 2, 32, generator=g)
    w_in_t, b_in_t = 0.3 * torch.randn(32, 8, 1, 1, generator=g), 0.5 * torch.randn(32, generator=g)
    w_out_t = 0.1 * torch.randn(8, 32, 3, 3, generator=g)
    F = torch.nn.functional
    intra = F.interpolate(up.permute(0, 3, 1, 2).double(), scale_factor=2, mode="nearest") + F.conv2d(lat.permute(0, 3, 1, 2).double(), w_in_t.double(), b_in_t.double())
    want = F.conv2d(intra, w_out_t.double(), padding=1).permute(0, 2, 3, 1)
    tab = ops.pack_fpn_folded(w_in_t.to(DEV), b_in_t.to(DEV), w_out_t.to(DEV))
    assert tab.numel() == ops.FPN_FOLDED_FLOATS
    got = ops.fpn_out_folded(lat.to(DEV), up.to(DEV), tab).cpu()
    scale = float(want.abs().max())
    assert float((got.double() - want).abs().max()) < 2e-6 * scale
    two = ops.conv2d(ops.conv2d(lat.to(DEV), ops.pack_conv2d_weight(w_in_t.to(DEV)), None, b_in_t.to(DEV), up_add=up.to(DEV)), ops.pack_conv2d_weight(w_out_t.to(DEV))).cpu()
    assert float((got - two).abs().max()) < 4e-6 * scale
    # the same level on the matrix cores (csrc/fpn_folded_mfma.hip: exact three-piece bf16 operands, persistent blocks), with its bound output
    img = ops.pack_fpn_folded_mfma(tab)
    bound = torch.zeros(ops.ABSMAX_FLOATS, device=DEV)
    got_m = ops.fpn_out_folded(lat.to(DEV), up.to(DEV), img, ysq_absmax=bound).cpu()
    assert float((got_m.double() - want).abs().max()) < 2e-6 * scale
    assert float(bound.max()) == float(got_m.abs().max() ** 2)          # (squared in fp32, as the kernel does)
    with pytest.raises(_lib.RcmvsError):
        ops.fpn_out_folded(lat.to(DEV)[:, :-1], up.to(DEV), tab)
