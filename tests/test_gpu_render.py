"""GPU parity tests of the rendering-consistency branch: every HIP
kernel against the CPU oracle (oracle/render.py, oracle/conv3d.py) and against the fixtures captured
from the reference's Rendering_Consistency_Net.forward with injected random draws."""
import types

import pytest
import torch

from conftest import load_golden, rel_err

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def gpu(t):
    return t.to(DEV).contiguous()


@pytest.fixture(scope="module")
def hip():
    from rc_mvsnet_amd import _lib, ops
    _lib.load()
    return ops


def _args(S):
    return types.SimpleNamespace(multires=10, i_embed=0, pts_dim=3, dir_dim=3, netdepth=6, netwidth=128, net_type="v0",
                                 netchunk=1024, ckpt=None, N_samples=S, N_importance=0, perturb=1.0, use_viewdirs=True,
                                 white_bkgd=False, raw_noise_std=0.0, pad=0, img_downscale=1.0, use_color_volume=False,
                                 multires_views=4)


def _net(S):
    from rc_mvsnet_amd import synthetic
    from rc_mvsnet_amd.render_consist_net import Rendering_Consistency_Net
    m = Rendering_Consistency_Net(_args(S))
    m.load_state_dict(synthetic.render_state_dict(1), strict=True)
    return m.to(DEV).eval()


def test_resize_planes(hip):
    from oracle import conv3d as oc
    g = torch.Generator().manual_seed(0)
    x = torch.randn(1, 41, 6, 8, 12, generator=g)
    ref = oc.resize_depth_align_corners(x, 128)
    y = hip.resize_planes(gpu(x), 128, pad_channels_to=44).cpu()
    assert y.shape == (1, 128, 8, 12, 44)
    assert float(y[..., 41:].abs().max()) == 0.0
    assert rel_err(y[..., :41].permute(0, 4, 1, 2, 3), ref) < 1e-6


def test_neural_volume_vs_golden(hip):
    g = load_golden("render")
    m = _net(16)
    with torch.no_grad():
        vol = m.MVSNet.forward_cl(gpu(g["vfw"]))               # (1,128,h,w,8)
        vol_nc = m.MVSNet(gpu(g["vfw"]))                        # reference layout through the module API
    v = vol.cpu().permute(0, 4, 1, 2, 3)
    assert rel_err(v[:, :, ::8], g["volume"]) < 5e-5
    assert torch.equal(vol_nc.cpu(), v.reshape(1, -1, *v.shape[2:]))


def _rays_case(S, seed=5, H=64, W=96, V=4):
    from rc_mvsnet_amd import synthetic
    from oracle import render as orr
    batch = synthetic.render_batch(V, H, W, 0)
    pix, eps, u = synthetic.render_randoms(H, W, 1024, S, seed)
    g = torch.Generator().manual_seed(seed)
    pseudo = 500.0 + 300.0 * torch.rand(H, W, generator=g)
    pseudo[:4] = 426.0
    imgs = orr.unpreprocess(batch["imgs"])
    w2cs, c2ws, intr, nf = batch["w2cs"][0], batch["c2ws"][0], batch["intrinsics"][0], batch["near_fars"][0]
    rays = orr.build_rays(imgs, pseudo, w2cs, c2ws, intr, nf, pix, eps, u)
    cam = torch.cat((intr[0].reshape(-1), c2ws[0].reshape(-1), w2cs[0].reshape(-1), intr[0].reshape(-1), nf[0]))
    return batch, imgs, pseudo, w2cs, c2ws, intr, nf, pix, eps, u, rays, cam


@pytest.mark.parametrize("S", [16, 128, 20])
def test_gu_sampler_vs_oracle(hip, S):
    batch, imgs, pseudo, w2cs, c2ws, intr, nf, pix, eps, u, rays, cam = _rays_case(S)
    z, pts, ndc, dirs, rdepth, target = hip.gu_sample(gpu(pseudo), gpu(imgs[0, 0]), gpu(pix.to(torch.int32)), gpu(eps), gpu(u), gpu(cam))
    assert torch.equal(rdepth.cpu(), rays["rays_depth"])
    assert torch.equal(target.cpu(), rays["target_s"])
    assert float((z.cpu() - rays["depth_candidates"]).abs().max()) < 2e-4          # mm of ~600: <= 3 ulp
    zc = z.cpu()
    assert bool((zc[:512, 1:] >= zc[:512, :-1]).all())                               # Gaussian half is sorted
    assert rel_err(dirs.cpu(), rays["rays_dir"]) < 1e-6
    assert rel_err(pts.cpu(), rays["rays_pts"]) < 1e-6
    assert float((ndc.cpu() - rays["rays_ndc"]).abs().max()) < 2e-6


def test_point_feats_vs_oracle(hip):
    from oracle import render as orr
    S = 16
    batch, imgs, pseudo, w2cs, c2ws, intr, nf, pix, eps, u, rays, cam = _rays_case(S)
    g = torch.Generator().manual_seed(3)
    vol = torch.randn(1, 8, 24, 16, 24, generator=g)
    # push some points outside the volume / images to exercise zeros / border padding and the mask
    rays["rays_ndc"][:8] = rays["rays_ndc"][:8] * 3.0 - 1.0
    rays["rays_pts"][8:16] = rays["rays_pts"][8:16] * 4.0
    ref = orr.point_features(vol, imgs[:, -3:], w2cs, intr, rays["rays_pts"], rays["rays_ndc"])
    poses = torch.cat((w2cs[:3].reshape(3, 16), intr[:3].reshape(3, 9)), dim=1)
    feat = hip.point_feats(gpu(vol[0].permute(1, 2, 3, 0)), gpu(imgs[0, -3:]), gpu(poses), gpu(rays["rays_pts"]), gpu(rays["rays_ndc"]), ldf=32)
    out = feat.cpu()[:, :20].reshape(1024, S, 20)
    assert rel_err(out[..., :8], ref[..., :8]) < 1e-5
    assert rel_err(out[..., 8:], ref[..., 8:]) < 1e-5
    assert torch.equal(out[..., 11], ref[..., 11]) and torch.equal(out[..., 19], ref[..., 19])   # masks


def test_nerf_mlp_vs_oracle(hip):
    """(the oracle MLP itself is pinned to the reference's RenderNet by tests/golden/nerf_mlp.npz on the CPU)"""
    from oracle import render as orr
    from rc_mvsnet_amd import synthetic
    sd = synthetic.render_state_dict(1)
    m = _net(16)
    gen = torch.Generator().manual_seed(11)
    N, S = 64, 16
    ndc = torch.rand(N, S, 3, generator=gen)
    feat20 = torch.randn(N * S, 20, generator=gen) * 0.5
    dirs = torch.randn(N, 3, generator=gen)
    w2c = synthetic.render_batch(4, 64, 96, 0)["w2cs"][0, 0]
    ang = (dirs / torch.norm(dirs, dim=-1, keepdim=True)) @ w2c[:3, :3].t()
    ref = orr.nerf_mlp(orr.embed(ndc).reshape(N * S, -1), feat20, ang[:, None].expand(-1, S, -1).reshape(N * S, 3), sd).reshape(N, S, 4)
    feat = torch.zeros(N * S, 32)
    feat[:, :20] = feat20
    feat[:, 20:] = 7.0                                                             # padding must be cleared by the kernel
    raw = hip.nerf_mlp(gpu(ndc), gpu(feat), gpu(dirs), gpu(w2c), m.network_fn.hip_blob()).cpu()
    print(f"MLP max err {float((raw - ref).abs().max()):.3e}")
    assert rel_err(raw, ref) < 5e-5


def test_rendernet_forward_on_its_own_vs_reference_golden(hip):
    """RenderNet.forward / Renderer_ours.forward / forward_alpha called directly (models/render_models.py:175-220,538-565) on rows that are
    already [embedded point | point feature | view direction]: against the reference's own `network_fn` on the same weights
    (tests/golden/nerf_mlp.npz), batch shape kept; the module explains itself on a foreign configuration, with gradients, off the device."""
    from rc_mvsnet_amd import synthetic
    from rc_mvsnet_amd.render_consist_net import RenderNet, Renderer_ours
    g = load_golden("nerf_mlp")
    net = RenderNet(D=6, W=128, input_ch_pts=63, input_ch_views=3, input_ch_feat=20, skips=[4], net_type="v0")
    net.load_state_dict({k[len("network_fn."):]: v for k, v in synthetic.render_state_dict(1).items() if k.startswith("network_fn.")}, strict=True)
    net = net.to(DEV).eval()
    x = gpu(g["x"])
    with torch.no_grad():
        out = net(x)
        alpha = net.forward_alpha(x[..., :83])
        out2 = net.nerf(x.reshape(-1, 86))
    assert out.shape == (64, 16, 4) and alpha.shape == (64, 16, 1)
    assert rel_err(out.cpu(), g["out"]) < 5e-5
    assert torch.equal(alpha[..., 0], out[..., 3]) and torch.equal(out2.reshape(64, 16, 4), out)
    with pytest.raises(Exception, match="no_grad|train|GPU"):
        net(x)                                                           # eval mode with autograd on
    with torch.no_grad():
        with pytest.raises(Exception, match="86"):
            net(x[..., :80])
        with pytest.raises(Exception, match="create_nerf_mvs"):
            Renderer_ours(D=8, W=256, use_viewdirs=True).to(DEV).eval()(x)


def test_composite_vs_oracle(hip):
    from oracle import render as orr
    gen = torch.Generator().manual_seed(2)
    for (N, S) in ((1024, 128), (64, 16), (10, 37)):
        raw = torch.rand(N, S, 4, generator=gen)
        raw[..., 3] = torch.relu(torch.randn(N, S, generator=gen)) * 0.3
        z = torch.sort(425 + 500 * torch.rand(N, S, generator=gen), dim=1).values
        ref = orr.composite(raw, z)
        rgb, depth, weights, alpha = hip.composite(gpu(raw), gpu(z))
        assert rel_err(alpha.cpu(), ref["alpha"]) < 1e-6
        assert rel_err(weights.cpu(), ref["weights"]) < 2e-6
        assert rel_err(rgb.cpu(), ref["rgb_map"]) < 1e-5
        assert float((depth.cpu() - ref["depth_map"]).abs().max()) < 2e-3
        assert float(weights.sum(1).max()) <= 1.0 + 1e-5


def test_render_forward_vs_reference_golden(hip):
    """Full Rendering_Consistency_Net.forward on the HIP path with the reference's captured draws."""
    from rc_mvsnet_amd import synthetic
    g = load_golden("render")
    H, W, V = int(g["H"]), int(g["W"]), int(g["V"])
    m = _net(16)
    batch = {k: gpu(v) for k, v in synthetic.render_batch(V, H, W, 0).items()}
    with torch.no_grad():
        rgb, feat, wts, dpred, alpha, _, rdepth, target = m(gpu(g["vfw"]), gpu(g["pseudo"]), batch,
                                                             randoms=(gpu(g["pix"]), gpu(g["eps"]), gpu(g["u"])))
    assert torch.equal(rdepth.cpu(), g["rays_depth"])
    assert rel_err(target.cpu(), g["target"]) < 1e-6
    # Rays through border pixels re-project onto |grid| == 1 exactly, where the strict in-bounds mask
    # (render_utils.py:273) is decided by the last ulp: compare those rays' masks statistically and
    # keep them out of the downstream comparisons.
    px, py = g["pix"][0], g["pix"][1]
    interior = (px > 0) & (px < W - 1) & (py > 0) & (py < H - 1)
    f, gf = feat.cpu()[::4], g["feat"]
    mask_cols = [11, 15, 19]
    other = [c for c in range(20) if c not in mask_cols]
    assert rel_err(f[..., other], gf[..., other]) < 2e-4
    assert float((f[..., mask_cols] != gf[..., mask_cols]).float().mean()) < 0.01
    assert torch.equal(f[interior[::4]][..., mask_cols], gf[interior[::4]][..., mask_cols])
    assert rel_err(alpha.cpu()[interior], g["alpha"][interior]) < 5e-4
    assert rel_err(wts.cpu()[interior], g["weights"][interior]) < 5e-4
    assert rel_err(rgb.cpu()[interior], g["rgb"][interior]) < 5e-4
    dd = (dpred.cpu() - g["depth"]).abs()[interior]
    print(f"render depth max err {float(dd.max()):.3e} mm over {int(interior.sum())} interior rays")
    assert float(dd.max()) / 500.0 < 5e-4


@pytest.mark.parametrize("ci,co,stride", [(41, 8, 1), (8, 16, 2), (16, 16, 1)])
def test_conv_bn_relu3d_block_on_its_own(hip, ci, co, stride):
    """ConvBnReLU3D (models/render_models.py:675-686: conv + norm, no ReLU) called as a module of its own -- what `CostReg` runs through
    its plan -- against the same layer in PyTorch-CPU fp64."""
    from rc_mvsnet_amd.render_consist_net import ConvBnReLU3D
    torch.manual_seed(ci + co)
    m = ConvBnReLU3D(ci, co, stride=stride)
    with torch.no_grad():
        m.bn.running_mean.normal_(0, 0.1); m.bn.running_var.uniform_(0.5, 1.5); m.bn.weight.uniform_(0.5, 1.5); m.bn.bias.normal_(0, 0.1)
    x = torch.randn(1, ci, 8, 16, 24)
    with torch.no_grad():
        want = torch.nn.functional.batch_norm(torch.nn.functional.conv3d(x.double(), m.conv.weight.double(), stride=stride, padding=1),
                                              m.bn.running_mean.double(), m.bn.running_var.double(), m.bn.weight.double(), m.bn.bias.double(),
                                              False, 0.0, m.bn.eps)
        got = m.to(DEV).eval()(gpu(x))
        again = m(gpu(x))
    assert tuple(got.shape) == tuple(want.shape) and torch.equal(got, again)
    assert rel_err(got.cpu().double(), want) < 2e-5


def test_render_forward_five_view_extension_vs_reference_golden(hip):
    """The flagged extension `args.num_views = 5` (BASELINE configs[2] as worded): the volume network takes the 44-channel warped volume
    feature of a five-view CascadeMVSNet pass; the renderer behind it is the reference's (last three images, first three poses of the batch).  Golden: the reference's own module with
    that one constructor argument changed (tests/golden/make_golden.py: render_v5_fixture), five-view batch, injected draws."""
    from rc_mvsnet_amd import synthetic
    from rc_mvsnet_amd.render_consist_net import Rendering_Consistency_Net
    g = load_golden("render_v5")
    H, W, V = int(g["H"]), int(g["W"]), int(g["V"])
    assert V == 5
    a = _args(16)
    a.num_views = 5
    m = Rendering_Consistency_Net(a)
    assert m.MVSNet.cost_reg_2.conv0.conv.weight.shape[1] == 44
    m.load_state_dict(synthetic.render_state_dict(1, vol_src=4), strict=True)
    m = m.to(DEV).eval()
    batch = {k: gpu(v) for k, v in synthetic.render_batch(V, H, W, 0).items()}
    pix, eps, u = synthetic.render_randoms(H, W, 1024, 16, int(g["seed"]))
    with torch.no_grad():
        vol = m.MVSNet(gpu(g["vfw"]))
        rgb, feat, wts, dpred, alpha, _, rdepth, target = m(gpu(g["vfw"]), gpu(g["pseudo"]), batch, randoms=(gpu(pix), gpu(eps), gpu(u)))
    assert rel_err(vol.cpu()[:, :, ::8], g["volume"]) < 5e-5
    assert torch.equal(rdepth.cpu(), g["rays_depth"]) and rel_err(target.cpu(), g["target"]) < 1e-6
    interior = (pix[0] > 0) & (pix[0] < W - 1) & (pix[1] > 0) & (pix[1] < H - 1)
    other = [c for c in range(20) if c not in (11, 15, 19)]
    assert rel_err(feat.cpu()[::4][..., other], g["feat"][..., other]) < 2e-4
    for got, key in ((alpha, "alpha"), (wts, "weights"), (rgb, "rgb")):
        assert rel_err(got.cpu()[interior], g[key][interior]) < 5e-4, key
    assert float((dpred.cpu() - g["depth"]).abs()[interior].max()) / 500.0 < 5e-4
    with pytest.raises(NotImplementedError):
        b = _args(16)
        b.num_views = 3
        Rendering_Consistency_Net(b)


def test_render_forward_full_size_properties(hip):
    """BASELINE config-3 shape (V=4, 512x640, 1024 rays x 128 samples): runs, finite, invariants hold."""
    from rc_mvsnet_amd import synthetic
    m = _net(128)
    H, W, V = 512, 640, 4
    batch = {k: gpu(v) for k, v in synthetic.render_batch(V, H, W, 0).items()}
    gen = torch.Generator().manual_seed(0)
    vfw = gpu(0.5 * torch.randn(1, 41, 48, H // 4, W // 4, generator=gen))
    pseudo = gpu(500.0 + 300.0 * torch.rand(1, H, W, generator=gen))
    with torch.no_grad():
        rgb, feat, wts, dpred, alpha, _, rdepth, target = m(vfw, pseudo, batch)
    for t in (rgb, feat, wts, dpred, alpha):
        assert torch.isfinite(t).all()
    assert rgb.shape == (1024, 3) and feat.shape == (1024, 128, 20) and wts.shape == (1024, 128)
    assert float(wts.sum(1).max()) <= 1.0 + 1e-4
    assert float(alpha.min()) >= 0.0 and float(alpha.max()) <= 1.0
    assert float(rgb.min()) >= 0.0 and float(rgb.max()) <= 1.0 + 1e-4
