"""CPU: the oracle (oracle/) against fixtures captured from the imported reference
(tests/golden/make_golden.py).  This is the pin that lets the GPU tests trust the oracle."""
import numpy as np
import pytest
import torch

from conftest import load_golden, rel_err
from oracle import warp, conv3d, depth_head, cascade, render
from rc_mvsnet_amd import synthetic

TOL = 2e-5


@pytest.mark.parametrize("name", ["warp_a1", "warp_a2", "warp_b1", "warp_b2", "warp_c1"])
def test_homo_warp(name):
    g = load_golden(name)
    out = warp.homo_warp(g["src"], warp.fold_intrinsics(g["src_proj"]), warp.fold_intrinsics(g["ref_proj"]), g["depth"])
    assert out.shape == g["out"].shape
    assert torch.isfinite(out).all()
    assert rel_err(out, g["out"]) < TOL


def test_warp_out_of_bounds_pixels_are_zero():
    """Planes whose projection leaves the source image (tiny / negative depths in fixture 'a')
    give exact zeros in the reference; the oracle must zero exactly the same voxels."""
    for name in ("warp_a1", "warp_a2"):
        g = load_golden(name)
        out = warp.homo_warp(g["src"], warp.fold_intrinsics(g["src_proj"]), warp.fold_intrinsics(g["ref_proj"]), g["depth"])
        zero_ref = (g["out"] == 0).all(dim=1)
        zero_orc = (out == 0).all(dim=1)
        assert int(zero_ref.sum()) > 0
        assert float((zero_ref != zero_orc).float().mean()) < 2e-3


@pytest.mark.parametrize("name,scale", [("planes_s2", 2), ("planes_s3", 1), ("planes_s2odd", 2)])
def test_hypothesis_planes(name, scale):
    g = load_golden(name)
    H, W = [int(v) for v in g["full_hw"]]
    dv = synthetic.depth_values(1)
    out = warp.stage_samples(g["prev"], dv, int(g["ndepth"]), int(g["ratio"]), (H, W), (H // scale, W // scale))
    assert float((out - g["out"]).abs().max()) < 2e-4            # mm; 1 ulp at 600 mm = 6e-5


def test_hypothesis_planes_stage1():
    g = load_golden("planes_s1")
    H, W = [int(v) for v in g["full_hw"]]
    out = warp.stage_samples(None, synthetic.depth_values(1), 48, 4, (H, W), (H // 4, W // 4))
    assert torch.equal(out, g["out"])


def test_conv_raw():
    g = load_golden("conv_raw")
    y = conv3d.conv3d(g["x"], g["w"], stride=2)
    assert rel_err(y, g["y"]) < TOL
    assert rel_err(conv3d.conv_transpose3d(g["y"], g["wt"]), g["yt"]) < TOL


def _cr_sd():
    return synthetic.cost_reg_state_dict(np.random.RandomState(3), "cr", 8)


def test_costreg_eval_layers():
    g = load_golden("costreg_eval")
    sd = _cr_sd()
    out = conv3d.cost_reg_net(g["x"], sd, "cr")
    assert rel_err(out, g["out"]) < 5e-5
    # first layers individually
    s, b = conv3d.bn_fold(sd, "cr.conv0.bn")
    c0 = torch.relu(conv3d.conv3d(g["x"], sd["cr.conv0.conv.weight"]) * s.view(1, -1, 1, 1, 1) + b.view(1, -1, 1, 1, 1))
    assert rel_err(c0, g["conv0"]) < TOL


def test_costreg_train_bn():
    g = load_golden("costreg_train")
    out = conv3d.cost_reg_net(g["x"], _cr_sd(), "cr", training=True)
    assert rel_err(out, g["out"]) < 2e-4


def test_depth_head():
    g = load_golden("depth_head")
    depth, conf, p = depth_head.depth_head(g["logits"], g["samples"])
    assert rel_err(p, g["prob"]) < 1e-6
    assert float((depth - g["depth"]).abs().max()) < 5e-4
    # confidence index flips only where sum p*k sits within rounding of an integer
    fidx = g["fidx"]
    safe = (fidx - fidx.round()).abs() > 1e-4
    assert float((conf - g["conf"]).abs()[safe].max()) < 1e-5


@pytest.mark.parametrize("name", ["cascade_c1", "cascade_small", "cascade_v5", "cascade_v7_d64", "cascade_v7_d64_smooth"])
@pytest.mark.parametrize("impl", ["spec", "aten"])
def test_cascade_eval(name, impl):
    g = load_golden(name)
    H, W, V = int(g["H"]), int(g["W"]), int(g["V"])
    nd, ra = [int(v) for v in g["ndepths"]], [int(v) for v in g["ratios"]]
    imgs, pm, dv = synthetic.cascade_inputs(1, V, H, W, 0)
    sd = synthetic.cascade_state_dict(0, prob_gain=float(g["prob_gain"])) if "prob_gain" in g else synthetic.cascade_state_dict(0)
    if len(nd) == 1:
        # 1-stage FeatureNet: only stage1 is produced/used
        pass
    out = cascade.forward_eval(imgs, pm, dv, sd, nd, ra, impl=impl)
    rng = float(dv[0, -1] - dv[0, 0])
    err = float((out["depth"] - g["depth"]).abs().mean()) / rng
    assert err < 1e-4, err
    cd = (out["photometric_confidence"] - g["conf"]).abs()
    assert float((cd > 1e-3).float().mean()) < 0.02


def test_train_extras():
    g = load_golden("train_extras")
    H, W, V = int(g["H"]), int(g["W"]), int(g["V"])
    imgs, pm, dv = synthetic.cascade_inputs(1, V, H, W, 0)
    sd = synthetic.cascade_state_dict(0)
    for training, key in ((True, "vf_train"), (False, "vf_eval")):
        vf = cascade.forward_train_extras(imgs, pm, dv, sd, int(g["ndepth"]), training)
        assert vf.shape == g[key].shape
        assert rel_err(vf, g[key]) < 5e-5, key


def test_nerf_mlp():
    g = load_golden("nerf_mlp")
    sd = synthetic.render_state_dict(1)
    x = g["x"].reshape(-1, 86)
    out = render.nerf_mlp(x[:, :63], x[:, 63:83], x[:, 83:], sd).reshape(64, 16, 4)
    assert rel_err(out, g["out"]) < 5e-5


def test_render_forward():
    g = load_golden("render")
    H, W, V = int(g["H"]), int(g["W"]), int(g["V"])
    sd = synthetic.render_state_dict(1)
    batch = synthetic.render_batch(V, H, W, 0)
    vol = conv3d.neural_volume_net(g["vfw"], sd)
    assert rel_err(vol[:, :, ::8], g["volume"]) < 5e-5
    rgb, feat, wts, dpred, alpha, _, rdepth, target = render.forward(g["vfw"], g["pseudo"], batch, sd, g["pix"], g["eps"], g["u"])
    assert torch.equal(rdepth, g["rays_depth"])
    assert rel_err(target, g["target"]) < 1e-6
    assert rel_err(feat[::4], g["feat"]) < 1e-4
    assert rel_err(alpha, g["alpha"]) < 2e-4
    assert rel_err(wts, g["weights"]) < 2e-4
    assert rel_err(rgb, g["rgb"]) < 2e-4
    assert float((dpred - g["depth"]).abs().max()) / 500.0 < 2e-4
