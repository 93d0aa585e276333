"""Self-supervised loss without a GPU: the oracle against the reference's golden values,
the shared per-pixel arithmetic of csrc/unsup_loss_math.h (compiled with g++ into a loop harness) against the oracle,
and the host-side plumbing of rc_mvsnet_amd.losses."""
import ctypes
import os
import subprocess

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import unsup_loss as O
from rc_mvsnet_amd import _lib, losses, synthetic

HERE = os.path.dirname(os.path.abspath(__file__))
GOLD = np.load(os.path.join(HERE, "golden", "unsup_loss.npz"))
STAGES = ("stage1", "stage2", "stage3")
DLOSSW = [0.5, 1.0, 2.0]


def case(tag):
    B, V, H, W, seed = [int(x) for x in GOLD[tag + ":dims"]]
    return B, V, H, W, synthetic.images(B, V, H, W, seed), synthetic.proj_matrices(B, V, H, W)


@pytest.mark.parametrize("tag", ["a", "b"])
def test_oracle_matches_reference_golden(tag):
    B, V, H, W, imgs, cams = case(tag)
    inputs = {k: {"depth": torch.tensor(GOLD[f"{tag}:depth:{k}"]).requires_grad_(True)} for k in STAGES}
    total, scalars = O.unsup_loss_multi_stage(inputs, imgs, cams, dlossw=DLOSSW)
    total.backward()
    assert abs(float(total) - float(GOLD[tag + ":total"])) <= 1e-6 * abs(float(total))
    for k, v in scalars.items():
        assert abs(float(v) - float(GOLD[f"{tag}:{k}"])) <= 1e-6 * abs(float(v)), k
    for k in STAGES:
        g = torch.tensor(GOLD[f"{tag}:grad:{k}"])
        assert float((inputs[k]["depth"].grad - g).abs().max()) <= 1e-6 * float(g.abs().max()), k
    src = O.stage_image(imgs[:, 1], 1)
    warped, mask = O.inverse_warp(src, cams["stage2"][:, 0], cams["stage2"][:, 1], torch.tensor(GOLD[f"{tag}:depth:stage2"]))
    assert torch.equal(mask, torch.tensor(GOLD[tag + ":mask2"]))
    assert float((warped - torch.tensor(GOLD[tag + ":warped2"])).abs().max()) < 1e-6
    # augmentation-consistency loss
    inputs = {k: {"depth": torch.tensor(GOLD[f"{tag}:depth:{k}"]).requires_grad_(True)} for k in STAGES}
    fmask = torch.ones(B, 3, H, W)
    fmask[:, :, H // 4:H // 2, W // 8:W // 2] = 0.0
    atotal, ascal = O.aug_loss_multi_stage(inputs, torch.tensor(GOLD[tag + ":aug:pseudo"]), fmask, dlossw=DLOSSW)
    atotal.backward()
    assert abs(float(atotal) - float(GOLD[tag + ":aug:total"])) <= 1e-6 * abs(float(atotal))
    for k in STAGES:
        assert torch.allclose(inputs[k]["depth"].grad, torch.tensor(GOLD[f"{tag}:aug:grad:{k}"]), rtol=1e-6, atol=1e-12)


@pytest.fixture(scope="module")
def harness(tmp_path_factory):
    out = str(tmp_path_factory.mktemp("ul") / "ul_harness.so")
    subprocess.run(["g++", "-O2", "-w", "-ffp-contract=off", "-shared", "-fPIC", "-o", out,
                    os.path.join(HERE, "harness", "unsup_loss_harness.cpp")], check=True)
    return ctypes.CDLL(out)


def _p(t):
    return ctypes.c_void_p(t.data_ptr())


@pytest.mark.parametrize("tag", ["a", "b"])
def test_kernel_arithmetic_matches_oracle(tag, harness):
    """The per-pixel functions the HIP kernels call (unsup_loss_math.h), looped on the CPU, against the oracle's autograd."""
    B, V, H, W, imgs, cams = case(tag)
    Vs = V - 1
    for idx, key in enumerate(STAGES):
        depth = torch.tensor(GOLD[f"{tag}:depth:{key}"]).requires_grad_(True)
        r = O.unsup_loss(imgs, cams[key], depth, idx)
        gw = torch.tensor([12.0, 6.0, 0.18]) * DLOSSW[idx]
        (gw[0] * r["reconstr"] + gw[1] * r["ssim"] + gw[2] * r["smooth"]).backward()
        ref = losses.stage_image(imgs[:, 0], idx)
        srcs = torch.stack([losses.stage_image(imgs[:, v], idx) for v in range(1, V)]).contiguous()
        coef = torch.stack([losses.inverse_warp_coefs(cams[key][:, 0], cams[key][:, v]) for v in range(1, V)]).contiguous()
        h, w = ref.shape[1:3]
        d = depth.detach().contiguous()
        warped, masks = torch.empty_like(srcs), torch.empty(Vs, B, h, w)
        sums, counts, out = torch.empty(4 * Vs + 2, dtype=torch.float64), torch.empty(Vs, dtype=torch.int32), torch.empty(4 + Vs)
        harness.h_unsup_loss_fwd(_p(ref), _p(srcs), _p(d), _p(coef), _p(warped), _p(masks), _p(sums), _p(counts), _p(out), B, Vs, h, w)
        for name, i in (("reconstr", 0), ("ssim", 1), ("smooth", 2), ("loss", 3)):
            assert abs(float(out[i]) - float(r[name])) <= 2e-5 * abs(float(r[name])), (key, name, float(out[i]), float(r[name]))
        ow, om = O.inverse_warp(srcs[0], cams[key][:, 0], cams[key][:, 1], d)
        flips = float((masks[0] != om[..., 0]).float().mean())
        assert flips <= 2e-3, flips                                    # knife-edge floor() decisions only
        same = (masks[0] == om[..., 0]).unsqueeze(-1)
        assert float(((warped[0] - ow) * same).abs().max()) < 2e-3     # coordinates differ by ~1e-4 px (fp64 vs fp32 composition)
        gd = torch.empty_like(d)
        harness.h_unsup_loss_bwd(_p(ref), _p(srcs), _p(d), _p(coef), _p(warped), _p(masks), _p(counts), _p(gw.contiguous()), _p(gd), B, Vs, h, w)
        err = (gd - depth.grad).abs()
        scale = float(depth.grad.abs().max())
        assert float(err.median()) <= 1e-5 * scale, (key, float(err.median()), scale)
        assert float((err > 1e-3 * scale).float().mean()) <= 5e-3, (key, float((err > 1e-3 * scale).float().mean()))


def test_inverse_warp_coefs_match_the_reference_composition():
    cams = synthetic.proj_matrices(2, 3, 64, 80)["stage2"]
    coef = losses.inverse_warp_coefs(cams[:, 0], cams[:, 2]).double()
    proj, k_inv = O.relative_projection(cams[:, 0], cams[:, 2])
    M = proj[:, :3, :3].double() @ k_inv.double()
    assert torch.allclose(coef[:, :9].reshape(2, 3, 3), M, rtol=1e-4, atol=1e-6)
    assert torch.allclose(coef[:, 9:], proj[:, :3, 3].double(), rtol=1e-5, atol=1e-4)


@pytest.mark.parametrize("hw", [(64, 80), (50, 67), (33, 21)])
@pytest.mark.parametrize("factor", [4, 2, 1])
def test_nearest_reduce_matches_interpolate(hw, factor):
    x = torch.randn(2, 3, *hw)
    want = x if factor == 1 else F.interpolate(x, scale_factor=1.0 / factor, recompute_scale_factor=True)
    assert torch.equal(losses.nearest_reduce(x, factor), want)


def test_losses_fail_loudly_without_a_gpu():
    B, V, H, W, imgs, cams = case("a")
    with pytest.raises(_lib.RcmvsError):
        losses.UnSupLoss()(imgs, cams["stage3"], torch.tensor(GOLD["a:depth:stage3"]), 2)
    with pytest.raises(_lib.RcmvsError):
        losses.SL1Loss()(torch.rand(4, 5), torch.rand(4, 5))


def _harness_stage(harness, imgs, cams, depth, idx, gw):
    B, V = imgs.shape[:2]
    Vs = V - 1
    ref = losses.stage_image(imgs[:, 0], idx)
    srcs = losses.nearest_reduce(imgs[:, 1:], (4, 2, 1)[idx]).permute(1, 0, 3, 4, 2).contiguous()
    coef = losses.inverse_warp_coefs(cams[:, 0], cams[:, 1:]).contiguous()
    h, w = ref.shape[1:3]
    d = depth.detach().contiguous()
    warped, masks = torch.empty_like(srcs), torch.empty(Vs, B, h, w)
    sums, counts, out = torch.empty(4 * Vs + 2, dtype=torch.float64), torch.empty(Vs, dtype=torch.int32), torch.empty(4 + Vs)
    harness.h_unsup_loss_fwd(_p(ref), _p(srcs), _p(d), _p(coef), _p(warped), _p(masks), _p(sums), _p(counts), _p(out), B, Vs, h, w)
    gd = torch.empty_like(d)
    gw = gw.contiguous()
    harness.h_unsup_loss_bwd(_p(ref), _p(srcs), _p(d), _p(coef), _p(warped), _p(masks), _p(counts), _p(gw), _p(gd), B, Vs, h, w)
    return out, counts, masks, gd


@pytest.mark.parametrize("case_name", ["one_source_view", "a_view_that_sees_nothing", "three_by_three", "five_source_views"])
def test_kernel_arithmetic_edge_cases(case_name, harness):
    """Edge cases of UnSupLoss.forward through the kernels' arithmetic: a single source view (SSIM of one view only), a source
    camera that looks away (fully masked: it must win no pixel and contribute no gradient), the smallest image SSIM accepts,
    and more views than the SSIM term uses (only the first two source views enter it, unsup_loss.py:70-72)."""
    g = torch.Generator().manual_seed(7)
    if case_name == "three_by_three":
        B, V, H, W = 1, 3, 3, 3
    elif case_name == "five_source_views":
        B, V, H, W = 1, 6, 24, 32
    else:
        B, V, H, W = 2, (2 if case_name == "one_source_view" else 3), 24, 32
    imgs = synthetic.images(B, V, H, W, 3)
    cams = synthetic.proj_matrices(B, V, H, W)["stage3"].clone()
    if case_name == "a_view_that_sees_nothing":
        cams[:, 2, 0, :3, 3] += torch.tensor([5.0e4, 0.0, 0.0])          # translate source view 2 far off to the side
    depth = (600.0 + 60.0 * torch.rand(B, H, W, generator=g)).requires_grad_(True)
    r = O.unsup_loss(imgs, cams, depth, 2)
    gw = torch.tensor([12.0, 6.0, 0.18])
    r["loss"].backward()
    out, counts, masks, gd = _harness_stage(harness, imgs, cams, depth, 2, gw)
    for name, i in (("reconstr", 0), ("ssim", 1), ("smooth", 2), ("loss", 3)):
        want = float(r[name])
        assert abs(float(out[i]) - want) <= 2e-5 * max(abs(want), 1e-6), (case_name, name, float(out[i]), want)
    scale = float(depth.grad.abs().max())
    err = (gd - depth.grad).abs()
    assert float(err.median()) <= 1e-5 * scale and float((err > 1e-3 * scale).float().mean()) <= 1e-2, (case_name, float(err.max()), scale)
    if case_name == "a_view_that_sees_nothing":
        assert float(masks[1].sum()) == 0.0 and int(counts[1]) == 0
    assert int(counts.sum()) <= B * H * W
