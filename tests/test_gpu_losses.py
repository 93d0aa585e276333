"""GPU: the self-supervised losses on the HIP path (rc_mvsnet_amd/losses.py over rcmvs_unsup_loss_fwd/_bwd,
rcmvs_inverse_warp, rcmvs_masked_sl1_*) against the reference's golden values (tests/golden/unsup_loss.npz, produced by
importing losses/unsup_loss.py, losses/aug_loss.py, losses/homography.py) and against the oracle's autograd."""
import os

import numpy as np
import pytest
import torch

from oracle import unsup_loss as O
from rc_mvsnet_amd import _lib, losses, synthetic

pytestmark = pytest.mark.gpu

HERE = os.path.dirname(os.path.abspath(__file__))
GOLD = np.load(os.path.join(HERE, "golden", "unsup_loss.npz"))
STAGES = ("stage1", "stage2", "stage3")
DLOSSW = [0.5, 1.0, 2.0]
DEV = "cuda:0"


def case(tag):
    B, V, H, W, seed = [int(x) for x in GOLD[tag + ":dims"]]
    return B, V, H, W, synthetic.images(B, V, H, W, seed), synthetic.proj_matrices(B, V, H, W)


def grad_check(got, want, med=1e-5, frac=5e-3):
    err = (got.cpu() - want).abs()
    scale = float(want.abs().max())
    assert float(err.median()) <= med * scale, (float(err.median()), scale)
    assert float((err > 1e-3 * scale).float().mean()) <= frac            # knife-edge floor() decisions only


@pytest.mark.parametrize("tag", ["a", "b"])
def test_unsup_loss_multi_stage_matches_reference(tag):
    _lib.load()
    B, V, H, W, imgs, cams = case(tag)
    inputs = {k: {"depth": torch.tensor(GOLD[f"{tag}:depth:{k}"]).to(DEV).requires_grad_(True)} for k in STAGES}
    total, scalars = losses.UnsupLossMultiStage()(inputs, imgs.to(DEV), {k: v.to(DEV) for k, v in cams.items()}, dlossw=DLOSSW)
    total.backward()
    if torch.cuda.is_available():
        torch.cuda.synchronize()
    assert abs(float(total) - float(GOLD[tag + ":total"])) <= 2e-5 * abs(float(GOLD[tag + ":total"]))
    for k, v in scalars.items():
        want = float(GOLD[f"{tag}:{k}"])
        assert abs(float(v) - want) <= 2e-5 * abs(want), (k, float(v), want)
    for k in STAGES:
        grad_check(inputs[k]["depth"].grad, torch.tensor(GOLD[f"{tag}:grad:{k}"]))


@pytest.mark.parametrize("tag", ["a", "b"])
def test_inverse_warping_matches_reference(tag):
    _lib.load()
    B, V, H, W, imgs, cams = case(tag)
    src = losses.stage_image(imgs[:, 1].to(DEV), 1)
    cam = cams["stage2"].to(DEV)
    warped, mask = losses.inverse_warping(src, cam[:, 0], cam[:, 1], torch.tensor(GOLD[f"{tag}:depth:stage2"]).to(DEV))
    want_w, want_m = torch.tensor(GOLD[tag + ":warped2"]), torch.tensor(GOLD[tag + ":mask2"])
    assert mask.shape == want_m.shape and warped.shape == want_w.shape
    same = mask.cpu() == want_m
    assert float((~same).float().mean()) <= 2e-3
    assert float(((warped.cpu() - want_w) * same).abs().max()) < 2e-3


@pytest.mark.parametrize("tag", ["a", "b"])
def test_aug_loss_and_sl1_match_reference(tag):
    _lib.load()
    B, V, H, W, imgs, cams = case(tag)
    inputs = {k: {"depth": torch.tensor(GOLD[f"{tag}:depth:{k}"]).to(DEV).requires_grad_(True)} for k in STAGES}
    fmask = torch.ones(B, 3, H, W)
    fmask[:, :, H // 4:H // 2, W // 8:W // 2] = 0.0
    total, scalars = losses.AugLossMultiStage()(inputs, torch.tensor(GOLD[tag + ":aug:pseudo"]).to(DEV), None, fmask.to(DEV), dlossw=DLOSSW)
    total.backward()
    assert abs(float(total) - float(GOLD[tag + ":aug:total"])) <= 1e-5 * abs(float(GOLD[tag + ":aug:total"]))
    for k, v in scalars.items():
        assert abs(float(v) - float(GOLD[f"{tag}:aug:{k}"])) <= 1e-5 * abs(float(GOLD[f"{tag}:aug:{k}"])), k
    for k in STAGES:
        assert torch.allclose(inputs[k]["depth"].grad.cpu(), torch.tensor(GOLD[f"{tag}:aug:grad:{k}"]), rtol=1e-4, atol=1e-9), k
    # SL1Loss (losses/sl1loss.py): default mask depth_gt > 0, factor 1/2
    g = torch.Generator().manual_seed(3)
    pred = (3.0 * torch.randn(1024, generator=g)).requires_grad_(True)
    gt = 3.0 * torch.randn(1024, generator=g)
    want = torch.nn.functional.smooth_l1_loss(pred[gt > 0], gt[gt > 0]) * 0.5
    want.backward()
    p2 = pred.detach().to(DEV).requires_grad_(True)
    got = losses.SL1Loss()(p2, gt.to(DEV))
    got.backward()
    assert abs(float(got) - float(want)) <= 1e-6 * abs(float(want))
    assert torch.allclose(p2.grad.cpu(), pred.grad, rtol=1e-5, atol=1e-9)


def test_unsup_loss_full_size_against_oracle():
    """BASELINE config 3 shape (4 views, 512x640, batch 1), stage 3: the three terms and the depth gradient against the
    oracle's autograd on the CPU."""
    _lib.load()
    B, V, H, W = 1, 4, 512, 640
    imgs, cams = synthetic.images(B, V, H, W, 5), synthetic.proj_matrices(B, V, H, W)["stage3"]
    yy, xx = torch.meshgrid(torch.linspace(0, 1, H), torch.linspace(0, 1, W), indexing="ij")
    depth = (620.0 + 110.0 * torch.sin(4.0 * xx) * torch.cos(3.0 * yy) + torch.randn(H, W, generator=torch.Generator().manual_seed(6))).unsqueeze(0)
    d_cpu = depth.clone().requires_grad_(True)
    r = O.unsup_loss(imgs, cams, d_cpu, 2)
    r["loss"].backward()
    d_gpu = depth.to(DEV).requires_grad_(True)
    mod = losses.UnSupLoss()
    loss = mod(imgs.to(DEV), cams.to(DEV), d_gpu, 2)
    loss.backward()
    for name, got in (("reconstr", mod.reconstr_loss), ("ssim", mod.ssim_loss), ("smooth", mod.smooth_loss), ("loss", loss)):
        assert abs(float(got) - float(r[name])) <= 2e-5 * abs(float(r[name])), (name, float(got), float(r[name]))
    grad_check(d_gpu.grad, d_cpu.grad)


def test_unsup_loss_argument_checks():
    _lib.load()
    imgs, cams = synthetic.images(1, 3, 32, 40, 0).to(DEV), synthetic.proj_matrices(1, 3, 32, 40)["stage3"].to(DEV)
    depth = torch.full((1, 32, 40), 600.0, device=DEV)
    with pytest.raises(_lib.RcmvsError):
        losses.UnSupLoss()(imgs, cams, depth.double(), 2)                        # fp32 only
    with pytest.raises(_lib.RcmvsError):
        losses.UnSupLoss()(imgs[:, :, :, :2], cams, depth[:, :2], 2)             # SSIM needs 3 rows
    with pytest.raises(_lib.RcmvsError):
        losses.UnSupLoss()(imgs.repeat(1, 4, 1, 1, 1)[:, :10], cams.repeat(1, 4, 1, 1, 1)[:, :10], depth, 2)   # 9 source views
    # a loss that does not depend on depth leaves a zero gradient, not garbage
    d = depth.clone().requires_grad_(True)
    m = losses.UnSupLoss()
    m(imgs, cams, d, 2)
    (0.0 * m.reconstr_loss + 0.0 * m.ssim_loss + 0.0 * m.smooth_loss).backward()
    assert float(d.grad.abs().max()) == 0.0
