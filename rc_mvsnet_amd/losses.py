"""Self-supervised losses of the training step on the HIP path.

Same classes, argument meaning and attribute names as the reference's ``losses`` package, so ``train_rcmvsnet.py`` can
import them from here unchanged:

    UnSupLoss, UnsupLossMultiStage      losses/unsup_loss.py:9-94, 423-451
    AugLossMultiStage, random_image_mask  losses/aug_loss.py:8-67
    SL1Loss                              losses/sl1loss.py:4-13
    inverse_warping                      losses/homography.py:6-63   (channels-last image in, warped + mask out)

One stage of ``UnSupLoss`` is two library calls (``rcmvs_unsup_loss_fwd`` / ``_bwd``): the inverse warps of all source
views, the photometric / gradient / SSIM / smoothness sums, the per-pixel best view and the three scalars stay on the
device, and the backward produces d loss / d depth directly (the images carry no gradient -- asking for one raises).
There is no CPU or eager fallback: tensors must live on the GPU and the library must be built.
"""
import numpy as np
import torch
import torch.nn as nn

from . import _lib
from .ops import _chk, _stream

MAX_VIEWS = 8     # RCMVS_UNSUP_MAX_VIEWS


# ---------------------------------------------------------------------------------------------------------- geometry
def inverse_warp_coefs(ref_cam, src_cam):
    """{M row-major, t} fp32 with p = M (x,y,1)^T d + t (homography.py:9-56 composed in fp64; the projection keeps the
    REFERENCE view's intrinsics, as the reference does).  ref_cam (B,2,4,4) with src_cam (B,2,4,4) -> (B,12), or with
    src_cam (B,Vs,2,4,4) -> (Vs,B,12) for all source views in one batch of small matrix products."""
    many = src_cam.dim() == 5
    src = (src_cam if many else src_cam.unsqueeze(1)).double()
    ref = ref_cam.double().unsqueeze(1)
    R_l, t_l, K = ref[:, :, 0, :3, :3], ref[:, :, 0, :3, 3:4], ref[:, :, 1, :3, :3]
    R_rel = src[:, :, 0, :3, :3] @ R_l.transpose(-1, -2)
    t_rel = src[:, :, 0, :3, 3:4] - R_rel @ t_l
    M = K @ R_rel @ torch.linalg.inv(K)
    t = K @ t_rel
    coef = torch.cat([M.flatten(-2), t.flatten(-2)], -1).float()          # (B,Vs,12)
    return coef.transpose(0, 1).contiguous() if many else coef[:, 0].contiguous()


def _nearest_index(n_in, n_out, device):
    # ATen nearest: src = min(floor(dst * (in / out)), in - 1) with the scale held in fp32
    scale = np.float32(n_in) / np.float32(n_out)
    idx = np.minimum(np.floor(np.arange(n_out, dtype=np.float32) * scale).astype(np.int64), n_in - 1)
    return torch.from_numpy(idx).to(device)


def nearest_reduce(x, factor):
    """F.interpolate(x, scale_factor=1/factor, recompute_scale_factor=True) (nearest) on (B,C,H,W): index plumbing."""
    if factor == 1:
        return x
    H, W = x.shape[-2:]
    h, w = int(np.floor(H * (1.0 / factor))), int(np.floor(W * (1.0 / factor)))
    if H % factor == 0 and W % factor == 0:
        return x[..., ::factor, ::factor]
    return x.index_select(-2, _nearest_index(H, h, x.device)).index_select(-1, _nearest_index(W, w, x.device))


def stage_image(img, stage_idx):
    """(B,3,H,W) -> channels-last (B,h,w,3) at the stage's resolution (unsup_loss.py:27-33)."""
    return nearest_reduce(img, (4, 2, 1)[stage_idx]).permute(0, 2, 3, 1).contiguous()


def inverse_warping(img, left_cam, right_cam, depth):
    """losses/homography.py:6-63: img (B,h,w,3) of the right (source) camera resampled into the left (reference) view
    through depth (B,h,w) -> warped (B,h,w,3), mask (B,h,w,1).  Forward only (the fused stage loss owns the backward)."""
    B, H, W, C = img.shape
    if C != 3:
        raise _lib.RcmvsError(f"inverse_warping: 3-channel images only, got {C}")
    coef = inverse_warp_coefs(left_cam, right_cam)
    img, depth = img.contiguous(), depth.detach().contiguous()
    warped = torch.empty_like(img)
    mask = torch.empty((B, H, W), device=img.device, dtype=torch.float32)
    _lib.check(_lib.load().rcmvs_inverse_warp(_chk(img, "img"), _chk(depth, "depth"), _chk(coef, "coef"), _chk(warped, "warped"),
                                              _chk(mask, "mask"), B, H, W, _stream()), "inverse_warp")
    return warped, mask.unsqueeze(-1)


# ------------------------------------------------------------------------------------------------- one stage, fused
class UnsupStageLossFn(torch.autograd.Function):
    """depth (B,h,w), ref (B,h,w,3), srcs (Vs,B,h,w,3), coef (Vs,B,12) -> tensor(3) = reconstr, ssim, smooth."""

    @staticmethod
    def forward(ctx, depth, ref, srcs, coef):
        Vs, B, H, W, _ = srcs.shape
        if not 1 <= Vs <= MAX_VIEWS:
            raise _lib.RcmvsError(f"unsup loss: {Vs} source views (1..{MAX_VIEWS})")
        dev = depth.device
        depth = depth.contiguous()
        warped = torch.empty_like(srcs)
        masks = torch.empty((Vs, B, H, W), device=dev, dtype=torch.float32)
        sums = torch.empty(4 * Vs + 2, device=dev, dtype=torch.float64)
        counts = torch.empty(Vs, device=dev, dtype=torch.int32)
        out = torch.empty(4 + Vs, device=dev, dtype=torch.float32)
        _lib.check(_lib.load().rcmvs_unsup_loss_fwd(
            _chk(ref, "ref"), _chk(srcs, "srcs"), _chk(depth, "depth"), _chk(coef, "coef"), _chk(warped, "warped"),
            _chk(masks, "masks"), _chk(sums, "sums", torch.float64), _chk(counts, "counts", torch.int32), _chk(out, "out"),
            B, Vs, H, W, _stream()), "unsup_loss_fwd")
        ctx.save_for_backward(depth, ref, srcs, coef, warped, masks, counts)
        ctx.view_losses = out[4:]
        return out[:3].clone()

    @staticmethod
    def backward(ctx, gout):
        depth, ref, srcs, coef, warped, masks, counts = ctx.saved_tensors
        if ctx.needs_input_grad[1] or ctx.needs_input_grad[2]:
            raise _lib.RcmvsError("unsup loss: gradients w.r.t. the images are not provided (the reference never asks for them)")
        Vs, B, H, W, _ = srcs.shape
        dev = depth.device
        gout = gout.contiguous().float()
        ws = torch.empty((B, H - 2, W - 2, 9), device=dev, dtype=torch.float32)
        kbuf = torch.empty(4 * Vs + 2, device=dev, dtype=torch.float32)
        gdepth = torch.empty_like(depth)
        _lib.check(_lib.load().rcmvs_unsup_loss_bwd(
            _chk(ref, "ref"), _chk(srcs, "srcs"), _chk(depth, "depth"), _chk(coef, "coef"), _chk(warped, "warped"),
            _chk(masks, "masks"), _chk(counts, "counts", torch.int32), _chk(gout, "gout"), _chk(ws, "ws"), _chk(kbuf, "kbuf"),
            _chk(gdepth, "gdepth"), B, Vs, H, W, _stream()), "unsup_loss_bwd")
        return gdepth, None, None, None


class UnSupLoss(nn.Module):
    """losses/unsup_loss.py:9-94.  forward(imgs (B,V,3,H,W), cams (B,V,2,4,4) at the stage scale, depth (B,h,w), stage_idx)
    -> 12 reconstr + 6 ssim + 0.18 smooth; the three terms are left on the module as in the reference."""

    def forward(self, imgs, cams, depth, stage_idx):
        V = imgs.shape[1]
        assert cams.shape[1] == V, "Different number of images and projection matrices"
        ref = stage_image(imgs[:, 0], stage_idx)
        srcs = nearest_reduce(imgs[:, 1:], (4, 2, 1)[stage_idx]).permute(1, 0, 3, 4, 2).contiguous()      # (Vs,B,h,w,3)
        coef = inverse_warp_coefs(cams[:, 0], cams[:, 1:])
        terms = UnsupStageLossFn.apply(depth, ref, srcs, coef)
        self.reconstr_loss, self.ssim_loss, self.smooth_loss = terms[0], terms[1], terms[2]
        self.unsup_loss = 12 * self.reconstr_loss + 6 * self.ssim_loss + 0.18 * self.smooth_loss
        return self.unsup_loss


class UnsupLossMultiStage(nn.Module):
    """losses/unsup_loss.py:423-451: forward(inputs, imgs, cams, dlossw=...) -> (total, scalar_outputs)."""

    def __init__(self):
        super().__init__()
        self.unsup_loss = UnSupLoss()

    def forward(self, inputs, imgs, cams, **kwargs):
        weights = kwargs.get("dlossw", None)
        total = torch.tensor(0.0, dtype=torch.float32, device=imgs.device, requires_grad=False)
        scalars = {}
        for key in [k for k in inputs.keys() if "stage" in k]:
            idx = int(key.replace("stage", "")) - 1
            loss = self.unsup_loss(imgs, cams[key], inputs[key]["depth"], idx)
            total = total + (weights[idx] if weights is not None else 1.0) * loss
            scalars["depth_loss_stage{}".format(idx + 1)] = loss
            scalars["reconstr_loss_stage{}".format(idx + 1)] = self.unsup_loss.reconstr_loss
            scalars["ssim_loss_stage{}".format(idx + 1)] = self.unsup_loss.ssim_loss
            scalars["smooth_loss_stage{}".format(idx + 1)] = self.unsup_loss.smooth_loss
        return total, scalars


# ------------------------------------------------------------------------------------------------- masked smooth-L1
class MaskedSmoothL1Fn(torch.autograd.Function):
    """mean of smooth_l1(pred - target) over mask > 0.5 (F.smooth_l1_loss(pred[mask], target[mask]))."""

    @staticmethod
    def forward(ctx, pred, target, mask):
        pred, target, mask = pred.contiguous(), target.contiguous(), mask.contiguous()
        sums = torch.empty(2, device=pred.device, dtype=torch.float64)
        _lib.check(_lib.load().rcmvs_masked_sl1_fwd(_chk(pred, "pred"), _chk(target, "target"), _chk(mask, "mask"),
                                                    _chk(sums, "sums", torch.float64), pred.numel(), _stream()), "masked_sl1_fwd")
        ctx.save_for_backward(pred, target, mask, sums)
        return (sums[0] / sums[1]).float()

    @staticmethod
    def backward(ctx, g):
        pred, target, mask, sums = ctx.saved_tensors
        g = g.reshape(1).contiguous().float()
        gp = torch.empty_like(pred)
        _lib.check(_lib.load().rcmvs_masked_sl1_bwd(_chk(pred, "pred"), _chk(target, "target"), _chk(mask, "mask"),
                                                    _chk(sums, "sums", torch.float64), _chk(g, "g"), _chk(gp, "grad"),
                                                    pred.numel(), _stream()), "masked_sl1_bwd")
        return gp, None, None


def masked_smooth_l1(pred, target, mask):
    if pred.shape != target.shape or pred.shape != mask.shape:
        raise _lib.RcmvsError(f"masked_smooth_l1: shapes {tuple(pred.shape)} {tuple(target.shape)} {tuple(mask.shape)}")
    return MaskedSmoothL1Fn.apply(pred, target.detach(), mask.detach().float())


def random_image_mask(img, filter_size):
    """losses/aug_loss.py:8-28: blank a random (fh, fw) rectangle of img (B,3,H,W); returns (img, filter_mask)."""
    fh, fw = filter_size
    _, _, h, w = img.size()
    if fh == h and fw == w:
        return img, None
    x = np.random.randint(0, w - fw)
    y = np.random.randint(0, h - fh)
    filter_mask = torch.ones_like(img)
    filter_mask[:, :, y:y + fh, x:x + fw] = 0.0
    return img * filter_mask, filter_mask


class AugLossMultiStage(nn.Module):
    """losses/aug_loss.py:31-67: forward(inputs, pseudo_depth, mask_ms, filter_mask, dlossw=...) -> (total, scalars)."""

    def forward(self, inputs, pseudo_depth, mask_ms, filter_mask, **kwargs):
        weights = kwargs.get("dlossw", None)
        total = torch.tensor(0.0, dtype=torch.float32, device=pseudo_depth.device, requires_grad=False)
        scalars = {}
        for key in [k for k in inputs.keys() if "stage" in k]:
            idx = int(key.replace("stage", "")) - 1
            f = (4, 2, 1)[idx]
            gt = nearest_reduce(pseudo_depth.unsqueeze(1), f).squeeze(1)
            fm = nearest_reduce(filter_mask, f)[:, 0]
            loss = masked_smooth_l1(inputs[key]["depth"], gt, fm)
            total = total + (weights[idx] if weights is not None else 1.0) * loss
            scalars["aug_loss_stage{}".format(idx + 1)] = loss
        return total, scalars


class SL1Loss(nn.Module):
    """losses/sl1loss.py:4-13: 0.5 * smooth_l1(depth_pred[mask], depth_gt[mask]); mask defaults to depth_gt > 0."""

    def forward(self, depth_pred, depth_gt, mask=None):
        if mask is None:
            mask = depth_gt > 0
        return masked_smooth_l1(depth_pred, depth_gt, mask.float()) * 2 ** (1 - 2)
