"""Oracle: the self-supervised photometric loss of the training step.

Test infrastructure (see oracle/__init__.py): a CPU restatement of what the reference computes in
losses/homography.py:6-200 (inverse warp of a source image into the reference view through the estimated depth, with
its own bilinear sampler), losses/modules.py:6-82 (SSIM, image-aware depth smoothness, photometric + gradient
smooth-L1) and losses/unsup_loss.py:9-94,423-451 (per-pixel best source view, stage weights).  Pinned by
tests/golden/unsup_loss.npz, which tests/golden/make_golden.py --only-unsup-loss produced by importing the reference.

The reference's quirks are kept because they change numbers:
  * the projection into the SOURCE view uses the REFERENCE view's intrinsics (homography.py:53-56 builds
    ``intrinsic_mat_hom`` from ``K_left``);
  * the validity mask tests ``y0 <= max_y`` where ``y1 <= max_y`` was probably meant (homography.py:148);
  * the bilinear weights are taken against the CLAMPED corner indices (homography.py:151-154,187-190), so samples that
    fall outside the image are not zero but an extrapolation -- they are masked out of the photometric term but do enter
    the SSIM windows;
  * compute_reconstr_loss returns a scalar, so the "per-pixel minimum over views" (unsup_loss.py:66-88) picks, at each
    pixel, the valid view with the smallest *mean* loss.
"""
import torch
import torch.nn.functional as F

SSIM_C1 = 0.01 ** 2
SSIM_C2 = 0.03 ** 2


def pixel_grid(h, w):
    """(3, h*w) homogeneous pixel coordinates the way homography.py:66-83 builds them: linspace(-1, 1) rescaled to
    [0, n-1] in fp32 (so not exactly integer-valued)."""
    xs = (torch.linspace(-1.0, 1.0, w) + 1.0) * 0.5 * (w - 1)
    ys = (torch.linspace(-1.0, 1.0, h) + 1.0) * 0.5 * (h - 1)
    gx = xs.reshape(1, w).expand(h, w).reshape(-1)
    gy = ys.reshape(h, 1).expand(h, w).reshape(-1)
    return torch.stack([gx, gy, torch.ones_like(gx)], 0)


def relative_projection(ref_cam, src_cam):
    """(B,4,4) map from reference-camera coordinates to "source pixels" (homography.py:9-56).
    cam (B,2,4,4): [:,0,:3,:3] = R, [:,0,:3,3] = t, [:,1,:3,:3] = K."""
    R_l, R_r = ref_cam[:, 0, :3, :3], src_cam[:, 0, :3, :3]
    t_l, t_r = ref_cam[:, 0, :3, 3:4], src_cam[:, 0, :3, 3:4]
    K_l = ref_cam[:, 1, :3, :3]
    B = R_l.shape[0]
    R_rel = torch.matmul(R_r, R_l.transpose(1, 2))
    t_rel = t_r - torch.matmul(R_rel, t_l)
    bottom = torch.tensor([0.0, 0.0, 0.0, 1.0]).reshape(1, 1, 4).repeat(B, 1, 1)
    motion = torch.cat([torch.cat([R_rel, t_rel], 2).float(), bottom], 1)
    k_hom = torch.cat([torch.cat([K_l.float(), torch.zeros(B, 3, 1)], 2), bottom], 1)
    return torch.matmul(k_hom, motion), torch.inverse(K_l)


def source_coords(ref_cam, src_cam, depth):
    """Absolute sampling coordinates (B,h,w) x, y in the source image (homography.py:41-59,86-101)."""
    B, h, w = depth.shape
    proj, k_inv = relative_projection(ref_cam, src_cam)
    grid = pixel_grid(h, w).unsqueeze(0).repeat(B, 1, 1)
    cam = torch.matmul(k_inv.float(), grid.float()) * depth.reshape(B, 1, h * w).float()
    cam = torch.cat([cam, torch.ones(B, 1, h * w)], 1)
    p = torch.matmul(proj, cam)
    x = p[:, 0] / (p[:, 2] + 1e-10)
    y = p[:, 1] / (p[:, 2] + 1e-10)
    return x.reshape(B, h, w), y.reshape(B, h, w)


def bilinear_sample(img, x, y):
    """img (B,h,w,C), absolute x, y (B,h,w) -> sampled (B,h,w,C), mask (B,h,w,1)  (homography.py:104-200, including the
    normalise / un-normalise round trip of _spatial_transformer)."""
    B, h, w, C = img.shape
    x = (x / (w - 1) * 2.0 - 1.0).reshape(-1).float()
    y = (y / (h - 1) * 2.0 - 1.0).reshape(-1).float()
    x = (x + 1.0) * (w - 1.0) / 2.0
    y = (y + 1.0) * (h - 1.0) / 2.0
    x0 = torch.floor(x).int()
    y0 = torch.floor(y).int()
    x1, y1 = x0 + 1, y0 + 1
    mask = ((x0 >= 0) & (x1 <= w - 1) & (y0 >= 0) & (y0 <= h - 1)).float()
    x0, x1 = x0.clamp(0, w - 1), x1.clamp(0, w - 1)
    y0, y1 = y0.clamp(0, h - 1), y1.clamp(0, h - 1)
    base = (torch.arange(B) * (h * w)).reshape(B, 1).repeat(1, h * w).reshape(-1)
    flat = img.reshape(-1, C).float()
    pa = flat[base + y0.long() * w + x0.long()]
    pb = flat[base + y1.long() * w + x0.long()]
    pc = flat[base + y0.long() * w + x1.long()]
    pd = flat[base + y1.long() * w + x1.long()]
    fx = x1.float() - x
    fy = y1.float() - y
    wa, wb = (fx * fy).unsqueeze(1), (fx * (1.0 - fy)).unsqueeze(1)
    wc, wd = ((1.0 - fx) * fy).unsqueeze(1), ((1.0 - fx) * (1.0 - fy)).unsqueeze(1)
    out = wa * pa + wb * pb + wc * pc + wd * pd
    return out.reshape(B, h, w, C), mask.reshape(B, h, w, 1)


def inverse_warp(src_img, ref_cam, src_cam, depth):
    """losses/homography.py:6-63: src_img (B,h,w,C) resampled into the reference view through ``depth`` (B,h,w)."""
    x, y = source_coords(ref_cam, src_cam, depth)
    return bilinear_sample(src_img, x, y)


def reconstr_loss(warped, ref, mask):
    """compute_reconstr_loss(simple=False), losses/modules.py:70-81: a scalar."""
    a, b = warped * mask, ref * mask
    photo = F.smooth_l1_loss(a, b, reduction="mean")
    gx = F.smooth_l1_loss(a[:, :, 1:] - a[:, :, :-1], b[:, :, 1:] - b[:, :, :-1], reduction="mean")
    gy = F.smooth_l1_loss(a[:, 1:] - a[:, :-1], b[:, 1:] - b[:, :-1], reduction="mean")
    return 0.5 * photo + 0.5 * (gx + gy)


def ssim(x, y, mask):
    """losses/modules.py:6-42 on channels-last input: (B,h-2,w-2,C)."""
    x, y, mask = (t.permute(0, 3, 1, 2) for t in (x, y, mask))
    pool = lambda t: F.avg_pool2d(t, 3, 1)  # noqa: E731
    mu_x, mu_y = pool(x), pool(y)
    s_x = pool(x ** 2) - mu_x ** 2
    s_y = pool(y ** 2) - mu_y ** 2
    s_xy = pool(x * y) - mu_x * mu_y
    n = (2 * mu_x * mu_y + SSIM_C1) * (2 * s_xy + SSIM_C2)
    d = (mu_x ** 2 + mu_y ** 2 + SSIM_C1) * (s_x + s_y + SSIM_C2)
    out = pool(mask) * torch.clamp((1 - n / d) / 2, 0, 1)
    return out.permute(0, 2, 3, 1)


def depth_smoothness(depth, img):
    """losses/modules.py:56-67 with lambda_wt = 1: depth (B,h,w,1), img (B,h,w,3)."""
    ddx = depth[:, :, :-1] - depth[:, :, 1:]
    ddy = depth[:, :-1] - depth[:, 1:]
    wx = torch.exp(-(img[:, :, :-1] - img[:, :, 1:]).abs().mean(3, keepdim=True))
    wy = torch.exp(-(img[:, :-1] - img[:, 1:]).abs().mean(3, keepdim=True))
    return (ddx * wx).abs().mean() + (ddy * wy).abs().mean()


def stage_image(img, stage_idx):
    """Nearest-neighbour reduction to the stage resolution (unsup_loss.py:27-32), channels-last."""
    if stage_idx == 0:
        img = F.interpolate(img, scale_factor=0.25, recompute_scale_factor=True)
    elif stage_idx == 1:
        img = F.interpolate(img, scale_factor=0.5, recompute_scale_factor=True)
    return img.permute(0, 2, 3, 1)


def unsup_loss(imgs, cams, depth, stage_idx):
    """UnSupLoss.forward (losses/unsup_loss.py:14-94).  imgs (B,V,3,H,W), cams (B,V,2,4,4) at the stage's scale,
    depth (B,h,w).  Returns dict(loss, reconstr, ssim, smooth)."""
    V = imgs.shape[1]
    ref = stage_image(imgs[:, 0], stage_idx)
    ssim_loss = 0
    per_view = []
    for v in range(1, V):
        src = stage_image(imgs[:, v], stage_idx)
        warped, mask = inverse_warp(src, cams[:, 0], cams[:, v], depth)
        per_view.append(reconstr_loss(warped, ref, mask) + 1e4 * (1 - mask))
        if v < 3:
            ssim_loss = ssim_loss + ssim(ref, warped, mask).mean()
    smooth = depth_smoothness(depth.unsqueeze(-1), ref)
    vol = torch.stack(per_view).permute(1, 2, 3, 4, 0)
    best = -torch.topk(-vol, k=1, sorted=False)[0]
    best = best * (best < 1e4).float()
    reconstr = best.sum(-1).mean()
    return {"loss": 12 * reconstr + 6 * ssim_loss + 0.18 * smooth, "reconstr": reconstr, "ssim": ssim_loss, "smooth": smooth}


def unsup_loss_multi_stage(inputs, imgs, cams, dlossw=None):
    """UnsupLossMultiStage.forward (losses/unsup_loss.py:423-451)."""
    total = torch.zeros((), dtype=torch.float32)
    scalars = {}
    for key in [k for k in inputs.keys() if "stage" in k]:
        idx = int(key.replace("stage", "")) - 1
        r = unsup_loss(imgs, cams[key], inputs[key]["depth"], idx)
        total = total + (dlossw[idx] if dlossw is not None else 1.0) * r["loss"]
        scalars["depth_loss_stage%d" % (idx + 1)] = r["loss"]
        scalars["reconstr_loss_stage%d" % (idx + 1)] = r["reconstr"]
        scalars["ssim_loss_stage%d" % (idx + 1)] = r["ssim"]
        scalars["smooth_loss_stage%d" % (idx + 1)] = r["smooth"]
    return total, scalars


def aug_loss_multi_stage(inputs, pseudo_depth, filter_mask, dlossw=None):
    """AugLossMultiStage.forward (losses/aug_loss.py:31-67): masked smooth-L1 between each stage's depth and the
    nearest-reduced pseudo depth of the un-augmented pass."""
    total = torch.zeros((), dtype=torch.float32)
    scalars = {}
    for key in [k for k in inputs.keys() if "stage" in k]:
        idx = int(key.replace("stage", "")) - 1
        gt = pseudo_depth.unsqueeze(1)
        fm = filter_mask
        if idx < 2:
            s = 0.25 if idx == 0 else 0.5
            gt = F.interpolate(gt, scale_factor=(s, s), recompute_scale_factor=True)
            fm = F.interpolate(fm, scale_factor=(s, s), recompute_scale_factor=True)
        m = fm[:, 0] > 0.5
        loss = F.smooth_l1_loss(inputs[key]["depth"][m], gt.squeeze(1)[m], reduction="mean")
        total = total + (dlossw[idx] if dlossw is not None else 1.0) * loss
        scalars["aug_loss_stage%d" % (idx + 1)] = loss
    return total, scalars
