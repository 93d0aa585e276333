"""Oracle: plane-sweep homography warp, hypothesis planes and the variance cost volume.

Test infrastructure (see oracle/__init__.py).  All arithmetic fp32 unless stated; no fused
multiply-add is assumed anywhere in the coordinate chain (products and sums are separate
torch ops), which is what the HIP kernel reproduces with ``fp contract(off)``.
"""
import torch


# --------------------------------------------------------------------------------------
# projection matrices  (models/casmvsnet.py:267-270, models/modules.py:314-316)
# --------------------------------------------------------------------------------------
def fold_intrinsics(proj):
    """proj (B,2,4,4): [:,0] = extrinsic E (4x4), [:,1,:3,:3] = intrinsic K.
    Returns the 4x4 whose top 3x4 block is K @ E[:3,:4]  (casmvsnet.py:267-270)."""
    out = proj[:, 0].clone()
    out[:, :3, :4] = torch.matmul(proj[:, 1, :3, :3], proj[:, 0, :3, :4])
    return out


def compose_homography(src_proj, ref_proj):
    """rot (B,3,3), trans (B,3) of  src_proj_new @ inverse(ref_proj_new)  (modules.py:314-316)."""
    p = torch.matmul(fold_intrinsics(src_proj), torch.inverse(fold_intrinsics(ref_proj)))
    return p[:, :3, :3].contiguous(), p[:, :3, 3].contiguous()


# --------------------------------------------------------------------------------------
# homo_warping  (models/modules.py:304-339) -- spelled out tap by tap
# --------------------------------------------------------------------------------------
def warp_coords(rot, trans, depth, h, w):
    """Source-image sampling coordinates, in *pixels*, for every (plane, ref pixel).

    rot (B,3,3), trans (B,3), depth (B,D,h,w).  Returns ix, iy of shape (B,D,h,w).
    Follows modules.py:318-333 and then grid_sample's align_corners=True un-normalisation
    ix = ((gx + 1) / 2) * (w - 1)  (ATen GridSampler.h grid_sampler_unnormalize)."""
    B, D = depth.shape[:2]
    dev = depth.device
    y, x = torch.meshgrid(torch.arange(h, dtype=torch.float32, device=dev),
                          torch.arange(w, dtype=torch.float32, device=dev), indexing="ij")
    x = x.reshape(1, 1, h, w)
    y = y.reshape(1, 1, h, w)
    r = rot.reshape(B, 9, 1, 1, 1)
    t = trans.reshape(B, 3, 1, 1, 1)
    # rot @ (x, y, 1):  (r0*x + r1*y) + r2
    rx = (r[:, 0] * x + r[:, 1] * y) + r[:, 2]
    ry = (r[:, 3] * x + r[:, 4] * y) + r[:, 5]
    rz = (r[:, 6] * x + r[:, 7] * y) + r[:, 8]
    px = rx * depth + t[:, 0]
    py = ry * depth + t[:, 1]
    pz = rz * depth + t[:, 2]
    u = px / pz
    v = py / pz
    gx = u / ((w - 1) / 2) - 1
    gy = v / ((h - 1) / 2) - 1
    ix = ((gx + 1) / 2) * (w - 1)
    iy = ((gy + 1) / 2) * (h - 1)
    return ix, iy


def bilinear_gather_zeros(src, ix, iy):
    """F.grid_sample(bilinear, zeros padding, align_corners=True) given pixel coordinates.

    src (B,C,h,w); ix, iy (B,D,h,w) -> (B,C,D,h,w).  Tap weights as ATen computes them:
    nw=(x1-ix)*(y1-iy), ne=(ix-x0)*(y1-iy), sw=(x1-ix)*(iy-y0), se=(ix-x0)*(iy-y0), each tap
    contributing only when its integer location is inside the image; non-finite coordinates
    (pz == 0) contribute nothing (CUDA/HIP grid_sampler semantics)."""
    B, C, h, w = src.shape
    D = ix.shape[1]
    finite = torch.isfinite(ix) & torch.isfinite(iy)
    ixs = torch.where(finite, ix, torch.full_like(ix, -10.0))
    iys = torch.where(finite, iy, torch.full_like(iy, -10.0))
    x0 = torch.floor(ixs)
    y0 = torch.floor(iys)
    x1 = x0 + 1
    y1 = y0 + 1
    wx1 = ixs - x0          # weight of the east column
    wx0 = x1 - ixs          # weight of the west column
    wy1 = iys - y0
    wy0 = y1 - iys
    flat = src.reshape(B, C, h * w)
    out = None
    for (xx, yy, wgt) in ((x0, y0, wx0 * wy0), (x1, y0, wx1 * wy0), (x0, y1, wx0 * wy1), (x1, y1, wx1 * wy1)):
        ok = (xx >= 0) & (xx <= w - 1) & (yy >= 0) & (yy <= h - 1)
        xi = xx.clamp(0, w - 1).long()
        yi = yy.clamp(0, h - 1).long()
        lin = (yi * w + xi).reshape(B, 1, -1).expand(B, C, -1)
        val = torch.gather(flat, 2, lin).reshape(B, C, D, h, w)
        contrib = val * torch.where(ok, wgt, torch.zeros_like(wgt)).unsqueeze(1)
        out = contrib if out is None else out + contrib
    return out


def homo_warp(src_fea, src_proj, ref_proj, depth_values):
    """models/modules.py:304-339 with already intrinsics-folded 4x4 matrices (B,4,4)."""
    B, C, h, w = src_fea.shape
    if depth_values.dim() == 2:
        depth_values = depth_values.reshape(B, -1, 1, 1).expand(B, depth_values.shape[1], h, w)
    p = torch.matmul(src_proj, torch.inverse(ref_proj))
    ix, iy = warp_coords(p[:, :3, :3], p[:, :3, 3], depth_values, h, w)
    return bilinear_gather_zeros(src_fea, ix, iy)


# --------------------------------------------------------------------------------------
# hypothesis planes  (models/modules.py:549-588, models/casmvsnet.py:357-359,383-404)
# --------------------------------------------------------------------------------------
def _linear_resize_axis(x, out_size, axis):
    """1-D linear resize along `axis`, align_corners=False (ATen upsample_linear semantics:
    src = max(scale*(dst+0.5)-0.5, 0), i0=floor(src), i1=min(i0+1,in-1), lam=src-i0)."""
    in_size = x.shape[axis]
    if in_size == out_size:
        return x
    scale = torch.tensor(in_size / out_size, dtype=torch.float32)
    dst = torch.arange(out_size, dtype=torch.float32)
    srcf = (scale * (dst + 0.5) - 0.5).clamp(min=0)
    i0 = srcf.floor().long().clamp(max=in_size - 1)
    i1 = (i0 + 1).clamp(max=in_size - 1)
    lam1 = srcf - i0.float()
    lam0 = 1.0 - lam1
    shape = [1] * x.dim()
    shape[axis] = out_size
    a = x.index_select(axis, i0)
    b = x.index_select(axis, i1)
    return lam0.reshape(shape) * a + lam1.reshape(shape) * b


def resize_linear(x, size):
    """F.interpolate(mode='bilinear'/'trilinear', align_corners=False) on the trailing
    len(size) axes; W innermost first, like ATen (h0*(w0*a+w1*b) + h1*(...))."""
    nd = len(size)
    for k in range(nd):          # last axis first
        axis = x.dim() - 1 - k
        x = _linear_resize_axis(x, size[nd - 1 - k], axis)
    return x


def depth_interval_from_values(depth_values):
    """casmvsnet.py:357-359 -- python doubles: (max - min) / 192 (NOT 191), batch item 0."""
    dmin = float(depth_values[0, 0])
    dmax = float(depth_values[0, -1])
    return dmin, dmax, (dmax - dmin) / depth_values.shape[1]


def stage1_samples(depth_values, ndepth, h, w):
    """modules.py:574-582 followed by the (no-op) trilinear resize of casmvsnet.py:399-404."""
    dmin = depth_values[:, 0]
    dmax = depth_values[:, -1]
    itv = (dmax - dmin) / (ndepth - 1)
    k = torch.arange(ndepth, dtype=torch.float32).reshape(1, -1)
    s = dmin.unsqueeze(1) + k * itv.unsqueeze(1)
    return s.reshape(-1, ndepth, 1, 1).repeat(1, 1, h, w)


def cur_depth_samples(cur_depth_full, ndepth, interval_pixel):
    """modules.py:549-566.  cur_depth_full (B,H,W) -> (B,D,H,W)."""
    cmin = cur_depth_full - ndepth / 2 * interval_pixel
    cmax = cur_depth_full + ndepth / 2 * interval_pixel
    new_itv = (cmax - cmin) / (ndepth - 1)
    k = torch.arange(ndepth, dtype=torch.float32).reshape(1, -1, 1, 1)
    return cmin.unsqueeze(1) + k * new_itv.unsqueeze(1)


def stage_samples(prev_depth, depth_values, ndepth, ratio, full_hw, stage_hw):
    """Per-pixel hypothesis planes of one cascade stage (casmvsnet.py:371-404).

    prev_depth: None for stage 1, else the previous stage's (B,hp,wp) depth map."""
    H, W = full_hw
    h, w = stage_hw
    if prev_depth is None:
        return stage1_samples(depth_values, ndepth, h, w)
    _, _, itv = depth_interval_from_values(depth_values)
    cur = resize_linear(prev_depth, (H, W))                       # bilinear up, :383-385
    full = cur_depth_samples(cur, ndepth, ratio * itv)            # :388-396
    return resize_linear(full, (ndepth, h, w))                    # trilinear down, :399-404


# --------------------------------------------------------------------------------------
# variance cost volume  (models/casmvsnet.py:257-288 eval, :70-101 train)
# --------------------------------------------------------------------------------------
def variance_volume(features, proj_matrices, depth_samples):
    """features: list of V tensors (B,C,h,w), ref first.  proj_matrices (B,V,2,4,4).
    depth_samples (B,D,h,w).  Returns (B,C,D,h,w) = sum(x^2)/V - (sum(x)/V)^2."""
    V = len(features)
    ref = features[0]
    D = depth_samples.shape[1]
    vol_sum = ref.unsqueeze(2).repeat(1, 1, D, 1, 1)
    vol_sq = vol_sum ** 2
    ref_new = fold_intrinsics(proj_matrices[:, 0])
    for v in range(1, V):
        src_new = fold_intrinsics(proj_matrices[:, v])
        warped = homo_warp(features[v], src_new, ref_new, depth_samples)
        vol_sum = vol_sum + warped
        vol_sq = vol_sq + warped ** 2
    return vol_sq / V - (vol_sum / V) ** 2


def volume_feature_no_ref(features, imgs_stage, proj_matrices, depth_samples, training=True):
    """The train-variant extra output (casmvsnet.py:59,82,89-101): for each source view the
    warped stage-resolution RGB, then the variance over the *source* views only, divided by
    V (not V-1).  imgs_stage: (V,B,3,h,w) images already resized to the stage resolution.

    training=False reproduces the eval-mode quirk of the train module (casmvsnet.py:92-96):
    ``warped.pow_(2)`` runs in place before the no-ref accumulation, so the "sum" holds
    squares and the "square sum" holds 4th powers."""
    V = len(features)
    ref_new = fold_intrinsics(proj_matrices[:, 0])
    s = 0
    sq = 0
    rgb = []
    for v in range(1, V):
        src_new = fold_intrinsics(proj_matrices[:, v])
        rgb.append(homo_warp(imgs_stage[v], src_new, ref_new, depth_samples))
        warped = homo_warp(features[v], src_new, ref_new, depth_samples)
        if not training:
            warped = warped ** 2
        s = s + warped
        sq = sq + warped ** 2
    var = sq / V - (s / V) ** 2
    return torch.cat(rgb + [var], dim=1)
