"""Oracle: the reference's op graph WITH autograd, evaluated on a product module's parameters.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  The product modules of ``rc_mvsnet_amd`` are parameter holders whose
forward runs hand-written HIP kernels (and raises when it cannot); this file evaluates the same networks through the stock
ATen composites the reference calls -- ``F.grid_sample``, the modules' own ``nn.Conv*`` / ``nn.BatchNorm*`` / ``nn.Linear``
children, ``F.softmax``, ``F.interpolate`` -- on whatever device and dtype the module lives on (fp32 on the GPU for speed,
fp64 on the CPU as ground truth).  It is the comparator of the gradient tests (tests/test_gpu_train.py), pinned itself to
gradients produced by the imported reference (tests/golden/train_grads.npz, tests/test_train_step_cpu.py).

Cited lines are relative to /root/reference.  Nothing under rc_mvsnet_amd/ imports this module.
"""
import torch
import torch.nn.functional as F

from . import render as orr

STAGE_SCALE = (4, 2, 1)                     # models/casmvsnet.py:140-152


# ------------------------------------------------------------------------------------------------ blocks
def unit(m, x):
    """One conv -> [norm] -> [relu] unit: the holders Conv2d / Conv3d / Deconv3d (models/modules.py:28-60,118-210), the
    renderer's ConvBnReLU3D (no ReLU, models/render_models.py:675-686) and its nn.Sequential(ConvTranspose3d, norm) pairs."""
    if isinstance(m, torch.nn.Sequential):
        return m[1](m[0](x))
    y = m.conv(x)
    if getattr(m, "bn", None) is not None:
        y = m.bn(y)
    return torch.relu(y) if getattr(m, "relu", False) else y


def chain(seq, x):
    for m in seq:
        x = unit(m, x)
    return x


def feature_pyramid(fnet, img):
    """FeatureNet.forward, arch_mode 'fpn' (models/modules.py:440-464): three-level trunk, 1x1 laterals added onto the
    nearest-neighbour up-sampled coarser level, one output conv per stage."""
    levels = [chain(fnet.conv0, img)]
    levels.append(chain(fnet.conv1, levels[0]))
    levels.append(chain(fnet.conv2, levels[1]))
    merged = levels[2]
    out = {"stage1": fnet.out1(merged)}
    laterals = [None, getattr(fnet, "inner1", None), getattr(fnet, "inner2", None)]
    heads = [None, getattr(fnet, "out2", None), getattr(fnet, "out3", None)]
    for s in range(1, fnet.num_stage):
        merged = F.interpolate(merged, scale_factor=2, mode="nearest") + laterals[s](levels[2 - s])
        out[f"stage{s + 1}"] = heads[s](merged)
    return out


def unet3d(cr, x, final=None):
    """The 3-D U-Net shared by CostRegNet (models/modules.py:484-501, units with ReLU, `prob` head) and the renderer's CostReg
    (models/render_models.py:720-734, units without ReLU, no head): three stride-2 levels, transposed convs back up, additive skips."""
    skips = [unit(cr.conv0, x)]
    skips.append(unit(cr.conv2, unit(cr.conv1, skips[0])))
    skips.append(unit(cr.conv4, unit(cr.conv3, skips[1])))
    t = unit(cr.conv6, unit(cr.conv5, skips[2]))
    for name, skip in (("conv7", skips[2]), ("conv9", skips[1]), ("conv11", skips[0])):
        t = skip + unit(getattr(cr, name), t)
    return final(t) if final is not None else t


# ------------------------------------------------------------------------------------------------ plane sweep
def _fold(proj):
    """(B,2,4,4) [extrinsic, intrinsic] -> 4x4 with K @ E[:3,:4] on top (models/casmvsnet.py:63-69)."""
    out = proj[:, 0].clone()
    out[:, :3, :4] = proj[:, 1, :3, :3] @ proj[:, 0, :3, :4]
    return out


def plane_sweep_warp(src, src_proj, ref_proj, samples):
    """homo_warping (models/modules.py:304-339): the source map resampled at the reference pixels' projections for every
    hypothesis plane.  src (B,C,h,w), samples (B,D,h,w) -> (B,C,D,h,w).  Coordinates carry no gradient (:313)."""
    B, C, h, w = src.shape
    D = samples.shape[1]
    grid = plane_sweep_grid(src_proj, ref_proj, samples, h, w)
    out = F.grid_sample(src, grid, mode="bilinear", padding_mode="zeros", align_corners=True)
    return out.reshape(B, C, D, h, w)


def plane_sweep_grid(src_proj, ref_proj, samples, h, w):
    """The normalised sampling grid of homo_warping (models/modules.py:313-331), (B, D*h, w, 2), in the dtype of its inputs."""
    B, D = samples.shape[:2]
    dt, dev = samples.dtype, samples.device
    with torch.no_grad():
        rel = src_proj @ torch.inverse(ref_proj)
        ys, xs = torch.meshgrid(torch.arange(h, dtype=dt, device=dev), torch.arange(w, dtype=dt, device=dev), indexing="ij")
        pix = torch.stack((xs.reshape(-1), ys.reshape(-1), torch.ones(h * w, dtype=dt, device=dev)))                   # (3, hw)
        ray = rel[:, :3, :3] @ pix                                                                                    # (B,3,hw)
        pts = ray.unsqueeze(2) * samples.reshape(B, 1, D, h * w) + rel[:, :3, 3].reshape(B, 3, 1, 1)
        uv = pts[:, :2] / pts[:, 2:3]
        return torch.stack((uv[:, 0] / ((w - 1) / 2) - 1, uv[:, 1] / ((h - 1) / 2) - 1), dim=-1).reshape(B, D * h, w, 2)


def stage_samples(prev_depth, depth_values, ndepth, ratio, full_hw, stage_hw):
    """Hypothesis planes of one stage (models/casmvsnet.py:181-222, modules.py:549-588): stage 1 spans [d_min, d_max] of the
    192-entry table; later stages put `ndepth` planes at ratio * (d_max - d_min) / 192 around the bilinearly up-sampled
    previous depth and resize the plane volume to the stage resolution."""
    B = depth_values.shape[0]
    k = torch.arange(ndepth, dtype=depth_values.dtype, device=depth_values.device)
    if prev_depth is None:
        lo, hi = depth_values[:, 0], depth_values[:, -1]
        planes = lo[:, None] + k[None] * ((hi - lo) / (ndepth - 1))[:, None]
        return planes.reshape(B, ndepth, 1, 1).expand(-1, -1, *stage_hw).contiguous()
    pixel = ratio * (depth_values[0, -1].double() - depth_values[0, 0].double()) / depth_values.shape[1]      # a python float in the reference
    cur = F.interpolate(prev_depth.unsqueeze(1), list(full_hw), mode="bilinear", align_corners=False).squeeze(1)
    lo = cur - ndepth / 2 * pixel
    hi = cur + ndepth / 2 * pixel
    planes = lo.unsqueeze(1) + k.reshape(1, -1, 1, 1) * ((hi - lo) / (ndepth - 1)).unsqueeze(1)
    return F.interpolate(planes.unsqueeze(1), [ndepth, *stage_hw], mode="trilinear", align_corners=False).squeeze(1)


def depth_stage(model, feats, proj, samples, cr, imgs):
    """DepthNet.forward (models/casmvsnet.py:49-124; eval twin :238-311): variance over the reference volume and the warped
    source volumes, cost regularisation, softmax, soft-argmin depth and the 4-plane confidence; the train variant also
    returns volume_feature_no_ref = [warped stage-resolution RGB of every source view, source-only variance / V]."""
    V, D = len(feats), samples.shape[1]
    B, C, h, w = feats[0].shape
    ref = _fold(proj[:, 0])
    total = feats[0].unsqueeze(2).expand(-1, -1, D, -1, -1)
    total_sq = total ** 2
    extra, s_src, q_src = [], 0, 0
    small = None
    if model.TRAIN_VARIANT:
        small = F.interpolate(imgs.reshape(B * V, *imgs.shape[2:]), (h, w), mode="bilinear", align_corners=False).reshape(B, V, -1, h, w)
    for v in range(1, V):
        src = _fold(proj[:, v])
        warped = plane_sweep_warp(feats[v], src, ref, samples)
        total = total + warped
        total_sq = total_sq + warped ** 2
        if model.TRAIN_VARIANT:
            extra.append(plane_sweep_warp(small[:, v], src, ref, samples))
            term = warped if model.training else warped ** 2        # eval mode squares in place first (models/casmvsnet.py:92-96)
            s_src = s_src + term
            q_src = q_src + term ** 2
    variance = total_sq / V - (total / V) ** 2
    logits = unet3d(cr, variance, cr.prob).squeeze(1)
    prob = F.softmax(logits, dim=1)
    depth = (prob * samples).sum(1)
    with torch.no_grad():                                           # models/casmvsnet.py:112-119
        padded = F.pad(prob, (0, 0, 0, 0, 1, 2))
        window = padded[:, 0:D] + padded[:, 1:D + 1] + padded[:, 2:D + 2] + padded[:, 3:D + 3]
        index = (prob * torch.arange(D, dtype=prob.dtype, device=prob.device).reshape(1, D, 1, 1)).sum(1).long().clamp(0, D - 1)
        conf = window.gather(1, index.unsqueeze(1)).squeeze(1)
    out = {"depth": depth, "photometric_confidence": conf}
    if model.TRAIN_VARIANT:
        out["volume_feature_no_ref"] = torch.cat(extra + [q_src / V - (s_src / V) ** 2], dim=1)
    return out


def cascade_forward(model, imgs, proj_matrices, depth_values):
    """CascadeMVSNet.forward / CascadeMVSNet_eval.forward (models/casmvsnet.py:171-231,356-417) for a product module:
    returns what the module's forward returns (the train variant: (outputs, stage-1 volume_feature_no_ref))."""
    B, V, _, H, W = imgs.shape
    feats = [feature_pyramid(model.feature, imgs[:, v]) for v in range(V)]
    outputs, depth = {}, None
    for s in range(model.num_stage):
        key = f"stage{s + 1}"
        sc = STAGE_SCALE[s]
        prev = None
        if depth is not None:
            prev = depth.detach() if model.grad_method == "detach" else depth
        samples = stage_samples(prev, depth_values, model.ndepths[s], model.depth_interals_ratio[s], (H, W), (H // sc, W // sc))
        cr = model.cost_regularization if model.share_cr else model.cost_regularization[s]
        out = depth_stage(model, [f[key] for f in feats], proj_matrices[key], samples, cr, imgs)
        depth = out["depth"]
        outputs[key] = out
        outputs.update(out)
    if model.TRAIN_VARIANT:
        return outputs, outputs["stage1"]["volume_feature_no_ref"]
    return outputs


# ------------------------------------------------------------------------------------------------ rendering branch
def neural_volume(nv, volume_feature):
    """Neural_Volume_Net.forward (models/render_models.py:753-760): depth axis to 128 planes (trilinear, align_corners), U-Net."""
    B, C, _, h, w = volume_feature.shape
    v = F.interpolate(volume_feature, size=[128, h, w], mode="trilinear", align_corners=True)
    v = unet3d(nv.cost_reg_2, v)
    return v.reshape(1, -1, *v.shape[2:])


def mlp(net, x):
    """Renderer_ours.forward with view directions (models/render_models.py:192-220) on the module's nn.Linear children:
    x = [63 encoded coordinates | 20 point features | 3 view direction]."""
    pts, feat, views = torch.split(x, [net.in_ch_pts, x.shape[-1] - net.in_ch_pts - net.in_ch_views, net.in_ch_views], dim=-1)
    gate = net.pts_bias(feat)
    h = pts
    for i, layer in enumerate(net.pts_linears):
        h = torch.relu(layer(h) * gate)
        if i in net.skips:
            h = torch.cat((pts, h), dim=-1)
    sigma = torch.relu(net.alpha_linear(h))
    h = torch.relu(net.views_linears[0](torch.cat((net.feature_linear(h), views), dim=-1)))
    return torch.cat((torch.sigmoid(net.rgb_linear(h)), sigma), dim=-1)


def render_forward(net, volume_feature_warp, pseudo_depth, batch, randoms):
    """Rendering_Consistency_Net.forward (models/render_consist_net.py:54-76) for a product module with injected random
    draws (pix (2,1024) int, eps (1024,S), u (512,S)): rays and samples, point features and compositing through the
    restatements of oracle/render.py (all differentiable torch ops), volume network and MLP through the module's children."""
    dev = volume_feature_warp.device
    pix, eps, u = randoms
    imgs = orr.unpreprocess(batch["imgs"].float().to(dev))
    w2cs, c2ws = batch["w2cs"].float().to(dev)[0], batch["c2ws"].float().to(dev)[0]
    intr, nf = batch["intrinsics"].float().to(dev)[0], batch["near_fars"].float().to(dev)[0]
    H, W = imgs.shape[-2:]
    volume = neural_volume(net.MVSNet, volume_feature_warp)
    rays = orr.build_rays(imgs, pseudo_depth.reshape(H, W).float(), w2cs, c2ws, intr, nf, pix, eps, u)
    feat = orr.point_features(volume, imgs[:, -3:], w2cs, intr, rays["rays_pts"], rays["rays_ndc"])
    N, S = feat.shape[:2]
    d = rays["rays_dir"]
    angle = (d / torch.norm(d, dim=-1, keepdim=True)) @ w2cs[0][:3, :3].t()
    x = torch.cat((orr.embed(rays["rays_ndc"]), feat, angle[:, None].expand(-1, S, -1)), dim=-1)
    raw = mlp(net.network_fn.nerf, x.reshape(N * S, -1)).reshape(N, S, 4)
    r = orr.composite(raw, rays["depth_candidates"])
    rgb = r["rgb_map"]
    if getattr(net, "white_bkgd", False):
        rgb = rgb + (1.0 - r["weights"].sum(-1, keepdim=True))
    return rgb, feat, r["weights"], r["depth_map"], r["alpha"], {}, rays["rays_depth"], rays["target_s"]
