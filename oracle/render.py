"""Oracle: the rendering-consistency branch -- Gaussian-Uniform ray sampling, point-feature
gathers, NeRF MLP and volumetric compositing (Rendering_Consistency_Net.forward).

Test infrastructure (see oracle/__init__.py).  RNG contract: the random
draws are *inputs* -- ``pix`` (2,N) integer pixel (x, y), ``eps`` (N,S) ~ N(0,1) and ``u``
(N/2,S) ~ U[0,1) -- so that sampler parity is well defined; the golden generator captures
them by patching torch.randint / torch.normal / torch.rand inside the reference.
"""
import torch

from . import conv3d

IMAGENET_MEAN = (0.485, 0.456, 0.406)
IMAGENET_STD = (0.229, 0.224, 0.225)


# --------------------------------------------------------------------------------------
def unpreprocess(imgs):
    """models/render_consist_net.py:44-51: (x - (-m/s)) / (1/s), per channel; imgs (N,V,3,H,W)."""
    mean = torch.tensor([-m / s for m, s in zip(IMAGENET_MEAN, IMAGENET_STD)], device=imgs.device).reshape(1, 1, 3, 1, 1)
    std = torch.tensor([1 / s for s in IMAGENET_STD], device=imgs.device).reshape(1, 1, 3, 1, 1)
    return (imgs - mean) / std


def ndc_coordinate(w2c, K, pts, inv_scale, near, far):
    """models/render_utils.py:112-146 (pad=0, lindisp=False).  pts (...,3) world points."""
    shp = pts.shape
    p = pts.reshape(-1, 3)
    p = torch.matmul(p, w2c[:3, :3].t()) + w2c[:3, 3].reshape(1, 3)
    q = p @ K.t()
    xy = (q[:, :2] / q[:, 2:3] + 0.0) / inv_scale.reshape(1, 2)
    z = (q[:, 2] - near) / (far - near)
    return torch.cat((xy, z.unsqueeze(1)), dim=1).reshape(shp)


def gaussian_uniform_samples(rays_depth, near, far, eps, u):
    """models/render_utils.py:201-226.  rays_depth (N,), eps (N,S), u (N/2,S) -> z (N,S).

    rays [0, N/2): sorted  mu + sigma*eps  with sigma = min(abs(far-mu), abs(mu-near)) / 3;
    rays [N/2, N): stratified uniform  lower + (upper-lower)*u  on linspace(near, far, S).
    (The reference draws Gaussians for every ray and overwrites the second half, :225.)"""
    N, S = eps.shape
    half = N // 2
    sigma = torch.min(torch.abs(far - rays_depth), torch.abs(rays_depth - near)) / 3
    g = rays_depth.unsqueeze(1) + sigma.unsqueeze(1) * eps
    g, _ = torch.sort(g, dim=1)
    t = torch.linspace(0.0, 1.0, steps=S, device=eps.device).reshape(1, S)
    lin = near * (1.0 - t) + far * t
    mids = 0.5 * (lin[:, 1:] + lin[:, :-1])
    upper = torch.cat([mids, lin[:, -1:]], -1)
    lower = torch.cat([lin[:, :1], mids], -1)
    z = g.clone()
    z[half:] = lower + (upper - lower) * u
    return z


def build_rays(imgs, pseudo_depth, w2cs, c2ws, intrinsics, near_fars, pix, eps, u):
    """build_rays_norm (models/render_utils.py:149-243) + get_rays_mvs (:86-108), pad=0.

    imgs (1,V,3,H,W) un-normalised; pseudo_depth (H,W); w2cs/c2ws (V,4,4); intrinsics (V,3,3);
    near_fars (V,2); pix (2,N) int64 rows (x, y).  Returns a dict with the reference's
    outputs: rays_pts (N,S,3), rays_dir (N,3), target_s (N,3), rays_ndc (N,S,3),
    depth_candidates (N,S), rays_o (3,N), rays_depth (N,)."""
    _, V, _, H, W = imgs.shape
    xs = pix[0].float()
    ys = pix[1].float()
    K = intrinsics[0]
    c2w = c2ws[0]
    dirs = torch.stack([(xs - K[0, 2]) / K[0, 0], (ys - K[1, 2]) / K[1, 1], torch.ones_like(xs)], -1)
    rays_d = dirs @ c2w[:3, :3].t()
    rays_o = c2w[:3, -1]
    target = imgs[0, 0][:, pix[1], pix[0]].permute(1, 0)
    rays_depth = pseudo_depth[pix[1], pix[0]]
    near, far = near_fars[0, 0], near_fars[0, 1]
    z = gaussian_uniform_samples(rays_depth, near, far, eps, u)
    pts = rays_o.reshape(1, 1, 3) + z.unsqueeze(-1) * rays_d.unsqueeze(1)
    inv_scale = torch.tensor([W - 1, H - 1], dtype=torch.float32, device=pts.device)
    ndc = ndc_coordinate(w2cs[0], intrinsics[0], pts, inv_scale, near, far)
    N = xs.shape[0]
    return {"rays_pts": pts, "rays_dir": rays_d, "target_s": target, "rays_ndc": ndc,
            "depth_candidates": z, "rays_o": rays_o.reshape(3, 1).expand(3, N), "rays_depth": rays_depth}


# --------------------------------------------------------------------------------------
def _unnorm(g, size):
    return ((g + 1) / 2) * (size - 1)


def trilinear_gather_zeros(vol, ndc):
    """index_point_feature (models/render_utils.py:304-330): 5-D grid_sample, trilinear,
    zeros padding, align_corners=True at grid = ndc*2-1 (x->W, y->H, z->D).
    vol (1,C,D,H,W); ndc (N,S,3) -> (N,S,C)."""
    _, C, D, H, W = vol.shape
    g = ndc * 2 - 1.0
    ix, iy, iz = _unnorm(g[..., 0], W), _unnorm(g[..., 1], H), _unnorm(g[..., 2], D)
    x0, y0, z0 = ix.floor(), iy.floor(), iz.floor()
    flat = vol.reshape(C, -1)
    out = torch.zeros(*ndc.shape[:2], C, device=vol.device, dtype=vol.dtype)
    for dz in (0, 1):
        for dy in (0, 1):
            for dx in (0, 1):
                xx, yy, zz = x0 + dx, y0 + dy, z0 + dz
                wx = (ix - x0) if dx else ((x0 + 1) - ix)
                wy = (iy - y0) if dy else ((y0 + 1) - iy)
                wz = (iz - z0) if dz else ((z0 + 1) - iz)
                ok = (xx >= 0) & (xx <= W - 1) & (yy >= 0) & (yy <= H - 1) & (zz >= 0) & (zz <= D - 1)
                lin = (zz.clamp(0, D - 1).long() * H + yy.clamp(0, H - 1).long()) * W + xx.clamp(0, W - 1).long()
                val = flat[:, lin.reshape(-1)].t().reshape(*ndc.shape[:2], C)
                wgt = torch.where(ok, wx * wy * wz, torch.zeros_like(wx))
                out = out + val * wgt.unsqueeze(-1)
    return out


def bilinear_gather_border(img, grid):
    """F.grid_sample(bilinear, padding_mode='border', align_corners=True).
    img (3,H,W); grid (N,S,2) normalised -> (N,S,3)."""
    C, H, W = img.shape
    ix = _unnorm(grid[..., 0], W).clamp(0, W - 1)
    iy = _unnorm(grid[..., 1], H).clamp(0, H - 1)
    x0, y0 = ix.floor(), iy.floor()
    flat = img.reshape(C, -1)
    out = torch.zeros(*grid.shape[:2], C, device=img.device, dtype=img.dtype)
    for dy in (0, 1):
        for dx in (0, 1):
            xx, yy = x0 + dx, y0 + dy
            wx = (ix - x0) if dx else ((x0 + 1) - ix)
            wy = (iy - y0) if dy else ((y0 + 1) - iy)
            ok = (xx <= W - 1) & (yy <= H - 1)
            lin = yy.clamp(0, H - 1).long() * W + xx.clamp(0, W - 1).long()
            val = flat[:, lin.reshape(-1)].t().reshape(*grid.shape[:2], C)
            out = out + val * torch.where(ok, wx * wy, torch.zeros_like(wx)).unsqueeze(-1)
    return out


def point_features(volume, imgs3, w2cs, intrinsics, rays_pts, rays_ndc):
    """gen_pts_feats (models/renderer.py:154-166) + build_color_volume
    (models/render_utils.py:247-279, with_mask=True).  volume (1,8,D,h,w); imgs3 (1,3,3,H,W) =
    the caller's ``imgs[:, -3:]`` -- views 1..3 sampled with the poses of views 0..2, a quirk
    of the reference (render_consist_net.py:74 vs render_utils.py:260) reproduced as is.
    Returns (N,S,20): 8 volume channels then, per image, r, g, b, in-bounds mask."""
    _, Vc, _, H, W = imgs3.shape
    inv_scale = torch.tensor([W - 1, H - 1], dtype=torch.float32, device=rays_pts.device)
    feats = [trilinear_gather_zeros(volume, rays_ndc)]
    for i in range(Vc):
        pix = ndc_coordinate(w2cs[i], intrinsics[i], rays_pts, inv_scale, 2, 6)
        grid = pix[..., :2] * 2.0 - 1.0
        rgb = bilinear_gather_border(imgs3[0, i], grid)
        inb = ((grid > -1.0) & (grid < 1.0))
        mask = (inb[..., 0] & inb[..., 1]).float()
        feats += [rgb, mask.unsqueeze(-1)]
    return torch.cat(feats, dim=-1)


# --------------------------------------------------------------------------------------
def embed(x, multires=10):
    """Embedder.embed (models/render_models.py:45-49): [x, sin(x*2^j)..., cos(x*2^j)...], j<10,
    frequency-major then coordinate (63 values for 3-D x)."""
    freqs = 2.0 ** torch.linspace(0.0, multires - 1, steps=multires, device=x.device, dtype=x.dtype)
    scaled = (x.unsqueeze(-2) * freqs.reshape(*([1] * (x.dim() - 1)), -1, 1)).reshape(*x.shape[:-1], -1)
    return torch.cat((x, torch.sin(scaled), torch.cos(scaled)), dim=-1)


def nerf_mlp(pts63, feat20, dirs3, sd, prefix="network_fn.nerf"):
    """Renderer_ours.forward, use_viewdirs=True (models/render_models.py:192-220).
    pts63 (M,63), feat20 (M,20), dirs3 (M,3) -> (M,4) = [sigmoid rgb, relu sigma]."""
    def lin(x, name):
        return x @ sd[f"{prefix}.{name}.weight"].t() + sd[f"{prefix}.{name}.bias"]
    bias = lin(feat20, "pts_bias")
    h = pts63
    for i in range(6):
        h = torch.relu(lin(h, f"pts_linears.{i}") * bias)
        if i == 4:
            h = torch.cat([pts63, h], -1)
    alpha = torch.relu(lin(h, "alpha_linear"))
    feature = lin(h, "feature_linear")
    h = torch.relu(lin(torch.cat([feature, dirs3], -1), "views_linears.0"))
    rgb = torch.sigmoid(lin(h, "rgb_linear"))
    return torch.cat([rgb, alpha], -1)


def composite(raw, z):
    """raw2alpha + raw2outputs (models/renderer.py:18-26,65-93): alpha = 1-exp(-sigma)
    (``dists`` is computed by the caller but never used), T = exclusive cumprod of
    (1 - alpha + 1e-10), w = alpha*T.  raw (N,S,4), z (N,S)."""
    sigma = raw[..., 3]
    alpha = 1.0 - torch.exp(-sigma)
    trans = torch.cumprod(torch.cat([torch.ones(alpha.shape[0], 1, device=alpha.device, dtype=alpha.dtype), 1.0 - alpha + 1e-10], -1), -1)[:, :-1]
    w = alpha * trans
    rgb_map = torch.sum(w[..., None] * raw[..., :3], -2)
    depth_map = torch.sum(w * z, -1)
    return {"rgb_map": rgb_map, "depth_map": depth_map, "weights": w, "alpha": alpha, "acc_map": torch.sum(w, -1)}


def rendering(volume, imgs_unnorm, w2cs, intrinsics, rays, sd):
    """rendering (models/renderer.py:168-195) with net_type 'v0', use_color_volume False."""
    rays_dir = rays["rays_dir"]
    cos = torch.norm(rays_dir, dim=-1)
    angle = (rays_dir / cos.unsqueeze(-1)) @ w2cs[0][:3, :3].t()
    feat = point_features(volume, imgs_unnorm[:, -3:], w2cs, intrinsics, rays["rays_pts"], rays["rays_ndc"])
    N, S = feat.shape[:2]
    pts = embed(rays["rays_ndc"])
    dirs = angle[:, None].expand(-1, S, -1)
    raw = nerf_mlp(pts.reshape(N * S, -1), feat.reshape(N * S, -1), dirs.reshape(N * S, -1), sd).reshape(N, S, 4)
    out = composite(raw, rays["depth_candidates"])
    out["input_feat"] = feat
    out["raw"] = raw
    return out


def forward(volume_feature_warp, pseudo_depth, batch, sd, pix, eps, u):
    """Rendering_Consistency_Net.forward (models/render_consist_net.py:54-76).

    batch: dict with 'imgs' (1,V,3,H,W) normalised, 'w2cs','c2ws' (1,V,4,4), 'intrinsics'
    (1,V,3,3), 'near_fars' (1,V,2).  Returns the reference's 8-tuple
    (rgb, input_feat, weights, depth_pred, alpha, {}, rays_depth, target_s)."""
    volume = conv3d.neural_volume_net(volume_feature_warp, sd)
    imgs = unpreprocess(batch["imgs"].float())
    w2cs, c2ws = batch["w2cs"][0].float(), batch["c2ws"][0].float()
    intr, nf = batch["intrinsics"][0].float(), batch["near_fars"][0].float()
    rays = build_rays(imgs, pseudo_depth.reshape(pseudo_depth.shape[-2:]), w2cs, c2ws, intr, nf, pix, eps, u)
    r = rendering(volume, imgs, w2cs, intr, rays, sd)
    return r["rgb_map"], r["input_feat"], r["weights"], r["depth_map"], r["alpha"], {}, rays["rays_depth"], rays["target_s"]
