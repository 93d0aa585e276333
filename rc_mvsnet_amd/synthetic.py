"""Seeded synthetic DTU-shaped inputs and non-degenerate weights (no dataset / checkpoint
is reachable here).  Shapes, camera model and depth range follow SURVEY.md section 8d:

  K(1/4 res) = [[361.54125*W/640, 0, 82.900625*W/640], [0, 360.3975*H/512, 66.383875*H/512], [0,0,1]]
  view v: rotation about y by 0.08*v rad, centre C = (60v, 10v, 0) mm, E = [R | -R C]
  proj[:, v, 0] = E, proj[:, v, 1, :3, :3] = K (x1, x2, x4 for stages 1..3, as
  datasets/dtu_test.py:215-224 builds them);  depth_values = 425 + 2.65*arange(192).

Everything is generated on the CPU with fixed seeds so that the golden generator, the
oracle, the tests and bench.py all see bit-identical tensors.
"""
import math

import numpy as np
import torch

NUM_DEPTH_VALUES = 192


def cameras(V, H, W):
    """Returns K_quarter (3,3) float64 and a list of V extrinsics (4,4) float64."""
    K = np.array([[361.54125 * W / 640.0, 0.0, 82.900625 * W / 640.0],
                  [0.0, 360.3975 * H / 512.0, 66.383875 * H / 512.0],
                  [0.0, 0.0, 1.0]])
    Es = []
    for v in range(V):
        a = 0.08 * v
        R = np.array([[math.cos(a), 0.0, math.sin(a)], [0.0, 1.0, 0.0], [-math.sin(a), 0.0, math.cos(a)]])
        C = np.array([60.0 * v, 10.0 * v, 0.0])
        E = np.eye(4)
        E[:3, :3] = R
        E[:3, 3] = -R @ C
        Es.append(E)
    return K, Es


def proj_matrices(B, V, H, W):
    """{'stage1','stage2','stage3'}: (B,V,2,4,4) float32."""
    K, Es = cameras(V, H, W)
    out = {}
    for s, mul in (("stage1", 1.0), ("stage2", 2.0), ("stage3", 4.0)):
        p = np.zeros((V, 2, 4, 4), dtype=np.float32)
        for v in range(V):
            p[v, 0] = Es[v]
            Ks = K.copy()
            Ks[:2] *= mul
            p[v, 1, :3, :3] = Ks
        out[s] = torch.from_numpy(np.broadcast_to(p, (B, V, 2, 4, 4)).copy())
    return out


def depth_values(B):
    return (425.0 + 2.65 * torch.arange(NUM_DEPTH_VALUES, dtype=torch.float32)).reshape(1, -1).repeat(B, 1)


def images(B, V, H, W, seed=0):
    """Smooth-ish seeded images, roughly ImageNet-normalised range: (B,V,3,H,W) float32."""
    g = torch.Generator().manual_seed(seed)
    out = torch.zeros(B * V, 3, H, W)
    for div, amp in ((16, 1.0), (4, 0.5), (1, 0.25)):
        n = torch.randn(B * V, 3, max(H // div, 1), max(W // div, 1), generator=g)
        out += amp * torch.nn.functional.interpolate(n, size=(H, W), mode="bilinear", align_corners=False)
    return out.reshape(B, V, 3, H, W).contiguous()


def cascade_inputs(B=1, V=3, H=512, W=640, seed=0):
    """(imgs, proj_matrices, depth_values) of CascadeMVSNet[_eval].forward."""
    return images(B, V, H, W, seed), proj_matrices(B, V, H, W), depth_values(B)


# ----------------------------------------------------------------------------------------
# weights with the reference's state_dict names
# ----------------------------------------------------------------------------------------
def _bn(sd, rng, name, c):
    sd[name + ".weight"] = torch.from_numpy(rng.uniform(0.5, 1.5, c).astype(np.float32))
    sd[name + ".bias"] = torch.from_numpy((0.1 * rng.standard_normal(c)).astype(np.float32))
    sd[name + ".running_mean"] = torch.from_numpy((0.1 * rng.standard_normal(c)).astype(np.float32))
    sd[name + ".running_var"] = torch.from_numpy(rng.uniform(0.5, 1.5, c).astype(np.float32))
    sd[name + ".num_batches_tracked"] = torch.tensor(0, dtype=torch.long)


def _w(rng, shape, fan_in, gain=1.0):
    bound = gain * math.sqrt(3.0 / fan_in)
    return torch.from_numpy(rng.uniform(-bound, bound, shape).astype(np.float32))


def cost_reg_specs(cin, base=8):
    """(name, kind, cin, cout) of CostRegNet (models/modules.py:470-489)."""
    b = base
    return [("conv0", "conv", cin, b), ("conv1", "conv", b, 2 * b), ("conv2", "conv", 2 * b, 2 * b),
            ("conv3", "conv", 2 * b, 4 * b), ("conv4", "conv", 4 * b, 4 * b), ("conv5", "conv", 4 * b, 8 * b),
            ("conv6", "conv", 8 * b, 8 * b), ("conv7", "deconv", 8 * b, 4 * b), ("conv9", "deconv", 4 * b, 2 * b),
            ("conv11", "deconv", 2 * b, b)]


def cascade_state_dict(seed=0, feat_channels=(32, 16, 8), prob_gain=20.0):
    """238 tensors named like CascadeMVSNet[_eval].state_dict() (fpn, 3 stages, cr base 8).
    BN statistics are randomised and prob.weight is scaled so that the probability volume
    is peaked instead of flat."""
    rng = np.random.RandomState(seed)
    sd = {}
    b = 8
    for name, ci, co, k in (("conv0.0", 3, b, 3), ("conv0.1", b, b, 3), ("conv1.0", b, 2 * b, 5), ("conv1.1", 2 * b, 2 * b, 3),
                            ("conv1.2", 2 * b, 2 * b, 3), ("conv2.0", 2 * b, 4 * b, 5), ("conv2.1", 4 * b, 4 * b, 3),
                            ("conv2.2", 4 * b, 4 * b, 3)):
        sd[f"feature.{name}.conv.weight"] = _w(rng, (co, ci, k, k), ci * k * k, math.sqrt(2.0))
        _bn(sd, rng, f"feature.{name}.bn", co)
    sd["feature.out1.weight"] = _w(rng, (4 * b, 4 * b, 1, 1), 4 * b)
    sd["feature.inner1.weight"] = _w(rng, (4 * b, 2 * b, 1, 1), 2 * b)
    sd["feature.inner1.bias"] = torch.from_numpy((0.1 * rng.standard_normal(4 * b)).astype(np.float32))
    sd["feature.inner2.weight"] = _w(rng, (4 * b, b, 1, 1), b)
    sd["feature.inner2.bias"] = torch.from_numpy((0.1 * rng.standard_normal(4 * b)).astype(np.float32))
    sd["feature.out2.weight"] = _w(rng, (2 * b, 4 * b, 3, 3), 4 * b * 9)
    sd["feature.out3.weight"] = _w(rng, (b, 4 * b, 3, 3), 4 * b * 9)
    for s, cin in enumerate(feat_channels):
        sd.update(cost_reg_state_dict(rng, f"cost_regularization.{s}", cin, prob_gain=prob_gain))
    return sd


def feature_unet_state_dict(seed=0, base=8, num_stage=3):
    """State dict of FeatureNet(base, num_stage, arch_mode='unet') (models/modules.py:363-401): the trunk of the 'fpn' form, the
    DeConv2dFuse merges and the 1x1 output convs; BN statistics randomised."""
    rng = np.random.RandomState(seed)
    sd, b = {}, base
    for name, ci, co, k in (("conv0.0", 3, b, 3), ("conv0.1", b, b, 3), ("conv1.0", b, 2 * b, 5), ("conv1.1", 2 * b, 2 * b, 3),
                            ("conv1.2", 2 * b, 2 * b, 3), ("conv2.0", 2 * b, 4 * b, 5), ("conv2.1", 4 * b, 4 * b, 3),
                            ("conv2.2", 4 * b, 4 * b, 3)):
        sd[f"{name}.conv.weight"] = _w(rng, (co, ci, k, k), ci * k * k, math.sqrt(2.0))
        _bn(sd, rng, f"{name}.bn", co)
    sd["out1.weight"] = _w(rng, (4 * b, 4 * b, 1, 1), 4 * b)
    for n, (name, ci, co) in enumerate((("deconv1", 4 * b, 2 * b), ("deconv2", 2 * b, b))[:num_stage - 1]):
        sd[f"{name}.deconv.conv.weight"] = _w(rng, (ci, co, 3, 3), ci * 9 / 4, math.sqrt(2.0))        # ConvTranspose2d: (in, out, k, k)
        _bn(sd, rng, f"{name}.deconv.bn", co)
        sd[f"{name}.conv.conv.weight"] = _w(rng, (co, 2 * co, 3, 3), 2 * co * 9, math.sqrt(2.0))
        _bn(sd, rng, f"{name}.conv.bn", co)
        sd[f"out{n + 2}.weight"] = _w(rng, (co, co, 1, 1), co)
    return sd


def cascade_unet_state_dict(seed=0, prob_gain=1.0):
    """State dict of CascadeMVSNet[_eval](arch_mode='unet') (3 stages, cr base 8): the 'unet' pyramid + the cost regularisations of
    cascade_state_dict (a smooth probability head by default)."""
    sd = {"feature." + k: v for k, v in feature_unet_state_dict(seed).items()}
    sd.update({k: v for k, v in cascade_state_dict(seed, prob_gain=prob_gain).items() if k.startswith("cost_regularization.")})
    return sd


def cost_reg_state_dict(rng, prefix, cin, base=8, prob_gain=20.0):
    sd = {}
    for name, kind, ci, co in cost_reg_specs(cin, base):
        shape = (co, ci, 3, 3, 3) if kind == "conv" else (ci, co, 3, 3, 3)
        sd[f"{prefix}.{name}.conv.weight"] = _w(rng, shape, ci * 27, math.sqrt(2.0))
        _bn(sd, rng, f"{prefix}.{name}.bn", co)
    sd[f"{prefix}.prob.weight"] = _w(rng, (1, base, 3, 3, 3), base * 27, prob_gain)
    return sd


def render_state_dict(seed=1, n_src=3, vol_src=None):
    """82 tensors named like Rendering_Consistency_Net.state_dict() (netdepth 6, width 128).  vol_src: source views behind the warped
    volume feature the volume network reads (32 + 3 vol_src input channels; default n_src = the reference's 3; 4 = the five-view
    extension of Neural_Volume_Net, models/render_models.py:750)."""
    rng = np.random.RandomState(seed)
    sd = {}
    p = "MVSNet.cost_reg_2"
    for name, kind, ci, co in cost_reg_specs(32 + 3 * (n_src if vol_src is None else vol_src), 8):
        if kind == "conv":
            sd[f"{p}.{name}.conv.weight"] = _w(rng, (co, ci, 3, 3, 3), ci * 27)
            _bn(sd, rng, f"{p}.{name}.bn", co)
        else:
            sd[f"{p}.{name}.0.weight"] = _w(rng, (ci, co, 3, 3, 3), ci * 27 / 8.0)
            _bn(sd, rng, f"{p}.{name}.1", co)
    q = "network_fn.nerf"

    def lin(name, ci, co, gain=math.sqrt(2.0)):
        sd[f"{q}.{name}.weight"] = _w(rng, (co, ci), ci, gain)
        sd[f"{q}.{name}.bias"] = torch.from_numpy((0.05 * rng.standard_normal(co)).astype(np.float32))
    lin("pts_linears.0", 63, 128)
    for i in range(1, 5):
        lin(f"pts_linears.{i}", 128, 128)
    lin("pts_linears.5", 191, 128)
    lin("pts_bias", 8 + 4 * n_src, 128, 1.0)
    lin("views_linears.0", 131, 64)
    lin("feature_linear", 128, 128, 1.0)
    lin("alpha_linear", 128, 1, 1.0)
    lin("rgb_linear", 64, 3, 1.0)
    # pts_bias centred on 1 so that the multiplicative bias does not kill the trunk
    sd[f"{q}.pts_bias.bias"] = sd[f"{q}.pts_bias.bias"] + 1.0
    return sd


# ----------------------------------------------------------------------------------------
# rendering-branch batch (keys of datasets/dtu_train.py:344-364 that the renderer reads)
# ----------------------------------------------------------------------------------------
def render_batch(V, H, W, seed=0):
    K, Es = cameras(V, H, W)
    Kfull = K.copy()
    Kfull[:2] *= 4.0
    w2cs = np.stack(Es).astype(np.float32)
    c2ws = np.stack([np.linalg.inv(E) for E in Es]).astype(np.float32)
    intr = np.stack([Kfull] * V).astype(np.float32)
    nf = np.stack([np.array([425.0, 425.0 + 2.65 * 191], dtype=np.float32)] * V)
    return {"imgs": images(1, V, H, W, seed), "w2cs": torch.from_numpy(w2cs)[None], "c2ws": torch.from_numpy(c2ws)[None],
            "intrinsics": torch.from_numpy(intr)[None], "near_fars": torch.from_numpy(nf)[None],
            "depths_h": torch.zeros(1, V, H, W), "proj_mats": torch.zeros(1, V, 3, 4)}


def render_randoms(H, W, n_rays=1024, n_samples=128, seed=0):
    """The injected random draws of the sampler: pix (2,N) int64 rows (x, y); eps; u."""
    g = torch.Generator().manual_seed(seed)
    xs = torch.randint(0, W, (n_rays,), generator=g)
    ys = torch.randint(0, H, (n_rays,), generator=g)
    eps = torch.randn(n_rays, n_samples, generator=g)
    u = torch.rand(n_rays // 2, n_samples, generator=g)
    return torch.stack((xs, ys)), eps, u


# ----------------------------------------------------------------------------------------
# a multi-view-consistent scan for the fusion filter
# ----------------------------------------------------------------------------------------
def _surface(X, Y):
    return 650.0 + 40.0 * np.sin(X / 80.0) * np.cos(Y / 60.0)


def fusion_scan(V=5, H=48, W=64, seed=0, n_src=4):
    """Depth maps of one smooth world surface seen from V cameras (so that they reproject onto each other), with seeded
    noise, a band of gross outliers per view, random confidences and 8-bit images.
    Returns dict: K (V,3,3) f32, E (V,4,4) f32, depth (V,H,W) f32, conf (V,H,W) f32, img (V,H,W,3) uint8, pairs."""
    Kq, Es = cameras(V, H, W)
    K = Kq.copy()
    K[:2] *= 4.0
    rng = np.random.default_rng(seed)
    ys, xs = np.meshgrid(np.arange(H, dtype=np.float64), np.arange(W, dtype=np.float64), indexing="ij")
    rays = np.linalg.inv(K) @ np.stack([xs.ravel(), ys.ravel(), np.ones(H * W)])
    depth = np.zeros((V, H, W), np.float32)
    for v in range(V):
        Ei = np.linalg.inv(Es[v])
        d = np.full(H * W, 650.0)
        for _ in range(25):                                   # fixed-point ray / surface intersection
            Pw = Ei[:3, :3] @ (rays * d) + Ei[:3, 3:4]
            d = d + (_surface(Pw[0], Pw[1]) - Pw[2])
        d = d + 0.25 * rng.standard_normal(H * W)
        d = d.reshape(H, W)
        d[:, (7 * v) % W:(7 * v) % W + 5] *= 1.06             # a band that fails the 1 % depth test
        depth[v] = d.astype(np.float32)
    conf = (0.55 + 0.45 * rng.random((V, H, W))).astype(np.float32)
    img = (255.0 * rng.random((V, H, W, 3))).astype(np.uint8)
    pairs = [(v, [(v + k) % V for k in range(1, n_src + 1)]) for v in range(V)]
    return {"K": np.broadcast_to(K.astype(np.float32), (V, 3, 3)).copy(), "E": np.stack(Es).astype(np.float32),
            "depth": depth, "conf": conf, "img": img, "pairs": pairs}


def write_fusion_scan(scan, pair_folder, out_folder, image_ext="png", depth_line="425.0 2.5"):
    """Lay the scan out the way the reference's filter_depth reads it (eval_rcmvsnet_dtu.py:341-368): pair.txt in
    pair_folder, cams/ + images/ + depth_est/ + confidence/ under out_folder (= its scan_folder)."""
    import os
    from PIL import Image
    from .data_io import save_pfm
    for sub in ("cams", "images", "depth_est", "confidence"):
        os.makedirs(os.path.join(out_folder, sub), exist_ok=True)
    os.makedirs(pair_folder, exist_ok=True)
    V = len(scan["depth"])
    with open(os.path.join(pair_folder, "pair.txt"), "w") as f:
        f.write("%d\n" % V)
        for ref, srcs in scan["pairs"]:
            f.write("%d\n%d %s\n" % (ref, len(srcs), " ".join("%d %.3f" % (s, 100.0 - i) for i, s in enumerate(srcs))))
    for v in range(V):
        with open(os.path.join(out_folder, "cams", "{:0>8}_cam.txt".format(v)), "w") as f:
            f.write("extrinsic\n")
            for row in scan["E"][v]:
                f.write(" ".join(repr(float(x)) for x in row) + "\n")
            f.write("\nintrinsic\n")
            for row in scan["K"][v]:
                f.write(" ".join(repr(float(x)) for x in row) + "\n")
            f.write("\n" + depth_line + "\n")
        # the reference opens '<view>.jpg'; PNG bytes under that name keep the fixture lossless (PIL sniffs the content)
        Image.fromarray(scan["img"][v]).save(os.path.join(out_folder, "images", "{:0>8}.jpg".format(v)), format=image_ext)
        save_pfm(os.path.join(out_folder, "depth_est", "{:0>8}.pfm".format(v)), scan["depth"][v])
        save_pfm(os.path.join(out_folder, "confidence", "{:0>8}.pfm".format(v)), scan["conf"][v])


def write_tanks_scan(scan, folder, depth_line="300.0 1100.0"):
    """The Tanks-and-Temples layout datasets/tanks.py reads: <folder>/{pair.txt, cams_1/<view:08d>_cam.txt, images/<view:08d>.jpg}
    with ``depth_min depth_max`` on the camera file's last line."""
    import os
    import shutil
    write_fusion_scan(scan, folder, folder, depth_line=depth_line)
    shutil.rmtree(os.path.join(folder, "cams_1"), ignore_errors=True)
    os.rename(os.path.join(folder, "cams"), os.path.join(folder, "cams_1"))


def tanks_fusion_scan(V=5, hw=(64, 96), orig_hw=(75, 100), seed=0, n_src=4):
    """A fusion scan in the Tanks-and-Temples situation (eval_rcmvsnet_tanks.py:269-330): depth maps at the network size
    ``hw`` while images and camera files describe the original size ``orig_hw`` (the filter rescales the intrinsics by
    img_wh / original and resizes the colour image)."""
    scan = fusion_scan(V=V, H=hw[0], W=hw[1], seed=seed, n_src=n_src)
    K = scan["K"].astype(np.float64)
    K[:, 0, :] *= orig_hw[1] / hw[1]
    K[:, 1, :] *= orig_hw[0] / hw[0]
    scan["K"] = K.astype(np.float32)
    rng = np.random.default_rng(seed + 100)
    scan["img"] = (255.0 * rng.random((V, orig_hw[0], orig_hw[1], 3))).astype(np.uint8)
    return scan


def write_tanks_fusion_scan(scan, scan_folder, out_folder):
    """scan_folder: pair.txt, cams_1/, images/ ; out_folder: depth_est/, confidence/ (eval_rcmvsnet_tanks.py:269-300)."""
    import os
    import shutil
    write_fusion_scan(scan, scan_folder, scan_folder)
    shutil.rmtree(os.path.join(scan_folder, "cams_1"), ignore_errors=True)
    os.rename(os.path.join(scan_folder, "cams"), os.path.join(scan_folder, "cams_1"))
    for sub in ("depth_est", "confidence"):
        os.makedirs(out_folder, exist_ok=True)
        shutil.rmtree(os.path.join(out_folder, sub), ignore_errors=True)
        shutil.move(os.path.join(scan_folder, sub), os.path.join(out_folder, sub))
