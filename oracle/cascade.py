"""Oracle: CascadeMVSNet_eval.forward / CascadeMVSNet.forward end to end on the CPU.

Test infrastructure (see oracle/__init__.py).  Two interchangeable op sets:

  impl="spec"  the index-by-index restatements of oracle/warp.py, conv3d.py, depth_head.py
               (what the HIP kernels are specified against);
  impl="aten"  the same graph through the ATen composites the reference itself calls
               (F.grid_sample, F.conv3d, F.conv_transpose3d, F.softmax, F.interpolate) --
               used as the timed ``cpu_baseline`` in bench.py because it is the op graph of
               models/casmvsnet.py:356-417 and runs at the reference's CPU speed; the two
               op sets are pinned against each other in tests/test_oracle_golden.py.
"""
import torch
import torch.nn.functional as F

from . import warp, conv3d, depth_head
from .feature_net import feature_net

STAGE_SCALE = {1: 4, 2: 2, 3: 1}            # casmvsnet.py:329-339


# ----------------------------------------------------------------------------- aten op set
def _aten_warp(src, src_proj, ref_proj, depth):
    B, C, h, w = src.shape
    D = depth.shape[1]
    p = torch.matmul(src_proj, torch.inverse(ref_proj))
    ix, iy = warp.warp_coords(p[:, :3, :3], p[:, :3, 3], depth, h, w)
    gx = ix / ((w - 1) / 2) - 1                     # back to normalised coords for grid_sample
    gy = iy / ((h - 1) / 2) - 1
    grid = torch.stack((gx, gy), dim=-1).reshape(B, D * h, w, 2)
    out = F.grid_sample(src, grid, mode="bilinear", padding_mode="zeros", align_corners=True)
    return out.reshape(B, C, D, h, w)


def _aten_costreg(x, sd, prefix):
    def bn(t, name):
        return F.batch_norm(t, sd[name + ".running_mean"], sd[name + ".running_var"], sd[name + ".weight"],
                            sd[name + ".bias"], False, 0.1, conv3d.BN_EPS)

    def block(t, name, stride=1):
        return torch.relu(bn(F.conv3d(t, sd[f"{prefix}.{name}.conv.weight"], None, stride, 1), f"{prefix}.{name}.bn"))

    def up(t, name):
        y = F.conv_transpose3d(t, sd[f"{prefix}.{name}.conv.weight"], None, 2, 1, 1)
        return torch.relu(bn(y, f"{prefix}.{name}.bn"))

    conv0 = block(x, "conv0")
    conv2 = block(block(conv0, "conv1", 2), "conv2")
    conv4 = block(block(conv2, "conv3", 2), "conv4")
    t = block(block(conv4, "conv5", 2), "conv6")
    t = conv4 + up(t, "conv7")
    t = conv2 + up(t, "conv9")
    t = conv0 + up(t, "conv11")
    return F.conv3d(t, sd[f"{prefix}.prob.weight"], None, 1, 1)


def _variance_aten(features, proj, samples):
    V = len(features)
    D = samples.shape[1]
    vs = features[0].unsqueeze(2).repeat(1, 1, D, 1, 1)
    vq = vs ** 2
    ref_new = warp.fold_intrinsics(proj[:, 0])
    for v in range(1, V):
        wv = _aten_warp(features[v], warp.fold_intrinsics(proj[:, v]), ref_new, samples)
        vs += wv
        vq += wv.pow_(2)
    return vq.div_(V).sub_(vs.div_(V).pow_(2))


# ----------------------------------------------------------------------------- stage
def depth_stage(features, proj, samples, sd, cr_prefix, impl="spec"):
    """DepthNet_eval.forward (casmvsnet.py:238-311) for one stage."""
    if impl == "aten":
        var = _variance_aten(features, proj, samples)
        logits = _aten_costreg(var, sd, cr_prefix).squeeze(1)
    else:
        var = warp.variance_volume(features, proj, samples)
        logits = conv3d.cost_reg_net(var, sd, cr_prefix).squeeze(1)
    depth, conf, prob = depth_head.depth_head(logits, samples)
    return {"depth": depth, "photometric_confidence": conf}, {"variance": var, "logits": logits, "prob": prob}


def forward_eval(imgs, proj_matrices, depth_values, sd, ndepths=(48, 32, 8), ratios=(4, 2, 1),
                 impl="spec", return_aux=False):
    """CascadeMVSNet_eval.forward (casmvsnet.py:356-417).

    imgs (B,V,3,H,W); proj_matrices {'stageK': (B,V,2,4,4)}; depth_values (B,192);
    sd: reference-named state dict.  Returns the reference's outputs dict."""
    B, V, _, H, W = imgs.shape
    feats = [feature_net(imgs[:, v], sd) for v in range(V)]
    outputs, aux_all = {}, {}
    depth = None
    for s in range(len(ndepths)):
        key = f"stage{s + 1}"
        sc = STAGE_SCALE[s + 1]
        h, w = H // sc, W // sc
        samples = warp.stage_samples(depth, depth_values, ndepths[s], ratios[s], (H, W), (h, w))
        out, aux = depth_stage([f[key] for f in feats], proj_matrices[key], samples, sd,
                               f"cost_regularization.{s}", impl)
        aux["samples"] = samples
        depth = out["depth"]
        outputs[key] = out
        outputs.update(out)
        aux_all[key] = aux
    return (outputs, aux_all) if return_aux else outputs


def forward_train_extras(imgs, proj_matrices, depth_values, sd, ndepth=48, training=True):
    """The stage-1 ``volume_feature_no_ref`` tensor CascadeMVSNet.forward returns next to the
    outputs dict (casmvsnet.py:59,62,82,99-101,231): (B, 3(V-1)+C, D, H/4, W/4).
    BatchNorm in the feature net runs in eval mode here (fixture weights are frozen)."""
    B, V, _, H, W = imgs.shape
    h, w = H // 4, W // 4
    feats = [feature_net(imgs[:, v], sd)["stage1"] for v in range(V)]
    samples = warp.stage1_samples(depth_values, ndepth, h, w)
    small = F.interpolate(imgs.reshape(B * V, 3, H, W), (h, w), mode="bilinear", align_corners=False)
    small = small.reshape(B, V, 3, h, w).permute(1, 0, 2, 3, 4)
    return warp.volume_feature_no_ref(feats, small, proj_matrices["stage1"], samples, training)
