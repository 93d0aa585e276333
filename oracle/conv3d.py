"""Oracle: 3-D convolution family and the two 3-D U-Nets built from it.

Test infrastructure (see oracle/__init__.py).  The convolutions are written as explicit
27-tap loops (one small channel contraction per tap) so that padding, stride and the
transposed-convolution index map are spelled out; ``tests/test_oracle_golden.py`` pins them
against the reference's nn.Conv3d / nn.ConvTranspose3d modules.
"""
import torch

BN_EPS = 1e-5


def conv3d(x, w, stride=1):
    """nn.Conv3d(k=3, padding=1, bias=False).  x (B,Ci,D,H,W), w (Co,Ci,3,3,3).
    out[b,co,d,h,w] = sum_{ci,kd,kh,kw} w[co,ci,kd,kh,kw] * x[b,ci,s*d+kd-1,s*h+kh-1,s*w+kw-1]
    (models/modules.py:145-146 with padding=1; out size floor((n-1)/s)+1)."""
    B, Ci, D, H, W = x.shape
    Co = w.shape[0]
    s = stride
    Do, Ho, Wo = (D - 1) // s + 1, (H - 1) // s + 1, (W - 1) // s + 1
    xp = torch.nn.functional.pad(x, (1, 1, 1, 1, 1, 1))
    out = torch.zeros(B, Co, Do, Ho, Wo, dtype=x.dtype)
    for kd in range(3):
        for kh in range(3):
            for kw in range(3):
                xs = xp[:, :, kd:kd + s * (Do - 1) + 1:s, kh:kh + s * (Ho - 1) + 1:s, kw:kw + s * (Wo - 1) + 1:s]
                out += torch.einsum("oc,bcdhw->bodhw", w[:, :, kd, kh, kw], xs)
    return out


def conv_transpose3d(x, w):
    """nn.ConvTranspose3d(k=3, stride=2, padding=1, output_padding=1, bias=False).
    x (B,Ci,D,H,W), w (Ci,Co,3,3,3) -> (B,Co,2D,2H,2W):
    out[b,co,2*id-1+kd, 2*ih-1+kh, 2*iw-1+kw] += x[b,ci,id,ih,iw] * w[ci,co,kd,kh,kw]
    (models/modules.py:189-190 with the kwargs of :483-487)."""
    B, Ci, D, H, W = x.shape
    Co = w.shape[1]
    full = torch.zeros(B, Co, 2 * D + 1, 2 * H + 1, 2 * W + 1, dtype=x.dtype)  # index = out + 1
    for kd in range(3):
        for kh in range(3):
            for kw in range(3):
                full[:, :, kd:kd + 2 * D:2, kh:kh + 2 * H:2, kw:kw + 2 * W:2] += \
                    torch.einsum("co,bcdhw->bodhw", w[:, :, kd, kh, kw], x)
    return full[:, :, 1:, 1:, 1:].contiguous()


def bn_fold(sd, prefix):
    """Eval-mode BatchNorm as y = x*scale + shift  (running statistics, eps=1e-5)."""
    scale = sd[prefix + ".weight"] / torch.sqrt(sd[prefix + ".running_var"] + BN_EPS)
    shift = sd[prefix + ".bias"] - sd[prefix + ".running_mean"] * scale
    return scale, shift


def bn_batch(x, sd, prefix):
    """Train-mode BatchNorm3d forward (biased batch variance; running stats not updated here)."""
    dims = (0, 2, 3, 4)
    mean = x.mean(dims, keepdim=True)
    var = x.var(dims, unbiased=False, keepdim=True)
    g = sd[prefix + ".weight"].reshape(1, -1, 1, 1, 1)
    b = sd[prefix + ".bias"].reshape(1, -1, 1, 1, 1)
    return (x - mean) / torch.sqrt(var + BN_EPS) * g + b


def _affine(x, sd, prefix, training):
    if training:
        return bn_batch(x, sd, prefix)
    scale, shift = bn_fold(sd, prefix)
    return x * scale.reshape(1, -1, 1, 1, 1) + shift.reshape(1, -1, 1, 1, 1)


def cost_reg_net(x, sd, prefix, training=False, return_feat=False):
    """CostRegNet.forward (models/modules.py:492-501): conv+BN+ReLU blocks, three stride-2
    levels, transposed convs with skip adds, final bias-free ``prob`` conv 8->1.
    sd: state dict with reference key names; prefix e.g. 'cost_regularization.0'."""
    def block(t, name, stride=1):
        y = conv3d(t, sd[f"{prefix}.{name}.conv.weight"], stride)
        return torch.relu(_affine(y, sd, f"{prefix}.{name}.bn", training))

    def up(t, name):
        y = conv_transpose3d(t, sd[f"{prefix}.{name}.conv.weight"])
        return torch.relu(_affine(y, sd, f"{prefix}.{name}.bn", training))

    conv0 = block(x, "conv0")
    conv2 = block(block(conv0, "conv1", 2), "conv2")
    conv4 = block(block(conv2, "conv3", 2), "conv4")
    t = block(block(conv4, "conv5", 2), "conv6")
    t = conv4 + up(t, "conv7")
    t = conv2 + up(t, "conv9")
    t = conv0 + up(t, "conv11")
    if return_feat:
        return t
    return conv3d(t, sd[f"{prefix}.prob.weight"])


def neural_volume_net(volume_feature, sd, prefix="MVSNet.cost_reg_2", training=False):
    """Neural_Volume_Net.forward + CostReg.forward (models/render_models.py:753-760,720-734):
    trilinear resize of the depth axis to 128 with align_corners=True, then the same U-Net
    shape as CostRegNet but conv+BN only (no ReLU, despite the class name, :675-686) and no
    final conv.  volume_feature (1,41,D,h,w) -> (1,8,128,h,w)."""
    x = resize_depth_align_corners(volume_feature, 128)

    def block(t, name, stride=1):
        return _affine(conv3d(t, sd[f"{prefix}.{name}.conv.weight"], stride), sd, f"{prefix}.{name}.bn", training)

    def up(t, name):
        return _affine(conv_transpose3d(t, sd[f"{prefix}.{name}.0.weight"]), sd, f"{prefix}.{name}.1", training)

    conv0 = block(x, "conv0")
    conv2 = block(block(conv0, "conv1", 2), "conv2")
    conv4 = block(block(conv2, "conv3", 2), "conv4")
    t = block(block(conv4, "conv5", 2), "conv6")
    t = conv4 + up(t, "conv7")
    t = conv2 + up(t, "conv9")
    t = conv0 + up(t, "conv11")
    return t.reshape(1, -1, *t.shape[2:])


def resize_depth_align_corners(x, out_d):
    """F.interpolate(size=[out_d,H,W], mode='trilinear', align_corners=True) when only the
    depth axis changes (render_models.py:756): src = dst*(in-1)/(out-1), lerp of the two
    neighbouring planes (the H and W axes map to themselves with weight 1)."""
    in_d = x.shape[2]
    if in_d == out_d:
        return x
    scale = torch.tensor((in_d - 1) / (out_d - 1), dtype=torch.float32)
    srcf = scale * torch.arange(out_d, dtype=torch.float32)
    i0 = srcf.floor().long().clamp(max=in_d - 1)
    i1 = (i0 + 1).clamp(max=in_d - 1)
    lam1 = (srcf - i0.float()).reshape(1, 1, -1, 1, 1)
    lam0 = 1.0 - lam1
    return lam0 * x.index_select(2, i0) + lam1 * x.index_select(2, i1)
