"""Oracle: the 2-D feature pyramid (FeatureNet, arch_mode='fpn', num_stage=3).

Test infrastructure (see oracle/__init__.py).  This component is *delegated* to
PyTorch-ROCm in the product, so the oracle simply restates its
graph with functional ATen ops from the reference state dict (models/modules.py:363-464).
"""
import torch
import torch.nn.functional as F

from .conv3d import BN_EPS


def _cbr(x, sd, name, stride, pad, training=False):
    y = F.conv2d(x, sd[name + ".conv.weight"], None, stride, pad)
    if training:
        y = F.batch_norm(y, None, None, sd[name + ".bn.weight"], sd[name + ".bn.bias"], True, 0.1, BN_EPS)
    else:
        y = F.batch_norm(y, sd[name + ".bn.running_mean"], sd[name + ".bn.running_var"],
                         sd[name + ".bn.weight"], sd[name + ".bn.bias"], False, 0.1, BN_EPS)
    return torch.relu(y)


def feature_net(img, sd, prefix="feature", training=False):
    """img (B,3,H,W) -> {'stage1': (B,32,H/4,W/4), 'stage2': (B,16,H/2,W/2), 'stage3': (B,8,H,W)}."""
    p = prefix
    c0 = _cbr(_cbr(img, sd, f"{p}.conv0.0", 1, 1, training), sd, f"{p}.conv0.1", 1, 1, training)
    c1 = _cbr(c0, sd, f"{p}.conv1.0", 2, 2, training)
    c1 = _cbr(_cbr(c1, sd, f"{p}.conv1.1", 1, 1, training), sd, f"{p}.conv1.2", 1, 1, training)
    c2 = _cbr(c1, sd, f"{p}.conv2.0", 2, 2, training)
    c2 = _cbr(_cbr(c2, sd, f"{p}.conv2.1", 1, 1, training), sd, f"{p}.conv2.2", 1, 1, training)
    out = {"stage1": F.conv2d(c2, sd[f"{p}.out1.weight"])}
    intra = F.interpolate(c2, scale_factor=2, mode="nearest") + F.conv2d(c1, sd[f"{p}.inner1.weight"], sd[f"{p}.inner1.bias"])
    out["stage2"] = F.conv2d(intra, sd[f"{p}.out2.weight"], None, 1, 1)
    intra = F.interpolate(intra, scale_factor=2, mode="nearest") + F.conv2d(c0, sd[f"{p}.inner2.weight"], sd[f"{p}.inner2.bias"])
    out["stage3"] = F.conv2d(intra, sd[f"{p}.out3.weight"], None, 1, 1)
    return out
