"""Oracle: softmax over the plane axis, soft-argmin depth and photometric confidence.

Test infrastructure (see oracle/__init__.py).
"""
import torch


def softmax_planes(logits):
    """F.softmax(prob_volume_pre, dim=1)  (models/casmvsnet.py:299): exp(x - max) / sum."""
    m = logits.max(dim=1, keepdim=True).values
    e = torch.exp(logits - m)
    return e / e.sum(dim=1, keepdim=True)


def depth_regression(p, depth_values):
    """models/modules.py:519-525: sum_k p_k * d_k over dim 1."""
    if depth_values.dim() <= 2:
        depth_values = depth_values.reshape(*depth_values.shape, 1, 1)
    return torch.sum(p * depth_values, 1)


def photometric_confidence(p):
    """models/casmvsnet.py:302-309: conf = p[i-1] + p[i] + p[i+1] + p[i+2] (zero outside
    [0,D)) at i = clamp(trunc(sum_k p_k * k), 0, D-1).  Returns (conf, index)."""
    B, D, h, w = p.shape
    k = torch.arange(D, dtype=torch.float32)
    idx = depth_regression(p, k).long().clamp(min=0, max=D - 1)
    pp = torch.nn.functional.pad(p, (0, 0, 0, 0, 1, 2))            # planes -1 .. D+1
    win = ((pp[:, 0:D] + pp[:, 1:D + 1]) + pp[:, 2:D + 2]) + pp[:, 3:D + 3]   # window i-1..i+2
    conf = torch.gather(win, 1, idx.unsqueeze(1)).squeeze(1)
    return conf, idx


def depth_head(logits, depth_samples):
    """logits (B,D,h,w) from the prob conv, depth_samples (B,D,h,w) -> depth, conf, prob."""
    p = softmax_planes(logits)
    depth = depth_regression(p, depth_samples)
    conf, _ = photometric_confidence(p)
    return depth, conf, p
