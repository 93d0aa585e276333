"""CPU: one full training iteration in the reference's call sequence (two cascade forwards, renderer,
single backward, Adam step) runs through the drop-in modules -- autograd reaches every trainable tensor of
both models and the update changes the weights.  (On CPU this exercises the delegated op graph; the same
driver runs on the GPU in tests/test_gpu_train.py.)"""
import warnings

import torch

from rc_mvsnet_amd import train_step as ts


def test_train_step_reaches_every_parameter():
    warnings.simplefilter("ignore")
    torch.manual_seed(0)
    dev = torch.device("cpu")
    model, model_nerf, opt = ts.build(dev, ndepths=(8, 8, 8), n_samples=8)
    imgs, proj, dv, batch = ts.synthetic_sample(dev, H=64, W=96, V=4)
    before = {n: p.detach().clone() for n, p in model.named_parameters()}
    losses = ts.train_step(model, model_nerf, opt, imgs, proj, dv, batch)
    assert all(torch.isfinite(torch.tensor(v)) for v in losses.values()), losses
    missing = [n for n, p in list(model.named_parameters()) + list(model_nerf.named_parameters()) if p.grad is None]
    assert not missing, missing
    bad = [n for n, p in list(model.named_parameters()) + list(model_nerf.named_parameters()) if not torch.isfinite(p.grad).all()]
    assert not bad, bad
    changed = sum(int(not torch.equal(before[n], p.detach())) for n, p in model.named_parameters())
    assert changed > 0.9 * len(before)
