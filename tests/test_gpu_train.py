"""GPU: the training iteration driver on the MI355X (BASELINE config 3 call sequence, reduced size), and
inference after the update still goes through the HIP path and sees the updated weights (plan caches are
keyed on tensor versions)."""
import warnings

import pytest
import torch

pytestmark = pytest.mark.gpu


def test_train_step_then_hip_inference():
    from rc_mvsnet_amd import train_step as ts, _lib
    from rc_mvsnet_amd.casmvsnet import CascadeMVSNet_eval
    _lib.load()
    warnings.simplefilter("ignore")
    dev = torch.device("cuda:0")
    model, model_nerf, opt = ts.build(dev, ndepths=(16, 8, 8), n_samples=32)
    imgs, proj, dv, batch = ts.synthetic_sample(dev, H=128, W=160, V=4)
    model.eval()
    with torch.no_grad():
        d0 = model(imgs, proj, dv)[0]["depth"].clone()
    l1 = ts.train_step(model, model_nerf, opt, imgs, proj, dv, batch)
    l2 = ts.train_step(model, model_nerf, opt, imgs, proj, dv, batch)
    for v in list(l1.values()) + list(l2.values()):
        assert v == v and abs(v) < 1e9
    grads_ok = all(p.grad is not None and torch.isfinite(p.grad).all() for p in list(model.parameters()) + list(model_nerf.parameters()))
    assert grads_ok
    model.eval()
    with torch.no_grad():
        d1 = model(imgs, proj, dv)[0]["depth"]
    assert torch.isfinite(d1).all()
    assert float((d1 - d0).abs().max()) > 0.0            # the HIP plan picked up the updated weights
