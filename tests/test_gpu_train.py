"""GPU: the training iteration driver on the MI355X (BASELINE config 3 call sequence, reduced size), and
inference after the update still goes through the HIP path and sees the updated weights (plan caches are
keyed on tensor versions)."""
import os
import warnings

import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"          # tests/test_emu_gpu_suite_cpu.py re-runs a subset of this file on the CPU emulation with DEV = "cpu"


def test_train_step_then_hip_inference():
    from rc_mvsnet_amd import train_step as ts, _lib
    from rc_mvsnet_amd.casmvsnet import CascadeMVSNet_eval
    _lib.load()
    warnings.simplefilter("ignore")
    dev = torch.device(DEV)
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


# ------------------------------------------------------------------------------------------------
# native training kernels (include/rcmvs.h, section "training") against torch autograd in float64 on the CPU
# ------------------------------------------------------------------------------------------------
def _rel(a, b):
    return float((a.double() - b.double()).abs().max() / b.double().abs().max().clamp_min(1e-30))


@pytest.mark.parametrize("ci,co,stride,transposed", [
    (8, 8, 1, False), (16, 8, 1, False), (32, 8, 1, False), (16, 16, 1, False), (32, 32, 1, False), (64, 64, 1, False),
    (8, 16, 2, False), (16, 32, 2, False), (32, 64, 2, False), (64, 32, 2, True), (32, 16, 2, True), (16, 8, 2, True)])
@pytest.mark.parametrize("relu,with_res", [(True, True), (False, False)])
def test_conv_bn_relu_block_forward_backward(ci, co, stride, transposed, relu, with_res):
    """One Conv3d / Deconv3d block in train mode (models/modules.py:149-157,196-204) + skip add: output, batch
    statistics and the gradients w.r.t. input, weight, gamma, beta and the skip tensor."""
    import torch.nn.functional as F
    from rc_mvsnet_amd import _lib, train_ops
    _lib.load()
    dev = DEV
    g = torch.Generator().manual_seed(ci * 100 + co + stride)
    B, D, H, W = (2, 8, 8, 16) if relu else (1, 4, 6, 10)             # second shape: ragged width (Wo = 10 or 5)
    x = torch.randn(B, ci, D, H, W, generator=g)
    w = torch.randn((ci, co, 3, 3, 3) if transposed else (co, ci, 3, 3, 3), generator=g) * (1.0 / (27 * ci) ** 0.5)
    gamma = 0.5 + torch.rand(co, generator=g)
    beta = 0.2 * torch.randn(co, generator=g)
    # ---- reference (float64)
    xr, wr, gr, br = (t.double().requires_grad_(True) for t in (x, w, gamma, beta))
    if transposed:
        yr = F.conv_transpose3d(xr, wr, stride=2, padding=1, output_padding=1)
    else:
        yr = F.conv3d(xr, wr, stride=stride, padding=1)
    zr = F.batch_norm(yr, None, None, gr, br, training=True, eps=1e-5)
    if relu:
        zr = F.relu(zr)
    res = torch.randn(zr.shape, generator=g) if with_res else None
    rr = res.double().requires_grad_(True) if with_res else None
    if with_res:
        zr = zr + rr
    G = torch.randn(zr.shape, generator=g)
    (zr * G.double()).sum().backward()
    # ---- HIP
    cl = lambda t: t.permute(0, 2, 3, 4, 1).contiguous().to(dev)
    xg = cl(x).requires_grad_(True)
    wg, gg, bg = (t.to(dev).requires_grad_(True) for t in (w, gamma, beta))
    rg = cl(res).requires_grad_(True) if with_res else None
    cfg = {"transposed": transposed, "stride": stride, "relu": relu, "eps": 1e-5, "momentum": 0.1, "group": None}
    rm, rv = torch.zeros(co, device=dev), torch.ones(co, device=dev)
    z = train_ops.ConvBnReluFn.apply(xg, wg, gg, bg, rg, rm, rv, cfg)
    (z * cl(G)).sum().backward()
    yr_d = yr.detach()
    assert _rel(z.detach().cpu().permute(0, 4, 1, 2, 3), zr.detach()) < 2e-5
    # running statistics: nn.BatchNorm's momentum update (unbiased variance)
    assert float((rm.cpu() - 0.1 * yr_d.mean(dim=(0, 2, 3, 4))).abs().max()) < 1e-5
    assert _rel(rv.cpu(), 0.9 + 0.1 * yr_d.var(dim=(0, 2, 3, 4), unbiased=True)) < 2e-5
    errs = {"dx": _rel(xg.grad.cpu().permute(0, 4, 1, 2, 3), xr.grad), "dw": _rel(wg.grad.cpu(), wr.grad),
            "dgamma": _rel(gg.grad.cpu(), gr.grad), "dbeta": _rel(bg.grad.cpu(), br.grad)}
    if with_res:
        errs["dres"] = _rel(rg.grad.cpu().permute(0, 4, 1, 2, 3), rr.grad)
    print(f"block {ci}->{co} s{stride} T={transposed}: " + " ".join(f"{k}={v:.1e}" for k, v in errs.items()))
    assert max(errs.values()) < 1e-4


def test_prob_depth_head_backward():
    import torch.nn.functional as F
    from rc_mvsnet_amd import _lib, train_ops
    _lib.load()
    dev = DEV
    g = torch.Generator().manual_seed(5)
    B, D, h, w = 2, 8, 16, 24
    x8 = torch.randn(B, 8, D, h, w, generator=g)
    wp = torch.randn(1, 8, 3, 3, 3, generator=g) * 0.3
    planes = torch.stack((450 + 100 * torch.rand(B, h, w, generator=g), 2.0 + torch.rand(B, h, w, generator=g)), dim=-1)
    G = torch.randn(B, h, w, generator=g)
    xr, wr = x8.double().requires_grad_(True), wp.double().requires_grad_(True)
    logits = F.conv3d(xr, wr, padding=1).squeeze(1)
    p = F.softmax(logits, dim=1)
    dvals = planes[..., 0].unsqueeze(1).double() + torch.arange(D).view(1, D, 1, 1).double() * planes[..., 1].unsqueeze(1).double()
    depth_r = (p * dvals).sum(1)
    (depth_r * G.double()).sum().backward()
    xg = x8.permute(0, 2, 3, 4, 1).contiguous().to(dev).requires_grad_(True)
    wg = wp.to(dev).requires_grad_(True)
    depth, conf = train_ops.ProbDepthHeadFn.apply(xg, wg, planes.to(dev).contiguous())
    (depth * G.to(dev)).sum().backward()
    assert float((depth.detach().cpu() - depth_r.detach()).abs().max()) < 2e-3
    e_dx, e_dw = _rel(xg.grad.cpu().permute(0, 4, 1, 2, 3), xr.grad), _rel(wg.grad.cpu(), wr.grad)
    print(f"depth head bwd: dx {e_dx:.1e} dw {e_dw:.1e}")
    assert e_dx < 1e-4 and e_dw < 1e-4


@pytest.mark.parametrize("shape", [(2, 5, 11, 21), (1, 20, 9, 37), (1, 3, 17, 16), (1, 35, 8, 16)])
def test_prob_conv_weight_gradient_marching_kernel(shape):
    """Weight gradient of the 8 -> 1 prob conv on the plane-marching kernel (persistent blocks over (tile, z chunk) items, rolling
    dy[z-1], dy[z], dy[z+1], butterfly reduction): against fp64 autograd on ragged tiles, several z chunks (D > 16), batch 2."""
    import torch.nn.functional as F
    from rc_mvsnet_amd import _lib, train_ops
    _lib.load()
    B, D, H, W = shape
    g = torch.Generator().manual_seed(D + W)
    x = torch.randn(B, 8, D, H, W, generator=g)
    dy = torch.randn(B, 1, D, H, W, generator=g)
    w = torch.zeros(1, 8, 3, 3, 3, dtype=torch.float64, requires_grad=True)
    (F.conv3d(x.double(), w, padding=1) * dy.double()).sum().backward()
    dw = train_ops.conv3d_wgrad(x.permute(0, 2, 3, 4, 1).contiguous().to(DEV), dy.permute(0, 2, 3, 4, 1).contiguous().to(DEV), 1).cpu()
    ref = w.grad[0].permute(1, 2, 3, 0).reshape(27, 8, 1)             # (27, Ci, Co)
    err = float((dw.double() - ref).abs().max() / ref.abs().max())
    assert err < 2e-5, err


@pytest.mark.parametrize("Ci", [4, 8, 16, 32])
@pytest.mark.parametrize("shape", [(2, 3, 5, 21), (1, 1, 3, 200), (1, 4, 9, 37), (1, 1, 16, 70)])
def test_conv3d_weight_gradient_cout8_paired_columns(Ci, shape):
    """Weight gradient of the Cout = 8 stride-1 layers: the idle eight MFMA columns take dy shifted by one cell along w, which
    yields tap kw - 1 from the rows of tap kw (18 row taps instead of 27).  Against fp64 autograd: ragged rows, batch 2, one-plane
    volumes whose rows are split across blocks in w (the shifted cell of a chunk's last x cell belongs to the next chunk)."""
    import torch.nn.functional as F
    from rc_mvsnet_amd import _lib, train_ops
    _lib.load()
    B, D, H, W = shape
    g = torch.Generator().manual_seed(Ci + W)
    x = torch.randn(B, Ci, D, H, W, generator=g)
    dy = torch.randn(B, 8, D, H, W, generator=g)
    w = torch.zeros(8, Ci, 3, 3, 3, dtype=torch.float64, requires_grad=True)
    (F.conv3d(x.double(), w, padding=1) * dy.double()).sum().backward()
    dw = train_ops.conv3d_wgrad(x.permute(0, 2, 3, 4, 1).contiguous().to(DEV), dy.permute(0, 2, 3, 4, 1).contiguous().to(DEV), 1).cpu()
    ref = w.grad.permute(2, 3, 4, 1, 0).reshape(27, Ci, 8)               # (27, Ci, Co)
    if D == 1:
        ref = ref.clone(); ref[:9] = 0; ref[18:] = 0                     # one-plane volumes: the kernel skips the dead tap planes (padding only)
        assert float(w.grad[:, :, 0].abs().max()) == 0.0 and float(w.grad[:, :, 2].abs().max()) == 0.0
    err = float((dw.double() - ref).abs().max() / ref.abs().max())
    assert err < 2e-5, err


def test_cascade_train_native_vs_delegated_gradients():
    """CascadeMVSNet in train mode: the HIP training path (WarpVarianceFn + ConvBnReluFn + ProbDepthHeadFn) and the
    reference's op graph with autograd (oracle/aten_graph.py, same device) produce the same outputs, parameter gradients
    and running statistics."""
    import copy
    from oracle import aten_graph
    from rc_mvsnet_amd import _lib, synthetic
    from rc_mvsnet_amd.casmvsnet import CascadeMVSNet
    _lib.load()
    warnings.simplefilter("ignore")
    dev = DEV
    m1 = CascadeMVSNet(ndepths=[16, 8, 8], depth_interals_ratio=[4, 2, 1])
    m1.load_state_dict(synthetic.cascade_state_dict(0, prob_gain=2.0), strict=True)
    m1 = m1.to(dev).train()
    m2 = copy.deepcopy(m1)
    imgs, pm, dv = synthetic.cascade_inputs(1, 3, 64, 96, 0)
    imgs, dv = imgs.to(dev), dv.to(dev)
    pm = {k: v.to(dev) for k, v in pm.items()}

    def run(model, forward):
        out, noref = forward(model, imgs, pm, dv)
        loss = sum((out[f"stage{s}"]["depth"] - 600.0).abs().mean() for s in (1, 2, 3)) + 1e-2 * noref.pow(2).mean()
        loss.backward()
        return out, noref, loss

    out1, nr1, l1 = run(m1, lambda m, *a: m(*a))
    out2, nr2, l2 = run(m2, aten_graph.cascade_forward)
    assert _rel(out1["stage1"]["depth"], out2["stage1"]["depth"]) < 1e-4
    assert _rel(nr1, nr2) < 1e-4
    worst = ("", 0.0)
    for (n1, p1), (n2, p2) in zip(m1.named_parameters(), m2.named_parameters()):
        assert p1.grad is not None and p2.grad is not None, n1
        if n1.startswith("cost_regularization.0") or n1.startswith("feature"):      # stage 1: identical hypothesis planes
            e = _rel(p1.grad, p2.grad)
            if e > worst[1]:
                worst = (n1, e)
    print(f"loss {float(l1):.6f} vs {float(l2):.6f}; worst stage-1/feature grad mismatch {worst[1]:.2e} at {worst[0]}")
    assert worst[1] < 3e-2          # ill-conditioned seeded network: see test_hip_training_path_vs_reference_gradients
    b1 = dict(m1.named_buffers())
    for n2, b in m2.named_buffers():
        if n2.startswith("cost_regularization.0") and "running" in n2:
            assert _rel(b1[n2], b) < 1e-4, n2


def test_neural_volume_net_train_native_vs_delegated():
    """Rendering branch, a8 in train mode: plane resize + CostReg (conv + batch-stat norm, no ReLU, 41 -> 44 padded
    input channels) on the HIP kernels vs the reference op graph over the module's children (oracle/aten_graph.py): volume, input gradient, parameter gradients."""
    import copy
    from oracle import aten_graph
    from rc_mvsnet_amd import _lib
    from rc_mvsnet_amd.render_consist_net import Neural_Volume_Net
    _lib.load()
    warnings.simplefilter("ignore")
    dev = DEV
    torch.manual_seed(3)
    m1 = Neural_Volume_Net().to(dev).train()
    for mod in m1.modules():
        if isinstance(mod, torch.nn.BatchNorm3d):
            mod.weight.data.uniform_(0.5, 1.5)
            mod.bias.data.normal_(0.0, 0.1)
    m2 = copy.deepcopy(m1)
    g = torch.Generator().manual_seed(0)
    x = torch.randn(1, 41, 12, 16, 24, generator=g).to(dev)
    G = torch.randn(1, 8, 128, 16, 24, generator=g).to(dev)

    def run(model, forward):
        xi = x.clone().requires_grad_(True)
        v = forward(model, xi)
        (v * G).sum().backward()
        return v.detach(), xi.grad

    v1, gx1 = run(m1, lambda m, t: m(t))
    v2, gx2 = run(m2, aten_graph.neural_volume)
    e_v, e_gx = _rel(v1, v2), _rel(gx1, gx2)
    worst = max(_rel(p1.grad, p2.grad) for p1, p2 in zip(m1.parameters(), m2.parameters()))
    print(f"Neural_Volume_Net train: volume {e_v:.1e}  d/d input {e_gx:.1e}  worst param grad {worst:.1e}")
    assert e_v < 1e-4 and e_gx < 1e-3 and worst < 1e-3


def test_renderer_train_native_vs_delegated():
    """Rendering_Consistency_Net in train mode with injected draws: HIP training path (volume network, point-feature
    scatter, NeRF MLP, compositing recurrence -- all on the library) vs the reference's op graph with autograd
    (oracle/aten_graph.py) -- outputs, gradient w.r.t. the warped volume feature, parameter gradients."""
    import copy
    from oracle import aten_graph
    from rc_mvsnet_amd import _lib, synthetic, train_step as ts
    from rc_mvsnet_amd.render_consist_net import Rendering_Consistency_Net
    _lib.load()
    warnings.simplefilter("ignore")
    dev = DEV
    H, W, V, S = 64, 96, 4, 16
    m1 = Rendering_Consistency_Net(ts.render_args(S))
    m1.load_state_dict(synthetic.render_state_dict(1), strict=True)
    m1 = m1.to(dev).train()
    m2 = copy.deepcopy(m1)
    batch = {k: v.to(dev) for k, v in synthetic.render_batch(V, H, W, 0).items()}
    pix, eps, u = (t.to(dev) for t in synthetic.render_randoms(H, W, 1024, S, 5))
    # rays through border pixels re-project onto |grid| == 1 exactly, where the strict in-bounds mask is decided by the
    # last ulp (see test_gpu_render.py): keep the draws in the interior so both paths see the same masks
    pix = torch.stack((pix[0].clamp(1, W - 2), pix[1].clamp(1, H - 2)))
    g = torch.Generator().manual_seed(2)
    vfw = (0.5 * torch.randn(1, 41, 12, H // 4, W // 4, generator=g)).to(dev)
    pseudo = (500.0 + 300.0 * torch.rand(1, H, W, generator=g)).to(dev)

    def run(model, forward):
        x = vfw.clone().requires_grad_(True)
        rgb, feat, wts, dpred, alpha, _, rdepth, target = forward(model, x, pseudo, dict(batch), (pix, eps, u))
        loss = torch.nn.functional.mse_loss(rgb, target) + 1e-3 * torch.nn.functional.smooth_l1_loss(dpred, rdepth) + \
            1e-2 * wts.pow(2).mean() + 1e-2 * alpha.mean()
        loss.backward()
        return (rgb.detach(), dpred.detach(), wts.detach()), x.grad, loss

    o1, gx1, l1 = run(m1, lambda m, x, p, b, r: m(x, p, b, randoms=r))
    o2, gx2, l2 = run(m2, aten_graph.render_forward)
    e_out = max(_rel(a, b) for a, b in zip(o1, o2))
    e_gx = _rel(gx1, gx2)
    worst = ("", 0.0)
    for (n1, p1), (n2, p2) in zip(m1.named_parameters(), m2.named_parameters()):
        assert (p1.grad is None) == (p2.grad is None), n1
        if p1.grad is not None:
            e = _rel(p1.grad, p2.grad)
            if e > worst[1]:
                worst = (n1, e)
    print(f"renderer train: loss {float(l1):.6f} vs {float(l2):.6f}; outputs {e_out:.1e}  d/d volume feature {e_gx:.1e}  worst param grad {worst[1]:.1e} ({worst[0]})")
    assert e_out < 1e-3 and e_gx < 2e-2 and worst[1] < 2e-2


def test_hip_training_path_vs_reference_gradients():
    """The HIP training path against gradients produced by the REFERENCE's own autograd on the well-conditioned fixture
    (tests/golden/train_grads_cond.npz, make_golden.py --only-train-grads: prob.weight x1, 128x160, D = 16/16/8): K1 scatter, conv
    data / weight gradients, batch-statistics BatchNorm forward and backward with running statistics, prob conv + softmax +
    soft-argmin backward.  FeatureNet gradients come through the K1 backward, so they check it end to end.

    Bounds (round 3).  What the error of this comparison is made of was bisected (profiles/r3_grad_outlier_bisect.txt): against the
    fp64 graph the HIP path is at 9e-6 at the gradient of the variance volume EXCEPT in one 3x3x3 neighbourhood -- ONE ReLU of
    cost_regularization.0.conv0 (327,680 output voxels) whose pre-activation sits within a rounding error of zero takes the other
    branch.  One element of N changes a gradient that sums N comparable terms by ~1 / sqrt(N) = 1.7e-3: measured 2.0e-3 at
    conv0's own weight gradient, 6e-4 at the variance gradient, 1.3e-3 at the FeatureNet BatchNorm parameters (everything at or
    upstream of that ReLU); the reference's fp32 graph on the CPU has its own such element in FeatureNet (9e-4 at
    feature.conv2.2.bn.bias against fp64).  So: every tensor DOWNSTREAM of conv0's ReLU (the rest of the 3-D U-Net, the prob conv)
    within 1e-4; the tensors at or upstream of it within 3e-3 (two single-element events), median over all 5e-4.  The older,
    ill-conditioned fixture (train_grads.npz: 64x96, D = 8, prob.weight x2, 6 voxels per channel at the deepest level) stays as the
    CPU pin of the restated graph only."""
    from conftest import load_golden
    from rc_mvsnet_amd import _lib, synthetic
    from rc_mvsnet_amd.casmvsnet import CascadeMVSNet
    _lib.load()
    warnings.simplefilter("ignore")
    dev = DEV
    g = load_golden("train_grads_cond")
    m = CascadeMVSNet(ndepths=[16, 16, 8], depth_interals_ratio=[4, 2, 1])
    m.load_state_dict(synthetic.cascade_state_dict(0, prob_gain=1.0), strict=True)
    m = m.to(dev).train()
    imgs, pm, dv = synthetic.cascade_inputs(1, 3, 128, 160, 0)
    outputs, noref = m(imgs.to(dev), {k: v.to(dev) for k, v in pm.items()}, dv.to(dev))
    loss = ((outputs["stage1"]["depth"] - 600.0) ** 2).mean() / 1e4 + 1e-2 * (noref ** 2).mean()
    loss.backward()
    print(f"loss {float(loss):.6f} vs reference {float(g['loss']):.6f}")
    assert abs(float(loss) - float(g["loss"])) < 1e-5 * abs(float(g["loss"]))
    assert float((outputs["stage1"]["depth"].cpu() - g["depth1"]).abs().max()) < 1e-2
    params = dict(m.named_parameters())
    errs = {}
    for k in g.keys():
        if k.startswith("grad:"):
            errs[k[5:]] = _rel(params[k[5:]].grad.cpu(), torch.as_tensor(g[k]))
    vals = sorted(errs.values())
    worst = max(errs, key=errs.get)
    print(f"gradient mismatch vs reference autograd: median {vals[len(vals) // 2]:.2e}, worst {errs[worst]:.2e} at {worst}")
    print("   " + ", ".join(f"{k.replace('cost_regularization.0.', 'cr0.')} {v:.1e}" for k, v in errs.items()))
    assert vals[len(vals) // 2] < 5e-4 and errs[worst] < 3e-3, errs
    for k, v in errs.items():                                                                # downstream of conv0's ReLU
        if k.startswith("cost_regularization.0.") and not k.startswith("cost_regularization.0.conv0."):
            assert v < 1e-4, (k, v)
    bufs = dict(m.named_buffers())
    assert float((bufs["cost_regularization.0.conv0.bn.running_mean"].cpu() - torch.as_tensor(g["running_mean_conv0"])).abs().max()) < 1e-5
    assert _rel(bufs["cost_regularization.0.conv0.bn.running_var"].cpu(), torch.as_tensor(g["running_var_conv0"])) < 1e-4


def test_featurenet_train_native_vs_delegated():
    """FeatureNet in train mode on the library (every layer as a one-plane volume on the 3-D conv family, 5x5 stride-2
    layers as space-to-depth + 3x3) vs the reference graph over the module's nn.Conv2d / BatchNorm2d children
    (oracle/aten_graph.feature_pyramid): feature maps, input-independent parameter gradients, running statistics."""
    import copy
    from oracle import aten_graph
    from rc_mvsnet_amd import _lib
    from rc_mvsnet_amd.casmvsnet import FeatureNet
    _lib.load()
    warnings.simplefilter("ignore")
    dev = DEV
    torch.manual_seed(1)
    m1 = FeatureNet(base_channels=8, stride=4, num_stage=3, arch_mode="fpn").to(dev).train()
    for mod in m1.modules():
        if isinstance(mod, torch.nn.BatchNorm2d):
            mod.weight.data.uniform_(0.5, 1.5)
            mod.bias.data.normal_(0.0, 0.1)
    m2 = copy.deepcopy(m1)
    g = torch.Generator().manual_seed(0)
    x = torch.randn(2, 3, 64, 96, generator=g).to(dev)
    G = {k: torch.randn(s, generator=g).to(dev) for k, s in (("stage1", (2, 16, 24, 32)), ("stage2", (2, 32, 48, 16)), ("stage3", (2, 64, 96, 8)))}
    # two "views" of one image each, normalised separately: one batched call with segments=2 vs two module calls
    o1 = m1.forward_train_cl(x, segments=2)
    sum((o1[k] * G[k]).sum() for k in G).backward()
    oa, ob = aten_graph.feature_pyramid(m2, x[:1]), aten_graph.feature_pyramid(m2, x[1:])
    o2 = {k: torch.cat((oa[k], ob[k])) for k in oa}
    sum((o2[k].permute(0, 2, 3, 1) * G[k]).sum() for k in G).backward()
    for k in G:
        assert _rel(o1[k].detach(), o2[k].detach().permute(0, 2, 3, 1)) < 1e-4, k
    worst = ("", 0.0)
    for (n1, p1), (n2, p2) in zip(m1.named_parameters(), m2.named_parameters()):
        assert p1.grad is not None and p2.grad is not None, n1
        e = _rel(p1.grad, p2.grad)
        if e > worst[1]:
            worst = (n1, e)
    print(f"FeatureNet train: worst parameter-gradient mismatch {worst[1]:.2e} at {worst[0]}")
    assert worst[1] < 2e-3
    b1 = dict(m1.named_buffers())
    for n2, b in m2.named_buffers():
        if "running" in n2:
            assert _rel(b1[n2], b) < 1e-4, n2


def test_sync_batchnorm_branch_with_simulated_replica(monkeypatch):
    """The SyncBatchNorm branch of ConvBnReluFn (statistics and backward sums all-reduced as fp64) on one GPU: with
    dist.all_reduce replaced by "add an identical replica" (t *= 2), the normalisation must be unchanged -- same mean /
    variance / running statistics, same output and input gradient -- while the parameter gradients stay the local ones."""
    import torch.distributed as dist
    from rc_mvsnet_amd import _lib, train_ops
    _lib.load()
    dev = DEV
    g = torch.Generator().manual_seed(3)
    x = torch.randn(2, 8, 8, 16, 16, generator=g).to(dev)
    w = (torch.randn(16, 16, 3, 3, 3, generator=g) * 0.05).to(dev)
    gamma, beta = (0.5 + torch.rand(16, generator=g)).to(dev), (0.1 * torch.randn(16, generator=g)).to(dev)
    G = torch.randn(2, 8, 8, 16, 16, generator=g).to(dev)
    calls = {"n": 0}

    def fake_all_reduce(t, group=None, op=None):
        calls["n"] += 1
        t.mul_(2.0)

    def run(group):
        xi, wi, gi, bi = (t.clone().requires_grad_(True) for t in (x, w, gamma, beta))
        rm, rv = torch.zeros(16, device=dev), torch.ones(16, device=dev)
        cfg = {"transposed": False, "stride": 1, "relu": True, "eps": 1e-5, "momentum": 0.1, "group": group}
        z = train_ops.ConvBnReluFn.apply(xi, wi, gi, bi, None, rm, rv, cfg)
        (z * G).sum().backward()
        return z.detach(), xi.grad, wi.grad, gi.grad, bi.grad, rm, rv

    ref = run(None)
    monkeypatch.setattr(dist, "all_reduce", fake_all_reduce)
    syn = run("fake-group")
    assert calls["n"] == 2                                            # one exchange forward, one backward
    names = ("z", "dx", "dw", "dgamma", "dbeta", "running_mean", "running_var")
    for n, a, b in zip(names, syn, ref):
        tol = 2e-4 if n == "running_var" else 1e-5                    # unbiased correction N/(N-1) sees the doubled count
        assert _rel(a, b) < tol, n


@pytest.mark.parametrize("N,S", [(64, 16), (37, 5), (130, 33)])
def test_nerf_mlp_train_forward_backward_vs_fp64_autograd(N, S):
    """rcmvs_nerf_mlp_train_fwd / rcmvs_nerf_mlp_bwd (11 linear layers with the multiplicative feature bias, the skip
    concatenation and the sigmoid / ReLU heads on the fp32 MFMA chain) against fp64 autograd of the oracle's restatement of
    Renderer_ours.forward: outputs, the gradient of the point features and all 22 parameter gradients, per tensor."""
    from oracle import render as orr
    from rc_mvsnet_amd import ops, train_ops
    from rc_mvsnet_amd.render_consist_net import Renderer_ours
    g = torch.Generator().manual_seed(N * 100 + S)
    torch.manual_seed(N * 100 + S)                     # the module's kaiming initialisation draws from the global generator
    net = Renderer_ours(D=6, W=128, input_ch=63, input_ch_views=3, output_ch=4, input_ch_feat=20, skips=[4], use_viewdirs=True)
    with torch.no_grad():
        for p in net.parameters():                     # biases are zero-initialised: give every tensor signal
            if p.dim() == 1:
                p.copy_(0.1 * torch.randn(p.shape, generator=g))
        net.pts_bias.bias.add_(1.0)                    # keeps the multiplicative bias away from 0 on most units
    M = N * S
    ndc = torch.rand(N, S, 3, generator=g)
    feat = torch.zeros(M, 32)
    feat[:, :20] = torch.randn(M, 20, generator=g)
    dirs = torch.randn(N, 3, generator=g)
    w2c = torch.eye(4)
    w2c[:3, :3] = torch.linalg.qr(torch.randn(3, 3, generator=g))[0]
    graw = torch.randn(N, S, 4, generator=g)
    # fp64 reference
    sd = {"network_fn.nerf." + k: v.detach().double().requires_grad_(True) for k, v in net.state_dict().items()}
    f64 = feat[:, :20].double().requires_grad_(True)
    angle = (dirs / dirs.norm(dim=-1, keepdim=True)).double() @ w2c[:3, :3].double().t()
    x63 = orr.embed(ndc.double().reshape(M, 3))
    ref = orr.nerf_mlp(x63, f64, angle[:, None].expand(-1, S, -1).reshape(M, 3), sd)
    (ref * graw.double().reshape(M, 4)).sum().backward()
    # the same graph in fp32 on the CPU: the yardstick for what single-precision sums over M points can deliver
    sd32 = {k: v.detach().float().requires_grad_(True) for k, v in sd.items()}
    f32 = feat[:, :20].clone().requires_grad_(True)
    ref32 = orr.nerf_mlp(x63.float(), f32, angle.float()[:, None].expand(-1, S, -1).reshape(M, 3), sd32)
    (ref32 * graw.reshape(M, 4)).sum().backward()
    # HIP
    net = net.to(DEV)
    fh = feat.to(DEV).requires_grad_(True)
    raw = train_ops.nerf_mlp_train(net, ndc.to(DEV), fh, dirs.to(DEV), w2c.to(DEV))
    (raw * graw.to(DEV)).sum().backward()
    scale = float(ref.detach().abs().max())
    assert float((raw.detach().cpu().double().reshape(M, 4) - ref.detach()).abs().max()) < 2e-5 * scale
    gf = fh.grad.cpu().double()
    assert float(gf[:, 20:].abs().max()) == 0.0
    assert float((gf[:, :20] - f64.grad).norm()) <= max(1e-5, 2.0 * float((f32.grad.double() - f64.grad).norm() / f64.grad.norm())) * float(f64.grad.norm())
    assert float((gf[:, :20] - f64.grad).abs().max()) < 2e-5 * float(f64.grad.abs().max())
    for name, p in net.named_parameters():
        want = sd["network_fn.nerf." + name].grad
        got = p.grad.cpu().double()
        assert got.shape == want.shape, name
        # relative Frobenius error: within 1e-5, or no worse than twice what fp32 autograd of the same graph reaches on the CPU
        e_hip = float((got - want).norm() / want.norm())
        e_32 = float((sd32["network_fn.nerf." + name].grad.double() - want).norm() / want.norm())
        assert e_hip <= max(1e-5, 2.0 * e_32), (name, e_hip, e_32)
        assert e_hip <= 5e-5, (name, e_hip)
    # inference forward on the same weights agrees with the training forward
    with torch.no_grad():
        blob = ops.pack_nerf_weights({n: (m.weight, m.bias) for n, m in
                                      [("pts_bias", net.pts_bias), ("alpha_linear", net.alpha_linear), ("feature_linear", net.feature_linear),
                                       ("views_linears.0", net.views_linears[0]), ("rgb_linear", net.rgb_linear)] +
                                      [(f"pts_linears.{i}", net.pts_linears[i]) for i in range(6)]})
        raw_inf = ops.nerf_mlp(ndc.to(DEV), feat.to(DEV).clone(), dirs.to(DEV), w2c.to(DEV), blob)
    assert torch.equal(raw_inf, raw.detach())


def test_train_step_full_size_config3():
    """BASELINE configs[2] at its full size under the driver's eyes: ONE training iteration with 4 views at 512x640,
    D = 48/32/8 and 1024 rays x 128 samples on the HIP path -- finite losses, finite gradients for every parameter of both
    models, BatchNorm running statistics and step counters updated, weights changed by the optimizer.  The self-supervised loss
    of forward #1 is cross-checked against the reference op graph (oracle/aten_graph.py on the host cores, forward only, same
    weights): the seeded network is chaotic (prob head x20), so the bound is loose; the tight per-block and per-tensor
    checks are the tests above."""
    import copy
    from oracle import aten_graph
    from rc_mvsnet_amd import losses, train_step as ts, _lib
    _lib.load()
    warnings.simplefilter("ignore")
    dev = torch.device(DEV)
    model, model_nerf, opt = ts.build(dev)
    imgs, proj, dv, batch = ts.synthetic_sample(dev, H=512, W=640, V=4)
    m2 = copy.deepcopy(model).cpu().train()                     # on the host cores: a fresh GPU box would spend a minute in MIOpen's kernel search
    with torch.no_grad():
        out2, _ = aten_graph.cascade_forward(m2, imgs.cpu(), {k: v.cpu() for k, v in proj.items()}, dv.cpu())
        out2 = {k: ({kk: vv.to(dev) for kk, vv in v.items()} if isinstance(v, dict) else v.to(dev)) for k, v in out2.items()}
        base2 = float(losses.UnsupLossMultiStage()(out2, imgs, proj, dlossw=list(ts.DLOSSW))[0])
    del m2, out2
    bn = model.cost_regularization[2].conv0.bn
    rm0, w0 = bn.running_mean.clone(), model.cost_regularization[0].conv0.conv.weight.detach().clone()
    l1 = ts.train_step(model, model_nerf, opt, imgs, proj, dv, batch)
    print("HIP iteration", l1, "| reference-graph forward #1 loss", base2)
    assert all(v == v and abs(v) < 1e9 for v in l1.values()), l1
    for n, p in list(model.named_parameters()) + list(model_nerf.named_parameters()):
        assert p.grad is not None and bool(torch.isfinite(p.grad).all()), n
    assert float((bn.running_mean - rm0).abs().max()) > 0.0
    assert int(bn.num_batches_tracked) == 2                                    # two cascade passes per iteration
    assert float((model.cost_regularization[0].conv0.conv.weight.detach() - w0).abs().max()) > 0.0
    assert abs(l1["base"] - base2) <= 5e-2 * abs(base2), (l1["base"], base2)


@pytest.mark.parametrize("grad_method", ["detach", "undetach"])
def test_cascade_train_gradients_conditioned_network(grad_method):
    """The HIP training path against the reference op graph evaluated in FP64 (oracle/aten_graph.py on the CPU) on a
    well-conditioned network: trained-like probability head (prob.weight x1) and volumes large enough that the deepest U-Net
    level holds dozens of voxels per channel (128x160 images, D = 16/16/8).  Losses and outputs within 1e-5 / 1e-4.

    Gradients: the map input -> gradient is piecewise smooth (ReLU masks, bilinear cells, the floor in the sampler), and at this
    size a RELATIVE INPUT PERTURBATION OF 1e-6 -- the size of the fp32 round-off every implementation carries in its
    activations -- already crosses some of those edges: in exact arithmetic it moves the parameter gradients by 2e-3 (median) to
    8e-3 (measured here, same fp64 graph, perturbed images).  An fp32 implementation therefore cannot be expected to agree with
    the exact gradient better than that, and agreement between two fp32 implementations at 1e-5 is luck with the edges, not
    accuracy.  The bar: every stage-1 / feature-pyramid parameter gradient of the HIP path is within the fixture's own
    conditioning (twice the worst fp64 gradient change over three draws of the 1e-6 perturbation) of the fp64 gradient, and the median error is
    below 1e-4 or the median of that change (undetach: hypothesis planes of stages 2 / 3 follow the previous depth, 6e-4).  grad_method='undetach' (models/casmvsnet.py:192) adds the gradient that later stages send back through the
    previous stage's depth (the loss then includes all three stages)."""
    import copy
    from oracle import aten_graph
    from rc_mvsnet_amd import _lib, synthetic
    from rc_mvsnet_amd.casmvsnet import CascadeMVSNet
    _lib.load()
    warnings.simplefilter("ignore")
    m0 = CascadeMVSNet(ndepths=[16, 16, 8], depth_interals_ratio=[4, 2, 1], grad_method=grad_method)
    m0.load_state_dict(synthetic.cascade_state_dict(0, prob_gain=1.0), strict=True)
    imgs, pm, dv = synthetic.cascade_inputs(1, 3, 128, 160, 0)
    stages = ("stage1",) if grad_method == "detach" else ("stage1", "stage2", "stage3")

    def run(forward, device, dtype, images):
        model = copy.deepcopy(m0).to(device=device, dtype=dtype).train()
        out, noref = forward(model, images.to(device=device, dtype=dtype), {k: v.to(device=device, dtype=dtype) for k, v in pm.items()},
                             dv.to(device=device, dtype=dtype))
        loss = sum(((out[k]["depth"] - 600.0) ** 2).mean() for k in stages) / 1e4 + 1e-2 * (noref ** 2).mean()
        loss.backward()
        grads = {n: (None if q.grad is None else q.grad.detach().double().cpu()) for n, q in model.named_parameters()}
        return float(loss), out["stage1"]["depth"].detach().double().cpu(), noref.detach().double().cpu(), grads

    l64, d64, nr64, g64 = run(aten_graph.cascade_forward, "cpu", torch.float64, imgs)
    gen = torch.Generator().manual_seed(0)
    g64p = [run(aten_graph.cascade_forward, "cpu", torch.float64, imgs * (1 + 1e-6 * torch.randn(imgs.shape, generator=gen)))[3]
            for _ in range(3)]          # edge crossings are discrete events: one draw moves the worst gradient by 1e-3, another by 8e-3
    l1, d1, nr1, g1 = run(lambda m, *a: m(*a), DEV, torch.float32, imgs)
    assert abs(l1 - l64) <= 1e-5 * abs(l64), (l1, l64)
    assert _rel(d1, d64) < 1e-5 and _rel(nr1, nr64) < 1e-4
    errs, sens = {}, {}
    for n in g64:
        assert (g1[n] is None) == (g64[n] is None), n
        if g64[n] is not None and (n.startswith("cost_regularization.0") or n.startswith("feature")):
            nrm = g64[n].norm().clamp_min(1e-300)
            errs[n] = float((g1[n] - g64[n]).norm() / nrm)
            sens[n] = max(float((gp[n] - g64[n]).norm() / nrm) for gp in g64p)
    worst = max(errs, key=errs.get)
    ev, sv = sorted(errs.values()), sorted(sens.values())
    print(f"conditioned network ({grad_method}): loss {l1:.6f} vs fp64 {l64:.6f}; gradient error vs fp64 (Frobenius) median {ev[len(ev) // 2]:.2e}, "
          f"worst {errs[worst]:.2e} at {worst}; fp64 gradient change under 1e-6 input perturbations (max of 3 draws): median {sv[len(sv) // 2]:.2e}, worst {sv[-1]:.2e}")
    # absolute bounds (round 3; the sensitivities above are printed for the record): what the worst case consists of is known --
    # single ReLU pre-activations within one rounding error of zero, profiles/r3_grad_outlier_bisect.txt
    assert errs[worst] <= 2e-3, (worst, errs[worst], sv[-1])
    assert ev[len(ev) // 2] <= (1e-4 if grad_method == "detach" else 1e-3), (ev[len(ev) // 2], sv[len(sv) // 2])


# ------------------------------------------------------------------------------------------------
# round 3: the bookkeeping around the training kernels (selective weight pack, pack reuse, fused BatchNorm forms, gradient finish)
# ------------------------------------------------------------------------------------------------
def test_selective_weight_pack_matches_full_blob_and_is_checked():
    """rcmvs_pack_conv3d_weight_sel writes only the image the production dispatch reads for a call; the result of that call equals
    the one on the full blob bit for bit, and a call that would read an image that was not written is refused on the host."""
    from rc_mvsnet_amd import ops, _lib
    _lib.load()
    g = torch.Generator().manual_seed(3)
    for (ci, co, stride, shape) in ((16, 8, 1, (1, 4, 9, 35)), (8, 16, 2, (1, 4, 10, 34)), (32, 32, 1, (2, 1, 12, 20)), (64, 64, 1, (1, 2, 6, 10))):
        x = torch.randn(*shape, ci, generator=g).to(DEV)
        w = (torch.randn(co, ci, 3, 3, 3, generator=g) / (ci * 27) ** 0.5).to(DEV)
        planar = stride == 1 and shape[1] == 1
        full, one = ops.pack_conv3d_weight(w), ops.pack_conv3d_weight(w, use=(stride, planar))
        assert one.images != ops.IMG_ALL and bin(one.images).count("1") == 1
        assert torch.equal(ops.conv3d(x, full, stride=stride), ops.conv3d(x, one, stride=stride))
        other = ops.pack_conv3d_weight(w, use=(stride, not planar)) if stride == 1 else None
        if other is not None and other.images != one.images:
            with pytest.raises(_lib.RcmvsError):
                ops.conv3d(x, other, stride=stride)
    wt = (torch.randn(16, 8, 3, 3, 3, generator=g) / 20).to(DEV)
    x = torch.randn(1, 3, 6, 18, 16, generator=g).to(DEV)
    assert torch.equal(ops.deconv3d(x, ops.pack_conv3d_weight(wt, transposed=True)), ops.deconv3d(x, ops.pack_conv3d_weight(wt, transposed=True, use=(2, False))))


def test_packed_weight_reuse_follows_the_parameter_version():
    """train_ops reuses a parameter's packed image while the parameter is unchanged and re-packs after an in-place update."""
    from rc_mvsnet_amd import train_ops, _lib
    _lib.load()
    train_ops.clear_pack_cache()
    g = torch.Generator().manual_seed(4)
    w = torch.nn.Parameter((torch.randn(8, 16, 3, 3, 3, generator=g) / 20).to(DEV))
    x = torch.randn(1, 4, 9, 20, 16, generator=g).to(DEV)
    a = train_ops._packed(w.detach(), False, 1, False, w)
    assert train_ops._packed(w.detach(), False, 1, False, w) is a                      # same version: the cached blob
    assert train_ops._packed(w.detach(), 2, 1, False, w) is not a                      # another pack mode: its own entry
    y0 = train_ops._conv_raw(x, w.detach(), False, 1, w)
    with torch.no_grad():
        w.mul_(2.0)                                                                   # what an optimizer step does: bumps the version
    b = train_ops._packed(w.detach(), False, 1, False, w)
    assert b is not a
    y1 = train_ops._conv_raw(x, w.detach(), False, 1, w)
    assert float((y1 - 2.0 * y0).abs().max()) <= 1e-5 * float(y1.abs().max())
    train_ops.clear_pack_cache()


def test_pack_cache_follows_data_writes_of_a_legacy_optimizer_and_dies_with_its_parameter():
    """A write through `.data` bumps no version counter.  An optimizer that updates that way (p.data.add_) is still followed, because every
    torch.optim.Optimizer.step() empties the cache (global post-step hook); a bare `.data` write needs clear_pack_cache(), as documented;
    moving the parameter's storage invalidates its entry; and entries do not outlive their parameter."""
    import gc
    from rc_mvsnet_amd import train_ops, _lib
    _lib.load()
    train_ops.clear_pack_cache()
    g = torch.Generator().manual_seed(9)
    w = torch.nn.Parameter((torch.randn(8, 16, 3, 3, 3, generator=g) / 20).to(DEV))
    x = torch.randn(1, 4, 9, 20, 16, generator=g).to(DEV)
    y0 = train_ops._conv_raw(x, w.detach(), False, 1, w)

    class LegacySGD(torch.optim.Optimizer):
        def __init__(self, params):
            super().__init__(params, {})

        def step(self, closure=None):
            for grp in self.param_groups:
                for p in grp["params"]:
                    p.data.mul_(2.0)                                                   # no version bump

    v = w._version
    LegacySGD([w]).step()
    assert w._version == v                                                             # the counter really did not move
    y1 = train_ops._conv_raw(x, w.detach(), False, 1, w)
    assert float((y1 - 2.0 * y0).abs().max()) <= 1e-5 * float(y1.abs().max())         # ... and the new weights were used all the same
    w.data.mul_(0.5)                                                                   # outside any optimizer: stale until told
    assert torch.equal(train_ops._conv_raw(x, w.detach(), False, 1, w), y1)
    train_ops.clear_pack_cache()
    y2 = train_ops._conv_raw(x, w.detach(), False, 1, w)
    assert float((y2 - y0).abs().max()) <= 1e-5 * float(y0.abs().max())
    a = train_ops._packed(w.detach(), False, 1, False, w)
    w.data = w.data.clone()                                                            # same values, other storage (what .to(device) does)
    assert train_ops._packed(w.detach(), False, 1, False, w) is not a
    n = len(train_ops._PACK_CACHE)
    assert n >= 1
    del w, a
    gc.collect()
    assert len(train_ops._PACK_CACHE) == n - 1                                         # the entry went with the parameter
    train_ops.clear_pack_cache()


def test_batchnorm_and_wgrad_scratch_survive_an_aborted_call(monkeypatch):
    """A call that dies between its accumulation launch and the launch that consumes / clears the buffers (OOM with a skip-batch handler,
    KeyboardInterrupt) must not poison the layer's next call: forward statistics, backward sums and the packed weight-gradient buffer."""
    from rc_mvsnet_amd import train_ops, _lib
    from rc_mvsnet_amd.casmvsnet import Conv3d
    _lib.load()
    torch.manual_seed(3)
    blk = Conv3d(16, 8).to(DEV).train()
    x = torch.randn(2, 4, 8, 12, 16, device=DEV)

    def run():
        xx = x.clone().requires_grad_(True)
        for p in blk.parameters():
            p.grad = None
        z = train_ops.conv_bn_relu_train(blk, xx)
        z.square().sum().backward()
        return z.detach().clone(), xx.grad.clone(), blk.conv.weight.grad.clone(), blk.bn.weight.grad.clone()

    run()                                                                              # buffers exist, ping-pong state advanced once
    blk.bn.running_mean.zero_(); blk.bn.running_var.fill_(1.0)
    want = run()
    real_stats, real_reduce, real_wgrad = train_ops.bn_stats, train_ops.bn_bwd_reduce, train_ops.conv3d_wgrad

    class Boom(RuntimeError):
        pass

    def failing(real):
        def f(*a, **k):
            real(*a, **k)                                                              # the accumulation launch is enqueued ...
            raise Boom()                                                               # ... and the call never reaches its finishing launch
        return f

    for name, real in (("bn_stats", real_stats), ("bn_bwd_reduce", real_reduce), ("conv3d_wgrad", real_wgrad)):
        monkeypatch.setattr(train_ops, name, failing(real))
        with pytest.raises(Boom):
            run()
        monkeypatch.setattr(train_ops, name, real)
        blk.bn.running_mean.zero_(); blk.bn.running_var.fill_(1.0)
        got = run()
        for a, b, what in zip(got, want, ("z", "dx", "dw", "dgamma")):
            if what in ("z", "dx"):
                assert torch.equal(a, b), (name, what, float((a - b).abs().max()))
            else:            # sums of atomics: the order, hence the last bits, vary from launch to launch on the GPU; stale sums would be O(1) off
                assert float((a - b).abs().max()) <= 1e-5 * float(b.abs().max()), (name, what, float((a - b).abs().max()))


@pytest.mark.parametrize("C,rows,relu", [(8, 1000, True), (32, 77, False), (64, 513, True)])
def test_fused_batchnorm_forms_equal_the_two_launch_forms(C, rows, relu):
    """rcmvs_bn_norm_fwd / _bwd = finalize + apply in one launch: bit-identical statistics, outputs and gradients; the accumulation
    buffer handed over as `clear` comes back zero and the one that was read is left intact."""
    import ctypes
    from rc_mvsnet_amd import train_ops, _lib
    lib = _lib.load()
    from rc_mvsnet_amd.ops import _stream
    ptr = lambda t: ctypes.c_void_p(t.data_ptr())
    g = torch.Generator().manual_seed(C + rows)
    y = (torch.randn(rows, C, generator=g) * 3 + 1).to(DEV)
    res = torch.randn(rows, C, generator=g).to(DEV)
    dz = torch.randn(rows, C, generator=g).to(DEV)
    gamma, beta = (0.5 + torch.rand(C, generator=g)).to(DEV), (0.1 * torch.randn(C, generator=g)).to(DEV)
    f64 = lambda n: torch.zeros(n, device=DEV, dtype=torch.float64)
    # ---- two-launch reference
    s_a = f64(2 * C + 1); train_ops.bn_stats(y, s_a)
    keep = s_a.clone()
    cnt_a = f64(1); st_a = torch.empty(5, C, device=DEV)
    rm_a, rv_a = torch.zeros(C, device=DEV), torch.ones(C, device=DEV)
    _lib.check(lib.rcmvs_bn_finalize(ptr(s_a), ptr(cnt_a), ptr(gamma), ptr(beta), 1e-5, 0.1, ptr(st_a[0]), ptr(st_a[1]), ptr(st_a[2]), ptr(st_a[3]),
                                     ptr(st_a[4]), ptr(rm_a), ptr(rv_a), C, _stream()), "bn_finalize")
    assert float(s_a.abs().max()) == 0.0 and float(cnt_a[0]) == rows                   # consumed and cleared, the row count handed on
    z_a = train_ops.scale_shift_relu(y, st_a[3], st_a[4], res, relu)
    # ---- fused
    s_b = keep.clone(); dirty = torch.full((2 * C + 1,), 7.0, device=DEV, dtype=torch.float64)
    cnt_b = f64(1); st_b = torch.empty(5, C, device=DEV); z_b = torch.empty_like(y)
    rm_b, rv_b = torch.zeros(C, device=DEV), torch.ones(C, device=DEV)
    _lib.check(lib.rcmvs_bn_norm_fwd(ptr(y), ptr(s_b), ptr(dirty), ptr(gamma), ptr(beta), 1e-5, 0.1, ptr(st_b), ptr(cnt_b), ptr(rm_b), ptr(rv_b),
                                     ptr(res), ptr(z_b), rows, C, int(relu), _stream()), "bn_norm_fwd")
    assert torch.equal(st_a, st_b) and torch.equal(z_a, z_b) and torch.equal(rm_a, rm_b) and torch.equal(rv_a, rv_b)
    assert float(dirty.abs().max()) == 0.0 and torch.equal(s_b, keep) and float(cnt_b[0]) == rows
    # ---- backward
    l_a = f64(2 * C); train_ops.bn_bwd_reduce(y, dz, st_a[3], st_a[4], st_a[0], st_a[2], l_a, relu)
    keep = l_a.clone()
    out_a = torch.empty(4, C, device=DEV)
    _lib.check(lib.rcmvs_bn_bwd_finalize(ptr(l_a), ptr(keep), ptr(cnt_a), ptr(out_a[0]), ptr(out_a[1]), ptr(out_a[2]), C, _stream()), "bn_bwd_finalize")
    dy_a = train_ops.bn_bwd_apply(y, dz, st_a[3], st_a[4], st_a[0], st_a[2], out_a[2:].reshape(-1), relu)
    l_b = keep.clone(); dirty = torch.full((2 * C,), 7.0, device=DEV, dtype=torch.float64)
    out_b = torch.empty(2, C, device=DEV); dy_b = torch.empty_like(y)
    _lib.check(lib.rcmvs_bn_norm_bwd(ptr(y), ptr(dz), ptr(st_b), ptr(l_b), ptr(l_b), ptr(cnt_b), ptr(dirty), ptr(out_b[0]), ptr(out_b[1]), ptr(dy_b),
                                     rows, C, int(relu), _stream()), "bn_norm_bwd")
    assert torch.equal(out_a[:2], out_b) and torch.equal(dy_a, dy_b) and float(dirty.abs().max()) == 0.0 and torch.equal(l_b, keep)


def test_weight_gradient_finish_permutes_and_clears():
    """rcmvs_wgrad_finish: packed [27][P][Q] -> (Q, Pk, 27) with the padding rows dropped, packed buffer zero afterwards; the layer-level
    path (persistent accumulation buffer) gives the same gradient twice in a row."""
    from rc_mvsnet_amd import train_ops, _lib
    from rc_mvsnet_amd.ops import _chk, _stream
    lib = _lib.load()
    g = torch.Generator().manual_seed(9)
    P, Q, Pk = 12, 8, 9
    packed = torch.randn(27, P, Q, generator=g).to(DEV)
    want = packed[:, :Pk].permute(2, 1, 0).contiguous()
    out = torch.empty(Q, Pk, 27, device=DEV)
    _lib.check(lib.rcmvs_wgrad_finish(_chk(packed, "packed"), _chk(out, "out"), P, Q, Pk, _stream()), "wgrad_finish")
    assert torch.equal(out, want) and float(packed.abs().max()) == 0.0
    x = torch.randn(1, 3, 7, 22, 16, generator=g).to(DEV)
    dy = torch.randn(1, 3, 7, 22, 8, generator=g).to(DEV)
    ref = train_ops.conv3d_wgrad(x, dy, 1).permute(2, 1, 0).reshape(8, 16, 3, 3, 3)
    for _ in range(2):                                                                  # second call: the buffer the first one cleared
        got = train_ops._conv_wgrad(x, dy, (8, 16, 3, 3, 3), False, 1)
        assert float((got - ref).abs().max()) <= 1e-5 * float(ref.abs().max())          # (atomic summation order differs between launches)
