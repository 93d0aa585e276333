"""The HIP kernels themselves on the CPU: rc_mvsnet_amd/csrc/*.hip compiled against the emulation of tests/emu (fibers for
threads, rendezvous for __syncthreads / wave collectives) and driven through the package's own Python path on CPU tensors,
checked against the oracle and the reference's goldens.  Logic only -- indexing, barriers, collectives, launch geometry; the
`-m gpu` tests remain the parity tests proper (the emulation's v_rcp and MFMA summation order are not the hardware's)."""
import os

import numpy as np
import pytest
import torch

from conftest import load_golden
from rc_mvsnet_amd import synthetic

HERE = os.path.dirname(os.path.abspath(__file__))


def test_unsup_loss_multi_stage_on_emulated_kernels(emu):
    from rc_mvsnet_amd import losses
    G = np.load(os.path.join(HERE, "golden", "unsup_loss.npz"))
    for tag in ("a", "b"):
        B, V, H, W, seed = [int(x) for x in G[tag + ":dims"]]
        imgs, cams = synthetic.images(B, V, H, W, seed), synthetic.proj_matrices(B, V, H, W)
        inputs = {k: {"depth": torch.tensor(G[f"{tag}:depth:{k}"]).requires_grad_(True)} for k in ("stage1", "stage2", "stage3")}
        total, scalars = losses.UnsupLossMultiStage()(inputs, imgs, cams, dlossw=[0.5, 1.0, 2.0])
        total.backward()
        assert abs(float(total.detach()) - float(G[tag + ":total"])) <= 2e-5 * abs(float(G[tag + ":total"]))
        for k in inputs:
            want = torch.tensor(G[f"{tag}:grad:{k}"])
            err = (inputs[k]["depth"].grad - want).abs()
            assert float(err.median()) <= 1e-5 * float(want.abs().max()) and float((err > 1e-3 * float(want.abs().max())).float().mean()) <= 5e-3


def test_fused_fpn_level_and_conv2d_on_emulated_kernels(emu):
    from rc_mvsnet_amd import ops
    g = torch.Generator().manual_seed(0)
    lat, up = torch.randn(2, 36, 44, 8, generator=g), torch.randn(2, 18, 22, 32, generator=g)
    w_in_t, b_in = 0.3 * torch.randn(32, 8, 1, 1, generator=g), 0.1 * torch.randn(32, generator=g)
    w_out_t = 0.1 * torch.randn(8, 32, 3, 3, generator=g)
    w_in, w_out = ops.pack_conv2d_weight(w_in_t), ops.pack_conv2d_weight(w_out_t)
    two = ops.conv2d(ops.conv2d(lat, w_in, None, b_in, up_add=up), w_out)
    one = ops.fpn_out_fused(lat, up, w_in, b_in, w_out)
    assert torch.equal(one, two)
    F = torch.nn.functional
    intra = F.interpolate(up.permute(0, 3, 1, 2), scale_factor=2, mode="nearest") + F.conv2d(lat.permute(0, 3, 1, 2), w_in_t, b_in)
    want = F.conv2d(intra, w_out_t, padding=1).permute(0, 2, 3, 1)
    assert float((one - want).abs().max()) < 2e-5


def test_cascade_eval_forward_on_emulated_kernels(emu):
    """CascadeMVSNet_eval.forward through every inference kernel (FeatureNet 2-D convs incl. the MFMA / space-to-depth layers,
    planes, homography, K1, the 3-D conv family, depth head) on the emulation, against the oracle -- __graft_entry__.smoke()'s
    check without a GPU."""
    from oracle import cascade
    from rc_mvsnet_amd.casmvsnet import CascadeMVSNet_eval
    sd = synthetic.cascade_state_dict(0)
    nd, ratios = (8, 8, 8), (4, 2, 1)
    model = CascadeMVSNet_eval(ndepths=list(nd), depth_interals_ratio=list(ratios))
    model.load_state_dict(sd, strict=True)
    model.eval()
    imgs, pm, dv = synthetic.cascade_inputs(1, 3, 32, 64, 0)
    with torch.no_grad():
        out = model(imgs, pm, dv)
    ref = cascade.forward_eval(imgs, pm, dv, sd, nd, ratios, impl="spec")
    rng = float(dv[0, -1] - dv[0, 0])
    for key in ("stage1", "stage2", "stage3"):
        err = float((out[key]["depth"] - ref[key]["depth"]).abs().mean()) / rng
        assert err < 1e-4, (key, err)
    assert float((out["photometric_confidence"] - ref["photometric_confidence"]).abs().max()) < 5e-3


@pytest.mark.parametrize("C,D,h,w,V", [(32, 16, 12, 21, 3), (16, 8, 17, 30, 4), (8, 12, 16, 40, 2), (8, 8, 12, 24, 7), (16, 8, 10, 14, 5)])
def test_k1_variants_on_emulated_kernels(C, D, h, w, V, emu):
    """K1 forward: production two-phase kernel (compile-time 2 / 4 / 6 source views and the general path) against the
    reference-order kernel (and its FMA-contracted build), and against the oracle's variance volume."""
    from oracle import warp
    from rc_mvsnet_amd import ops
    g = torch.Generator().manual_seed(C + V)
    feats = torch.randn(2, V, h, w, C, generator=g)
    pm = synthetic.proj_matrices(2, V, h * 4, w * 4)["stage1"]
    rot, trans = ops.compose_homography(pm)
    planes = torch.stack((425.0 + 100.0 * torch.rand(2, h, w, generator=g), 2.0 + 8.0 * torch.rand(2, h, w, generator=g)), dim=-1).contiguous()
    vref = ops.warp_variance(feats, rot, trans, planes, D, variant=2)
    outs = {var: ops.warp_variance(feats, rot, trans, planes, D, variant=var) for var in (0, 1)}
    for var, v in outs.items():
        assert float((v - vref).abs().max()) <= 2e-6 * max(1.0, float(vref.abs().max())), var
    samples = planes[..., 0].unsqueeze(1) + planes[..., 1].unsqueeze(1) * torch.arange(D, dtype=torch.float32).view(1, D, 1, 1)
    want = warp.variance_volume([feats[:, v].permute(0, 3, 1, 2) for v in range(V)], pm, samples)        # (B,C,D,h,w)
    got = outs[0].permute(0, 4, 1, 2, 3)
    assert float((got - want).abs().max()) <= 1e-4 * max(1.0, float(want.abs().max()))


@pytest.mark.parametrize("C,D,h,w,smooth", [(32, 8, 20, 37, True), (16, 7, 33, 50, True), (8, 12, 24, 96, True), (32, 5, 9, 13, False),
                                            (16, 8, 6, 70, False), (8, 4, 5, 3, False)])
def test_k1_window_form_on_emulated_kernels(C, D, h, w, smooth, emu):
    """K1, LDS-window form (csrc/k1_win.h; two source views): windows loaded ahead of the coordinate phase (5, what the
    uniform-planes hint launches) and after the fit test (6) against the reference-order kernel -- identical sampling positions, FMA
    blend -- on plane tables whose tiles fit the windows (smooth: the window path must really run) and on rough ones (per-tile
    fallback to gathers, ragged edges, D not a multiple of the plane chunk, images smaller than a tile)."""
    from rc_mvsnet_amd import ops
    g = torch.Generator().manual_seed(C + h)
    B, V = 2, 3
    feats = torch.randn(B, V, h, w, C, generator=g)
    pm = synthetic.proj_matrices(B, V, h * 4, w * 4)["stage1"]
    rot, trans = ops.compose_homography(pm)
    if smooth:
        planes = torch.stack((500.0 + 3.0 * torch.rand(B, h, w, generator=g), torch.full((B, h, w), 5.0)), dim=-1).contiguous()
    else:
        planes = torch.stack((300.0 + 600.0 * torch.rand(B, h, w, generator=g), 2.0 + 40.0 * torch.rand(B, h, w, generator=g)), dim=-1).contiguous()
    vref = ops.warp_variance(feats, rot, trans, planes, D, variant=2)
    tol = 2e-6 * max(1.0, float(vref.abs().max()))
    assert float((ops.warp_variance(feats, rot, trans, planes, D, variant=7) - vref).abs().max()) <= tol      # plane-pipelined gather form
    for var in (5, 6):
        v, blocks, on_window = ops.warp_variance_win(feats, rot, trans, planes, D, variant=var)
        assert float((v - vref).abs().max()) <= tol, var
        assert blocks > 0 and (on_window == blocks if smooth else on_window <= blocks), (var, blocks, on_window)
        assert torch.equal(v, ops.warp_variance(feats, rot, trans, planes, D, variant=var))
    hinted = ops.warp_variance(feats, rot, trans, planes, D, uniform_planes=True)
    assert torch.equal(hinted, ops.warp_variance(feats, rot, trans, planes, D, variant=5))
    assert torch.equal(ops.warp_variance(feats, rot, trans, planes, D), ops.warp_variance(feats, rot, trans, planes, D, variant=7 if C == 8 else 0))
    with pytest.raises(Exception):
        ops.warp_variance(feats[:, :2].contiguous(), rot[:, :1].contiguous(), trans[:, :1].contiguous(), planes, D, variant=5)   # V = 2: not built
    # ... but the hint itself is only a hint: other view counts take the gather kernel
    v2 = ops.warp_variance(feats[:, :2].contiguous(), rot[:, :1].contiguous(), trans[:, :1].contiguous(), planes, D, uniform_planes=True)
    assert torch.equal(v2, ops.warp_variance(feats[:, :2].contiguous(), rot[:, :1].contiguous(), trans[:, :1].contiguous(), planes, D))


@pytest.mark.parametrize("seed", [0, 1, 2])
def test_k1_forms_on_random_shapes_and_plane_tables(seed, emu):
    """Seeded fuzz of the three-view K1 forms (window 5 / 6, plane-pipelined 7) against the reference-order kernel: ragged image sizes down to
    2 x 2, 1-19 planes, 1-3 batch items, and plane tables that are smooth, rough, uniform, or cross z = 0 (taps dropped, non-finite
    positions): same values to 2e-6 of the range and the same finite / non-finite pattern."""
    import random
    from rc_mvsnet_amd import ops
    rnd = random.Random(seed)
    for n in range(10):
        C, B, D = rnd.choice([8, 16, 32]), rnd.choice([1, 1, 2, 3]), rnd.randint(1, 19)
        h, w = rnd.randint(2, 40), rnd.randint(2, 90)
        g = torch.Generator().manual_seed(1000 * seed + n)
        feats = torch.randn(B, 3, h, w, C, generator=g)
        rot, trans = ops.compose_homography(synthetic.proj_matrices(B, 3, h * 4, w * 4)["stage1"])
        mode = rnd.choice(["smooth", "rough", "uniform", "behind"])
        if mode == "smooth":
            planes = torch.stack((500.0 + 30.0 * torch.rand(B, h, w, generator=g), torch.full((B, h, w), 5.0)), -1)
        elif mode == "uniform":
            planes = torch.stack((torch.full((B, h, w), 430.0), torch.full((B, h, w), 10.6)), -1)
        elif mode == "behind":
            planes = torch.stack((-200.0 + 400.0 * torch.rand(B, h, w, generator=g), 40.0 * torch.randn(B, h, w, generator=g)), -1)
        else:
            planes = torch.stack((300.0 + 600.0 * torch.rand(B, h, w, generator=g), 2.0 + 40.0 * torch.rand(B, h, w, generator=g)), -1)
        planes = planes.contiguous()
        vref = ops.warp_variance(feats, rot, trans, planes, D, variant=2)
        fin = torch.isfinite(vref)
        tol = 2e-6 * max(1.0, float(vref[fin].abs().max()) if bool(fin.any()) else 1.0)
        for var in (5, 6, 7):
            v = ops.warp_variance(feats, rot, trans, planes, D, variant=var)
            assert torch.equal(torch.isfinite(v), fin), (C, B, D, h, w, mode, var)
            assert float((torch.nan_to_num(v) - torch.nan_to_num(vref)).abs().max()) <= tol, (C, B, D, h, w, mode, var)


def test_results_do_not_depend_on_the_thread_schedule(emu):
    """Missing-barrier detector: between synchronisation points the emulation may run a block's threads in any order; ascending,
    descending and wave-reversed schedules must give bit-identical results for kernels without float atomics (the whole inference
    cascade, the fused FPN level, K1, ordered compaction) and equal results up to summation order for the
    ones that accumulate with atomics (loss sums, K1 backward)."""
    from rc_mvsnet_amd import fusion, losses, ops
    from rc_mvsnet_amd.casmvsnet import CascadeMVSNet_eval
    model = CascadeMVSNet_eval(ndepths=[8, 8, 8], depth_interals_ratio=[4, 2, 1])
    model.load_state_dict(synthetic.cascade_state_dict(0), strict=True)
    model.eval()
    imgs, pm, dv = synthetic.cascade_inputs(1, 3, 32, 32, 0)
    g = torch.Generator().manual_seed(5)
    feats = torch.randn(1, 3, 12, 20, 16, generator=g)
    rot, trans = ops.compose_homography(synthetic.proj_matrices(1, 3, 48, 80)["stage1"])
    planes = torch.stack((425.0 + 100.0 * torch.rand(1, 12, 20, generator=g), 2.0 + 8.0 * torch.rand(1, 12, 20, generator=g)), dim=-1).contiguous()
    mask = (torch.rand(37, 53, generator=g) < 0.4).to(torch.uint8)
    xyz = torch.randn(37, 53, 3, generator=g)
    gvar = torch.randn(1, 8, 12, 20, 16, generator=g)
    G = np.load(os.path.join(HERE, "golden", "unsup_loss.npz"))
    B, V, H, W, seed = [int(x) for x in G["a:dims"]]
    limgs, lcams = synthetic.images(B, V, H, W, seed), synthetic.proj_matrices(B, V, H, W)

    def run():
        with torch.no_grad():
            out = model(imgs, pm, dv)
            staged = ops.warp_variance(feats, rot, trans, planes, 8)
            pts, _ = fusion.compact_points(mask, xyz)
            gfe = ops.warp_variance_bwd(feats, rot, trans, planes, gvar, None)
        inputs = {k: {"depth": torch.tensor(G[f"a:depth:{k}"]).requires_grad_(True)} for k in ("stage1", "stage2", "stage3")}
        total, _ = losses.UnsupLossMultiStage()(inputs, limgs, lcams, dlossw=[0.5, 1.0, 2.0])
        total.backward()
        exact = {"depth": out["depth"], "conf": out["photometric_confidence"], "staged": staged, "pts": pts}
        approx = {"k1_bwd": gfe, "loss": total.detach().reshape(1), "loss_grad": inputs["stage3"]["depth"].grad}
        return exact, approx

    try:
        exact0, approx0 = run()
        for order in ((1, 2) if os.environ.get("RCMVS_EMU_FULL", "0") == "1" else (1,)):      # (2 = wave-reversed: with RCMVS_EMU_FULL=1)
            emu.rcmvs_emu_set_order(order)
            exact, approx = run()
            for k in exact0:
                assert torch.equal(exact[k], exact0[k]), (order, k)
            for k in approx0:
                assert float((approx[k] - approx0[k]).abs().max()) <= 1e-5 * max(1e-6, float(approx0[k].abs().max())), (order, k)
    finally:
        emu.rcmvs_emu_set_order(0)
