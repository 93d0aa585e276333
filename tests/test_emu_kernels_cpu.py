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
    reference-order kernel and the LDS-staged variants (per-wave bounding boxes through DPP row shifts + readlane), and against
    the oracle's variance volume."""
    from oracle import warp
    from rc_mvsnet_amd import ops
    g = torch.Generator().manual_seed(C + V)
    feats = torch.randn(2, V, h, w, C, generator=g)
    pm = synthetic.proj_matrices(2, V, h * 4, w * 4)["stage1"]
    rot, trans = ops.compose_homography(pm)
    planes = torch.stack((425.0 + 100.0 * torch.rand(2, h, w, generator=g), 2.0 + 8.0 * torch.rand(2, h, w, generator=g)), dim=-1).contiguous()
    try:
        emu.rcmvs_debug_k1_variant(2)
        vref = ops.warp_variance(feats, rot, trans, planes, D)
        outs = {}
        for var in (0, 4, 6):
            emu.rcmvs_debug_k1_variant(var)
            outs[var] = ops.warp_variance(feats, rot, trans, planes, D)
    finally:
        emu.rcmvs_debug_k1_variant(0)
    for var, v in outs.items():
        assert float((v - vref).abs().max()) <= 2e-6 * max(1.0, float(vref.abs().max())), var
    samples = planes[..., 0].unsqueeze(1) + planes[..., 1].unsqueeze(1) * torch.arange(D, dtype=torch.float32).view(1, D, 1, 1)
    want = warp.variance_volume([feats[:, v].permute(0, 3, 1, 2) for v in range(V)], pm, samples)        # (B,C,D,h,w)
    got = outs[0].permute(0, 4, 1, 2, 3)
    assert float((got - want).abs().max()) <= 1e-4 * max(1.0, float(want.abs().max()))


def test_results_do_not_depend_on_the_thread_schedule(emu):
    """Missing-barrier detector: between synchronisation points the emulation may run a block's threads in any order; ascending,
    descending and wave-reversed schedules must give bit-identical results for kernels without float atomics (the whole inference
    cascade, the fused FPN level, the LDS-staged K1 variant, ordered compaction) and equal results up to summation order for the
    ones that accumulate with atomics (loss sums, K1 backward)."""
    from rc_mvsnet_amd import fusion, losses, ops
    from rc_mvsnet_amd.casmvsnet import CascadeMVSNet_eval
    model = CascadeMVSNet_eval(ndepths=[8, 8, 8], depth_interals_ratio=[4, 2, 1])
    model.load_state_dict(synthetic.cascade_state_dict(0), strict=True)
    model.eval()
    imgs, pm, dv = synthetic.cascade_inputs(1, 3, 32, 64, 0)
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
            emu.rcmvs_debug_k1_variant(6)
            staged = ops.warp_variance(feats, rot, trans, planes, 8)
            emu.rcmvs_debug_k1_variant(0)
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
        for order in (1, 2):
            emu.rcmvs_emu_set_order(order)
            exact, approx = run()
            for k in exact0:
                assert torch.equal(exact[k], exact0[k]), (order, k)
            for k in approx0:
                assert float((approx[k] - approx0[k]).abs().max()) <= 1e-5 * max(1e-6, float(approx0[k].abs().max())), (order, k)
    finally:
        emu.rcmvs_emu_set_order(0)
        emu.rcmvs_debug_k1_variant(0)
        emu.rcmvs_debug_k1_ps_config(0, 0, 0)


@pytest.mark.parametrize("C,D,h,w,V", [(32, 16, 12, 21, 3), (16, 8, 17, 30, 3), (8, 12, 16, 40, 2), (8, 10, 9, 140, 3), (32, 5, 6, 9, 2)])
def test_k1_pipelined_staged_variant_on_emulated_kernels(C, D, h, w, V, emu):
    """Debug variants 8 / 9 (persistent over plane chunks, double-buffered tap tables and windows, direct-to-LDS loads): the
    exact build must equal the reference-order kernel bit for bit -- ragged plane counts, ragged tiles, windows that do not fit
    the LDS budget (wide maps: global fallback), one and two source views -- under every thread schedule."""
    from rc_mvsnet_amd import _lib, ops
    g = torch.Generator().manual_seed(C + V + D)
    feats = torch.randn(2, V, h, w, C, generator=g)
    pm = synthetic.proj_matrices(2, V, h * 4, w * 4)["stage1"]
    rot, trans = ops.compose_homography(pm)
    planes = torch.stack((425.0 + 100.0 * torch.rand(2, h, w, generator=g), 2.0 + 8.0 * torch.rand(2, h, w, generator=g)), dim=-1).contiguous()
    try:
        emu.rcmvs_debug_k1_variant(2)
        vref = ops.warp_variance(feats, rot, trans, planes, D)
        for order in (0, 1, 2):
            emu.rcmvs_emu_set_order(order)
            for var in (8, 10, 12):                                                   # windows by direct-to-LDS loads / held in registers / static LDS sets
                emu.rcmvs_debug_k1_variant(var)
                assert torch.equal(ops.warp_variance(feats, rot, trans, planes, D), vref), (var, order)
        emu.rcmvs_emu_set_order(0)
        knobs = ((2, 0, 0), (8, 64, 16), (4, 16, 0), (2, 64, 32), (4, 0, 16)) if (C, V) in ((32, 3), (8, 2)) else ((4, 16, 16),)
        for dkb, ptex, pad in knobs:                                                       # other chunk depths; a budget small enough to force the global fallback
            emu.rcmvs_debug_k1_ps_config(dkb, ptex, pad)
            for var in (8, 10):
                emu.rcmvs_debug_k1_variant(var)
                assert torch.equal(ops.warp_variance(feats, rot, trans, planes, D), vref), (var, dkb, ptex, pad)
        emu.rcmvs_debug_k1_ps_config(0, 0, 0)
        for dkb, ptex in ((2, 0), (4, 0), (4, 72)):                                   # static-set form: compile-time budgets per chunk depth
            emu.rcmvs_debug_k1_ps_config(dkb, ptex, 0)
            emu.rcmvs_debug_k1_variant(12)
            assert torch.equal(ops.warp_variance(feats, rot, trans, planes, D), vref), (dkb, ptex)
        emu.rcmvs_debug_k1_ps_config(0, 0, 0)
        for var in (9, 11, 13):
            emu.rcmvs_debug_k1_variant(var)
            v9 = ops.warp_variance(feats, rot, trans, planes, D)
            assert float((v9 - vref).abs().max()) <= 2e-6 * max(1.0, float(vref.abs().max())), var
        emu.rcmvs_debug_k1_variant(8)
        with pytest.raises(_lib.RcmvsError):
            ops.warp_variance(torch.randn(1, 4, h, w, C), rot[:1].repeat(1, 2, 1)[:, :3], trans[:1].repeat(1, 2, 1)[:, :3], planes[:1], D)
    finally:
        emu.rcmvs_emu_set_order(0)
        emu.rcmvs_debug_k1_variant(0)
        emu.rcmvs_debug_k1_ps_config(0, 0, 0)


@pytest.mark.parametrize("C,interval", [(32, 40.0), (16, 60.0), (8, 400.0)])
def test_k1_pipelined_variants_window_overflow_path(C, interval, emu):
    """Plane spacing so wide that a chunk's source window exceeds the LDS budget: the block-uniform mixed body (global gathers for
    the view that does not fit) of the pipelined variants, including the static-set form whose budget is a compile-time constant.
    The test first shows, with the oracle's coordinates, that such windows do occur for these inputs."""
    from oracle import warp
    from rc_mvsnet_amd import ops
    g = torch.Generator().manual_seed(C)
    V, D, h, w = 3, 8, 8, 200
    budget, tile_w, chunk = {32: (112, 8, 4), 16: (160, 16, 4), 8: (320, 32, 2)}[C]          # the static form's PTEX, TW, DKB
    feats = torch.randn(1, V, h, w, C, generator=g)
    pm = synthetic.proj_matrices(1, V, h * 4, w * 4)["stage1"]
    rot, trans = ops.compose_homography(pm)
    planes = torch.stack((430.0 + 5.0 * torch.rand(1, h, w, generator=g), interval + 5.0 * torch.rand(1, h, w, generator=g)), dim=-1).contiguous()
    samples = planes[..., 0].unsqueeze(1) + planes[..., 1].unsqueeze(1) * torch.arange(D, dtype=torch.float32).view(1, D, 1, 1)
    largest = 0
    for v in (1, 2):
        r, t = warp.compose_homography(pm[:, v], pm[:, 0])
        ix, iy = warp.warp_coords(r, t, samples, h, w)
        x0, y0 = ix[0].floor().clamp(0, w - 1), iy[0].floor().clamp(0, h - 1)
        for k0 in range(0, D, chunk):
            for tx in range(0, w, tile_w):
                xs, ys = x0[k0:k0 + chunk, 0:4, tx:tx + tile_w], y0[k0:k0 + chunk, 0:4, tx:tx + tile_w]
                largest = max(largest, int((xs.max() - xs.min() + 2) * (ys.max() - ys.min() + 2)))
    assert largest > budget, (largest, budget)
    try:
        emu.rcmvs_debug_k1_variant(2)
        vref = ops.warp_variance(feats, rot, trans, planes, D)
        assert float(vref.abs().max()) > 0
        for var in (8, 10, 12):
            emu.rcmvs_debug_k1_variant(var)
            assert torch.equal(ops.warp_variance(feats, rot, trans, planes, D), vref), var
    finally:
        emu.rcmvs_debug_k1_variant(0)
