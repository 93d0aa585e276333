"""Re-run the GPU parity tests on the CPU emulation of the kernels (tests/emu): the very same test functions as
`pytest -m gpu` (cloned from tests/test_gpu_*.py; the module-level gpu mark does not travel with them), with their DEV switched
to "cpu" and the package routed to librcmvs_emu.so.  The default CPU run takes the tests the emulation finishes in seconds;
RCMVS_EMU_FULL=1 adds the minute-long ones (rendering forward / training graphs, wide conv blocks).  Full-size property tests
stay GPU-only."""
import functools
import os
import types

import pytest
import torch

import test_gpu_dataset as GD
import test_gpu_fusion as GF
import test_gpu_losses as GL
import test_gpu_parity as GP
import test_gpu_render as GR
import test_gpu_train as GT

MODULES = (GP, GR, GT, GL, GF, GD)


@pytest.fixture(autouse=True)
def _on_the_emulation(emu, monkeypatch):
    for m in MODULES:
        if hasattr(m, "DEV"):
            monkeypatch.setattr(m, "DEV", "cpu")
    from rc_mvsnet_amd import _lib
    monkeypatch.setattr(_lib, "load", lambda: emu)
    monkeypatch.setattr(torch.cuda, "synchronize", lambda *a, **k: None)
    from rc_mvsnet_amd import fusion, mvs_dataset
    for name in ("check_geometric_consistency", "filter_depth", "filter_depth_tanks"):        # their default device is cuda:0
        monkeypatch.setattr(fusion, name, functools.partial(getattr(fusion, name), device="cpu"))
    prepare = mvs_dataset.prepare_image
    monkeypatch.setattr(mvs_dataset, "prepare_image", lambda img, hw, device, **kw: prepare(img, hw, "cpu", **kw))
    yield


@pytest.fixture
def hip(emu):
    from rc_mvsnet_amd import ops
    return ops


FULL = os.environ.get("RCMVS_EMU_FULL", "0") == "1"


def reuse(module, name, slow=False):
    """A copy of a GPU test function (same code, same globals, same parametrisation) that this module can collect and mark
    without touching the original."""
    f = getattr(module, name)
    g = types.FunctionType(f.__code__, f.__globals__, name, f.__defaults__, f.__closure__)
    g.__dict__.update(f.__dict__)
    g.pytestmark = list(getattr(f, "pytestmark", []))
    if slow:
        g.pytestmark.append(pytest.mark.skipif(not FULL, reason="minutes on the emulation: set RCMVS_EMU_FULL=1").mark)
    g.__doc__ = f.__doc__
    return g


FAST = {
    GP: ["test_layout_roundtrip", "test_compose_homography", "test_hypothesis_planes_vs_golden", "test_hypothesis_planes_stage1_exact",
         "test_warp_variance_vs_oracle", "test_resize_rgb_cl_is_torch_bilinear", "test_conv2d_pair_vs_fp64", "test_conv2d_stem_vs_fp64", "test_conv2d_tile_vs_fp64", "test_warp_variance_golden_fixture", "test_warp_variance_backward_vs_oracle_autograd",
         "test_conv3d_vs_oracle", "test_conv3d_x3_item_schedule_is_bit_exact", "test_conv3d_x3_planar_vs_fp64", "test_conv3d_x3h_vs_fp64", "test_conv0_stream_cuts_are_bit_exact", "test_conv1_stream_cuts_are_bit_exact", "test_conv11_prob_vs_fp64", "test_prob_conv_marching_kernel", "test_prob_conv_z_chunk_does_not_change_the_result", "test_conv2d_s2d_is_the_5x5_stride2_layer", "test_conv3d_lds_halo_kernel", "test_deconv3d_vs_oracle", "test_conv3d_golden_and_linearity",
         "test_costreg_vs_golden", "test_conv2d_vs_torch_cpu", "test_depth_head_vs_oracle", "test_depth_head_forms_are_bit_identical", "test_depth_head_matrix_core_form_vs_fp32_form", "test_depth_head_golden",
         "test_fpn_out_fused_is_bit_identical", "test_fpn_out_folded_matches_the_unfused_path", "test_conv1x1_matrix_core_form", "test_feature_output_convs_keep_the_variance_bound", "test_first_layer_reads_the_planar_images_itself", "test_standalone_blocks_vs_reference_golden",
         "test_deconv2d_fuse_and_unet_pyramid_vs_reference_golden", "test_standalone_conv3d_block_trains_like_torch",
         "test_depthnet_on_its_own_vs_reference_golden", "test_execution_plans_follow_their_parameters"],
    GR: ["test_resize_planes", "test_gu_sampler_vs_oracle", "test_nerf_mlp_vs_oracle", "test_rendernet_forward_on_its_own_vs_reference_golden", "test_composite_vs_oracle"],
    GT: ["test_prob_depth_head_backward", "test_prob_conv_weight_gradient_marching_kernel", "test_conv3d_weight_gradient_cout8_paired_columns",
         "test_selective_weight_pack_matches_full_blob_and_is_checked", "test_packed_weight_reuse_follows_the_parameter_version", "test_pack_cache_follows_data_writes_of_a_legacy_optimizer_and_dies_with_its_parameter", "test_batchnorm_and_wgrad_scratch_survive_an_aborted_call",
         "test_fused_batchnorm_forms_equal_the_two_launch_forms", "test_weight_gradient_finish_permutes_and_clears"],
    GL: ["test_unsup_loss_multi_stage_matches_reference", "test_inverse_warping_matches_reference", "test_aug_loss_and_sl1_match_reference",
         "test_unsup_loss_argument_checks"],
    GF: ["test_check_geometric_consistency_matches_reference", "test_filter_depth_matches_reference", "test_fuse_view_argument_checks",
         "test_filter_depth_tanks_matches_reference"],
    GD: ["test_tanks_loader_items_match_reference"],          # the DTU twin asserts `.is_cuda`; tests/test_dataset_cpu.py covers it
}
SLOW = {
    GP: ["test_train_variant_volume_feature", "test_warp_variance_variants_agree", "test_feature_net_vs_oracle", "test_feature_net_fused_kernels_match_the_one_layer_launches",
         "test_cascade_batch_two_equals_two_singles", "test_conv3d_x3_vs_fp64", "test_conv3d_x3_strided_vs_fp64",
         "test_cascade_on_the_unet_pyramid_vs_reference_golden"],
    GR: ["test_neural_volume_vs_golden", "test_render_forward_vs_reference_golden", "test_render_forward_five_view_extension_vs_reference_golden"],
    GT: ["test_conv_bn_relu_block_forward_backward", "test_neural_volume_net_train_native_vs_delegated",
         "test_renderer_train_native_vs_delegated", "test_featurenet_train_native_vs_delegated",
         "test_cascade_train_native_vs_delegated_gradients", "test_hip_training_path_vs_reference_gradients",
         "test_sync_batchnorm_branch_with_simulated_replica"],
    GF: ["test_compact_points_is_ordered_boolean_indexing"],
}
for _table, _slow in ((FAST, False), (SLOW, True)):
    for _mod, _names in _table.items():
        for _n in _names:
            globals()[_n] = reuse(_mod, _n, _slow)


# ---- the same functions with the parameter sets the emulation finishes quickly (the GPU runs all of them)
@pytest.mark.parametrize("Ci,Co,mode", [(8, 16, "s2"), (16, 32, "s2"), (32, 64, "s2"), (32, 32, "s1"), (64, 64, "s1"), (64, 32, "t2"), (32, 16, "t2")])
def test_conv3d_mfma_matches_direct_small(hip, Ci, Co, mode):
    GP.test_conv3d_mfma_matches_direct(hip, Ci, Co, mode, False)


@pytest.mark.parametrize("name", ["cascade_c1", pytest.param("cascade_small", marks=pytest.mark.skipif(not FULL, reason="RCMVS_EMU_FULL=1"))])
def test_cascade_vs_reference_golden_small(hip, name):
    GP.test_cascade_vs_reference_golden(hip, name)


@pytest.mark.skipif(not FULL, reason="a minute on the emulation: RCMVS_EMU_FULL=1")
def test_cascade_fp16_pair_form_small(hip, monkeypatch):
    GP.test_cascade_fp16_pair_form_vs_reference_golden(hip, monkeypatch, "cascade_c1", 1e-4)


@pytest.mark.parametrize("ci,co,stride,transposed", [(8, 8, 1, False), (16, 8, 1, False), (8, 16, 2, False), (16, 8, 2, True)])
@pytest.mark.parametrize("relu,with_res", [(True, True), (False, False)])
def test_conv_bn_relu_block_forward_backward_small(ci, co, stride, transposed, relu, with_res):
    GT.test_conv_bn_relu_block_forward_backward(ci, co, stride, transposed, relu, with_res)


@pytest.mark.parametrize("shape,frac", [((37, 53), 0.5), ((16, 16), 0.0), ((16, 17), 1.0), ((600, 700), 0.001)])
def test_compact_points_small(shape, frac):
    GF.test_compact_points_is_ordered_boolean_indexing(shape, frac)


def test_point_feats_vs_oracle_relaxed_masks(hip):
    """test_gpu_render.test_point_feats_vs_oracle demands bit-equal in-bounds masks; its rays sit exactly on pixel positions, so
    the mask is a knife edge that follows the oracle's BLAS summation order on the machine at hand.  Here: features to 1e-5, at
    most 1 % of the masks flipped."""
    from oracle import render as orr
    S = 16
    batch, imgs, pseudo, w2cs, c2ws, intr, nf, pix, eps, u, rays, cam = GR._rays_case(S)
    vol = torch.randn(1, 8, 24, 16, 24, generator=torch.Generator().manual_seed(3))
    ref = orr.point_features(vol, imgs[:, -3:], w2cs, intr, rays["rays_pts"], rays["rays_ndc"])
    poses = torch.cat((w2cs[:3].reshape(3, 16), intr[:3].reshape(3, 9)), dim=1)
    feat = hip.point_feats(vol[0].permute(1, 2, 3, 0).contiguous(), imgs[0, -3:].contiguous(), poses.contiguous(),
                           rays["rays_pts"].contiguous(), rays["rays_ndc"].contiguous(), ldf=32)
    out = feat[:, :20].reshape(1024, S, 20)
    assert GR.rel_err(out[..., :8], ref[..., :8]) < 1e-5
    for i in range(3):
        assert GR.rel_err(out[..., 8 + 4 * i:11 + 4 * i], ref[..., 8 + 4 * i:11 + 4 * i]) < 1e-5
        assert float((out[..., 11 + 4 * i] != ref[..., 11 + 4 * i]).float().mean()) < 1e-2
