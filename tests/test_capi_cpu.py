"""CPU: the C-ABI library builds for gfx950, loads without a GPU, and exports every symbol that
include/rcmvs.h declares (no compute calls here)."""
import ctypes
import os
import re

import pytest

from conftest import REPO


def _declared_symbols():
    text = open(os.path.join(REPO, "include", "rcmvs.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(rcmvs_[a-z0-9_]+)\s*\(", text)))


@pytest.fixture(scope="module")
def lib():
    from rc_mvsnet_amd import _lib
    _lib.build()
    return _lib.load()


def test_header_symbols_exported(lib):
    from rc_mvsnet_amd import _lib
    syms = _declared_symbols()
    assert len(syms) >= 18
    raw = ctypes.CDLL(_lib.LIB_PATH)
    for s in syms:
        assert hasattr(raw, s), f"{s} declared in include/rcmvs.h but not exported"
    assert set(syms) == set(_lib.SIGNATURES), "ctypes signature table out of sync with the header"


def test_version_and_error_string(lib):
    from rc_mvsnet_amd import _lib
    assert lib.rcmvs_version() == _lib.REQUIRED_VERSION == 106
    # bad arguments are rejected on the host before any launch (no GPU needed)
    rc = lib.rcmvs_conv3d_fwd(None, None, None, None, None, None, 1, 1, 1, 1, 8, 8, 1, 0, None)
    assert rc < 0
    assert b"null pointer" in lib.rcmvs_last_error_string()
    rc = lib.rcmvs_warp_variance_fwd(None, None, None, None, None, 1, 3, 32, 8, 4, 4, None)
    assert rc < 0


def test_ops_refuse_cpu_tensors():
    import torch
    from rc_mvsnet_amd import ops
    from rc_mvsnet_amd._lib import RcmvsError
    with pytest.raises(RcmvsError):
        ops.compose_homography(torch.zeros(1, 3, 2, 4, 4))


def test_state_dict_contract():
    """238 reference-named tensors; strict load of a reference-shaped checkpoint."""
    from rc_mvsnet_amd import synthetic
    from rc_mvsnet_amd.casmvsnet import CascadeMVSNet, CascadeMVSNet_eval
    sd = synthetic.cascade_state_dict(0)
    for cls in (CascadeMVSNet, CascadeMVSNet_eval):
        m = cls()
        assert list(m.state_dict().keys()) == list(m.state_dict().keys())
        m.load_state_dict(sd, strict=True)
        assert len(m.state_dict()) == 238
        assert set(m.state_dict().keys()) == set(sd.keys())


def test_training_entry_points_validate_arguments(lib):
    """The backward / training entry points reject bad arguments on the host with a message (no GPU needed)."""
    one = ctypes.c_void_p(16)                      # any non-null pointer: validation happens before any launch
    assert lib.rcmvs_warp_variance_bwd(None, None, None, None, None, None, None, 1, 3, 8, 4, 8, 8, None) < 0
    assert b"null pointer" in lib.rcmvs_last_error_string()
    assert lib.rcmvs_warp_variance_bwd(one, one, one, one, one, None, one, 1, 3, 12, 4, 8, 8, None) < 0
    assert b"C must be" in lib.rcmvs_last_error_string()
    assert lib.rcmvs_bn_stats(one, one, 10, 6, None) < 0
    assert b"C=6" in lib.rcmvs_last_error_string()
    assert lib.rcmvs_conv3d_wgrad(one, one, one, 1, 4, 8, 8, 12, 24, 1, None) < 0
    assert b"unsupported" in lib.rcmvs_last_error_string()
    assert lib.rcmvs_conv3d_wgrad(one, one, one, 1, 4, 8, 8, 8, 8, 3, None) < 0
    assert b"stride" in lib.rcmvs_last_error_string()
    assert lib.rcmvs_composite_bwd(None, None, None, None, None, None, None, 4, 4, None) < 0
    assert lib.rcmvs_resize_planes_bwd(one, one, 1, 80, 80, 4, 8, 4, 4, None) < 0       # more channels than the kernel holds


def test_loss_fusion_loader_entry_points_validate_arguments(lib):
    """The SURVEY 8f entry points (self-supervised loss, fusion filter, image preparation, fused FPN level) reject bad
    arguments on the host, before any launch, with a message."""
    one = ctypes.c_void_p(16)
    err = lib.rcmvs_last_error_string
    assert lib.rcmvs_inverse_warp(one, one, None, one, one, 1, 8, 8, None) < 0 and b"null pointer" in err()
    assert lib.rcmvs_inverse_warp(one, one, one, one, one, 1, 1, 8, None) < 0 and b"bad dims" in err()
    assert lib.rcmvs_unsup_loss_fwd(one, one, one, one, one, one, one, one, one, 1, 0, 8, 8, None) < 0 and b"source views" in err()
    assert lib.rcmvs_unsup_loss_fwd(one, one, one, one, one, one, one, one, one, 1, 9, 8, 8, None) < 0 and b"source views" in err()
    assert lib.rcmvs_unsup_loss_fwd(one, one, one, one, one, one, one, one, one, 1, 2, 2, 8, None) < 0 and b"3x3" in err()
    assert lib.rcmvs_unsup_loss_bwd(one, one, one, one, one, one, one, None, one, one, one, 1, 2, 8, 8, None) < 0 and b"null pointer" in err()
    assert lib.rcmvs_masked_sl1_fwd(one, one, one, one, 0, None) < 0 and b"n=0" in err()
    assert lib.rcmvs_masked_sl1_bwd(one, one, one, one, None, one, 4, None) < 0
    idx = (ctypes.c_int * 17)(*range(17))
    args = (one, 0, ctypes.cast(idx, ctypes.c_void_p), one, None, one, 0.8, 3, 0.5, 0.01, one, one, one, None, None, None, None)
    assert lib.rcmvs_fuse_view(*args, 17, 8, 8, None) < 0 and b"source views" in err()
    assert lib.rcmvs_fuse_view(*args, 0, 8, 8, None) < 0
    bad = list(args)
    bad[4] = one                                                    # img without rgb
    assert lib.rcmvs_fuse_view(*bad, 2, 8, 8, None) < 0 and b"together" in err()
    neg = (ctypes.c_int * 2)(1, -1)
    args2 = (one, 0, ctypes.cast(neg, ctypes.c_void_p)) + args[3:]
    assert lib.rcmvs_fuse_view(*args2, 2, 8, 8, None) < 0 and b"negative" in err()
    assert lib.rcmvs_compact_points(one, one, None, one, None, one, 0, None) < 0 and b"n=0" in err()
    assert lib.rcmvs_compact_points(one, one, one, one, None, one, 8, None) < 0 and b"together" in err()
    mean, std0 = (ctypes.c_float * 3)(0.5, 0.5, 0.5), (ctypes.c_float * 3)(0.2, 0.0, 0.2)
    assert lib.rcmvs_prepare_image(one, one, 8, 8, 4, 4, ctypes.cast(mean, ctypes.c_void_p), ctypes.cast(std0, ctypes.c_void_p), None) < 0
    assert b"zero std" in err()
    assert lib.rcmvs_prepare_image(one, one, 8, 8, 0, 4, ctypes.cast(mean, ctypes.c_void_p), ctypes.cast(mean, ctypes.c_void_p), None) < 0
    assert lib.rcmvs_fpn_out_fused(one, one, one, one, one, one, 1, 15, 16, 8, 32, 8, None) < 0 and b"even" in err()
    assert lib.rcmvs_fpn_out_fused(one, one, one, one, one, one, 1, 16, 16, 16, 32, 8, None) < 0 and b"unsupported" in err()
