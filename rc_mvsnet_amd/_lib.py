"""ctypes binding of librcmvs_hip.so (C ABI declared in include/rcmvs.h).

The library is built in-tree by ``__graft_entry__.build()`` (hipcc --offload-arch=gfx950).
There is no fallback: if the shared object is missing or a call fails, an exception is raised.
"""
import ctypes
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "librcmvs_hip.so")
REQUIRED_VERSION = 106      # RCMVS_VERSION of include/rcmvs.h this binding was written against (106: rcmvs_conv2d_stem_fwd)
CSRC = os.path.join(_HERE, "csrc")
HEADER = os.path.join(os.path.dirname(_HERE), "include", "rcmvs.h")

_p = ctypes.c_void_p
_i = ctypes.c_int
_f = ctypes.c_float
_ll = ctypes.c_longlong
_d = ctypes.c_double

# name -> argtypes (restype is int unless listed in _RESTYPES); mirrors include/rcmvs.h
SIGNATURES = {
    "rcmvs_version": [],
    "rcmvs_last_error_string": [],
    "rcmvs_nchw_to_nhwc": [_p, _p, _i, _i, _ll, _p],
    "rcmvs_nhwc_to_nchw": [_p, _p, _i, _i, _ll, _p],
    "rcmvs_compose_homography": [_p, _p, _p, _i, _i, _p],
    "rcmvs_compose_homography_stages": [_p, _p, _p, _p, _i, _p, _p, _i, _i, _p, ctypes.c_longlong, _p],
    "rcmvs_hypothesis_planes": [_p, _p, _p, _i, _i, _i, _i, _i, _i, _i, _f, _i, _p],
    "rcmvs_warp_variance_fwd": [_p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _p],
    "rcmvs_debug_warp_variance_fwd": [_p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _i, _p],
    "rcmvs_warp_variance_hint_fwd": [_p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _i, _p],
    "rcmvs_warp_variance_timed_fwd": [_p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _i, _p, _p, _p],
    "rcmvs_debug_warp_variance_win_fwd": [_p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _i, _p, _p],
    "rcmvs_absmax_fwd": [_p, _ll, _i, _p, _p],
    "rcmvs_conv3d_scaled_fwd": [_p, _p, _p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _i, _i, _i, _p],
    "rcmvs_deconv3d_scaled_fwd": [_p, _p, _p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _i, _i, _p],
    "rcmvs_conv11_prob_fwd": [_p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _p],
    "rcmvs_softmax_head_fwd": [_p, _p, _p, _p, _i, _i, _i, _i, _i, _p],
    "rcmvs_conv2d_s2d_fwd": [_p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _p],
    "rcmvs_warp_variance_bwd": [_p, _p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _p],
    "rcmvs_debug_warp_variance_bwd": [_p, _p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _i, _p],
    "rcmvs_resize_planes_bwd": [_p, _p, _i, _i, _i, _i, _i, _i, _i, _p],
    "rcmvs_composite_bwd": [_p, _p, _p, _p, _p, _p, _p, _i, _i, _p],
    "rcmvs_point_feats_bwd": [_p, _p, _p, _i, _i, _i, _i, _i, _p],
    "rcmvs_inverse_warp": [_p, _p, _p, _p, _p, _i, _i, _i, _p],
    "rcmvs_unsup_loss_fwd": [_p, _p, _p, _p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _p],
    "rcmvs_unsup_loss_bwd": [_p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _p],
    "rcmvs_masked_sl1_fwd": [_p, _p, _p, _p, _ll, _p],
    "rcmvs_masked_sl1_bwd": [_p, _p, _p, _p, _p, _p, _ll, _p],
    "rcmvs_fuse_view": [_p, _i, _p, _p, _p, _p, _f, _i, _d, _f, _p, _p, _p, _p, _p, _p, _p, _i, _i, _i, _p],
    "rcmvs_compact_points": [_p, _p, _p, _p, _p, _p, _ll, _p],
    "rcmvs_prepare_image": [_p, _p, _i, _i, _i, _i, _p, _p, _p],
    "rcmvs_resize_rgb_cl": [_p, _p, _i, _i, _i, _i, _i, _p],
    "rcmvs_conv2d_pair_weight_floats": [],
    "rcmvs_pack_conv2d_pair": [_p, _p, _p, _p],
    "rcmvs_conv2d_pair_fwd": [_p, _p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _p],
    "rcmvs_conv2d_tile_weight_floats": [_i],
    "rcmvs_pack_conv2d_tile": [_p, _p, _i, _p],
    "rcmvs_conv2d_tile_fwd": [_p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _p, _p],
    "rcmvs_conv2d_stem_weight_floats": [],
    "rcmvs_pack_conv2d_stem": [_p, _p, _p],
    "rcmvs_conv2d_stem_fwd": [_p, _p, _p, _p, _p, _p, _p, _p, _i, _i, _i, _p],
    "rcmvs_conv1x1_fwd": [_p, _p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _p],
    "rcmvs_conv1x1_mfma_fwd": [_p, _p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _p],
    "rcmvs_fpn_out_fused": [_p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _p],
    "rcmvs_fpn_out_folded": [_p, _p, _p, _p, _p, _i, _i, _i, _p],
    "rcmvs_fpn_folded_mfma_floats": [],
    "rcmvs_fpn_folded_mfma_pack": [_p, _p, _p],
    "rcmvs_fpn_out_folded_mfma": [_p, _p, _p, _p, _p, _i, _i, _i, _p],
    "rcmvs_bn_stats": [_p, _p, _ll, _i, _p],
    "rcmvs_bn_finalize": [_p, _p, _p, _p, _f, _f, _p, _p, _p, _p, _p, _p, _p, _i, _p],
    "rcmvs_bn_bwd_finalize": [_p, _p, _p, _p, _p, _p, _i, _p],
    "rcmvs_scale_shift_relu": [_p, _p, _p, _p, _p, _ll, _i, _i, _p],
    "rcmvs_bn_norm_fwd": [_p, _p, _p, _p, _p, _f, _f, _p, _p, _p, _p, _p, _p, _ll, _i, _i, _p],
    "rcmvs_bn_norm_bwd": [_p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _ll, _i, _i, _p],
    "rcmvs_bn_bwd_reduce": [_p, _p, _p, _p, _p, _p, _p, _ll, _i, _i, _p],
    "rcmvs_bn_bwd_apply": [_p, _p, _p, _p, _p, _p, _p, _p, _ll, _i, _i, _p],
    "rcmvs_conv3d_wgrad": [_p, _p, _p, _i, _i, _i, _i, _i, _i, _i, _p],
    "rcmvs_wgrad_finish": [_p, _p, _i, _i, _i, _p],
    "rcmvs_conv3d_dgrad_c1": [_p, _p, _p, _i, _i, _i, _i, _i, _p],
    "rcmvs_depth_head_bwd": [_p, _p, _p, _p, _p, _i, _i, _i, _i, _p],
    "rcmvs_warp_noref_fwd": [_p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _i, _p],
    "rcmvs_packed_weight_floats": [_i, _i],
    "rcmvs_pack_conv3d_weight": [_p, _p, _i, _i, _i, _p],
    "rcmvs_pack_conv3d_weight_sel": [_p, _p, _i, _i, _i, _i, _p],
    "rcmvs_conv3d_images": [_i, _i, _i, _i, _i],
    "rcmvs_debug_conv3d_fwd": [_p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _i, _i, _i, _p],
    "rcmvs_debug_deconv3d_fwd": [_p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _i, _i, _p],
    "rcmvs_conv3d_fwd": [_p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _i, _i, _p],
    "rcmvs_deconv3d_fwd": [_p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _i, _p],
    "rcmvs_rgb_to_nhwc4": [_p, _p, _i, _i, _i, _p],
    "rcmvs_pack_conv2d_weight": [_p, _p, _i, _i, _i, _i, _p],
    "rcmvs_conv2d_fwd": [_p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _i, _i, _p],
    "rcmvs_depth_head_fwd": [_p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _p],
    "rcmvs_depth_head_scaled_fwd": [_p, _p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _p],
    "rcmvs_resize_planes_fwd": [_p, _p, _i, _i, _i, _i, _i, _i, _i, _p],
    "rcmvs_gu_sample_fwd": [_p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _p],
    "rcmvs_point_feats_fwd": [_p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _i, _i, _p],
    "rcmvs_nerf_mlp_fwd": [_p, _p, _i, _p, _p, _p, _p, _p, _i, _i, _p],
    "rcmvs_nerf_mlp_embedded_fwd": [_p, _i, _p, _p, _p, _p, _ll, _p],
    "rcmvs_nerf_mlp_train_fwd": [_p, _p, _i, _p, _p, _p, _p, _p, _i, _i, _p],
    "rcmvs_nerf_mlp_bwd": [_p, _p, _i, _p, _p, _p, _p, _p, _p, _i, _i, _p],
    "rcmvs_nerf_train_workspace_floats": [_ll],
    "rcmvs_nerf_bwd_workspace_floats": [_ll],
    "rcmvs_nerf_weight_floats": [],
    "rcmvs_nerf_workspace_floats": [_ll],
    "rcmvs_pack_nerf_weights": [_p, _p, _p],
    "rcmvs_composite_fwd": [_p, _p, _p, _p, _p, _p, _i, _i, _p],
}
_RESTYPES = {"rcmvs_last_error_string": ctypes.c_char_p, "rcmvs_nerf_weight_floats": _ll, "rcmvs_nerf_workspace_floats": _ll, "rcmvs_nerf_train_workspace_floats": _ll, "rcmvs_nerf_bwd_workspace_floats": _ll,
             "rcmvs_packed_weight_floats": _ll, "rcmvs_fpn_folded_mfma_floats": _ll, "rcmvs_conv2d_pair_weight_floats": _ll, "rcmvs_conv2d_stem_weight_floats": _ll, "rcmvs_conv2d_tile_weight_floats": _ll}

_lib = None


class RcmvsError(RuntimeError):
    pass


def sources():
    return sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".hip"))


def build(force=False, verbose=False):
    """Compile every HIP source into librcmvs_hip.so for gfx950 (cross-compiles without a GPU).  One object per source
    (csrc/_obj/*.o, rebuilt when the source or any header is newer), compiled in parallel, then one link step."""
    from concurrent.futures import ThreadPoolExecutor
    srcs = sources()
    hdrs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")] + [HEADER]
    if not force and os.path.exists(LIB_PATH) and all(os.path.getmtime(LIB_PATH) >= os.path.getmtime(d) for d in srcs + hdrs):
        return LIB_PATH
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    objdir = os.path.join(_HERE, "_obj")
    os.makedirs(objdir, exist_ok=True)
    hdr_time = max(os.path.getmtime(h) for h in hdrs)
    jobs, objs = [], []
    for src in srcs:
        obj = os.path.join(objdir, os.path.basename(src)[:-4] + ".o")
        objs.append(obj)
        if force or not os.path.exists(obj) or os.path.getmtime(obj) < max(os.path.getmtime(src), hdr_time):
            jobs.append([hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-c", src, "-o", obj])

    def run(cmd):
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.run(cmd, check=True)

    with ThreadPoolExecutor(max_workers=min(8, os.cpu_count() or 1)) as ex:
        list(ex.map(run, jobs))
    run([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB_PATH] + objs)
    return LIB_PATH


def load():
    """Load the shared library (raises RcmvsError when it has not been built)."""
    global _lib
    if _lib is not None:
        return _lib
    path = os.environ.get("RCMVS_LIB") or LIB_PATH       # RCMVS_LIB: another build of the same ABI (developer A/B of kernel variants, tools/dev/build_variant.sh)
    if not os.path.exists(path):
        raise RcmvsError(f"{path} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                         "(hipcc --offload-arch=gfx950). There is no CPU / eager fallback.")
    lib = ctypes.CDLL(path)
    for name, argtypes in SIGNATURES.items():
        fn = getattr(lib, name)          # AttributeError if the symbol is not exported
        fn.argtypes = argtypes
        fn.restype = _RESTYPES.get(name, ctypes.c_int)
    if lib.rcmvs_version() < REQUIRED_VERSION:
        raise RcmvsError(f"librcmvs_hip.so (version {lib.rcmvs_version()}) is older than this package needs ({REQUIRED_VERSION}): rebuild it")
    _lib = lib
    return lib


def check(rc, what):
    if rc != 0:
        msg = load().rcmvs_last_error_string()
        raise RcmvsError(f"{what} failed (rc={rc}): {msg.decode() if msg else ''}")
