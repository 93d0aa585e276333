"""Thin tensor-level wrappers over the C ABI (include/rcmvs.h).

PyTorch is plumbing here: it owns device memory and the current HIP stream; every function
below checks its inputs, allocates the outputs with torch.empty and enqueues exactly one
library call on the current stream.  Layout convention inside the library is channels-last
(maps (B,h,w,C), volumes (B,D,h,w,C)).
"""
import ctypes
import os

import torch

from . import _lib

# bench.py sets this to a list to have every K1 launch bracketed by HIP events recorded on the
# launch stream (torch's current stream); None = no instrumentation.
K1_EVENTS = None
K1_UNIFORM_PLANES = 1          # RCMVS_K1_UNIFORM_PLANES (include/rcmvs.h)
K1_FAST_BLEND = 2              # RCMVS_K1_FAST_BLEND: the FMA-contracted forms (<= 2e-6 of the value range from the exact kernel)
# same for the 3-D convolutions: list of (event0, event1, key) with key = (kind, B, D, H, W, Ci, Co), kind 's1' | 's2' | 't2'
CONV_EVENTS = None


# One HIP stream per device and process.  With kernels of one process in flight on two hardware queues of an MI355X, a consumer kernel can
# read STALE 64-byte granules of a tensor its predecessor on the same stream has just rewritten completely (profiles/r6_two_streams.txt: seen by
# an ATen copy kernel as well as by ours, any producer / consumer pair, gone with GPU_MAX_HW_QUEUES=1 or AMD_SERIALIZE_KERNEL=3) -- wrong numbers,
# no error.  So the binding remembers the first stream it launched on per device and refuses a second one; two worker PROCESSES per GPU
# (sharding.py) are the supported overlap.  GPU_MAX_HW_QUEUES=1 (all streams of the process share one hardware queue) lifts the refusal.
_FIRST_STREAM = {}


def _multi_stream_allowed():
    return os.environ.get("GPU_MAX_HW_QUEUES") == "1" or os.environ.get("RCMVS_ALLOW_MULTI_STREAM") == "1"


def _stream():
    st = torch.cuda.current_stream()
    key = st.device_index
    first = _FIRST_STREAM.setdefault(key, st.cuda_stream)
    if first != st.cuda_stream and not _multi_stream_allowed():
        raise _lib.RcmvsError(
            f"rc_mvsnet_amd launches on ONE HIP stream per device and process: cuda:{key} started on stream {first:#x}, this call is on "
            f"{st.cuda_stream:#x}.  Two hardware queues of one process give silently wrong results on this platform "
            "(profiles/r6_two_streams.txt); use one stream, or one worker process per stream (rc_mvsnet_amd.sharding), or run the "
            "process with GPU_MAX_HW_QUEUES=1.")
    return ctypes.c_void_p(st.cuda_stream)


def _chk(t, name, dtype=torch.float32):
    if not t.is_cuda:
        raise _lib.RcmvsError(f"{name}: expected a tensor on the GPU (the HIP path has no CPU fallback)")
    if t.dtype != dtype:
        raise _lib.RcmvsError(f"{name}: expected {dtype}, got {t.dtype}")
    if not t.is_contiguous():
        raise _lib.RcmvsError(f"{name}: expected a contiguous tensor")
    return ctypes.c_void_p(t.data_ptr())


def _opt(t, name):
    return ctypes.c_void_p(0) if t is None else _chk(t, name)


# ------------------------------------------------------------------------------- layout
def to_channels_last(x):
    """(N,C,*spatial) -> (N,*spatial,C)."""
    N, C = x.shape[:2]
    S = x[0, 0].numel()
    out = torch.empty((N, *x.shape[2:], C), device=x.device, dtype=torch.float32)
    _lib.check(_lib.load().rcmvs_nchw_to_nhwc(_chk(x, "x"), _chk(out, "out"), N, C, S, _stream()), "nchw_to_nhwc")
    return out


def to_channels_first(x):
    """(N,*spatial,C) -> (N,C,*spatial)."""
    N, C = x.shape[0], x.shape[-1]
    S = x[0, ..., 0].numel()
    out = torch.empty((N, C, *x.shape[1:-1]), device=x.device, dtype=torch.float32)
    _lib.check(_lib.load().rcmvs_nhwc_to_nchw(_chk(x, "x"), _chk(out, "out"), N, C, S, _stream()), "nhwc_to_nchw")
    return out


# ------------------------------------------------------------------------------- geometry
def compose_homography(proj):
    """proj (B,V,2,4,4) -> rot (B,V-1,9), trans (B,V-1,3)."""
    B, V = proj.shape[:2]
    rot = torch.empty((B, V - 1, 9), device=proj.device, dtype=torch.float32)
    trans = torch.empty((B, V - 1, 3), device=proj.device, dtype=torch.float32)
    _lib.check(_lib.load().rcmvs_compose_homography(_chk(proj, "proj"), _chk(rot, "rot"), _chk(trans, "trans"), B, V, _stream()),
               "compose_homography")
    return rot, trans


def compose_homography_stages(projs, zero=None):
    """[proj (B,V,2,4,4)] x nstage (<= 4) -> rot (nstage,B,V-1,9), trans (nstage,B,V-1,3): the cascade's stages in one launch.
    zero: an optional fp32 tensor the launch clears on the side (the scene's activation-bound rows)."""
    B, V = projs[0].shape[:2]
    n = len(projs)
    rot = torch.empty((n, B, V - 1, 9), device=projs[0].device, dtype=torch.float32)
    trans = torch.empty((n, B, V - 1, 3), device=projs[0].device, dtype=torch.float32)
    ptrs = [_chk(p, "proj") for p in projs] + [ctypes.c_void_p(0)] * (4 - n)
    _lib.check(_lib.load().rcmvs_compose_homography_stages(ptrs[0], ptrs[1], ptrs[2], ptrs[3], n, _chk(rot, "rot"), _chk(trans, "trans"), B, V,
                                                           _opt(zero, "zero"), 0 if zero is None else zero.numel(), _stream()), "compose_homography_stages")
    return rot, trans


def hypothesis_planes(prev_depth, depth_values, full_hw, scale, ndepth, ratio):
    """-> planes (B, H/scale, W/scale, 2) = {d_0, delta}."""
    B, ND = depth_values.shape
    H, W = full_hw
    planes = torch.empty((B, H // scale, W // scale, 2), device=depth_values.device, dtype=torch.float32)
    hp, wp = (prev_depth.shape[-2:] if prev_depth is not None else (0, 0))
    _lib.check(_lib.load().rcmvs_hypothesis_planes(_opt(prev_depth, "prev_depth"), _chk(depth_values, "depth_values"),
                                                   _chk(planes, "planes"), B, hp, wp, H, W, scale, ndepth, float(ratio), ND,
                                                   _stream()), "hypothesis_planes")
    return planes


# ------------------------------------------------------------------------------- K1
ABSMAX_FLOATS = 1024      # RCMVS_ABSMAX_FLOATS of include/rcmvs.h: a bound is 64 slots, 16 floats apart


def absmax(x, square=False, out=None):
    """Bound of max|x| (or its square) in the slot format conv3d(x_absmax=) takes: a (1024,) float vector whose maximum is the bound."""
    if out is None:
        out = torch.zeros(ABSMAX_FLOATS, device=x.device, dtype=torch.float32)
    _lib.check(_lib.load().rcmvs_absmax_fwd(_chk(x, "x"), x.numel(), int(bool(square)), _chk(out, "absmax"), _stream()), "absmax_fwd")
    return out


def warp_variance(feats, rot, trans, planes, ndepth, variant=None, uniform_planes=False):
    """feats (B,V,h,w,C) -> variance volume (B,D,h,w,C).  uniform_planes: the caller's hint that the plane table is the same for every
    pixel (stage 1 of the cascade): with two source views the kernel then stages the tiles' source windows in LDS (csrc/k1_win.h;
    same results within ~4e-7 of the value range, per-tile fallback).  variant (None = the production call): the test / profiling code
    variants of rcmvs_debug_warp_variance_fwd (0 two-phase exact kernel, 1 its FMA build, 2 reference-order kernel, 3 store-only,
    5 / 6 window form, 7 plane-pipelined gather form)."""
    B, V, h, w, C = feats.shape
    var = torch.empty((B, ndepth, h, w, C), device=feats.device, dtype=torch.float32)
    if variant is not None:
        _lib.check(_lib.load().rcmvs_debug_warp_variance_fwd(_chk(feats, "feats"), _chk(rot, "rot"), _chk(trans, "trans"),
                                                             _chk(planes, "planes"), _chk(var, "var"), B, V, C, ndepth, h, w, int(variant),
                                                             _stream()), "debug_warp_variance_fwd")
        return var
    hint = K1_FAST_BLEND | (K1_UNIFORM_PLANES if uniform_planes else 0)
    if K1_EVENTS is not None:
        # the kernel's own start / stop timestamps (hipExtLaunchKernelGGL through rcmvs_warp_variance_timed_fwd): what rocprofv3 reports as the launch's
        # duration.  Event RECORDS around the launch measure 3-6 us more per launch (two marker packets).  A torch event owns its hipEvent_t only after a
        # first record, so both are recorded once here and then overwritten by the launch.
        ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
        ev[0].record()
        ev[1].record()
        _lib.check(_lib.load().rcmvs_warp_variance_timed_fwd(_chk(feats, "feats"), _chk(rot, "rot"), _chk(trans, "trans"),
                                                             _chk(planes, "planes"), _chk(var, "var"), B, V, C, ndepth, h, w, hint,
                                                             ctypes.c_void_p(ev[0].cuda_event), ctypes.c_void_p(ev[1].cuda_event), _stream()),
                   "warp_variance_timed_fwd")
        K1_EVENTS.append(ev)
        return var
    _lib.check(_lib.load().rcmvs_warp_variance_hint_fwd(_chk(feats, "feats"), _chk(rot, "rot"), _chk(trans, "trans"),
                                                        _chk(planes, "planes"), _chk(var, "var"), B, V, C, ndepth, h, w, hint, _stream()),
               "warp_variance_hint_fwd")
    return var


def warp_variance_win(feats, rot, trans, planes, ndepth, variant=5):
    """The window-form K1 kernel with its tile statistics -> (variance volume, blocks launched, blocks on the LDS-window path)."""
    B, V, h, w, C = feats.shape
    var = torch.empty((B, ndepth, h, w, C), device=feats.device, dtype=torch.float32)
    stats = torch.zeros(2, device=feats.device, dtype=torch.int32)
    _lib.check(_lib.load().rcmvs_debug_warp_variance_win_fwd(_chk(feats, "feats"), _chk(rot, "rot"), _chk(trans, "trans"),
                                                                 _chk(planes, "planes"), _chk(var, "var"), B, V, C, ndepth, h, w, int(variant),
                                                                 _chk(stats, "stats", torch.int32), _stream()), "debug_warp_variance_win_fwd")
    n = stats.cpu()
    return var, int(n[0]), int(n[1])


def warp_noref(feats, imgs, rot, trans, planes, ndepth, square_first):
    """feats (B,V,h,w,C), imgs (B,V,h,w,3) -> (B, 3(V-1)+C, D, h, w) in the reference's NCDHW."""
    B, V, h, w, C = feats.shape
    out = torch.empty((B, 3 * (V - 1) + C, ndepth, h, w), device=feats.device, dtype=torch.float32)
    _lib.check(_lib.load().rcmvs_warp_noref_fwd(_chk(feats, "feats"), _chk(imgs, "imgs"), _chk(rot, "rot"), _chk(trans, "trans"),
                                                _chk(planes, "planes"), _chk(out, "out"), B, V, C, ndepth, h, w,
                                                int(bool(square_first)), _stream()), "warp_noref_fwd")
    return out


def warp_variance_bwd(feats, rot, trans, planes, grad_var, grad_noref=None, variant=0):
    """d loss / d feats (B,V,h,w,C) from d loss / d var (and, optionally, d loss / d no-ref variance), both
    (B,D,h,w,C) channels-last."""
    B, V, h, w, C = feats.shape
    D = grad_var.shape[1]
    if tuple(grad_var.shape) != (B, D, h, w, C) or (grad_noref is not None and grad_noref.shape != grad_var.shape):
        raise _lib.RcmvsError(f"warp_variance_bwd: gradient shape {tuple(grad_var.shape)} does not match (B,D,h,w,C)")
    gf = torch.zeros_like(feats)
    if variant:          # ablation twin (bit 0: no scatter -- timing only; bit 1: no run-length merging)
        _lib.check(_lib.load().rcmvs_debug_warp_variance_bwd(_chk(feats, "feats"), _chk(rot, "rot"), _chk(trans, "trans"),
                                                             _chk(planes, "planes"), _chk(grad_var, "grad_var"),
                                                             _opt(grad_noref, "grad_noref"), _chk(gf, "grad_feats"),
                                                             B, V, C, D, h, w, int(variant), _stream()), "debug_warp_variance_bwd")
        return gf
    _lib.check(_lib.load().rcmvs_warp_variance_bwd(_chk(feats, "feats"), _chk(rot, "rot"), _chk(trans, "trans"),
                                                   _chk(planes, "planes"), _chk(grad_var, "grad_var"),
                                                   _opt(grad_noref, "grad_noref"), _chk(gf, "grad_feats"),
                                                   B, V, C, D, h, w, _stream()), "warp_variance_bwd")
    return gf


class WarpVarianceFn(torch.autograd.Function):
    """Differentiable K1: feats (B,V,h,w,C) -> var (B,D,h,w,C) [, volume_feature_no_ref (B,3(V-1)+C,D,h,w)].

    Forward = rcmvs_warp_variance_fwd (+ rcmvs_warp_noref_fwd when `imgs` is given: the train variant of
    models/casmvsnet.py:59-101 in train mode); backward = rcmvs_warp_variance_bwd.  Only `feats` is
    differentiable (the reference's sampling grid is built under no_grad)."""

    @staticmethod
    def forward(ctx, feats, rot, trans, planes, ndepth, imgs):
        feats = feats.contiguous()
        ctx.save_for_backward(feats, rot, trans, planes)
        ctx.ndepth = ndepth
        ctx.set_materialize_grads(False)      # an unused output (stage 2/3 no-ref volumes) must not cost a zero-filled gradient tensor
        var = warp_variance(feats, rot, trans, planes, ndepth)
        if imgs is None:
            return var
        noref = warp_noref(feats, imgs, rot, trans, planes, ndepth, square_first=False)
        return var, noref

    @staticmethod
    def backward(ctx, gvar, gnoref=None):
        feats, rot, trans, planes = ctx.saved_tensors
        C = feats.shape[-1]
        if gvar is None and gnoref is None:
            return None, None, None, None, None, None
        if gvar is None:
            gvar = torch.zeros((feats.shape[0], ctx.ndepth, *feats.shape[2:]), device=feats.device)
        gnr = None
        if gnoref is not None:
            gnr = gnoref[:, -C:].permute(0, 2, 3, 4, 1).contiguous()
        gf = warp_variance_bwd(feats, rot, trans, planes, gvar.contiguous(), gnr)
        return gf, None, None, None, None, None


# ------------------------------------------------------------------------------- K2/K3
IMG_ALL = 0x7fffffff


class PackedWeight:
    """Device blob produced by rcmvs_pack_conv3d_weight plus its channel counts."""
    __slots__ = ("blob", "ci", "co", "k", "transposed", "images")

    def __init__(self, blob, ci, co, transposed=0, images=IMG_ALL):
        # transposed: the pack mode (0 conv weight, 1 ConvTranspose3d weight, 2 flipped adjoint of a stride-1 conv).  The blob holds
        # the matrix-core images of the matching kernels only (rcmvs_pack_conv3d_weight), so conv3d / deconv3d check it.
        # images: which images were written (rcmvs_pack_conv3d_weight_sel; IMG_ALL = every image the channel pair has)
        self.blob, self.ci, self.co, self.transposed, self.images = blob, ci, co, int(transposed), int(images)


_CONV_IMPL = 0      # test / A-B hook (force_direct_conv): kernel selection handed to the rcmvs_debug_* conv entry points


def force_direct_conv(on):
    """Test / bench hook: select the 3-D conv kernels for the following ops.conv3d / ops.deconv3d calls (bit 0 = direct kernels,
    bit 6 = no split-bf16 MFMA kernels, other bits see include/rcmvs.h; 0 / False = production dispatch).  The selection lives
    here, on the Python side: the library itself is stateless."""
    global _CONV_IMPL
    _CONV_IMPL = int(on)


def conv3d_images(ci, co, stride=1, transposed=False, planar=False):
    """Mask of the blob image the production dispatch reads for this layer (rcmvs_conv3d_images)."""
    return int(_lib.load().rcmvs_conv3d_images(co, ci, int(stride), int(bool(transposed)), int(bool(planar))))


def pack_conv3d_weight(w, transposed=False, use=None):
    """conv (Co,Ci,3,3,3) / deconv (Ci,Co,3,3,3) -> PackedWeight (direct [27][Ci][Co] + MFMA image).
    transposed=2: the adjoint of a stride-1 conv whose weight is (Ci,Co,3,3,3) (taps flipped as well).
    use=(stride, planar): write only the image the production dispatch reads for a call with that stride on a volume that
    is (planar=True) / is not one plane deep -- training re-packs every weight every step for exactly one call; inference plans keep
    the full blob (use=None).  Ignored while a test hook selects other kernels (force_direct_conv)."""
    w = w.detach().contiguous().float()
    Ci, Co = (w.shape[0], w.shape[1]) if transposed else (w.shape[1], w.shape[0])
    n = _lib.load().rcmvs_packed_weight_floats(Co, Ci)
    blob = torch.empty((n,), device=w.device, dtype=torch.float32)
    images = IMG_ALL
    if use is not None and not _CONV_IMPL:
        images = conv3d_images(Ci, Co, stride=use[0], transposed=(int(transposed) == 1), planar=use[1])
    _lib.check(_lib.load().rcmvs_pack_conv3d_weight_sel(_chk(w, "w"), _chk(blob, "packed"), Co, Ci, int(transposed), images, _stream()),
               "pack_conv3d_weight")
    return PackedWeight(blob, Ci, Co, int(transposed), images)


def _check_image(w_packed, ci, co, stride, transposed, planar, scaled, who):
    if w_packed.images == IMG_ALL:
        return
    if _CONV_IMPL or scaled or not (w_packed.images & conv3d_images(ci, co, stride, transposed, planar)):
        raise _lib.RcmvsError(f"{who}: the weight was packed for another call (pack_conv3d_weight(use=...)): the image this call reads was not written")


def conv3d(x, w_packed, scale=None, shift=None, residual=None, stride=1, relu=False, x_absmax=None, y_absmax=None, y_absmax_square=False):
    """x (B,D,H,W,Ci) -> (B,Do,Ho,Wo,Co) with fused [relu](v*scale+shift) + residual.
    x_absmax / y_absmax ((1024,) bound vectors, ops.absmax / rcmvs_conv3d_scaled_fwd): a bound of max|x| selects the fp16-pair
    matrix-core form where the channel pair has one; y_absmax (zero-filled by the caller) receives max|y| for the next layer --
    its square with y_absmax_square (one-plane volumes only: the FeatureNet output convs, whose maps the variance volume is built from)."""
    B, D, H, W, Ci = x.shape
    Co = w_packed.co
    if w_packed.ci != Ci:
        raise _lib.RcmvsError(f"conv3d: input has {Ci} channels, weight expects {w_packed.ci}")
    if w_packed.transposed == 1:
        raise _lib.RcmvsError("conv3d: the weight was packed as a ConvTranspose3d weight (transposed=1); its blob holds no conv images")
    _check_image(w_packed, Ci, Co, stride, False, D == 1, x_absmax is not None, "conv3d")
    y = torch.empty((B, (D - 1) // stride + 1, (H - 1) // stride + 1, (W - 1) // stride + 1, Co), device=x.device,
                    dtype=torch.float32)
    if residual is not None and residual.shape != y.shape:
        raise _lib.RcmvsError(f"conv3d: residual {tuple(residual.shape)} != output {tuple(y.shape)}")
    ev = None
    if CONV_EVENTS is not None:
        ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
        ev[0].record()
    if x_absmax is not None or y_absmax is not None:
        _lib.check(_lib.load().rcmvs_conv3d_scaled_fwd(_chk(x, "x"), _opt(x_absmax, "x_absmax"), _chk(w_packed.blob, "w"), _opt(scale, "scale"),
                                                       _opt(shift, "shift"), _opt(residual, "residual"), _chk(y, "y"), _opt(y_absmax, "y_absmax"),
                                                       B, D, H, W, Ci, Co, stride, int(relu), _CONV_IMPL | ((1 << 24) if y_absmax_square else 0), _stream()),
                   "conv3d_scaled_fwd")
    elif _CONV_IMPL:
        _lib.check(_lib.load().rcmvs_debug_conv3d_fwd(_chk(x, "x"), _chk(w_packed.blob, "w"), _opt(scale, "scale"), _opt(shift, "shift"),
                                                      _opt(residual, "residual"), _chk(y, "y"), B, D, H, W, Ci, Co, stride, int(relu),
                                                      _CONV_IMPL, _stream()), "debug_conv3d_fwd")
    else:
        _lib.check(_lib.load().rcmvs_conv3d_fwd(_chk(x, "x"), _chk(w_packed.blob, "w"), _opt(scale, "scale"), _opt(shift, "shift"),
                                                _opt(residual, "residual"), _chk(y, "y"), B, D, H, W, Ci, Co, stride, int(relu),
                                                _stream()), "conv3d_fwd")
    if ev is not None:
        ev[1].record()
        CONV_EVENTS.append(ev + (("s1" if stride == 1 else "s2", B, D, H, W, Ci, Co),))      # only complete pairs are listed
    return y


def deconv3d(x, w_packed, scale=None, shift=None, residual=None, relu=False, x_absmax=None, y_absmax=None):
    """x (B,D,H,W,Ci) -> (B,2D,2H,2W,Co).  x_absmax / y_absmax: as in conv3d (rcmvs_deconv3d_scaled_fwd)."""
    B, D, H, W, Ci = x.shape
    Co = w_packed.co
    if w_packed.ci != Ci:
        raise _lib.RcmvsError(f"deconv3d: input has {Ci} channels, weight expects {w_packed.ci}")
    if w_packed.transposed != 1:
        raise _lib.RcmvsError("deconv3d: the weight was not packed with transposed=True; its blob holds no transposed-conv image")
    _check_image(w_packed, Ci, Co, 2, True, False, x_absmax is not None, "deconv3d")
    y = torch.empty((B, 2 * D, 2 * H, 2 * W, Co), device=x.device, dtype=torch.float32)
    if residual is not None and residual.shape != y.shape:
        raise _lib.RcmvsError(f"deconv3d: residual {tuple(residual.shape)} != output {tuple(y.shape)} "
                              "(volume sizes must be divisible by 8, as in the reference)")
    ev = None
    if CONV_EVENTS is not None:
        ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
        ev[0].record()
    if x_absmax is not None or y_absmax is not None:
        _lib.check(_lib.load().rcmvs_deconv3d_scaled_fwd(_chk(x, "x"), _opt(x_absmax, "x_absmax"), _chk(w_packed.blob, "w"), _opt(scale, "scale"),
                                                         _opt(shift, "shift"), _opt(residual, "residual"), _chk(y, "y"), _opt(y_absmax, "y_absmax"),
                                                         B, D, H, W, Ci, Co, int(relu), _CONV_IMPL, _stream()), "deconv3d_scaled_fwd")
    elif _CONV_IMPL:
        _lib.check(_lib.load().rcmvs_debug_deconv3d_fwd(_chk(x, "x"), _chk(w_packed.blob, "w"), _opt(scale, "scale"), _opt(shift, "shift"),
                                                        _opt(residual, "residual"), _chk(y, "y"), B, D, H, W, Ci, Co, int(relu),
                                                        _CONV_IMPL, _stream()), "debug_deconv3d_fwd")
    else:
        _lib.check(_lib.load().rcmvs_deconv3d_fwd(_chk(x, "x"), _chk(w_packed.blob, "w"), _opt(scale, "scale"), _opt(shift, "shift"),
                                                  _opt(residual, "residual"), _chk(y, "y"), B, D, H, W, Ci, Co, int(relu),
                                                  _stream()), "deconv3d_fwd")
    if ev is not None:
        ev[1].record()
        CONV_EVENTS.append(ev + (("t2", B, D, H, W, Ci, Co),))
    return y


def conv2d_s2d(x, w_packed, scale=None, shift=None, relu=False):
    """5x5 stride-2 conv as a 3x3 conv of the space-to-depth view of x (N,H,W,C), the view never materialised: -> (N,H/2,W/2,Co).
    w_packed: pack_conv3d_weight of the (Co,4C,3,3,3) re-indexed weight."""
    N, H, W, C = x.shape
    Co = w_packed.co
    if w_packed.ci != 4 * C:
        raise _lib.RcmvsError(f"conv2d_s2d: input has {C} channels, the packed weight expects {w_packed.ci} = 4 x C")
    y = torch.empty((N, H // 2, W // 2, Co), device=x.device, dtype=torch.float32)
    _lib.check(_lib.load().rcmvs_conv2d_s2d_fwd(_chk(x, "x"), _chk(w_packed.blob, "w"), _opt(scale, "scale"), _opt(shift, "shift"),
                                                _chk(y, "y"), N, H, W, C, Co, int(relu), _stream()), "conv2d_s2d_fwd")
    return y


# ------------------------------------------------------------------------------- 2-D feature pyramid
def pack_conv2d_pair(wa, wb):
    """Two (16,16,3,3) Conv2d weights -> the fragment image of conv2d_pair (csrc/conv2d_pair.hip)."""
    if tuple(wa.shape) != (16, 16, 3, 3) or tuple(wb.shape) != (16, 16, 3, 3):
        raise _lib.RcmvsError("pack_conv2d_pair: built for two 16 -> 16 3x3 layers")
    lib = _lib.load()
    img = torch.empty(int(lib.rcmvs_conv2d_pair_weight_floats()), device=wa.device, dtype=torch.float32)
    _lib.check(lib.rcmvs_pack_conv2d_pair(_chk(wa.detach().float().contiguous(), "wa"), _chk(wb.detach().float().contiguous(), "wb"), _chk(img, "image"), _stream()),
               "pack_conv2d_pair")
    return img


def conv2d_pair(x, image, scale_a, shift_a, scale_b, shift_b):
    """x (N,H,W,16) -> relu(bn_b(conv_b(relu(bn_a(conv_a(x)))))) (N,H,W,16): two 3x3 Conv2d blocks in one launch, the map between them in LDS."""
    N, H, W, C = x.shape
    y = torch.empty_like(x)
    _lib.check(_lib.load().rcmvs_conv2d_pair_fwd(_chk(x, "x"), _chk(image, "image"), _chk(scale_a, "scale_a"), _chk(shift_a, "shift_a"),
                                                 _chk(scale_b, "scale_b"), _chk(shift_b, "shift_b"), _chk(y, "y"), N, H, W, C, _stream()), "conv2d_pair_fwd")
    return y


def resize_rgb_cl(x, hw):
    """(N,3,H,W) -> (N,h,w,3): F.interpolate(x, hw, mode="bilinear", align_corners=False) + the channels-last view, one launch (no gradient)."""
    N, C, H, W = x.shape
    if C != 3:
        raise _lib.RcmvsError("resize_rgb_cl: expects RGB images (N,3,H,W)")
    h, w = int(hw[0]), int(hw[1])
    y = torch.empty((N, h, w, 3), device=x.device, dtype=torch.float32)
    _lib.check(_lib.load().rcmvs_resize_rgb_cl(_chk(x, "x"), _chk(y, "y"), N, H, W, h, w, _stream()), "resize_rgb_cl")
    return y


def rgb_to_nhwc4(x):
    """(N,3,H,W) -> (N,H,W,4), zero 4th channel."""
    N, _, H, W = x.shape
    y = torch.empty((N, H, W, 4), device=x.device, dtype=torch.float32)
    _lib.check(_lib.load().rcmvs_rgb_to_nhwc4(_chk(x, "x"), _chk(y, "y"), N, H, W, _stream()), "rgb_to_nhwc4")
    return y


def pack_conv2d_weight(w, pad_in_to=None):
    """(Co,Ci,K,K) -> PackedWeight ([K*K][Cip][Co])."""
    w = w.detach().contiguous().float()
    Co, Ci, K, _ = w.shape
    Cip = Ci if pad_in_to is None else pad_in_to
    blob = torch.empty((K * K * Cip * Co,), device=w.device, dtype=torch.float32)
    _lib.check(_lib.load().rcmvs_pack_conv2d_weight(_chk(w, "w"), _chk(blob, "packed"), Co, Ci, Cip, K, _stream()), "pack_conv2d_weight")
    pw = PackedWeight(blob, Cip, Co)
    pw.k = K
    return pw


def conv2d(x, w_packed, scale=None, shift=None, up_add=None, stride=1, relu=False):
    """x (N,H,W,Ci) -> (N,Ho,Wo,Co) = [relu](up2(up_add) + conv*scale + shift)."""
    N, H, W, Ci = x.shape
    K, Co = w_packed.k, w_packed.co
    if w_packed.ci != Ci:
        raise _lib.RcmvsError(f"conv2d: input has {Ci} channels, weight expects {w_packed.ci}")
    Ho, Wo = (H + 2 * (K // 2) - K) // stride + 1, (W + 2 * (K // 2) - K) // stride + 1
    y = torch.empty((N, Ho, Wo, Co), device=x.device, dtype=torch.float32)
    if up_add is not None and tuple(up_add.shape) != (N, Ho // 2, Wo // 2, Co):
        raise _lib.RcmvsError(f"conv2d: up_add {tuple(up_add.shape)} does not match half of the output {tuple(y.shape)}")
    _lib.check(_lib.load().rcmvs_conv2d_fwd(_chk(x, "x"), _chk(w_packed.blob, "w"), _opt(scale, "scale"), _opt(shift, "shift"),
                                            _opt(up_add, "up_add"), _chk(y, "y"), N, H, W, Ci, Co, K, stride, int(relu), _stream()),
               "conv2d_fwd")
    return y


def conv1x1(x, w_packed, scale=None, shift=None, up_add=None, relu=False, ysq_absmax=None, mfma=False):
    """x (N,H,W,Ci) -> (N,H,W,Co) = [relu](up2(up_add) + conv1x1 * scale + shift) on the streaming 1x1 kernel (what conv2d routes its K = 1
    layers to); ysq_absmax: a zero-filled (1024,) bound vector that receives (max|y|)^2.  mfma=True: the same layer on the matrix cores
    (16 -> 32 and 32 -> 32; exact split operands, equal up to fp32 summation order)."""
    N, H, W, Ci = x.shape
    Co = w_packed.co
    if w_packed.ci != Ci or w_packed.k != 1:
        raise _lib.RcmvsError(f"conv1x1: input has {Ci} channels, weight is {w_packed.ci} -> {Co} with k={w_packed.k}")
    if up_add is not None and tuple(up_add.shape) != (N, H // 2, W // 2, Co):
        raise _lib.RcmvsError(f"conv1x1: up_add {tuple(up_add.shape)} does not match half of the output")
    y = torch.empty((N, H, W, Co), device=x.device, dtype=torch.float32)
    fn = _lib.load().rcmvs_conv1x1_mfma_fwd if mfma else _lib.load().rcmvs_conv1x1_fwd
    _lib.check(fn(_chk(x, "x"), _chk(w_packed.blob, "w"), _opt(scale, "scale"), _opt(shift, "shift"), _opt(up_add, "up_add"),
                  _chk(y, "y"), _opt(ysq_absmax, "ysq_absmax"), N, H, W, Ci, Co, int(relu), _stream()), "conv1x1_fwd")
    return y


def conv2d_rgb(x, w_packed, scale=None, shift=None, relu=False):
    """FeatureNet's first layer on the planar input itself: x (N,3,H,W) -> (N,H,W,8) channels-last; w_packed = the 3 -> 8 3x3 weight packed
    with pad_in_to=4 (rcmvs_conv2d_fwd with Ci = 3: the NCHW -> NHWC4 pass is folded into the tile staging)."""
    N, C, H, W = x.shape
    if C != 3 or w_packed.ci != 4 or w_packed.co != 8 or w_packed.k != 3:
        raise _lib.RcmvsError(f"conv2d_rgb: expected a (N,3,H,W) input and a 3 -> 8 3x3 weight packed to 4 input channels (got {tuple(x.shape)}, {w_packed.ci} -> {w_packed.co}, k={w_packed.k})")
    y = torch.empty((N, H, W, 8), device=x.device, dtype=torch.float32)
    _lib.check(_lib.load().rcmvs_conv2d_fwd(_chk(x, "x"), _chk(w_packed.blob, "w"), _opt(scale, "scale"), _opt(shift, "shift"), None,
                                            _chk(y, "y"), N, H, W, 3, 8, 3, 1, int(relu), _stream()), "conv2d_fwd")
    return y


def pack_conv2d_tile(w):
    """A (Co,32,3,3) Conv2d weight, Co = 16 or 32 -> the fragment image of conv2d_tile (csrc/conv2d_tile.hip)."""
    Co = w.shape[0]
    if Co not in (16, 32) or tuple(w.shape) != (Co, 32, 3, 3):
        raise _lib.RcmvsError("pack_conv2d_tile: built for 32 -> 16 and 32 -> 32 3x3 layers")
    lib = _lib.load()
    img = torch.empty(int(lib.rcmvs_conv2d_tile_weight_floats(Co)), device=w.device, dtype=torch.float32)
    _lib.check(lib.rcmvs_pack_conv2d_tile(_chk(w.detach().float().contiguous(), "w"), _chk(img, "image"), Co, _stream()), "pack_conv2d_tile")
    return img


def conv2d_tile(x, image, scale=None, shift=None, relu=False, s2d=False, ysq_absmax=None):
    """32 -> Co 3x3 conv on a tile kernel (Co = 16 or 32, read off the image's size): x (N,H,W,32) -> (N,H,W,Co), or (s2d=True, Co = 16) x (N,2H,2W,8) read through
    its space-to-depth view (a 5x5 stride-2 layer as a 3x3 one, weight re-indexed as FeatureNet._w5s2 does).  ysq_absmax: zero-filled bound vector that receives (max|y|)^2."""
    N, Hx, Wx, C = x.shape
    lib = _lib.load()
    Co = {int(lib.rcmvs_conv2d_tile_weight_floats(c)): c for c in (16, 32)}.get(image.numel())
    if Co is None or C != (8 if s2d else 32) or (s2d and (Hx % 2 or Wx % 2 or Co != 16)):
        raise _lib.RcmvsError(f"conv2d_tile: unexpected input {tuple(x.shape)} (s2d={s2d}) or image of {image.numel()} floats")
    H, W = (Hx // 2, Wx // 2) if s2d else (Hx, Wx)
    y = torch.empty((N, H, W, Co), device=x.device, dtype=torch.float32)
    _lib.check(lib.rcmvs_conv2d_tile_fwd(_chk(x, "x"), _chk(image, "image"), _opt(scale, "scale"), _opt(shift, "shift"), _chk(y, "y"), N, H, W, Co, int(s2d), int(relu),
                                         _opt(ysq_absmax, "ysq_absmax"), _stream()), "conv2d_tile_fwd")
    return y


def pack_conv2d_stem(wb):
    """The (8,8,3,3) Conv2d weight of FeatureNet's conv0.1 -> the fragment image of conv2d_stem (csrc/conv2d_stem.hip)."""
    if tuple(wb.shape) != (8, 8, 3, 3):
        raise _lib.RcmvsError("pack_conv2d_stem: built for an 8 -> 8 3x3 second layer")
    lib = _lib.load()
    img = torch.empty(int(lib.rcmvs_conv2d_stem_weight_floats()), device=wb.device, dtype=torch.float32)
    _lib.check(lib.rcmvs_pack_conv2d_stem(_chk(wb.detach().float().contiguous(), "wb"), _chk(img, "image"), _stream()), "pack_conv2d_stem")
    return img


def conv2d_stem(x, w_a_packed, scale_a, shift_a, image_b, scale_b, shift_b):
    """FeatureNet's conv0 block in one launch: x (N,3,H,W) planar -> (N,H,W,8) channels-last = relu(bn_b(conv_b(relu(bn_a(conv_a(x)))))).
    w_a_packed: the 3 -> 8 3x3 weight packed with pad_in_to=4 (pack_conv2d_weight); image_b: pack_conv2d_stem."""
    N, C, H, W = x.shape
    if C != 3 or w_a_packed.ci != 4 or w_a_packed.co != 8 or w_a_packed.k != 3:
        raise _lib.RcmvsError(f"conv2d_stem: expected a (N,3,H,W) input and a 3 -> 8 3x3 weight packed to 4 input channels (got {tuple(x.shape)}, {w_a_packed.ci} -> {w_a_packed.co}, k={w_a_packed.k})")
    y = torch.empty((N, H, W, 8), device=x.device, dtype=torch.float32)
    _lib.check(_lib.load().rcmvs_conv2d_stem_fwd(_chk(x, "x"), _chk(w_a_packed.blob, "w_a"), _chk(scale_a, "scale_a"), _chk(shift_a, "shift_a"), _chk(image_b, "image_b"),
                                                 _chk(scale_b, "scale_b"), _chk(shift_b, "shift_b"), _chk(y, "y"), N, H, W, _stream()), "conv2d_stem_fwd")
    return y


def fpn_out_fused(lat, up, w_inner_packed, b_inner, w_out_packed):
    """conv3x3(up2(up) + conv1x1(lat) + bias): lat (N,H,W,8), up (N,H/2,W/2,32) -> (N,H,W,8), the 32-channel merge never stored."""
    N, H, W, CL = lat.shape
    CM, CO = w_inner_packed.co, w_out_packed.co
    if w_inner_packed.ci != CL or w_inner_packed.k != 1 or w_out_packed.ci != CM or w_out_packed.k != 3:
        raise _lib.RcmvsError("fpn_out_fused: expected a 1x1 lateral conv followed by a 3x3 output conv on matching channels")
    if tuple(up.shape) != (N, H // 2, W // 2, CM):
        raise _lib.RcmvsError(f"fpn_out_fused: up {tuple(up.shape)} does not match half of {tuple(lat.shape)} with {CM} channels")
    y = torch.empty((N, H, W, CO), device=lat.device, dtype=torch.float32)
    _lib.check(_lib.load().rcmvs_fpn_out_fused(_chk(lat, "lat"), _chk(up, "up"), _chk(w_inner_packed.blob, "w_inner"), _chk(b_inner, "b_inner"),
                                               _chk(w_out_packed.blob, "w_out"), _chk(y, "y"), N, H, W, CL, CM, CO, _stream()), "fpn_out_fused")
    return y


FPN_FOLDED_FLOATS = 4744      # RCMVS_FPN_FOLDED_FLOATS of include/rcmvs.h


def pack_fpn_folded(w_inner, b_inner, w_out):
    """Tables of rcmvs_fpn_out_folded from the modules' weights: w_inner (32,8,1,1), b_inner (32,), w_out (8,32,3,3) -> (4744,) fp32.
    The products are formed in fp64 and rounded once."""
    wi = w_inner.detach().double().reshape(32, 8)                      # [cm][ci]
    wo = w_out.detach().double()                                       # [co][cm][ky][kx]
    b = b_inner.detach().double()
    wb = torch.einsum("mi,omyx->yxio", wi, wo)                         # [ky][kx][ci][co]
    bt = torch.einsum("m,omyx->yxo", b, wo)                            # [ky][kx][co]
    inside = {0: (1, 2), 1: (0, 1, 2), 2: (0, 1)}                     # taps inside the image for the first / an interior / the last row (column)
    bs = torch.stack([torch.stack([sum(bt[ky, kx] for ky in inside[cy] for kx in inside[cx]) for cx in range(3)]) for cy in range(3)])
    rows = lambda p: ((0,), (1, 2)) if p == 0 else ((0, 1), (2,))     # 3x3 taps that land on source row r of the 2x2 block, per output parity
    wa = torch.zeros(2, 2, 2, 2, 32, 8, dtype=torch.float64, device=wo.device)
    for py in range(2):
        for px in range(2):
            for ry in range(2):
                for rx in range(2):
                    for ky in rows(py)[ry]:
                        for kx in rows(px)[rx]:
                            wa[py, px, ry, rx] += wo[:, :, ky, kx].t()
    tab = torch.cat((wb.reshape(-1), bs.reshape(-1), wa.reshape(-1))).float().contiguous()
    assert tab.numel() == FPN_FOLDED_FLOATS
    return tab


def pack_fpn_folded_mfma(tables):
    """The tables of pack_fpn_folded as the weight image of the matrix-core form of the level (rcmvs_fpn_folded_mfma_pack: A fragments,
    three bf16 pieces per weight -- exact)."""
    if tables.numel() != FPN_FOLDED_FLOATS:
        raise _lib.RcmvsError("pack_fpn_folded_mfma: expects the tables of pack_fpn_folded")
    img = torch.empty((_lib.load().rcmvs_fpn_folded_mfma_floats(),), device=tables.device, dtype=torch.float32)
    _lib.check(_lib.load().rcmvs_fpn_folded_mfma_pack(_chk(tables, "tables"), _chk(img, "image"), _stream()), "fpn_folded_mfma_pack")
    return img


def fpn_out_folded(lat, up, tables, ysq_absmax=None):
    """conv3x3(up2(up) + conv1x1(lat) + bias) with the two convolutions folded: lat (N,H,W,8), up (N,H/2,W/2,32) -> (N,H,W,8).
    tables: pack_fpn_folded's fp32 tables (rcmvs_fpn_out_folded: fp32 FMA chains) or pack_fpn_folded_mfma's image
    (rcmvs_fpn_out_folded_mfma: the matrix cores, exact split operands).  ysq_absmax: a zero-filled (1024,) bound vector that receives (max|y|)^2."""
    N, H, W, CL = lat.shape
    mfma = tables.numel() == _lib.load().rcmvs_fpn_folded_mfma_floats()
    if CL != 8 or tuple(up.shape) != (N, H // 2, W // 2, 32) or not (mfma or tables.numel() == FPN_FOLDED_FLOATS):
        raise _lib.RcmvsError(f"fpn_out_folded: lat {tuple(lat.shape)} / up {tuple(up.shape)} / {tables.numel()} table floats do not fit 8 -> 32 -> 8")
    y = torch.empty((N, H, W, 8), device=lat.device, dtype=torch.float32)
    fn = _lib.load().rcmvs_fpn_out_folded_mfma if mfma else _lib.load().rcmvs_fpn_out_folded
    _lib.check(fn(_chk(lat, "lat"), _chk(up, "up"), _chk(tables, "tables"), _chk(y, "y"), _opt(ysq_absmax, "ysq_absmax"), N, H, W, _stream()), "fpn_out_folded")
    return y


# ------------------------------------------------------------------------------- K4
DEPTH_HEAD_IMPL = 0          # tests / A-B timing: bit 0 = two launches also for D = 8, bit 1 = the generic marching prob conv,
                             # bit 2 = the fp32 form although a bound was given, bits 8-15 = z chunk of the prob conv


def depth_head(x8, w_prob_packed, planes, want_prob=False, x_absmax=None):
    """x8 (B,D,h,w,8) -> depth (B,h,w), confidence (B,h,w)[, prob (B,D,h,w)].  With D = 8 (the cascade's last stage) the head is one
    launch; the probability volume is then stored only when asked for.  x_absmax (a bound of max|x8|, ops.absmax format or the
    y_absmax of the layer that produced x8): the prob conv runs on the matrix cores in fp16 pairs (csrc/prob_pair.hip)."""
    B, D, h, w, C = x8.shape
    if C != 8:
        raise _lib.RcmvsError("depth_head: the prob conv takes 8 channels")
    if x_absmax is not None and w_prob_packed.images != IMG_ALL:
        raise _lib.RcmvsError("depth_head: the fp16-pair form reads an image that a selectively packed weight does not hold")
    depth = torch.empty((B, h, w), device=x8.device, dtype=torch.float32)
    conf = torch.empty((B, h, w), device=x8.device, dtype=torch.float32)
    one_launch = D == 8 and not (DEPTH_HEAD_IMPL & 1)
    prob = None if one_launch and not want_prob else torch.empty((B, D, h, w), device=x8.device, dtype=torch.float32)     # logit scratch -> probabilities
    _lib.check(_lib.load().rcmvs_depth_head_scaled_fwd(_chk(x8, "x8"), _opt(x_absmax, "x_absmax"), _chk(w_prob_packed.blob, "w_prob"), _chk(planes, "planes"),
                                                       _chk(depth, "depth"), _chk(conf, "conf"), _opt(prob, "prob"), B, D, h, w,
                                                       DEPTH_HEAD_IMPL | (0 if want_prob else 8), _stream()),
               "depth_head_scaled_fwd")
    return (depth, conf, prob) if want_prob else (depth, conf)


def conv11_prob(t, t_absmax, w11_packed, scale, shift, res, res_absmax, coef, w_prob_packed, zchunk=0, planes=None):
    """The last transposed layer (16 -> 8, BatchNorm, ReLU, + skip `res`) and the prob conv in one pass (csrc/conv11_prob.hip):
    t (B,Dt,Ht,Wt,16), res (B,2Dt,2Ht,2Wt,8) -> logits (B,2Dt,2Ht,2Wt).  coef: (2,) device floats {c1, c2} (include/rcmvs.h).
    planes (B,2Ht,2Wt,2) with 2 Dt = 8: the whole head in the launch -> (depth, conf) instead of the logits."""
    B, Dt, Ht, Wt, C = t.shape
    if C != 16 or tuple(res.shape) != (B, 2 * Dt, 2 * Ht, 2 * Wt, 8):
        raise _lib.RcmvsError(f"conv11_prob: t {tuple(t.shape)} / res {tuple(res.shape)} are not (B,Dt,Ht,Wt,16) / (B,2Dt,2Ht,2Wt,8)")
    if w11_packed.images != IMG_ALL or w_prob_packed.images != IMG_ALL:
        raise _lib.RcmvsError("conv11_prob: needs the full weight blobs (a selectively packed weight does not hold the fp16-pair images)")
    fused = planes is not None and Dt == 4
    logits = depth = conf = None
    if fused:
        depth = torch.empty((B, 2 * Ht, 2 * Wt), device=t.device, dtype=torch.float32)
        conf = torch.empty((B, 2 * Ht, 2 * Wt), device=t.device, dtype=torch.float32)
    else:
        logits = torch.empty((B, 2 * Dt, 2 * Ht, 2 * Wt), device=t.device, dtype=torch.float32)
    _lib.check(_lib.load().rcmvs_conv11_prob_fwd(_chk(t, "t"), _chk(t_absmax, "t_absmax"), _chk(w11_packed.blob, "w11"), _chk(scale, "scale"),
                                                 _chk(shift, "shift"), _chk(res, "res"), _chk(res_absmax, "res_absmax"), _chk(coef, "coef"),
                                                 _chk(w_prob_packed.blob, "w_prob"), _opt(logits, "logits"), _opt(planes if fused else None, "planes"),
                                                 _opt(depth, "depth"), _opt(conf, "conf"), B, Dt, Ht, Wt, int(zchunk), _stream()),
               "conv11_prob_fwd")
    if fused:
        return depth, conf
    if planes is not None:
        return softmax_head(logits, planes)
    return logits


def softmax_head(logits, planes, keep_prob=False):
    """logits (B,D,h,w) -> depth (B,h,w), confidence (B,h,w); keep_prob: the logits tensor holds the probabilities afterwards."""
    B, D, h, w = logits.shape
    depth = torch.empty((B, h, w), device=logits.device, dtype=torch.float32)
    conf = torch.empty((B, h, w), device=logits.device, dtype=torch.float32)
    _lib.check(_lib.load().rcmvs_softmax_head_fwd(_chk(logits, "logits"), _chk(planes, "planes"), _chk(depth, "depth"), _chk(conf, "conf"),
                                                  B, D, h, w, int(bool(keep_prob)), _stream()), "softmax_head_fwd")
    return depth, conf


# ------------------------------------------------------------------------------- rendering branch
def resize_planes(x, out_planes, pad_channels_to=None):
    """x (B,C,D,h,w) NCDHW -> (B,out_planes,h,w,Cp) channels-last, trilinear along D, align_corners=True."""
    B, C, D, h, w = x.shape
    Cp = C if pad_channels_to is None else pad_channels_to
    y = torch.empty((B, out_planes, h, w, Cp), device=x.device, dtype=torch.float32)
    _lib.check(_lib.load().rcmvs_resize_planes_fwd(_chk(x, "x"), _chk(y, "y"), B, C, Cp, D, out_planes, h, w, _stream()),
               "resize_planes_fwd")
    return y


def gu_sample(pseudo_depth, img0, pix, eps, u, cam):
    """Gaussian-Uniform sampler.  pseudo_depth (H,W); img0 (3,H,W); pix (2,N) int32; eps (N,S); u (N/2,S); cam (52,)."""
    H, W = pseudo_depth.shape
    N, S = eps.shape
    dev = eps.device
    f = lambda *shape: torch.empty(shape, device=dev, dtype=torch.float32)
    z, pts, ndc, dirs, rdepth, target = f(N, S), f(N, S, 3), f(N, S, 3), f(N, 3), f(N), f(N, 3)
    _lib.check(_lib.load().rcmvs_gu_sample_fwd(_chk(pseudo_depth, "pseudo_depth"), _chk(img0, "img0"), _chk(pix, "pix", torch.int32),
                                               _chk(eps, "eps"), _chk(u, "u"), _chk(cam, "cam"), _chk(z, "z"), _chk(pts, "pts"),
                                               _chk(ndc, "ndc"), _chk(dirs, "dirs"), _chk(rdepth, "rays_depth"), _chk(target, "target"),
                                               N, S, H, W, _stream()), "gu_sample_fwd")
    return z, pts, ndc, dirs, rdepth, target


def point_feats(volume_cl, imgs, poses, pts, ndc, ldf=32):
    """volume_cl (Dv,hv,wv,8); imgs (nimg,3,H,W); poses (nimg,25); pts/ndc (N,S,3) -> feat (N*S, ldf), 8+4*nimg cols used."""
    Dv, hv, wv, C = volume_cl.shape
    if C != 8:
        raise _lib.RcmvsError("point_feats: the neural volume has 8 channels")
    nimg, _, H, W = imgs.shape
    M = pts.shape[0] * pts.shape[1]
    feat = torch.empty((M, ldf), device=pts.device, dtype=torch.float32)
    _lib.check(_lib.load().rcmvs_point_feats_fwd(_chk(volume_cl, "volume"), _chk(imgs, "imgs"), _chk(poses, "poses"), _chk(pts, "pts"),
                                                 _chk(ndc, "ndc"), _chk(feat, "feat"), M, Dv, hv, wv, nimg, H, W, ldf, _stream()),
               "point_feats_fwd")
    return feat


NERF_ORDER = ("pts_bias", "pts_linears.0", "pts_linears.1", "pts_linears.2", "pts_linears.3", "pts_linears.4", "pts_linears.5",
              "alpha_linear", "feature_linear", "views_linears.0", "rgb_linear")


def pack_nerf_weights(named):
    """named: {name: (weight, bias)} for NERF_ORDER -> packed blob."""
    lib = _lib.load()
    tensors = []
    for n in NERF_ORDER:
        w, b = named[n]
        tensors += [w.detach().contiguous().float(), b.detach().contiguous().float()]
    arr = (ctypes.c_void_p * 22)(*[_chk(t, "nerf weight").value for t in tensors])
    blob = torch.empty((lib.rcmvs_nerf_weight_floats(),), device=tensors[0].device, dtype=torch.float32)
    _lib.check(lib.rcmvs_pack_nerf_weights(arr, _chk(blob, "blob"), _stream()), "pack_nerf_weights")
    return blob


def nerf_mlp(ndc, feat, dirs, w2c_ref, blob):
    """ndc (N,S,3), feat (N*S,32), dirs (N,3), w2c_ref (4,4) -> raw (N,S,4) = [rgb, sigma]."""
    lib = _lib.load()
    N, S = ndc.shape[:2]
    M = N * S
    ws = torch.empty((lib.rcmvs_nerf_workspace_floats(M),), device=ndc.device, dtype=torch.float32)
    raw = torch.empty((N, S, 4), device=ndc.device, dtype=torch.float32)
    _lib.check(lib.rcmvs_nerf_mlp_fwd(_chk(ndc, "ndc"), _chk(feat, "feat"), feat.shape[1], _chk(dirs, "dirs"), _chk(w2c_ref, "w2c_ref"),
                                      _chk(blob, "weights"), _chk(ws, "workspace"), _chk(raw, "raw"), N, S, _stream()), "nerf_mlp_fwd")
    return raw


def nerf_mlp_embedded(x, blob):
    """Renderer_ours.forward on its own: x (M, 86) rows [embedded point 63 | feature 20 | view direction 3] -> raw (M, 4) = [rgb, sigma]."""
    lib = _lib.load()
    M, ldx = x.shape
    if ldx != 86:
        raise _lib.RcmvsError(f"nerf_mlp_embedded: rows of 63 + 20 + 3 = 86 columns expected (got {ldx})")
    ws = torch.empty((lib.rcmvs_nerf_workspace_floats(M),), device=x.device, dtype=torch.float32)
    feat = torch.empty((M, 32), device=x.device, dtype=torch.float32)
    raw = torch.empty((M, 4), device=x.device, dtype=torch.float32)
    _lib.check(lib.rcmvs_nerf_mlp_embedded_fwd(_chk(x, "x"), ldx, _chk(blob, "weights"), _chk(ws, "workspace"), _chk(feat, "feat"), _chk(raw, "raw"), M, _stream()),
               "nerf_mlp_embedded_fwd")
    return raw


def composite(raw, z):
    """raw (N,S,4), z (N,S) -> rgb (N,3), depth (N), weights (N,S), alpha (N,S)."""
    N, S = z.shape
    dev = z.device
    rgb = torch.empty((N, 3), device=dev, dtype=torch.float32)
    depth = torch.empty((N,), device=dev, dtype=torch.float32)
    weights = torch.empty((N, S), device=dev, dtype=torch.float32)
    alpha = torch.empty((N, S), device=dev, dtype=torch.float32)
    _lib.check(_lib.load().rcmvs_composite_fwd(_chk(raw, "raw"), _chk(z, "z"), _chk(rgb, "rgb"), _chk(depth, "depth"),
                                               _chk(weights, "weights"), _chk(alpha, "alpha"), N, S, _stream()), "composite_fwd")
    return rgb, depth, weights, alpha
