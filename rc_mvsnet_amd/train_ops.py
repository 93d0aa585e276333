"""Training-mode building blocks on the HIP kernels: autograd Functions whose forward AND backward are
library calls (include/rcmvs.h, section "training"), so that ``loss.backward()`` through the cost
regularisation never touches an ATen convolution / batch-norm kernel.

    ConvBnReluFn      Conv3d / Deconv3d block in train mode (models/modules.py:149-157,196-204):
                      conv -> BatchNorm3d with batch statistics -> ReLU [+ skip], channels-last
    ProbDepthHeadFn   prob conv + softmax + soft-argmin depth (+ confidence, no grad)
                      (models/modules.py:489,500, models/casmvsnet.py:103-122)
    (the differentiable fused warp + variance is ops.WarpVarianceFn)

PyTorch's role: it owns the tensors, runs the tiny per-channel vector arithmetic that turns the fp64 sums
into mean / invstd / running statistics (a few dozen floats), and all-reduces those sums when the module is
a SyncBatchNorm replica.
"""
import ctypes
import weakref

import torch
import torch.distributed as dist

from . import _lib, ops
from .ops import _chk, _opt, _stream


# --------------------------------------------------------------------------------------- thin wrappers
def bn_stats(x, sums):
    C = x.shape[-1]
    _lib.check(_lib.load().rcmvs_bn_stats(_chk(x, "x"), _chk(sums, "sums", torch.float64), x.numel() // C, C, _stream()), "bn_stats")


def scale_shift_relu(x, scale, shift, residual, relu, out=None):
    C = x.shape[-1]
    y = torch.empty_like(x) if out is None else out
    _lib.check(_lib.load().rcmvs_scale_shift_relu(_chk(x, "x"), _opt(scale, "scale"), _opt(shift, "shift"), _opt(residual, "residual"),
                                                  _chk(y, "y"), x.numel() // C, C, int(bool(relu)), _stream()), "scale_shift_relu")
    return y


def bn_bwd_reduce(y, dz, scale, shift, mean, invstd, sums, relu):
    C = y.shape[-1]
    _lib.check(_lib.load().rcmvs_bn_bwd_reduce(_chk(y, "y"), _chk(dz, "dz"), _chk(scale, "scale"), _chk(shift, "shift"),
                                               _chk(mean, "mean"), _chk(invstd, "invstd"), _chk(sums, "sums", torch.float64),
                                               y.numel() // C, C, int(bool(relu)), _stream()), "bn_bwd_reduce")


def bn_bwd_apply(y, dz, scale, shift, mean, invstd, coef, relu, out=None):
    C = y.shape[-1]
    dy = torch.empty_like(y) if out is None else out
    _lib.check(_lib.load().rcmvs_bn_bwd_apply(_chk(y, "y"), _chk(dz, "dz"), _chk(scale, "scale"), _chk(shift, "shift"),
                                              _chk(mean, "mean"), _chk(invstd, "invstd"), _chk(coef, "coef"), _chk(dy, "dy"),
                                              y.numel() // C, C, int(bool(relu)), _stream()), "bn_bwd_apply")
    return dy


def conv3d_wgrad(x, dy, stride, out=None):
    """x (B,D,H,W,Ci), dy (B,Do,Ho,Wo,Co) -> packed weight gradient (27,Ci,Co); `out`: a ZERO (27,Ci,Co) buffer to accumulate into."""
    B, D, H, W, Ci = x.shape
    Co = dy.shape[-1]
    exp = (B, (D - 1) // stride + 1, (H - 1) // stride + 1, (W - 1) // stride + 1)
    if tuple(dy.shape[:4]) != exp:
        raise _lib.RcmvsError(f"conv3d_wgrad: dy {tuple(dy.shape)} does not match x {tuple(x.shape)} at stride {stride}")
    dw = torch.zeros((27, Ci, Co), device=x.device, dtype=torch.float32) if out is None else out
    _lib.check(_lib.load().rcmvs_conv3d_wgrad(_chk(x, "x"), _chk(dy, "dy"), _chk(dw, "dw"), B, D, H, W, Ci, Co, stride, _stream()),
               "conv3d_wgrad")
    return dw


# packed-gradient accumulation buffers by (shape, device): [buffer, known-zero flag]; zero when created, re-zeroed by rcmvs_wgrad_finish
# after every use (stream-ordered); a use that aborts before its finish launch leaves the flag down and the next use fills the buffer first
_WGRAD_SCRATCH = {}


def _wgrad_to_param_layout(big, small, stride, w_shape):
    """Weight gradient in the parameter's layout (w_shape = (Q, Pk, 3, 3, 3)): accumulate [27][P][Q] into the persistent zero buffer
    of that shape, then one launch that permutes it out and clears the buffer (no zero fill, no permute copy)."""
    P, Q = big.shape[-1], small.shape[-1]
    key = (P, Q, str(big.device))
    ent = _WGRAD_SCRATCH.get(key)
    if ent is None:
        ent = _WGRAD_SCRATCH[key] = [torch.zeros((27, P, Q), device=big.device, dtype=torch.float32), True]
    buf = ent[0]
    if not ent[1]:                   # the previous use of this shape aborted before its finish launch cleared the buffer
        buf.zero_()
    ent[1] = False
    conv3d_wgrad(big, small, stride, out=buf)
    out = torch.empty(tuple(w_shape), device=big.device, dtype=torch.float32)
    _lib.check(_lib.load().rcmvs_wgrad_finish(_chk(buf, "packed"), _chk(out, "dw"), P, Q, int(w_shape[1]), _stream()), "wgrad_finish")
    ent[1] = True
    return out


def conv3d_dgrad_c1(dy, w):
    """dy (B,D,H,W) gradient of the 1-channel prob conv output, w (1,Ci,3,3,3) -> dx (B,D,H,W,Ci)."""
    B, D, H, W = dy.shape
    Ci = w.shape[1]
    dx = torch.empty((B, D, H, W, Ci), device=dy.device, dtype=torch.float32)
    _lib.check(_lib.load().rcmvs_conv3d_dgrad_c1(_chk(dy, "dy"), _chk(w, "w"), _chk(dx, "dx"), B, D, H, W, Ci, _stream()),
               "conv3d_dgrad_c1")
    return dx


def depth_head_bwd(prob, planes, depth, gdepth):
    B, D, h, w = prob.shape
    dl = torch.empty_like(prob)
    _lib.check(_lib.load().rcmvs_depth_head_bwd(_chk(prob, "prob"), _chk(planes, "planes"), _chk(depth, "depth"), _chk(gdepth, "gdepth"),
                                                _chk(dl, "dlogits"), B, D, h, w, _stream()), "depth_head_bwd")
    return dl


# --------------------------------------------------------------------------------------- conv + BN + ReLU
def _pad_in_channels(w, cx):
    """Conv weight (Co,Ci,3,3,3) -> (Co,cx,3,3,3): zero taps for the input's padding channels (41 -> 44)."""
    if w.shape[1] == cx:
        return w
    return torch.cat((w, w.new_zeros(w.shape[0], cx - w.shape[1], 3, 3, 3)), dim=1)


# Packed images of PARAMETERS, reused while the parameter is unchanged: an iteration runs two cascade passes (standard and augmented
# views) over the same weights, forward and backward.  An entry is valid while the parameter object is alive (weak reference: a model
# that is dropped takes its entries with it) and its autograd version counter, storage pointer and device are what they were -- an
# in-place optimizer update bumps the counter, `.to(device)` / a dtype change moves the storage.  A write through `.data`
# (`p.data.add_(...)`: legacy optimizers, EMA / weight-swap code) changes NONE of these, so the cache is ALSO emptied after every
# torch.optim.Optimizer.step() (a global post-step hook, whatever the optimizer subclass does inside); code that writes `.data`
# outside an optimizer step must call clear_pack_cache() itself.  Derived weight tensors (FeatureNet's
# embedded 2-D kernels, channel-padded copies) are new tensors every call and are packed every call.
_PACK_CACHE = {}
_PACK_HOOK = [None]


def clear_pack_cache():
    _PACK_CACHE.clear()


def _install_pack_hook():
    if _PACK_HOOK[0] is None:
        try:
            from torch.optim.optimizer import register_optimizer_step_post_hook
            _PACK_HOOK[0] = register_optimizer_step_post_hook(lambda opt, args, kwargs: _PACK_CACHE.clear())
        except Exception:                      # an older torch without global optimizer hooks: the version counter alone
            _PACK_HOOK[0] = False


def _param_of(w):
    return w if isinstance(w, torch.nn.Parameter) else None


def _packed(w, transposed, stride, planar, param=None):
    """PackedWeight of `w` holding the one image a production call (stride, planar) reads."""
    if param is None or ops._CONV_IMPL:
        return ops.pack_conv3d_weight(w, transposed=transposed, use=(stride, planar))
    _install_pack_hook()
    key = (int(transposed), int(stride), bool(planar))
    pid = id(param)
    state = (param._version, param.data_ptr(), param.device)
    ent = _PACK_CACHE.get(pid)
    if ent is None or ent[0]() is not param or ent[1] != state:
        ref = weakref.ref(param, lambda _, pid=pid: _PACK_CACHE.pop(pid, None))      # the id may be recycled once the parameter is gone
        ent = _PACK_CACHE[pid] = (ref, state, {})
    pk = ent[2].get(key)
    if pk is None:
        pk = ent[2][key] = ops.pack_conv3d_weight(w, transposed=transposed, use=(stride, planar))
    return pk


def _conv_raw(x, w, transposed, stride, param=None):
    """The block's convolution with an identity epilogue (weights re-packed when they change: every step)."""
    if transposed:
        return ops.deconv3d(x, _packed(w, True, 2, False, param))
    wp = _pad_in_channels(w, x.shape[-1])
    return ops.conv3d(x, _packed(wp, False, stride, stride == 1 and x.shape[1] == 1, param if wp.shape == w.shape else None), stride=stride)


def _conv_dgrad(dy, w, transposed, stride, cx, param=None):
    """d loss / d x of the block's convolution, on the forward kernels with re-packed weights (cx = channels of x)."""
    w = w.detach()
    if transposed:                                   # adjoint of ConvTranspose3d(stride 2) = Conv3d(stride 2), same weight tensor
        return ops.conv3d(dy, _packed(w, False, 2, False, param), stride=2)
    if stride == 2:                                  # adjoint of Conv3d(stride 2) = ConvTranspose3d(stride 2, output_padding 1)
        return ops.deconv3d(dy, _packed(w, True, 2, False, param))
    # adjoint of Conv3d(stride 1, pad 1) = Conv3d with flipped taps and swapped channel roles: pack mode 2 does both
    co = w.shape[1]                                  # output channels of the adjoint = input channels of the layer
    if co not in (1, 8) and co % 16:                 # e.g. 41 input channels: the MFMA kernels want a multiple of 16 outputs
        w = _pad_in_channels(w, (co + 15) // 16 * 16)
        param = None
    dx = ops.conv3d(dy, _packed(w, 2, 1, dy.shape[1] == 1, param), stride=1)
    if dx.shape[-1] != cx:                           # back to the (padded) channel count of x; padding channels get zero
        dx = dx[..., :cx].contiguous() if dx.shape[-1] > cx else torch.nn.functional.pad(dx, (0, cx - dx.shape[-1]))
    return dx


def _conv_wgrad(x, dy, w_shape, transposed, stride):
    if transposed:                                   # roles swap: the large tensor (dy) is strided over -> (27, Cout_T, Cin_T)
        return _wgrad_to_param_layout(dy, x, 2, w_shape)
    return _wgrad_to_param_layout(x, dy, stride, w_shape)       # (27, Cx, Co), Cx >= Ci when the input carries padding channels


def _bn_scratch(cfg, which, S, n, device):
    """-> (cur, other, done): the layer's two fp64 accumulation buffers (S segments x n doubles each), created zero once and kept on the
    BatchNorm module.  A call accumulates into `cur` and hands `other` -- the buffer the previous call of this layer consumed -- to
    the fused normalisation kernel, which clears it (rcmvs_bn_norm_fwd / _bwd): no call fills a buffer again.  Calls of one module are
    stream-ordered.  A call that ABORTS between its accumulation and its normalisation launch (out of memory with a skip-batch
    handler, KeyboardInterrupt, a failing all_reduce) would leave `cur` part-filled and `other` uncleared: each buffer therefore
    carries a host-side "known zero" flag -- cleared when a call starts accumulating into it, set by `done()` once the launch that
    clears it has been enqueued -- and a buffer that is not known to be zero is re-zeroed before use (one extra fill, after an abort only)."""
    store = cfg.get("scratch")
    if store is None:
        z = torch.zeros((2, S, n), device=device, dtype=torch.float64)
        return z[0], z[1], (lambda: None)
    key = (which, S, n, str(device))
    ent = store.get(key)
    if ent is None:
        ent = store[key] = [torch.zeros((2, S, n), device=device, dtype=torch.float64), 0, [True, True]]
    ent[1] ^= 1
    i = ent[1]
    if not ent[2][i]:
        ent[0][i].zero_()
    ent[2][i] = False

    def done(ent=ent, j=i ^ 1):
        ent[2][j] = True
    return ent[0][i], ent[0][i ^ 1], done


class ConvBnReluFn(torch.autograd.Function):
    """z = [relu](batchnorm_train(conv(x, w))) [+ residual], channels-last.  `cfg` = dict(transposed, stride, relu,
    eps, momentum, group, segments): group = a process group for SyncBatchNorm statistics, or None; segments = S splits
    the batch into S consecutive groups that are normalised independently, in order (the reference runs FeatureNet once
    per view, so each view has its own batch statistics and the running statistics are updated once per view) while the
    convolution and both of its gradients run once over the whole batch.  `running_mean` / `running_var` (may be None)
    receive nn.BatchNorm's momentum update inside the same kernel that folds the statistics."""

    @staticmethod
    def forward(ctx, x, w, gamma, beta, residual, running_mean, running_var, cfg):
        x = x.contiguous()
        lib = _lib.load()
        y = _conv_raw(x, w.detach(), cfg["transposed"], cfg["stride"], _param_of(w))
        C = y.shape[-1]
        S = int(cfg.get("segments", 1))
        nb = y.shape[0] // S
        pack, spent, done = _bn_scratch(cfg, "fwd", S, 2 * C + 1, x.device)      # per segment [sum | sum of squares | rows]; `spent`: the previous call's
        cnt = torch.empty((S,), device=x.device, dtype=torch.float64)           # rows behind the statistics (all ranks), for the backward pass
        stats = torch.empty((S, 5, C), device=x.device, dtype=torch.float32)     # mean, var, invstd, scale, shift
        g32, b32 = gamma.detach().float().contiguous(), beta.detach().float().contiguous()
        res = None if residual is None else residual.contiguous()
        z = torch.empty_like(y)
        ptr = lambda t: ctypes.c_void_p(t.data_ptr())
        for sgm in range(S):
            ys = y[sgm * nb:(sgm + 1) * nb]
            bn_stats(ys, pack[sgm])
            if cfg.get("group") is not None:
                dist.all_reduce(pack[sgm], group=cfg["group"])
            rs = None if res is None else res[sgm * nb:(sgm + 1) * nb]
            zs = z[sgm * nb:(sgm + 1) * nb]
            _lib.check(lib.rcmvs_bn_norm_fwd(_chk(ys, "y"), ptr(pack[sgm]), ptr(spent[sgm]), _chk(g32, "gamma"), _chk(b32, "beta"),
                                             float(cfg["eps"]), float(cfg.get("momentum", 0.0)), ptr(stats[sgm]), ptr(cnt[sgm:]),
                                             _opt(running_mean, "running_mean"), _opt(running_var, "running_var"), _opt(rs, "residual"),
                                             _chk(zs, "z"), ys.numel() // C, C, int(bool(cfg["relu"])), _stream()), "bn_norm_fwd")
        done()                                                                   # every segment's `spent` row has its clearing launch enqueued
        ctx.save_for_backward(x, w, y, stats, cnt)
        ctx.cfg = cfg
        ctx.has_res = residual is not None
        return z

    @staticmethod
    def backward(ctx, dz):
        x, w, y, stats, cnt = ctx.saved_tensors
        cfg = ctx.cfg
        C = y.shape[-1]
        S = int(cfg.get("segments", 1))
        nb = y.shape[0] // S
        dz = dz.contiguous()
        sums, spent, done = _bn_scratch(cfg, "bwd", S, 2 * C, y.device)
        out = torch.empty((S, 2, C), device=y.device, dtype=torch.float32)        # per segment: dgamma, dbeta
        dy = torch.empty_like(y)
        ptr = lambda t: ctypes.c_void_p(t.data_ptr())
        for sgm in range(S):
            sl = slice(sgm * nb, (sgm + 1) * nb)
            mean, invstd, scale, shift = stats[sgm, 0], stats[sgm, 2], stats[sgm, 3], stats[sgm, 4]
            bn_bwd_reduce(y[sl], dz[sl], scale, shift, mean, invstd, sums[sgm], cfg["relu"])
            tot = sums[sgm]
            if cfg.get("group") is not None:
                tot = sums[sgm].clone()
                dist.all_reduce(tot, group=cfg["group"])
            _lib.check(_lib.load().rcmvs_bn_norm_bwd(_chk(y[sl], "y"), _chk(dz[sl], "dz"), ptr(stats[sgm]), ptr(sums[sgm]), ptr(tot), ptr(cnt[sgm:]),
                                                     ptr(spent[sgm]), ptr(out[sgm, 0]), ptr(out[sgm, 1]), _chk(dy[sl], "dy"), y[sl].numel() // C, C,
                                                     int(bool(cfg["relu"])), _stream()), "bn_norm_bwd")
        done()
        dgamma, dbeta = (out[0, 0], out[0, 1]) if S == 1 else (out[:, 0].sum(0), out[:, 1].sum(0))
        dx = _conv_dgrad(dy, w, cfg["transposed"], cfg["stride"], x.shape[-1], _param_of(w)) if ctx.needs_input_grad[0] else None
        dw = _conv_wgrad(x, dy, w.shape, cfg["transposed"], cfg["stride"]) if ctx.needs_input_grad[1] else None
        return dx, dw, dgamma, dbeta, (dz if ctx.has_res else None), None, None, None


def conv_bn_relu_train(block, x, residual=None):
    """Run a Conv3d / Deconv3d module (casmvsnet.py) in train mode on the HIP kernels, including the
    running-statistics update of its BatchNorm3d / SyncBatchNorm (momentum semantics of torch.nn)."""
    return conv_bn_train(block.conv, block.bn, x, relu=bool(block.relu), residual=residual)


def conv_bn_train(conv, bn, x, relu, residual=None):
    """conv (nn.Conv3d | nn.ConvTranspose3d, k=3, pad 1, no bias) -> bn (batch statistics) -> [ReLU] [+ residual]."""
    return conv_bn_train_w(conv.weight, bn, x, relu, residual, isinstance(conv, torch.nn.ConvTranspose3d), conv.stride[0])


def conv_bn_train_w(weight, bn, x, relu, residual=None, transposed=False, stride=1, segments=1):
    """Same with an explicit (Co,Ci,3,3,3) weight tensor (it may be a differentiable function of the module's parameter,
    e.g. FeatureNet's 2-D weights embedded as one-plane 3-D kernels)."""
    group = None
    if isinstance(bn, torch.nn.SyncBatchNorm) and dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        group = bn.process_group if bn.process_group is not None else dist.group.WORLD
    track = bn.track_running_stats and bn.running_mean is not None
    mom = 0.0
    if track:
        with torch.no_grad():
            bn.num_batches_tracked += segments
        # momentum=None means a cumulative average; that needs the step count on the host (one sync) -- the reference
        # always sets a momentum (modules.py:146, bn_momentum=0.1), so this branch is cold
        mom = bn.momentum if bn.momentum is not None else 1.0 / float(bn.num_batches_tracked)
    scratch = bn.__dict__.get("_rcmvs_scratch")
    if scratch is None:
        scratch = bn.__dict__["_rcmvs_scratch"] = {}
    cfg = {"transposed": transposed, "stride": stride, "relu": bool(relu), "eps": bn.eps, "momentum": mom, "group": group,
           "segments": segments, "scratch": scratch}
    return ConvBnReluFn.apply(x, weight, bn.weight, bn.bias, residual, bn.running_mean if track else None,
                              bn.running_var if track else None, cfg)


class ConvPlainFn(torch.autograd.Function):
    """y = conv(x, w) [+ bias], no normalisation (FeatureNet's out* / inner* layers as one-plane volumes): forward and
    both gradients on the 3-D conv family."""

    @staticmethod
    def forward(ctx, x, w, bias):
        x = x.contiguous()
        ctx.save_for_backward(x, w)
        ctx.has_bias = bias is not None
        wp = _pad_in_channels(w.detach(), x.shape[-1])
        pk = _packed(wp, False, 1, x.shape[1] == 1, _param_of(w) if wp.shape == w.shape else None)
        if bias is None:
            return ops.conv3d(x, pk)
        b32 = bias.detach().float().contiguous()
        return ops.conv3d(x, pk, torch.ones_like(b32), b32)

    @staticmethod
    def backward(ctx, dy):
        x, w = ctx.saved_tensors
        dy = dy.contiguous()
        dx = _conv_dgrad(dy, w, False, 1, x.shape[-1], _param_of(w)) if ctx.needs_input_grad[0] else None
        dw = _conv_wgrad(x, dy, w.shape, False, 1) if ctx.needs_input_grad[1] else None
        db = dy.sum(dim=(0, 1, 2, 3)) if ctx.has_bias else None
        return dx, dw, db


# --------------------------------------------------------------------------------------- plane resize (renderer)
class ResizePlanesFn(torch.autograd.Function):
    """F.interpolate(trilinear, align_corners=True) along the plane axis (models/render_models.py:756): NCDHW in ->
    (B,Do,h,w,Cp) channels-last out (channels zero-padded to Cp), with the exact adjoint as backward."""

    @staticmethod
    def forward(ctx, x, out_planes, cp):
        ctx.shape = tuple(x.shape)
        return ops.resize_planes(x.contiguous().float(), out_planes, pad_channels_to=cp)

    @staticmethod
    def backward(ctx, g):
        B, C, D, h, w = ctx.shape
        g = g.contiguous()
        gx = torch.empty(ctx.shape, device=g.device, dtype=torch.float32)
        _lib.check(_lib.load().rcmvs_resize_planes_bwd(_chk(g, "g"), _chk(gx, "gx"), B, C, g.shape[-1], D, g.shape[1], h, w, _stream()),
                   "resize_planes_bwd")
        return gx, None, None


# --------------------------------------------------------------------------------------- rendering tail
class PointFeatsFn(torch.autograd.Function):
    """ops.point_feats with the gradient w.r.t. the neural volume (trilinear scatter); images, poses and point
    coordinates are data (render_utils.py:304-330 samples under the graph only through the volume)."""

    @staticmethod
    def forward(ctx, volume_cl, imgs, poses, pts, ndc, ldf):
        volume_cl = volume_cl.contiguous()
        ndc = ndc.contiguous()
        ctx.save_for_backward(ndc)
        ctx.vshape = tuple(volume_cl.shape)
        return ops.point_feats(volume_cl, imgs, poses, pts, ndc, ldf=ldf)

    @staticmethod
    def backward(ctx, gfeat):
        (ndc,) = ctx.saved_tensors
        Dv, hv, wv, _ = ctx.vshape
        gfeat = gfeat.contiguous()
        gvol = torch.zeros(ctx.vshape, device=gfeat.device, dtype=torch.float32)
        _lib.check(_lib.load().rcmvs_point_feats_bwd(_chk(ndc, "ndc"), _chk(gfeat, "grad_feat"), _chk(gvol, "grad_volume"),
                                                     gfeat.shape[0], Dv, hv, wv, gfeat.shape[1], _stream()), "point_feats_bwd")
        return gvol, None, None, None, None, None


class CompositeFn(torch.autograd.Function):
    """ops.composite (renderer.py:18-26,65-93) with its backward w.r.t. raw = [rgb, sigma]; z is data."""

    @staticmethod
    def forward(ctx, raw, z):
        raw, z = raw.contiguous(), z.contiguous()
        ctx.save_for_backward(raw, z)
        return ops.composite(raw, z)

    @staticmethod
    def backward(ctx, g_rgb, g_depth, g_w, g_alpha):
        raw, z = ctx.saved_tensors
        N, S = z.shape
        c = lambda t: None if t is None else t.contiguous()
        g_rgb, g_depth, g_w, g_alpha = c(g_rgb), c(g_depth), c(g_w), c(g_alpha)
        graw = torch.empty_like(raw)
        _lib.check(_lib.load().rcmvs_composite_bwd(_chk(raw, "raw"), _chk(z, "z"), _opt(g_rgb, "g_rgb"), _opt(g_depth, "g_depth"),
                                                   _opt(g_w, "g_w"), _opt(g_alpha, "g_alpha"), _chk(graw, "grad_raw"), N, S, _stream()),
                   "composite_bwd")
        return graw, None


class NerfMlpFn(torch.autograd.Function):
    """The NeRF MLP (Renderer_ours.forward through run_network_mvs, models/render_models.py:192-220, renderer.py:42-63)
    forward and backward on the MFMA kernels of csrc/nerf_mlp.hip: raw (N,S,4) = f(ndc, feat, dirs; 22 parameters),
    differentiable w.r.t. the point features (first 20 columns of the (M,32) buffer) and every weight / bias; ndc and the
    view directions are data.  Parameters are passed in ops.NERF_ORDER as (weight, bias) pairs."""

    @staticmethod
    def forward(ctx, ndc, feat, dirs, w2c_ref, *params):
        import ctypes
        lib = _lib.load()
        N, S = ndc.shape[:2]
        M = N * S
        ps = [p.detach().contiguous().float() for p in params]
        arr = (ctypes.c_void_p * 22)(*[_chk(t, "nerf weight").value for t in ps])
        blob = torch.empty((lib.rcmvs_nerf_weight_floats(),), device=ndc.device, dtype=torch.float32)
        _lib.check(lib.rcmvs_pack_nerf_weights(arr, _chk(blob, "blob"), _stream()), "pack_nerf_weights")
        feat = feat.detach().contiguous().clone()            # the forward zeroes its padding columns in place
        tws = torch.empty((lib.rcmvs_nerf_train_workspace_floats(M),), device=ndc.device, dtype=torch.float32)
        raw = torch.empty((N, S, 4), device=ndc.device, dtype=torch.float32)
        _lib.check(lib.rcmvs_nerf_mlp_train_fwd(_chk(ndc.contiguous(), "ndc"), _chk(feat, "feat"), feat.shape[1], _chk(dirs.contiguous(), "dirs"),
                                                _chk(w2c_ref.contiguous(), "w2c_ref"), _chk(blob, "weights"), _chk(tws, "workspace"),
                                                _chk(raw, "raw"), N, S, _stream()), "nerf_mlp_train_fwd")
        ctx.save_for_backward(feat, tws, raw, *ps)
        ctx.dims = (N, S)
        return raw

    @staticmethod
    def backward(ctx, graw):
        import ctypes
        lib = _lib.load()
        feat, tws, raw, *ps = ctx.saved_tensors
        N, S = ctx.dims
        M = N * S
        graw = graw.contiguous().float()
        key = (M, raw.device)              # backward scratch (415 MB at 1024 x 128 points): one buffer per (size, device), reused every step
        gws = _NERF_BWD_SCRATCH.pop(key, None)
        if gws is None:
            while len(_NERF_BWD_SCRATCH) >= _NERF_BWD_SCRATCH_MAX:          # least recently used first (dicts keep insertion order)
                _NERF_BWD_SCRATCH.pop(next(iter(_NERF_BWD_SCRATCH)))
            gws = torch.empty((lib.rcmvs_nerf_bwd_workspace_floats(M),), device=raw.device, dtype=torch.float32)
        _NERF_BWD_SCRATCH[key] = gws                                        # (re-)inserted as the most recent
        dfeat = torch.empty_like(feat)
        grads = [torch.empty_like(p) for p in ps]
        warr = (ctypes.c_void_p * 22)(*[_chk(t, "nerf weight").value for t in ps])
        garr = (ctypes.c_void_p * 22)(*[_chk(t, "nerf grad").value for t in grads])
        _lib.check(lib.rcmvs_nerf_mlp_bwd(warr, _chk(feat, "feat"), feat.shape[1], _chk(tws, "workspace"), _chk(raw, "raw"), _chk(graw, "grad_raw"),
                                          _chk(gws, "scratch"), _chk(dfeat, "grad_feat"), garr, N, S, _stream()), "nerf_mlp_bwd")
        return (None, dfeat, None, None, *grads)


_NERF_BWD_SCRATCH = {}             # (points, device) -> workspace, at most _NERF_BWD_SCRATCH_MAX entries (two devices / two ray counts alternate without re-allocating)
_NERF_BWD_SCRATCH_MAX = 4


def nerf_mlp_train(net, ndc, feat, dirs, w2c_ref):
    """net: Renderer_ours (use_viewdirs head).  feat (M,32) point-feature buffer (20 used columns)."""
    mods = {"pts_bias": net.pts_bias, "alpha_linear": net.alpha_linear, "feature_linear": net.feature_linear,
            "views_linears.0": net.views_linears[0], "rgb_linear": net.rgb_linear}
    for i in range(6):
        mods[f"pts_linears.{i}"] = net.pts_linears[i]
    params = []
    for n in ops.NERF_ORDER:
        params += [mods[n].weight, mods[n].bias]
    return NerfMlpFn.apply(ndc, feat, dirs, w2c_ref, *params)


# --------------------------------------------------------------------------------------- depth head
class ProbDepthHeadFn(torch.autograd.Function):
    """x8 (B,D,h,w,8), prob weight (1,8,3,3,3), planes (B,h,w,2) -> depth (B,h,w) [differentiable],
    photometric confidence (B,h,w) [the reference computes it under no_grad, casmvsnet.py:115]."""

    @staticmethod
    def forward(ctx, x8, w, planes):
        x8 = x8.contiguous()
        depth, conf, prob = ops.depth_head(x8, ops.pack_conv3d_weight(w.detach()), planes, want_prob=True)
        ctx.save_for_backward(x8, w, planes, prob, depth)
        ctx.mark_non_differentiable(conf)
        return depth, conf

    @staticmethod
    def backward(ctx, gdepth, _gconf):
        x8, w, planes, prob, depth = ctx.saved_tensors
        dl = depth_head_bwd(prob, planes, depth, gdepth.contiguous())
        dx = conv3d_dgrad_c1(dl, w.detach().contiguous()) if ctx.needs_input_grad[0] else None
        dw = None
        if ctx.needs_input_grad[1]:
            dw = conv3d_wgrad(x8, dl.unsqueeze(-1), 1).permute(2, 1, 0).reshape(w.shape)
        return dx, dw, None
