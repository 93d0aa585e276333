"""Drop-in modules for the reference's ``models.casmvsnet`` / ``models.modules`` surface:
``CascadeMVSNet``, ``CascadeMVSNet_eval`` (+ ``FeatureNet``, ``CostRegNet``, ``Conv3d`` ...).

Same constructor kwargs, forward signatures, output dict keys and ``state_dict`` names as
models/casmvsnet.py:126-231,313-417 and models/modules.py:28-210,363-501, so the reference's
``train_rcmvsnet.py`` / ``eval_rcmvsnet_*.py`` run unchanged and its checkpoints load strict.

Execution:
  * inference (module in eval mode under ``torch.no_grad()``, tensors on the GPU) runs entirely on the
    hand-written HIP kernels of librcmvs_hip.so: 2-D feature pyramid (conv2d.hip), fused warp+variance
    (K1), 3-D conv family with folded BatchNorm (K2/K3), prob-conv/softmax/regression/confidence (K4),
    all in channels-last layout, no host synchronisation anywhere in forward();
  * training (module in train mode on the GPU) runs the feature pyramid and the three cascade stages
    forward AND backward on the library through autograd Functions (ops.WarpVarianceFn,
    train_ops.ConvBnReluFn / ConvPlainFn / ProbDepthHeadFn: batch-statistics BatchNorm, data / weight
    gradients, K1 scatter); PyTorch supplies element-wise glue only;
  * anything else -- CPU tensors, eval mode with autograd enabled, a missing library -- raises RcmvsError: the modules are
    parameter holders (reference ``state_dict`` names) plus HIP execution plans, there is no PyTorch op graph behind them.
    (The reference's op graph with autograd lives in oracle/aten_graph.py, for the tests only.)
"""
import os

FP16_PAIR_DEFAULT = True       # inference CostRegNet on the fp16-pair matrix-core form unless RCMVS_FP16_PAIR=0 (the exact bf16 triple)
ONE_BY_ONE_MFMA = os.environ.get("RCMVS_1X1_MFMA", "1") != "0"  # FeatureNet's 32 -> 32 1x1 output conv (out1) on the matrix cores, exact split operands; 0 = fp32 FMA chains
HEAD_PAIR = os.environ.get("RCMVS_HEAD_PAIR", "1") != "0"       # ... and the depth head's prob conv (csrc/prob_pair.hip); 0 = fp32 FMA chains there
# the last transposed layer + the prob conv (+ the head, D = 8) in one pass (csrc/conv11_prob.hip): 1 = at the cascade's last stage (D = 8: 50 against 70 us on
# a DTU scene; at the other stages the two launches are as fast, profiles/r6_conv11_prob.txt), 2 = at every stage, 0 = never (the 8-channel volume in memory)
CONV11_PROB = int(os.environ.get("RCMVS_CONV11_PROB", "1"))
CONV_TILE = os.environ.get("RCMVS_CONV_TILE", "1") != "0"       # FeatureNet's conv1.0 (5x5 stride 2 as space-to-depth) and out2 (32 -> 16) on the tile kernel of csrc/conv2d_tile.hip; 0 = the planar kernel
CONV_STEM = os.environ.get("RCMVS_CONV_STEM", "1") != "0"       # FeatureNet's conv0.0 -> conv0.1 as one launch (csrc/conv2d_stem.hip); 0 = the tile kernel + the planar kernel
CONV_PAIR = os.environ.get("RCMVS_CONV_PAIR", "1") != "0"       # FeatureNet's conv1.1 -> conv1.2 as one launch (csrc/conv2d_pair.hip); 0 = two launches of the planar kernel
DEEP_PAIR = os.environ.get("RCMVS_DEEP_PAIR", "1") != "0"       # ... its deep levels (conv5-7) included (csrc/conv3d_deep.hip); 0 = fp32 MFMAs there

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import ops
from ._lib import RcmvsError


# ----------------------------------------------------------------------------------------------
# 2-D blocks (models/modules.py:28-116, 342-360).  Inside FeatureNet they are parameter holders (its HIP plan reads their conv / bn
# children); called on their own they run the same kernels one block at a time (eval mode, autograd off, NCHW tensors on the GPU).
# ----------------------------------------------------------------------------------------------
def _nhwc(x):
    """(N,C,H,W) -> channels-last; an RGB image gets a zero fourth channel (the first layer's kernel reads 4-channel pixels)."""
    x = x.contiguous().float()
    return ops.rgb_to_nhwc4(x) if x.shape[1] == 3 else ops.to_channels_last(x)


def _param(mod, name):
    """A parameter by name, dictionary-fast: nn.Module._parameters for a real module, the plain attribute for a DataParallel replica
    (torch.nn.parallel.replicate empties a replica's _parameters and sets the per-device copies as ordinary attributes)."""
    p = mod._parameters.get(name)
    return p if p is not None else getattr(mod, name)


def _packed_of(module, make):
    """Packed weight of a stand-alone block, rebuilt when its parameter changes."""
    w = module.conv.weight
    key = (w.data_ptr(), w._version, str(w.device))
    c = module.__dict__.get("_rcmvs_pack")
    if c is None or c[0] != key:
        c = module.__dict__["_rcmvs_pack"] = (key, make(w.detach()))
    return c[1]


def _affine_of(m):
    """What follows the convolution in eval mode as (scale, shift): the folded BatchNorm, or the conv's bias."""
    if getattr(m, "gn", None) is not None:
        raise RcmvsError(f"{type(m).__name__}: GroupNorm blocks are not provided (the reference builds every block with norm='batch_norm')")
    if m.bn is not None:
        return _bn_fold(m.bn)
    b = m.conv.bias
    if b is None:
        return None, None
    return torch.ones_like(b, dtype=torch.float32), b.detach().float().contiguous()


def _same2(v, n):
    return tuple(v) == (n, n)


def _conv2d_block_cl(m, t):
    """One Conv2d block (conv -> [BatchNorm] -> [ReLU], models/modules.py:56-62) on a channels-last map t (N,H,W,Ci).
    5x5 stride 2 and the RGB layer: the 2-D kernel (csrc/conv2d.hip); 1x1 / 3x3: a one-plane volume on the 3-D family."""
    conv = m.conv
    K = conv.kernel_size[0]
    st = conv.stride[0]
    if (not _same2(conv.kernel_size, K) or K not in (1, 3, 5) or not _same2(conv.padding, K // 2) or not _same2(conv.stride, st)
            or st not in (1, 2) or not _same2(conv.dilation, 1) or conv.groups != 1 or conv.padding_mode != "zeros"):
        raise RcmvsError(f"Conv2d: only square 1x1 / 3x3 / 5x5 kernels with 'same' zero padding, stride 1 or 2, no dilation or groups "
                         f"(the layers of models/modules.py) run on the HIP path; got {conv}")
    scale, shift = _affine_of(m)
    relu = bool(m.relu)
    if K == 5 or conv.in_channels == 3:
        pw = _packed_of(m, lambda w: ops.pack_conv2d_weight(w, pad_in_to=4 if w.shape[1] == 3 else None))
        return ops.conv2d(t, pw, scale, shift, stride=st, relu=relu)
    pw = _packed_of(m, lambda w: ops.pack_conv3d_weight(FeatureNet._w3(w)))
    return ops.conv3d(t.unsqueeze(1), pw, scale, shift, stride=st, relu=relu).squeeze(1)


def _deconv2d_block_cl(m, t):
    """One Deconv2d block (models/modules.py:100-110: ConvTranspose2d k3 s2 p1 op1 -> BatchNorm -> [ReLU]) on a channels-last map:
    the transposed 3-D kernel on a one-plane volume whose weight has only its middle depth slice -- output plane 0 is the 2-D result
    (plane 1 sees no tap and is discarded: twice the arithmetic, no new kernel for the reference's non-default pyramid)."""
    conv = m.conv
    if (not _same2(conv.kernel_size, 3) or not _same2(conv.stride, 2) or not _same2(conv.padding, 1) or not _same2(conv.output_padding, 1)
            or not _same2(conv.dilation, 1) or conv.groups != 1):
        raise RcmvsError(f"Deconv2d: only kernel 3, stride 2, padding 1, output_padding 1 (DeConv2dFuse, models/modules.py:346) runs on the "
                         f"HIP path; got {conv}")
    if m.bn is None:
        raise RcmvsError("Deconv2d: bn=False is not provided (the reference's forward returns its INPUT in that case, models/modules.py:106-110)")
    scale, shift = _bn_fold(m.bn)
    pw = _packed_of(m, lambda w: ops.pack_conv3d_weight(F.pad(w.unsqueeze(2), (0, 0, 0, 0, 1, 1)), transposed=True))
    return ops.deconv3d(t.unsqueeze(1), pw, scale, shift, relu=bool(m.relu))[:, 0]


class Conv2d(nn.Module):
    def __init__(self, in_channels, out_channels, kernel_size, stride=1, relu=True, bn=True, bn_momentum=0.1,
                 init_method="xavier", **kwargs):
        super().__init__()
        self.conv = nn.Conv2d(in_channels, out_channels, kernel_size, stride=stride, bias=(not bn), **kwargs)
        self.kernel_size = kernel_size
        self.stride = stride
        self.bn = nn.BatchNorm2d(out_channels, momentum=bn_momentum) if bn else None
        self.relu = relu

    def forward(self, x):
        """models/modules.py:56-62 as a stand-alone module: (N,Ci,H,W) -> (N,Co,Ho,Wo), eval mode, autograd off."""
        if _hip_inference(self, x):
            return ops.to_channels_first(_conv2d_block_cl(self, _nhwc(x)).contiguous())
        _unsupported(self, x)


class Deconv2d(nn.Module):
    """models/modules.py:71-116 (only DeConv2dFuse of the 'unet' pyramid uses it)."""

    def __init__(self, in_channels, out_channels, kernel_size, stride=1, relu=True, bn=True, bn_momentum=0.1,
                 init_method="xavier", **kwargs):
        super().__init__()
        assert stride in [1, 2]
        self.out_channels = out_channels
        self.stride = stride
        self.conv = nn.ConvTranspose2d(in_channels, out_channels, kernel_size, stride=stride, bias=(not bn), **kwargs)
        self.bn = nn.BatchNorm2d(out_channels, momentum=bn_momentum) if bn else None
        self.relu = relu

    def forward(self, x):
        if _hip_inference(self, x):
            return ops.to_channels_first(_deconv2d_block_cl(self, _nhwc(x)).contiguous())
        _unsupported(self, x)


class DeConv2dFuse(nn.Module):
    """models/modules.py:342-360: up-sample x by the transposed conv, concatenate with the skip map, 3x3 conv."""

    def __init__(self, in_channels, out_channels, kernel_size, relu=True, bn=True, bn_momentum=0.1):
        super().__init__()
        self.deconv = Deconv2d(in_channels, out_channels, kernel_size, stride=2, padding=1, output_padding=1, bn=True, relu=relu,
                               bn_momentum=bn_momentum)
        self.conv = Conv2d(2 * out_channels, out_channels, kernel_size, stride=1, padding=1, bn=bn, relu=relu, bn_momentum=bn_momentum)

    def forward_cl(self, x_pre, x):
        """channels-last twin of forward: x_pre (N,2h,2w,Co), x (N,h,w,Ci) -> (N,2h,2w,Co)."""
        up = _deconv2d_block_cl(self.deconv, x)
        if up.shape != x_pre.shape:
            raise RcmvsError(f"DeConv2dFuse: the skip map {tuple(x_pre.shape)} is not twice the size of the up-sampled one {tuple(up.shape)}")
        return _conv2d_block_cl(self.conv, torch.cat((up, x_pre), dim=-1))

    def forward(self, x_pre, x):
        if _hip_inference(self, x_pre, x):
            return ops.to_channels_first(self.forward_cl(_nhwc(x_pre), _nhwc(x)).contiguous())
        _unsupported(self, x_pre, x)


class FeatureNet(nn.Module):
    """models/modules.py:363-464.  arch_mode='fpn' is the reference's shipped configuration (CascadeMVSNet passes it,
    models/casmvsnet.py:354) and the tuned one; 'unet' (the class default of the reference: DeConv2dFuse merges, 1x1 output convs) runs in
    eval mode on the same kernel families, one block at a time."""

    def __init__(self, base_channels, num_stage=3, stride=4, arch_mode="fpn"):
        super().__init__()
        if arch_mode not in ("unet", "fpn"):
            raise ValueError("mode must be in 'unet' or 'fpn', but get:{}".format(arch_mode))
        if num_stage not in (1, 2, 3):
            raise ValueError("num_stage must be 1, 2 or 3")
        self.arch_mode = arch_mode
        self.stride = stride
        self.base_channels = base_channels
        self.num_stage = num_stage
        b = base_channels
        self.conv0 = nn.Sequential(Conv2d(3, b, 3, 1, padding=1), Conv2d(b, b, 3, 1, padding=1))
        self.conv1 = nn.Sequential(Conv2d(b, b * 2, 5, stride=2, padding=2), Conv2d(b * 2, b * 2, 3, 1, padding=1),
                                   Conv2d(b * 2, b * 2, 3, 1, padding=1))
        self.conv2 = nn.Sequential(Conv2d(b * 2, b * 4, 5, stride=2, padding=2), Conv2d(b * 4, b * 4, 3, 1, padding=1),
                                   Conv2d(b * 4, b * 4, 3, 1, padding=1))
        self.out1 = nn.Conv2d(b * 4, b * 4, 1, bias=False)
        self.out_channels = [4 * b]
        final_chs = b * 4
        if arch_mode == "unet":
            if num_stage >= 2:
                self.deconv1 = DeConv2dFuse(b * 4, b * 2, 3)
                self.out2 = nn.Conv2d(b * 2, b * 2, 1, bias=False)
                self.out_channels.append(2 * b)
            if num_stage == 3:
                self.deconv2 = DeConv2dFuse(b * 2, b, 3)
                self.out3 = nn.Conv2d(b, b, 1, bias=False)
                self.out_channels.append(b)
        elif num_stage == 3:
            self.inner1 = nn.Conv2d(b * 2, final_chs, 1, bias=True)
            self.inner2 = nn.Conv2d(b * 1, final_chs, 1, bias=True)
            self.out2 = nn.Conv2d(final_chs, b * 2, 3, padding=1, bias=False)
            self.out3 = nn.Conv2d(final_chs, b, 3, padding=1, bias=False)
            self.out_channels += [b * 2, b]
        elif num_stage == 2:
            self.inner1 = nn.Conv2d(b * 2, final_chs, 1, bias=True)
            self.out2 = nn.Conv2d(final_chs, b, 3, padding=1, bias=False)
            self.out_channels.append(b)

    # -- native inference path: channels-last HIP kernels (conv + folded BN + ReLU, FPN merge fused) --------
    def hip_plan(self):
        # (tensors read straight from the module dictionaries: nn.Module.__getattr__ made this walk 0.1 ms of host time per scene)
        sub = self._modules
        mods = [m for seq in ("conv0", "conv1", "conv2") for m in sub[seq]._modules.values()]
        tens = []
        for m in mods:
            # running statistics are updated by the kernels through raw pointers (no version bump): num_batches_tracked is
            # incremented in place by every train-mode forward and stands in for them
            cv, bn = m._modules["conv"], m._modules["bn"]
            bb = bn._buffers
            tens += [_param(cv, "weight"), _param(bn, "weight"), _param(bn, "bias"), bb["running_mean"], bb["running_var"], bb["num_batches_tracked"]]
        unet = self.arch_mode == "unet"              # (its merge blocks keep their own packed weights: _packed_of)
        heads = ["out1"] + (["out2", "out3"] if unet else ["inner1", "out2", "inner2", "out3"])
        extra = [t for n in heads if n in sub for t in (_param(sub[n], "weight"), getattr(sub[n], "bias", None)) if t is not None]
        key = tuple((t.data_ptr(), t._version) for t in tens + extra)
        if getattr(self, "_plan", None) is None or key != self._plan_key:
            plan = {}
            names = ["conv0.0", "conv0.1", "conv1.0", "conv1.1", "conv1.2", "conv2.0", "conv2.1", "conv2.2"]
            for n, m in zip(names, mods):
                pad_to = 4 if m.conv.in_channels == 3 else None
                w = m.conv.weight
                if tuple(w.shape) in ((8, 8, 3, 3), (16, 16, 3, 3), (32, 32, 3, 3)) and m.stride == 1:
                    # square 3x3 layers: as a one-plane 3-D conv the layer goes to the planar form of the split-bf16 matrix-core
                    # kernel (csrc/conv3d_x3.hip, kd = 1 taps only); the scalar-weight VALU kernel runs these at 15-25 TF
                    w3 = w.detach().new_zeros(w.shape[0], w.shape[1], 3, 3, 3)
                    w3[:, :, 1] = w.detach()
                    plan[n] = ("mfma3d", ops.pack_conv3d_weight(w3)) + _bn_fold(m.bn)
                    continue
                if tuple(w.shape) in ((32, 16, 5, 5), (16, 8, 5, 5)) and m.stride == 2:
                    # 5x5 stride 2 (12 % VALU busy on the scalar-weight kernel): space-to-depth turns it into a 4C -> Co 3x3
                    # stride-1 layer (tap k = 2t + parity; the k = 5 taps are zero) for the planar split-bf16 matrix-core kernel,
                    # which reads the space-to-depth view straight from the un-rearranged map (ops.conv2d_s2d)
                    w3 = self._w3(self._w5s2(w.detach())).contiguous()
                    plan[n] = ("s2d_mfma3d", ops.pack_conv3d_weight(w3)) + _bn_fold(m.bn)
                    continue
                plan[n] = (ops.pack_conv2d_weight(w, pad_in_to=pad_to),) + _bn_fold(m.bn) + (m.stride,)
            # conv1.1 -> conv1.2 (16 -> 16 -> 16, stride 1) as ONE launch with the map between them in LDS (csrc/conv2d_pair.hip)
            m11, m12 = mods[3], mods[4]
            if CONV_PAIR and all(tuple(m.conv.weight.shape) == (16, 16, 3, 3) and m.stride == 1 for m in (m11, m12)):
                plan["pair1"] = (ops.pack_conv2d_pair(m11.conv.weight, m12.conv.weight),) + _bn_fold(m11.bn) + _bn_fold(m12.bn)
            # conv0.0 -> conv0.1 (3 -> 8 -> 8, stride 1) as ONE launch from the planar images (csrc/conv2d_stem.hip)
            m00, m01 = mods[0], mods[1]
            if CONV_STEM and tuple(m00.conv.weight.shape) == (8, 3, 3, 3) and tuple(m01.conv.weight.shape) == (8, 8, 3, 3) and m00.stride == 1 and m01.stride == 1:
                plan["stem"] = (plan["conv0.0"][0],) + _bn_fold(m00.bn) + (ops.pack_conv2d_stem(m01.conv.weight),) + _bn_fold(m01.bn)
            m10 = mods[2]
            if CONV_TILE and tuple(m10.conv.weight.shape) == (16, 8, 5, 5) and m10.stride == 2:
                plan["tile10"] = (ops.pack_conv2d_tile(self._w5s2(m10.conv.weight.detach()).contiguous()),) + _bn_fold(m10.bn)
            for n, m in (("conv2.1", mods[6]), ("conv2.2", mods[7])):           # 32 -> 32 at quarter resolution: the eight-wave form of the tile kernel
                if CONV_TILE and tuple(m.conv.weight.shape) == (32, 32, 3, 3) and m.stride == 1:
                    plan["tile:" + n] = (ops.pack_conv2d_tile(m.conv.weight),) + _bn_fold(m.bn)
            plan["out1"] = ops.pack_conv2d_weight(self.out1.weight)
            if unet:
                for n in ("out2", "out3")[:self.num_stage - 1]:           # 1x1, Co = Ci: the middle tap of a one-plane 3-D kernel
                    plan[n] = ops.pack_conv3d_weight(self._w3(getattr(self, n).weight.detach()))
            elif self.num_stage >= 2:
                plan["inner1"] = (ops.pack_conv2d_weight(self.inner1.weight), self.inner1.bias.detach().float().contiguous())
                if tuple(self.out2.weight.shape) == (16, 32, 3, 3):       # 32 -> 16 3x3: planar split-bf16 kernel, as the trunk's square layers
                    w3 = self.out2.weight.detach().new_zeros(16, 32, 3, 3, 3)
                    w3[:, :, 1] = self.out2.weight.detach()
                    plan["out2"] = ("mfma3d", ops.pack_conv3d_weight(w3))
                    if CONV_TILE:
                        plan["out2t"] = ops.pack_conv2d_tile(self.out2.weight)
                else:
                    plan["out2"] = ops.pack_conv2d_weight(self.out2.weight)
            if self.num_stage == 3 and not unet:
                plan["inner2"] = (ops.pack_conv2d_weight(self.inner2.weight), self.inner2.bias.detach().float().contiguous())
                plan["out3"] = ops.pack_conv2d_weight(self.out3.weight)
                plan["fuse_out3"] = tuple(self.inner2.weight.shape[:2]) == (32, 8) and tuple(self.out3.weight.shape[:2]) == (8, 32)
                # the same level with the 1x1 lateral conv folded into the 3x3 output conv (1600 instead of 2628 multiply-adds per pixel;
                # equal up to fp32 rounding; ops.fpn_out_fused is the bit-identical-to-unfused kernel, kept under test)
                if plan["fuse_out3"]:
                    plan["fold_out3"] = ops.pack_fpn_folded(self.inner2.weight, self.inner2.bias, self.out3.weight)
                    if os.environ.get("RCMVS_FPN_MFMA", "1") != "0":           # ... on the matrix cores, exact split operands (csrc/fpn_folded_mfma.hip)
                        plan["fold_out3"] = ops.pack_fpn_folded_mfma(plan["fold_out3"])
            self._plan, self._plan_key = plan, key
        return self._plan

    def forward_cl(self, x, lazy=False):
        """x (N,3,H,W) NCHW -> {'stageK': (N,h,w,C)} channels-last feature maps (HIP path, eval-mode BN).
        lazy=True returns {'stageK': thunk}: the trunk and the FPN merges run now, each stage's output conv
        runs when its thunk is called -- the cascade calls it right before that stage's warp so the maps are
        still in L2 / Infinity Cache when K1 gathers from them (they are produced ~1 ms and ~400 MB of volume
        traffic earlier otherwise)."""
        p = self.hip_plan()

        def cbr(t, n):
            if "tile:" + n in p and not ops._CONV_IMPL:
                img, sc, sh = p["tile:" + n]
                return ops.conv2d_tile(t, img, sc, sh, relu=True)
            if p[n][0] == "mfma3d":
                _, w3, sc, sh = p[n]
                return ops.conv3d(t.unsqueeze(1), w3, sc, sh, relu=True).squeeze(1)
            if p[n][0] == "s2d_mfma3d":
                if t.shape[1] % 2 or t.shape[2] % 2:
                    raise RcmvsError(f"FeatureNet: image height and width must be multiples of 4 (got a {t.shape[1]}x{t.shape[2]} map "
                                     "at the second stride-2 layer), as the reference's three-level pyramid requires")
                _, w3, sc, sh = p[n]
                if ops._CONV_IMPL:          # tests / A-B with the split-bf16 kernels switched off: the materialised view on the selected kernels
                    return ops.conv3d(self._s2d(t).contiguous().unsqueeze(1), w3, sc, sh, relu=True).squeeze(1)
                return ops.conv2d_s2d(t, w3, sc, sh, relu=True)
            w, sc, sh, stride = p[n]
            return ops.conv2d(t, w, sc, sh, stride=stride, relu=True)

        w00 = p["conv0.0"][0] if len(p["conv0.0"]) == 4 else None
        if "stem" in p and not ops._CONV_IMPL:
            c0 = ops.conv2d_stem(x.contiguous().float(), *p["stem"])
        else:
            c0 = None
        # the planar first layer is built for the reference's 3 -> 8, k = 3, stride 1 (base_channels = 8); other widths take the NHWC4 path
        if c0 is not None:
            pass
        elif w00 is not None and w00.ci == 4 and w00.co == 8 and w00.k == 3 and p["conv0.0"][3] == 1 and not ops._CONV_IMPL:
            w, sc, sh, _ = p["conv0.0"]                              # the first layer reads the planar images itself (no NHWC4 pass)
            c0 = cbr(ops.conv2d_rgb(x.contiguous().float(), w, sc, sh, relu=True), "conv0.1")
        else:
            c0 = cbr(cbr(ops.rgb_to_nhwc4(x.contiguous().float()), "conv0.0"), "conv0.1")
        if "tile10" in p and not ops._CONV_IMPL and c0.shape[1] % 2 == 0 and c0.shape[2] % 2 == 0:
            c10 = ops.conv2d_tile(c0, p["tile10"][0], p["tile10"][1], p["tile10"][2], relu=True, s2d=True)
        else:
            c10 = cbr(c0, "conv1.0")
        if "pair1" in p and not ops._CONV_IMPL:
            c1 = ops.conv2d_pair(c10, *p["pair1"])
        else:
            c1 = cbr(cbr(c10, "conv1.1"), "conv1.2")
        c2 = cbr(cbr(cbr(c1, "conv2.0"), "conv2.1"), "conv2.2")
        # A thunk takes an optional activation-bound vector (ops.ABSMAX_FLOATS floats, zero-filled): the output conv then leaves (max|f|)^2
        # there -- the bound of the variance volume built from the map, which the fp16-pair cost regularisation needs -- in its epilogue
        # and returns (map, True); (map, False) = no bound was kept (the caller runs ops.absmax over the map).
        def out1(bound=None, t=c2):
            if tuple(self.out1.weight.shape) == (32, 32, 1, 1) and (bound is not None or ONE_BY_ONE_MFMA):      # the 1x1 kernels keep the bound in their epilogue
                return ops.conv1x1(t, p["out1"], ysq_absmax=bound, mfma=ONE_BY_ONE_MFMA), bound is not None
            return ops.conv2d(t, p["out1"]), False
        out = {"stage1": out1}
        plain = lambda f: (lambda bound=None: (f(), False))
        if self.arch_mode == "unet":                     # models/modules.py:449-457
            one = lambda t, n: ops.conv3d(t.unsqueeze(1), p[n]).squeeze(1)
            if self.num_stage >= 2:
                intra = self.deconv1.forward_cl(c1, c2)
                out["stage2"] = plain(lambda t=intra: one(t, "out2"))
            if self.num_stage == 3:
                intra = self.deconv2.forward_cl(c0, intra)
                out["stage3"] = plain(lambda t=intra: one(t, "out3"))
            return out if lazy else {k: f()[0] for k, f in out.items()}
        if self.num_stage >= 2:
            # (the 16 -> 32 lateral merge stays on the fp32 streaming kernel: its matrix-core form -- K = 16, v_mfma_f32_16x16x16_bf16 --
            # measured 15.7 against 14.9 us: the layer waits for its 55 MB, not for arithmetic; the 32 -> 32 output conv above: 14.1 -> 7.1 us)
            intra = ops.conv2d(c1, p["inner1"][0], None, p["inner1"][1], up_add=c2)
            if isinstance(p["out2"], tuple):
                def out2(bound=None, t=intra):
                    keep = bound is not None and not ops._CONV_IMPL
                    if "out2t" in p and not ops._CONV_IMPL:
                        return ops.conv2d_tile(t, p["out2t"], ysq_absmax=bound), keep
                    return ops.conv3d(t.unsqueeze(1), p["out2"][1], y_absmax=bound if keep else None, y_absmax_square=keep).squeeze(1), keep
                out["stage2"] = out2
            else:
                out["stage2"] = plain(lambda t=intra: ops.conv2d(t, p["out2"]))
        if self.num_stage == 3:
            if p.get("fuse_out3") and c0.shape[1] % 2 == 0 and c0.shape[2] % 2 == 0:
                # the full-resolution 32-channel merge is never stored: 1x1 lateral + up-add + 3x3 output conv in one launch
                if p.get("fold_out3") is not None:
                    out["stage3"] = (lambda bound=None, a=c0, b=intra: (ops.fpn_out_folded(a, b, p["fold_out3"], ysq_absmax=bound), bound is not None))
                else:
                    out["stage3"] = plain(lambda a=c0, b=intra: ops.fpn_out_fused(a, b, p["inner2"][0], p["inner2"][1], p["out3"]))
            else:
                intra = ops.conv2d(c0, p["inner2"][0], None, p["inner2"][1], up_add=intra)
                out["stage3"] = plain(lambda t=intra: ops.conv2d(t, p["out3"]))
        return out if lazy else {k: f()[0] for k, f in out.items()}

    # ---- training on the library: every layer as a one-plane volume on the 3-D conv family -----------------------
    @staticmethod
    def _w3(w):
        """(Co,Ci,k,k) with k = 1 or 3 -> (Co,Ci,3,3,3) whose only non-zero depth slice is the middle one (differentiable)."""
        p = (3 - w.shape[-1]) // 2
        return F.pad(w.unsqueeze(2), (p, p, p, p, 1, 1))

    @staticmethod
    def _w5s2(w):
        """5x5 stride-2 weight (Co,Ci,5,5) -> the equivalent 3x3 stride-1 weight (Co,4Ci,3,3) on the space-to-depth input:
        tap k = 2t + a per axis (t = 3x3 tap, a = pixel parity), the k = 5 combinations are zero (differentiable)."""
        Co, Ci = w.shape[:2]
        w6 = F.pad(w, (0, 1, 0, 1))                                        # (Co,Ci,6,6)
        return w6.reshape(Co, Ci, 3, 2, 3, 2).permute(0, 3, 5, 1, 2, 4).reshape(Co, 4 * Ci, 3, 3)

    @staticmethod
    def _s2d(x):
        """(N,H,W,C) -> (N,H/2,W/2,4C), channel order (row parity, column parity, c)."""
        N, H, W, C = x.shape
        return x.reshape(N, H // 2, 2, W // 2, 2, C).permute(0, 1, 3, 2, 4, 5).reshape(N, H // 2, W // 2, 4 * C)

    def forward_train_cl(self, x, segments=1):
        """Train-mode twin of ``forward_cl``: x (N,3,H,W) holds `segments` consecutive groups of images (the V views) that
        are normalised independently, exactly as if the module were called once per group (models/casmvsnet.py:364-366),
        while every convolution and gradient runs once over all N images.  Batch-statistics BatchNorm, autograd through
        train_ops (conv / norm / gradients on the HIP kernels; the 5x5 stride-2 layers as space-to-depth + 3x3, the 2x
        nearest up-sampling and the adds of the FPN merge as PyTorch element-wise ops).  Returns channels-last maps."""
        from .train_ops import ConvPlainFn, conv_bn_train_w
        if self.arch_mode != "fpn":
            raise RcmvsError("FeatureNet(arch_mode='unet'): the train-mode HIP path covers the shipped 'fpn' pyramid only (the one-plane form of "
                             "its transposed convs would put the discarded plane into the batch statistics)")
        N, _, H, W = x.shape
        if H % 4 or W % 4:
            raise RcmvsError("FeatureNet (training): image height and width must be multiples of 4")
        t = F.pad(x.permute(0, 2, 3, 1), (0, 5)).contiguous()              # (N,H,W,8): RGB + zero channels

        def cbr(t, m):
            w = m.conv.weight
            if m.stride == 2:
                t, w = self._s2d(t), self._w5s2(w)
            return conv_bn_train_w(self._w3(w), m.bn, t.unsqueeze(1), relu=True, segments=segments).squeeze(1)

        def plain(t, conv):
            return ConvPlainFn.apply(t.unsqueeze(1), self._w3(conv.weight), conv.bias).squeeze(1)

        def up2(t):
            return t.repeat_interleave(2, dim=1).repeat_interleave(2, dim=2)

        c0 = t
        for m in self.conv0: c0 = cbr(c0, m)
        c1 = c0
        for m in self.conv1: c1 = cbr(c1, m)
        c2 = c1
        for m in self.conv2: c2 = cbr(c2, m)
        out = {"stage1": plain(c2, self.out1)}
        intra = c2
        if self.num_stage >= 2:
            intra = up2(intra) + plain(c1, self.inner1)
            out["stage2"] = plain(intra, self.out2)
        if self.num_stage == 3:
            intra = up2(intra) + plain(c0, self.inner2)
            out["stage3"] = plain(intra, self.out3)
        return out

    def forward(self, x):
        """x (N,3,H,W) -> {'stageK': (N,C,h,w)} like the reference module (models/modules.py:440-464), on the HIP kernels."""
        if _hip_inference(self, x):
            return {k: ops.to_channels_first(v) for k, v in self.forward_cl(x).items()}
        if _hip_training(self, x):
            return {k: v.permute(0, 3, 1, 2) for k, v in self.forward_train_cl(x).items()}
        _unsupported(self, x)


# ----------------------------------------------------------------------------------------------
# 3-D blocks (models/modules.py:118-210): inside CostRegNet the network's HIP plan reads their conv / bn children; called on their own
# they run the same kernels one block at a time (_block3d)
# ----------------------------------------------------------------------------------------------
class Conv3d(nn.Module):
    def __init__(self, in_channels, out_channels, kernel_size=3, stride=1, relu=True, bn=True, bn_momentum=0.1, **kwargs):
        super().__init__()
        assert stride in [1, 2]
        self.out_channels = out_channels
        self.kernel_size = kernel_size
        self.stride = stride
        self.conv = nn.Conv3d(in_channels, out_channels, kernel_size, stride=stride, bias=(not bn), **kwargs)
        self.bn = nn.BatchNorm3d(out_channels, momentum=bn_momentum)
        self.gn = None
        self.relu = relu

    def forward(self, x):
        """models/modules.py:149-157 as a stand-alone module: (B,Ci,D,H,W) -> (B,Co,Do,Ho,Wo) on the 3-D conv family (eval mode with
        autograd off: folded BatchNorm; train mode: batch statistics + autograd, train_ops.ConvBnReluFn)."""
        return _block3d(self, x)


class Deconv3d(nn.Module):
    def __init__(self, in_channels, out_channels, kernel_size=3, stride=1, relu=True, bn=True, bn_momentum=0.1, **kwargs):
        super().__init__()
        assert stride in [1, 2]
        self.out_channels = out_channels
        self.stride = stride
        self.conv = nn.ConvTranspose3d(in_channels, out_channels, kernel_size, stride=stride, bias=(not bn), **kwargs)
        self.bn = nn.BatchNorm3d(out_channels, momentum=bn_momentum)
        self.gn = None
        self.relu = relu

    def forward(self, x):
        """models/modules.py:196-204 as a stand-alone module: (B,Ci,D,H,W) -> (B,Co,2D,2H,2W)."""
        return _block3d(self, x)


def _block3d(m, x):
    """A Conv3d / Deconv3d block called on its own (inside CostRegNet the plan of the network runs it)."""
    conv = m.conv
    transposed = isinstance(conv, nn.ConvTranspose3d)
    ok = (_same3(conv.kernel_size, 3) and _same3(conv.padding, 1) and _same3(conv.dilation, 1) and conv.groups == 1
          and (_same3(conv.stride, 2) and _same3(conv.output_padding, 1) if transposed
               else conv.stride[0] in (1, 2) and _same3(conv.stride, conv.stride[0]) and conv.padding_mode == "zeros"))
    if not ok:
        raise RcmvsError(f"{type(m).__name__}: only kernel 3, padding 1 (transposed: stride 2, output_padding 1) -- the layers of "
                         f"CostRegNet, models/modules.py:473-487 -- run on the HIP path; got {conv}")
    if _hip_inference(m, x):
        scale, shift = _affine_of(m)
        pw = _packed_of(m, lambda w: ops.pack_conv3d_weight(w, transposed=transposed))
        t = ops.to_channels_last(x.contiguous().float())
        y = ops.deconv3d(t, pw, scale, shift, relu=bool(m.relu)) if transposed else ops.conv3d(t, pw, scale, shift, stride=conv.stride[0], relu=bool(m.relu))
        return ops.to_channels_first(y)
    if _hip_training(m, x):
        if m.bn is None or conv.bias is not None:
            raise RcmvsError(f"{type(m).__name__}: the train-mode HIP path is conv -> BatchNorm (batch statistics) -> [ReLU] without a bias")
        from .train_ops import conv_bn_relu_train
        return conv_bn_relu_train(m, x.float().permute(0, 2, 3, 4, 1).contiguous()).permute(0, 4, 1, 2, 3)
    _unsupported(m, x)


def _same3(v, n):
    return tuple(v) == (n, n, n)


def _bn_fold(bn):
    """eval-mode BatchNorm as (scale, shift), computed on the device without a sync."""
    scale = bn.weight.detach() / torch.sqrt(bn.running_var + bn.eps)
    shift = bn.bias.detach() - bn.running_mean * scale
    return scale.float().contiguous(), shift.float().contiguous()


class CostRegNet(nn.Module):
    """models/modules.py:470-501.  ``forward`` keeps the reference contract ((B,C,D,h,w) ->
    (B,1,D,h,w) logits); the cascade uses ``features_cl`` + the fused depth head instead."""

    _LAYERS = ("conv0", "conv1", "conv2", "conv3", "conv4", "conv5", "conv6", "conv7", "conv9", "conv11")
    BOUND_ROWS = 11         # rows of the activation-bound buffer of the fp16-pair form: the input volume + ten layer outputs (the last one: the depth head's input)

    def __init__(self, in_channels, base_channels):
        super().__init__()
        b = base_channels
        self.conv0 = Conv3d(in_channels, b, padding=1)
        self.conv1 = Conv3d(b, b * 2, stride=2, padding=1)
        self.conv2 = Conv3d(b * 2, b * 2, padding=1)
        self.conv3 = Conv3d(b * 2, b * 4, stride=2, padding=1)
        self.conv4 = Conv3d(b * 4, b * 4, padding=1)
        self.conv5 = Conv3d(b * 4, b * 8, stride=2, padding=1)
        self.conv6 = Conv3d(b * 8, b * 8, padding=1)
        self.conv7 = Deconv3d(b * 8, b * 4, stride=2, padding=1, output_padding=1)
        self.conv9 = Deconv3d(b * 4, b * 2, stride=2, padding=1, output_padding=1)
        self.conv11 = Deconv3d(b * 2, b * 1, stride=2, padding=1, output_padding=1)
        self.prob = nn.Conv3d(b, 1, 3, stride=1, padding=1, bias=False)
        self._plan = None
        self._plan_key = None

    # -- HIP execution plan: packed weights + folded BN, rebuilt only when a tensor changes -----
    def _tensors(self):
        # read straight from the module dictionaries: nn.Module.__getattr__ costs ~0.4 us per attribute and this list is walked for every
        # plan validation (61 tensors, six lookups deep, three networks per scene -- it was 0.4 ms of host time per scene in round 3)
        ts = []
        mods = self._modules
        for n in self._LAYERS:
            m = mods[n]._modules
            cv, bn = m["conv"], m["bn"]
            bb = bn._buffers
            ts += [_param(cv, "weight"), _param(bn, "weight"), _param(bn, "bias"), bb["running_mean"], bb["running_var"], bb["num_batches_tracked"]]
        ts.append(_param(mods["prob"], "weight"))
        return ts

    def hip_plan(self):
        key = tuple((t.data_ptr(), t._version) for t in self._tensors())
        if self._plan is None or key != self._plan_key:
            plan = {}
            for n in self._LAYERS:
                m = getattr(self, n)
                w = ops.pack_conv3d_weight(m.conv.weight, transposed=isinstance(m, Deconv3d))
                s, b = _bn_fold(m.bn)
                plan[n] = (w, s, b)
            plan["prob"] = ops.pack_conv3d_weight(self.prob.weight)
            plan["conv11_coef"] = self._conv11_bound_coef(plan["conv11"][1], plan["conv11"][2])
            self._plan, self._plan_key = plan, key
        return self._plan

    def _conv11_bound_coef(self, scale, shift):
        """{c1, c2} of ops.conv11_prob: max|relu(bn(conv11(t)))| <= c1 max|t| + c2.  An output voxel of parity class (pd, ph, pw) sums the taps
        k = 1 (parity 0) or k in {0, 2} (parity 1) per axis over the 16 input channels: c1 = max_co |scale_co| x (the largest of the eight class-wise
        L1 norms of the weights), c2 = max_co |shift_co|; computed on the device, no sync, x (1 + 2^-10) against the rounding of the sums."""
        w = self.conv11.conv.weight.detach().float().abs().sum(0)          # ConvTranspose3d weight (Ci, Co, 3, 3, 3) -> (Co, 3, 3, 3)
        taps = ([1], [0, 2])
        l1 = None
        for pd in range(2):
            for ph in range(2):
                for pw in range(2):
                    v = w[:, taps[pd]][:, :, taps[ph]][:, :, :, taps[pw]].sum((1, 2, 3))
                    l1 = v if l1 is None else torch.maximum(l1, v)
        c1 = (scale.abs() * l1).max() * (1.0 + 2.0 ** -10)
        c2 = shift.abs().max() * (1.0 + 2.0 ** -10)
        return torch.stack((c1, c2)).float().contiguous()

    def features_cl(self, x, x_absmax=None, plan=None, logits=False, planes=None):
        """x (B,D,h,w,C) channels-last -> the 8-channel volume fed to ``prob`` (B,D,h,w,8); logits=True (fp16-pair form only): -> the prob conv's
        output (B,D,h,w) instead, the last transposed layer and the prob conv fused; with `planes` (B,h,w,2) as well: (depth, confidence), the head included
        (one launch for D = 8, the logits + ops.softmax_head otherwise).
        x_absmax: None = the exact three-piece bf16 form of the matrix-core kernels; or a (BOUND_ROWS, ops.ABSMAX_FLOATS) tensor whose row 0
        is a bound of max|x| (ops.absmax format; the cascade derives it from the feature maps) and whose other rows are ZERO: the
        layers then run on the fp16-pair form (half the matrix-pipe work) and every layer leaves the bound of its output in the
        next row for its consumer (one atomic max per block, inside the kernel)."""
        B, D, h, w, _ = x.shape
        if D % 8 or h % 8 or w % 8:
            raise RcmvsError(f"CostRegNet: volume {D}x{h}x{w} must be divisible by 8 in every axis "
                             "(three stride-2 levels with skip connections, models/modules.py:492-499)")
        p = self.hip_plan() if plan is None else plan          # (the cascade validates a stage's plan once and hands it in)
        if logits and (x_absmax is None or B != 1):
            raise RcmvsError("CostRegNet.features_cl(logits=True): the fused conv11 + prob pass is the fp16-pair form of a B = 1 scene")
        if x_absmax is None:
            conv0 = ops.conv3d(x, *p["conv0"], relu=True)
            conv2 = ops.conv3d(ops.conv3d(conv0, *p["conv1"], stride=2, relu=True), *p["conv2"], relu=True)
            conv4 = ops.conv3d(ops.conv3d(conv2, *p["conv3"], stride=2, relu=True), *p["conv4"], relu=True)
            t = ops.conv3d(ops.conv3d(conv4, *p["conv5"], stride=2, relu=True), *p["conv6"], relu=True)
            t = ops.deconv3d(t, *p["conv7"], residual=conv4, relu=True)
            t = ops.deconv3d(t, *p["conv9"], residual=conv2, relu=True)
            return ops.deconv3d(t, *p["conv11"], residual=conv0, relu=True)
        b = [x_absmax[i] for i in range(1, self.BOUND_ROWS)]      # bounds of conv0, 1, 2, 3, 9, 7, 4, 5, 6, 11 (zero on entry: the caller's one fill per scene)
        x_absmax = x_absmax[0]
        conv0 = ops.conv3d(x, *p["conv0"], relu=True, x_absmax=x_absmax, y_absmax=b[0])
        conv1 = ops.conv3d(conv0, *p["conv1"], stride=2, relu=True, x_absmax=b[0], y_absmax=b[1])
        conv2 = ops.conv3d(conv1, *p["conv2"], relu=True, x_absmax=b[1], y_absmax=b[2])
        conv3 = ops.conv3d(conv2, *p["conv3"], stride=2, relu=True, x_absmax=b[2], y_absmax=b[3])
        if DEEP_PAIR:               # the deep levels on their own fp16-pair kernels (csrc/conv3d_deep.hip)
            conv4 = ops.conv3d(conv3, *p["conv4"], relu=True, x_absmax=b[3], y_absmax=b[6])
            t = ops.conv3d(conv4, *p["conv5"], stride=2, relu=True, x_absmax=b[6], y_absmax=b[7])
            t = ops.conv3d(t, *p["conv6"], relu=True, x_absmax=b[7], y_absmax=b[8])
            t = ops.deconv3d(t, *p["conv7"], residual=conv4, relu=True, x_absmax=b[8], y_absmax=b[5])
        else:                       # RCMVS_DEEP_PAIR=0: fp32-MFMA deep levels, no bounds needed (the kernel keeps the bound of conv7 too)
            conv4 = ops.conv3d(conv3, *p["conv4"], relu=True, x_absmax=b[3])
            t = ops.conv3d(ops.conv3d(conv4, *p["conv5"], stride=2, relu=True), *p["conv6"], relu=True)
            t = ops.deconv3d(t, *p["conv7"], residual=conv4, relu=True, y_absmax=b[5])
        t = ops.deconv3d(t, *p["conv9"], residual=conv2, relu=True, x_absmax=b[5], y_absmax=b[4])
        if logits:                  # conv11 + prob in one pass: the 8-channel volume never reaches memory (csrc/conv11_prob.hip); with `planes`: the head too
            return ops.conv11_prob(t, b[4], p["conv11"][0], p["conv11"][1], p["conv11"][2], conv0, b[0], p["conv11_coef"], p["prob"], planes=planes)
        return ops.deconv3d(t, *p["conv11"], residual=conv0, relu=True, x_absmax=b[4], y_absmax=b[9])      # (the depth head's bound)

    def features_cl_train(self, x):
        """Train-mode twin of ``features_cl`` (batch-statistics BatchNorm, autograd through the HIP kernels:
        train_ops.ConvBnReluFn); updates the running statistics like nn.BatchNorm3d does."""
        from .train_ops import conv_bn_relu_train as blk
        B, D, h, w, _ = x.shape
        if D % 8 or h % 8 or w % 8:
            raise RcmvsError(f"CostRegNet: volume {D}x{h}x{w} must be divisible by 8 in every axis")
        conv0 = blk(self.conv0, x)
        conv2 = blk(self.conv2, blk(self.conv1, conv0))
        conv4 = blk(self.conv4, blk(self.conv3, conv2))
        t = blk(self.conv6, blk(self.conv5, conv4))
        t = blk(self.conv7, t, residual=conv4)
        t = blk(self.conv9, t, residual=conv2)
        return blk(self.conv11, t, residual=conv0)

    def forward(self, x):
        if _hip_inference(self, x):
            feat = self.features_cl(ops.to_channels_last(x.contiguous().float()))
            logits = ops.conv3d(feat, self.hip_plan()["prob"])
            return ops.to_channels_first(logits)
        _unsupported(self, x)


# ----------------------------------------------------------------------------------------------
def _hip_inference(module, *tensors):
    """The inference path: eval mode, autograd off, tensors on the GPU."""
    return (not module.training) and (not torch.is_grad_enabled()) and all(t.is_cuda for t in tensors)


def _hip_training(module, *tensors):
    """The training path (autograd Functions over the HIP kernels): train mode, tensors on the GPU."""
    return module.training and all(t.is_cuda for t in tensors)


def _unsupported(module, *tensors):
    name = type(module).__name__
    if not all(t.is_cuda for t in tensors):
        raise RcmvsError(f"{name}: the HIP path needs its tensors on the GPU (there is no CPU / eager fallback)")
    raise RcmvsError(f"{name}: eval-mode forward with autograd enabled is not provided -- wrap inference in torch.no_grad(), "
                     "or put the module in train() mode for the differentiable HIP path")


def _holder_only(module):
    raise RcmvsError(f"{type(module).__name__} is a parameter holder: its conv / bn children are executed by the enclosing "
                     "network's HIP plan (Rendering_Consistency_Net.forward and its volume / MLP chains)")


class DepthNet(nn.Module):
    """models/casmvsnet.py:45-124 (train variant: also returns volume_feature_no_ref) and 234-311 (DepthNet_eval): one stage's cost volume
    + regularisation + depth head.  Parameter-free.  Inside the cascade the stage loop of _CascadeBase runs these ops itself (it keeps
    the channels-last maps, the batched homographies and the activation bounds across stages); called on its own it is the same
    kernels for one stage."""

    def __init__(self, train_variant):
        super().__init__()
        self.train_variant = train_variant

    def forward(self, features, proj_matrices, depth_values, num_depth, cost_regularization, imgs=None, pad=0, prob_volume_init=None):
        """features: V maps (B,C,h,w), reference view first; proj_matrices (B,V,2,4,4); depth_values (B,D,h,w), uniformly spaced along D
        at every pixel (what get_depth_range_samples produces, models/modules.py:560-620: the kernels take them as {d_0, delta} planes);
        cost_regularization: a CostRegNet; imgs (B,V,3,H,W), needed by the train variant only."""
        if pad != 0 or prob_volume_init is not None:
            raise RcmvsError("DepthNet: pad and prob_volume_init are not provided (no caller of the reference passes them)")
        V = len(features)
        B, C, h, w = features[0].shape
        D = int(num_depth)
        if proj_matrices.shape[1] != V:
            raise RcmvsError("DepthNet: different number of images and projection matrices")
        if tuple(depth_values.shape) != (B, D, h, w):
            raise RcmvsError(f"DepthNet: depth_values {tuple(depth_values.shape)} is not (B, num_depth, h, w) = {(B, D, h, w)}")
        dv = depth_values.float()
        d0 = dv[:, 0]
        delta = (dv[:, 1] - d0) if D > 1 else torch.zeros_like(d0)
        k = torch.arange(D, device=dv.device, dtype=torch.float32).view(1, D, 1, 1)
        if float((dv - (d0.unsqueeze(1) + k * delta.unsqueeze(1))).abs().max()) > 1e-5 * float(dv.abs().max()):
            raise RcmvsError("DepthNet: depth_values must be uniformly spaced along the depth axis at every pixel (d_0 + k * delta)")
        planes = torch.stack((d0, delta), dim=-1).contiguous()
        with torch.no_grad():
            rot, trans = ops.compose_homography(proj_matrices.contiguous().float())
            small_cl = None
            if self.train_variant:
                if imgs is None:
                    raise RcmvsError("DepthNet (train variant): imgs (B,V,3,H,W) are needed for volume_feature_no_ref")
                small_cl = ops.resize_rgb_cl(imgs.float().reshape(B * V, *imgs.shape[2:]).contiguous(), (h, w)).view(B, V, h, w, 3)
        if _hip_inference(self, *features, depth_values):
            f_cl = torch.stack([ops.to_channels_last(f.contiguous().float()) for f in features], dim=1)
            var = ops.warp_variance(f_cl, rot, trans, planes, D)
            x8 = cost_regularization.features_cl(var)
            depth, conf = ops.depth_head(x8, cost_regularization.hip_plan()["prob"], planes)
            out = {"depth": depth, "photometric_confidence": conf}
            if self.train_variant:          # eval mode: the reference's in-place pow_ quirk applies (models/casmvsnet.py:95-97)
                out["volume_feature_no_ref"] = ops.warp_noref(f_cl, small_cl, rot, trans, planes, D, square_first=True)
            return out
        if _hip_training(self, *features, depth_values):
            from . import train_ops
            f_cl = torch.stack([f.float().permute(0, 2, 3, 1) for f in features], dim=1).contiguous()
            res = ops.WarpVarianceFn.apply(f_cl, rot, trans, planes, D, small_cl)
            var, noref = res if self.train_variant else (res, None)
            x8 = cost_regularization.features_cl_train(var)
            depth, conf = train_ops.ProbDepthHeadFn.apply(x8, cost_regularization.prob.weight, planes)
            out = {"depth": depth, "photometric_confidence": conf}
            if self.train_variant:
                out["volume_feature_no_ref"] = noref
            return out
        _unsupported(self, *features, depth_values)


class _CascadeBase(nn.Module):
    TRAIN_VARIANT = False

    def __init__(self, refine=False, ndepths=[48, 32, 8], depth_interals_ratio=[4, 2, 1], share_cr=False,
                 grad_method="detach", arch_mode="fpn", cr_base_chs=[8, 8, 8]):
        super().__init__()
        if refine:
            raise NotImplementedError("refine=True: the reference's RefineNet is not runnable (models/modules.py:511 calls F.cat)")
        self.refine = refine
        self.share_cr = share_cr
        self.ndepths = ndepths
        self.depth_interals_ratio = depth_interals_ratio
        self.grad_method = grad_method
        self.arch_mode = arch_mode
        self.cr_base_chs = cr_base_chs
        self.num_stage = len(ndepths)
        assert len(ndepths) == len(depth_interals_ratio)
        self.stage_infos = {"stage1": {"scale": 4.0}, "stage2": {"scale": 2.0}, "stage3": {"scale": 1.0}}
        self.feature = FeatureNet(base_channels=8, stride=4, num_stage=self.num_stage, arch_mode=self.arch_mode)
        if self.share_cr:
            self.cost_regularization = CostRegNet(in_channels=self.feature.out_channels, base_channels=8)
        else:
            self.cost_regularization = nn.ModuleList([CostRegNet(in_channels=self.feature.out_channels[i],
                                                                 base_channels=self.cr_base_chs[i])
                                                      for i in range(self.num_stage)])
        self.DepthNet = DepthNet(self.TRAIN_VARIANT)

    def _cr(self, s):
        return self.cost_regularization if self.share_cr else self.cost_regularization[s]

    # ---------------------------------------------------------------- native inference path
    def _forward_hip(self, imgs, proj_matrices, depth_values, features=None, homographies=None):
        """Test hooks: `features` {'stageK': (B*V, C, h, w)} bypasses the feature pyramid;
        `homographies` {'stageK': (rot (B,V-1,9), trans (B,V-1,3))} bypasses the fp64 composer (to
        feed the hot path the reference's own fp32 values)."""
        B, V, _, H, W = imgs.shape
        imgs = imgs.float()
        depth_values = depth_values.contiguous().float()
        # eval-mode BN: batching the V views is exact.  `features` (test hook) are NCHW maps; the native pyramid
        # already produces the channels-last maps K1 reads.
        feats_cl = None
        if features is None:
            feats_cl = self.feature.forward_cl(imgs.reshape(B * V, 3, H, W), lazy=True)
        outputs = {}
        depth = None
        bounds = None
        need_fill = False
        # arithmetic of the cost regularisation: the module attribute `fp16_pair` (True / False) if the caller set one, else RCMVS_FP16_PAIR
        # (read per forward so that a process can switch), else the default
        pair = getattr(self, "fp16_pair", None)
        if pair is None:
            pair = FP16_PAIR_DEFAULT if os.environ.get("RCMVS_FP16_PAIR") is None else os.environ["RCMVS_FP16_PAIR"] == "1"
        if pair and B == 1:          # (the bounds are per launch: with B > 1 a sample's rounding would depend on its batch mates -> exact form)
            # activation bounds of the fp16-pair kernels: one persistent (stage, BOUND_ROWS, 1024) buffer per model, cleared ONCE per scene -- by the
            # homography launch below, on the side -- (row 0 of a stage: bound of the variance volume, left there by FeatureNet's output
            # conv; the other rows: written by the layers).  Not re-entrant across streams.
            bounds = getattr(self, "_pair_bounds", None)
            if getattr(self, "_is_replica", False):                  # a DataParallel replica shares its attributes with its siblings (shallow copy) and runs beside them
                bounds = torch.zeros(self.num_stage, CostRegNet.BOUND_ROWS, ops.ABSMAX_FLOATS, device=imgs.device, dtype=torch.float32)
            elif bounds is None or bounds.device != imgs.device:
                bounds = self._pair_bounds = torch.zeros(self.num_stage, CostRegNet.BOUND_ROWS, ops.ABSMAX_FLOATS, device=imgs.device, dtype=torch.float32)
            else:
                need_fill = True
        if homographies is None:                                     # every stage's homographies in one launch (they differ in the intrinsics scale only)
            if self.num_stage <= 4:
                rots, transs = ops.compose_homography_stages([proj_matrices["stage{}".format(k + 1)].contiguous().float() for k in range(self.num_stage)],
                                                             zero=bounds if need_fill else None)
                need_fill = False
            else:
                rt = [ops.compose_homography(proj_matrices["stage{}".format(k + 1)].contiguous().float()) for k in range(self.num_stage)]
                rots, transs = [r for r, _ in rt], [t for _, t in rt]
        if need_fill:
            bounds.zero_()
        for s in range(self.num_stage):
            key = "stage{}".format(s + 1)
            scale = int(self.stage_infos[key]["scale"])
            D = self.ndepths[s]
            bound_kept = False
            if feats_cl is not None:
                # this stage's output conv, just in time; with the fp16-pair form it also leaves the bound of the variance volume
                # ((max|f|)^2 over the V maps) in row 0 of the stage's bounds
                fk, bound_kept = feats_cl[key](bounds[s][0] if bounds is not None else None)
                h, w, C = fk.shape[1:]
                f_cl = fk.view(B, V, h, w, C)
            else:
                C, h, w = features[key].shape[1:]
                f_cl = ops.to_channels_last(features[key].contiguous()).view(B, V, h, w, C)
            if homographies is not None:
                rot, trans = homographies[key]
            else:
                rot, trans = rots[s], transs[s]
            planes = ops.hypothesis_planes(depth, depth_values, (H, W), scale, D, self.depth_interals_ratio[s])
            # stage 1 sweeps the same planes at every pixel (models/modules.py:549-566): K1 can stage its source windows in LDS
            var = ops.warp_variance(f_cl, rot, trans, planes, D, uniform_planes=depth is None)
            cr = self._cr(s)
            # The cost regularisation runs on the two-piece fp16 form of the matrix-core kernels (half the MFMAs of the exact bf16
            # triple; RCMVS_FP16_PAIR=0 selects the exact form).  It needs a bound of max|var|: var = E[f^2] - E[f]^2 <= max f^2, from
            # the feature maps, no pass over the volume; the layers keep the bounds of their outputs themselves.
            vmax = None
            if bounds is not None:
                vmax = bounds[s]
                if not bound_kept:                                   # (test hooks, pyramids whose output conv keeps no bound)
                    ops.absmax(f_cl, square=True, out=vmax[0])
            plan = cr.hip_plan()
            if vmax is not None and HEAD_PAIR and (CONV11_PROB == 2 or (CONV11_PROB == 1 and D == 8)):
                # conv11 + prob conv in one pass (csrc/conv11_prob.hip), then softmax / soft-argmin / confidence over the logits
                depth, conf = cr.features_cl(var, vmax, plan, logits=True, planes=planes)
            else:
                x8 = cr.features_cl(var, vmax, plan)
                # ... and so does the depth head (prob conv on the matrix cores, csrc/prob_pair.hip; RCMVS_HEAD_PAIR=0: the fp32 form)
                depth, conf = ops.depth_head(x8, plan["prob"], planes, x_absmax=vmax[CostRegNet.BOUND_ROWS - 1] if vmax is not None and HEAD_PAIR else None)
            out = {"depth": depth, "photometric_confidence": conf}
            if self.TRAIN_VARIANT:
                small_cl = ops.resize_rgb_cl(imgs.reshape(B * V, 3, H, W).contiguous(), (h, w)).view(B, V, h, w, 3)
                # module is in eval mode on this path -> the reference's in-place pow_ quirk applies
                out["volume_feature_no_ref"] = ops.warp_noref(f_cl, small_cl, rot, trans, planes, D, square_first=True)
            outputs[key] = out
            outputs.update(out)
        return outputs

    # ---------------------------------------------------------------- native training path
    def _forward_train_hip(self, imgs, proj_matrices, depth_values):
        """Train mode with autograd: the feature pyramid and every op of the three cascade stages -- warp + variance (and
        the train variant's volume_feature_no_ref), the cost regularisation with batch-statistics BatchNorm, prob conv +
        softmax + soft-argmin -- run forward AND backward on the HIP kernels (ops.WarpVarianceFn, train_ops.*).  The
        pyramid sees all V views in one pass but normalises them per view, like models/casmvsnet.py:364-366."""
        from . import train_ops
        B, V, _, H, W = imgs.shape
        imgs = imgs.float()
        depth_values = depth_values.contiguous().float()
        # all V views in one pass, normalised per view (segments) like the reference's per-view calls (casmvsnet.py:364-366)
        fv = self.feature.forward_train_cl(imgs.transpose(0, 1).reshape(V * B, 3, H, W), segments=V)
        features = [{k: f[v * B:(v + 1) * B] for k, f in fv.items()} for v in range(V)]
        outputs = {}
        depth = None
        for s in range(self.num_stage):
            key = "stage{}".format(s + 1)
            scale = int(self.stage_infos[key]["scale"])
            D = self.ndepths[s]
            f_cl = torch.stack([f[key] for f in features], dim=1).contiguous()                       # (B,V,h,w,C), differentiable
            h, w = f_cl.shape[2:4]
            with torch.no_grad():
                rot, trans = ops.compose_homography(proj_matrices[key].contiguous().float())
                prev = depth.detach() if depth is not None else None
                planes = ops.hypothesis_planes(prev, depth_values, (H, W), scale, D, self.depth_interals_ratio[s])
                small_cl = None
                if self.TRAIN_VARIANT:
                    small_cl = ops.resize_rgb_cl(imgs.reshape(B * V, 3, H, W).contiguous(), (h, w)).view(B, V, h, w, 3)
            res = ops.WarpVarianceFn.apply(f_cl, rot, trans, planes, D, small_cl)
            var, noref = res if self.TRAIN_VARIANT else (res, None)
            cr = self._cr(s)
            x8 = cr.features_cl_train(var)
            prev_live = depth
            depth, conf = train_ops.ProbDepthHeadFn.apply(x8, cr.prob.weight, planes)
            if prev_live is not None and self.grad_method != "detach":
                # grad_method='undetach' (models/casmvsnet.py:192): the hypothesis planes are prev-depth + constants, resized
                # linearly (weights sum to 1), and the probabilities sum to 1, so d depth / d planes reaches the previous
                # stage's depth as the adjoint of "bilinear up to (H,W), bilinear down to (h,w)"; the warp coordinates carry
                # no gradient in the reference either (modules.py:313).  Value unchanged, gradient through two tiny resizes.
                up = F.interpolate(prev_live.unsqueeze(1), [H, W], mode="bilinear", align_corners=False)
                base = F.interpolate(up, [h, w], mode="bilinear", align_corners=False).squeeze(1)
                depth = depth + (base - base.detach())
            out = {"depth": depth, "photometric_confidence": conf}
            if self.TRAIN_VARIANT:
                out["volume_feature_no_ref"] = noref
            outputs[key] = out
            outputs.update(out)
        return outputs

    def _run(self, imgs, proj_matrices, depth_values):
        if _hip_inference(self, imgs, depth_values):
            return self._forward_hip(imgs, proj_matrices, depth_values)
        if _hip_training(self, imgs, depth_values):
            return self._forward_train_hip(imgs, proj_matrices, depth_values)
        _unsupported(self, imgs, depth_values)


class CascadeMVSNet_eval(_CascadeBase):
    """models/casmvsnet.py:313-417."""
    TRAIN_VARIANT = False

    def forward(self, imgs, proj_matrices, depth_values):
        return self._run(imgs, proj_matrices, depth_values)


class CascadeMVSNet(_CascadeBase):
    """models/casmvsnet.py:126-231: returns (outputs, stage-1 volume_feature_no_ref)."""
    TRAIN_VARIANT = True

    def forward(self, imgs, proj_matrices, depth_values):
        outputs = self._run(imgs, proj_matrices, depth_values)
        return outputs, outputs["stage1"]["volume_feature_no_ref"]
