"""Drop-in for the reference's ``models.render_consist_net.Rendering_Consistency_Net``
(models/render_consist_net.py:11-76) with its children ``Neural_Volume_Net`` / ``CostReg``
(models/render_models.py:690-760) and ``RenderNet`` / ``Renderer_ours`` (:143-220,538-565).

Same constructor (``args`` namespace; ``args.feat_dim`` is written like the reference does), same
``forward(volume_feature_warp, pseudo_depth, batch)`` 8-tuple and the same 82 ``state_dict`` names
(``MVSNet.cost_reg_2.*``, ``network_fn.nerf.*``), so ``train_rcmvsnet.py`` and the shipped
``model_000014_nerf.ckpt`` work unchanged.

Execution mirrors casmvsnet.py: eval() under no_grad runs on the HIP kernels (plane resize,
neural-volume U-Net on the 3-D conv family without ReLU, Gaussian-Uniform sampler, point features,
MFMA MLP, wave-scan compositing).  In train mode on the GPU the volume network, the point-feature
gather / scatter, the NeRF MLP (forward, data and weight gradients on the MFMA chain) and the compositing run forward and
backward on the library through autograd Functions (train_ops.py).  CPU tensors or eval mode with autograd enabled raise
RcmvsError: there is no PyTorch op graph behind these modules (the tests' comparator lives in oracle/aten_graph.py).
Random draws (pixel indices, Gaussian eps, stratified u) come from torch's generator on the device and
are passed INTO the sampler kernel -- the RNG contract of SURVEY.md 8a-9; ``forward`` accepts them
through the optional ``randoms=(pix, eps, u)`` argument so tests can inject the reference's draws.
"""
import torch
import torch.nn as nn
import torch.nn.functional as F

from . import ops
from .casmvsnet import RcmvsError, _block3d, _bn_fold, _hip_inference, _hip_training, _holder_only, _unsupported

N_RAYS = 1024                               # hard-coded in the reference (render_consist_net.py:68)
IMAGENET_MEAN = (0.485, 0.456, 0.406)
IMAGENET_STD = (0.229, 0.224, 0.225)


class ConvBnReLU3D(nn.Module):
    """conv + norm, NO ReLU despite the name (models/render_models.py:675-686)."""

    def __init__(self, in_channels, out_channels, kernel_size=3, stride=1, pad=1, norm_act=nn.BatchNorm3d):
        super().__init__()
        self.conv = nn.Conv3d(in_channels, out_channels, kernel_size, stride=stride, padding=pad, bias=False)
        self.bn = norm_act(out_channels)
        self.relu = False
        self.gn = None

    def forward(self, x):
        """Called on its own (inside CostReg the network's plan runs it): (B,Ci,D,H,W) -> (B,Co,Do,Ho,Wo) on the 3-D conv family, eval mode
        under no_grad (folded norm) or train mode (batch statistics + autograd); a channel count that is not a multiple of 4 (the 41
        channels of the volume network's first layer) is zero-padded on both operands."""
        pad = (-x.shape[1]) % 4
        if pad and _hip_inference(self, x):
            w = self.conv.weight
            key = (w.data_ptr(), w._version, str(w.device))
            c = self.__dict__.get("_rcmvs_pack")
            if c is None or c[0] != key:
                wp = torch.cat((w.detach(), w.new_zeros(w.shape[0], pad, *w.shape[2:])), dim=1)
                self.__dict__["_rcmvs_pack"] = (key, ops.pack_conv3d_weight(wp))
            x = F.pad(x, (0, 0, 0, 0, 0, 0, 0, pad))
        return _block3d(self, x)


class CostReg(nn.Module):
    """models/render_models.py:690-734."""

    _LAYERS = ("conv0", "conv1", "conv2", "conv3", "conv4", "conv5", "conv6", "conv7", "conv9", "conv11")

    def __init__(self, in_channels, norm_act=nn.BatchNorm3d, base_channels=4):
        super().__init__()
        b = base_channels
        self.in_channels = in_channels
        self.conv0 = ConvBnReLU3D(in_channels, b, norm_act=norm_act)
        self.conv1 = ConvBnReLU3D(b, b * 2, stride=2, norm_act=norm_act)
        self.conv2 = ConvBnReLU3D(b * 2, b * 2, norm_act=norm_act)
        self.conv3 = ConvBnReLU3D(b * 2, b * 4, stride=2, norm_act=norm_act)
        self.conv4 = ConvBnReLU3D(b * 4, b * 4, norm_act=norm_act)
        self.conv5 = ConvBnReLU3D(b * 4, b * 8, stride=2, norm_act=norm_act)
        self.conv6 = ConvBnReLU3D(b * 8, b * 8, norm_act=norm_act)
        self.conv7 = nn.Sequential(nn.ConvTranspose3d(b * 8, b * 4, 3, padding=1, output_padding=1, stride=2, bias=False), norm_act(32))
        self.conv9 = nn.Sequential(nn.ConvTranspose3d(b * 4, b * 2, 3, padding=1, output_padding=1, stride=2, bias=False), norm_act(16))
        self.conv11 = nn.Sequential(nn.ConvTranspose3d(b * 2, b, 3, padding=1, output_padding=1, stride=2, bias=False), norm_act(8))
        self._plan = None
        self._plan_key = None

    def _conv_bn(self, name):
        m = getattr(self, name)
        return (m[0], m[1], True) if isinstance(m, nn.Sequential) else (m.conv, m.bn, False)

    def hip_plan(self):
        ts = []
        for n in self._LAYERS:
            conv, bn, _ = self._conv_bn(n)
            ts += [conv.weight, bn.weight, bn.bias, bn.running_mean, bn.running_var]
            if getattr(bn, "num_batches_tracked", None) is not None:      # bumped by every train-mode forward (the kernels update the statistics in place)
                ts.append(bn.num_batches_tracked)
        key = tuple((t.data_ptr(), t._version) for t in ts)
        if self._plan is None or key != self._plan_key:
            plan = {}
            for n in self._LAYERS:
                conv, bn, transposed = self._conv_bn(n)
                w = conv.weight.detach()
                if n == "conv0" and self.in_channels % 4:                  # 41 -> 44 zero channels: 16-byte input vectors
                    pad = 4 - self.in_channels % 4
                    w = torch.cat((w, w.new_zeros(w.shape[0], pad, 3, 3, 3)), dim=1)
                plan[n] = (ops.pack_conv3d_weight(w, transposed=transposed),) + _bn_fold(bn)
            self._plan, self._plan_key = plan, key
        return self._plan

    def forward_cl(self, x):
        """x (B,D,h,w,Cin padded to a multiple of 4) channels-last -> (B,D,h,w,8)."""
        p = self.hip_plan()
        conv0 = ops.conv3d(x, *p["conv0"])
        conv2 = ops.conv3d(ops.conv3d(conv0, *p["conv1"], stride=2), *p["conv2"])
        conv4 = ops.conv3d(ops.conv3d(conv2, *p["conv3"], stride=2), *p["conv4"])
        t = ops.conv3d(ops.conv3d(conv4, *p["conv5"], stride=2), *p["conv6"])
        t = ops.deconv3d(t, *p["conv7"], residual=conv4)
        t = ops.deconv3d(t, *p["conv9"], residual=conv2)
        return ops.deconv3d(t, *p["conv11"], residual=conv0)

    def forward_cl_train(self, x):
        """Train-mode twin of ``forward_cl``: batch-statistics norm layers, autograd through the HIP kernels."""
        from .train_ops import conv_bn_train

        def blk(name, t, residual=None):
            conv, bn, _ = self._conv_bn(name)
            return conv_bn_train(conv, bn, t, relu=False, residual=residual)

        conv0 = blk("conv0", x)
        conv2 = blk("conv2", blk("conv1", conv0))
        conv4 = blk("conv4", blk("conv3", conv2))
        t = blk("conv6", blk("conv5", conv4))
        t = blk("conv7", t, residual=conv4)
        t = blk("conv9", t, residual=conv2)
        return blk("conv11", t, residual=conv0)

    def forward(self, x):
        """x (B,C,D,h,w) -> (B,8,D,h,w) like the reference module (models/render_models.py:720-734), on the HIP kernels."""
        if _hip_inference(self, x):
            C = x.shape[1]
            xcl = ops.to_channels_last(x.contiguous().float())
            if C % 4:
                xcl = F.pad(xcl, (0, 4 - C % 4))
            return ops.to_channels_first(self.forward_cl(xcl))
        _unsupported(self, x)


class Neural_Volume_Net(nn.Module):
    """models/render_models.py:736-760: D -> 128 planes (trilinear, align_corners), then CostReg(32+9)."""

    def __init__(self, num_groups=1, norm_act=nn.BatchNorm3d, levels=1, in_channels=32 + 9):
        super().__init__()
        self.levels = levels
        self.n_depths = [128, 32, 8]
        self.G = num_groups
        self.N_importance = 0
        self.chunk = 1024
        self.cost_reg_2 = CostReg(in_channels, norm_act, base_channels=8)

    def forward_cl(self, volume_feature):
        """NCDHW in -> (B,128,h,w,8) channels-last (HIP path)."""
        C = volume_feature.shape[1]
        x = ops.resize_planes(volume_feature.contiguous().float(), 128, pad_channels_to=(C + 3) // 4 * 4)
        return self.cost_reg_2.forward_cl(x)

    def hip_trainable(self, volume_feature):
        """Train mode on the GPU with 5-D-capable norm layers (the reference's BatchNorm2d default only works after
        SyncBatchNorm conversion, train_rcmvsnet.py:525)."""
        return _hip_training(self, volume_feature) and all(isinstance(m, (nn.BatchNorm3d, nn.SyncBatchNorm)) for m in self.modules()
                                                           if isinstance(m, nn.modules.batchnorm._NormBase))

    def forward_cl_train(self, volume_feature):
        """Train-mode twin of ``forward_cl``: autograd through the HIP kernels (train_ops)."""
        from .train_ops import ResizePlanesFn
        C = volume_feature.shape[1]
        x = ResizePlanesFn.apply(volume_feature, 128, (C + 3) // 4 * 4)
        return self.cost_reg_2.forward_cl_train(x)

    def forward(self, volume_feature, pad=0):
        if _hip_inference(self, volume_feature):
            return ops.to_channels_first(self.forward_cl(volume_feature)).reshape(1, -1, 128, *volume_feature.shape[-2:])
        if self.hip_trainable(volume_feature):
            v = self.forward_cl_train(volume_feature).permute(0, 4, 1, 2, 3)      # NCDHW view of the channels-last result
            return v.reshape(1, -1, *v.shape[2:])
        _unsupported(self, volume_feature)


class Renderer_ours(nn.Module):
    """models/render_models.py:143-220 (use_viewdirs=True head)."""

    def __init__(self, D=8, W=256, input_ch=3, input_ch_views=3, output_ch=4, input_ch_feat=8, skips=[4], use_viewdirs=False):
        super().__init__()
        self.D, self.W = D, W
        self.input_ch, self.input_ch_views, self.skips, self.use_viewdirs = input_ch, input_ch_views, skips, use_viewdirs
        self.in_ch_pts, self.in_ch_views, self.in_ch_feat = input_ch, input_ch_views, input_ch_feat
        self.pts_linears = nn.ModuleList([nn.Linear(input_ch, W, bias=True)] +
                                         [nn.Linear(W, W, bias=True) if i not in skips else nn.Linear(W + input_ch, W) for i in range(D - 1)])
        self.pts_bias = nn.Linear(input_ch_feat, W)
        self.views_linears = nn.ModuleList([nn.Linear(input_ch_views + W, W // 2)])
        if use_viewdirs:
            self.feature_linear = nn.Linear(W, W)
            self.alpha_linear = nn.Linear(W, 1)
            self.rgb_linear = nn.Linear(W // 2, 3)
        else:
            self.output_linear = nn.Linear(W, output_ch)
        for m in self.modules():
            if isinstance(m, nn.Linear):
                nn.init.kaiming_normal_(m.weight.data)
                nn.init.zeros_(m.bias.data)

        self._blob = None
        self._blob_key = None

    def hip_blob(self):
        """The eleven layers as one packed MFMA weight blob (rebuilt when a parameter changes)."""
        named = {"pts_bias": self.pts_bias, "alpha_linear": self.alpha_linear, "feature_linear": self.feature_linear,
                 "views_linears.0": self.views_linears[0], "rgb_linear": self.rgb_linear}
        for i in range(6):
            named[f"pts_linears.{i}"] = self.pts_linears[i]
        key = tuple((m.weight.data_ptr(), m.weight._version, m.bias._version) for m in named.values())
        if self._blob is None or key != self._blob_key:
            self._blob = ops.pack_nerf_weights({k: (m.weight, m.bias) for k, m in named.items()})
            self._blob_key = key
        return self._blob

    def _mvs_form(self):
        return (self.use_viewdirs and self.D == 6 and self.W == 128 and self.in_ch_pts == 63 and self.in_ch_feat == 20 and self.in_ch_views == 3
                and list(self.skips) == [4])

    def forward(self, x):
        """models/render_models.py:192-220 on its own: x (..., 63 + 20 + 3) = [embedded point | point feature | view direction] ->
        (..., 4) = [rgb, sigma].  Inference only (inside the rendering branch the layers run fused with the embedding and, in
        training, through train_ops.nerf_mlp_train); the form create_nerf_mvs builds (D = 6, W = 128, 63 / 20 / 3 columns)."""
        if not self._mvs_form():
            raise RcmvsError("Renderer_ours.forward: only the configuration of create_nerf_mvs runs on the HIP kernels "
                             "(D=6, W=128, input_ch=63, input_ch_feat=20, input_ch_views=3, skips=[4], use_viewdirs=True)")
        if not _hip_inference(self, x):
            _unsupported(self, x)
        if x.shape[-1] != 86:
            raise RcmvsError(f"Renderer_ours.forward: rows of 63 + 20 + 3 = 86 columns expected (got {x.shape[-1]})")
        raw = ops.nerf_mlp_embedded(x.reshape(-1, 86).contiguous().float(), self.hip_blob())
        return raw.reshape(*x.shape[:-1], 4)

    def forward_alpha(self, x):
        """models/render_models.py:175-190: sigma only, from rows without the view direction (..., 63 + 20) -> (..., 1)."""
        if x.shape[-1] != 83:
            raise RcmvsError(f"Renderer_ours.forward_alpha: rows of 63 + 20 = 83 columns expected (got {x.shape[-1]})")
        xv = torch.cat((x, torch.zeros(*x.shape[:-1], 3, device=x.device, dtype=x.dtype)), dim=-1)      # sigma does not depend on the direction
        return self.forward(xv)[..., 3:4]


class RenderNet(nn.Module):
    """models/render_models.py:538-565, net_type 'v0'."""

    def __init__(self, D=8, W=256, input_ch_pts=3, input_ch_views=3, input_ch_feat=8, skips=[4], net_type="v0"):
        super().__init__()
        if net_type != "v0":
            raise NotImplementedError("only net_type='v0' (the reference's default) is provided")
        self.in_ch_pts, self.in_ch_views, self.in_ch_feat = input_ch_pts, input_ch_views, input_ch_feat
        self.nerf = Renderer_ours(D=D, W=W, input_ch_feat=input_ch_feat, input_ch=input_ch_pts, output_ch=4, skips=skips,
                                  input_ch_views=input_ch_views, use_viewdirs=True)

    def hip_blob(self):
        return self.nerf.hip_blob()

    def forward_alpha(self, x):
        return self.nerf.forward_alpha(x)

    def forward(self, x):
        return self.nerf(x)


class Rendering_Consistency_Net(nn.Module):
    """models/render_consist_net.py:11-76.  One flagged extension beyond the reference: ``args.num_views`` (default 4).  The reference's
    volume network is built for the warped volume feature of a FOUR-view CascadeMVSNet pass (CostReg(32+9), models/render_models.py:750),
    so its training script cannot feed it a five-view pass (BASELINE configs[2] as worded); with ``args.num_views = V`` the volume network
    takes the 32 + 3 (V - 1) channels of a V-view pass.  Everything after it is the reference's, quirks included: decode_batch's
    idx = arange(4) selects nothing (models/render_utils.py:378 tests the key, not the tensor), so the renderer reads the images of the
    LAST three views of the batch (imgs[:, -3:], :74) with the poses of the FIRST three (render_utils.py:260)."""

    def __init__(self, args):
        super().__init__()
        self.args = args
        self.args.feat_dim = 8 + 3 * 4
        self.idx = 0
        self.num_views = int(getattr(args, "num_views", 4))
        if self.num_views < 4:
            raise NotImplementedError("the renderer reads three source views (models/render_consist_net.py:74): num_views >= 4")
        if getattr(args, "net_type", "v0") != "v0" or getattr(args, "N_importance", 0) != 0:
            raise NotImplementedError("only net_type='v0', N_importance=0 (the shipped configuration) is provided")
        if getattr(args, "multires", 10) != 10 or getattr(args, "netdepth", 6) != 6 or getattr(args, "netwidth", 128) != 128:
            raise NotImplementedError("the MLP kernels are built for multires=10, netdepth=6, netwidth=128 (the shipped configuration)")
        self.MVSNet = Neural_Volume_Net(in_channels=32 + 3 * (self.num_views - 1))
        self.network_fn = RenderNet(D=args.netdepth, W=args.netwidth, input_ch_pts=63, skips=[4], input_ch_views=args.dir_dim,
                                    input_ch_feat=self.args.feat_dim, net_type=args.net_type)
        self.white_bkgd = getattr(args, "white_bkgd", False)

    # ---- helpers shared by both paths ----------------------------------------------------------
    @staticmethod
    def _unpreprocess(imgs):
        mean = torch.tensor([-m / s for m, s in zip(IMAGENET_MEAN, IMAGENET_STD)], device=imgs.device).view(1, 1, 3, 1, 1)
        std = torch.tensor([1 / s for s in IMAGENET_STD], device=imgs.device).view(1, 1, 3, 1, 1)
        return (imgs - mean) / std

    def _draw(self, H, W, S, dev):
        xs = torch.randint(0, W, (N_RAYS,), device=dev)
        ys = torch.randint(0, H, (N_RAYS,), device=dev)
        return torch.stack((xs, ys)), torch.randn(N_RAYS, S, device=dev), torch.rand(N_RAYS // 2, S, device=dev)

    def forward(self, volume_feature_warp, pseudo_depth, batch, randoms=None):
        if "scan" in batch:
            batch.pop("scan")
        dev = volume_feature_warp.device
        # (decode_batch's idx = arange(4) selects nothing: sub_selete_data tests torch.is_tensor on the KEY, models/render_utils.py:378 --
        # every view of the batch reaches the renderer, which reads the LAST three images and the FIRST three poses)
        imgs = batch["imgs"].float().to(dev)
        w2cs = batch["w2cs"].float().to(dev).squeeze(0)
        c2ws = batch["c2ws"].float().to(dev).squeeze(0)
        intr = batch["intrinsics"].float().to(dev).squeeze(0)
        nf = batch["near_fars"].float().to(dev).squeeze(0)
        _, V, _, H, W = imgs.shape
        S = self.args.N_samples
        pix, eps, u = randoms if randoms is not None else self._draw(H, W, S, dev)
        imgs = self._unpreprocess(imgs)
        pseudo = pseudo_depth.reshape(H, W).float()
        if _hip_inference(self, volume_feature_warp, pseudo):
            return self._forward_hip(volume_feature_warp, pseudo, imgs, w2cs, c2ws, intr, nf, pix, eps, u)
        if self.MVSNet.hip_trainable(volume_feature_warp) and _hip_training(self.MVSNet, pseudo):
            return self._forward_train_hip(volume_feature_warp, pseudo, imgs, w2cs, c2ws, intr, nf, pix, eps, u)
        _unsupported(self, volume_feature_warp, pseudo)

    # ---- training path: HIP kernels with autograd ---------------------------------------------------------
    def _forward_train_hip(self, vfw, pseudo, imgs, w2cs, c2ws, intr, nf, pix, eps, u):
        """Same data flow as ``_forward_hip``.  Differentiable w.r.t. the warped volume feature and every parameter:
        volume network (train_ops.ConvBnReluFn ...), point features (PointFeatsFn: trilinear scatter), compositing
        (CompositeFn: reverse recurrence) and the NeRF MLP (NerfMlpFn: eleven layers, data and weight gradients on the MFMA
        chain) run forward and backward on the library.  Rays, samples and image taps carry no gradient."""
        from .train_ops import CompositeFn, PointFeatsFn, nerf_mlp_train
        vol = self.MVSNet.forward_cl_train(vfw)[0]                                 # (128,h,w,8)
        with torch.no_grad():
            cam = torch.cat((intr[0].reshape(-1), c2ws[0].reshape(-1), w2cs[0].reshape(-1), intr[0].reshape(-1), nf[0])).contiguous()
            z, pts, ndc, dirs, rdepth, target = ops.gu_sample(pseudo.contiguous(), imgs[0, 0].contiguous(),
                                                              pix.to(torch.int32).contiguous(), eps.contiguous(), u.contiguous(), cam)
            imgs3 = imgs[0, -3:].contiguous()
            poses = torch.cat((w2cs[:3].reshape(3, 16), intr[:3].reshape(3, 9)), dim=1).contiguous()
        S = z.shape[1]
        feat32 = PointFeatsFn.apply(vol, imgs3, poses, pts, ndc, 32)               # (M,32), 20 used columns
        raw = nerf_mlp_train(self.network_fn.nerf, ndc, feat32, dirs, w2cs[0])
        feat = feat32[:, :20].reshape(N_RAYS, S, 20)
        rgb, depth, weights, alpha = CompositeFn.apply(raw, z)
        if self.white_bkgd:
            rgb = rgb + (1.0 - torch.sum(weights, -1)[..., None])
        return rgb, feat, weights, depth, alpha, {}, rdepth, target

    # ---- native path ------------------------------------------------------------------------------
    def _forward_hip(self, vfw, pseudo, imgs, w2cs, c2ws, intr, nf, pix, eps, u):
        vol = self.MVSNet.forward_cl(vfw)[0].contiguous()                       # (128,h,w,8)
        cam = torch.cat((intr[0].reshape(-1), c2ws[0].reshape(-1), w2cs[0].reshape(-1), intr[0].reshape(-1), nf[0])).contiguous()
        z, pts, ndc, dirs, rdepth, target = ops.gu_sample(pseudo.contiguous(), imgs[0, 0].contiguous(), pix.to(torch.int32).contiguous(),
                                                          eps.contiguous(), u.contiguous(), cam)
        # quirk reproduced (render_consist_net.py:74 vs render_utils.py:260): images of views 1..3, poses of views 0..2
        imgs3 = imgs[0, -3:].contiguous()
        poses = torch.cat((w2cs[:3].reshape(3, 16), intr[:3].reshape(3, 9)), dim=1).contiguous()
        feat = ops.point_feats(vol, imgs3, poses, pts, ndc, ldf=32)
        raw = ops.nerf_mlp(ndc, feat, dirs, w2cs[0].contiguous(), self.network_fn.hip_blob())
        rgb, depth, weights, alpha = ops.composite(raw, z)
        S = z.shape[1]
        return rgb, feat[:, :20].reshape(N_RAYS, S, 20), weights, depth, alpha, {}, rdepth, target
