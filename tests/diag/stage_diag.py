#!/usr/bin/env python
"""Diagnostic (GPU box): isolate each HIP kernel's numerical error at BASELINE config-2 stage-1 size
by feeding it the CPU oracle's exact inputs; fp64 CPU references give the 'true' values."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import torch.nn.functional as F
from rc_mvsnet_amd import synthetic, ops
from rc_mvsnet_amd.casmvsnet import CascadeMVSNet_eval
from oracle import warp, conv3d as oc, depth_head as od, cascade
from oracle.feature_net import feature_net
torch.set_num_threads(16)
dev = "cuda:0"
H, W, V = 512, 640, 3
sd = synthetic.cascade_state_dict(0)
imgs, pm, dv = synthetic.cascade_inputs(1, V, H, W, 0)
m = CascadeMVSNet_eval(); m.load_state_dict(sd, strict=True); m = m.to(dev).eval()
stage = int(sys.argv[1]) if len(sys.argv) > 1 else 1
key = f"stage{stage}"; sc = {1: 4, 2: 2, 3: 1}[stage]; D = {1: 48, 2: 32, 3: 8}[stage]; h, w = H // sc, W // sc
with torch.no_grad():
    feats = [feature_net(imgs[:, v], sd)[key] for v in range(V)]
    prev = None
    if stage > 1:
        prev = 600.0 + 50.0 * torch.rand(1, h // 2, w // 2, generator=torch.Generator().manual_seed(1))
    samples = warp.stage_samples(prev, dv, D, {1: 4, 2: 2, 3: 1}[stage], (H, W), (h, w))
    # ---- K1
    var_ref = warp.variance_volume(feats, pm[key], samples)
    f_cl = torch.stack([f.permute(0, 2, 3, 1) for f in feats], dim=1).contiguous().to(dev)
    rots, transs = zip(*[warp.compose_homography(pm[key][:, v], pm[key][:, 0]) for v in range(1, V)])
    rot = torch.stack([r.reshape(1, 9) for r in rots], dim=1).to(dev); trans = torch.stack(transs, dim=1).to(dev)
    planes = ops.hypothesis_planes(prev.to(dev) if prev is not None else None, dv.to(dev), (H, W), sc, D, {1: 4, 2: 2, 3: 1}[stage])
    k = torch.arange(D, dtype=torch.float32).reshape(1, D, 1, 1)
    pl = planes.cpu(); smp_hip = pl[..., 0].unsqueeze(1) + k * pl[..., 1].unsqueeze(1)
    print(f"planes: max |hip - oracle| = {float((smp_hip - samples).abs().max()):.3e} mm")
    # use the HIP planes for both so that K1 is isolated
    var_ref2 = warp.variance_volume(feats, pm[key], smp_hip)
    var_hip = ops.warp_variance(f_cl, rot, trans, planes, D)
    dvv = (var_hip.cpu().permute(0, 4, 1, 2, 3) - var_ref2).abs()
    print(f"K1 ({tuple(var_ref2.shape)}): max|d| {float(dvv.max()):.3e}  bit-identical {float((dvv == 0).float().mean()):.6f}")
    # ---- cost regularisation on the ORACLE variance
    cr = f"cost_regularization.{stage - 1}"
    t0 = time.time()
    feat_ref32 = oc.cost_reg_net(var_ref2, sd, cr, return_feat=True) if False else None
    def costreg_feat(x, sdd):
        def bn(t, name):
            return F.batch_norm(t, sdd[name + ".running_mean"], sdd[name + ".running_var"], sdd[name + ".weight"], sdd[name + ".bias"], False, 0.1, 1e-5)
        def block(t, name, stride=1):
            return torch.relu(bn(F.conv3d(t, sdd[f"{cr}.{name}.conv.weight"], None, stride, 1), f"{cr}.{name}.bn"))
        def up(t, name):
            return torch.relu(bn(F.conv_transpose3d(t, sdd[f"{cr}.{name}.conv.weight"], None, 2, 1, 1), f"{cr}.{name}.bn"))
        c0 = block(x, "conv0"); c2 = block(block(c0, "conv1", 2), "conv2"); c4 = block(block(c2, "conv3", 2), "conv4")
        t = block(block(c4, "conv5", 2), "conv6"); t = c4 + up(t, "conv7"); t = c2 + up(t, "conv9"); t = c0 + up(t, "conv11")
        return c0, t, F.conv3d(t, sdd[f"{cr}.prob.weight"], None, 1, 1)
    c0_32, f_32, lg_32 = costreg_feat(var_ref2, sd)
    sd64 = {kk: (vv.double() if vv.is_floating_point() else vv) for kk, vv in sd.items()}
    c0_64, f_64, lg_64 = costreg_feat(var_ref2.double(), sd64)
    net = m.cost_regularization[stage - 1]
    p = net.hip_plan()
    xcl = var_ref2.permute(0, 2, 3, 4, 1).contiguous().to(dev)
    c0_hip = ops.conv3d(xcl, *p["conv0"], relu=True).cpu().permute(0, 4, 1, 2, 3)
    f_hip_cl = net.features_cl(xcl)
    f_hip = f_hip_cl.cpu().permute(0, 4, 1, 2, 3)
    lg_hip = ops.conv3d(f_hip_cl, p["prob"]).cpu().permute(0, 4, 1, 2, 3)
    def rep(name, a32, ahip, a64):
        s = float(a64.abs().max())
        print(f"{name}: |true|max {s:.3f}   cpu-fp32 err max {float((a32 - a64).abs().max()):.3e} mean {float((a32 - a64).abs().mean()):.3e}"
              f"   HIP err max {float((ahip - a64).abs().max()):.3e} mean {float((ahip - a64).abs().mean()):.3e}")
    rep("conv0 ", c0_32, c0_hip, c0_64)
    rep("feat8 ", f_32, f_hip, f_64)
    rep("logits", lg_32, lg_hip, lg_64)
    # ---- depth head on the ORACLE logits (1-hot prob conv trick)
    depth_ref, conf_ref, p_ref = od.depth_head(lg_32.squeeze(1), smp_hip)
    depth64 = (torch.softmax(lg_32.squeeze(1).double(), 1) * smp_hip.double()).sum(1)
    x = torch.zeros(1, D, h, w, 8); x[..., 0] = lg_32.squeeze(1)
    wprob = torch.zeros(1, 8, 3, 3, 3); wprob[0, 0, 1, 1, 1] = 1.0
    dep, conf = ops.depth_head(x.to(dev), ops.pack_conv3d_weight(wprob.to(dev)), planes)
    print(f"depth head: cpu-fp32 err max {float((depth_ref - depth64).abs().max()):.3e}   HIP err max {float((dep.cpu() - depth64).abs().max()):.3e} mm")
    # full stage from HIP logits
    dep2, _ = ops.depth_head(f_hip_cl, p["prob"], planes)
    d_true = (torch.softmax(lg_64.squeeze(1), 1) * smp_hip.double()).sum(1)
    d_cpu = od.depth_head(lg_32.squeeze(1), smp_hip)[0]
    print(f"stage depth vs fp64 chain: cpu-fp32 max {float((d_cpu - d_true).abs().max()):.3e} mean {float((d_cpu - d_true).abs().mean()):.3e}"
          f"   HIP max {float((dep2.cpu() - d_true).abs().max()):.3e} mean {float((dep2.cpu() - d_true).abs().mean()):.3e} mm")
