"""Developer probe (GPU box): where does the 1e-3 outlier of the FeatureNet parameter gradients (profiles/r2_grad_probe.txt) enter?
K1's backward in isolation at the fixture's stage-1 shape, on the fixture's own feature maps: d loss / d feature maps of
   (a) the HIP kernel (rcmvs_warp_variance_bwd, fp32 sampling positions = the reference's fp32 op chain),
   (b) torch autograd through the reference op chain in fp32 on this GPU (F.grid_sample),
against fp64 autograd with (1) the SAME fp32 sampling positions and (2) positions from the chain evaluated in fp64.
If (a) matches (1) to ~1e-6 but is ~1e-4 off (2), the backward kernel is exact and the gradient is simply that sensitive to the
last bits of the sampling positions; (b) tells what the reference's own fp32 graph does under the same measure."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import torch.nn.functional as F
from oracle import aten_graph, warp as ow
from rc_mvsnet_amd import _lib, ops, synthetic
from rc_mvsnet_amd.casmvsnet import CascadeMVSNet

_lib.load()
dev = "cuda:0"
H, W, V, D = 128, 160, 3, 16
m = CascadeMVSNet(ndepths=[16, 16, 8], depth_interals_ratio=[4, 2, 1]).train()
m.load_state_dict(synthetic.cascade_state_dict(0, prob_gain=1.0), strict=True)
imgs, pm, dv = synthetic.cascade_inputs(1, V, H, W, 0)
with torch.no_grad():
    feats = [aten_graph.feature_pyramid(m.feature, imgs[:, v])["stage1"] for v in range(V)]        # (1,32,h,w) fp32, CPU
C, h, w = feats[0].shape[1:]
proj = pm["stage1"]
samples = aten_graph.stage_samples(None, dv, D, 4, (H, W), (h, w))                                    # (1,D,h,w)
planes = torch.stack((samples[:, 0], samples[:, 1] - samples[:, 0]), dim=-1).contiguous()
g = torch.Generator().manual_seed(0)
gvar = torch.randn(1, D, h, w, C, generator=g)
gnr = torch.randn(1, C, D, h, w, generator=g)


def graph(dtype, device, pos_dtype):
    """variance + source-only variance from the feature maps through the reference op chain; sampling positions computed in
    pos_dtype (then cast), everything else in dtype."""
    f = [t.to(device=device, dtype=dtype).requires_grad_(True) for t in feats]
    ref = aten_graph._fold(proj[:, 0].to(device=device, dtype=pos_dtype))
    s = f[0].unsqueeze(2).expand(-1, -1, D, -1, -1)
    q = s ** 2
    sn = qn = 0
    for v in range(1, V):
        src = aten_graph._fold(proj[:, v].to(device=device, dtype=pos_dtype))
        grid = aten_graph.plane_sweep_grid(src, ref, samples.to(device=device, dtype=pos_dtype), h, w).to(dtype)
        warped = F.grid_sample(f[v], grid.view(1, D * h, w, 2), mode="bilinear", padding_mode="zeros", align_corners=True).view(1, C, D, h, w)
        s = s + warped
        q = q + warped ** 2
        sn = sn + warped
        qn = qn + warped ** 2
    var = q / V - (s / V) ** 2
    nr = qn / V - (sn / V) ** 2
    loss = (var.permute(0, 2, 3, 4, 1) * gvar.to(device=device, dtype=dtype)).sum() + (nr * gnr.to(device=device, dtype=dtype)).sum()
    loss.backward()
    return torch.stack([t.grad[0].permute(1, 2, 0) for t in f]).double().cpu()        # (V,h,w,C)


if not hasattr(aten_graph, "plane_sweep_grid"):
    raise SystemExit("oracle/aten_graph.py: plane_sweep_grid (the sampling grid of plane_sweep_warp) is needed by this probe")
truth_pos32 = graph(torch.float64, "cpu", torch.float32)
truth_pos64 = graph(torch.float64, "cpu", torch.float64)
ref32 = graph(torch.float32, dev, torch.float32)
f_cl = torch.stack([t[0].permute(1, 2, 0) for t in feats]).unsqueeze(0).contiguous().to(dev)
rot, trans = ops.compose_homography(proj.to(dev))
gnr_full = torch.zeros(1, 3 * (V - 1) + C, D, h, w)
gnr_full[:, -C:] = gnr
gf = ops.warp_variance_bwd(f_cl, rot, trans, planes.to(dev), gvar.to(dev), gnr_full[:, -C:].permute(0, 2, 3, 4, 1).contiguous().to(dev))[0].double().cpu()
rel = lambda a, b: float((a - b).norm() / b.norm())
print(f"fp64 truth: fp32 positions vs fp64 positions          {rel(truth_pos32, truth_pos64):.2e}   (what the last bits of the sampling positions are worth)")
for name, t in (("HIP K1 backward", gf), ("reference graph fp32 on this GPU", ref32)):
    print(f"{name:34s} vs fp64 @ fp32 positions {rel(t, truth_pos32):.2e}   vs fp64 @ fp64 positions {rel(t, truth_pos64):.2e}   "
          + "  ".join(f"view {v}: {rel(t[v], truth_pos64[v]):.2e}" for v in range(V)))
