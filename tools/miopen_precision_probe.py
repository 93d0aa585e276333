"""Is PyTorch-ROCm's (MIOpen) fp32 convolution backward exact fp32?  Compares conv2d / conv3d gradients on the GPU with
float64 CPU autograd under the default settings and with torch.backends.cudnn.allow_tf32 = False."""
import torch, torch.nn.functional as F
torch.manual_seed(0)
dev = "cuda:0"
def probe(tag):
    for name, fn, xs, ws, kw in (("conv2d 32->32 3x3", F.conv2d, (3, 32, 16, 24), (32, 32, 3, 3), dict(padding=1)),
                                 ("conv2d 16->32 5x5 s2", F.conv2d, (3, 16, 32, 48), (32, 16, 5, 5), dict(padding=2, stride=2)),
                                 ("conv3d 8->8 3x3x3", F.conv3d, (1, 8, 8, 16, 24), (8, 8, 3, 3, 3), dict(padding=1))):
        x = torch.randn(xs); w = torch.randn(ws) * 0.1
        xr, wr = x.double().requires_grad_(True), w.double().requires_grad_(True)
        yr = fn(xr, wr, **kw); G = torch.randn(yr.shape)
        (yr * G.double()).sum().backward()
        xg, wg = x.to(dev).requires_grad_(True), w.to(dev).requires_grad_(True)
        yg = fn(xg, wg, **kw); (yg * G.to(dev)).sum().backward()
        rel = lambda a, b: float((a.cpu().double() - b).abs().max() / b.abs().max())
        print(f"[{tag}] {name:22s} fwd {rel(yg.detach(), yr.detach()):.1e}  dgrad {rel(xg.grad, xr.grad):.1e}  wgrad {rel(wg.grad, wr.grad):.1e}")
print("cudnn.allow_tf32 default:", torch.backends.cudnn.allow_tf32, " matmul.allow_tf32:", torch.backends.cuda.matmul.allow_tf32)
probe("default")
torch.backends.cudnn.allow_tf32 = False
torch.backends.cuda.matmul.allow_tf32 = False
probe("allow_tf32=False")
