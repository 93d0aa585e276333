"""Developer check (GPU box), round 5: the window-form K1 kernel alone on two HIP streams of one process.  Each stream, in a loop: write a
fresh feature block (an elementwise kernel on that stream; the allocator reuses blocks), run the window form and the gather kernel on it,
keep max |difference| on the GPU.  Both must agree to ~4e-7 of the value range whatever the other stream is doing."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
os.environ.setdefault("RCMVS_ALLOW_MULTI_STREAM", "1")      # (ops._stream() refuses a second stream otherwise: this script studies exactly that)
import torch
from rc_mvsnet_amd import _lib, ops, synthetic
_lib.load()
dev = "cuda:0"
C, D, h, w = 32, 48, 128, 160
g = torch.Generator().manual_seed(1)
base = [torch.randn(1, 3, h, w, C, generator=g).to(dev) for _ in range(2)]
rot, trans = ops.compose_homography(synthetic.proj_matrices(1, 3, 512, 640)["stage1"].to(dev))
planes = ops.hypothesis_planes(None, synthetic.depth_values(1).to(dev), (512, 640), 4, D, 4)
torch.cuda.synchronize()
variant = int(os.environ.get("K1_VARIANT", "5"))
for nstreams in (1, 2, 2):
    streams = [torch.cuda.Stream() for _ in range(nstreams)]
    errs = [[] for _ in range(nstreams)]
    for it in range(60):
        for s in range(nstreams):
            with torch.cuda.stream(streams[s]):
                x = base[s] * (1.0 + 0.01 * it)                       # fresh contents, reused allocator blocks
                a = ops.warp_variance(x, rot, trans, planes, D, variant=variant)
                b = ops.warp_variance(x, rot, trans, planes, D, variant=0)
                errs[s].append((a - b).abs().max() / b.abs().max())
    torch.cuda.synchronize()
    for s in range(nstreams):
        e = torch.stack(errs[s]).cpu()
        print(f"variant {variant}, {nstreams} stream(s), stream {s}: max rel difference over 60 rounds {float(e.max()):.3e}; rounds above 1e-5: {int((e > 1e-5).sum())}")
