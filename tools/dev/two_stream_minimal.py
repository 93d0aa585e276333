"""Developer probe (GPU box) for item (0) of the next round: which concurrent work makes the hypothesis-planes kernel read stale pieces of
the previous stage's depth map?  (profiles/r3_two_streams.txt (i)-(o): with PLAIN loads in that kernel ~7 % of the scenes of the two-stream
pipeline were corrupted, first wrong op = the planes of stage 3.)

Run with the pre-fix library:  bash tools/dev/build_plain_planes_variant.sh   (build container), then on the box
    RCMVS_LIB=tools/dev/_variants/lib_plain.so python tools/dev/two_stream_minimal.py [aggressor ...]

VICTIM, stream A, per iteration -- the three launches around the spot that went wrong, on tensors allocated like the pipeline allocates them
(torch.empty per call, so the caching allocator hands the depth / confidence blocks back and forth):
    depth head of stage 2 (prob conv + softmax/regress: writes depth (1,256,320))  ->  [a filler launch]  ->  hypothesis planes of stage 3
and, still on the stream, a comparison of the planes with what the same depth map gives when nothing else runs.
AGGRESSOR, stream B, in a loop beside it (argument; default: all, one after the other):
    none | cascade (a whole scene of a second model replica) | stage3 (stage 3 only: K1 + cost regularisation + head on persistent inputs)
    | x3 (the 8 -> 8 full-resolution conv of stage 3) | k1 (warp + variance, stage 3) | head (depth head, stage 3) | fill (torch fills of 80 MiB)
Prints, per aggressor, how many of the victim's iterations produced a wrong planes tensor."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from rc_mvsnet_amd import _lib, ops, synthetic
from rc_mvsnet_amd.casmvsnet import CascadeMVSNet_eval
if os.environ.get("RCMVS_LIB"): _lib.LIB_PATH = os.path.abspath(os.environ["RCMVS_LIB"])
_lib.load()
dev = "cuda:0"
ITERS = int(os.environ.get("ITERS", "400"))
H, W = 512, 640


def make():
    m = CascadeMVSNet_eval(ndepths=[48, 32, 8], depth_interals_ratio=[4, 2, 1])
    m.load_state_dict(synthetic.cascade_state_dict(0)); return m.to(dev).eval()


import warnings; warnings.simplefilter("ignore")
with torch.no_grad():
    model = make()
    imgs, pm, dv = synthetic.cascade_inputs(1, 3, H, W, 0)
    scene = (imgs.to(dev), {k: v.to(dev) for k, v in pm.items()}, dv.to(dev))
    dvd = scene[2]
    cr2, cr3 = model._cr(1), model._cr(2)
    g = torch.Generator().manual_seed(1)
    # the victim's persistent inputs: two different stage-2 regularised volumes (so consecutive depth maps differ) and stage-2 planes
    x8 = [(torch.randn(1, 32, H // 2, W // 2, 8, generator=g) * 0.5).to(dev) for _ in range(2)]
    planes2 = ops.hypothesis_planes(None, dvd, (H, W), 2, 32, 2.0)
    wprob2 = cr2.hip_plan()["prob"]
    filler_in = torch.randn(1, 1, H, W, 8, generator=g).to(dev)
    wfill = cr3.hip_plan()["conv0"]
    # references, computed alone
    want = []
    for v in x8:
        d, _ = ops.depth_head(v, wprob2, planes2)
        want.append(ops.hypothesis_planes(d, dvd, (H, W), 1, 8, 1.0).clone())
    torch.cuda.synchronize()
    # the aggressors' persistent inputs
    rot, trans = ops.compose_homography(scene[1]["stage3"].contiguous().float())
    f3 = torch.randn(1, 3, H, W, 8, generator=g).to(dev)
    planes3 = want[0]
    var3 = ops.warp_variance(f3, rot, trans, planes3, 8)
    x8_3 = cr3.features_cl(var3)
    big = torch.empty(20 * 1024 * 1024, device=dev)
    replica = make()
    replica(*scene)
    torch.cuda.synchronize()

    def aggress(kind):
        if kind == "cascade": replica(*scene)
        elif kind == "stage3":
            v = ops.warp_variance(f3, rot, trans, planes3, 8); ops.depth_head(cr3.features_cl(v), cr3.hip_plan()["prob"], planes3)
        elif kind == "x3": ops.conv3d(var3, wfill[0], wfill[1], wfill[2], relu=True)
        elif kind == "k1": ops.warp_variance(f3, rot, trans, planes3, 8)
        elif kind == "head": ops.depth_head(x8_3, cr3.hip_plan()["prob"], planes3)
        elif kind == "fill": big.fill_(1.0)

    sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
    kinds = sys.argv[1:] or ["none", "fill", "k1", "x3", "head", "stage3", "cascade", "none"]
    for kind in kinds:
        bad = torch.zeros(ITERS, device=dev, dtype=torch.int32)
        keep = []
        t0 = time.perf_counter()
        for i in range(ITERS):
            with torch.cuda.stream(sa):
                d, c = ops.depth_head(x8[i % 2], wprob2, planes2)
                ops.conv3d(filler_in, wfill[0], wfill[1], wfill[2], relu=True)           # the launch between the two in the pipeline (there: the pyramid's output conv)
                p = ops.hypothesis_planes(d, dvd, (H, W), 1, 8, 1.0)
                bad[i] = (p != want[i % 2]).any()
                keep = [d, c, p][: 1 + i % 3]                # outputs die at different times, like a scene's: the small blocks change roles
            if kind != "none":
                with torch.cuda.stream(sb):
                    aggress(kind)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        print(f"aggressor {kind:8s}: {int(bad.sum())} of {ITERS} victim iterations with wrong planes; {dt * 1e3 / ITERS:.3f} ms per iteration", flush=True)
