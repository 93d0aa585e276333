"""Developer check (GPU box), round 6: discriminators for the two-streams-in-one-process hazard (profiles/r5_two_streams.txt).
24 config-2 scenes alternately on two HIP streams (one model replica each) against the one-stream outputs; for every scene that
differs: which tensor differs FIRST along the stage-1 chain (feature map -> activation bounds -> variance volume -> depth), and
how large the difference is (histogram of |delta| / range).  Run it three ways (tools/visits/r6_two_streams.sh): plain,
RCMVS_FP16_PAIR=0 (no activation bounds on the path), AMD_SERIALIZE_KERNEL=3, GPU_MAX_HW_QUEUES=1.
argv[1] = 'probe' (default) | 'spacer' (an unrelated tiny kernel between the stage-1 producer and K1)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
os.environ.setdefault("RCMVS_ALLOW_MULTI_STREAM", "1")      # (ops._stream() refuses a second stream otherwise: this script studies exactly that)
import torch
from rc_mvsnet_amd import _lib, ops, synthetic
from rc_mvsnet_amd.casmvsnet import CascadeMVSNet_eval
if os.environ.get("RCMVS_LIB"):                  # a variant library (tools/dev/build_plain_store_variant.sh)
    _lib.LIB_PATH = os.path.abspath(os.environ["RCMVS_LIB"])
_lib.load()
dev = "cuda:0"
mode = sys.argv[1] if len(sys.argv) > 1 else "probe"
NSC = int(os.environ.get("NSCENES", "24"))


def make():
    m = CascadeMVSNet_eval(ndepths=[48, 32, 8], depth_interals_ratio=[4, 2, 1])
    m.load_state_dict(synthetic.cascade_state_dict(0), strict=True)
    return m.to(dev).eval()


scenes = []
for seed in range(4):
    i, p, d = synthetic.cascade_inputs(1, 3, 512, 640, seed)
    scenes.append((i.to(dev), {k: v.to(dev) for k, v in p.items()}, d.to(dev)))
rng = float(scenes[0][2][0, -1] - scenes[0][2][0, 0])

# ---- taps on the stage-1 chain: copies taken AFTER the K1 launch on the launching stream (a copy in front of it would change the
# boundary under test); the spacer mode puts a 1-element kernel between the producer of the feature map and K1
_orig_wv = ops.warp_variance
_taps = None
_spacer = torch.zeros(64, device=dev)


def _wv(feats, rot, trans, planes, ndepth, variant=None, uniform_planes=False):
    if mode == "spacer" and uniform_planes:
        _spacer.add_(1.0)
    var = _orig_wv(feats, rot, trans, planes, ndepth, variant=variant, uniform_planes=uniform_planes)
    if _taps is not None and uniform_planes:
        _taps["feats"] = feats.clone()
        _taps["var"] = var.clone()
        _taps["planes"] = planes.clone()
        _taps["rot"] = rot.clone()
        # the same launch again, after the copies: reads what the first launch read, a few microseconds later
        _taps["var_again"] = _orig_wv(feats, rot, trans, planes, ndepth, variant=variant, uniform_planes=uniform_planes)
    elif _taps is not None and feats.shape[-1] == 8:          # stage 3: the same taps
        _taps["feats3"] = feats.clone()
        _taps["var3"] = var.clone()
        _taps["planes3"] = planes.clone()
    return var


_orig_head = ops.depth_head


def _head(x8, *a, **k):
    out = _orig_head(x8, *a, **k)
    if _taps is not None and x8.shape[1] == 8:
        _taps["x8_3"] = x8.clone()
    return out


ops.depth_head = _head


ops.warp_variance = _wv
import rc_mvsnet_amd.casmvsnet as cm
cm.ops.warp_variance = _wv


def run(model, scene):
    global _taps
    _taps = {}
    o = model(*scene)
    t = _taps
    _taps = None
    t["bounds"] = model._pair_bounds.clone() if getattr(model, "_pair_bounds", None) is not None else torch.zeros(1, device=dev)
    t["s1depth"] = o["stage1"]["depth"].clone()
    t["s2depth"] = o["stage2"]["depth"].clone()
    t["depth"] = o["depth"].clone()
    return t


ORDER = ["feats", "planes", "rot", "bounds", "var", "var_again", "s1depth", "s2depth", "planes3", "feats3", "var3", "x8_3", "depth"]
with torch.no_grad():
    one = make()
    run(one, scenes[0])                        # (first call: packs the weights, allocates the bounds)
    want = [run(one, s) for s in scenes]
    again = [run(one, s) for s in scenes]
    torch.cuda.synchronize()
    rep = sum(not torch.equal(a[k], w[k]) for a, w in zip(again, want) for k in ORDER)
    print(f"[{mode}] one stream, second pass over the 4 scenes: {rep} of {4 * len(ORDER)} tensors differ from the first pass")
    models = [make(), make()]
    streams = [torch.cuda.Stream(), torch.cuda.Stream()]
    got = []
    for i in range(2 + NSC):
        with torch.cuda.stream(streams[i % 2]):
            t = run(models[i % 2], scenes[i % 4])
        if i >= 2:
            got.append((i, t))
    torch.cuda.synchronize()
nbad = 0
for i, t in got:
    w = want[i % 4]
    diff = [k for k in ORDER if not torch.equal(t[k], w[k])]
    if not diff:
        continue
    nbad += 1
    line = f"  scene {i:2d} (stream {i % 2}): differing tensors {diff}"
    for k in diff:
        d = (t[k] - w[k]).abs().float()
        scale = rng if "depth" in k else float(w[k].abs().max()) + 1e-30
        rel = d / scale
        frac = float((d > 0).float().mean())
        hist = [float((rel > th).float().mean()) for th in (0, 1e-7, 1e-6, 1e-5, 1e-4, 1e-3, 1e-2)]
        line += f"\n      {k}: {100 * frac:.2f} % of elements, max |d|/scale {float(rel.max()):.3e}, frac above (0,1e-7..1e-2): " + " ".join(f"{h:.4f}" for h in hist)
        if k in ("var", "var3", "x8_3", "feats3"):                      # where: (plane, row, column) boxes of the differing elements, 4-plane x 4-row x 8-column K1 blocks touched
            idx = (t[k] != w[k]).nonzero()
            blocks = {(int(a[1]) // 4, int(a[2]) // 4, int(a[3]) // 8) for a in idx[:: max(1, len(idx) // 4000)]}
            line += f"\n      {k}: {len(idx)} elements in >= {len(blocks)} K1 blocks; planes {int(idx[:, 1].min())}-{int(idx[:, 1].max())}, rows {int(idx[:, 2].min())}-{int(idx[:, 2].max())}, " \
                    f"columns {int(idx[:, 3].min())}-{int(idx[:, 3].max())}; first blocks (plane chunk, tile row, tile column) {sorted(blocks)[:12]}"
        if k == "bounds":
            b0, b1 = t[k].amax(dim=-1), w[k].amax(dim=-1)
            line += f"\n      bound rows that differ (stage, row): {[(int(a), int(b)) for a, b in (b0 != b1).nonzero().tolist()]}"
    print(line)
print(f"[{mode}] env FP16_PAIR={os.environ.get('RCMVS_FP16_PAIR')} SERIALIZE={os.environ.get('AMD_SERIALIZE_KERNEL')} HWQ={os.environ.get('GPU_MAX_HW_QUEUES')}: "
      f"{nbad} of {len(got)} two-stream scenes differ from the one-stream outputs")
