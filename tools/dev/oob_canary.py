"""Developer check (GPU box): does any kernel of a scene WRITE outside its output tensor?
Every tensor the ops layer allocates (torch.empty / torch.zeros on the GPU) is placed in the middle of a buffer with PAD bytes of guard band
on both sides, the guard bands hold a sentinel, every tensor is kept alive for the scene (no block is handed out twice), and after the scene
the guard bands are compared with the sentinel.  One stream, deterministic; STREAMS=2 runs the scenes through the two-stream pipeline."""
import os, sys, math
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from rc_mvsnet_amd import _lib, synthetic
from rc_mvsnet_amd.casmvsnet import CascadeMVSNet_eval
from rc_mvsnet_amd.scene_pipeline import ScenePipeline
_lib.load()
dev = "cuda:0"
PAD = int(os.environ.get("PAD", str(24 << 20)))
SMALLPAD = int(os.environ.get("SMALLPAD", str(1 << 20)))          # tensors under 1 MB
NST = int(os.environ.get("STREAMS", "1"))
SENT = 0x7fc0beef - (1 << 32) if 0x7fc0beef >= (1 << 31) else 0x7fc0beef
real_empty, real_zeros = torch.empty, torch.zeros
LIVE = []


def guarded(fn, zero):
    def alloc(*a, **k):
        if "device" not in k or not str(k["device"]).startswith("cuda"):
            return fn(*a, **k)
        shape = tuple(a[0]) if len(a) == 1 and isinstance(a[0], (tuple, list, torch.Size)) else tuple(a)
        dt = k.get("dtype") or torch.float32
        es = real_empty((), dtype=dt).element_size()
        n = math.prod(shape) if len(shape) else 1
        pad = PAD if n * es >= (1 << 20) else SMALLPAD
        pe = pad // es
        kk = dict(k); kk.pop("dtype", None)
        raw = real_empty(((n + 2 * pe) * es + 3) // 4, dtype=torch.int32, **kk)
        raw.fill_(SENT)
        flat = raw.view(torch.uint8)[: (n + 2 * pe) * es].view(dt)
        t = flat[pe:pe + n].view(shape)
        if zero: t.zero_()
        f = sys._getframe(1)
        LIVE.append((raw, pe * es // 4, (pe + n) * es // 4, "{}:{} {} {}".format(f.f_code.co_name, f.f_lineno, shape, str(dt).replace("torch.", ""))))
        return t
    return alloc


def check(label):
    torch.cuda.synchronize()
    hits = 0
    for idx, (raw, lo, hi, tag) in enumerate(LIVE):
        for name, g, base in (("before", raw[:lo], 0), ("after", raw[hi:], hi)):
            badm = g != SENT
            if bool(badm.any()):
                w = badm.nonzero().flatten()
                hits += 1
                dist = (lo - 1 - int(w[-1]), lo - int(w[0])) if name == "before" else (int(w[0]), int(w[-1]) + 1)
                print(f"  {label}: tensor #{idx} [{tag}]: {w.numel()} guard words {name} it overwritten, {dist[0] * 4}..{dist[1] * 4} bytes {'ahead of its start' if name == 'before' else 'past its end'}; "
                      f"values e.g. {g[w[:4]].view(torch.float32).tolist()}")
    print(f"{label}: {len(LIVE)} tensors, {hits} damaged guard bands")
    LIVE.clear()
    return hits


def make():
    m = CascadeMVSNet_eval(ndepths=[48, 32, 8], depth_interals_ratio=[4, 2, 1])
    m.load_state_dict(synthetic.cascade_state_dict(0)); return m.to(dev).eval()


import warnings; warnings.simplefilter("ignore")
scenes = []
for seed in range(2):
    imgs, pm, dv = synthetic.cascade_inputs(1, 3, 512, 640, seed)
    scenes.append((imgs.to(dev), {k: v.to(dev) for k, v in pm.items()}, dv.to(dev)))
torch.empty, torch.zeros = guarded(real_empty, False), guarded(real_zeros, True)
with torch.no_grad():
    if NST == 1:
        m = make()
        for i in range(3):
            out = m(*scenes[i % 2])
            check(f"one stream, scene {i}" + (" (plans built here)" if i == 0 else ""))
        for pair in ("0", "1"):
            os.environ["RCMVS_FP16_PAIR"] = pair
            out = m(*scenes[0]); check(f"one stream, RCMVS_FP16_PAIR={pair}")
    else:
        pipe = ScenePipeline(make, NST, dev)
        for rnd in range(3):
            outs = [pipe(*scenes[i % 2])[0] for i in range(4)]
            pipe.synchronize()
            check(f"{NST} streams, round {rnd} (4 scenes)")
