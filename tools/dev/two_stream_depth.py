"""Developer check (GPU box): is the two-stream corruption a matter of memory REUSE or of how far the host runs AHEAD of the GPU?
  SCENES   scenes per round (12 in the earlier probes; the delayed-free result could be the hipMallocs of the warm-up scenes acting as syncs)
  KEEPALIVE=1 hold every tensor the ops layer allocates for the round; =2 with DELAY=n: release a scene's tensors n scenes later
  MINBYTES / MAXBYTES   hold only tensors in that size range (which tensors' reuse matters)
  AHEAD=n  before issuing scene i the host waits for scene i-n to finish (both streams stay busy, the queues stay shallow)
  PAD=bytes   every tensor of MINBYTES or more gets PAD bytes of guard band on both sides (an out-of-bounds write next to a volume lands there)
  FENCE=1     every library launch is preceded by an explicit dependency on everything issued before it on its stream, routed through a
              helper stream (record on the stream -> helper waits -> record on the helper -> the stream waits): in-stream order no longer rests
              on the queue's own packet ordering; FENCE=2 = the same host work without the final wait (control for the host overhead)
  LOGADDR=1   record (stream, address range) of every such tensor and report overlaps between the two streams' ranges
Prints the indices of the corrupted scenes per round and the wall time per scene."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
os.environ.setdefault("RCMVS_ALLOW_MULTI_STREAM", "1")      # (ops._stream() refuses a second stream otherwise: this script studies exactly that)
import torch
from rc_mvsnet_amd import _lib, synthetic
from rc_mvsnet_amd.casmvsnet import CascadeMVSNet_eval
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from scene_pipeline import ScenePipeline
if os.environ.get('RCMVS_LIB'): _lib.LIB_PATH = os.path.abspath(os.environ['RCMVS_LIB'])       # a library variant (A/B of a kernel change)
_lib.load()
dev = "cuda:0"
E = lambda k, d: int(os.environ.get(k, d))
NR, NS, DELAY, KA, AHEAD = E("ROUNDS", "8"), E("SCENES", "12"), E("DELAY", "0"), E("KEEPALIVE", "0"), E("AHEAD", "0")
LO, HI = E("MINBYTES", "0"), E("MAXBYTES", str(1 << 62))
KEEP = []
real_empty, real_zeros = torch.empty, torch.zeros
PAD, LOGADDR = E("PAD", "0"), E("LOGADDR", "0")
RANGES = {}
if PAD or LOGADDR:
    import math

    def padded(fn):
        def alloc(*a, **k):
            shape = a[0] if len(a) == 1 and isinstance(a[0], (tuple, list, torch.Size)) else a
            dt = k.get("dtype") or torch.float32
            n = math.prod(shape) if len(shape) else 1
            nb = n * real_empty((), dtype=dt).element_size()
            if nb < LO or "device" not in k or not str(k["device"]).startswith("cuda"):
                return fn(*a, **k)
            if PAD:
                pe = PAD // 4
                flat = fn(n + 2 * pe, **k)
                t = flat[pe:pe + n].view(*shape)
            else:
                t = fn(*a, **k)
            if LOGADDR:
                RANGES.setdefault(torch.cuda.current_stream().cuda_stream, set()).add((t.data_ptr(), t.data_ptr() + nb))
            return t
        return alloc
    torch.empty, torch.zeros = padded(real_empty), padded(real_zeros)
elif KA:
    def hold(t):
        if LO <= t.numel() * t.element_size() < HI: KEEP.append(t)
        return t
    torch.empty = lambda *a, **k: hold(real_empty(*a, **k))
    torch.zeros = lambda *a, **k: hold(real_zeros(*a, **k))
FENCE = E("FENCE", "0")
if FENCE:
    from rc_mvsnet_amd import ops as _ops
    import ctypes
    HELPERS = {}

    def fenced_stream():
        st = torch.cuda.current_stream()
        if st.cuda_stream:
            h = HELPERS.get(st.cuda_stream)
            if h is None: h = HELPERS[st.cuda_stream] = torch.cuda.Stream()
            e1, e2 = torch.cuda.Event(), torch.cuda.Event()
            e1.record(st); h.wait_event(e1); e2.record(h)
            if FENCE == 1: st.wait_event(e2)
        return ctypes.c_void_p(st.cuda_stream)
    _ops._stream = fenced_stream
scenes = []
for seed in range(4):
    imgs, pm, dv = synthetic.cascade_inputs(1, 3, 512, 640, seed)
    scenes.append((imgs.to(dev), {k: v.to(dev) for k, v in pm.items()}, dv.to(dev)))


def make():
    m = CascadeMVSNet_eval(ndepths=[48, 32, 8], depth_interals_ratio=[4, 2, 1])
    m.load_state_dict(synthetic.cascade_state_dict(0)); return m.to(dev).eval()


import warnings; warnings.simplefilter("ignore")
with torch.no_grad():
    ref = make()
    want = [ref(*s)["depth"].clone() for s in scenes]
    torch.cuda.synchronize(); KEEP.clear()
    bad, where, ms = 0, [], []
    for rnd in range(NR):
        pipe = ScenePipeline(make, 2, dev, experimental=True)
        for i in range(2): pipe(*scenes[i])            # plans / packed weights of both replicas outside the timed, checked part
        pipe.synchronize(); KEEP.clear()
        got, hist, done = [], [], []
        t0 = time.perf_counter()
        for i in range(NS):
            if AHEAD and i >= AHEAD: done[i - AHEAD].synchronize()
            out, st = pipe(*scenes[i % 4])
            got.append(out["depth"])
            ev = torch.cuda.Event(); ev.record(st); done.append(ev)
            if DELAY:
                hist.append(list(KEEP)); KEEP.clear()
                if len(hist) > DELAY: hist.pop(0)
        pipe.synchronize()
        ms.append((time.perf_counter() - t0) * 1e3 / NS)
        w = [i for i, o in enumerate(got) if not torch.equal(o, want[i % 4])]
        bad += bool(w); where.append(w)
        if LOGADDR:
            for st in pipe.streams:
                r = sorted(RANGES.get(st.cuda_stream, ()))
                G = 1 << 32
                print(f"  round {rnd}: stream {st.cuda_stream:#x}: {len(r)} volume ranges in {r[0][0]:#x}..{r[-1][1]:#x}; crossing a 4 GiB line: {[hex(l) for l, h in r if l // G != (h - 1) // G]}; "
                      f"{ms[-1]:.3f} ms/scene; corrupted {[i for i in w if pipe.streams[i % 2] is st]}")
                for l, h in r: print(f"      {l:#x} +{(h - l) / 2 ** 20:.1f} MiB")
        n = len(KEEP) + sum(len(h) for h in hist); KEEP.clear(); hist.clear()
if LOGADDR:
    ks = [k for k in RANGES if k != 0]
    print("streams with volumes:", {k: len(v) for k, v in RANGES.items()})
    for a in range(len(ks)):
        for b in range(a + 1, len(ks)):
            ov = sum(1 for (l0, h0) in RANGES[ks[a]] for (l1, h1) in RANGES[ks[b]] if l0 < h1 and l1 < h0)
            gap = min((abs(l1 - h0) if l1 >= h0 else abs(l0 - h1)) for (l0, h0) in RANGES[ks[a]] for (l1, h1) in RANGES[ks[b]] if not (l0 < h1 and l1 < h0))
            print(f"  streams {ks[a]:#x} / {ks[b]:#x}: {ov} overlapping range pairs, smallest gap between two ranges {gap} bytes")
cfg = " ".join(f"{k}={os.environ[k]}" for k in ("SCENES", "KEEPALIVE", "DELAY", "MINBYTES", "MAXBYTES", "AHEAD", "PAD", "LOGADDR", "FENCE", "RCMVS_LIB") if k in os.environ)
print(f"[{cfg}] {bad} of {NR} rounds corrupted; {sorted(ms)[len(ms) // 2]:.3f} ms/scene (median round); held {n}; corrupted scene indices per round: {where}")
