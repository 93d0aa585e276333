"""Developer tool (no GPU needed): host-side cost of one config-2 scene -- the library replaced by a stub whose entry points return 0
immediately, CPU tensors, so what is timed is the Python of the forward (plan validation, allocations, ctypes argument marshalling).
Round 3: 1.38 ms per scene before the plan keys were read from the module dictionaries, 0.79 ms after (this container's CPU)."""
import os, sys, time, ctypes, cProfile, pstats, io
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from rc_mvsnet_amd import _lib, ops, synthetic, casmvsnet
from rc_mvsnet_amd.casmvsnet import CascadeMVSNet_eval

class Stub:
    def __getattr__(self, name):
        f = lambda *a: 0
        return f
stub = Stub()
_lib._lib = stub
_lib.load = lambda: stub
chk = lambda t, name, dtype=torch.float32: ctypes.c_void_p(t.data_ptr())
opt = lambda t, name: ctypes.c_void_p(0) if t is None else chk(t, name)
ops._chk = chk; ops._opt = opt; ops._stream = lambda: ctypes.c_void_p(0)
casmvsnet._hip_inference = lambda m, *t: True
m = CascadeMVSNet_eval(ndepths=[48, 32, 8], depth_interals_ratio=[4, 2, 1])
m.load_state_dict(synthetic.cascade_state_dict(0)); m.eval()
H, W = 64, 96          # sizes do not matter to the host cost (allocations are empty())
imgs, pm, dv = synthetic.cascade_inputs(1, 3, H, W, 0)
with torch.no_grad():
    for _ in range(3): m(imgs, pm, dv)
    t0 = time.perf_counter()
    N = 200
    for _ in range(N): m(imgs, pm, dv)
    dt = (time.perf_counter() - t0) / N
    print(f"host time per scene with a null library: {dt * 1e3:.3f} ms")
    pr = cProfile.Profile(); pr.enable()
    for _ in range(100): m(imgs, pm, dv)
    pr.disable()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(22); print(s.getvalue()[:4500])
