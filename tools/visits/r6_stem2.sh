#!/bin/bash
# stem kernel occupancy variants: kernel time under rocprofv3 for each library (RCMVS_LIB honoured by bench.py? no: via layer-free direct timing below)
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -s -k "conv2d_stem" 2>&1 | grep "conv2d stem\|passed\|failed\|Error" | tee gpurun_out/r6_stem2.log
for lib in "$@"; do
RCMVS_LIB=$lib python - <<'PY' 2>&1 | grep -v Warning | tee -a gpurun_out/r6_stem2.log
import os, sys, torch
sys.path.insert(0, os.getcwd())
from rc_mvsnet_amd import _lib
if os.environ.get("RCMVS_LIB") and os.environ["RCMVS_LIB"] != "product":
    _lib.LIB_PATH = os.path.abspath(os.environ["RCMVS_LIB"])
from rc_mvsnet_amd import ops
g = torch.Generator().manual_seed(0)
x = torch.randn(3, 3, 512, 640, generator=g).cuda()
wa, wb = (torch.randn(8, 3, 3, 3, generator=g) / 5).cuda(), (torch.randn(8, 8, 3, 3, generator=g) / 8).cuda()
sa, sb, ha, hb = (torch.rand(8, generator=g).cuda() + 0.5 for _ in range(4))
pa, img = ops.pack_conv2d_weight(wa, pad_in_to=4), ops.pack_conv2d_stem(wb)
for _ in range(10): y = ops.conv2d_stem(x, pa, sa, ha, img, sb, hb)
torch.cuda.synchronize()
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
ts = []
for rep in range(5):
    a.record()
    for _ in range(50): y = ops.conv2d_stem(x, pa, sa, ha, img, sb, hb)
    b.record(); torch.cuda.synchronize()
    ts.append(a.elapsed_time(b) / 50 * 1e3)
print(os.environ.get("RCMVS_LIB"), "conv2d_stem 3x512x640: %.1f us (min of 5 x 50 back-to-back launches)" % min(ts), "checksum %.6e" % float(y.double().sum()))
PY
done
