"""Developer check (GPU box): eval_driver with one process against two worker processes on the one GPU, N rounds; every depth / confidence file must be
byte-identical (tests/test_gpu_wrappers.py::test_eval_driver_two_worker_processes_per_gpu_write_the_same_files, repeated, with the size of a difference printed)."""
import os, subprocess, sys, tempfile
import numpy as np
root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, root)
from rc_mvsnet_amd.sharding import clean_env
from rc_mvsnet_amd import data_io
common = ["--scans", "2", "--ref-views", "3", "--views", "3", "--height", "128", "--width", "160", "--ndepths", "16,8,8"]
N = int(sys.argv[1]) if len(sys.argv) > 1 else 10
bad = 0
with tempfile.TemporaryDirectory() as tmp:
    one = os.path.join(tmp, "one")
    out = subprocess.run([sys.executable, "-m", "rc_mvsnet_amd.eval_driver", "--outdir", one] + common, cwd=root, env=clean_env(), capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-2000:]
    names = sorted(os.path.relpath(os.path.join(d, f), one) for d, _, fs in os.walk(one) for f in fs)
    for it in range(N):
        two = os.path.join(tmp, f"two{it}")
        extra = ["--gpus", "1", "--procs-per-gpu", "2"] if it % 2 == 0 else []
        out = subprocess.run([sys.executable, "-m", "rc_mvsnet_amd.eval_driver", "--outdir", two] + common + extra, cwd=root, env=clean_env(), capture_output=True, text=True, timeout=900)
        if out.returncode != 0:
            print("round", it, "returncode", out.returncode, out.stderr[-1500:]); bad += 1; continue
        for n in names:
            a, b = open(os.path.join(one, n), "rb").read(), open(os.path.join(two, n), "rb").read()
            if a != b:
                bad += 1
                try:
                    x, y = data_io.read_pfm(os.path.join(one, n))[0], data_io.read_pfm(os.path.join(two, n))[0]
                    d = np.abs(x - y)
                    print("round", it, "procs", 2 if extra else 1, n, "differs: pixels", int((d > 0).sum()), "of", d.size, "max", float(d.max()))
                except Exception as e:
                    print("round", it, n, "differs", e)
print("rounds", N, "differences", bad)
