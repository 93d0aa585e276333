import os
import sys

import numpy as np
import pytest
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)

GOLDEN = os.path.join(REPO, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_golden(name):
    z = np.load(os.path.join(GOLDEN, name + ".npz"))
    out = {}
    for k in z.files:
        a = z[k]
        out[k] = torch.from_numpy(a) if a.ndim > 0 and a.dtype.kind in "fi" and a.size > 8 else a
    return out


@pytest.fixture(scope="session")
def golden():
    return load_golden


def rel_err(a, b):
    """max |a-b| / max(1, max|b|)"""
    a = torch.as_tensor(a, dtype=torch.float64)
    b = torch.as_tensor(b, dtype=torch.float64)
    return float((a - b).abs().max() / max(1.0, float(b.abs().max())))


# ---------------------------------------------------------------------------------------------------------------------
# CPU emulation of the HIP kernels (tests/emu): the product's own Python path -- ops.*, the autograd Functions, the drop-in
# modules -- runs on CPU tensors against librcmvs_emu.so, so kernel logic is checked against the oracle without a GPU.
# ---------------------------------------------------------------------------------------------------------------------
def load_emu_lib():
    """ctypes handle of librcmvs_emu.so (built on demand under tests/emu/_build) with the product's signature table."""
    import ctypes
    sys.path.insert(0, os.path.join(REPO, "tests", "emu"))
    import build as emu_build
    from rc_mvsnet_amd import _lib
    lib = ctypes.CDLL(emu_build.build_cached())
    for name, argtypes in _lib.SIGNATURES.items():
        fn = getattr(lib, name)
        fn.argtypes = argtypes
        fn.restype = _lib._RESTYPES.get(name, ctypes.c_int)
    # RCMVS_EMU_ORDER=1|2 runs every block's threads in another order between synchronisation points: a whole-suite missing-barrier check
    lib.rcmvs_emu_set_order(int(os.environ.get("RCMVS_EMU_ORDER", "0")))
    return lib


def route_to_emulation(lib, setattr_):
    """Point the package at the emulated library: CPU tensors are accepted where the real path demands GPU ones, the 'stream'
    is NULL, and the modules take their HIP branches regardless of the device.  setattr_(obj, name, value) does the patching
    (monkeypatch.setattr in a test, plain setattr in a spawned worker process)."""
    import ctypes
    from rc_mvsnet_amd import _lib, casmvsnet, fusion, losses, mvs_dataset, ops, render_consist_net, train_ops

    def chk(t, name, dtype=torch.float32):
        if t.is_cuda or t.dtype != dtype or not t.is_contiguous():
            raise _lib.RcmvsError(f"{name}: emulation expects a contiguous CPU tensor of {dtype}")
        return ctypes.c_void_p(t.data_ptr())

    def opt(t, name):
        return ctypes.c_void_p(0) if t is None else chk(t, name)

    setattr_(_lib, "_lib", lib)
    for mod in (ops, fusion, losses, mvs_dataset, train_ops):
        setattr_(mod, "_chk", chk)
        setattr_(mod, "_stream", lambda: ctypes.c_void_p(0))
    setattr_(ops, "_opt", opt)
    setattr_(train_ops, "_opt", opt)
    infer = lambda module, *tensors: (not module.training) and (not torch.is_grad_enabled())            # noqa: E731
    train = lambda module, *tensors: module.training                                                      # noqa: E731
    for mod in (casmvsnet, render_consist_net):
        setattr_(mod, "_hip_inference", infer)
        setattr_(mod, "_hip_training", train)


@pytest.fixture(scope="session")
def emu_lib():
    return load_emu_lib()


@pytest.fixture
def emu(emu_lib, monkeypatch):
    """Route the package to the emulated library for one test (see route_to_emulation)."""
    route_to_emulation(emu_lib, monkeypatch.setattr)
    return emu_lib
