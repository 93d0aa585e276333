"""Self-test of the CPU emulation of the HIP programming model (tests/emu): the primitives the product kernels rely on behave as
the hardware's do, and the schedule-permutation check exposes a kernel with a missing barrier."""
import ctypes
import os
import subprocess

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
EMU = os.path.join(HERE, "emu")
CLANG = os.environ.get("EMU_CXX", "/opt/rocm/lib/llvm/bin/clang++")


@pytest.fixture(scope="module")
def st(tmp_path_factory):
    out = str(tmp_path_factory.mktemp("emu_st") / "selftest.so")
    subprocess.run([CLANG, "-std=c++17", "-O1", "-w", "-fPIC", "-shared", "-I", EMU, "-o", out,
                    os.path.join(EMU, "selftest.cpp"), os.path.join(EMU, "engine.cpp")], check=True)
    return ctypes.CDLL(out)


def _p(a):
    return ctypes.c_void_p(a.ctypes.data)


def test_block_reduction_and_missing_barrier_detection(st):
    x = np.random.default_rng(0).random((5, 256)).astype(np.float32)
    want = x.reshape(5, 4, 64).astype(np.float64)
    results = {}
    for racy in (0, 1):
        for order in (0, 1, 2):
            st.rcmvs_emu_set_order(order)
            out = np.zeros(5, np.float32)
            st.st_block_sum(_p(x), _p(out), 5, racy)
            results[(racy, order)] = out.copy()
    st.rcmvs_emu_set_order(0)
    for order in (0, 1, 2):
        assert np.allclose(results[(0, order)], want.sum((1, 2)), rtol=1e-5)
        assert np.array_equal(results[(0, order)], results[(0, 0)])                       # with the barrier: schedule-independent
    assert not np.array_equal(results[(1, 1)], results[(1, 0)])                           # without it: the permutation shows


def test_wave_primitives(st):
    out = np.zeros((8, 64), np.int32)
    st.st_wave_ops(_p(out))
    l = np.arange(64)
    assert np.array_equal(out[0], l ^ 5)
    assert np.array_equal(out[1], np.where(l >= 3, l - 3, l))
    assert np.array_equal(out[2], np.where(l % 16 + 7 < 16, l + 7, l))                    # width 16: stays inside its group
    assert np.array_equal(out[3], np.array([sum(1 for k in range(i) if k % 3 == 0) for i in l]))
    assert np.array_equal(out[4], np.full(64, 170))
    assert np.array_equal(out[5], np.where(l % 16 >= 2, l - 2, -1))                       # row_shr:2, bound_ctrl off keeps `old`
    assert np.array_equal(out[6], np.full(64, 20))
    assert np.array_equal(out[7], np.full(64, 100))


@pytest.mark.parametrize("n", [64, 37, 1])
def test_exited_lanes_do_not_take_part(st, n):
    out = np.full(192, -5, np.int32)
    st.st_partial_wave(_p(out), n)
    l = np.arange(n)
    assert np.array_equal(out[:n], np.full(n, n))                                         # ballot counts live lanes only
    assert np.array_equal(out[64:64 + n], np.where(l + 1 < n, l + 1, l))                  # reading an exited lane returns own value
    assert np.array_equal(out[128:128 + n], np.full(n, 7)) and np.all(out[n:64] == -5)


def test_mfma_16x16x4_layout(st):
    g = np.random.default_rng(1)
    A, B, C = (g.standard_normal(s).astype(np.float32) for s in ((16, 4), (4, 16), (16, 16)))
    D = np.zeros((16, 16), np.float32)
    st.st_mfma(_p(A), _p(B), _p(C), _p(D))
    assert np.allclose(D, A @ B + C, rtol=1e-6, atol=1e-6)


def test_buffer_loads_dynamic_lds_and_2d_geometry(st):
    src = np.arange(100, dtype=np.float32)                    # 400 bytes: lanes 0..24 are in range (16 bytes each)
    out = np.zeros((6, 64), np.float32)
    st.st_buffer_and_lds(_p(src), 100, _p(out))
    t = 63 - np.arange(64)
    want = np.where(t * 16 + 16 <= 400, t * 4.0, 0.0).astype(np.float32)
    for b in range(6):
        assert np.array_equal(out[b], want)


def test_more_lane_primitives(st):
    out, fout = np.zeros((10, 64), np.int32), np.zeros(64, np.float32)
    st.st_wave_ops2(_p(out), _p(fout))
    l = np.arange(64)
    i, row = l % 16, l // 16 * 16
    assert np.array_equal(out[0], np.where(i + 3 <= 15, l + 3, -1))
    assert np.array_equal(out[1], row + (i - 4) % 16)
    assert np.array_equal(out[2], row + 15 - i)
    assert np.array_equal(out[3], np.where((l // 16) % 2 == 1, row - 1, -1))              # rows 1 and 3 receive lane 15 of the row before
    assert np.array_equal(out[4], np.where(l >= 32, 31, -1))
    assert np.array_equal(out[5], (l & ~3) + 3 - (l & 3))
    assert np.array_equal(out[6], np.where(i >= 1, l - 1, 0))
    assert np.array_equal(out[7], ((l * 7) % 64) * 2)
    assert np.array_equal(out[8], row + 5)
    assert np.array_equal(out[9], np.full(64, 2))
    assert np.array_equal(fout, np.full(64, 4.5, np.float32))


def test_direct_to_lds_loads(st):
    src = np.arange(400, dtype=np.float32)                    # 1600 bytes: threads 0..99 in range, 100..127 read zeros
    out = np.full(512, -1.0, np.float32)
    st.st_dma(_p(src), 400, _p(out))
    want = np.where(np.arange(512) < 400, np.arange(512, dtype=np.float32), 0.0)
    assert np.array_equal(out, want)


def test_mfma_32x32x2_layout(st):
    g = np.random.default_rng(2)
    A, B, C = (g.standard_normal(s).astype(np.float32) for s in ((32, 2), (2, 32), (32, 32)))
    D = np.zeros((32, 32), np.float32)
    st.st_mfma32(_p(A), _p(B), _p(C), _p(D))
    assert np.allclose(D, A @ B + C, rtol=1e-6, atol=1e-6)
