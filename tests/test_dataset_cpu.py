"""Evaluation loader without a GPU: the oracle against the reference's golden items, the
image-preparation arithmetic of csrc/image_prep_math.h (g++ loop harness) against the oracle, and the loader's host logic
(rc_mvsnet_amd/mvs_dataset.py) against the golden with the harness standing in for the kernel launch."""
import ctypes
import os
import subprocess

import numpy as np
import pytest
import torch

from oracle import dataset as O
from rc_mvsnet_amd import _lib, mvs_dataset, scan_io, synthetic

HERE = os.path.dirname(os.path.abspath(__file__))
GOLD = np.load(os.path.join(HERE, "golden", "dataset.npz"))
CASES = {"a": (["scan1"], 3, 1200, 1600), "b": (["scan1", "scan2"], 6, 64, 64)}


@pytest.fixture(scope="module")
def folder(tmp_path_factory):
    V, H, W, seed, n_src = [int(x) for x in GOLD["dims"]]
    scan = synthetic.fusion_scan(V=V, H=H, W=W, seed=seed, n_src=n_src)
    d = str(tmp_path_factory.mktemp("scans"))
    for name, line in (("scan1", "425.0 2.5"), ("scan2", "425.0 2.5 256 1065.0")):
        synthetic.write_fusion_scan(scan, os.path.join(d, name), os.path.join(d, name), depth_line=line)
    return d, scan


@pytest.fixture(scope="module")
def harness(tmp_path_factory):
    out = str(tmp_path_factory.mktemp("ip") / "ip_harness.so")
    subprocess.run(["g++", "-O2", "-w", "-ffp-contract=off", "-shared", "-fPIC", "-o", out,
                    os.path.join(HERE, "harness", "image_prep_harness.cpp")], check=True)
    return ctypes.CDLL(out)


def _p(a):
    return ctypes.c_void_p(a.ctypes.data)


def harness_prepare(h):
    def prepare(img_u8, out_hw, device, mean=mvs_dataset.MEAN, std=mvs_dataset.STD):
        H, W = img_u8.shape[:2]
        out = np.empty((3, int(out_hw[0]), int(out_hw[1])), np.float32)
        img_u8 = np.ascontiguousarray(img_u8)
        mean, std = np.array(mean, np.float32), np.array(std, np.float32)
        h.h_prepare_image(_p(img_u8), _p(out), H, W, out.shape[1], out.shape[2], _p(mean), _p(std))
        return torch.from_numpy(out)
    return prepare


def test_oracle_matches_reference_golden(folder):
    d, scan = folder
    for tag, (scans, nviews, max_h, max_w) in CASES.items():
        pairs = scan_io.read_pair_file(os.path.join(d, scans[0], "pair.txt"))
        ref, srcs = pairs[0]
        srcs = srcs + [srcs[0]] * max(0, nviews - len(srcs))
        views = [ref] + srcs[:nviews - 1]
        imgs, projs = [], []
        for v in views:
            K, E, dmin, dint = scan_io.read_cam_file(os.path.join(d, scans[0], "cams", "{:0>8}_cam.txt".format(v)), 1.06, 192)
            imgs.append(O.prepare_view(scan["img"][v], K, max_h, max_w))
            p = np.zeros((2, 4, 4), np.float32)
            p[0], p[1, :3, :3] = E, K
            projs.append(p)
            if v == ref:
                dv = O.depth_values(dmin, dint, 192)
        assert np.array_equal(np.stack(imgs), GOLD[tag + ":0:imgs"])
        for k, m in O.stage_matrices(np.stack(projs)).items():
            assert np.array_equal(m, GOLD["%s:0:%s" % (tag, k)]), (tag, k)
        assert np.array_equal(dv, GOLD[tag + ":0:depth_values"])


@pytest.mark.parametrize("shape,limits", [((75, 100), (1200, 1600)), ((75, 100), (64, 64)), ((64, 96), (64, 96)), ((130, 70), (100, 100)),
                                           ((1200, 1600), (1200, 1600))])
def test_image_preparation_arithmetic_matches_oracle(shape, limits, harness):
    g = np.random.default_rng(shape[0])
    img = (255 * g.random(shape + (3,))).astype(np.uint8)
    want = O.prepare_view(img, np.eye(3, dtype=np.float32), *limits)
    new_h, new_w = mvs_dataset.scaled_size(shape[0], shape[1], *limits)
    assert (int(new_h), int(new_w)) == want.shape[1:]
    got = harness_prepare(harness)(img, (new_h, new_w), None).numpy()
    assert np.array_equal(got, want)


def test_loader_items_match_reference(folder, harness, monkeypatch):
    d, _ = folder
    monkeypatch.setattr(mvs_dataset, "prepare_image", harness_prepare(harness))
    for tag, (scans, nviews, max_h, max_w) in CASES.items():
        ds = mvs_dataset.MVSDataset(d, scans, "test", nviews, 192, 1.06, device="cpu", max_h=max_h, max_w=max_w, fix_res=False)
        assert len(ds) == int(GOLD[tag + ":len"])
        for idx in (0, len(ds) - 1):
            item = ds[idx]
            if "%s:%d:imgs" % (tag, idx) in GOLD:
                assert np.array_equal(item["imgs"].numpy(), GOLD["%s:%d:imgs" % (tag, idx)])
            for k in ("stage1", "stage2", "stage3"):
                assert np.array_equal(item["proj_matrices"][k], GOLD["%s:%d:%s" % (tag, idx, k)]), (tag, idx, k)
            assert np.array_equal(item["depth_values"], GOLD["%s:%d:depth_values" % (tag, idx)])
            assert item["filename"] == str(GOLD["%s:%d:filename" % (tag, idx)])


def test_loader_fails_loudly_without_a_gpu(folder):
    d, _ = folder
    ds = mvs_dataset.MVSDataset(d, ["scan1"], "test", 3, 192, 1.06, device="cpu", max_h=1200, max_w=1600)
    with pytest.raises(_lib.RcmvsError):
        ds[0]
    with pytest.raises(_lib.RcmvsError):
        mvs_dataset.prepare_image(np.zeros((4, 4), np.uint8), (4, 4), "cpu")


def _check_tanks(ds):
    assert len(ds) == int(GOLD["t:len"])
    for idx in (0, len(ds) - 1):
        item = ds[idx]
        if "t:%d:imgs" % idx in GOLD:
            got = item["imgs"].cpu().numpy()
            assert np.allclose(got, GOLD["t:%d:imgs" % idx], rtol=0, atol=1e-6 if item["imgs"].is_cuda else 0)
        for k in ("stage1", "stage2", "stage3"):
            assert np.array_equal(item["proj_matrices"][k], GOLD["t:%d:%s" % (idx, k)]), (idx, k)
        assert np.array_equal(item["depth_values"], GOLD["t:%d:depth_values" % idx])
        assert item["filename"] == str(GOLD["t:%d:filename" % idx])


@pytest.fixture(scope="module")
def tanks_folder(tmp_path_factory):
    V, H, W, seed, n_src = [int(x) for x in GOLD["dims"]]
    scan = synthetic.fusion_scan(V=V, H=H, W=W, seed=seed, n_src=n_src)
    d = str(tmp_path_factory.mktemp("tt"))
    for name in mvs_dataset.TANKS_SCANS["intermediate"]:
        synthetic.write_tanks_scan(scan, os.path.join(d, "intermediate", name))
    return d


def test_tanks_loader_items_match_reference(tanks_folder, harness, monkeypatch):
    monkeypatch.setattr(mvs_dataset, "prepare_image", harness_prepare(harness))
    _check_tanks(mvs_dataset.TanksDataset(tanks_folder, "intermediate", 3, (96, 64), 192, device="cpu"))
    one = mvs_dataset.TanksDataset(tanks_folder, "intermediate", 3, (96, 64), 192, device="cpu", scans=["Horse"])
    assert len(one) == 5 and one[0]["filename"].startswith("Horse/")


def test_prefetch_keeps_order_and_propagates_errors(folder, harness, monkeypatch):
    """prefetch(): host halves run ahead on threads, items come back in order and equal to direct indexing; a failing item
    raises at its position."""
    d, _ = folder
    monkeypatch.setattr(mvs_dataset, "prepare_image", harness_prepare(harness))
    ds = mvs_dataset.MVSDataset(d, ["scan1", "scan2"], "test", 3, 192, 1.06, device="cpu", max_h=1200, max_w=1600)
    direct = [ds[i] for i in range(len(ds))]
    for workers, depth in ((1, 1), (4, 8), (3, 2)):
        got = list(mvs_dataset.prefetch(ds, workers=workers, depth=depth))
        assert [g["filename"] for g in got] == [x["filename"] for x in direct]
        assert all(torch.equal(g["imgs"], x["imgs"]) and np.array_equal(g["depth_values"], x["depth_values"]) for g, x in zip(got, direct))
    assert [g["filename"] for g in mvs_dataset.prefetch(ds, indices=[7, 2])] == [direct[7]["filename"], direct[2]["filename"]]
    orig = ds.load_host

    def flaky(i):
        if i == 3:
            raise ValueError("broken image")
        return orig(i)

    monkeypatch.setattr(ds, "load_host", flaky)
    seen = []
    with pytest.raises(ValueError, match="broken image"):
        for item in mvs_dataset.prefetch(ds, workers=4, depth=8):
            seen.append(item["filename"])
    assert seen == [x["filename"] for x in direct[:3]]


def test_async_writer_flushes_and_reports_failures(tmp_path):
    paths = [str(tmp_path / ("f%d.txt" % i)) for i in range(20)]
    with mvs_dataset.AsyncWriter(3) as w:
        for i, p in enumerate(paths):
            w.submit(lambda p=p, i=i: open(p, "w").write(str(i)))
    assert [open(p).read() for p in paths] == [str(i) for i in range(20)]        # everything is on disk once the block exits

    def boom():
        raise OSError("disk full")

    w = mvs_dataset.AsyncWriter(2)
    w.submit(boom)
    w.submit(lambda: open(paths[0], "w").write("late"))
    with pytest.raises(OSError, match="disk full"):
        w.close()
    assert open(paths[0]).read() == "late"                                        # the other jobs still ran
