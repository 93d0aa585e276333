"""Depth-map fusion filter and the scan file formats (rank 4) without a GPU: the oracle
against the reference's golden outputs, the shared per-pixel arithmetic of csrc/fusion_math.h (compiled with g++ into a loop
harness) against the oracle, and the host-side parsers / writers."""
import ctypes
import os
import subprocess

import numpy as np
import pytest
import torch

from oracle import fusion as O
from rc_mvsnet_amd import _lib, fusion, scan_io, synthetic

HERE = os.path.dirname(os.path.abspath(__file__))
GOLD = np.load(os.path.join(HERE, "golden", "fusion.npz"))
PROB, NCONS, DIST, DEPTH = 0.8, 3, 0.5, 0.01


def scan():
    V, H, W, seed, n_src = [int(x) for x in GOLD["dims"]]
    return synthetic.fusion_scan(V=V, H=H, W=W, seed=seed, n_src=n_src)


def test_oracle_matches_reference_golden():
    s = scan()
    m, back, xs, ys = O.check_geometric_consistency(s["depth"][0], s["K"][0], s["E"][0], s["depth"][2], s["K"][2], s["E"][2], DIST, DEPTH)
    assert np.array_equal(m, GOLD["pair02:mask"])
    assert np.array_equal(back, GOLD["pair02:depth"])
    assert np.array_equal(xs, GOLD["pair02:x_src"]) and np.array_equal(ys, GOLD["pair02:y_src"])
    pts, cols = [], []
    for ref, srcs in s["pairs"]:
        r = O.fuse_view(s["depth"][ref], s["conf"][ref], s["img"][ref].astype(np.float32) / 255.0, s["K"][ref], s["E"][ref],
                        [s["depth"][i] for i in srcs], [s["K"][i] for i in srcs], [s["E"][i] for i in srcs], PROB, NCONS, DIST, DEPTH)
        for kind in ("photo", "geo", "final"):
            assert np.array_equal(r[kind], GOLD["mask:%d:%s" % (ref, kind)]), (ref, kind)
        pts.append(r["xyz"])
        cols.append(r["rgb"])
    assert np.array_equal(np.concatenate(pts), GOLD["xyz"])
    assert np.array_equal(np.concatenate(cols), GOLD["rgb"])


def test_remap_linear_properties():
    """The restated cv2.remap: integer positions return the pixel, positions are quantised to 1/32 pixel, taps outside the
    image contribute 0, NaN positions give 0."""
    g = np.random.default_rng(0)
    img = g.random((6, 7)).astype(np.float32)
    yy, xx = np.meshgrid(np.arange(6, dtype=np.float32), np.arange(7, dtype=np.float32), indexing="ij")
    assert np.array_equal(O.remap_linear(img, xx, yy), img)
    half = O.remap_linear(img, xx[:, :-1] + 0.5, yy[:, :-1])
    assert np.allclose(half, 0.5 * (img[:, :-1] + img[:, 1:]), atol=1e-7)
    assert np.array_equal(O.remap_linear(img, xx + 0.01, yy), img)                          # 0.01 px rounds to 0/32
    edge = O.remap_linear(img, np.full((1, 1), -0.5, np.float32), np.zeros((1, 1), np.float32))
    assert np.allclose(edge, 0.5 * img[0, 0])
    bad = O.remap_linear(img, np.array([[np.nan, 1e30, -3.0]], np.float32), np.zeros((1, 3), np.float32))
    assert np.array_equal(bad, np.zeros((1, 3), np.float32))


@pytest.fixture(scope="module")
def harness(tmp_path_factory):
    out = str(tmp_path_factory.mktemp("fu") / "fu_harness.so")
    subprocess.run(["g++", "-O2", "-w", "-ffp-contract=off", "-shared", "-fPIC", "-o", out,
                    os.path.join(HERE, "harness", "fusion_harness.cpp")], check=True)
    return ctypes.CDLL(out)


def _p(a):
    return ctypes.c_void_p(a.ctypes.data) if a is not None else ctypes.c_void_p(0)


def test_kernel_arithmetic_matches_oracle(harness):
    s = scan()
    V, H, W = s["depth"].shape
    depth_all = np.ascontiguousarray(s["depth"])
    total_flips = 0
    for ref, srcs in s["pairs"]:
        N = len(srcs)
        mats = fusion.fusion_matrices(s["K"][ref], s["E"][ref], [s["K"][i] for i in srcs], [s["E"][i] for i in srcs])
        img = np.ascontiguousarray(s["img"][ref].astype(np.float32) / 255.0)
        masks, avg = np.empty((3, H, W), np.uint8), np.empty((H, W), np.float32)
        xyz, rgb = np.empty((H, W, 3), np.float32), np.empty((H, W, 3), np.uint8)
        dd, dg, dxy = np.empty((N, H, W), np.float32), np.empty((N, H, W), np.uint8), np.empty((N, H, W, 2), np.float32)
        idx = np.array(srcs, np.int32)
        harness.h_fuse_view(_p(depth_all), ref, _p(idx), _p(np.ascontiguousarray(s["conf"][ref])), _p(img), _p(mats),
                            ctypes.c_float(PROB), NCONS, ctypes.c_double(DIST), ctypes.c_float(DEPTH),
                            _p(masks), _p(avg), _p(xyz), _p(rgb), _p(dd), _p(dg), _p(dxy), N, H, W)
        r = O.fuse_view(s["depth"][ref], s["conf"][ref], img, s["K"][ref], s["E"][ref], [s["depth"][i] for i in srcs],
                        [s["K"][i] for i in srcs], [s["E"][i] for i in srcs], PROB, NCONS, DIST, DEPTH)
        assert np.array_equal(masks[0].astype(bool), r["photo"])
        flips = int((masks[1].astype(bool) != r["geo"]).sum())
        total_flips += flips
        same = masks[2].astype(bool) == r["final"]
        assert np.allclose(avg[same], r["depth_avg"].astype(np.float32)[same], rtol=1e-6)
        both = masks[2].astype(bool) & r["final"]
        want = np.zeros((H, W, 3), np.float32)
        want[r["final"]] = r["xyz"]
        assert np.allclose(xyz[both], want[both], rtol=1e-5, atol=1e-3)
        wrgb = np.zeros((H, W, 3), np.uint8)
        wrgb[r["final"]] = r["rgb"]
        assert np.array_equal(rgb[both], wrgb[both])
        for n, src in enumerate(srcs):
            m, back, xs, ys = O.check_geometric_consistency(s["depth"][ref], s["K"][ref], s["E"][ref], s["depth"][src], s["K"][src],
                                                            s["E"][src], DIST, DEPTH)
            assert np.allclose(dxy[n, ..., 0], xs, rtol=1e-6, atol=1e-4) and np.allclose(dxy[n, ..., 1], ys, rtol=1e-6, atol=1e-4)
            agree = dg[n].astype(bool) == m
            assert agree.mean() > 0.995
            assert np.allclose(dd[n][agree], back[agree], rtol=1e-6, atol=1e-3)
    assert total_flips <= 3                                             # threshold knife edges only


def test_scan_files_round_trip(tmp_path):
    s = scan()
    pair_folder, out_folder = str(tmp_path / "data" / "scan1"), str(tmp_path / "out" / "scan1")
    synthetic.write_fusion_scan(s, pair_folder, out_folder)
    assert scan_io.read_pair_file(os.path.join(pair_folder, "pair.txt")) == s["pairs"]
    K, E = scan_io.read_camera_parameters(os.path.join(out_folder, "cams", "{:0>8}_cam.txt".format(3)))
    assert K.dtype == np.float32 and E.dtype == np.float32
    assert np.array_equal(K, s["K"][3]) and np.array_equal(E, s["E"][3])
    Kq, E2, dmin, dint = scan_io.read_cam_file(os.path.join(out_folder, "cams", "{:0>8}_cam.txt".format(3)), interval_scale=1.06)
    assert np.allclose(Kq[:2] * 4.0, s["K"][3][:2]) and np.array_equal(Kq[2], s["K"][3][2]) and np.array_equal(E2, E)
    assert dmin == 425.0 and abs(dint - 2.5 * 1.06) < 1e-12
    img = scan_io.read_img(os.path.join(out_folder, "images", "{:0>8}.jpg".format(1)))
    assert img.dtype == np.float32 and np.array_equal((img * 255).astype(np.uint8), s["img"][1])
    # a camera file that carries a plane count rescales the interval to ndepths planes (datasets/dtu_test.py:98-101)
    cam = np.zeros((2, 4, 4), np.float32)
    cam[0], cam[1, :3, :3], cam[1, 3] = E, K, [425.0, 2.5, 256, 1065.0]
    path = str(tmp_path / "cam.txt")
    scan_io.write_cam(path, cam)
    K3, E3 = scan_io.read_camera_parameters(path)
    assert np.array_equal(K3, K) and np.array_equal(E3, E)
    _, _, dmin, dint = scan_io.read_cam_file(path, interval_scale=1.0, ndepths=192)
    assert dmin == 425.0 and abs(dint - 256 * 2.5 / 192) < 1e-12
    m = np.zeros((4, 5), bool)
    m[1, 2] = True
    scan_io.save_mask(str(tmp_path / "m.png"), m)
    assert np.array_equal(scan_io.read_mask(str(tmp_path / "m.png")), m)
    with pytest.raises(TypeError):
        scan_io.save_mask(str(tmp_path / "m2.png"), m.astype(np.uint8))


def test_ply_bytes():
    xyz, rgb = GOLD["xyz"][:100], GOLD["rgb"][:100]
    b = fusion.ply_bytes(xyz, rgb)
    assert b == O.ply_bytes(xyz, rgb)
    head, body = b.split(b"end_header\n", 1)
    assert head.startswith(b"ply\nformat binary_little_endian 1.0\nelement vertex 100\nproperty float x\n") and len(body) == 100 * 15
    rec = np.frombuffer(body, dtype=[("p", "<f4", 3), ("c", "u1", 3)])
    assert np.array_equal(rec["p"], xyz) and np.array_equal(rec["c"], rgb)


def test_fusion_fails_loudly_without_a_gpu(tmp_path):
    s = scan()
    with pytest.raises((_lib.RcmvsError, RuntimeError, AssertionError)):
        fusion.check_geometric_consistency(s["depth"][0], s["K"][0], s["E"][0], s["depth"][1], s["K"][1], s["E"][1], DIST, DEPTH, device="cpu")
    mats = fusion.fusion_matrices(s["K"][0], s["E"][0], [s["K"][1]], [s["E"][1]])
    assert mats.dtype == np.float64 and mats.shape == (72,)
    with pytest.raises(_lib.RcmvsError):
        fusion.fuse_view(torch.zeros(2, 4, 4), 0, list(range(17)), torch.zeros(4, 4), None, torch.zeros(72, dtype=torch.float64), 0.8, 3, 0.5, 0.01)


def test_filter_scans_shards_scans_round_robin(monkeypatch):
    """pcd_filter's pool over scans -> every world-th scan per rank, no collective."""
    seen = []
    monkeypatch.setattr(fusion, "filter_depth", lambda device, **job: seen.append((device, job["plyfilename"])) or job["plyfilename"])
    jobs = [{"plyfilename": "scan%d.ply" % i} for i in range(5)]
    monkeypatch.setenv("LOCAL_RANK", "1")
    assert fusion.filter_scans(jobs, rank=1, world=2) == ["scan1.ply", "scan3.ply"]
    assert seen == [("cuda:1", "scan1.ply"), ("cuda:1", "scan3.ply")]
    assert fusion.filter_scans(jobs, rank=0, world=2) == ["scan0.ply", "scan2.ply", "scan4.ply"]
    all_ranks = sorted(fusion.filter_scans(jobs, rank=r, world=3)[i] for r in range(3) for i in range(len(jobs[r::3])))
    assert all_ranks == sorted(j["plyfilename"] for j in jobs)


def test_tanks_filter_host_logic_matches_reference(tmp_path, harness, monkeypatch):
    """filter_depth_tanks (eval_rcmvsnet_tanks.py:269-380) end to end on the CPU, with the g++ harnesses standing in for the
    three kernel launches: file layout, intrinsics rescaling, image resize, thresholds, vertex list."""
    from PIL import Image
    from rc_mvsnet_amd import mvs_dataset
    V, h, w, oh, ow, seed, n_src = [int(x) for x in GOLD["tanks:dims"]]
    pix, dth, photo, ncons = [float(x) for x in GOLD["tanks:thresholds"]]
    s = synthetic.tanks_fusion_scan(V=V, hw=(h, w), orig_hw=(oh, ow), seed=seed, n_src=n_src)
    scan_folder, out_folder = str(tmp_path / "tt" / "intermediate" / "Horse"), str(tmp_path / "out" / "Horse")
    synthetic.write_tanks_fusion_scan(s, scan_folder, out_folder)
    ip = str(tmp_path / "ip.so")
    subprocess.run(["g++", "-O2", "-w", "-ffp-contract=off", "-shared", "-fPIC", "-o", ip, os.path.join(HERE, "harness", "image_prep_harness.cpp")], check=True)
    ip = ctypes.CDLL(ip)

    def prepare_cpu(img_u8, out_hw, device, mean=mvs_dataset.MEAN, std=mvs_dataset.STD):
        out = np.empty((3, int(out_hw[0]), int(out_hw[1])), np.float32)
        img_u8, mean, std = np.ascontiguousarray(img_u8), np.array(mean, np.float32), np.array(std, np.float32)
        ip.h_prepare_image(_p(img_u8), _p(out), img_u8.shape[0], img_u8.shape[1], out.shape[1], out.shape[2], _p(mean), _p(std))
        return torch.from_numpy(out)

    def fuse_cpu(depth_all, ref_idx, src_idx, conf, img, mats, prob, ncons_, dist, depth_t, debug=False):
        d, c, im, m = depth_all.numpy(), conf.numpy(), img.numpy(), mats.numpy()
        N, (H, W) = len(src_idx), d.shape[1:]
        masks, avg = np.empty((3, H, W), np.uint8), np.empty((H, W), np.float32)
        xyz, rgb = np.empty((H, W, 3), np.float32), np.empty((H, W, 3), np.uint8)
        idx = np.array(src_idx, np.int32)
        harness.h_fuse_view(_p(d), int(ref_idx), _p(idx), _p(c), _p(im), _p(m), ctypes.c_float(prob), int(ncons_), ctypes.c_double(dist),
                            ctypes.c_float(depth_t), _p(masks), _p(avg), _p(xyz), _p(rgb), None, None, None, N, H, W)
        return {"masks": torch.from_numpy(masks), "depth_avg": torch.from_numpy(avg), "xyz": torch.from_numpy(xyz), "rgb": torch.from_numpy(rgb)}

    monkeypatch.setattr(mvs_dataset, "prepare_image", prepare_cpu)
    monkeypatch.setattr(fusion, "fuse_view", fuse_cpu)
    monkeypatch.setattr(fusion, "compact_points", lambda mask, xyz, rgb=None: (xyz[mask.bool()], rgb[mask.bool()]))
    ply = str(tmp_path / "ply" / "Horse.ply")
    xyz, rgb = fusion.filter_depth_tanks(scan_folder, out_folder, ply, pix, dth, photo, (w, h), (ow, oh), int(ncons), V, "Horse",
                                         device="cpu", verbose=False)
    flips = 0
    for v in range(V):
        got = np.array(Image.open(os.path.join(out_folder, "mask", "{:0>8}_final.png".format(v)))) > 0
        flips += int((got != GOLD["tanks:mask:%d:final" % v]).sum())
    assert flips <= 2, flips
    if flips == 0:
        assert np.allclose(xyz, GOLD["tanks:xyz"], rtol=1e-5, atol=1e-3) and np.array_equal(rgb, GOLD["tanks:rgb"])
    assert open(ply, "rb").read(3) == b"ply"


@pytest.mark.parametrize("case_name", ["zero_depth_holes", "source_out_of_frame", "sixteen_sources", "one_by_one"])
def test_kernel_arithmetic_edge_cases(case_name, harness):
    """Holes (depth 0: 0/0 in the relative test), a source camera that sees nothing, the maximum source count, a 1x1 map."""
    s = synthetic.fusion_scan(V=5, H=(1 if case_name == "one_by_one" else 20), W=(1 if case_name == "one_by_one" else 28), seed=9, n_src=4)
    ref, srcs = s["pairs"][0]
    depth = s["depth"].copy()
    K, E = s["K"].copy(), s["E"].copy()
    if case_name == "zero_depth_holes":
        depth[ref, 3:9, 5:11] = 0.0
        depth[srcs[0], :, :7] = 0.0
    if case_name == "source_out_of_frame":
        E[srcs[1], :3, 3] += np.array([4.0e4, 0.0, 0.0], np.float32)
    if case_name == "sixteen_sources":
        srcs = (srcs * 4)[:16]
    H, W = depth.shape[1:]
    N = len(srcs)
    img = np.ascontiguousarray(s["img"][ref].astype(np.float32) / 255.0)
    mats = fusion.fusion_matrices(K[ref], E[ref], [K[i] for i in srcs], [E[i] for i in srcs])
    masks, avg = np.empty((3, H, W), np.uint8), np.empty((H, W), np.float32)
    xyz, rgb = np.empty((H, W, 3), np.float32), np.empty((H, W, 3), np.uint8)
    dg = np.empty((N, H, W), np.uint8)
    idx, conf, dall = np.array(srcs, np.int32), np.ascontiguousarray(s["conf"][ref]), np.ascontiguousarray(depth)
    harness.h_fuse_view(_p(dall), ref, _p(idx), _p(conf), _p(img), _p(mats), ctypes.c_float(PROB), NCONS, ctypes.c_double(DIST),
                        ctypes.c_float(DEPTH), _p(masks), _p(avg), _p(xyz), _p(rgb), None, _p(dg), None, N, H, W)
    with np.errstate(all="ignore"):
        r = O.fuse_view(depth[ref], s["conf"][ref], img, K[ref], E[ref], [depth[i] for i in srcs], [K[i] for i in srcs], [E[i] for i in srcs],
                        PROB, NCONS, DIST, DEPTH)
    assert np.array_equal(masks[0].astype(bool), r["photo"])
    assert (masks[1].astype(bool) != r["geo"]).sum() <= 1 and (masks[2].astype(bool) != r["final"]).sum() <= 1
    both = masks[2].astype(bool) & r["final"]
    want = np.zeros((H, W, 3), np.float32)
    want[r["final"]] = r["xyz"]
    assert np.allclose(xyz[both], want[both], rtol=1e-5, atol=1e-3)
    if case_name == "zero_depth_holes":
        assert not masks[1][3:9, 5:11].any()                    # a hole in the reference map is never consistent
    if case_name == "source_out_of_frame":
        assert not dg[1].any()
