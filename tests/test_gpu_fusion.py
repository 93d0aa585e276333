"""GPU: the depth-map fusion filter on the HIP path (rc_mvsnet_amd/fusion.py over rcmvs_fuse_view / rcmvs_compact_points)
against the reference's golden outputs (tests/golden/fusion.npz, produced by importing eval_rcmvsnet_dtu.py) and the oracle."""
import os

import numpy as np
import pytest
import torch

from oracle import fusion as O
from rc_mvsnet_amd import _lib, fusion, synthetic

pytestmark = pytest.mark.gpu

HERE = os.path.dirname(os.path.abspath(__file__))
GOLD = np.load(os.path.join(HERE, "golden", "fusion.npz"))
PROB, NCONS, DIST, DEPTH = 0.8, 3, 0.5, 0.01
DEV = "cuda:0"


def scan():
    V, H, W, seed, n_src = [int(x) for x in GOLD["dims"]]
    return synthetic.fusion_scan(V=V, H=H, W=W, seed=seed, n_src=n_src)


def test_check_geometric_consistency_matches_reference():
    _lib.load()
    s = scan()
    m, back, xs, ys = fusion.check_geometric_consistency(s["depth"][0], s["K"][0], s["E"][0], s["depth"][2], s["K"][2], s["E"][2], DIST, DEPTH)
    assert m.dtype == bool and m.shape == GOLD["pair02:mask"].shape
    agree = m == GOLD["pair02:mask"]
    assert (~agree).sum() <= 1
    assert np.allclose(back[agree], GOLD["pair02:depth"][agree], rtol=1e-6, atol=1e-3)
    assert np.allclose(xs, GOLD["pair02:x_src"], rtol=1e-6, atol=1e-4) and np.allclose(ys, GOLD["pair02:y_src"], rtol=1e-6, atol=1e-4)


def test_filter_depth_matches_reference(tmp_path):
    """The whole scan through the files the reference reads and writes: mask images and the fused vertex list."""
    _lib.load()
    from PIL import Image
    s = scan()
    pair_folder, out_folder = str(tmp_path / "data" / "scan1"), str(tmp_path / "out" / "scan1")
    synthetic.write_fusion_scan(s, pair_folder, out_folder)
    ply = str(tmp_path / "out" / "fused.ply")
    xyz, rgb = fusion.filter_depth(pair_folder, out_folder, out_folder, ply, PROB, NCONS, DIST, DEPTH, verbose=False)
    flips = 0
    for v in range(len(s["depth"])):
        for kind in ("photo", "geo", "final"):
            got = np.array(Image.open(os.path.join(out_folder, "mask", "{:0>8}_{}.png".format(v, kind)))) > 0
            flips += int((got != GOLD["mask:%d:%s" % (v, kind)]).sum())
    assert flips <= 4, flips                                              # threshold knife edges only
    if flips == 0:
        assert xyz.shape == GOLD["xyz"].shape
        assert np.allclose(xyz, GOLD["xyz"], rtol=1e-5, atol=1e-3)
        assert np.array_equal(rgb, GOLD["rgb"])
    else:
        assert abs(len(xyz) - len(GOLD["xyz"])) <= flips
    body = open(ply, "rb").read().split(b"end_header\n", 1)[1]
    rec = np.frombuffer(body, dtype=[("p", "<f4", 3), ("c", "u1", 3)])
    assert np.array_equal(rec["p"], xyz) and np.array_equal(rec["c"], rgb)


@pytest.mark.parametrize("shape,frac", [((1184, 1600), 0.3), ((37, 53), 0.5), ((16, 16), 0.0), ((16, 17), 1.0), ((600, 700), 0.001)])
def test_compact_points_is_ordered_boolean_indexing(shape, frac):
    _lib.load()
    g = torch.Generator().manual_seed(shape[0])
    mask = (torch.rand(shape, generator=g) < frac).to(torch.uint8)
    xyz = torch.randn(*shape, 3, generator=g)
    rgb = (255 * torch.rand(*shape, 3, generator=g)).to(torch.uint8)
    oxyz, orgb = fusion.compact_points(mask.to(DEV), xyz.to(DEV), rgb.to(DEV))
    assert torch.equal(oxyz.cpu(), xyz[mask.bool()]) and torch.equal(orgb.cpu(), rgb[mask.bool()])
    oxyz2, none = fusion.compact_points(mask.to(DEV), xyz.to(DEV))
    assert none is None and torch.equal(oxyz2.cpu(), xyz[mask.bool()])


def test_fuse_view_full_size_against_oracle():
    """The reference's DTU evaluation shape (1184 x 1600 depth maps, 4 of the 10 source views to keep the CPU oracle to
    seconds): masks, averaged depth and world points."""
    _lib.load()
    s = synthetic.fusion_scan(V=5, H=1184, W=1600, seed=3, n_src=4)
    ref, srcs = s["pairs"][1]
    img = s["img"][ref].astype(np.float32) / 255.0
    r = O.fuse_view(s["depth"][ref], s["conf"][ref], img, s["K"][ref], s["E"][ref], [s["depth"][i] for i in srcs],
                    [s["K"][i] for i in srcs], [s["E"][i] for i in srcs], PROB, NCONS, DIST, DEPTH)
    mats = torch.from_numpy(fusion.fusion_matrices(s["K"][ref], s["E"][ref], [s["K"][i] for i in srcs], [s["E"][i] for i in srcs])).to(DEV)
    g = fusion.fuse_view(torch.from_numpy(s["depth"]).to(DEV), ref, srcs, torch.from_numpy(s["conf"][ref]).to(DEV),
                         torch.from_numpy(img).to(DEV), mats, PROB, NCONS, DIST, DEPTH)
    masks = g["masks"].cpu().numpy().astype(bool)
    assert np.array_equal(masks[0], r["photo"])
    assert (masks[1] != r["geo"]).mean() < 1e-4 and (masks[2] != r["final"]).mean() < 1e-4
    both = masks[2] & r["final"]
    assert np.allclose(g["depth_avg"].cpu().numpy()[both], r["depth_avg"].astype(np.float32)[both], rtol=1e-6)
    want = np.zeros(masks[2].shape + (3,), np.float32)
    want[r["final"]] = r["xyz"]
    assert np.allclose(g["xyz"].cpu().numpy()[both], want[both], rtol=1e-5, atol=1e-3)
    xyz, rgb = fusion.compact_points(g["masks"][2], g["xyz"], g["rgb"])
    assert len(xyz) == int(masks[2].sum()) and torch.equal(rgb.cpu(), g["rgb"].cpu()[torch.from_numpy(masks[2])])


def test_fuse_view_argument_checks():
    _lib.load()
    d = torch.zeros(2, 8, 8, device=DEV)
    mats = torch.zeros(72, dtype=torch.float64, device=DEV)
    with pytest.raises(_lib.RcmvsError):
        fusion.fuse_view(d, 0, [2], d[0], None, mats, PROB, NCONS, DIST, DEPTH)                   # view index beyond depth_all
    with pytest.raises(_lib.RcmvsError):
        fusion.fuse_view(d, 0, [1], d[0], None, mats[:70], PROB, NCONS, DIST, DEPTH)              # wrong matrix count
    with pytest.raises(_lib.RcmvsError):
        fusion.fuse_view(d, 0, [1], d[0].double(), None, mats, PROB, NCONS, DIST, DEPTH)          # fp32 confidence only
    # all-zero depth (division by zero in the relative test): nothing is consistent, nothing is NaN in the masks
    r = fusion.fuse_view(d, 0, [1], d[0], None, torch.ones(72, dtype=torch.float64, device=DEV), PROB, 1, DIST, DEPTH)
    assert int(r["masks"][1].sum()) == 0


def test_filter_depth_tanks_matches_reference(tmp_path):
    """The Tanks-and-Temples form (eval_rcmvsnet_tanks.py:269-380): original-size cameras and images, network-size depth maps."""
    _lib.load()
    from PIL import Image
    V, h, w, oh, ow, seed, n_src = [int(x) for x in GOLD["tanks:dims"]]
    pix, dth, photo, ncons = [float(x) for x in GOLD["tanks:thresholds"]]
    s = synthetic.tanks_fusion_scan(V=V, hw=(h, w), orig_hw=(oh, ow), seed=seed, n_src=n_src)
    scan_folder, out_folder = str(tmp_path / "tt" / "intermediate" / "Horse"), str(tmp_path / "out" / "Horse")
    synthetic.write_tanks_fusion_scan(s, scan_folder, out_folder)
    xyz, rgb = fusion.filter_depth_tanks(scan_folder, out_folder, str(tmp_path / "ply" / "Horse.ply"), pix, dth, photo, (w, h), (ow, oh),
                                         int(ncons), V, "Horse", verbose=False)
    flips = 0
    for v in range(V):
        got = np.array(Image.open(os.path.join(out_folder, "mask", "{:0>8}_final.png".format(v)))) > 0
        flips += int((got != GOLD["tanks:mask:%d:final" % v]).sum())
    assert flips <= 2, flips
    if flips == 0:
        assert np.allclose(xyz, GOLD["tanks:xyz"], rtol=1e-5, atol=1e-3)
        assert (rgb.astype(int) - GOLD["tanks:rgb"].astype(int)).__abs__().max() <= 1      # colour: resize restated, 1 level of 255
    with pytest.raises(_lib.RcmvsError):
        fusion.filter_depth_tanks(scan_folder, out_folder, str(tmp_path / "x.ply"), pix, dth, photo, (w + 32, h), (ow, oh), int(ncons), V, "Horse",
                                  verbose=False)
