"""Depth-map fusion filter on the HIP path.

Same entry points as the reference's evaluation script (eval_rcmvsnet_dtu.py:281-446; eval_rcmvsnet_tanks.py:206-382 is the
same code): ``check_geometric_consistency`` for one pair of views and ``filter_depth`` for a scan laid out on disk
(pair.txt, cams/, images/, depth_est/*.pfm, confidence/*.pfm -> mask/*.png and one fused .ply).  The reference runs this
on the CPU, a ``multiprocessing.Pool`` of one process per scan, ~40 numpy passes per (reference, source) pair; here every
depth map of the scan is uploaded ONCE and stays in HBM, a reference view is one kernel over all of its source views
(``rcmvs_fuse_view``) plus an ordered compaction of the surviving points (``rcmvs_compact_points``), and scans shard over
GPUs with no collective (``filter_scans``).  No CPU fallback: without a GPU / the built library these functions raise.
"""
import ctypes
import os

import numpy as np
import torch
from PIL import Image

from . import _lib, scan_io
from .data_io import read_pfm
from .ops import _chk, _stream

MAX_SRC = 16      # RCMVS_FUSE_MAX_SRC
REF_MATS, SRC_MATS = 30, 42


def fusion_matrices(K_ref, E_ref, src_K, src_E):
    """The matrices of rcmvs_fuse_view as one float64 vector (30 + 42 N).  Inverses and products are taken in the dtype the
    reference has them in (float32 camera files -> float32 np.linalg.inv / np.matmul, eval_rcmvsnet_dtu.py:289-313) and only
    then promoted, so the per-pixel chain sees the same numbers."""
    K_ref, E_ref = np.asarray(K_ref), np.asarray(E_ref)
    E_ref_inv = np.linalg.inv(E_ref)
    out = [np.linalg.inv(K_ref).ravel(), K_ref.ravel(), E_ref_inv[:3, :4].ravel()]
    for K, E in zip(src_K, src_E):
        K, E = np.asarray(K), np.asarray(E)
        out += [np.matmul(E, E_ref_inv)[:3, :4].ravel(), K.ravel(), np.linalg.inv(K).ravel(),
                np.matmul(E_ref, np.linalg.inv(E))[:3, :4].ravel()]
    return np.concatenate([o.astype(np.float64) for o in out])


def _ptr(t, name, dtype):
    return ctypes.c_void_p(0) if t is None else _chk(t, name, dtype)


def fuse_view(depth_all, ref_idx, src_idx, conf, img, mats, prob_threshold, num_consistent, img_dist_thresh, depth_thresh, debug=False):
    """Device tensors in, device tensors out.  depth_all (n_views,H,W) fp32, conf (H,W) fp32, img (H,W,3) fp32 in [0,1] or
    None, mats float64 (30 + 42 N).  Returns dict: masks (3,H,W) uint8 [photo, geo, final], depth_avg (H,W), xyz (H,W,3),
    rgb (H,W,3) uint8 or None, and with debug=True depth_reprojected (N,H,W), geo (N,H,W) uint8, xy_src (N,H,W,2)."""
    N = len(src_idx)
    if not 1 <= N <= MAX_SRC:
        raise _lib.RcmvsError(f"fuse_view: {N} source views (1..{MAX_SRC})")
    n_views, H, W = depth_all.shape
    if max([ref_idx] + list(src_idx)) >= n_views:
        raise _lib.RcmvsError("fuse_view: view index beyond depth_all")
    if mats.numel() != REF_MATS + SRC_MATS * N:
        raise _lib.RcmvsError(f"fuse_view: mats has {mats.numel()} values, expected {REF_MATS + SRC_MATS * N}")
    dev = depth_all.device
    masks = torch.empty((3, H, W), device=dev, dtype=torch.uint8)
    depth_avg = torch.empty((H, W), device=dev, dtype=torch.float32)
    xyz = torch.empty((H, W, 3), device=dev, dtype=torch.float32)
    rgb = torch.empty((H, W, 3), device=dev, dtype=torch.uint8) if img is not None else None
    dbg_d = torch.empty((N, H, W), device=dev, dtype=torch.float32) if debug else None
    dbg_g = torch.empty((N, H, W), device=dev, dtype=torch.uint8) if debug else None
    dbg_xy = torch.empty((N, H, W, 2), device=dev, dtype=torch.float32) if debug else None
    idx = (ctypes.c_int * N)(*[int(i) for i in src_idx])
    _lib.check(_lib.load().rcmvs_fuse_view(
        _chk(depth_all, "depth_all"), int(ref_idx), ctypes.cast(idx, ctypes.c_void_p), _chk(conf, "conf"), _ptr(img, "img", torch.float32),
        _chk(mats, "mats", torch.float64), float(prob_threshold), int(num_consistent), float(img_dist_thresh), float(depth_thresh),
        _chk(masks, "masks", torch.uint8), _chk(depth_avg, "depth_avg"), _chk(xyz, "xyz"), _ptr(rgb, "rgb", torch.uint8),
        _ptr(dbg_d, "dbg_depth", torch.float32), _ptr(dbg_g, "dbg_geo", torch.uint8), _ptr(dbg_xy, "dbg_xy", torch.float32),
        N, H, W, _stream()), "fuse_view")
    out = {"masks": masks, "depth_avg": depth_avg, "xyz": xyz, "rgb": rgb}
    if debug:
        out.update({"depth_reprojected": dbg_d, "geo": dbg_g, "xy_src": dbg_xy})
    return out


def compact_points(mask, xyz, rgb=None):
    """numpy's ``xyz[mask]`` / ``rgb[mask]`` on the device, row-major order kept.  mask (H,W) uint8 -> (n,3) fp32[, (n,3) uint8]."""
    n = mask.numel()
    dev = mask.device
    out_xyz = torch.empty((n, 3), device=dev, dtype=torch.float32)
    out_rgb = torch.empty((n, 3), device=dev, dtype=torch.uint8) if rgb is not None else None
    offsets = torch.empty((n + 255) // 256 + 1, device=dev, dtype=torch.int32)
    _lib.check(_lib.load().rcmvs_compact_points(_chk(mask, "mask", torch.uint8), _chk(xyz, "xyz"), _ptr(rgb, "rgb", torch.uint8),
                                                _chk(out_xyz, "out_xyz"), _ptr(out_rgb, "out_rgb", torch.uint8),
                                                _chk(offsets, "offsets", torch.int32), n, _stream()), "compact_points")
    kept = int(offsets[-1])                                     # the one host synchronisation of a reference view
    return out_xyz[:kept], (out_rgb[:kept] if rgb is not None else None)


def check_geometric_consistency(depth_ref, intrinsics_ref, extrinsics_ref, depth_src, intrinsics_src, extrinsics_src,
                                img_dist_thresh, depth_thresh, device="cuda:0"):
    """eval_rcmvsnet_dtu.py:324-338 for one (reference, source) pair of numpy arrays:
    -> mask (H,W) bool, depth_reprojected (H,W) fp32 (0 where inconsistent), x2d_src, y2d_src (H,W) fp32."""
    dev = torch.device(device)
    depth_all = torch.from_numpy(np.stack([depth_ref, depth_src]).astype(np.float32)).to(dev)
    mats = torch.from_numpy(fusion_matrices(intrinsics_ref, extrinsics_ref, [intrinsics_src], [extrinsics_src])).to(dev)
    conf = torch.zeros(depth_ref.shape, device=dev, dtype=torch.float32)
    r = fuse_view(depth_all, 0, [1], conf, None, mats, 0.0, 1, img_dist_thresh, depth_thresh, debug=True)
    xy = r["xy_src"][0].cpu().numpy()
    return r["geo"][0].cpu().numpy().astype(bool), r["depth_reprojected"][0].cpu().numpy(), xy[..., 0].copy(), xy[..., 1].copy()


def ply_bytes(xyz, rgb):
    """Binary little-endian PLY with vertex properties x y z (float) red green blue (uchar): the file the reference writes
    through plyfile (eval_rcmvsnet_dtu.py:433-446)."""
    n = len(xyz)
    head = ("ply\nformat binary_little_endian 1.0\nelement vertex %d\nproperty float x\nproperty float y\nproperty float z\n"
            "property uchar red\nproperty uchar green\nproperty uchar blue\nend_header\n" % n).encode("ascii")
    rec = np.empty(n, dtype=[("x", "<f4"), ("y", "<f4"), ("z", "<f4"), ("red", "u1"), ("green", "u1"), ("blue", "u1")])
    for i, k in enumerate(("x", "y", "z")):
        rec[k] = xyz[:, i]
    for i, k in enumerate(("red", "green", "blue")):
        rec[k] = rgb[:, i]
    return head + rec.tobytes()


def stage_colour(img, num_stage, shape):
    """The colour image at the depth map's resolution (eval_rcmvsnet_dtu.py:410-415)."""
    if num_stage == 1:
        img = img[1::4, 1::4, :]
    elif num_stage == 2:
        img = img[1::2, 1::2, :]
    if img.shape[:2] != tuple(shape):
        raise _lib.RcmvsError(f"filter_depth: image {img.shape[:2]} does not match the depth map {tuple(shape)}")
    return np.ascontiguousarray(img)


def filter_depth(pair_folder, scan_folder, out_folder, plyfilename, prob_threshold, num_consistent, img_dist_thresh, depth_thresh,
                 num_stage=3, device="cuda:0", save_masks=True, verbose=True):
    """eval_rcmvsnet_dtu.py:341-446: fuse one scan into ``plyfilename``; returns (xyz (n,3) fp32, rgb (n,3) uint8)."""
    dev = torch.device(device)
    pairs = scan_io.read_pair_file(os.path.join(pair_folder, "pair.txt"))
    views = sorted({v for ref, srcs in pairs for v in [ref] + list(srcs)})
    slot = {v: i for i, v in enumerate(views)}
    cams = {v: scan_io.read_camera_parameters(os.path.join(scan_folder, "cams/{:0>8}_cam.txt".format(v))) for v in views}
    depth_all = torch.from_numpy(np.stack([read_pfm(os.path.join(out_folder, "depth_est/{:0>8}.pfm".format(v)))[0] for v in views])).to(dev)
    if save_masks:
        os.makedirs(os.path.join(out_folder, "mask"), exist_ok=True)
    pts, cols = [], []
    for ref, srcs in pairs:
        if len(srcs) > MAX_SRC:
            raise _lib.RcmvsError(f"filter_depth: view {ref} lists {len(srcs)} source views (at most {MAX_SRC})")
        conf = torch.from_numpy(read_pfm(os.path.join(out_folder, "confidence/{:0>8}.pfm".format(ref)))[0]).to(dev)
        img = scan_io.read_img(os.path.join(scan_folder, "images/{:0>8}.jpg".format(ref)))
        img = torch.from_numpy(stage_colour(img, num_stage, conf.shape)).to(dev)
        mats = torch.from_numpy(fusion_matrices(cams[ref][0], cams[ref][1], [cams[s][0] for s in srcs], [cams[s][1] for s in srcs])).to(dev)
        r = fuse_view(depth_all, slot[ref], [slot[s] for s in srcs], conf, img, mats, prob_threshold, num_consistent, img_dist_thresh, depth_thresh)
        xyz, rgb = compact_points(r["masks"][2], r["xyz"], r["rgb"])
        pts.append(xyz.cpu().numpy())
        cols.append(rgb.cpu().numpy())
        if save_masks or verbose:
            m = r["masks"].cpu().numpy().astype(bool)
            if save_masks:
                for k, kind in enumerate(("photo", "geo", "final")):
                    scan_io.save_mask(os.path.join(out_folder, "mask/{:0>8}_{}.png".format(ref, kind)), m[k])
            if verbose:
                print("processing {}, ref-view{:0>2}, photo/geo/final-mask:{}/{}/{}".format(scan_folder, ref, m[0].mean(), m[1].mean(), m[2].mean()))
    xyz, rgb = np.concatenate(pts, 0), np.concatenate(cols, 0)
    with open(plyfilename, "wb") as f:
        f.write(ply_bytes(xyz, rgb))
    if verbose:
        print("saving the final model to", plyfilename)
    return xyz, rgb


def filter_depth_tanks(scan_folder, out_folder, plyfilename, geo_pixel_thres, geo_depth_thres, photo_thres, img_wh, image_sizes,
                       geo_mask_thres, n_views=None, scan="", device="cuda:0", save_masks=True, verbose=True):
    """eval_rcmvsnet_tanks.py:269-380, the Tanks-and-Temples form of filter_depth: pair.txt, cams_1/ and images/ live in
    ``scan_folder`` at the ORIGINAL image size ``image_sizes`` = (w, h); the depth maps under ``out_folder`` have the network
    size ``img_wh`` = (w, h), so the intrinsics' first two rows are rescaled and the colour image is resized (cv2.resize's
    rule, on the device).  Same kernels as filter_depth.  Returns (xyz (n,3) fp32, rgb (n,3) uint8)."""
    from .mvs_dataset import prepare_image
    dev = torch.device(device)
    pairs = scan_io.read_pair_file(os.path.join(scan_folder, "pair.txt"))
    views = sorted({v for ref, srcs in pairs for v in [ref] + list(srcs)})
    slot = {v: i for i, v in enumerate(views)}
    ow, oh = image_sizes
    cams = {}
    for v in views:
        K, E = scan_io.read_camera_parameters(os.path.join(scan_folder, "cams_1/{:0>8}_cam.txt".format(v)))
        K[0] *= img_wh[0] / ow
        K[1] *= img_wh[1] / oh
        cams[v] = (K, E)
    depth_all = torch.from_numpy(np.stack([read_pfm(os.path.join(out_folder, "depth_est/{:0>8}.pfm".format(v)))[0] for v in views])).to(dev)
    if tuple(depth_all.shape[1:]) != (img_wh[1], img_wh[0]):
        raise _lib.RcmvsError(f"filter_depth_tanks: depth maps are {tuple(depth_all.shape[1:])}, img_wh says {(img_wh[1], img_wh[0])}")
    if save_masks:
        os.makedirs(os.path.join(out_folder, "mask"), exist_ok=True)
    pts, cols = [], []
    for ref, srcs in pairs:
        if len(srcs) > MAX_SRC:
            raise _lib.RcmvsError(f"filter_depth_tanks: view {ref} lists {len(srcs)} source views (at most {MAX_SRC})")
        conf = torch.from_numpy(read_pfm(os.path.join(out_folder, "confidence/{:0>8}.pfm".format(ref)))[0]).to(dev)
        raw = np.array(Image.open(os.path.join(scan_folder, "images/{:0>8}.jpg".format(ref))), dtype=np.uint8)
        img = prepare_image(raw, (img_wh[1], img_wh[0]), dev, mean=(0.0, 0.0, 0.0), std=(1.0, 1.0, 1.0)).permute(1, 2, 0).contiguous()
        mats = torch.from_numpy(fusion_matrices(cams[ref][0], cams[ref][1], [cams[s][0] for s in srcs], [cams[s][1] for s in srcs])).to(dev)
        r = fuse_view(depth_all, slot[ref], [slot[s] for s in srcs], conf, img, mats, photo_thres, geo_mask_thres, geo_pixel_thres, geo_depth_thres)
        xyz, rgb = compact_points(r["masks"][2], r["xyz"], r["rgb"])
        pts.append(xyz.cpu().numpy())
        cols.append(rgb.cpu().numpy())
        if save_masks or verbose:
            m = r["masks"].cpu().numpy().astype(bool)
            if save_masks:
                for k, kind in enumerate(("photo", "geo", "final")):
                    scan_io.save_mask(os.path.join(out_folder, "mask/{:0>8}_{}.png".format(ref, kind)), m[k])
            if verbose:
                print("processing {}, ref-view{:0>2}, geo_mask:{:3f} photo_mask:{:3f} final_mask: {:3f}".format(
                    scan_folder, ref, m[1].mean(), m[0].mean(), m[2].mean()))
    xyz, rgb = np.concatenate(pts, 0), np.concatenate(cols, 0)
    os.makedirs(os.path.dirname(os.path.abspath(plyfilename)), exist_ok=True)
    with open(plyfilename, "wb") as f:
        f.write(ply_bytes(xyz, rgb))
    if verbose:
        print("saving the final model to", plyfilename)
    return xyz, rgb


def filter_scans(jobs, rank=None, world=None, **kwargs):
    """pcd_filter (eval_rcmvsnet_dtu.py:503-515) without the process pool: jobs = [dict of filter_depth arguments]; each rank
    (one process per GPU, RANK / WORLD_SIZE / LOCAL_RANK from the environment) takes every world-th scan, no collective."""
    from .sharding import shard_items
    rank = int(os.environ.get("RANK", "0")) if rank is None else rank
    world = int(os.environ.get("WORLD_SIZE", "1")) if world is None else world
    device = "cuda:%d" % int(os.environ.get("LOCAL_RANK", "0"))
    return [filter_depth(device=device, **dict(job, **kwargs)) for job in shard_items(list(jobs), rank, world)]
