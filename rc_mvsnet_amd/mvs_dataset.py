"""Evaluation loaders for MVSNet-style scan folders: the test-mode ``MVSDataset`` of
datasets/dtu_test.py:11-229 and the Tanks-and-Temples one of datasets/tanks.py:11-186 (``TanksDataset`` here), with the
per-image work moved to the GPU.

Same constructor arguments, item order and item dict (``imgs`` (V,3,h,w), ``proj_matrices`` {stage1..3: (V,2,4,4)},
``depth_values`` (ndepths,), ``filename``) as the reference, so the loop of ``eval_rcmvsnet_dtu.py:174-197`` consumes the
items unchanged -- with ``DataLoader(num_workers=0)`` or the ``prefetch()`` iterator below: ``__getitem__`` launches HIP
kernels, so a forked loader worker (the reference's ``num_workers=1``) cannot run it ("Cannot re-initialize CUDA in forked
subprocess").  The
host parses the text files and decodes the JPEG; ``/255``, the ``cv2.resize`` of ``scale_mvs_input`` / the common-size resize,
``ToTensor`` and ``Normalize`` are one kernel per image (``rcmvs_prepare_image``) on the uploaded bytes -- that work costs the
reference's single loader worker ~10x the network's time per item.  ``imgs`` is therefore a CUDA tensor; everything else is
numpy like the reference's.  No CPU fallback: ``device`` must be a GPU.
"""
import ctypes
import os

import numpy as np
import torch
from PIL import Image

from . import _lib, scan_io
from .ops import _chk, _stream

MEAN = (0.485, 0.456, 0.406)      # transforms.Normalize of datasets/dtu_test.py:78-81
STD = (0.229, 0.224, 0.225)


def scaled_size(h, w, max_h, max_w, base=32):
    """Target (new_h, new_w) of scale_mvs_input (datasets/dtu_test.py:127-137): fit inside (max_h, max_w), then round both
    sides down to a multiple of ``base``.  Floats, as the reference computes them."""
    if h > max_h or w > max_w:
        scale = 1.0 * max_h / h
        if scale * w > max_w:
            scale = 1.0 * max_w / w
        return scale * h // base * base, scale * w // base * base
    return 1.0 * h // base * base, 1.0 * w // base * base


def prepare_image(img_u8, out_hw, device, mean=MEAN, std=STD):
    """Decoded image (H,W,3) uint8 numpy -> (3,h,w) fp32 CUDA tensor, resized and normalised on the device
    (mean 0 / std 1 gives the plain resized image in [0,1])."""
    if img_u8.dtype != np.uint8 or img_u8.ndim != 3 or img_u8.shape[2] != 3:
        raise _lib.RcmvsError(f"prepare_image: expected an (H,W,3) uint8 image, got {img_u8.dtype} {img_u8.shape}")
    H, W = img_u8.shape[:2]
    h, w = int(out_hw[0]), int(out_hw[1])
    src = torch.from_numpy(np.ascontiguousarray(img_u8)).to(device, non_blocking=True)
    out = torch.empty((3, h, w), device=device, dtype=torch.float32)
    mean, std = (ctypes.c_float * 3)(*mean), (ctypes.c_float * 3)(*std)
    _lib.check(_lib.load().rcmvs_prepare_image(_chk(src, "src", torch.uint8), _chk(out, "out"), H, W, h, w,
                                               ctypes.cast(mean, ctypes.c_void_p), ctypes.cast(std, ctypes.c_void_p), _stream()),
               "prepare_image")
    return out


class MVSDataset(torch.utils.data.Dataset):
    def __init__(self, datapath, listfile, mode, nviews, ndepths=192, interval_scale=1.06, device="cuda:0", **kwargs):
        super().__init__()
        assert mode == "test"
        self.datapath, self.listfile, self.mode, self.nviews, self.ndepths = datapath, listfile, mode, nviews, ndepths
        self.max_h, self.max_w = kwargs["max_h"], kwargs["max_w"]
        if kwargs.get("fix_res", False):
            raise ValueError("fix_res=True (one resolution latched for the whole list, datasets/dtu_test.py:201-205) is not provided: "
                             "every item is scaled by its own size, the reference's default")
        self.fix_res = False
        self.device = torch.device(device)
        self.interval_scale = {scan: (interval_scale if isinstance(interval_scale, float) else interval_scale[scan]) for scan in listfile}
        self.metas = self.build_list()

    def build_list(self):
        """[(scan, ref_view, src_views, scan)]; short source lists are padded with their first entry (dtu_test.py:28-57)."""
        metas = []
        for scan in self.listfile:
            for ref, srcs in scan_io.read_pair_file(os.path.join(self.datapath, "{}/pair.txt".format(scan))):
                if len(srcs) < self.nviews:
                    srcs = srcs + [srcs[0]] * (self.nviews - len(srcs))
                metas.append((scan, ref, srcs, scan))
        return metas

    def __len__(self):
        return len(self.metas)

    def load_host(self, idx):
        """Everything of item ``idx`` that needs no GPU: file parsing, JPEG decoding, target size, scaled intrinsics, depth values.
        Thread-safe (``prefetch`` runs it on worker threads; PIL releases the GIL while decoding)."""
        scan, ref_view, src_views, scene = self.metas[idx]
        view_ids = [ref_view] + src_views[:self.nviews - 1]
        raws, projs, depth_values, size = [], [], None, None
        for i, vid in enumerate(view_ids):
            name = os.path.join(self.datapath, "{}/images_post/{:0>8}.jpg".format(scan, vid))
            if not os.path.exists(name):
                name = os.path.join(self.datapath, "{}/images/{:0>8}.jpg".format(scan, vid))
            K, E, depth_min, depth_interval = scan_io.read_cam_file(
                os.path.join(self.datapath, "{}/cams/{:0>8}_cam.txt".format(scan, vid)), self.interval_scale[scene], self.ndepths)
            raw = np.array(Image.open(name), dtype=np.uint8)
            h0, w0 = raw.shape[:2]
            new_h, new_w = scaled_size(h0, w0, self.max_h, self.max_w)
            K[0, :] *= 1.0 * new_w / w0
            K[1, :] *= 1.0 * new_h / h0
            c_h, c_w = int(new_h), int(new_w)
            if i == 0:
                size = (c_h, c_w)
            elif (c_h, c_w) != size:
                # dtu_test.py:171-189 has a "resize to the standard size" branch, but it measures the already channels-first
                # tensor ((3, h) instead of (h, w)) and would hand that tensor to cv2.resize: views of one item must agree
                raise _lib.RcmvsError(f"view {vid} of {scan}: size {(c_h, c_w)} differs from the reference view's {size}")
            raws.append(raw)
            p = np.zeros((2, 4, 4), dtype=np.float32)
            p[0, :4, :4] = E
            p[1, :3, :3] = K
            projs.append(p)
            if i == 0:
                depth_values = np.arange(depth_min, depth_interval * (self.ndepths - 0.5) + depth_min, depth_interval, dtype=np.float32)
        return {"raw": raws, "size": size, "proj": np.stack(projs), "depth_values": depth_values,
                "filename": scan + "/{}/" + "{:0>8}".format(view_ids[0]) + "{}"}

    def to_device(self, host):
        return _finish_item(host, self.device)

    def __getitem__(self, idx):
        return self.to_device(self.load_host(idx))


def _finish_item(host, device):
    """The device half of an item: one rcmvs_prepare_image launch per view, then the three-stage projection matrices."""
    imgs = [prepare_image(raw, host["size"], device) for raw in host["raw"]]
    proj = host["proj"]
    stages = {"stage1": proj}
    for key, mul in (("stage2", 2), ("stage3", 4)):
        q = proj.copy()
        q[:, 1, :2, :] = proj[:, 1, :2, :] * mul
        stages[key] = q
    return {"imgs": torch.stack(imgs), "proj_matrices": stages, "depth_values": host["depth_values"], "filename": host["filename"]}


def prefetch(dataset, indices=None, workers=4, depth=8):
    """Items of ``dataset`` in order, with the host half (``load_host``: parsing + JPEG decoding, ~15 ms per 1200x1600 image, i.e.
    several times the network's time per item) of up to ``depth`` items running ahead on ``workers`` threads -- the reference's
    DataLoader(num_workers=1) serialises it with the GPU.  The device half runs on the calling thread / current stream."""
    from concurrent.futures import ThreadPoolExecutor
    idx = list(range(len(dataset))) if indices is None else list(indices)
    with ThreadPoolExecutor(max_workers=max(1, workers)) as pool:
        pending = []
        nxt = 0
        while nxt < len(idx) or pending:
            while nxt < len(idx) and len(pending) < max(1, depth):
                pending.append(pool.submit(dataset.load_host, idx[nxt]))
                nxt += 1
            host = pending.pop(0).result()                          # re-raises a worker's exception here, in order
            yield dataset.to_device(host)


class AsyncWriter:
    """Runs output writers (PFM / camera / JPEG files) on background threads so that disk I/O overlaps the next item's network
    pass; ``close()`` (or leaving the ``with`` block) waits for all of them and re-raises the first failure."""

    def __init__(self, workers=2):
        from concurrent.futures import ThreadPoolExecutor
        self.pool = ThreadPoolExecutor(max_workers=max(1, workers))
        self.jobs = []

    def submit(self, fn, *args, **kwargs):
        self.jobs.append(self.pool.submit(fn, *args, **kwargs))

    def close(self):
        jobs, self.jobs = self.jobs, []
        err = None
        for j in jobs:
            try:
                j.result()
            except Exception as e:                                  # noqa: BLE001 -- keep draining, report the first
                err = err or e
        self.pool.shutdown(wait=True)
        if err is not None:
            raise err

    def __enter__(self):
        return self

    def __exit__(self, exc_type, exc, tb):
        if exc_type is None:
            self.close()
        else:
            self.pool.shutdown(wait=True)
        return False


TANKS_SCANS = {     # scan -> (width, height) of the original images (datasets/tanks.py:24-48)
    "intermediate": {"Family": (1920, 1080), "Francis": (1920, 1080), "Horse": (1920, 1080), "Lighthouse": (2048, 1080),
                     "M60": (2048, 1080), "Panther": (2048, 1080), "Playground": (1920, 1080), "Train": (1920, 1080)},
    "advanced": {"Auditorium": (1920, 1080), "Ballroom": (1920, 1080), "Courtroom": (1920, 1080), "Museum": (1920, 1080),
                 "Palace": (1920, 1080), "Temple": (1920, 1080)},
}


class TanksDataset(torch.utils.data.Dataset):
    """datasets/tanks.py ``MVSDataset``: <datapath>/<split>/<scan>/{pair.txt, cams_1/, images/}; every image is resized to
    ``img_wh`` (no rounding to multiples of 32), the camera file's last line is ``depth_min depth_max`` and the ``ndepths``
    planes span exactly that range.  ``scans`` restricts the split's scan list (the reference always walks all of them)."""

    def __init__(self, datapath, split="intermediate", nviews=3, img_wh=(1920, 1056), ndepths=192, device="cuda:0", scans=None):
        super().__init__()
        self.datapath, self.split, self.nviews, self.img_wh, self.ndepths = datapath, split, nviews, img_wh, ndepths
        self.device = torch.device(device)
        self.image_sizes = dict(TANKS_SCANS[split])
        self.scans = list(self.image_sizes) if scans is None else list(scans)
        self.metas = []
        for scan in self.scans:
            for ref, srcs in scan_io.read_pair_file(os.path.join(datapath, split, scan, "pair.txt")):
                self.metas.append((scan, ref, srcs, scan))

    def __len__(self):
        return len(self.metas)

    @staticmethod
    def read_cam_file(filename):
        """-> intrinsics (first two rows / 4), extrinsics, depth_min, depth_max (datasets/tanks.py:66-80)."""
        lines = scan_io._cam_lines(filename)
        K, E = scan_io._matrix(lines[7:10], 3, 3), scan_io._matrix(lines[1:5], 4, 4)
        K[:2, :] /= 4.0
        tail = lines[11].split()
        return K, E, float(tail[0]), float(tail[1])

    def load_host(self, idx):
        scan, ref_view, src_views, _ = self.metas[idx]
        view_ids = [ref_view] + src_views[:self.nviews - 1]
        new_w, new_h = self.img_wh
        raws, projs, depth_values = [], [], None
        for i, vid in enumerate(view_ids):
            folder = os.path.join(self.datapath, self.split, scan)
            K, E, depth_min, depth_max = self.read_cam_file(os.path.join(folder, "cams_1/{:08d}_cam.txt".format(vid)))
            raw = np.array(Image.open(os.path.join(folder, "images/{:08d}.jpg".format(vid))), dtype=np.uint8)
            h0, w0 = raw.shape[:2]
            K[0, :] *= 1.0 * new_w / w0
            K[1, :] *= 1.0 * new_h / h0
            raws.append(raw)
            p = np.zeros((2, 4, 4), dtype=np.float32)
            p[0, :4, :4] = E
            p[1, :3, :3] = K
            projs.append(p)
            if i == 0:
                interval = (depth_max - depth_min) / (self.ndepths - 1)
                depth_values = np.arange(depth_min, interval * (self.ndepths - 0.5) + depth_min, interval, dtype=np.float32)
        return {"raw": raws, "size": (int(new_h), int(new_w)), "proj": np.stack(projs), "depth_values": depth_values,
                "filename": scan + "/{}/" + "{:0>8}".format(view_ids[0]) + "{}"}

    def to_device(self, host):
        return _finish_item(host, self.device)

    def __getitem__(self, idx):
        return self.to_device(self.load_host(idx))
