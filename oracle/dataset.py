"""Oracle: the evaluation loader's item assembly.

Test infrastructure (see oracle/__init__.py): numpy restatement of datasets/dtu_test.py:85-229 (read_cam_file, read_img,
scale_mvs_input, ToTensor + Normalize, the per-view loop of __getitem__ and the three-stage projection matrices).  Pinned by
tests/golden/dataset.npz, produced by tests/golden/make_golden.py --only-dataset from the reference's MVSDataset.

One step cannot be pinned by running the reference: ``cv2.resize`` (opencv-python 4.5.5.62, requirements.txt:31) is absent from
the reference tree and from this image; ``resize_linear`` restates its published float32 INTER_LINEAR algorithm
(modules/imgproc/src/resize.cpp, resizeGeneric_ with HResizeLinear / VResizeLinear: fx = (dx + 0.5) * scale - 0.5, floor,
border clamps with fx = 0, taps {1 - fx, fx}, horizontal pass then vertical), the golden generator hands it to the reference as
``cv2.resize``, and the restatement is pinned by KNOWN ANSWERS derived from that algorithm (exact halving, identity, border
replication and interior taps of an enlargement, a linear ramp at non-integer scales): tests/test_cv_known_answers_cpu.py.  torchvision's ToTensor / Normalize (also absent) are a
transpose and ``(x - mean) / std`` in float32.
"""
import numpy as np

MEAN = np.array([0.485, 0.456, 0.406], np.float32)
STD = np.array([0.229, 0.224, 0.225], np.float32)


def _taps(n_dst, n_src, horizontal):
    scale = 1.0 / (float(n_dst) / float(n_src))
    f = ((np.arange(n_dst, dtype=np.float64) + 0.5) * scale - 0.5).astype(np.float32)
    s = np.floor(f).astype(np.int64)
    f = (f - s.astype(np.float32)).astype(np.float32)
    if horizontal:
        f = np.where(s < 0, np.float32(0), f)
        s = np.maximum(s, 0)
        last = s >= n_src - 1
        i0 = np.where(last, n_src - 1, s)
        i1 = np.where(last, n_src - 1, s + 1)
        w0 = np.where(last, np.float32(1), np.float32(1) - f).astype(np.float32)
        w1 = np.where(last, np.float32(0), f).astype(np.float32)
        return i0, i1, w0, w1, last
    i0, i1 = np.clip(s, 0, n_src - 1), np.clip(s + 1, 0, n_src - 1)
    return i0, i1, (np.float32(1) - f).astype(np.float32), f, None


def resize_linear(img, dsize):
    """cv2.resize(img, (new_w, new_h)) (INTER_LINEAR) for float32 (H,W,C): horizontal pass, then vertical."""
    img = np.asarray(img, np.float32)
    new_w, new_h = int(dsize[0]), int(dsize[1])
    H, W = img.shape[:2]
    if (new_h, new_w) == (H, W):
        return img.copy()
    x0, x1, a0, a1, last = _taps(new_w, W, True)
    y0, y1, b0, b1, _ = _taps(new_h, H, False)
    a0, a1 = a0[None, :, None], a1[None, :, None]
    rows = np.where(last[None, :, None], img[:, x0] * np.float32(1), img[:, x0] * a0 + img[:, x1] * a1).astype(np.float32)
    return (rows[y0] * b0[:, None, None] + rows[y1] * b1[:, None, None]).astype(np.float32)


def scaled_size(h, w, max_h, max_w, base=32):
    """scale_mvs_input's target size (datasets/dtu_test.py:127-137), python float arithmetic kept."""
    if h > max_h or w > max_w:
        scale = 1.0 * max_h / h
        if scale * w > max_w:
            scale = 1.0 * max_w / w
        new_w, new_h = scale * w // base * base, scale * h // base * base
    else:
        new_w, new_h = 1.0 * w // base * base, 1.0 * h // base * base
    return new_h, new_w


def prepare_view(img_u8, K, max_h, max_w):
    """read_img + scale_mvs_input + ToTensor + Normalize for one view: -> (3,h,w) float32, intrinsics scaled in place."""
    img = img_u8.astype(np.float32) / 255.0
    h, w = img.shape[:2]
    new_h, new_w = scaled_size(h, w, max_h, max_w)
    K[0, :] *= 1.0 * new_w / w
    K[1, :] *= 1.0 * new_h / h
    img = resize_linear(img, (int(new_w), int(new_h)))
    t = np.ascontiguousarray(img.transpose(2, 0, 1))
    return ((t - MEAN[:, None, None]) / STD[:, None, None]).astype(np.float32)


def stage_matrices(proj):
    """proj (V,2,4,4) at 1/4 resolution -> dict of the three stages (datasets/dtu_test.py:211-220)."""
    out = {"stage1": proj}
    for key, mul in (("stage2", 2), ("stage3", 4)):
        p = proj.copy()
        p[:, 1, :2, :] = proj[:, 1, :2, :] * mul
        out[key] = p
    return out


def depth_values(depth_min, depth_interval, ndepths):
    return np.arange(depth_min, depth_interval * (ndepths - 0.5) + depth_min, depth_interval, dtype=np.float32)
