"""On-disk scan layout of the reference's evaluation: MVSNet-style camera files, pair files,
images and binary masks.  Same function names and return values as the helpers in eval_rcmvsnet_dtu.py:99-155 and
datasets/dtu_test.py:85-112, so the fusion driver (rc_mvsnet_amd/fusion.py) and a dataset loader read real DTU /
Tanks-and-Temples folders:

    <scan>/pair.txt                      "<n views>" then per view "<ref id>" and "<n src> <id> <score> <id> <score> ..."
    <scan>/cams/<view:08d>_cam.txt       "extrinsic" + 4 rows, blank, "intrinsic" + 3 rows, blank, "<depth_min> <interval> [<n> ...]"
    <scan>/images/<view:08d>.jpg
"""
import numpy as np
from PIL import Image


def _matrix(lines, rows, cols):
    # the reference parses with np.fromstring(..., dtype=float32, sep=' '): text -> double -> float32
    vals = np.array(" ".join(lines).split(), dtype=np.float64).astype(np.float32)
    if vals.size != rows * cols:
        raise ValueError(f"camera file: expected {rows * cols} numbers, found {vals.size}")
    return vals.reshape(rows, cols)


def _cam_lines(filename):
    with open(filename) as f:
        return [line.rstrip() for line in f.readlines()]


def read_camera_parameters(filename):
    """-> intrinsics (3,3) float32, extrinsics (4,4) float32 (eval_rcmvsnet_dtu.py:99-109; no 1/4 scaling)."""
    lines = _cam_lines(filename)
    return _matrix(lines[7:10], 3, 3), _matrix(lines[1:5], 4, 4)


def read_cam_file(filename, interval_scale=1.0, ndepths=192):
    """-> intrinsics (first two rows / 4), extrinsics, depth_min, depth_interval (datasets/dtu_test.py:85-105): when the last
    line carries a plane count the interval is rescaled so that ``ndepths`` planes span the same range."""
    lines = _cam_lines(filename)
    intrinsics, extrinsics = _matrix(lines[7:10], 3, 3), _matrix(lines[1:5], 4, 4)
    intrinsics[:2, :] /= 4.0
    tail = lines[11].split()
    depth_min, depth_interval = float(tail[0]), float(tail[1])
    if len(tail) >= 3:
        depth_max = depth_min + int(float(tail[2])) * depth_interval
        depth_interval = (depth_max - depth_min) / ndepths
    return intrinsics, extrinsics, depth_min, depth_interval * interval_scale


def write_cam(filename, cam):
    """cam (2,4,4): [0] = extrinsic, [1][:3,:3] = intrinsic, [1][3] = depth line (eval_rcmvsnet_dtu.py:139-155)."""
    with open(filename, "w") as f:
        f.write("extrinsic\n")
        for row in cam[0]:
            f.write("".join(str(v) + " " for v in row[:4]) + "\n")
        f.write("\nintrinsic\n")
        for row in cam[1][:3]:
            f.write("".join(str(v) + " " for v in row[:3]) + "\n")
        f.write("\n" + " ".join(str(v) for v in cam[1][3][:4]) + "\n")


def read_pair_file(filename):
    """-> [(ref_view, [src_view, ...]), ...], views without sources dropped (eval_rcmvsnet_dtu.py:126-137)."""
    pairs = []
    with open(filename) as f:
        n = int(f.readline())
        for _ in range(n):
            ref = int(f.readline().rstrip())
            srcs = [int(t) for t in f.readline().rstrip().split()[1::2]]
            if srcs:
                pairs.append((ref, srcs))
    return pairs


def read_img(filename):
    """-> float32 (H,W,3) in [0,1] (eval_rcmvsnet_dtu.py:112-116)."""
    return np.array(Image.open(filename), dtype=np.float32) / 255.0


def read_mask(filename):
    return read_img(filename) > 0.5


def save_mask(filename, mask):
    """Boolean (H,W) -> 8-bit image with 255 where set (eval_rcmvsnet_dtu.py:119-123)."""
    if mask.dtype != np.bool_:
        raise TypeError("save_mask: boolean mask expected")
    Image.fromarray(mask.astype(np.uint8) * 255).save(filename)
