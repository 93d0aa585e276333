"""PFM depth / confidence map files, the on-disk format the reference's evaluation writes and its fusion step reads
(datasets/data_io.py:8-68, eval_rcmvsnet_dtu.py:212-260).

Format (Portable Float Map): three text lines -- ``Pf`` (one channel) or ``PF`` (three), ``<width> <height>``, a scale
whose SIGN gives the byte order (negative = little-endian; the reference prints it with ``%f``) -- followed by the rows
as raw float32, BOTTOM row first.  Byte-for-byte compatible with the reference's writer (tests/golden/pfm.npz).
"""
import re
import sys

import numpy as np


def save_pfm(filename, image, scale=1):
    """image: float32 (H,W), (H,W,1) or (H,W,3)."""
    image = np.asarray(image)
    if image.dtype != np.float32:
        raise ValueError("save_pfm: image dtype must be float32")
    if image.ndim == 3 and image.shape[2] == 3:
        color = True
    elif image.ndim == 2 or (image.ndim == 3 and image.shape[2] == 1):
        color = False
    else:
        raise ValueError("save_pfm: image must be H x W x 3, H x W x 1 or H x W")
    little = image.dtype.byteorder == "<" or (image.dtype.byteorder == "=" and sys.byteorder == "little")
    with open(filename, "wb") as f:
        f.write(b"PF\n" if color else b"Pf\n")
        f.write(f"{image.shape[1]} {image.shape[0]}\n".encode("utf-8"))
        f.write(("%f\n" % (-scale if little else scale)).encode("utf-8"))
        np.ascontiguousarray(image[::-1]).tofile(f)          # bottom row first


def read_pfm(filename):
    """-> (array (H,W) or (H,W,3) float32 with the top row first, scale)."""
    with open(filename, "rb") as f:
        header = f.readline().decode("utf-8").rstrip()
        if header not in ("PF", "Pf"):
            raise ValueError(f"{filename}: not a PFM file")
        m = re.match(r"^(\d+)\s(\d+)\s$", f.readline().decode("utf-8"))
        if not m:
            raise ValueError(f"{filename}: malformed PFM header")
        width, height = int(m.group(1)), int(m.group(2))
        scale = float(f.readline().rstrip())
        endian = "<" if scale < 0 else ">"
        data = np.fromfile(f, endian + "f")
    shape = (height, width, 3) if header == "PF" else (height, width)
    if data.size != int(np.prod(shape)):
        raise ValueError(f"{filename}: expected {int(np.prod(shape))} floats, found {data.size}")
    return np.flipud(data.reshape(shape)).astype(np.float32), abs(scale)
