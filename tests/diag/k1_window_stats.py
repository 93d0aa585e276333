#!/usr/bin/env python
"""CPU diagnostic for the window-staged K1 design ("next"): how large is the source-map window that a block
of reference pixels x a chunk of hypothesis planes samples, relative to the bytes its gathers move today?

Uses the oracle's coordinate chain (oracle/warp.py) on the bench's own scene (config 2, seed 0, random-init weights: the plane
tables of stages 2 and 3 come from the previous stage's -- noisy -- depth map) and, for contrast, on plane tables centred on a
smooth depth map (what a trained network produces).  For every (stage, tile shape, chunk depth) it reports the median and 90th
percentile of   window texels / (tile pixels x planes x 4 taps)   over all tiles and source views: the fraction of today's
gather traffic that staging the window once would move instead.
"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle import cascade, warp                         # noqa: E402
from rc_mvsnet_amd import synthetic                      # noqa: E402

H, W, V = 512, 640, 3
NDEPTHS, RATIOS = (48, 32, 8), (4, 2, 1)
TILES = ((4, 64), (4, 32), (8, 32), (8, 16), (16, 16))
CHUNKS = (2, 4, 8)


def window_ratio(ix, iy, th, tw, dk):
    """ix, iy (D,h,w) sampling positions in one source view -> array of window/gather ratios over tiles and plane chunks."""
    D, h, w = ix.shape
    x0 = np.clip(np.floor(ix), -1, w - 1).astype(np.int32)
    y0 = np.clip(np.floor(iy), -1, h - 1).astype(np.int32)
    out = []
    for k0 in range(0, D - dk + 1, dk):
        xa, ya = x0[k0:k0 + dk], y0[k0:k0 + dk]
        for ty in range(0, h - th + 1, th):
            xs, ys = xa[:, ty:ty + th], ya[:, ty:ty + th]
            nx = w // tw
            xs = xs[:, :, :nx * tw].reshape(dk, th, nx, tw)
            ys = ys[:, :, :nx * tw].reshape(dk, th, nx, tw)
            wx = xs.max((0, 1, 3)) - xs.min((0, 1, 3)) + 2
            wy = ys.max((0, 1, 3)) - ys.min((0, 1, 3)) + 2
            out.append(wx * wy / float(th * tw * dk * 4))
    return np.concatenate(out)


def main():
    torch.set_num_threads(min(16, os.cpu_count() or 1))
    sd = synthetic.cascade_state_dict(0)
    imgs, pm, dv = synthetic.cascade_inputs(1, V, H, W, 0)
    with torch.no_grad():
        _, aux = cascade.forward_eval(imgs, pm, dv, sd, NDEPTHS, RATIOS, impl="aten", return_aux=True)
    yy, xx = torch.meshgrid(torch.linspace(0, 1, H), torch.linspace(0, 1, W), indexing="ij")
    smooth = (650.0 + 90.0 * torch.sin(3.0 * xx) * torch.cos(2.0 * yy)).unsqueeze(0)
    lines = ["window texels / (tile pixels x planes x 4 taps): median (p90) over tiles, chunks and the 2 source views; config 2, seed 0",
             "%-28s %-8s " % ("plane tables", "tile") + " ".join("DK=%d          " % c for c in CHUNKS)]
    for label in ("bench (random weights)", "smooth depth"):
        prev = None
        for s in range(3):
            key = "stage%d" % (s + 1)
            sc = 4 >> s
            h, w = H // sc, W // sc
            if label.startswith("bench"):
                samples = aux[key]["samples"]
            else:
                samples = warp.stage_samples(prev, dv, NDEPTHS[s], RATIOS[s], (H, W), (h, w))
                prev = smooth
            coords = []
            for v in range(1, V):
                rot, trans = warp.compose_homography(pm[key][:, v], pm[key][:, 0])
                ix, iy = warp.warp_coords(rot, trans, samples, h, w)
                coords.append((ix[0].numpy(), iy[0].numpy()))
            for th, tw in TILES:
                cells = []
                for dk in CHUNKS:
                    if dk > NDEPTHS[s]:
                        cells.append("-")
                        continue
                    r = np.concatenate([window_ratio(ix, iy, th, tw, dk) for ix, iy in coords])
                    cells.append("%.3f (%.3f)" % (np.median(r), np.percentile(r, 90)))
                lines.append("%-28s %-8s " % ("%s S%d C=%d" % (label[:14], s + 1, (32, 16, 8)[s]), "%dx%d" % (th, tw)) + " ".join("%-14s" % c for c in cells))
    text = "\n".join(lines)
    print(text)
    if len(sys.argv) > 1:
        open(sys.argv[1], "w").write(text + "\n")


if __name__ == "__main__":
    main()
