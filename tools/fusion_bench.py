"""Fusion filter at the reference's DTU evaluation shape (1184 x 1600 depth maps, 10 source views per reference view): ms per
reference view on the HIP path (fuse + compaction, maps resident) -- kernels only, e.g. under rocprofv3.  The CPU baseline
beside it is ``python bench.py --workload fusion``."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rc_mvsnet_amd import _lib, fusion, synthetic        # noqa: E402


def main():
    _lib.load()
    dev = "cuda:0"
    V, H, W, n_src = 11, 1184, 1600, 10
    s = synthetic.fusion_scan(V=V, H=H, W=W, seed=0, n_src=n_src)
    depth_all = torch.from_numpy(s["depth"]).to(dev)
    jobs = []
    for ref, srcs in s["pairs"]:
        mats = torch.from_numpy(fusion.fusion_matrices(s["K"][ref], s["E"][ref], [s["K"][i] for i in srcs], [s["E"][i] for i in srcs])).to(dev)
        jobs.append((ref, srcs, torch.from_numpy(s["conf"][ref]).to(dev), torch.from_numpy(s["img"][ref].astype(np.float32) / 255.0).to(dev), mats))

    def run():
        n = 0
        for ref, srcs, conf, img, mats in jobs:
            r = fusion.fuse_view(depth_all, ref, srcs, conf, img, mats, 0.8, 3, 0.5, 0.01)
            xyz, rgb = fusion.compact_points(r["masks"][2], r["xyz"], r["rgb"])
            n += len(xyz)
        return n

    run()
    torch.cuda.synchronize()
    t = time.perf_counter()
    reps = 5
    for _ in range(reps):
        n = run()
    torch.cuda.synchronize()
    hip_ms = (time.perf_counter() - t) * 1e3 / (reps * len(jobs))
    print(f"fusion, {n_src} source views, {H}x{W}: HIP {hip_ms:.3f} ms per reference view ({H * W * n_src / hip_ms / 1e6:.1f} G pixel-pairs/s, "
          f"{n // len(jobs)} points kept)")


if __name__ == "__main__":
    main()
