"""Evaluation driver: the loop of the reference's ``save_scene_depth`` (eval_rcmvsnet_dtu.py:163-260) -- run
``CascadeMVSNet_eval`` over (scan, reference view) items and write ``<outdir>/<scan>/depth_est/<view:08d>.pfm`` and
``.../confidence/<view:08d>.pfm`` -- sharded one process per GPU with no collective.

The datasets are out of scope, so the items here are seeded synthetic DTU-shaped scenes; a loader
yielding the same ``(imgs, proj_matrices, depth_values)`` triple drops in through ``make_sample``.

    python -m rc_mvsnet_amd.eval_driver --outdir out --scans 4 --views 3 --height 512 --width 640
    python -m torch.distributed.run --nproc-per-node 8 --master-addr 127.0.0.1 -m rc_mvsnet_amd.eval_driver --outdir out ...
"""
import argparse
import os
import time

import torch

from . import synthetic
from .data_io import save_pfm
from .sharding import shard_items


def output_paths(outdir, scan, view):
    """The reference's ``filename.format('depth_est', '.pfm')`` layout (datasets/dtu_test.py:227: scan + '/{}/' + '{:0>8}' + '{}')."""
    name = "{:0>8}".format(view)
    return (os.path.join(outdir, scan, "depth_est", name + ".pfm"), os.path.join(outdir, scan, "confidence", name + ".pfm"))


def save_outputs(outdir, scan, view, depth, confidence):
    for path, t in zip(output_paths(outdir, scan, view), (depth, confidence)):
        os.makedirs(os.path.dirname(path), exist_ok=True)
        save_pfm(path, t.detach().float().cpu().numpy())


def run(model, items, make_sample, outdir, device):
    """items: [(scan, view)]; make_sample(scan, view) -> (imgs, proj_matrices, depth_values) on the CPU, batch 1."""
    times = []
    with torch.no_grad():
        for scan, view in items:
            imgs, proj, dv = make_sample(scan, view)
            imgs, dv = imgs.to(device), dv.to(device)
            proj = {k: v.to(device) for k, v in proj.items()}
            t0 = time.perf_counter()
            out = model(imgs, proj, dv)
            depth, conf = out["depth"][0], out["photometric_confidence"][0]
            if device.type == "cuda":
                torch.cuda.synchronize(device)
            times.append(time.perf_counter() - t0)
            save_outputs(outdir, scan, view, depth, conf)
    return times


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--outdir", required=True)
    ap.add_argument("--scans", type=int, default=2)
    ap.add_argument("--ref-views", type=int, default=2, help="reference views per scan")
    ap.add_argument("--views", type=int, default=3, help="images per item (1 reference + sources)")
    ap.add_argument("--height", type=int, default=512)
    ap.add_argument("--width", type=int, default=640)
    ap.add_argument("--ndepths", default="48,32,8")
    ap.add_argument("--depth_inter_r", default="4,2,1")
    ap.add_argument("--loadckpt", default=None, help="a reference checkpoint ({'model': state_dict}); seeded weights otherwise")
    args = ap.parse_args(argv)

    from .casmvsnet import CascadeMVSNet_eval
    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    device = torch.device("cuda", local) if torch.cuda.is_available() else torch.device("cpu")
    if device.type == "cuda":
        torch.cuda.set_device(device)
    model = CascadeMVSNet_eval(ndepths=[int(n) for n in args.ndepths.split(",")],
                               depth_interals_ratio=[float(r) for r in args.depth_inter_r.split(",")])
    sd = torch.load(args.loadckpt, map_location="cpu")["model"] if args.loadckpt else synthetic.cascade_state_dict(0)
    model.load_state_dict(sd, strict=True)
    model = model.to(device).eval()

    items = [("scan{}".format(s + 1), v) for s in range(args.scans) for v in range(args.ref_views)]
    mine = shard_items(items, rank, world)

    def make_sample(scan, view):
        seed = int(scan[4:]) * 1000 + view
        return synthetic.cascade_inputs(1, args.views, args.height, args.width, seed)

    times = run(model, mine, make_sample, args.outdir, device)
    if times:
        warm = times[1:] or times
        print(f"rank {rank}/{world}: {len(mine)} of {len(items)} items, {1.0 / (sum(warm) / len(warm)):.1f} ref-views/s "
              f"(model time, first item excluded), outputs under {args.outdir}")


if __name__ == "__main__":
    main()
