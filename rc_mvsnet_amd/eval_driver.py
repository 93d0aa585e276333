"""Evaluation driver: the loop of the reference's ``save_scene_depth`` (eval_rcmvsnet_dtu.py:163-260) -- run
``CascadeMVSNet_eval`` over (scan, reference view) items and write ``<outdir>/<scan>/depth_est/<view:08d>.pfm`` and
``.../confidence/<view:08d>.pfm`` -- sharded one process per GPU with no collective.

Two item sources: seeded synthetic DTU-shaped scenes (default), or real MVSNet-style scan folders through
``rc_mvsnet_amd.mvs_dataset.MVSDataset`` (``--testpath`` + ``--testlist``), in which case the reference view's camera and image
are written next to the depth maps as the reference does (eval_rcmvsnet_dtu.py:238-253) and ``--filter`` runs the fusion step
(``rc_mvsnet_amd.fusion.filter_depth``, the reference's step 2) on this rank's scans.  With real data the shard unit is the scan,
so that a rank owns every depth map its fusion needs.

    python -m rc_mvsnet_amd.eval_driver --outdir out --scans 4 --views 3 --height 512 --width 640
    python -m rc_mvsnet_amd.eval_driver --outdir out --testpath /data/dtu_test --testlist lists/dtu/test.txt --loadckpt model.ckpt --filter
    python -m rc_mvsnet_amd.eval_driver --gpus 8 --procs-per-gpu 2 --outdir out ...          (starts its own 16 ranks, two per GPU)
    python -m torch.distributed.run --nproc-per-node 8 --master-addr 127.0.0.1 -m rc_mvsnet_amd.eval_driver --outdir out ...
"""
import argparse
import os
import sys
import time

import torch

from . import synthetic
from .data_io import save_pfm
from .sharding import device_index, launch_ranks, launched, rank_env, shard_items


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


def save_reference_view(outdir, filename, cam, img):
    """cams/<view>_cam.txt and images/<view>.jpg of the reference view (eval_rcmvsnet_dtu.py:203-253): cam (2,4,4) at the last
    stage's scale, img (3,h,w) normalised -- de-normalised with the reference's constants (its blue std is 0.255, kept)."""
    import numpy as np
    from PIL import Image
    from .scan_io import write_cam
    cam_path = os.path.join(outdir, filename.format("cams", "_cam.txt"))
    img_path = os.path.join(outdir, filename.format("images", ".jpg"))
    for path in (cam_path, img_path):
        os.makedirs(os.path.dirname(path), exist_ok=True)
    write_cam(cam_path, cam)
    mean = torch.tensor([-0.485 / 0.229, -0.456 / 0.224, -0.406 / 0.255], device=img.device).view(3, 1, 1)
    std = torch.tensor([1 / 0.229, 1 / 0.224, 1 / 0.255], device=img.device).view(3, 1, 1)
    rgb = ((img - mean) / std).permute(1, 2, 0).mul(255).clamp(0, 255).to(torch.uint8).cpu().numpy()
    Image.fromarray(np.ascontiguousarray(rgb)).save(img_path, format="JPEG", quality=95)


def run_scans(model, args, device, rank, world):
    """Real data: this rank's scans through the loader, the model, the writers and (``--filter``) the fusion step."""
    from . import fusion
    from .mvs_dataset import AsyncWriter, MVSDataset, prefetch
    with open(args.testlist) as f:
        scans = [line.rstrip() for line in f.readlines() if line.strip()]
    mine = shard_items(scans, rank, world)
    nstage = len(args.ndepths.split(","))
    times = []
    for scan in mine:
        ds = MVSDataset(args.testpath, [scan], "test", args.num_view, args.numdepth, args.interval_scale, device=device,
                        max_h=args.max_h, max_w=args.max_w)
        # decoding runs ahead on worker threads, file writing trails on others: the GPU only waits for its own kernels
        with torch.no_grad(), AsyncWriter(args.io_threads) as writer:
            for item in prefetch(ds, workers=args.io_threads, depth=2 * args.io_threads):
                imgs = item["imgs"].unsqueeze(0)
                proj = {k: torch.from_numpy(v).unsqueeze(0).to(device) for k, v in item["proj_matrices"].items()}
                dv = torch.from_numpy(item["depth_values"]).unsqueeze(0).to(device)
                t0 = time.perf_counter()
                out = model(imgs, proj, dv)
                torch.cuda.synchronize(device)
                times.append(time.perf_counter() - t0)
                name = item["filename"]
                for kind, t in (("depth_est", out["depth"][0]), ("confidence", out["photometric_confidence"][0])):
                    path = os.path.join(args.outdir, name.format(kind, ".pfm"))
                    os.makedirs(os.path.dirname(path), exist_ok=True)
                    writer.submit(save_pfm, path, t.float().cpu().numpy())
                cam = item["proj_matrices"]["stage{}".format(nstage)][0]
                writer.submit(save_reference_view, args.outdir, name, cam, item["imgs"][0].cpu())
        if args.filter:
            folder = os.path.join(args.outdir, scan)
            fusion.filter_depth(os.path.join(args.testpath, scan), folder, folder, os.path.join(args.outdir, scan + ".ply"),
                                args.prob_thres, args.num_consistency, args.img_dist_thres, args.depth_thres,
                                num_stage=nstage, device=str(device))
    if times:
        warm = times[1:] or times
        print(f"rank {rank}/{world}: {len(mine)} of {len(scans)} scans, {len(times)} reference views, "
              f"{1.0 / (sum(warm) / len(warm)):.1f} ref-views/s (model time), outputs under {args.outdir}")


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--testpath", default=None, help="folder of MVSNet-style scans (real data instead of synthetic scenes)")
    ap.add_argument("--testlist", default=None, help="text file, one scan name per line")
    ap.add_argument("--num_view", type=int, default=5)
    ap.add_argument("--numdepth", type=int, default=192)
    ap.add_argument("--interval_scale", type=float, default=1.06)
    ap.add_argument("--max_h", type=int, default=1200)
    ap.add_argument("--max_w", type=int, default=1600)
    ap.add_argument("--io_threads", type=int, default=4, help="threads decoding input images ahead / writing outputs behind the GPU")
    ap.add_argument("--filter", action="store_true", help="fuse each scan's depth maps into <outdir>/<scan>.ply afterwards")
    ap.add_argument("--prob_thres", type=float, default=0.8)
    ap.add_argument("--num_consistency", type=int, default=3)
    ap.add_argument("--img_dist_thres", type=float, default=0.5)
    ap.add_argument("--depth_thres", type=float, default=0.01)
    ap.add_argument("--outdir", required=True)
    ap.add_argument("--scans", type=int, default=2)
    ap.add_argument("--ref-views", type=int, default=2, help="reference views per scan")
    ap.add_argument("--views", type=int, default=3, help="images per item (1 reference + sources)")
    ap.add_argument("--height", type=int, default=512)
    ap.add_argument("--width", type=int, default=640)
    ap.add_argument("--ndepths", default="48,32,8")
    ap.add_argument("--depth_inter_r", default="4,2,1")
    ap.add_argument("--loadckpt", default=None, help="a reference checkpoint ({'model': state_dict}); seeded weights otherwise")
    ap.add_argument("--gpus", type=int, default=1, help="GPUs of this node; above 1 rank (with --procs-per-gpu) the driver starts its own ranks "
                                                        "unless a launcher already did (WORLD_SIZE in the environment)")
    ap.add_argument("--procs-per-gpu", type=int, default=1, help="worker processes per GPU: items are independent, two processes per GPU overlap "
                                                                 "each other's latency-bound phases (rc_mvsnet_amd/sharding.py)")
    args = ap.parse_args(argv)
    nproc = args.gpus * args.procs_per_gpu
    if nproc > 1 and not launched():
        raise SystemExit(launch_ranks("rc_mvsnet_amd.eval_driver", nproc, sys.argv[1:] if argv is None else argv, module=True))

    from .casmvsnet import CascadeMVSNet_eval
    rank, local, world = rank_env()
    # Under an external launcher the launcher owns the rank layout: `torch.distributed.run --nproc-per-node 8 -m rc_mvsnet_amd.eval_driver --outdir out`
    # (no --gpus) and multi-node launches (WORLD_SIZE = nodes x local ranks) shard by (rank, world) as they are.  Only an
    # EXPLICIT --gpus / --procs-per-gpu is checked, and against the ranks of THIS node (LOCAL_WORLD_SIZE), not the job.
    explicit = any(a.split("=")[0] in ("--gpus", "--procs-per-gpu") for a in (sys.argv[1:] if argv is None else argv))
    local_world = int(os.environ.get("LOCAL_WORLD_SIZE", world))
    if launched() and explicit and local_world != nproc:
        raise SystemExit(f"eval_driver: --gpus {args.gpus} x --procs-per-gpu {args.procs_per_gpu} = {nproc} ranks per node, but the launcher started "
                         f"{local_world} (launch torch.distributed.run with --nproc-per-node {nproc}, drop the two flags, or drop the launcher and let the driver start the ranks)")
    device = torch.device("cpu")
    if torch.cuda.is_available():
        index = device_index(local, args.procs_per_gpu)
        if index >= torch.cuda.device_count():
            raise SystemExit(f"eval_driver: local rank {local} maps to GPU {index}, but this node shows {torch.cuda.device_count()} GPU(s)")
        device = torch.device("cuda", index)
        torch.cuda.set_device(device)
    model = CascadeMVSNet_eval(ndepths=[int(n) for n in args.ndepths.split(",")],
                               depth_interals_ratio=[float(r) for r in args.depth_inter_r.split(",")])
    sd = torch.load(args.loadckpt, map_location="cpu")["model"] if args.loadckpt else synthetic.cascade_state_dict(0)
    model.load_state_dict(sd, strict=True)
    model = model.to(device).eval()

    if args.testpath:
        if device.type != "cuda":
            raise SystemExit("eval_driver: real data needs a GPU (the loader's image preparation has no CPU fallback)")
        return run_scans(model, args, device, rank, world)

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
