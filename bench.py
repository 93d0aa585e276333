#!/usr/bin/env python
"""bench.py -- ref-scenes/s of the MI355X plane-sweep hot path (BASELINE.json metric).

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A "step" is one CascadeMVSNet_eval.forward at batch 1 = one reference view ("ref-scene") of
BASELINE config 2: DTU-shaped, 3 views, 512x640, D = 48/32/8, fp32, synthetic seeded inputs and
weights already resident in HBM.  Reference views are independent, so N GPUs run N independent
streams of scenes with no data-path collective ("weak" scaling): value = N*K / max-over-ranks time.

The JSON line also carries
  roofline      the fused warp+variance kernel (K1): algorithmic bytes of its three per-scene
                launches (137.6 + 194.0 + 125.8 MB) / their HIP-event durations
                recorded on the launch stream inside the timed region, vs the 8 TB/s HBM peak;
  cpu_baseline  the oracle's ATen op graph (= the reference's CPU path) timed on the host cores
                of this box on a bounded sample, rank 0, N=1 only -- a reported baseline, plus
                the depth-L1 parity of the HIP output against it on the same inputs.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import torch  # noqa: E402

HBM_PEAK_GBS = 8000.0           # MI355X_MICROARCH.md: 8 TB/s spec (~6.3 TB/s achievable)
H, W, V = 512, 640, 3
NDEPTHS, RATIOS = (48, 32, 8), (4, 2, 1)
FEAT_C = (32, 16, 8)


def host_threads():
    """Threads for the CPU baseline: the cores this process may really use (affinity mask and
    cgroup CPU quota), capped at 32 -- PyTorch-CPU does not scale past that on this workload and
    oversubscribing a quota-limited container is pathological."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except Exception:
        pass
    return max(1, min(n, 32))


def k1_algorithmic_bytes():
    """read (V-1) source maps + the reference map + the plane table, write the variance volume once
    (planes counted as the reference does: one (D,h,w) fp32 tensor)."""
    per_stage = []
    for s, (D, C) in enumerate(zip(NDEPTHS, FEAT_C)):
        sc = 4 >> s
        h, w = H // sc, W // sc
        per_stage.append(4 * ((V - 1) * C * h * w + C * h * w + D * h * w + C * D * h * w))
    return per_stage


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-scenes", type=int, default=3, help="scenes timed on the CPU baseline (bounded sample)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    assert torch.cuda.is_available(), "bench.py needs a GPU (the HIP path has no CPU fallback)"
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        import torch.distributed as dist
        dist.init_process_group(backend="nccl")          # RCCL; used only for the barrier / max-time reduction

    from rc_mvsnet_amd import _lib, ops, synthetic
    from rc_mvsnet_amd.casmvsnet import CascadeMVSNet_eval
    _lib.load()

    sd = synthetic.cascade_state_dict(0)
    model = CascadeMVSNet_eval(ndepths=list(NDEPTHS), depth_interals_ratio=list(RATIOS))
    model.load_state_dict(sd, strict=True)
    model = model.to(dev).eval()

    # a few distinct scenes resident in HBM; rank r starts at a different one
    scenes = []
    for seed in range(4):
        imgs, pm, dv = synthetic.cascade_inputs(1, V, H, W, seed)
        scenes.append((imgs.to(dev), {k: v.to(dev) for k, v in pm.items()}, dv.to(dev)))

    def step(i):
        imgs, pm, dv = scenes[(i + rank) % len(scenes)]
        return model(imgs, pm, dv)

    def barrier():
        if world > 1:
            dist.barrier()

    with torch.no_grad():
        for i in range(args.warmup):
            out = step(i)
        torch.cuda.synchronize()
        barrier()
        ops.K1_EVENTS = []                                  # HIP events around every K1 launch
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(args.steps):
            out = step(i)
        torch.cuda.synchronize()
        barrier()
        elapsed = time.perf_counter() - t0
    events, ops.K1_EVENTS = ops.K1_EVENTS, None

    if world > 1:
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        dist.barrier()
        dist.destroy_process_group()                         # before rank 0 spends ~25 s on the CPU baseline alone

    if rank != 0:
        return

    # ---- roofline of K1 from the events recorded inside the timed region -------------------
    k1_ms = [e0.elapsed_time(e1) for (e0, e1) in events]
    nstage = len(NDEPTHS)
    per_stage_ms = [sum(k1_ms[s::nstage]) / max(1, len(k1_ms[s::nstage])) for s in range(nstage)]
    bytes_stage = k1_algorithmic_bytes()
    tot_ms = sum(per_stage_ms)
    achieved = sum(bytes_stage) / (tot_ms * 1e-3) / 1e9 if tot_ms > 0 else 0.0
    traffic = None                      # HBM bytes per scene from the committed PMC profile (FETCH_SIZE x2 + WRITE_SIZE)
    try:
        traffic = json.load(open(os.path.join(ROOT, "profiles", "r1_k1_traffic.json")))["bytes_per_scene"]
    except Exception:
        pass
    roofline = {"bound": "hbm", "kernel": "rcmvs::warp_variance_tp_kernel (K1, 3 launches per scene)",
                "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic,
                "traffic_note": "bytes per scene (sum of the 3 launches), rocprofv3 PMC passes of profiles/r1_k1_traffic.json",
                "algorithmic_bytes_per_scene": sum(bytes_stage),
                "per_stage_us": [round(m * 1e3, 2) for m in per_stage_ms],
                "per_stage_GBs": [round(b / (m * 1e-3) / 1e9, 1) if m > 0 else 0.0 for b, m in zip(bytes_stage, per_stage_ms)]}

    result = {
        "metric": "ref-scenes/sec (DTU 3-view 512x640, D=48/32/8)",
        "value": round(world * args.steps / elapsed, 3),
        "unit": "ref-scenes/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": round(elapsed / args.steps * 1e3, 4),
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "f32",
        "data": "synthetic",
        "config": {"workload": "BASELINE configs[1]: CascadeMVSNet_eval.forward, DTU-shaped 3 views 512x640, "
                               "D=(48,32,8), batch 1 per GPU, fp32, random-init seeded weights",
                   "views": V, "height": H, "width": W, "ndepths": list(NDEPTHS), "parallelism": f"scene-per-gpu x{world}"},
        "roofline": roofline,
    }

    # ---- CPU baseline (oracle, ATen op graph of the reference) + parity on the same inputs -----
    if world == 1 and not args.no_cpu_baseline:
        from oracle import cascade
        nthreads = host_threads()
        torch.set_num_threads(nthreads)
        imgs, pm, dv = synthetic.cascade_inputs(1, V, H, W, 0)
        times = []
        with torch.no_grad():
            budget_t0 = time.perf_counter()
            for i in range(1 + args.cpu_scenes):              # first pass = warm-up unless it is all we can afford
                c0 = time.perf_counter()
                ref = cascade.forward_eval(imgs, pm, dv, sd, NDEPTHS, RATIOS, impl="aten")
                times.append(time.perf_counter() - c0)
                if time.perf_counter() - budget_t0 > 25.0:    # bounded sample: ~10-30 s of CPU work
                    break
            hip = model(*scenes[0])
        timed = times[1:] if len(times) > 1 else times
        cpu_s = sorted(timed)[len(timed) // 2]
        rng = float(dv[0, -1] - dv[0, 0])
        dd = (hip["depth"].cpu() - ref["depth"]).abs()
        result["cpu_baseline"] = {"value": round(1.0 / cpu_s, 4), "unit": "ref-scenes/s", "cores": nthreads, "kind": "port",
                                  "sample": f"median of {len(timed)} scene(s) of the same config-2 workload"
                                            f"{' after 1 warm-up' if len(times) > 1 else ' (cold, no warm-up fit the time bound)'}, "
                                            f"oracle impl='aten' (reference op graph on PyTorch-CPU), {nthreads} threads"}
        result["parity"] = {"depth_l1_over_range": float(dd.mean()) / rng, "depth_l1_mm": float(dd.mean()),
                            "depth_max_abs_mm": float(dd.max()), "frac_pixels_over_0.1mm": float((dd > 0.1).float().mean()),
                            "tolerance": 1e-4}
    print(json.dumps(result))


if __name__ == "__main__":
    main()
