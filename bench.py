#!/usr/bin/env python
"""bench.py -- ref-scenes/s of the MI355X plane-sweep hot path (BASELINE.json metric).

    python bench.py [--gpus N] [--steps K] [--warmup W]                    (N > 1 without WORLD_SIZE: bench.py starts its own N ranks)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A "step" is one CascadeMVSNet_eval.forward at batch 1 = one reference view ("ref-scene") of
BASELINE config 2: DTU-shaped, 3 views, 512x640, D = 48/32/8, fp32, synthetic seeded inputs and
weights already resident in HBM.  Reference views are independent, so N GPUs run N independent
streams of scenes with no data-path collective ("weak" scaling): value = N*K / max-over-ranks time.

The JSON line also carries
  roofline      the fused warp+variance kernel (K1): algorithmic bytes of its three per-scene
                launches (137.6 + 194.0 + 125.8 MB) / their HIP-event durations
                recorded on the launch stream in a separate pass over the same scenes right after the timed region
                (6 event records per scene would sit inside `value` otherwise), vs the 8 TB/s HBM peak;
                `roofline.smooth_scene` = the same kernel on the same scene with prob.weight x1 (depth maps of stages 2 / 3
                that are not noise: the gathers a trained network produces);
  roofline_conv the 3-D convolutions that run on the matrix cores at fp32 accuracy (csrc/conv3d_x3.hip: operands split into two
                fp16 pieces after an exact power-of-two pre-scale -- the default for a B = 1 scene -- or into three bf16 pieces):
                algorithmic flops of every such launch of a scene / HIP-event durations from a separate untimed pass, vs the
                fp32 dense peak (157 TF; the three / six MFMAs per product are priced against the matrix-pipe peak as
                `matrix_pipe_frac`);
  train_step    a short timing of the config-3 training iteration (5 iterations after 2 warm-ups; --no-train-step skips it);
  two_procs_per_gpu      N = 1 only, a side pass after the timed region, never `value`: `bench.py --procs-per-gpu 2` run as a child -- two worker
                processes on the GPU, whole scenes each (--no-side-pass skips it);
  timed_rounds  when K steps take less than 0.5 s the barrier-bracketed region of exactly K steps is repeated and the MEDIAN round is
                reported (`value`, `ms_per_step`, `timed_region_s`), so that a short --steps still keeps the GPU busy for half a second;
  cpu_baseline  the oracle's ATen op graph (= the reference's CPU path) timed on the host cores
                of this box on a bounded sample, rank 0, N=1 only -- a reported baseline, plus
                the depth-L1 parity of the HIP output against it on the same inputs.

``--workload train_step`` times one training iteration of BASELINE config 3 (train_rcmvsnet.py's call sequence: two
CascadeMVSNet passes over 4 views at 512x640, D = 48/32/8, the rendering-consistency branch on 1024 rays x 128 samples, the
reference's losses, one backward, Adam) -- with N > 1 ranks as data-parallel training (config 4): SyncBatchNorm-converted
models, gradients of both models averaged with ONE reduce-scatter + all-gather message over RCCL (parallel.GradSync).
``--workload unsup_loss`` / ``--workload fusion`` time the two callers either side of the path that SURVEY.md section 8f ranks
next (the self-supervised loss of the training step, the depth-map fusion filter of the evaluation) with the same contract and
their own ``roofline`` / ``cpu_baseline`` objects; the default workload, and the only one BASELINE.json's metric is quoted
on, is the cascade.
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
SCENE_ALGORITHMIC_BYTES = 457441280 + 1870000000 + 49900000 + 300000000      # K1, 3-D CNN (ideal activations), depth head, FeatureNet
SCENE_PARTS_MB = {"k1": 457.4, "cnn3d_ideal_activations": 1870.0, "depth_head": 49.9, "feature_net": 300.0}

# --shape: the cascade workload at the shapes the reference's own workflows run (the default is BASELINE configs[1], the one `metric` is quoted on)
SHAPES = {
    "dtu_bench": {"V": 3, "H": 512, "W": 640, "ndepths": (48, 32, 8), "steps": 600,
                  "workload": "BASELINE configs[1]: CascadeMVSNet_eval.forward, DTU-shaped 3 views 512x640, D=(48,32,8), batch 1 per GPU, fp32, random-init seeded weights",
                  "metric": "ref-scenes/sec (DTU 3-view 512x640, D=48/32/8)"},
    "dtu_eval": {"V": 5, "H": 1184, "W": 1600, "ndepths": (48, 32, 8), "steps": 60,
                 "workload": "the reference's DTU evaluation shape (eval_rcmvsnet_dtu.py:49-51: 5 views, 1184x1600, D=(48,32,8)): CascadeMVSNet_eval.forward, "
                             "batch 1 per GPU, fp32, random-init seeded weights",
                 "metric": "ref-scenes/sec (DTU evaluation shape, 5-view 1184x1600, D=48/32/8)"},
    "tanks": {"V": 7, "H": 1056, "W": 1920, "ndepths": (64, 32, 8), "steps": 40,
              "workload": "BASELINE configs[4], one GPU's share (eval_rcmvsnet_tanks.py:47,53-55: Tanks and Temples intermediate, 7 views, 1056x1920, D=(64,32,8)): "
                          "CascadeMVSNet_eval.forward, batch 1 per GPU, fp32, random-init seeded weights",
              "metric": "ref-scenes/sec (Tanks and Temples shape, 7-view 1056x1920, D=64/32/8)"},
}


def set_shape(name):
    """Switch the module's workload constants to SHAPES[name]; the scene's algorithmic bytes follow SURVEY.md 8d's per-layer formulas."""
    global H, W, V, NDEPTHS, SCENE_ALGORITHMIC_BYTES, SCENE_PARTS_MB
    sh = SHAPES[name]
    H, W, V, NDEPTHS = sh["H"], sh["W"], sh["V"], tuple(sh["ndepths"])
    if name == "dtu_bench":
        return
    k1 = sum(k1_algorithmic_bytes())
    cnn = head = 0
    for s_, (D, C) in enumerate(zip(NDEPTHS, FEAT_C)):
        n = D * (H // (4 >> s_)) * (W // (4 >> s_))
        # every layer of the 3-D U-Net reads its input once and writes its output once (the transposed layers also read their skip tensor),
        # channels 8 / 16 / 32 / 64 at 1 / 1/8 / 1/64 / 1/512 of the voxels; + the prob conv's read of the 8-channel volume
        cnn += 4 * (n * (C + 8) + n * 10 + n * 4 + n * 2.5 + n * 1 + n * 0.625 + n * 0.25 + n * 1.125 + n * 4.5 + n * 18 + n * 8)
        head += 4 * 2 * n
    fnet = int(300e6 * (V * H * W) / (3 * 512 * 640))
    SCENE_ALGORITHMIC_BYTES = int(k1 + cnn + head + fnet)
    SCENE_PARTS_MB = {"k1": round(k1 / 1e6, 1), "cnn3d_ideal_activations": round(cnn / 1e6, 1), "depth_head": round(head / 1e6, 1), "feature_net": round(fnet / 1e6, 1)}


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


# layer shapes served by the split-bf16 matrix-core kernels (csrc/conv3d_x3.hip): (kind, Ci, Co); one-plane stride-1 volumes use
# the planar form (9 taps)
X3_LAYERS = {("s1", 8, 8), ("s1", 16, 8), ("s1", 32, 8), ("s1", 16, 16), ("s1", 32, 32), ("s2", 8, 16), ("s2", 16, 32), ("t2", 16, 8), ("t2", 32, 16)}
FP32_PEAK_TFLOPS = 157.3           # MI355X_MICROARCH: fp32 vector = fp32 MFMA dense peak
BF16_PEAK_TFLOPS = 2500.0          # dense bf16 MFMA peak


def conv_roofline(conv_events, nscenes):
    """Second roofline object: the 3-D convolutions that run on the bf16 matrix cores at fp32 accuracy.  `achieved` counts the
    layer's ALGORITHMIC flops (2 x taps x Ci x Co per output voxel / input cell: what an fp32 convolution does) against the
    fp32 dense peak -- the precision class the results are in; `matrix_pipe_frac` prices the MFMAs actually issued per product
    (three on the fp16-pair form the 3-D layers of a B = 1 scene take by default, six on the exact bf16 triple; block-Toeplitz
    padding not counted) against the dense bf16 / fp16 matrix-pipe peak."""
    if not conv_events or nscenes <= 0:
        return None
    from rc_mvsnet_amd import casmvsnet
    pair = casmvsnet.FP16_PAIR_DEFAULT if os.environ.get("RCMVS_FP16_PAIR") is None else os.environ["RCMVS_FP16_PAIR"] == "1"
    PAIR_LAYERS = {("s1", 8, 8), ("s1", 16, 8), ("s1", 32, 8), ("s1", 16, 16), ("s2", 8, 16), ("s2", 16, 32), ("t2", 16, 8), ("s1", 32, 32), ("t2", 32, 16)}
    flops = ms = products = act_bytes = 0.0
    per_layer = {}
    for e0, e1, (kind, B, D, H, W, Ci, Co) in conv_events:
        if (kind, Ci, Co) not in X3_LAYERS:
            continue
        taps = 9 if (kind == "s1" and D == 1) else 27
        if kind == "t2":
            cells = B * D * H * W                                  # every input cell meets all 27 taps once over its 8 outputs
        elif kind == "s2":
            cells = B * ((D - 1) // 2 + 1) * ((H - 1) // 2 + 1) * ((W - 1) // 2 + 1)
        else:
            cells = B * D * H * W
        f = 2.0 * taps * Ci * Co * cells
        t = e0.elapsed_time(e1)
        # ideal activation traffic of the layer: read the input once, write the output once, and -- the transposed layers of CostRegNet add
        # their skip connection in the epilogue (models/modules.py:496-498) -- read the skip tensor once
        vin = B * D * H * W
        vout = 8 * vin if kind == "t2" else cells
        act_bytes += 4.0 * (vin * Ci + vout * Co * (2 if kind == "t2" else 1))
        flops += f
        products += f * (3.0 if (pair and B == 1 and taps == 27 and (kind, Ci, Co) in PAIR_LAYERS) else 6.0)
        ms += t
        k = f"{kind} {Ci}->{Co} {D}x{H}x{W}" + (f" x{B}" if B > 1 else "")
        a = per_layer.setdefault(k, [0.0, 0.0])
        a[0] += f
        a[1] += t
    if ms <= 0:
        return None
    tf = flops / (ms * 1e-3) / 1e12
    top = sorted(per_layer.items(), key=lambda kv: -kv[1][1])[:6]
    return {"bound": "mfma", "kernel": "rcmvs::conv3d_x3_kernel family (split-operand matrix-core 3-D / planar convolutions, all launches of a scene)",
            "achieved": round(tf, 1), "peak": FP32_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": round(tf / FP32_PEAK_TFLOPS, 4),
            "peak_note": "fp32 dense peak: the results are fp32-accurate (3-D layers: two fp16 pieces per operand after an exact power-of-two "
                         "pre-scale, 3 MFMAs per product" + ("" if pair else " -- disabled by RCMVS_FP16_PAIR=0") + "; planar layers and the exact form: three bf16 pieces, 6 MFMAs)",
            "arithmetic": "fp16 pair (default)" if pair else "exact bf16 triple (RCMVS_FP16_PAIR=0)",
            "matrix_pipe_frac": round(products / (ms * 1e-3) / 1e12 / BF16_PEAK_TFLOPS, 4),
            # once the arithmetic is on the 2.5 PF pipe the layers' bound is their activation traffic: ideal bytes / time against the 8 TB/s HBM peak
            "frac_hbm": round(act_bytes / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4), "activation_GB_per_scene": round(act_bytes / nscenes / 1e9, 3),
            "hbm_bound_us_per_scene": round(act_bytes / nscenes / (HBM_PEAK_GBS * 1e9) * 1e6, 1),
            "us_per_scene": round(ms * 1e3 / nscenes, 1),
            "gflop_per_scene": round(flops / nscenes / 1e9, 2),
            "largest_layers_us_tflops": {k: [round(v[1] * 1e3 / nscenes, 1), round(v[0] / (v[1] * 1e-3) / 1e12, 1)] for k, v in top},
            "timing": "HIP events on the launch stream around every launch, separate untimed pass of 10 scenes"}


def k1_algorithmic_bytes():
    """read (V-1) source maps + the reference map + the plane table, write the variance volume once
    (planes counted as the reference does: one (D,h,w) fp32 tensor)."""
    per_stage = []
    for s, (D, C) in enumerate(zip(NDEPTHS, FEAT_C)):
        sc = 4 >> s
        h, w = H // sc, W // sc
        per_stage.append(4 * ((V - 1) * C * h * w + C * h * w + D * h * w + C * D * h * w))
    return per_stage


MIN_TIMED_S = 0.5               # a timed region shorter than this is repeated in rounds (each exactly K steps) and the median round reported
MIN_TIMED_S_SHORT = 6.0         # ... and with --steps < 100 (the driver's --steps 20 is ~23 ms of cascade forwards) rounds are repeated for 6 s, so that a
                                # utilisation sampler with a 5-second period sees a busy GPU (BENCH_r05.gpu_busy had 0 of 5 samples active)
MAX_ROUNDS = 4000


def timed_rounds(world, dev, steps, step, sync):
    """The contract's timed region -- exactly K steps between barrier + synchronize pairs, max over ranks -- repeated while the
    rounds so far add up to less than MIN_TIMED_S (the driver's --steps 20 is 28 ms of cascade forwards: too short for an external
    utilisation sampler to see the GPU busy).  Returns (median round's max-over-ranks seconds, this rank's own seconds of that
    round, every round's max-over-ranks seconds).  The process group stays up."""
    import torch.distributed as dist
    rounds, owns = [], []
    while True:
        sync()
        if world > 1:
            dist.barrier()
        sync()
        t0 = time.perf_counter()
        for i in range(steps):
            step(i)
        sync()
        own = time.perf_counter() - t0
        if world > 1:
            dist.barrier()
        elapsed = time.perf_counter() - t0
        if world > 1:
            t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)          # every rank sees the same round time, hence takes the same decision below
            elapsed = float(t.item())
        rounds.append(elapsed)
        owns.append(own)
        if sum(rounds) >= (MIN_TIMED_S_SHORT if (steps < 100 and dev.type == "cuda") else MIN_TIMED_S) or len(rounds) >= MAX_ROUNDS:
            break
    order = sorted(range(len(rounds)), key=lambda i: rounds[i])
    mid = order[len(order) // 2]
    return rounds[mid], owns[mid], rounds


def rounds_fields(rounds):
    return {"timed_rounds": len(rounds), "rounds_total_s": round(sum(rounds), 4),
            "round_s_min_max": [round(min(rounds), 5), round(max(rounds), 5)]}


def timed_region(world, dev, warmup, steps, step):
    """W untimed steps, then timed_rounds; tears the process group down.  Returns (median round seconds, all rounds)."""
    import torch.distributed as dist
    for i in range(warmup):
        step(i)
    sync = torch.cuda.synchronize if dev.type == "cuda" else (lambda: None)
    elapsed, _, rounds = timed_rounds(world, dev, steps, step, sync)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    return elapsed, rounds


def two_procs_side_pass(steps):
    """`python bench.py --gpus 1 --procs-per-gpu 2` as a child of the N = 1 run (its two ranks rendezvous over gloo on 127.0.0.1)."""
    import subprocess
    from rc_mvsnet_amd.sharding import clean_env
    env = clean_env()
    cmd = [sys.executable, os.path.abspath(__file__), "--gpus", "1", "--procs-per-gpu", "2", "--steps", str(steps), "--warmup", "10",
           "--no-cpu-baseline", "--no-train-step", "--no-side-pass"]
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    if out.returncode != 0 or not lines:
        return {"error": f"child exited with {out.returncode}: {out.stderr[-300:]}"}
    b = json.loads(lines[-1])
    return {"value": b["value"], "unit": b["unit"], "ms_per_step": b["ms_per_step"], "steps_per_process": b["steps"], "processes": 2,
            "per_rank_scenes_per_s": b.get("per_rank_scenes_per_s"), "timed_rounds": b.get("timed_rounds"),
            "note": "side pass, NOT `value`: two worker processes on the one GPU, each running whole scenes one at a time (python bench.py --procs-per-gpu 2)"}


def event_ms(fn, reps=20):
    """Average duration of ``fn`` (library launches on torch's current stream) from HIP events on that stream."""
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


def bench_unsup_loss(args, rank, world, dev):
    """SURVEY.md 8f-2.  Step = UnsupLossMultiStage forward + backward over the three stage depth maps of one sample of
    BASELINE config 3's shape (batch 1, 4 views, 512x640); images / cameras / depth maps resident in HBM."""
    from rc_mvsnet_amd import losses, synthetic
    B, Vl = 1, 4
    imgs, cams = synthetic.images(B, Vl, H, W, rank), synthetic.proj_matrices(B, Vl, H, W)
    dep = {}
    for i, sc in enumerate((4, 2, 1)):
        h, w = H // sc, W // sc
        yy, xx = torch.meshgrid(torch.linspace(0, 1, h), torch.linspace(0, 1, w), indexing="ij")
        d = 620.0 + 110.0 * torch.sin(4.0 * xx + i) * torch.cos(3.0 * yy) + torch.randn(h, w, generator=torch.Generator().manual_seed(i))
        dep["stage%d" % (i + 1)] = d.unsqueeze(0).repeat(B, 1, 1)
    gi, gc = imgs.to(dev), {k: v.to(dev) for k, v in cams.items()}
    gd = {k: v.to(dev) for k, v in dep.items()}
    mod = losses.UnsupLossMultiStage()
    last = {}

    def step(i):
        inputs = {k: {"depth": v.clone().requires_grad_(True)} for k, v in gd.items()}
        total, _ = mod(inputs, gi, gc, dlossw=[0.5, 1.0, 2.0])
        total.backward()
        last["total"] = total

    elapsed, rounds = timed_region(world, dev, args.warmup, args.steps, step)
    if rank != 0:
        return None
    # dominant launch group: one full-resolution stage forward (rcmvs_unsup_loss_fwd, 2 Vs + 3 launches)
    Vs = Vl - 1
    ref = losses.stage_image(gi[:, 0], 2)
    srcs = losses.nearest_reduce(gi[:, 1:], 1).permute(1, 0, 3, 4, 2).contiguous()
    coef = losses.inverse_warp_coefs(gc["stage3"][:, 0], gc["stage3"][:, 1:])
    ms = event_ms(lambda: losses.UnsupStageLossFn.apply(gd["stage3"], ref, srcs, coef))
    alg = B * H * W * (Vs * (12 + 4 + 12 + 4 + 12 + 12 + 4) + 16 + 4 * Vs)      # warp r/w, terms reads, smoothness, best view
    roofline = {"bound": "hbm", "kernel": "rcmvs_unsup_loss_fwd at 512x640 (inverse_warp + photo_terms per view, smooth_terms, best_view, finalize)",
                "achieved": round(alg / (ms * 1e-3) / 1e9, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(alg / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4), "traffic": None, "algorithmic_bytes": alg,
                "us": round(ms * 1e3, 1), "note": "9 short launches over 0.33 M pixels: launch / latency bound, not bandwidth bound"}
    result = {"metric": "loss steps/sec (UnsupLossMultiStage forward+backward, 4 views 512x640, 3 stages)", "value": round(world * args.steps / elapsed, 2),
              "unit": "steps/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 4),
              "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
              "config": {"workload": "SURVEY 8f-2: losses/unsup_loss.py UnsupLossMultiStage on BASELINE configs[2] shapes (batch 1 per GPU)",
                         "views": Vl, "height": H, "width": W, "parallelism": f"sample-per-gpu x{world}"},
              "roofline": roofline, **rounds_fields(rounds)}
    if world == 1 and not args.no_cpu_baseline:
        from oracle import unsup_loss as O
        nthreads = host_threads()
        torch.set_num_threads(nthreads)
        times = []
        for _ in range(3):
            c0 = time.perf_counter()
            inputs = {k: {"depth": v.clone().requires_grad_(True)} for k, v in dep.items()}
            ctotal, _ = O.unsup_loss_multi_stage(inputs, imgs, cams, dlossw=[0.5, 1.0, 2.0])
            ctotal.backward()
            times.append(time.perf_counter() - c0)
        cpu_s = sorted(times[1:])[len(times[1:]) // 2]
        result["cpu_baseline"] = {"value": round(1.0 / cpu_s, 4), "unit": "steps/s", "cores": nthreads, "kind": "port",
                                  "sample": f"median of 2 steps of the same workload after 1 warm-up, oracle/unsup_loss.py (the reference's op graph on PyTorch-CPU), {nthreads} threads"}
        result["parity"] = {"loss_hip": float(last["total"]), "loss_oracle": float(ctotal),
                            "rel": abs(float(last["total"]) - float(ctotal)) / abs(float(ctotal)), "tolerance": 2e-5}
    return result


def bench_fusion(args, rank, world, dev):
    """SURVEY.md 8f-3.  Step = one reference view of the fusion filter at the reference's DTU evaluation shape (1184x1600 depth
    maps, 10 source views): rcmvs_fuse_view + ordered compaction, every depth map of the scan resident in HBM."""
    import numpy as np
    from rc_mvsnet_amd import fusion, synthetic
    Hf, Wf, n_src, Vf = 1184, 1600, 10, 11
    s = synthetic.fusion_scan(V=Vf, H=Hf, W=Wf, seed=rank, n_src=n_src)
    depth_all = torch.from_numpy(s["depth"]).to(dev)
    jobs = []
    for ref, srcs in s["pairs"]:
        mats = torch.from_numpy(fusion.fusion_matrices(s["K"][ref], s["E"][ref], [s["K"][i] for i in srcs], [s["E"][i] for i in srcs])).to(dev)
        jobs.append((ref, srcs, torch.from_numpy(s["conf"][ref]).to(dev), torch.from_numpy(s["img"][ref].astype(np.float32) / 255.0).to(dev), mats))
    kept = {}

    def fuse(i):
        ref, srcs, conf, img, mats = jobs[i % len(jobs)]
        return fusion.fuse_view(depth_all, ref, srcs, conf, img, mats, 0.8, 3, 0.5, 0.01)

    def step(i):
        r = fuse(i)
        xyz, _ = fusion.compact_points(r["masks"][2], r["xyz"], r["rgb"])
        kept["n"] = len(xyz)

    elapsed, rounds = timed_region(world, dev, args.warmup, args.steps, step)
    if rank != 0:
        return None
    ms = event_ms(lambda: fuse(0))
    alg = Hf * Wf * (4 + 4 + 12 + 4 * n_src + 3 + 4 + 12 + 3)        # depth, conf, img, each source map once; masks, avg, xyz, rgb
    roofline = {"bound": "hbm", "kernel": "rcmvs::fuse_view_kernel (1 launch per reference view)", "achieved": round(alg / (ms * 1e-3) / 1e9, 1),
                "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(alg / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4), "traffic": None,
                "algorithmic_bytes": alg, "us": round(ms * 1e3, 1),
                "note": "fp64 chain with fp32 cast points, ~250 fp64 operations per (pixel, source view): closer to the fp64 vector rate than to HBM"}
    result = {"metric": "fused reference views/sec (filter_depth body, 1184x1600, 10 source views)", "value": round(world * args.steps / elapsed, 1),
              "unit": "ref-views/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 4),
              "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
              "config": {"workload": "SURVEY 8f-3: eval_rcmvsnet_dtu.py filter_depth per-reference-view body, DTU evaluation shape", "height": Hf,
                         "width": Wf, "source_views": n_src, "points_kept": kept.get("n"), "parallelism": f"scan-per-gpu x{world}"},
              "roofline": roofline, **rounds_fields(rounds)}
    if world == 1 and not args.no_cpu_baseline:
        from oracle import fusion as O
        ref, srcs = s["pairs"][0]
        times = []
        for _ in range(2):
            c0 = time.perf_counter()
            o = O.fuse_view(s["depth"][ref], s["conf"][ref], s["img"][ref].astype(np.float32) / 255.0, s["K"][ref], s["E"][ref],
                            [s["depth"][i] for i in srcs], [s["K"][i] for i in srcs], [s["E"][i] for i in srcs], 0.8, 3, 0.5, 0.01)
            times.append(time.perf_counter() - c0)
        g = fuse(0)
        result["cpu_baseline"] = {"value": round(1.0 / min(times), 4), "unit": "ref-views/s", "cores": 1, "kind": "port",
                                  "sample": "best of 2 reference views of the same workload, oracle/fusion.py (the reference's numpy path; numpy is "
                                            "single-threaded here apart from BLAS in the 3xN matmuls)"}
        result["parity"] = {"final_mask_mismatch_frac": float((g["masks"][2].cpu().numpy().astype(bool) != o["final"]).mean()), "tolerance": 1e-4}
    return result


def bench_train_step(args, rank, world, dev):
    """BASELINE configs[2] (N = 1) / configs[3] (N > 1, one sample per GPU, RCCL gradient exchange).  Step = one training
    iteration (rc_mvsnet_amd/train_step.py = train_rcmvsnet.py:279-312,330-446) on synthetic inputs resident in HBM."""
    import torch.distributed as dist
    from rc_mvsnet_amd import ops, train_step as ts
    Vt = 4
    model, model_nerf, opt = ts.build(dev, seed=0)
    sync = None
    if world > 1:
        (model, model_nerf), opt, sync = ts.make_data_parallel([model, model_nerf])     # SyncBatchNorm + one Adam + GradSync (train_rcmvsnet.py:524-525)
    imgs, proj, dv, batch = ts.synthetic_sample(dev, H=H, W=W, V=Vt, seed=rank)
    last = {}

    def step(i):
        last.update(ts.train_step(model, model_nerf, opt, imgs, proj, dv, batch, grad_sync=sync))

    elapsed, rounds = timed_region(world, dev, args.warmup, args.steps, step)
    if rank != 0:
        return None
    # dominant memory-bound kernel of the iteration: the K1 backward scatter at stage 3 (two calls per iteration)
    C, D, h, w = FEAT_C[2], NDEPTHS[2], H, W
    g = torch.Generator().manual_seed(0)
    feats = torch.randn(1, Vt, h, w, C, generator=g).to(dev)
    rot, trans = ops.compose_homography(proj["stage3"].contiguous().float())
    prev = (600.0 + 100.0 * torch.rand(1, h // 2, w // 2, generator=g)).to(dev)
    planes = ops.hypothesis_planes(prev, dv, (H, W), 1, D, 1.0)
    gvar = torch.randn(1, D, h, w, C, generator=g).to(dev)
    ms = event_ms(lambda: ops.warp_variance_bwd(feats, rot, trans, planes, gvar, None), reps=10)
    alg = 4 * (C * D * h * w + 2 * Vt * C * h * w + 2 * D * h * w)       # gradient volume in, feature maps in, feature gradients out, planes
    roofline = {"bound": "hbm", "kernel": "rcmvs::warp_variance_bwd (K1 backward scatter, stage 3: 8 planes x 512x640 x 8 channels, 3 source views)",
                "achieved": round(alg / (ms * 1e-3) / 1e9, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(alg / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4), "traffic": None, "algorithmic_bytes": alg, "us": round(ms * 1e3, 1),
                "note": "4 fp32 atomics per (tap, channel quad): atomic-rate bound, not bandwidth bound"}
    result = {"metric": "training iterations/sec (DTU-shaped 4 views 512x640, D=48/32/8, rendering branch 1024 rays x 128 samples)",
              "value": round(world * args.steps / elapsed, 3), "unit": "iterations/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
              "ms_per_step": round(elapsed / args.steps * 1e3, 3), "timed_region_s": round(elapsed, 4), "higher_is_better": True, "scaling": "weak",
              "vs_baseline": None, "dtype": "f32", "data": "synthetic",
              "config": {"workload": f"BASELINE configs[{2 if world == 1 else 3}]: train_rcmvsnet.py iteration (2 x CascadeMVSNet.forward, "
                                     "Rendering_Consistency_Net.forward, UnsupLoss + AugLoss + render losses, backward, Adam), batch 1 per GPU",
                         "views": Vt, "height": H, "width": W, "ndepths": list(NDEPTHS), "rays": 1024, "samples": 128,
                         "parallelism": f"dp{world}" + (" (SyncBatchNorm + one reduce-scatter/all-gather gradient message over RCCL)" if world > 1 else ""),
                         "rccl_ranks": world, "gpus": args.gpus},
              "roofline": roofline, "losses": {k: round(v, 5) for k, v in last.items()}, **rounds_fields(rounds)}
    if world == 1 and not args.no_cpu_baseline:
        # bounded sample of the reference's CPU path: forward + backward of ONE of the iteration's two CascadeMVSNet passes
        # (oracle/aten_graph.py on the host cores); the iteration has two of them plus the renderer and the losses, so
        # 1 / (2 t) is an upper bound of the CPU rate
        from oracle import aten_graph
        nthreads = host_threads()
        torch.set_num_threads(nthreads)
        cm, _, _ = ts.build(torch.device("cpu"), seed=0)
        cm.train()
        ci, cp, cd = imgs.cpu(), {k: v.cpu() for k, v in proj.items()}, dv.cpu()
        c0 = time.perf_counter()
        out, noref = aten_graph.cascade_forward(cm, ci, cp, cd)
        (sum(out[f"stage{s}"]["depth"].mean() for s in (1, 2, 3)) + noref.mean()).backward()
        t = time.perf_counter() - c0
        result["cpu_baseline"] = {"value": round(1.0 / (2.0 * t), 5), "unit": "iterations/s", "cores": nthreads, "kind": "port",
                                  "sample": f"one CascadeMVSNet forward+backward at the full config-3 size through oracle/aten_graph.py ({t:.1f} s, "
                                            f"{nthreads} threads, no warm-up); an iteration holds two such passes plus the renderer and the losses, "
                                            "so the value 1/(2 t) is an upper bound of the reference's CPU rate"}
    return result


def bench_stub(args, rank, world, dev):
    """The contract without a GPU (tests/test_bench_contract_cpu.py): a step is one small CPU matmul, ranks rendezvous over gloo.  Exercises
    what the real workloads share -- the self-spawn of N ranks, the barrier-bracketed rounds, max over ranks, one JSON line from rank 0."""
    x = torch.randn(64, 64, generator=torch.Generator().manual_seed(rank))
    acc = {"n": 0}

    def step(i):
        acc["y"] = x @ x
        acc["n"] += 1

    elapsed, rounds = timed_region(world, dev, args.warmup, args.steps, step)
    if rank != 0:
        return None
    return {"metric": "stub steps/sec (64x64 CPU matmul; contract self-test, not a benchmark)", "value": round(world * args.steps / elapsed, 3),
            "unit": "steps/s", "n_gpus": args.gpus, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 6),
            "timed_region_s": round(elapsed, 6), "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "stub", "parallelism": f"x{world}", "ranks": world, "procs_per_gpu": args.procs_per_gpu},
            "steps_run_by_rank0": acc["n"], **rounds_fields(rounds)}


def main(argv=None):
    argv = list(sys.argv[1:] if argv is None else argv)
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="cascade", choices=["cascade", "train_step", "unsup_loss", "fusion", "stub"],
                    help="cascade = BASELINE.json's metric (default); train_step / unsup_loss / fusion are the SURVEY 8f rows either side of the "
                         "path; stub = the contract's plumbing on CPU + gloo (tests)")
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--procs-per-gpu", type=int, default=1,
                    help="worker processes per GPU (cascade workload): independent (scene, view) items need no collective, so P processes on one "
                         "GPU overlap each other's latency-bound phases like HIP streams would, without sharing an address space")
    ap.add_argument("--shape", default="dtu_bench", choices=sorted(SHAPES),
                    help="cascade workload: dtu_bench = BASELINE configs[1] (the metric's configuration, default); dtu_eval = the reference's DTU evaluation shape "
                         "(5 views, 1184x1600); tanks = configs[4]'s single-GPU workload (7 views, 1056x1920, D=64/32/8).  The other shapes print a full line "
                         "(roofline with that shape's algorithmic bytes, parity against the oracle on a down-scaled twin) without the CPU timing / side passes.")
    ap.add_argument("--steps", type=int, default=None, help="timed steps (default: ~1 s of cascade forwards: 600 at the default shape)")
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-train-step", action="store_true", help="skip the short config-3 training-iteration timing appended to the default line")
    ap.add_argument("--no-side-pass", action="store_true", help="skip the side pass of the cascade workload (N = 1): two worker processes on the GPU")
    ap.add_argument("--cpu-scenes", type=int, default=5, help="scenes timed on the CPU baseline after 2 warm-ups (bounded sample, BASELINE.md section 3)")
    args = ap.parse_args(argv)
    if args.gpus < 1 or args.procs_per_gpu < 1:
        ap.error("--gpus and --procs-per-gpu must be >= 1")
    default_steps = args.steps is None
    if default_steps:
        args.steps = SHAPES[args.shape]["steps"] if args.workload == "cascade" else 600
    if args.shape != "dtu_bench":
        if args.workload != "cascade":
            ap.error("--shape applies to the cascade workload")
        set_shape(args.shape)
        args.no_train_step = args.no_side_pass = True          # appendices of the headline configuration only
    nproc = args.gpus * args.procs_per_gpu
    if "WORLD_SIZE" not in os.environ and nproc > 1:
        from rc_mvsnet_amd.sharding import launch_ranks
        sys.exit(launch_ranks(__file__, nproc, argv))          # our own N (x P) ranks, like train_rcmvsnet.py:632-636 spawns its own

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != nproc:
        raise SystemExit(f"bench.py: --gpus {args.gpus} x --procs-per-gpu {args.procs_per_gpu} = {nproc} ranks, but WORLD_SIZE={world} "
                         f"(launch torch.distributed.run with --nproc-per-node {nproc}, or drop WORLD_SIZE and let bench.py start them)")
    if args.workload == "stub":
        dev = torch.device("cpu")
        if world > 1:
            from rc_mvsnet_amd.sharding import init_process_group
            dist = init_process_group("gloo")
        result = bench_stub(args, rank, world, dev)
        if result is not None:
            print(json.dumps(result), flush=True)
        return
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the HIP path has no CPU fallback): torch.cuda.is_available() is False")
    if args.gpus > torch.cuda.device_count():
        raise SystemExit(f"bench.py: --gpus {args.gpus} but this node shows {torch.cuda.device_count()} GPU(s) "
                         f"(HIP_VISIBLE_DEVICES / ROCR_VISIBLE_DEVICES = {os.environ.get('HIP_VISIBLE_DEVICES', os.environ.get('ROCR_VISIBLE_DEVICES', 'unset'))})")
    from rc_mvsnet_amd.sharding import device_index as _device_index
    device_index = _device_index(local_rank, args.procs_per_gpu)           # ranks g*P .. g*P+P-1 share GPU g
    torch.cuda.set_device(device_index)
    dev = torch.device("cuda", device_index)
    if world > 1:
        import torch.distributed as dist
        # RCCL needs one device per rank: with several processes per GPU the rendezvous (barrier, max-time reduction -- there is no data-path
        # collective in the cascade workload) goes over gloo instead
        from rc_mvsnet_amd.sharding import init_process_group
        dist = init_process_group("nccl" if args.procs_per_gpu == 1 else "gloo")
        assert dist.get_world_size() == world
    if args.procs_per_gpu > 1 and args.workload != "cascade":
        raise SystemExit("bench.py: --procs-per-gpu applies to the cascade workload (independent items); training ranks own one GPU each")

    from rc_mvsnet_amd import _lib, ops, synthetic
    from rc_mvsnet_amd.casmvsnet import CascadeMVSNet_eval
    _lib.load()
    if args.workload != "cascade":
        if args.workload == "train_step" and default_steps and args.warmup == 20:
            args.steps, args.warmup = 10, 3                       # an iteration is ~80 ms: the defaults of the cascade workload are overkill
        result = {"unsup_loss": bench_unsup_loss, "fusion": bench_fusion, "train_step": bench_train_step}[args.workload](args, rank, world, dev)
        if result is not None:
            print(json.dumps(result), flush=True)
        return

    sd = synthetic.cascade_state_dict(0)

    def make_model(state):
        m = CascadeMVSNet_eval(ndepths=list(NDEPTHS), depth_interals_ratio=list(RATIOS))
        m.load_state_dict(state, strict=True)
        return m.to(dev).eval()

    model = make_model(sd)

    # a few distinct scenes resident in HBM; rank r starts at a different one
    scenes = []
    for seed in range(4):
        imgs, pm, dv = synthetic.cascade_inputs(1, V, H, W, seed)
        scenes.append((imgs.to(dev), {k: v.to(dev) for k, v in pm.items()}, dv.to(dev)))

    def step(i):                                              # one CascadeMVSNet_eval.forward at batch 1 on the current stream
        return model(*scenes[(i + rank) % len(scenes)])

    cdev = dev if args.procs_per_gpu == 1 else torch.device("cpu")      # where the rendezvous tensors live (RCCL / gloo)
    with torch.no_grad():
        for i in range(args.warmup):
            out = step(i)
        torch.cuda.synchronize()
        elapsed, own, rounds = timed_rounds(world, cdev, args.steps, step, torch.cuda.synchronize)
    # second, UNTIMED pass over the same scenes with HIP events (torch's current stream = the launch stream) around every K1 launch and
    # every 3-D convolution launch: ~90 event records per scene would perturb `value` inside the timed region
    events, conv_events, smooth_events = [], [], []
    nprobe = min(10, args.steps)
    if rank == 0:
        with torch.no_grad():
            ops.K1_EVENTS, ops.CONV_EVENTS = events, conv_events
            for i in range(nprobe):
                step(i)
            torch.cuda.synchronize()
            ops.K1_EVENTS = ops.CONV_EVENTS = None
            # the smooth-head twin of the scene (prob.weight x1 instead of BASELINE.md's x20): K1 on depth maps that are not noise
            sd1 = model1 = None
            if world == 1:
                sd1 = synthetic.cascade_state_dict(0, prob_gain=1.0)
                model1 = make_model(sd1)
                for i in range(3):
                    model1(*scenes[i % len(scenes)])
                torch.cuda.synchronize()
                ops.K1_EVENTS = smooth_events
                for i in range(nprobe):
                    model1(*scenes[i % len(scenes)])
                torch.cuda.synchronize()
                ops.K1_EVENTS = None

    # SIDE PASS (N = 1, one process; never `value`).  two_procs_per_gpu: the same workload as `python bench.py --procs-per-gpu 2` in a
    # child -- two worker processes on this GPU, each issuing whole scenes on its own queue from its own address space (the item list is
    # sharded over the processes like over GPUs: rc_mvsnet_amd/sharding.py, eval_driver --procs-per-gpu)
    two_procs = None
    if rank == 0 and world == 1 and not args.no_side_pass:
        try:
            two_procs = two_procs_side_pass(min(300, max(20, args.steps)))
        except Exception as e:                               # never at the expense of the headline line
            two_procs = {"error": f"{type(e).__name__}: {e}"[:300]}

    rank_rates = None
    if world > 1:
        mine = torch.tensor([own], device=cdev, dtype=torch.float64)
        every = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(every, mine)                        # every rank's own time of the median round: per-rank scenes/s (min / max) in the line
        rank_rates = [args.steps / float(t.item()) for t in every]
        dist.barrier()
        dist.destroy_process_group()                         # before rank 0 spends ~25 s on the CPU baseline alone

    if rank != 0:
        return

    # ---- roofline of K1 from the events of the probe pass -------------------------------------
    nstage = len(NDEPTHS)
    bytes_stage = k1_algorithmic_bytes()

    def k1_summary(evs):
        ms = [e0.elapsed_time(e1) for (e0, e1) in evs]
        per_stage = [sum(ms[s::nstage]) / max(1, len(ms[s::nstage])) for s in range(nstage)]
        tot = sum(per_stage)
        ach = sum(bytes_stage) / (tot * 1e-3) / 1e9 if tot > 0 else 0.0
        return per_stage, ach

    per_stage_ms, achieved = k1_summary(events)
    traffic, traffic_file = None, None  # HBM bytes per scene from the committed PMC profile (FETCH_SIZE x2 + WRITE_SIZE)
    for name in ("r6_k1_traffic.json", "r5_k1_traffic.json", "r4_k1_traffic.json"):
        try:
            traffic = json.load(open(os.path.join(ROOT, "profiles", name)))["bytes_per_scene"]
            traffic_file = name
            break
        except Exception:
            pass
    k1_kernels = {3: "rcmvs::warp_variance_win_kernel (stage 1: pixel-invariant planes, source windows in LDS), warp_variance_tp_kernel (stage 2: two-phase "
                     "gathers), warp_variance_pp_kernel (stage 3: plane-pipelined gathers)",
                  5: "rcmvs::warp_variance_pp_kernel (stages 1 and 3: plane-pipelined gathers, four source views), warp_variance_tp_kernel (stage 2: two-phase "
                     "gathers, FMA build)",
                  7: "rcmvs::warp_variance_pp_kernel (plane-pipelined gathers, six source views in two groups of three)"}
    roofline = {"bound": "hbm", "kernel": "K1, 3 launches per scene: " + k1_kernels[V],
                "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic,
                "traffic_source": f"profiles/{traffic_file} -- rocprofv3 PMC passes over this command line in an earlier visit "
                                  "(FETCH_SIZE x2 + WRITE_SIZE per launch, summed over the 3 launches of a scene); "
                                  "a committed measurement, NOT taken in this run (a profiler cannot wrap its own process)",
                "algorithmic_bytes_per_scene": sum(bytes_stage),
                "per_stage_us": [round(m * 1e3, 2) for m in per_stage_ms],
                "per_stage_GBs": [round(b / (m * 1e-3) / 1e9, 1) if m > 0 else 0.0 for b, m in zip(bytes_stage, per_stage_ms)],
                "timing": f"HIP events carrying each K1 kernel's own start / stop timestamps (hipExtLaunchKernelGGL through rcmvs_warp_variance_timed_fwd: the "
                          f"duration rocprofv3 reports; event records AROUND a launch add 3-6 us of marker packets), separate untimed pass of {nprobe} scenes "
                          "right after the timed region"}
    try:        # the window path depends on the geometry only (homographies, plane table): count stage 1's tiles on it with blank feature maps
        if V != 3:
            raise RuntimeError("stage 1 runs the plane-pipelined gather form at this view count (profiles/r6_k1_views.txt)")
        with torch.no_grad():
            imgs0, pm0, dv0 = scenes[0]
            rot1, trans1 = ops.compose_homography(pm0["stage1"].contiguous().float())
            planes1 = ops.hypothesis_planes(None, dv0, (H, W), 4, NDEPTHS[0], RATIOS[0])
            _, tiles, on_window = ops.warp_variance_win(torch.zeros(1, V, H // 4, W // 4, FEAT_C[0], device=dev), rot1, trans1, planes1, NDEPTHS[0], variant=5)
        roofline["stage1_window_path"] = {"tiles": tiles, "on_lds_windows": on_window, "fallback_to_gathers": tiles - on_window}
    except Exception as e:
        roofline["stage1_window_path"] = {"error": f"{type(e).__name__}: {e}"[:200]}
    if smooth_events:
        sm_ms, sm_ach = k1_summary(smooth_events)
        roofline["smooth_scene"] = {"frac": round(sm_ach / HBM_PEAK_GBS, 4), "achieved": round(sm_ach, 1),
                                    "per_stage_us": [round(m * 1e3, 2) for m in sm_ms],
                                    "note": "same inputs and weights with prob.weight x1 instead of BASELINE.md's x20: stage-2 / 3 depth maps that are not "
                                            "noise, i.e. the gather locality of a trained network (the x20 scene's neighbouring pixels sample +-20 px apart)"}

    roofline_conv = conv_roofline(conv_events, min(10, args.steps))

    result = {
        "metric": SHAPES[args.shape]["metric"],
        "value": round(world * args.steps / elapsed, 3),
        "unit": "ref-scenes/s",
        "n_gpus": args.gpus,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": round(elapsed / args.steps * 1e3, 4),
        "timed_region_s": round(elapsed, 4),
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "f32",
        "data": "synthetic",
        "config": {"workload": SHAPES[args.shape]["workload"], "shape": args.shape,
                   "views": V, "height": H, "width": W, "ndepths": list(NDEPTHS),
                   "parallelism": f"scene-per-gpu x{args.gpus}" + (f", {args.procs_per_gpu} worker processes per GPU" if args.procs_per_gpu > 1 else ""),
                   "ranks": world, "procs_per_gpu": args.procs_per_gpu,
                   "rccl_ranks": world if (world > 1 and args.procs_per_gpu == 1) else 0},
        "roofline": roofline,
        "roofline_conv": roofline_conv,
        # the scene as a whole against HBM: what every layer must move at least (K1 + the 3-D CNN's ideal activation traffic + the
        # depth head + FeatureNet's maps) over the wall time of a step -- the honest summary next to the
        # per-kernel objects: the scene is bound by its kernels' tick structures and launch latencies, not by a pipe
        "roofline_scene": {"bound": "hbm", "algorithmic_bytes_per_scene": SCENE_ALGORITHMIC_BYTES,
                           "achieved": round(SCENE_ALGORITHMIC_BYTES / (elapsed / args.steps) / 1e9, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                           "frac": round(SCENE_ALGORITHMIC_BYTES / (elapsed / args.steps) / 1e9 / HBM_PEAK_GBS, 4),
                           "parts_MB": SCENE_PARTS_MB},
    }
    result.update(rounds_fields(rounds))
    if two_procs is not None:
        result["two_procs_per_gpu"] = two_procs
    if rank_rates:
        result["per_rank_scenes_per_s"] = {"min": round(min(rank_rates), 2), "max": round(max(rank_rates), 2), "ranks": world}

    # ---- the other shapes: no CPU timing at full size (a 1184x1600 five-view scene is minutes of PyTorch-CPU per pass: stated, not measured);
    # parity of the same kernels against the oracle on a DOWN-SCALED TWIN (same view count, same depth counts, 160x192)
    if world == 1 and args.shape != "dtu_bench" and not args.no_cpu_baseline:
        from oracle import cascade
        torch.set_num_threads(host_threads())
        th, tw = 160, 192
        imgs, pm, dv = synthetic.cascade_inputs(1, V, th, tw, 0)
        with torch.no_grad():
            c0 = time.perf_counter()
            ref = cascade.forward_eval(imgs, pm, dv, sd, NDEPTHS, RATIOS, impl="aten")
            twin_s = time.perf_counter() - c0
            hip = model(imgs.to(dev), {k: v.to(dev) for k, v in pm.items()}, dv.to(dev))
        rng = float(dv[0, -1] - dv[0, 0])
        dd = (hip["depth"].cpu() - ref["depth"]).abs()
        result["cpu_baseline"] = None
        result["cpu_baseline_note"] = (f"not timed at {H}x{W} (minutes of PyTorch-CPU per scene); the oracle's ATen op graph took {twin_s:.2f} s for the "
                                       f"{th}x{tw} twin on {host_threads()} threads")
        result["parity"] = {"depth_l1_over_range": float(dd.mean()) / rng, "depth_l1_mm": float(dd.mean()), "depth_max_abs_mm": float(dd.max()),
                            "tolerance": 1e-4, "on": f"down-scaled twin: {V} views {th}x{tw}, D={list(NDEPTHS)}, same seeded weights (prob.weight x20), "
                                                     "HIP path vs oracle impl='aten'"}
    # ---- CPU baseline (oracle, ATen op graph of the reference) + parity on the same inputs -----
    if world == 1 and args.shape == "dtu_bench" and not args.no_cpu_baseline:
        from oracle import cascade
        nthreads = host_threads()
        torch.set_num_threads(nthreads)
        imgs, pm, dv = synthetic.cascade_inputs(1, V, H, W, 0)
        times = []
        with torch.no_grad():
            budget_t0 = time.perf_counter()
            for i in range(2 + args.cpu_scenes):              # 2 warm-ups + >= 5 timed passes, median
                c0 = time.perf_counter()
                ref = cascade.forward_eval(imgs, pm, dv, sd, NDEPTHS, RATIOS, impl="aten")
                times.append(time.perf_counter() - c0)
                if time.perf_counter() - budget_t0 > 40.0:    # bounded sample
                    break
            torch.set_num_threads(1)                           # the 1-thread figure: one pass
            c0 = time.perf_counter()
            cascade.forward_eval(imgs, pm, dv, sd, NDEPTHS, RATIOS, impl="aten")
            one_thread_s = time.perf_counter() - c0
            torch.set_num_threads(nthreads)
            hip = model(*scenes[0])
        timed = times[2:] if len(times) > 2 else times[-1:]
        cpu_s = sorted(timed)[len(timed) // 2]
        rng = float(dv[0, -1] - dv[0, 0])
        dd = (hip["depth"].cpu() - ref["depth"]).abs()
        result["cpu_baseline"] = {"value": round(1.0 / cpu_s, 4), "unit": "ref-scenes/s", "cores": nthreads, "kind": "port",
                                  "sample": f"median of {len(timed)} scene(s) of the same config-2 workload after {len(times) - len(timed)} warm-up(s), "
                                            f"oracle impl='aten' (reference op graph on PyTorch-CPU), {nthreads} threads",
                                  "one_thread": {"value": round(1.0 / one_thread_s, 4), "unit": "ref-scenes/s", "sample": "one scene, 1 thread"}}
        result["parity"] = {"depth_l1_over_range": float(dd.mean()) / rng, "depth_l1_mm": float(dd.mean()),
                            "depth_max_abs_mm": float(dd.max()), "frac_pixels_over_0.1mm": float((dd > 0.1).float().mean()),
                            "tolerance": 1e-4,
                            "note": "BASELINE.md's seeded weights scale prob.weight x20: a chaotic soft-argmin in which single-ulp logit "
                                    "differences move isolated pixels by millimetres; 'smooth_head' is the same scene with prob.weight x1"}
        # the same scene with a well-conditioned (trained-like) probability head: HIP vs the CPU op graph
        with torch.no_grad():
            ref1 = cascade.forward_eval(imgs, pm, dv, sd1, NDEPTHS, RATIOS, impl="aten")
            hip1 = model1(*scenes[0])
        d1 = (hip1["depth"].cpu() - ref1["depth"]).abs()
        result["parity"]["smooth_head"] = {"depth_l1_over_range": float(d1.mean()) / rng, "depth_max_abs_mm": float(d1.max()),
                                           "frac_pixels_over_0.1mm": float((d1 > 0.1).float().mean())}
    # ---- BASELINE configs[2] next to the headline: a short timing of the training iteration (full size), so that the driver's
    # default run carries a number for it too (`--workload train_step` is the full line with its own roofline / CPU baseline)
    if world == 1 and not args.no_train_step:
        try:
            from rc_mvsnet_amd import train_step as ts
            del model, scenes
            model1 = None
            torch.cuda.empty_cache()
            tm, tn, topt = ts.build(dev, seed=0)
            ti, tp, td, tb = ts.synthetic_sample(dev, H=H, W=W, V=4, seed=0)
            for _ in range(2):
                ts.train_step(tm, tn, topt, ti, tp, td, tb)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(5):
                last = ts.train_step(tm, tn, topt, ti, tp, td, tb)
            torch.cuda.synchronize()
            t_it = (time.perf_counter() - t0) / 5
            result["train_step"] = {"ms_per_iteration": round(t_it * 1e3, 2), "iterations": 5, "warmup": 2, "loss": round(last["loss"], 5),
                                    "config": "BASELINE configs[2]: train_rcmvsnet.py iteration, 4 views 512x640, D=48/32/8, rendering branch "
                                              "1024 rays x 128 samples, reference losses, backward, Adam; batch 1, fp32, synthetic"}
        except Exception as e:                               # the headline number must not depend on this appendix
            result["train_step"] = {"error": repr(e)[:200]}
    print(json.dumps(result))


if __name__ == "__main__":
    main()
