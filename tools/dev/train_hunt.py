"""Developer probe (GPU box): which ATen ops launch device kernels / copies inside one training iteration, by call site."""
import os, sys, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench


class A: pass


def main():
    from torch.profiler import profile, ProfilerActivity
    captured = {}
    orig = bench.time.perf_counter
    args = A(); args.steps, args.warmup, args.gpus, args.no_cpu_baseline = 1, 2, 1, True
    # reuse bench_train_step's set-up by running it with the profiler around its timed call
    import rc_mvsnet_amd.train_step as ts
    real = ts.train_step
    state = {"n": 0}

    def wrapped(*a, **k):
        state["n"] += 1
        if state["n"] == 3:                       # the timed step (after 2 warm-ups)
            with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True, record_shapes=True) as prof:
                out = real(*a, **k)
                torch.cuda.synchronize()
            captured["prof"] = prof
            return out
        return real(*a, **k)

    ts.train_step = wrapped
    bench.bench_train_step(args, 0, 1, torch.device("cuda", 0))
    prof = captured["prof"]
    agg = collections.OrderedDict()
    for e in prof.events():
        if not e.name.startswith("aten::"): continue
        dt = getattr(e, "self_device_time_total", None)
        if dt is None: dt = getattr(e, "self_cuda_time_total", 0)
        if not dt: continue
        site = next((s for s in (e.stack or []) if "rc_mvsnet_amd" in s or "bench.py" in s), "?")
        shp = str([tuple(x) for x in (e.input_shapes or []) if x])[:70]
        key = (e.name, site.strip()[-70:] + " " + shp)
        a = agg.setdefault(key, [0, 0.0]); a[0] += 1; a[1] += dt
    tot = sum(v[1] for v in agg.values())
    print(f"ATen ops with device time in one iteration: {sum(v[0] for v in agg.values())} calls, {tot / 1e3:.2f} ms")
    for (name, site), (n, dt) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:70]:
        print(f"{dt:8.1f} us x{n:4d} {name:24s} {site}")


main()
