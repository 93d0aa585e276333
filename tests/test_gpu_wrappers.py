"""GPU: the reference's own wrapping of the modules, replayed literally on the real drop-in modules.
  * eval_rcmvsnet_dtu.py:179-197 -- load_state_dict(state_dict['model'], strict=True), nn.DataParallel(model), .cuda(), .eval(), tocuda,
    no_grad forward: equals the bare module bit for bit (one visible device: DataParallel calls the module itself);
  * the same through DataParallel's REPLICA path (device_ids=[0, 0], batch 2: replicate + scatter + parallel_apply threads + gather);
  * train_rcmvsnet.py:524-525,565-578 on a world-size-1 "nccl" (= RCCL) group: SyncBatchNorm.convert_sync_batchnorm +
    DistributedDataParallel(find_unused_parameters=False) around CascadeMVSNet and Rendering_Consistency_Net, with
    parallel.flat_allreduce_hook registered on both; one train_step -> the gradients of the un-wrapped step.  The hook's reduce-scatter +
    all-gather pair and GradSync's run on RCCL here (parallel.RUN_AT_WORLD_ONE), a degenerate but real launch of each collective."""
import os
import socket

import pytest
import torch
import torch.nn as nn

pytestmark = pytest.mark.gpu


def tocuda(v):
    """utils.tocuda of the reference (utils.py: recursive over dicts / lists; tensors -> .cuda())."""
    if isinstance(v, torch.Tensor):
        return v.cuda()
    if isinstance(v, dict):
        return {k: tocuda(x) for k, x in v.items()}
    if isinstance(v, (list, tuple)):
        return [tocuda(x) for x in v]
    return v


def _eval_model(nd=(16, 8, 8)):
    from rc_mvsnet_amd import synthetic
    from rc_mvsnet_amd.casmvsnet import CascadeMVSNet_eval
    model = CascadeMVSNet_eval(refine=False, ndepths=list(nd), depth_interals_ratio=[4.0, 2.0, 1.0], share_cr=False, cr_base_chs=[8, 8, 8],
                               grad_method="detach")
    state_dict = {"model": synthetic.cascade_state_dict(0)}                  # what torch.load(args.loadckpt) hands back
    model.load_state_dict(state_dict["model"], strict=True)
    return model


def test_eval_script_wrapping_dataparallel_equals_the_bare_module():
    from rc_mvsnet_amd import synthetic
    imgs, pm, dv = synthetic.cascade_inputs(1, 3, 128, 160, 0)
    sample = {"imgs": imgs, "proj_matrices": pm, "depth_values": dv}
    model = _eval_model()
    model = nn.DataParallel(model)
    model.cuda()
    model.eval()
    with torch.no_grad():
        sample_cuda = tocuda(sample)
        outputs = model(sample_cuda["imgs"], sample_cuda["proj_matrices"], sample_cuda["depth_values"])
        again = model(sample_cuda["imgs"], sample_cuda["proj_matrices"], sample_cuda["depth_values"])
    bare = _eval_model().cuda().eval()
    with torch.no_grad():
        want = bare(imgs.cuda(), {k: v.cuda() for k, v in pm.items()}, dv.cuda())
    for key in ("depth", "photometric_confidence"):
        assert torch.equal(outputs[key], want[key]) and torch.equal(again[key], want[key]), key
        assert torch.equal(outputs["stage1"][key], want["stage1"][key]), key
    assert outputs["depth"].shape == (1, 128, 160)


def test_dataparallel_replica_path_batch_two():
    """DataParallel with two 'devices' (the one GPU listed twice): replicate() copies the module per forward, scatter() splits the batch of 2,
    the replicas run on two host threads, gather() concatenates -- every sample equals the bare module's result on it, bit for bit.  (A
    replica is a fresh shallow copy each call: its packed weights / plans are rebuilt per forward -- correct, and the reason INTEGRATION.md
    recommends one process per GPU, rc_mvsnet_amd/sharding.py, over DataParallel for throughput.)"""
    from rc_mvsnet_amd import synthetic
    a, pm, dv = synthetic.cascade_inputs(1, 3, 64, 96, 0)
    b, _, _ = synthetic.cascade_inputs(1, 3, 64, 96, 1)
    imgs = torch.cat((a, b))
    pm2 = {k: v.repeat(2, 1, 1, 1, 1) for k, v in pm.items()}
    dv2 = dv.repeat(2, 1)
    model = nn.DataParallel(_eval_model(), device_ids=[0, 0])
    model.cuda()
    model.eval()
    with torch.no_grad():
        out = model(imgs.cuda(), tocuda(pm2), dv2.cuda())
        out2 = model(imgs.cuda(), tocuda(pm2), dv2.cuda())
    bare = _eval_model().cuda().eval()
    with torch.no_grad():
        for i, im in enumerate((a, b)):
            want = bare(im.cuda(), tocuda(pm), dv.cuda())
            for key in ("depth", "photometric_confidence"):
                assert torch.equal(out[key][i], want[key][0]) and torch.equal(out2[key][i], want[key][0]), (i, key)


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


@pytest.fixture
def nccl_world_one():
    import torch.distributed as dist
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{_free_port()}", rank=0, world_size=1)
    from rc_mvsnet_amd import parallel
    parallel.RUN_AT_WORLD_ONE = True
    yield dist
    parallel.RUN_AT_WORLD_ONE = False
    dist.destroy_process_group()


def _grads(*models):
    return {f"{i}.{n}": p.grad.detach().clone() for i, m in enumerate(models) for n, p in m.named_parameters() if p.grad is not None}


def _one_step(wrap, dist=None):
    from rc_mvsnet_amd import train_step as ts
    dev = torch.device("cuda", 0)
    model, model_nerf, opt = ts.build(dev, ndepths=(16, 8, 8), n_samples=32)
    imgs, proj, dv, batch = ts.synthetic_sample(dev, H=128, W=160, V=4)
    sync = None
    if wrap in ("ddp", "ddp_hook"):
        from torch.nn.parallel import DistributedDataParallel as DDP
        from rc_mvsnet_amd import parallel
        model = nn.SyncBatchNorm.convert_sync_batchnorm(model)                          # train_rcmvsnet.py:524-525
        model_nerf = nn.SyncBatchNorm.convert_sync_batchnorm(model_nerf)
        assert any(isinstance(m, nn.SyncBatchNorm) for m in model.modules())
        opt = torch.optim.Adam(list(filter(lambda p: p.requires_grad, model.parameters())) + list(model_nerf.parameters()), lr=1e-4,
                               betas=(0.9, 0.999))                                      # :532-533
        model = DDP(model, device_ids=[0], find_unused_parameters=False)                # :565-578
        model_nerf = DDP(model_nerf, device_ids=[0], find_unused_parameters=False)
        if wrap == "ddp_hook":
            model.register_comm_hook(None, parallel.flat_allreduce_hook)
            model_nerf.register_comm_hook(None, parallel.flat_allreduce_hook)
    elif wrap == "gradsync":
        from rc_mvsnet_amd import parallel
        model = nn.SyncBatchNorm.convert_sync_batchnorm(model)
        model_nerf = nn.SyncBatchNorm.convert_sync_batchnorm(model_nerf)
        opt = torch.optim.Adam(list(model.parameters()) + list(model_nerf.parameters()), lr=1e-4, betas=(0.9, 0.999))
        sync = parallel.GradSync([model, model_nerf])
    import numpy as np
    torch.manual_seed(1234)                                                             # the ray draws of the iteration
    np.random.seed(1234)                                                                # the blanked rectangle (losses/aug_loss.py draws it with numpy)
    losses = ts.train_step(model, model_nerf, opt, imgs, proj, dv, batch, grad_sync=sync)
    torch.cuda.synchronize()
    inner = [m.module if hasattr(m, "module") else m for m in (model, model_nerf)]
    assert all(p.grad is not None for m in inner for p in m.parameters() if p.requires_grad)      # find_unused_parameters=False holds: every parameter took part
    weights = {f"{i}.{n}": p.detach().clone() for i, m in enumerate(inner) for n, p in m.named_parameters()}
    return losses, _grads(*inner), weights


@pytest.mark.parametrize("wrap", ["ddp", "ddp_hook", "gradsync"])
def test_training_script_wrapping_on_a_world_one_rccl_group(nccl_world_one, wrap):
    base_losses, base_g, base_w = _one_step("bare")
    losses, g, w = _one_step(wrap, nccl_world_one)
    assert set(g) == set(base_g) and len(g) > 150
    for k in ("loss", "base", "aug", "render"):
        assert abs(losses[k] - base_losses[k]) <= 1e-5 * max(1.0, abs(base_losses[k])), (k, losses, base_losses)
    worst = 0.0
    for name, want in base_g.items():
        scale = float(want.abs().max())
        if scale == 0.0:
            assert float(g[name].abs().max()) == 0.0, name
            continue
        worst = max(worst, float((g[name] - want).abs().max()) / scale)
    print(f"{wrap}: worst relative gradient difference to the un-wrapped step = {worst:.2e}")
    # the SyncBatchNorm branch sums in fp64 over the (one-rank) group where plain BatchNorm sums per launch: not bit-equal, but tight
    assert worst < 2e-5
    for name, want in base_w.items():                                   # and the Adam step landed on the wrapped module's parameters
        assert torch.allclose(w[name], want, rtol=0, atol=2.5e-4), name


def test_eval_driver_two_worker_processes_per_gpu_write_the_same_files(tmp_path):
    """`eval_driver --gpus 1 --procs-per-gpu 2`: the driver starts its own two ranks on the one GPU, the (scan, view) items are sharded over
    them (sharding.shard_items) and every depth / confidence map equals the one-process run's, byte for byte."""
    import subprocess
    import sys
    from rc_mvsnet_amd.sharding import clean_env
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    common = ["--scans", "2", "--ref-views", "3", "--views", "3", "--height", "128", "--width", "160", "--ndepths", "16,8,8"]
    one, two = str(tmp_path / "one"), str(tmp_path / "two")
    for outdir, extra in ((one, []), (two, ["--gpus", "1", "--procs-per-gpu", "2"])):
        out = subprocess.run([sys.executable, "-m", "rc_mvsnet_amd.eval_driver", "--outdir", outdir] + common + extra, cwd=root, env=clean_env(),
                             capture_output=True, text=True, timeout=900)
        assert out.returncode == 0, out.stderr[-3000:]
        if extra:
            assert "rank 0/2: 3 of 6 items" in out.stdout and "rank 1/2: 3 of 6 items" in out.stdout, out.stdout
    names = sorted(os.path.relpath(os.path.join(d, f), one) for d, _, fs in os.walk(one) for f in fs)
    assert len(names) == 12                                              # 6 items x (depth_est, confidence)
    for n in names:
        a, b = open(os.path.join(one, n), "rb").read(), open(os.path.join(two, n), "rb").read()
        if a != b:                                                       # (say how different: a wrong pixel and a truncated file are different bugs)
            import numpy as np
            from rc_mvsnet_amd import data_io
            x, y = data_io.read_pfm(os.path.join(one, n))[0], data_io.read_pfm(os.path.join(two, n))[0]
            d = np.abs(np.asarray(x, dtype=np.float64) - np.asarray(y, dtype=np.float64)) if np.shape(x) == np.shape(y) else None
            raise AssertionError(f"{n}: {len(a)} vs {len(b)} bytes" + ("" if d is None else f", {int((d > 0).sum())} of {d.size} values differ, max {float(d.max()):.3e}"))


def test_a_second_stream_is_refused(monkeypatch):
    """Two hardware queues of one process give silently wrong results on this platform (profiles/r6_two_streams.txt), and the boundary's
    contract is "the CURRENT stream": the binding remembers the first stream per device and raises on a second one instead of returning wrong
    numbers; GPU_MAX_HW_QUEUES=1 (one hardware queue for the process: the mode in which the hazard does not occur) or the developer override
    lift the refusal."""
    from rc_mvsnet_amd import _lib, ops
    monkeypatch.delenv("GPU_MAX_HW_QUEUES", raising=False)
    monkeypatch.delenv("RCMVS_ALLOW_MULTI_STREAM", raising=False)
    x = torch.randn(4, 8, 16, 16, device="cuda")
    first = ops.absmax(x)                                # (the process's stream so far: whatever the earlier tests of this process ran on)
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        if ops._FIRST_STREAM[x.device.index] != side.cuda_stream:
            with pytest.raises(_lib.RcmvsError, match="ONE HIP stream"):
                ops.absmax(x)
        monkeypatch.setenv("RCMVS_ALLOW_MULTI_STREAM", "1")
        again = ops.absmax(x)
    side.synchronize()
    assert torch.equal(first, again)
    monkeypatch.delenv("RCMVS_ALLOW_MULTI_STREAM")
    assert torch.equal(ops.absmax(x), first)             # the first stream keeps working
