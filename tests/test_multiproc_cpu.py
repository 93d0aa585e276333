"""CPU, world_size 2 over gloo: the N>1 path of the scene-sharded inference harness -- per-rank
scene partition with no data-path collective, barrier + max-over-ranks timing reduction."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from rc_mvsnet_amd.sharding import shard_items
    items = [(scan, view) for scan in range(3) for view in range(7)]
    mine = shard_items(items, rank, world)
    dist.barrier()
    t = torch.tensor([0.5 + rank], dtype=torch.float64)       # pretend elapsed seconds
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    gathered = [None] * world
    dist.all_gather_object(gathered, mine)
    q.put((rank, float(t.item()), gathered))
    dist.destroy_process_group()


def test_scene_sharding_world2():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, tmax, gathered in res:
        assert tmax == 1.5                                     # max over ranks
        flat = [tuple(x) for part in gathered for x in part]
        assert sorted(flat) == sorted((s, v) for s in range(3) for v in range(7))   # a partition: no loss, no overlap
        assert abs(len(gathered[0]) - len(gathered[1])) <= 1


def _ddp_worker(rank, world, port, q, compose=False):
    import torch.nn as nn
    from torch.nn.parallel import DistributedDataParallel as DDP
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle import aten_graph
    from rc_mvsnet_amd import parallel
    from rc_mvsnet_amd.parallel import GradSync, flat_allreduce_hook
    from rc_mvsnet_amd.casmvsnet import CostRegNet
    parallel.COMPOSE_ON_GLOO = compose           # True: the reduce-scatter + all-gather branch (what runs on RCCL) on gloo

    class Graph(nn.Module):                      # the product module refuses CPU tensors: run its parameters through the oracle graph
        def __init__(self, net):
            super().__init__()
            self.net = net
            self.gain = nn.Parameter(torch.ones(2))          # 292824 + 2 parameters: a multiple of neither 4 nor 3 (padding path)

        def forward(self, x):
            return aten_graph.unet3d(self.net, x, self.net.prob) * self.gain.mean()

    torch.manual_seed(0)
    net = CostRegNet(8, 8)                       # a real sub-module of the path (3-D U-Net, BN included)
    torch.manual_seed(1000 + rank)               # the GradSync replica starts from DIFFERENT weights on every rank ...
    ref = Graph(CostRegNet(8, 8))
    gnet = Graph(net)
    ddp = DDP(gnet)
    ddp.register_comm_hook(state=None, hook=flat_allreduce_hook)
    g = torch.Generator().manual_seed(100 + rank)
    x = torch.randn(1, 8, 16, 16, 16, generator=g)
    ddp(x).square().mean().backward()
    sync = GradSync([ref])                       # one flat buffer, .grad are views into it; ... and is broadcast from rank 0
    assert all(p.grad.data_ptr() >= sync.flat.data_ptr() for p in ref.parameters())
    assert sync.flat.numel() % world == 0 and sync.flat.numel() - sync.numel < world
    first = [None] * world
    dist.all_gather_object(first, float(sum(p.double().abs().sum() for p in ref.parameters()) + sum(b.double().abs().sum() for b in ref.buffers())))
    assert all(v == first[0] for v in first), first          # identical replicas after construction
    ref.load_state_dict(gnet.state_dict())
    if world == 3:
        assert sync.flat.numel() != sync.numel               # the padding path really runs
    ref(x).square().mean().backward()
    local = [p.grad.clone() for p in ref.parameters()]
    sync.sync()                                  # must agree with the DDP hook
    err = max(float((a.grad - b.grad).abs().max()) for a, b in zip(gnet.parameters(), ref.parameters()))
    # and with the plain average of the two ranks' local gradients
    gathered = [None] * world
    dist.all_gather_object(gathered, [t.numpy() for t in local])
    mean = [sum(torch.as_tensor(gathered[r][i]) for r in range(world)) / world for i in range(len(local))]
    err = max(err, max(float((p.grad - m).abs().max()) for p, m in zip(ref.parameters(), mean)))
    gsum = float(sum(p.grad.abs().sum() for p in gnet.parameters()))
    sync.zero()
    zeroed = all(float(p.grad.abs().max()) == 0.0 for p in ref.parameters())
    next(ref.parameters()).grad = torch.zeros_like(next(ref.parameters()))
    try:
        sync.sync()
        detected = False
    except RuntimeError:
        detected = True
    q.put((rank, err, gsum, zeroed and detected))
    dist.destroy_process_group()


def _run_ddp(world, compose):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_ddp_worker, args=(r, world, port, q, compose)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=240) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(err < 1e-6 for _, err, _, _ in res), res
    assert all(ok for _, _, _, ok in res), res
    assert all(abs(r[2] - res[0][2]) < 1e-4 * max(1.0, res[0][2]) for r in res)      # same averaged gradients on every rank


def test_flat_allreduce_hook_world2():
    """Gradient exchange over gloo, world size 2: the DDP comm hook and GradSync (one flat buffer for all parameters, .grad
    views into it, replicas broadcast from rank 0 at construction) both leave every rank with the average of the ranks'
    gradients; GradSync.zero() clears through the views and a replaced .grad is detected."""
    _run_ddp(2, compose=False)


def test_direct_allreduce_branch_world2():
    """The branch RCCL takes -- reduce-scatter of the flat message into 1/W shards, all-gather of the reduced shards, and in the
    DDP hook the all-gather chained on the reduce-scatter's future -- executed on gloo with a composed reduce-scatter
    (parallel.COMPOSE_ON_GLOO): same averaged gradients as the plain all_reduce."""
    _run_ddp(2, compose=True)


def test_direct_allreduce_branch_world3_padding():
    """World size 3: neither the flat GradSync buffer nor DDP's bucket is a multiple of the world size, so the padding of the
    message and the un-padding after the all-gather execute."""
    _run_ddp(3, compose=True)


def _syncbn_worker(rank, world, port, q):
    """Two ranks, each with HALF of a batch, run the product's train-mode conv -> SyncBatchNorm -> ReLU block (HIP kernels on
    the CPU emulation; the batch statistics and the backward sums go through train_ops' fp64 all-reduce over gloo)."""
    import sys
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import conftest
    conftest.route_to_emulation(conftest.load_emu_lib(), setattr)
    import torch.nn as nn
    from rc_mvsnet_amd import train_ops
    from rc_mvsnet_amd.casmvsnet import Conv3d
    torch.manual_seed(0)
    blk = nn.SyncBatchNorm.convert_sync_batchnorm(Conv3d(16, 16, padding=1)).train()
    g = torch.Generator().manual_seed(7)
    x_all = torch.randn(2, 4, 8, 8, 16, generator=g)            # channels-last (B,D,H,W,C), one batch item per rank
    G_all = torch.randn(2, 4, 8, 8, 16, generator=g)
    x = x_all[rank:rank + 1].clone().requires_grad_(True)
    z = train_ops.conv_bn_train(blk.conv, blk.bn, x, relu=True)
    (z * G_all[rank:rank + 1]).sum().backward()
    local = (blk.conv.weight.grad.numpy().copy(), blk.bn.weight.grad.numpy().copy())
    # ... and one data-parallel optimizer step on top (config 4: SyncBatchNorm + GradSync + Adam): the gradients of the two ranks are
    # averaged through the flat buffer (copied in: they were produced before the buffer existed), every rank takes the same step
    # (through train_step.make_data_parallel: the setup `bench.py --workload train_step --gpus N` runs on the real node)
    from rc_mvsnet_amd import train_step as ts
    grads = [p.grad.clone() for p in blk.parameters()]
    (blk2,), opt, sync = ts.make_data_parallel([blk], lr=1e-3)
    assert blk2 is blk                            # already converted: convert_sync_batchnorm returns the module itself
    for p, gcopy in zip(blk.parameters(), grads):
        p.grad.copy_(gcopy)
    sync.sync()
    opt.step()
    q.put((rank, z.detach().numpy(), x.grad.numpy(), local[0], local[1],
           blk.bn.running_mean.numpy().copy(), blk.bn.running_var.numpy().copy(),
           blk.conv.weight.detach().numpy().copy(), blk.bn.weight.detach().numpy().copy(), blk.bn.bias.detach().numpy().copy()))
    dist.destroy_process_group()


def test_sync_batchnorm_statistics_world2():
    """SyncBatchNorm-converted block (train_rcmvsnet.py:524-525) over two gloo ranks == the same block with plain BatchNorm on
    the concatenated batch: outputs, input gradients and running statistics per rank; the parameter gradients of the ranks
    sum to the full-batch ones (what the gradient all-reduce then averages)."""
    import numpy as np
    import torch.nn.functional as F
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_syncbn_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted((q.get(timeout=300) for _ in procs), key=lambda t: t[0])
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    # full-batch reference in fp64 with torch ops (same seeds as the workers)
    from rc_mvsnet_amd.casmvsnet import Conv3d
    torch.manual_seed(0)
    blk = Conv3d(16, 16, padding=1).double().train()
    g = torch.Generator().manual_seed(7)
    x_all = torch.randn(2, 4, 8, 8, 16, generator=g)
    G_all = torch.randn(2, 4, 8, 8, 16, generator=g)
    x = x_all.double().permute(0, 4, 1, 2, 3).clone().requires_grad_(True)
    z = torch.relu(blk.bn(blk.conv(x)))
    (z * G_all.double().permute(0, 4, 1, 2, 3)).sum().backward()
    rel = lambda a, b: float(np.abs(a - b).max() / max(1e-30, np.abs(b).max()))
    for rank, zr, gx, gw, gg, rm, rv, *_ in res:
        assert rel(zr[0], z[rank].detach().permute(1, 2, 3, 0).numpy()) < 1e-5
        assert rel(gx[0], x.grad[rank].permute(1, 2, 3, 0).numpy()) < 1e-4
        assert rel(rm, blk.bn.running_mean.numpy()) < 1e-5 and rel(rv, blk.bn.running_var.numpy()) < 1e-4
    assert rel(res[0][3] + res[1][3], blk.conv.weight.grad.numpy()) < 1e-4
    assert rel(res[0][4] + res[1][4], blk.bn.weight.grad.numpy()) < 1e-4
    # the data-parallel step: both ranks hold the same updated weights, equal to ONE single-process Adam step on the two-sample
    # batch with the averaged gradient (what DistributedDataParallel + SyncBatchNorm compute in train_rcmvsnet.py:524-525,565-578)
    w0 = {n: p.detach().clone() for n, p in blk.named_parameters()}
    for p in blk.parameters():
        p.grad = p.grad / 2                                  # mean over the two ranks of the per-rank (per-sample) losses
    torch.optim.Adam(blk.parameters(), lr=1e-3).step()
    for a, b in zip(res[0][7:], res[1][7:]):
        assert np.array_equal(a, b)                           # identical replicas after the step
    got = dict(zip(("conv.weight", "bn.weight", "bn.bias"), res[0][7:]))
    for n, p in blk.named_parameters():
        step_ref = (p.detach() - w0[n]).numpy()
        step_got = got[n] - w0[n].numpy()
        assert np.abs(step_got - step_ref).max() <= 3e-7, (n, np.abs(step_got - step_ref).max())      # fp32 weights of magnitude ~1 vs the fp64 step
