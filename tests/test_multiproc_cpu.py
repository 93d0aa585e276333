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


def _ddp_worker(rank, world, port, q):
    import torch.nn as nn
    from torch.nn.parallel import DistributedDataParallel as DDP
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from rc_mvsnet_amd.parallel import flat_allreduce_hook, allreduce_gradients
    from rc_mvsnet_amd.casmvsnet import CostRegNet
    torch.manual_seed(0)
    net = CostRegNet(8, 8)                       # a real sub-module of the path (3-D U-Net, BN included)
    ref = CostRegNet(8, 8)
    ref.load_state_dict(net.state_dict())
    ddp = DDP(net)
    ddp.register_comm_hook(state=None, hook=flat_allreduce_hook)
    g = torch.Generator().manual_seed(100 + rank)
    x = torch.randn(1, 8, 16, 16, 16, generator=g)
    ddp(x).square().mean().backward()
    ref(x).square().mean().backward()
    allreduce_gradients([ref])                   # manual flat all-reduce must agree with the hook
    err = max(float((a.grad - b.grad).abs().max()) for a, b in zip(net.parameters(), ref.parameters()))
    gsum = float(sum(p.grad.abs().sum() for p in net.parameters()))
    q.put((rank, err, gsum))
    dist.destroy_process_group()


def test_flat_allreduce_hook_world2():
    """DDP over gloo, world size 2: the flat-buffer comm hook averages gradients across ranks (every rank
    ends with identical gradients) and matches the manual flat all-reduce."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_ddp_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(err < 1e-6 for _, err, _ in res), res
    assert abs(res[0][2] - res[1][2]) < 1e-4 * max(1.0, res[0][2])      # same averaged gradients on both ranks
