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
