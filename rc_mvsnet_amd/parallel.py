"""Data-parallel training support (SURVEY.md sections 5, 8e): one process per GPU, the reference's DDP
(`train_rcmvsnet.py:565-578`), with the gradient exchange reshaped for MI355X's point-to-point xGMI.

The two models carry 3.74 MB + 1.7 MB of fp32 gradients -- latency-bound, and DDP's default 25 MB buckets
already give one bucket per model.  `flat_allreduce_hook` is a DDP communication hook that keeps it to
exactly ONE collective per bucket on the bucket's own flat buffer (no per-parameter copies, no extra
launches) and divides once; with `average_on_device=True` the division is folded into the same
in-place op.  `backend="nccl"` is RCCL on ROCm; on CPU test rigs the same hook runs over gloo.

    model = DDP(model.to(rank), device_ids=[rank])
    model.register_comm_hook(state=None, hook=flat_allreduce_hook)
"""
import torch
import torch.distributed as dist


def flat_allreduce_hook(state, bucket):
    """DDP comm hook: one all-reduce over the bucket's flat gradient buffer, averaged over the world."""
    group = state if isinstance(state, dist.ProcessGroup) else dist.group.WORLD
    world = dist.get_world_size(group)
    buf = bucket.buffer()
    buf.div_(world)                                   # pre-divide: keeps the sum in range, one elementwise op
    fut = dist.all_reduce(buf, group=group, async_op=True).get_future()
    return fut.then(lambda f: f.value()[0])


def allreduce_gradients(modules, group=None):
    """Manual variant for loops that do not use DDP: flatten every gradient of `modules` into one buffer,
    all-reduce once, scatter back (the reference's 5.4 MB total fits one message)."""
    group = group or dist.group.WORLD
    grads = [p.grad for m in modules for p in m.parameters() if p.grad is not None]
    if not grads:
        return
    flat = torch.cat([g.reshape(-1) for g in grads])
    flat.div_(dist.get_world_size(group))
    dist.all_reduce(flat, group=group)
    off = 0
    for g in grads:
        n = g.numel()
        g.copy_(flat[off:off + n].view_as(g))
        off += n
