"""Data-parallel training support (SURVEY.md sections 5, 8e): one process per GPU, the reference's DDP
(`train_rcmvsnet.py:524-525,565-578`), with the gradient exchange shaped for MI355X's point-to-point xGMI.

The two models carry 3.74 MB + 1.7 MB of fp32 gradients: the exchange is latency-bound, so it should be ONE message, and on a
fully connected 8-GPU xGMI node (7 links x ~153 GB/s per GPU, no switch) the cheapest all-reduce of one small message is the
two-step direct form -- reduce-scatter (every rank receives its 1/W slice from the W-1 peers over W-1 links at once), then
all-gather of the reduced slices -- instead of a 2(W-1)-step ring.

`GradSync` owns one flat fp32 buffer holding the gradients of BOTH models (padded to a multiple of the world size) and makes
every parameter's ``.grad`` a view into it, so backward accumulates straight into the message (no per-parameter copies, no
bucketing, no extra launches); ``sync()`` is `div` + `reduce_scatter_tensor` + `all_gather_into_tensor` on that buffer
(RCCL: `backend="nccl"`), or one `all_reduce` where the backend has no reduce-scatter (gloo on the CPU test rig).

    sync = GradSync([model, model_nerf])          # after .to(device); use opt.zero_grad(set_to_none=False)
    loss.backward(); sync.sync(); opt.step()

`flat_allreduce_hook` is the DDP-communication-hook form of the same exchange for code that keeps `DistributedDataParallel`
(one bucket per model with the default 25 MB cap).  The SyncBatchNorm statistics of the converted models
(`nn.SyncBatchNorm.convert_sync_batchnorm`, train_rcmvsnet.py:524-525) are exchanged inside train_ops.ConvBnReluFn
(one fp64 all-reduce per layer forward, one backward).
"""
import torch
import torch.distributed as dist


def _has_reduce_scatter(group):
    return dist.get_backend(group) != "gloo"


def _direct_allreduce(flat, group):
    """In-place sum of `flat` (numel divisible by the world size) over the group: reduce-scatter + all-gather."""
    world = dist.get_world_size(group)
    if world == 1:
        return
    if not _has_reduce_scatter(group):
        dist.all_reduce(flat, group=group)
        return
    shard = torch.empty(flat.numel() // world, dtype=flat.dtype, device=flat.device)
    dist.reduce_scatter_tensor(shard, flat, group=group)
    dist.all_gather_into_tensor(flat, shard, group=group)


class GradSync:
    """One flat gradient buffer for a list of modules + the one-message exchange over it (see the module docstring)."""

    def __init__(self, modules, group=None):
        self.group = group or dist.group.WORLD
        self.world = dist.get_world_size(self.group)
        self.params = [p for m in modules for p in m.parameters() if p.requires_grad]
        if not self.params:
            raise ValueError("GradSync: no trainable parameters")
        dev, dt = self.params[0].device, self.params[0].dtype
        if any(p.device != dev or p.dtype != dt for p in self.params):
            raise ValueError("GradSync: parameters must share one device and dtype")
        n = sum(p.numel() for p in self.params)
        self.numel = n
        self.flat = torch.zeros((n + self.world - 1) // self.world * self.world, dtype=dt, device=dev)
        off = 0
        for p in self.params:
            p.grad = self.flat[off:off + p.numel()].view_as(p)       # backward accumulates into the message itself
            off += p.numel()

    def zero(self):
        self.flat.zero_()

    def sync(self):
        """Average the gradients over the ranks, in place."""
        for p, (a, b) in zip(self.params, self._spans()):
            if p.grad is None or p.grad.data_ptr() != self.flat[a:b].data_ptr():
                raise RuntimeError("GradSync: a parameter's .grad was replaced (use optimizer.zero_grad(set_to_none=False) or GradSync.zero())")
        self.flat.div_(self.world)
        _direct_allreduce(self.flat, self.group)

    def _spans(self):
        off = 0
        for p in self.params:
            yield off, off + p.numel()
            off += p.numel()


def flat_allreduce_hook(state, bucket):
    """DDP comm hook: the bucket's flat buffer averaged with one direct all-reduce (reduce-scatter + all-gather on RCCL)."""
    group = state if isinstance(state, dist.ProcessGroup) else dist.group.WORLD
    world = dist.get_world_size(group)
    buf = bucket.buffer()
    buf.div_(world)
    if _has_reduce_scatter(group) and buf.numel() % world == 0 and world > 1:
        shard = torch.empty(buf.numel() // world, dtype=buf.dtype, device=buf.device)
        fut = dist.reduce_scatter_tensor(shard, buf, group=group, async_op=True).get_future()

        def gather(_):
            dist.all_gather_into_tensor(buf, shard, group=group)
            return buf

        return fut.then(gather)
    fut = dist.all_reduce(buf, group=group, async_op=True).get_future()
    return fut.then(lambda f: f.value()[0])
