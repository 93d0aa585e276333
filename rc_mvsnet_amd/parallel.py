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


# Test hook (tests/test_multiproc_cpu.py): gloo has no reduce_scatter_tensor, so on the CPU rig every exchange takes the plain
# all_reduce branch and the two-step branch -- shard sizes, padding to the world size, the chained future of the DDP hook --
# would never execute before the first RCCL run.  With COMPOSE_ON_GLOO = True the two-step branch runs on gloo as well, its
# reduce-scatter composed from the collectives gloo does have (`_reduce_scatter` below).
COMPOSE_ON_GLOO = False
# Test hook (tests/test_gpu_wrappers.py): a group of ONE rank needs no exchange, so both entry points return early and the one-GPU test box
# would never issue an RCCL collective.  With RUN_AT_WORLD_ONE = True the reduce-scatter + all-gather pair runs on a world-size-1 "nccl"
# group too (a degenerate but real RCCL launch of each).
RUN_AT_WORLD_ONE = False


def _has_reduce_scatter(group):
    return COMPOSE_ON_GLOO or dist.get_backend(group) != "gloo"


class _Done:
    """A completed piece of work with the Work.get_future() surface the DDP hook chains on."""

    def __init__(self, value):
        self._fut = torch.futures.Future()
        self._fut.set_result(value)

    def get_future(self):
        return self._fut

    def wait(self):
        return True


def _reduce_scatter(shard, flat, group, async_op=False):
    """shard <- this rank's 1/W slice of the sum of `flat` over the group.  RCCL: reduce_scatter_tensor.  gloo (test rig only):
    one reduce per slice towards the rank that owns it -- the same result and the same shard arithmetic."""
    if dist.get_backend(group) != "gloo":
        return dist.reduce_scatter_tensor(shard, flat, group=group, async_op=async_op)
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    n = shard.numel()
    for r in range(world):
        piece = flat[r * n:(r + 1) * n].clone()
        dist.reduce(piece, dst=dist.get_global_rank(group, r), group=group)
        if r == rank:
            shard.copy_(piece)
    return _Done([shard]) if async_op else None


def _direct_allreduce(flat, group):
    """In-place sum of `flat` (numel divisible by the world size) over the group: reduce-scatter + all-gather.  Which
    algorithm RCCL runs underneath each of the two collectives (direct / ring / tree over the xGMI links) is RCCL's choice
    (NCCL_ALGO / its tuner); nothing here controls it."""
    world = dist.get_world_size(group)
    if world == 1 and not RUN_AT_WORLD_ONE:
        return
    if not _has_reduce_scatter(group):
        dist.all_reduce(flat, group=group)
        return
    if flat.numel() % world:
        raise ValueError(f"_direct_allreduce: {flat.numel()} elements do not split over {world} ranks (pad the buffer)")
    shard = torch.empty(flat.numel() // world, dtype=flat.dtype, device=flat.device)
    _reduce_scatter(shard, flat, group)
    dist.all_gather_into_tensor(flat, shard, group=group)


class GradSync:
    """One flat gradient buffer for a list of modules + the one-message exchange over it (see the module docstring)."""

    def __init__(self, modules, group=None, broadcast=True):
        self.group = group or dist.group.WORLD
        self.world = dist.get_world_size(self.group)
        if broadcast and self.world > 1:
            # what DistributedDataParallel does at construction (train_rcmvsnet.py:565-578 relies on it): every replica starts
            # from rank 0's parameters and buffers, as ONE flat message per dtype
            src = dist.get_global_rank(self.group, 0)
            with torch.no_grad():
                by_dtype = {}
                for m in modules:
                    for t in list(m.parameters()) + list(m.buffers()):
                        by_dtype.setdefault((t.dtype, t.device), []).append(t)
                for ts in by_dtype.values():
                    flat = torch.cat([t.detach().reshape(-1) for t in ts])
                    dist.broadcast(flat, src=src, group=self.group)
                    off = 0
                    for t in ts:
                        t.copy_(flat[off:off + t.numel()].view_as(t))
                        off += t.numel()
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
        self._offsets = []
        for p in self.params:
            p.grad = self.flat[off:off + p.numel()].view_as(p)       # backward accumulates into the message itself
            self._offsets.append(off)
            off += p.numel()

    def zero(self):
        self.flat.zero_()

    def sync(self):
        """Average the gradients over the ranks, in place."""
        base, esz = self.flat.data_ptr(), self.flat.element_size()
        for p, off in zip(self.params, self._offsets):          # pointer arithmetic only: no tensor slices per step
            if p.grad is None or p.grad.data_ptr() != base + off * esz:
                raise RuntimeError("GradSync: a parameter's .grad was replaced (use optimizer.zero_grad(set_to_none=False) or GradSync.zero())")
        self.flat.div_(self.world)
        _direct_allreduce(self.flat, self.group)


def flat_allreduce_hook(state, bucket):
    """DDP comm hook: the bucket's flat buffer averaged with one direct all-reduce (reduce-scatter + all-gather on RCCL)."""
    group = state if isinstance(state, dist.ProcessGroup) else dist.group.WORLD
    world = dist.get_world_size(group)
    buf = bucket.buffer()
    buf.div_(world)
    if _has_reduce_scatter(group) and (world > 1 or RUN_AT_WORLD_ONE):
        # DDP's bucket size is whatever its parameters add up to: pad the message to a multiple of the world size
        n = buf.numel()
        padded = (n + world - 1) // world * world
        msg = buf if padded == n else torch.cat([buf, buf.new_zeros(padded - n)])
        shard = torch.empty(padded // world, dtype=buf.dtype, device=buf.device)
        fut = _reduce_scatter(shard, msg, group, async_op=True).get_future()

        def gather(_):
            dist.all_gather_into_tensor(msg, shard, group=group)
            if msg is not buf:
                buf.copy_(msg[:n])
            return buf

        return fut.then(gather)
    fut = dist.all_reduce(buf, group=group, async_op=True).get_future()
    return fut.then(lambda f: f.value()[0])
