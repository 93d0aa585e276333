"""GPU, >= 2 devices (skipped on the one-GPU test box): the gradient exchange of the data-parallel training path on RCCL itself
(backend "nccl" = RCCL over xGMI) -- GradSync.sync() and the DDP comm hook against a plain all_reduce, the parameter broadcast at
construction, and the non-divisible padding -- what tests/test_multiproc_cpu.py exercises on gloo with a composed reduce-scatter."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, q):
    import torch.nn as nn
    from torch.nn.parallel import DistributedDataParallel as DDP
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    torch.cuda.set_device(rank)
    dev = torch.device("cuda", rank)
    dist.init_process_group("nccl", rank=rank, world_size=world)
    from rc_mvsnet_amd.parallel import GradSync, flat_allreduce_hook
    torch.manual_seed(100 + rank)                                   # DIFFERENT initial weights per rank: the broadcast must fix that
    net = nn.Sequential(nn.Linear(13, 7), nn.ReLU(), nn.Linear(7, 5), nn.BatchNorm1d(5)).to(dev)      # 13*7+7+7*5+5+5+5 = 148 parameters: not a multiple of 3
    sync = GradSync([net])
    w0 = torch.cat([p.detach().reshape(-1) for p in net.parameters()])
    ref0 = w0.clone()
    dist.broadcast(ref0, src=0)
    same_start = bool(torch.equal(w0, ref0))
    g = torch.Generator().manual_seed(7 + rank)
    x = torch.randn(11, 13, generator=g).to(dev)
    net(x).square().sum().backward()
    mine = sync.flat[:sync.numel].clone()
    want = mine.clone()
    dist.all_reduce(want)
    want /= world
    sync.sync()
    err_sync = float((sync.flat[:sync.numel] - want).abs().max() / want.abs().max())
    # the DDP comm-hook form of the same exchange
    net2 = nn.Sequential(nn.Linear(13, 7), nn.ReLU(), nn.Linear(7, 5)).to(dev)
    ddp = DDP(net2, device_ids=[rank])
    ddp.register_comm_hook(None, flat_allreduce_hook)
    ddp(x).square().sum().backward()
    got = torch.cat([p.grad.reshape(-1) for p in net2.parameters()])
    with torch.no_grad():
        net3 = nn.Sequential(nn.Linear(13, 7), nn.ReLU(), nn.Linear(7, 5)).to(dev)
        net3.load_state_dict(net2.state_dict())
    net3(x).square().sum().backward()
    ref = torch.cat([p.grad.reshape(-1) for p in net3.parameters()])
    dist.all_reduce(ref)
    ref /= world
    err_hook = float((got - ref).abs().max() / ref.abs().max())
    q.put((rank, same_start, err_sync, err_hook))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_gradsync_and_ddp_hook_on_rccl(world):
    if torch.cuda.device_count() < world:
        pytest.skip(f"needs {world} GPUs (the data-parallel path is covered on CPU over gloo: tests/test_multiproc_cpu.py)")
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=300) for _ in procs]
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    for rank, same_start, err_sync, err_hook in res:
        assert same_start, f"rank {rank}: GradSync did not broadcast rank 0's parameters"
        assert err_sync < 1e-6 and err_hook < 1e-6, (rank, err_sync, err_hook)
