"""Multi-GPU inference = independent (scene, reference-view) items, one process per rank, no data-path collective:
each rank takes a round-robin slice of the item list (what eval_rcmvsnet_dtu.py:157-165 iterates serially on one GPU).

Ranks are one per GPU by default; ``procs_per_gpu`` P > 1 puts P worker processes on every GPU (ranks g*P .. g*P+P-1 share GPU g): the items
are independent, so two processes on one GPU overlap each other's latency-bound phases (the deep levels of the cost regularisation run on a
fraction of the CUs) the way two HIP streams would, without sharing an address space -- measured +27 % scenes/s on one MI355X
(profiles/r3_two_streams.txt, two_process_check).  The reference starts its own worker processes (train_rcmvsnet.py:632-636, mp.spawn);
``launch_ranks`` does the same for the evaluation driver and bench.py through torch.distributed.run on 127.0.0.1."""
import os
import socket
import subprocess
import sys

_RANK_ENV = ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "GROUP_RANK", "LOCAL_WORLD_SIZE", "ROLE_RANK",
             "ROLE_WORLD_SIZE", "TORCHELASTIC_RUN_ID")


def shard_items(items, rank, world_size):
    """Round-robin partition: rank r gets items r, r+W, r+2W, ...  (balanced to within one item)."""
    if not (0 <= rank < world_size):
        raise ValueError(f"rank {rank} outside world of size {world_size}")
    return list(items[rank::world_size])


def device_index(local_rank, procs_per_gpu=1):
    """GPU of a local rank when every GPU hosts ``procs_per_gpu`` consecutive ranks."""
    if procs_per_gpu < 1 or local_rank < 0:
        raise ValueError(f"local_rank {local_rank}, procs_per_gpu {procs_per_gpu}")
    return local_rank // procs_per_gpu


def rank_env():
    """(rank, local_rank, world_size) from the launcher's environment; (0, 0, 1) when there is none."""
    return int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))


def launched():
    """True inside a rank started by a launcher (torch.distributed.run exports WORLD_SIZE)."""
    return "WORLD_SIZE" in os.environ


def clean_env(env=None):
    """The environment without a launcher's rank variables (for a child that starts its own ranks)."""
    return {k: v for k, v in (os.environ if env is None else env).items() if k not in _RANK_ENV}


def free_port():
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


# init_process_group's timeout is also the default timeout of every later collective and barrier of the group (gloo and NCCL / RCCL alike): ranks that
# skew by more than this -- one of them building the HIP library for the first time, a slow warm-up, a long CPU baseline on rank 0 -- abort the run
# mid-way.  So the drivers (bench.py, eval_driver) first check that the rendezvous port ANSWERS within RENDEZVOUS_TIMEOUT_S (a readable failure instead of
# a 30-minute hang when MASTER_ADDR / MASTER_PORT are wrong), then create the group with PyTorch's long default.
RENDEZVOUS_TIMEOUT_S = 120
GROUP_TIMEOUT_S = 1800


def _wait_for_master(timeout_s):
    """Ranks other than 0: poll MASTER_ADDR:MASTER_PORT until the store of rank 0 listens (True) or the time is up (False)."""
    import time
    addr, port = os.environ.get("MASTER_ADDR"), os.environ.get("MASTER_PORT")
    if not addr or not port or os.environ.get("RANK", "0") == "0":
        return True
    t0 = time.time()
    while time.time() - t0 < timeout_s:
        try:
            with socket.create_connection((addr, int(port)), timeout=2.0):
                return True
        except OSError:
            time.sleep(0.2)
    return False


def init_process_group(backend):
    """torch.distributed.init_process_group with the drivers' timeout; a failed rendezvous names the backend and the rank."""
    import datetime
    import torch.distributed as dist
    try:
        if not _wait_for_master(RENDEZVOUS_TIMEOUT_S):
            raise TimeoutError("the rendezvous port does not answer")
        dist.init_process_group(backend=backend, timeout=datetime.timedelta(seconds=GROUP_TIMEOUT_S))
    except Exception as e:
        raise SystemExit(f"rank {os.environ.get('RANK', '?')} of {os.environ.get('WORLD_SIZE', '?')}: {backend} rendezvous on "
                         f"{os.environ.get('MASTER_ADDR', '?')}:{os.environ.get('MASTER_PORT', '?')} failed within {RENDEZVOUS_TIMEOUT_S} s: "
                         f"{type(e).__name__}: {e}")
    return dist


def launch_ranks(target, nproc, argv, module=False):
    """Start ``nproc`` ranks of ``target`` (a script path, or a module name with ``module=True``) on this node under
    torch.distributed.run, rendezvous on 127.0.0.1 (the container hostname may not resolve), and return the launcher's exit code."""
    env = clean_env()
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")        # the host driver supports dmabuf IPC only (RCCL, tensor sharing)
    env.setdefault("OMP_NUM_THREADS", "4")
    env.setdefault("NCCL_DEBUG", "WARN")                     # a first contact with N > 1 GPUs should say WHY a rendezvous failed
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={int(nproc)}", "--master-addr", "127.0.0.1",
           "--master-port", str(free_port())]
    cmd += (["-m", target] if module else [os.path.abspath(target)]) + list(argv)
    return subprocess.run(cmd, env=env).returncode
