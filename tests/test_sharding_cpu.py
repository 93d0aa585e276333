"""rc_mvsnet_amd/sharding.py without a GPU: the item partition over N x P ranks, the rank -> GPU map, the launcher's environment."""
import os
import subprocess
import sys

import pytest

from rc_mvsnet_amd import sharding

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("gpus,ppg,nitems", [(1, 1, 5), (1, 2, 7), (8, 1, 49), (8, 2, 49), (4, 3, 5), (2, 2, 0)])
def test_every_item_goes_to_exactly_one_rank_and_gpus_are_balanced(gpus, ppg, nitems):
    items = [("scan%d" % (i // 7), i % 7) for i in range(nitems)]
    world = gpus * ppg
    shards = [sharding.shard_items(items, r, world) for r in range(world)]
    assert sorted(sum(shards, [])) == sorted(items)
    assert max(len(s) for s in shards) - min(len(s) for s in shards) <= 1
    per_gpu = [0] * gpus
    for r in range(world):
        g = sharding.device_index(r, ppg)                       # one node: local rank = rank
        assert 0 <= g < gpus
        per_gpu[g] += 1
    assert per_gpu == [ppg] * gpus                              # every GPU hosts exactly P consecutive ranks
    with pytest.raises(ValueError):
        sharding.shard_items(items, world, world)
    with pytest.raises(ValueError):
        sharding.device_index(0, 0)


def test_rank_env_and_clean_env(monkeypatch):
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE"):
        monkeypatch.delenv(k, raising=False)
    assert sharding.rank_env() == (0, 0, 1) and not sharding.launched()
    monkeypatch.setenv("RANK", "5"); monkeypatch.setenv("LOCAL_RANK", "1"); monkeypatch.setenv("WORLD_SIZE", "8"); monkeypatch.setenv("MASTER_PORT", "1")
    assert sharding.rank_env() == (5, 1, 8) and sharding.launched()
    env = sharding.clean_env()
    assert not {"RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT"} & set(env) and "PATH" in env


def test_launch_ranks_starts_n_processes_on_localhost(tmp_path):
    """launch_ranks through torch.distributed.run: N ranks, each sees its RANK / WORLD_SIZE and the 127.0.0.1 rendezvous."""
    script = tmp_path / "who.py"
    script.write_text("import os, sys\n"
                      "open(os.path.join(sys.argv[1], 'rank%s' % os.environ['RANK']), 'w').write(os.environ['WORLD_SIZE'] + ' ' + os.environ['MASTER_ADDR'])\n")
    code = ("import sys; sys.path.insert(0, %r); from rc_mvsnet_amd.sharding import launch_ranks; "
            "sys.exit(launch_ranks(%r, 3, [%r]))" % (ROOT, str(script), str(tmp_path)))
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    assert sorted(p.name for p in tmp_path.glob("rank*")) == ["rank0", "rank1", "rank2"]
    assert (tmp_path / "rank2").read_text() == "3 127.0.0.1"
