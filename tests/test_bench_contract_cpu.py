"""bench.py on a machine without a GPU: the module imports, its pure helpers give SURVEY.md section 8d's figures, the command line
parses, and the committed line of the round's closing run (profiles/) carries every field of the driver's contract with consistent values."""
import glob
import importlib.util
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def bench():
    spec = importlib.util.spec_from_file_location("bench_under_test", os.path.join(ROOT, "bench.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def test_k1_algorithmic_bytes_are_the_survey_figures(bench):
    per_stage = bench.k1_algorithmic_bytes()
    assert [round(b / 1e6, 1) for b in per_stage] == [137.6, 194.0, 125.8]
    assert sum(per_stage) == 457441280


def test_command_line_parses_without_a_gpu():
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--help"], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0
    for flag in ("--gpus", "--steps", "--warmup", "--workload", "--no-side-pass", "--no-cpu-baseline", "--procs-per-gpu"):
        assert flag in out.stdout


def _clean_env():
    return {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}


def _json_line(stdout):
    lines = [ln for ln in stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, stdout                       # ONE line, from rank 0
    return json.loads(lines[0])


@pytest.mark.parametrize("gpus,ppg", [(2, 1), (1, 2), (2, 2)])
def test_bench_starts_its_own_ranks(gpus, ppg):
    """`python bench.py --gpus N` the way the driver invokes N = 1 (no torch.distributed.run in front, no WORLD_SIZE): bench.py starts the
    N x P ranks itself; the stub workload runs the shared plumbing (rendezvous on 127.0.0.1, barrier-bracketed rounds of exactly K steps,
    max over ranks, one JSON line from rank 0) on CPU + gloo."""
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--workload", "stub", "--gpus", str(gpus), "--procs-per-gpu", str(ppg),
                          "--steps", "40", "--warmup", "3"], capture_output=True, text=True, timeout=300, env=_clean_env())
    assert out.returncode == 0, out.stderr[-2000:]
    b = _json_line(out.stdout)
    assert b["n_gpus"] == gpus and b["config"]["ranks"] == gpus * ppg and b["config"]["procs_per_gpu"] == ppg
    assert b["steps"] == 40 and b["warmup"] == 3 and b["scaling"] == "weak" and b["higher_is_better"] is True
    assert abs(b["value"] * b["ms_per_step"] * 1e-3 - gpus * ppg) < 1e-2 * gpus * ppg                 # value = ranks * K / median round
    # a 40-step round of the stub is far below 0.5 s on an idle box: the region is then repeated in rounds of exactly K steps and the
    # median is reported (how many rounds fit depends on the machine's load: only the bookkeeping is asserted)
    assert 1 <= b["timed_rounds"] <= 4000 and b["round_s_min_max"][0] <= b["timed_region_s"] <= b["round_s_min_max"][1]
    assert b["steps_run_by_rank0"] == 3 + 40 * b["timed_rounds"]


def test_bench_under_torch_distributed_run_still_works():
    """The driver's N > 1 form: torch.distributed.run in front, RANK / WORLD_SIZE in the environment -- no second spawn."""
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                          "--master-port", "29731", os.path.join(ROOT, "bench.py"), "--workload", "stub", "--gpus", "2", "--steps", "30",
                          "--warmup", "1"], capture_output=True, text=True, timeout=300, env=_clean_env())
    assert out.returncode == 0, out.stderr[-2000:]
    b = _json_line(out.stdout)
    assert b["n_gpus"] == 2 and b["config"]["ranks"] == 2 and b["steps"] == 30


def test_world_size_mismatch_is_an_error_message_not_a_crash():
    env = dict(_clean_env(), WORLD_SIZE="3", RANK="0", LOCAL_RANK="0")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--workload", "stub", "--gpus", "2"], capture_output=True, text=True,
                         timeout=120, env=env)
    assert out.returncode != 0 and "WORLD_SIZE=3" in out.stderr and "Traceback" not in out.stderr


def _latest_line():
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_run*_bench.json")),
                   key=lambda p: [int(x) for x in os.path.basename(p).replace("r", "", 1).replace("_run", " ").replace("_bench.json", "").split()])
    return files[-1], json.load(open(files[-1]))


def test_committed_closing_line_meets_the_contract():
    path, b = _latest_line()
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype",
              "data", "config", "roofline", "cpu_baseline"):
        assert k in b, (path, k)
    assert b["unit"] == "ref-scenes/s" and b["higher_is_better"] is True and b["scaling"] == "weak" and b["data"] == "synthetic"
    assert b["dtype"] == "f32" and b["vs_baseline"] is None and "workload" in b["config"] and "model" not in b["config"]
    assert abs(b["value"] * b["ms_per_step"] * 1e-3 - b["n_gpus"]) < 1e-3 * b["n_gpus"]            # value = N * K / time
    r = b["roofline"]
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3 and 0.0 < r["frac"] < 1.0
    assert r["traffic"] is None or 0.9 < r["traffic"] / r["algorithmic_bytes_per_scene"] < 1.2     # nothing re-read from HBM
    c = b["cpu_baseline"]
    assert c["kind"] in ("port", "reference") and c["cores"] >= 1 and c["value"] > 0 and c["sample"]

