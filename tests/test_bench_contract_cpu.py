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
    assert bench.STREAMS_DEFAULT == 1                                                # more streams are experimental (scene_pipeline.py)


def test_command_line_parses_without_a_gpu():
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--help"], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0
    for flag in ("--gpus", "--steps", "--warmup", "--workload", "--streams", "--no-side-pass", "--no-cpu-baseline"):
        assert flag in out.stdout


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
    side = b.get("two_scenes_in_flight")
    if side is not None and "error" not in side:
        assert side["outputs_identical_to_single_stream"] is True and side["value"] > 0
