"""GPU suite (-m gpu): bench.py's one-line contract, on a small batch (the driver runs the full size).

One JSON object on stdout and nothing else: the keys the driver reads, a roofline object whose fraction is formed on the
bytes the resident layout has to read and never exceeds 1, the secondary workloads, the CPU baseline."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(*extra):
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--tracks", "640", "--rows", "800", "--steps", "5", "--warmup", "2", *extra],
                       capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [l for l in p.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, p.stdout[-2000:]                       # ONE line on stdout
    return json.loads(lines[0])


def test_bench_line_has_what_the_driver_reads():
    d = _run()
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 5 and d["warmup"] == 2 and d["higher_is_better"] is True
    assert d["scaling"] == "strong" and d["vs_baseline"] is None and d["dtype"] == "f64" and d["data"] == "synthetic"
    assert d["unit"] == "track-timesteps/s" and d["value"] > 0 and d["ms_per_step"] > 0
    assert abs(d["value"] - 640 * 800 * 5 / (d["ms_per_step"] * 5e-3)) <= 1e-6 * d["value"]
    assert "workload" in d["config"] and "model" not in d["config"]
    r = d["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in r, k
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0 and r["traffic"] is None
    assert 0 < r["frac"] <= 1.0 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-12
    assert r["required_bytes_per_row"] == 16.0 and r["algo_bytes_per_row"] == 24.0 and r["kernel_ms"] > 0
    assert d["ms_per_step_stamped"] > 0 and "strong scaling" in d["config"]["workload"] and d["config"]["total_rows"] == 640 * 800
    c = d["cpu_baseline"]
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in c, k
    assert c["kind"] in ("port", "reference") and c["value"] > 0 and c["cores"] >= 1
    assert len(d["secondary"]) == 6 and all(s["value"] > 0 and s["window_check_max"] <= 1e-11 for s in d["secondary"])
    assert "H_array" in d["secondary"][5]["workload"] and d["secondary"][5]["path"] == "isotropic-register"     # the one-wave full-covariance kernel
    assert "tau and nu smooth" in d["secondary"][4]["workload"] and d["secondary"][4]["path"] == "isotropic-register"   # the lane = track kernel
    assert "one missing row" in d["secondary"][2]["workload"] and d["secondary"][2]["quiet_window"] > 0             # quiet rows of the general kernel


def test_one_rank_communicator_rehearsal_prints_one_line_too():
    env = dict(os.environ, SSDE_BENCH_SELF_LAUNCH="1", SSDE_BENCH_FORCE_COMM="1")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--tracks", "640", "--rows", "800", "--steps", "4", "--warmup", "1",
                        "--no-cpu-baseline", "--no-secondary"], capture_output=True, text=True, timeout=600, env=env)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [l for l in p.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, p.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 1 and "ncclAllReduce" in d["config"]["parallelism"]


def test_strong_and_weak_modes_with_one_rank_time_the_same_batch():
    """One rank: both modes cut nothing; the line says which one ran and both see the same batch (same nllk at the last step)."""
    a = _run("--no-cpu-baseline", "--no-secondary", "--scaling", "strong")
    b = _run("--no-cpu-baseline", "--no-secondary", "--scaling", "weak")
    assert a["scaling"] == "strong" and b["scaling"] == "weak"
    assert a["config"]["nllk_at_last_step"] == b["config"]["nllk_at_last_step"]
    assert a["config"]["total_rows"] == b["config"]["total_rows"] == 640 * 800


def test_config_c4_and_c5_lines():
    d = _run("--no-cpu-baseline", "--config", "c4")
    assert d["config"]["config"] == "c4" and "C4" in d["config"]["workload"] and 0 < d["roofline"]["frac"] <= 1.0
    d = _run("--no-cpu-baseline", "--config", "c5")
    c = d["config"]
    assert c["config"] == "c5" and "concurrently" in c["workload"] and len(c["n_free_par"]) == 3
    # ragged lengths U[T/2, T]: three sub-batches of 640 tracks
    assert 3 * 640 * 400 <= c["total_rows"] <= 3 * 640 * 800
    hs = d["roofline"]["handles"]
    assert [h["kernel"].split("(")[1].rstrip(")") for h in hs] == ["BM_SSM", "OU_SSM", "CTCRW"]
    assert all(0 < h["frac"] <= 1.0 and h["kernel_ms"] > 0 for h in hs)
    assert c["window_check"] <= 1e-11 and d["value"] > 0
    assert abs(d["value"] - c["total_rows"] * 5 / (d["ms_per_step"] * 5e-3)) <= 1e-6 * d["value"]
