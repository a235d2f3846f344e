"""GPU suite (-m gpu): bench.py's one-line contract, on a small batch (the driver runs the full size).

One JSON object on stdout and nothing else: the keys the driver reads, a roofline object whose fraction is formed on the
bytes the resident layout has to read and never exceeds 1, the secondary workloads, the CPU baseline."""
import json

import numpy as np
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
    # (`traffic` is the PMC figure of the committed profile of THIS command at ITS size -- null for any other launch, such as this small one)
    assert r["bound"] in ("hbm", "fp64_issue") and r["unit"] == "GB/s" and r["peak"] == 8000.0 and r["traffic"] is None
    assert 0 < r["frac"] <= 1.0 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-12
    # two roofs: the HBM fraction and the fp64-issue fraction side by side, neither above 1; `bound` names the larger
    assert r["frac_hbm"] == r["frac"] and r["kernel"].startswith("iso_shared_kernel") and r["kernel_id"] == 3
    assert r["frac_fp64_issue"] is None or 0 < r["frac_fp64_issue"] <= 1.0
    if r["frac_fp64_issue"] is not None:
        assert r["bound"] == ("fp64_issue" if r["frac_fp64_issue"] > r["frac_hbm"] else "hbm") and "SQ_INSTS_VALU" in r["valu_source"]
    # the line carries its own spread, and the per-rank account exists at N = 1 too
    ex = d["extra"]
    assert 0 < ex["ms_per_step_min"] <= ex["ms_per_step_median"] <= ex["ms_per_step_p90"] <= ex["ms_per_step_max"]
    pr = d["per_rank"]
    assert len(pr) == 1 and pr[0]["rank"] == 0 and pr[0]["rows"] == 640 * 800 and pr[0]["kernel_ms"] > 0
    assert pr[0]["finalize_ms"] > 0 and pr[0]["allreduce_wait_ms"] == 0.0 and pr[0]["host_total_ms"] >= pr[0]["kernel_stamp_ms"] > 0
    assert d["config"]["comm_ranks_reported"] == 0 and d["config"]["engine_kernel"] == "iso_shared_kernel"
    assert r["required_bytes_per_row"] == 16.0 and r["algo_bytes_per_row"] == 24.0 and r["kernel_ms"] > 0
    assert d["ms_per_step_stamped"] > 0 and "strong scaling" in d["config"]["workload"] and d["config"]["total_rows"] == 640 * 800
    c = d["cpu_baseline"]
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in c, k
    assert c["kind"] in ("port", "reference") and c["value"] > 0 and c["cores"] >= 1
    assert len(d["secondary"]) == 10 and all(s["value"] > 0 and s["window_check_max"] <= 1e-11 for s in d["secondary"])
    for s in d["secondary"]:                                       # every entry names its kernel and carries both roofs, each <= 1
        assert s["kernel"] != "?", s
        if s["kernel"] == "tv_filter_kernel":                      # replayed from a hipGraph: no event pair around the launch, no roofs
            assert s["bound"] is None and s["frac_hbm"] is None, s
            continue
        assert s["bound"] in ("hbm", "fp64_issue") and 0 < s["frac_hbm"] <= 1.0, s
        assert s["frac_fp64_issue"] is None or 0 < s["frac_fp64_issue"] <= 1.0, s
    kern = [s["kernel"] for s in d["secondary"]]
    assert kern[0] == "iso_mask_kernel" and kern[2] == "iso_quiet_kernel" and kern[5] == "iso_full_kernel", kern
    assert "H_array" in d["secondary"][5]["workload"] and d["secondary"][5]["path"] == "isotropic-register"     # the one-wave full-covariance kernel
    # tau and nu smooth: 640 x 80 rows x 32 lanes per track is below the rows rule of ssde_create (4.5e6 lane-rows): the lane = direction
    # path here, the eight-wave pipeline at the default size (tests/test_gpu_colvar.py pins the rule on both sides)
    assert "tau and nu smooth" in d["secondary"][4]["workload"] and kern[4] in ("tv_filter_kernel", "iso_colvar_kernel", "iso_few_kernel", "iso_adj_kernel"), kern
    # BASELINE configurations 1, 2, 3 (+ 3 with the block evaluated from its table): driver-timed entries with their kernels
    assert [s.get("baseline_config") for s in d["secondary"][6:]] == [1, 2, 3, 3] and d["secondary"][9]["form"] == "table"
    assert kern[6] == "tv_filter_kernel" and kern[7] == "iso_shared_kernel" and kern[8] == "direct_fast_kernel" and kern[9] == "direct_fast_kernel", kern
    # (regular grid: no time stamps resident -- 8 B of observation + 72 B of columns; from the table: the observation and the covariate)
    assert d["secondary"][8]["required_bytes_per_row"] == 80.0 and d["secondary"][8]["algo_bytes_per_row"] == 88.0
    assert d["secondary"][9]["required_bytes_per_row"] <= 24.0
    assert d["secondary"][4]["path"] == ("isotropic-row-varying" if kern[4] == "tv_filter_kernel" else "isotropic-register")
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
    # what makes the first real N > 1 run diagnosable (VERDICT r03 #1b): RCCL's own count of the communicator, and per rank
    # the kernel, the finalising launch, the wait inside the all-reduce and the read-back
    assert d["config"]["comm_ranks_requested"] == 1 and d["config"]["comm_ranks_reported"] == 1
    pr = d["per_rank"]
    assert len(pr) == 1 and pr[0]["comm_ranks_reported"] == 1
    for k in ("rows", "kernel_ms", "finalize_ms", "allreduce_wait_ms", "readback_ms", "host_total_ms"):
        assert k in pr[0], k
    assert pr[0]["allreduce_wait_ms"] > 0 and pr[0]["kernel_ms"] > 0 and pr[0]["finalize_ms"] > 0


def test_strong_scaling_line_carries_the_weak_scaling_reading_of_the_same_launch():
    """at N > 1 (rehearsed here with ONE rank and a real one-rank communicator) the line of the default, strong-scaling run carries
    a `secondary` entry that times the weak-scaling batch -- every rank a whole configuration's worth of tracks -- through the same
    engines and collective; its failure modes end in an entry with `error`, never in a missing line"""
    env = dict(os.environ, SSDE_BENCH_SELF_LAUNCH="1", SSDE_BENCH_FORCE_COMM="1", SSDE_BENCH_WEAK_PROBE="1")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--tracks", "640", "--rows", "800", "--steps", "4", "--warmup", "1",
                        "--no-cpu-baseline"], capture_output=True, text=True, timeout=600, env=env)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [l for l in p.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, p.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["scaling"] == "strong"
    w = d["secondary"][0]
    assert "error" not in w, w
    assert w["scaling"] == "weak" and w["n_gpus"] == 1 and w["rows_total"] == 640 * 800 and w["value"] > 0 and w["comm_ranks_reported"] == 1
    assert "ncclAllReduce" in w["workload"]
    assert abs(w["nllk_at_last_step"]) > 0 and np.isfinite(w["nllk_at_last_step"])


def test_c5_with_a_communicator_sums_the_three_handles_in_one_collective():
    env = dict(os.environ, SSDE_BENCH_SELF_LAUNCH="1", SSDE_BENCH_FORCE_COMM="1")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--tracks", "640", "--rows", "800", "--steps", "4", "--warmup", "1",
                        "--no-cpu-baseline", "--config", "c5"], capture_output=True, text=True, timeout=600, env=env)
    assert p.returncode == 0, p.stderr[-2000:]
    d = json.loads([l for l in p.stdout.splitlines() if l.strip()][0])
    assert "ONE ncclAllReduce" in d["config"]["parallelism"] and d["config"]["comm_ranks_reported"] == 1
    # the same batch without a communicator: same numbers (a one-rank sum changes nothing)
    q = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--tracks", "640", "--rows", "800", "--steps", "4", "--warmup", "1",
                        "--no-cpu-baseline", "--config", "c5"], capture_output=True, text=True, timeout=600)
    assert q.returncode == 0, q.stderr[-2000:]
    e = json.loads([l for l in q.stdout.splitlines() if l.strip()][0])
    assert d["config"]["nllk_at_last_step"] == e["config"]["nllk_at_last_step"] and d["config"]["total_rows"] == e["config"]["total_rows"]


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
