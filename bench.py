#!/usr/bin/env python3
"""bench.py -- track-timesteps/s of one nllk + gradient evaluation (BASELINE.json metric).

A "step" is one evaluation of the hot path (value + full gradient, one ssde_eval_device, the
RCCL all-reduce of the 1+p doubles when N > 1, and the D2H of the result) over one resident
batch of synthetic tracks, each step at a different parameter vector.  Workload at every N:
10^4 two-dimensional CTCRW tracks x 10^4 rows PER GPU (constant coefficients, sigma_obs free,
mu fixed as in the vignette) -- weak scaling: tracks shard over ranks with no data-path
collective other than the scalar all-reduce.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured copy)


def _usable_cores():
    """Cores this process may use: the affinity mask, capped by the cgroup CPU quota when one is set."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        q, p = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            n = min(n, max(1, int(float(q) / float(p))))
    except (OSError, ValueError):
        pass
    try:
        q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
        p = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
        if q > 0 and p > 0:
            n = min(n, max(1, q // p))
    except (OSError, ValueError):
        pass
    return n


def cpu_baseline(seconds_target=15.0):
    """The CPU oracle (port of the reference arithmetic, forward-mode gradient) on a bounded
    sample of the same workload, on all host cores of this box.  Baseline, not the target."""
    from oracle_lib import oracle_eval
    from smoothsde_amd import capi
    from smoothsde_amd.synth import simulate
    cores = _usable_cores()
    tracks, rows = 8 * cores, 2000
    ID, times, obs = simulate("CTCRW", tracks, rows, 2, tau=2.0, nu=1.0, sigma_obs=0.1, seed=1)
    fixed = np.array([0, 1, 1, 0, 0], dtype=np.uint8)
    pb = capi.Problem("CTCRW", ID, times, obs, par_fixed=fixed)
    par = np.array([0.0, 0.0, 0.0, 0.0, 0.0])
    # the affinity mask can promise more cores than the box's CPU share grants: measure how many
    # cores the threads really got (process CPU time / wall time) and size the pool to that
    oracle_eval(pb, par, order=1, threads=cores)  # warm (library load, first-touch)
    w0, c0 = time.perf_counter(), time.process_time()
    oracle_eval(pb, par, order=1, threads=cores)
    busy = (time.process_time() - c0) / max(time.perf_counter() - w0, 1e-9)
    if busy < 0.6 * cores:
        cores = max(1, int(round(busy)))
    t0 = time.perf_counter()
    reps = 0
    while True:
        oracle_eval(pb, par + 0.01 * reps, order=1, threads=cores)
        reps += 1
        el = time.perf_counter() - t0
        if el > seconds_target or reps >= 50:
            break
    rate = tracks * rows * reps / el
    return {"value": rate, "unit": "track-timesteps/s", "cores": cores, "kind": "port",
            "sample": f"{tracks} CTCRW tracks x {rows} rows, {reps} nllk+grad evaluations, "
                      f"oracle/liboracle.so (g++ -O2, {cores} threads over track shards)"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--tracks", type=int, default=10_000, help="tracks per GPU")
    ap.add_argument("--rows", type=int, default=10_000, help="rows per track")
    ap.add_argument("--model", default="CTCRW")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if rank == 0:
            print(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}; launch with torch.distributed.run",
                  file=sys.stderr)
        if args.gpus > 1 and world == 1:
            sys.exit(2)
    # rehearsal on a one-GPU box (never used by the driver): SSDE_BENCH_SHARE_GPU=1 puts every rank on cuda:0 and
    # SSDE_BENCH_BACKEND=gloo carries the all-reduce (RCCL refuses two ranks on one device)
    if os.environ.get("SSDE_BENCH_SHARE_GPU"):
        local_rank = 0
    backend = os.environ.get("SSDE_BENCH_BACKEND", "nccl")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    from smoothsde_amd import capi
    from smoothsde_amd.synth import simulate

    M, T, d = args.tracks, args.rows, 2
    # synthetic batch built directly in HBM (SURVEY.md 8(d) C2': tau=2, nu=1, mu=0, sigma_obs=0.1, dt=1)
    ID, times, obs = simulate(args.model, M, T, d, mu=0.0, tau=2.0, nu=1.0, kappa=1.0, sigma=1.0, sigma_obs=0.1,
                              seed=1 + rank, backend="torch", device=dev)
    q = capi.n_sde_par(args.model, d)
    fixed = np.zeros(1 + q, dtype=np.uint8)
    fixed[1:1 + d] = 1  # fixpar = c("mu1","mu2") as in the vignette (smoothSDE.rmd:486-490)
    pb = capi.Problem.from_torch(args.model, ID, times, obs, par_fixed=fixed)
    eng = capi.Engine(pb)
    del ID, times, obs
    info = eng.info()
    npar = pb.n_par_full
    out = torch.zeros(2 + npar, dtype=torch.float64, device=dev)  # [nllk, grad..., window_check]
    stream = torch.cuda.current_stream(dev)
    out_pinned = torch.zeros(2 + npar, dtype=torch.float64).pin_memory()

    def theta(k):
        base = np.zeros(npar)
        base[0] = np.log(0.1)            # log sigma_obs
        base[1 + d] = np.log(2.0)        # log tau
        if q > d + 1:
            base[2 + d] = 0.0            # log nu
        return base + 0.01 * np.sin(np.arange(npar) + 0.7 * k)

    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]

    host_enqueue_s = []

    thetas = {k: theta(k) for k in range(-args.warmup - 1, args.steps)}   # built outside the timed region

    # N = 1: the synchronous C-ABI call the reference's fn/gr pair maps to (ssde_eval: kernels, hand-over check,
    # reduction, result in host memory when it returns).  N > 1: ssde_eval_device + RCCL all-reduce + D2H.
    sync_api = world == 1 and not os.environ.get("SSDE_BENCH_ASYNC")

    last_info = [None]
    d2h_blocking = os.environ.get("SSDE_BENCH_D2H", "blocking") == "blocking"   # ("pinned": async copy + stream sync, 2-3 % slower)

    def step(k, events=None, force_async=False):
        th = thetas[k]
        if sync_api and not force_async:
            val, grad = eng.eval(th, order=1)
            last_info[0] = eng.info()
            return np.concatenate([[val], grad, [last_info[0]["window_check"]]])
        if events:
            events[0].record(stream)
        t_h = time.perf_counter()
        eng.eval_device(th, out.data_ptr(), order=1, stream=stream.cuda_stream)
        host_enqueue_s.append(time.perf_counter() - t_h)
        if events:
            events[1].record(stream)
        if world > 1:
            dist.all_reduce(out)          # RCCL sum of [nllk, grad] over xGMI: 1+p doubles
        if d2h_blocking:
            return out.cpu().numpy()                 # blocking D2H (one hipMemcpy, its own synchronisation)
        out_pinned.copy_(out, non_blocking=True)   # D2H of the result into pinned memory ...
        stream.synchronize()                         # ... and the one synchronisation of the step
        return out_pinned.numpy().copy()

    for k in range(args.warmup):
        res = step(-1 - k)
        for _ in range(4):  # overlapping time windows must agree (k_iso.hip); widen the overlap if not
            if res[-1] <= capi.WINDOW_TOL * world or os.environ.get("SSDE_DIAG_TIMING_ONLY"):
                break
            eng.widen_windows(4)
            res = step(-1 - k)
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    main_ms = []
    for k in range(args.steps):
        res = step(k, None if sync_api else ev[k])
        main_ms.append((last_info[0] if sync_api else eng.info())["main_kernel_ms"])   # HIP events around the dominant kernel, on its own stream
    torch.cuda.synchronize(dev)
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        tmax = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())
    if sync_api:   # GPU time of a whole evaluation: the same evaluations again, asynchronously, between HIP events
        for k in range(args.steps):
            step(k, ev[k], force_async=True)
        torch.cuda.synchronize(dev)
    if not os.environ.get("SSDE_DIAG_TIMING_ONLY"):   # (timing-only diagnostic kernels produce wrong numbers)
        assert np.all(np.isfinite(res)), res
        assert res[-1] <= capi.WINDOW_TOL * world, f"window hand-over check failed: {res[-1]}"
    info = eng.info()

    rows_per_gpu = info["n_rows"]
    total_rows = rows_per_gpu * world
    value = total_rows * args.steps / elapsed
    eval_ms = float(np.mean([a.elapsed_time(b) for a, b in ev]))      # all kernels of one evaluation
    kern_ms = float(np.mean(main_ms))                                   # the dominant kernel alone
    algo_bytes = info["algo_bytes_per_row"] * info["main_kernel_rows"]  # bytes of the rows that launch scores
    achieved = algo_bytes / (kern_ms * 1e-3) / 1e9
    eval_achieved = info["algo_bytes_per_row"] * rows_per_gpu / (eval_ms * 1e-3) / 1e9
    traffic = traffic_eval = None
    pmc = os.path.join(ROOT, "profiles", "pmc_latest.json")
    if os.path.exists(pmc):
        try:
            pj = json.load(open(pmc))
            traffic = pj.get("main_kernel_bytes")          # PMC bytes of the dominant kernel (profiles/pmc_latest.json)
            traffic_eval = pj.get("hbm_bytes_per_launch")  # ... and of all kernels of one evaluation
        except Exception:
            traffic = None
    line = {
        "metric": "track-timesteps/s nllk+grad",
        "value": value, "unit": "track-timesteps/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": 1e3 * elapsed / args.steps,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f64", "data": "synthetic",
        "config": {"workload": f"{M} {args.model} tracks x {T} rows per GPU, d=2, constant coefficients, "
                               f"sigma_obs/tau/nu free, mu fixed, dt=1 (SURVEY 8(d) C2')",
                   "tracks_per_gpu": M, "rows_per_track": T, "n_free_par": info["n_free"],
                   "engine_path": capi.PATH_NAMES[info["path"]],
                   "uniform_dt": info["uniform_dt"], "workgroups": info["n_kernel_blocks"],
                   "lanes_per_track": info["lanes_per_track"], "window_rows": info["window"],
                   "window_check": float(res[-1]), "parallelism": f"tracks x{world}"},
        "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                     "kernel": "iso_shared_kernel<stationary>" if info["uniform_dt"] else "iso_kernel",
                     "kernel_ms": kern_ms, "algo_bytes_per_launch": algo_bytes,
                     "rows_in_launch": info["main_kernel_rows"],
                     "whole_evaluation": {"gpu_ms": eval_ms, "achieved": eval_achieved,
                                          "frac": eval_achieved / HBM_PEAK_GBS, "traffic": traffic_eval,
                                          "host_enqueue_ms": 1e3 * float(np.mean(host_enqueue_s[-args.steps:])),
                                          "api": "ssde_eval (synchronous); gpu_ms / host_enqueue_ms from an untimed "
                                                 "ssde_eval_device pass over the same evaluations"
                                                 if sync_api else "ssde_eval_device + D2H",
                                          "note": "all kernels of one evaluation incl. the concurrent transient-window "
                                                  "launch, the hand-over check and the reduction"}},
    }
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        try:
            line["cpu_baseline"] = cpu_baseline()
        except Exception as e:  # the baseline must never take the bench line down
            line["cpu_baseline"] = {"value": None, "unit": "track-timesteps/s", "cores": os.cpu_count(),
                                    "kind": "port", "sample": f"failed: {e}"}
    if rank == 0:
        print(json.dumps(line), flush=True)
    eng.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
