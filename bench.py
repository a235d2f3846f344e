#!/usr/bin/env python3
"""bench.py -- track-timesteps/s of one nllk + gradient evaluation (BASELINE.json metric).

A "step" is one synchronous evaluation of the hot path through the C ABI (ssde_eval, order 1: value + full
gradient, the hand-over check of the time windows, the reduction, at N > 1 the in-engine ncclAllReduce of the
2 + p doubles over xGMI, and the D2H of the result) over one resident batch of synthetic tracks, each step at a
different parameter vector (the engine memoises the last one).  Workload at every N: 10^4 two-dimensional CTCRW
tracks x 10^4 rows PER GPU (constant coefficients, sigma_obs free, mu fixed as in the vignette) -- weak scaling:
tracks shard over ranks with no data-path collective other than that all-reduce.

    python bench.py --gpus N --steps K --warmup W        (N > 1 without a launcher: starts its own N ranks)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

One process per GPU; the ranks' engines are joined by ssde_comm_init_rank (the ncclUniqueId travels over a gloo
group, which also carries the barriers and the max-over-ranks of the elapsed time: torch.distributed is plumbing
here, the collective of the data path is the engine's own).  Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import subprocess
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


def cpu_baseline(seconds_target=12.0):
    """The CPU restatement SURVEY 8(d)(i) asks for -- fp64, analytic gradient, threads over tracks on the cores this box
    grants (oracle/cpu_fast.cpp: the hand-derived step compiled for the host, -O3 with hardware FMA) -- on a bounded
    sample of the same workload; the literal dual-number oracle (the checker) is timed next to it on a smaller sample.
    A baseline, not the target."""
    from oracle_lib import cpu_fast_eval, oracle_eval
    from smoothsde_amd import capi
    from smoothsde_amd.synth import simulate
    cores = _usable_cores()
    fixed = np.array([0, 1, 1, 0, 0], dtype=np.uint8)
    par = np.array([np.log(0.1), 0.0, 0.0, np.log(2.0), 0.0])

    def busy_cores(fn):
        # the affinity mask can promise more cores than the box's CPU share grants: measure how many the threads
        # really got (process CPU time / wall time)
        w0, c0 = time.perf_counter(), time.process_time()
        fn()
        return (time.process_time() - c0) / max(time.perf_counter() - w0, 1e-9)

    # --- the analytic-gradient restatement: 64 tracks per core x 10^4 rows (the bench's track length) ----------------
    tracks, rows = 64 * cores, 10_000
    ID, times, obs = simulate("CTCRW", tracks, rows, 2, tau=2.0, nu=1.0, sigma_obs=0.1, seed=1)
    pb = capi.Problem("CTCRW", ID, times, obs, par_fixed=fixed)
    cpu_fast_eval(pb, par, threads=cores)                      # warm (library load, first touch)
    busy = busy_cores(lambda: [cpu_fast_eval(pb, par + 1e-3 * k, threads=cores) for k in range(3)])
    if busy < 0.6 * cores:
        # the CPU gets its best shot: the granted thread count or the measured one, whichever evaluates faster
        def rate_with(t):
            t_ = time.perf_counter()
            for k in range(3):
                cpu_fast_eval(pb, par + 2e-3 * k, threads=t)
            return time.perf_counter() - t_
        alt = max(1, int(round(busy)))
        cores = cores if rate_with(cores) <= rate_with(alt) else alt
    t0 = time.perf_counter()
    reps = 0
    while True:
        cpu_fast_eval(pb, par + 0.01 * reps, threads=cores)
        reps += 1
        el = time.perf_counter() - t0
        if el > seconds_target or reps >= 200:
            break
    rate = tracks * rows * reps / el
    out = {"value": rate, "unit": "track-timesteps/s", "cores": cores, "kind": "port",
           "sample": f"{tracks} CTCRW tracks x {rows} rows, regular grid, {reps} nllk+grad evaluations, oracle/libcpu_fast.so "
                     f"(analytic forward-sensitivity gradient, transition hoisted, g++ -O3 -mfma, {cores} threads over tracks)"}
    # --- the literal restatement (dual numbers over dense matrices): what the tests check against -------------------------
    try:
        t2, r2 = 8 * cores, 2000
        ID, times, obs = simulate("CTCRW", t2, r2, 2, tau=2.0, nu=1.0, sigma_obs=0.1, seed=1)
        pb2 = capi.Problem("CTCRW", ID, times, obs, par_fixed=fixed)
        oracle_eval(pb2, par, order=1, threads=cores)
        t0 = time.perf_counter()
        n2 = 0
        while time.perf_counter() - t0 < 4.0 and n2 < 50:
            oracle_eval(pb2, par + 0.01 * n2, order=1, threads=cores)
            n2 += 1
        out["literal_oracle"] = {"value": t2 * r2 * n2 / (time.perf_counter() - t0), "unit": "track-timesteps/s",
                                 "sample": f"{t2} tracks x {r2} rows, {n2} evaluations, oracle/liboracle.so (forward-mode duals "
                                           f"over 8x8 dense matrices, g++ -O2, {cores} threads)"}
    except Exception as e:  # noqa: BLE001
        out["literal_oracle"] = {"value": None, "sample": f"failed: {e}"}
    return out


def self_launch(args):
    """`python bench.py --gpus N` without a launcher: start the N ranks ourselves (one process per GPU under
    torch.distributed.run) and relay rank 0's JSON line.  This parent never imports torch and never touches a GPU."""
    import socket
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    proc = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, text=True)
    line = None
    for ln in proc.stdout.splitlines():
        if ln.startswith("{") and '"metric"' in ln:
            line = ln
        else:
            print(ln, file=sys.stderr)
    if line:
        print(line, flush=True)
    sys.exit(proc.returncode if proc.returncode else (0 if line else 1))


def theta_for(npar, d, q, k):
    base = np.zeros(npar)
    base[0] = np.log(0.1)            # log sigma_obs
    base[1 + d] = np.log(2.0)        # log tau (log sigma for BM_SSM)
    if q > d + 1:
        base[2 + d] = 0.0            # log nu / log kappa
    return base + 0.01 * np.sin(np.arange(npar) + 0.7 * k)


def secondary_workload(name, model, M, T, dev, steps, mutate):
    """The same batch shape with what real data have -- an irregular time grid, missing rows -- so that the driver's
    run times the general per-lane kernel too (BASELINE's metric configuration is the engine's best case)."""
    import torch
    from smoothsde_amd import capi
    from smoothsde_amd.synth import simulate
    d = 2
    ID, times, obs = simulate(model, M, T, d, mu=0.0, tau=2.0, nu=1.0, kappa=1.0, sigma=1.0, sigma_obs=0.1, seed=11,
                              backend="torch", device=dev)
    ID, times, obs = mutate(ID, times, obs)
    q = capi.n_sde_par(model, d)
    fixed = np.zeros(1 + q, dtype=np.uint8)
    fixed[1:1 + d] = 1
    eng = capi.Engine(capi.Problem.from_torch(model, ID, times, obs, par_fixed=fixed))
    del ID, times, obs
    npar = 1 + q
    for k in range(2):
        eng.eval(theta_for(npar, d, q, -1 - k))
    ssde_eval = eng.bound_eval(order=1)
    ths = [np.ascontiguousarray(theta_for(npar, d, q, k)) for k in range(steps)]
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    kms = []
    for k in range(steps):
        ssde_eval(ths[k])
        kms.append(eng.last_kernel_ms())
    torch.cuda.synchronize(dev)
    el = time.perf_counter() - t0
    inf = eng.info()
    rows, chk = inf["n_rows"], inf["window_check_max"]
    eng.close()
    kern = float(np.mean(kms))
    return {"workload": name, "value": rows * steps / el, "unit": "track-timesteps/s", "steps": steps,
            "ms_per_step": 1e3 * el / steps, "kernel_ms": kern,
            "required_bytes_per_row": inf["required_bytes_per_row"],
            "frac": inf["required_bytes_per_row"] * inf["main_kernel_rows"] / (kern * 1e-3) / 1e9 / HBM_PEAK_GBS,
            "window_check_max": chk, "window_retries": inf["window_retries"],
            "rows_tiled": inf["n_rows_tiled"], "groups": inf["n_groups"], "clean_groups": inf["n_clean_groups"]}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--tracks", type=int, default=10_000, help="tracks per GPU")
    ap.add_argument("--rows", type=int, default=10_000, help="rows per track")
    ap.add_argument("--model", default="CTCRW")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-secondary", action="store_true", help="skip the irregular-grid / missing-row workloads")
    args = ap.parse_args()

    # (SSDE_BENCH_SELF_LAUNCH / SSDE_BENCH_FORCE_COMM: rehearse the N > 1 plumbing -- own launcher, gloo group,
    # ncclCommInitRank, the all-reduce inside ssde_eval -- with ONE rank on a one-GPU box)
    if (args.gpus > 1 or os.environ.get("SSDE_BENCH_SELF_LAUNCH")) and "WORLD_SIZE" not in os.environ:
        self_launch(args)          # before anything touches the GPU (the children are fresh processes)

    # stdout carries ONE line, rank 0's JSON: everything any library prints there (gloo announces its connections on
    # stdout) is sent to stderr; the JSON goes to the saved descriptor at the end
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)

    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and rank == 0:
        print(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}: running {world} rank(s)", file=sys.stderr)
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    use_comm = world > 1 or bool(os.environ.get("SSDE_BENCH_FORCE_COMM"))
    if use_comm:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        dist.init_process_group("gloo", rank=rank, world_size=world)     # plumbing: id exchange, barriers, max of the clock

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
    npar = pb.n_par_full
    host_reduce = None                                    # set only if RCCL could not be brought up (see below)
    if use_comm:
        # ncclCommInitRank: the engines of all ranks, one communicator.  If RCCL cannot be brought up on this node (an
        # exception on ANY rank -- agreed on over the gloo group), the run still produces a line, with the sum over ranks
        # done by the host over gloo after every evaluation and SAID SO in config.parallelism: a slower collective, the same
        # per-rank engine.  (The engine itself has no such fallback: ssde_comm_init_rank fails loudly.)
        err = ""
        try:
            if os.environ.get("SSDE_BENCH_FAKE_RCCL_FAILURE"):        # rehearsal of the fallback on a one-GPU box
                raise RuntimeError("faked for a rehearsal")
            box = [capi.comm_unique_id() if rank == 0 else None]
        except Exception as e:                            # rank 0 could not even make an id
            box, err = [None], f"ssde_comm_unique_id: {e}"
        dist.broadcast_object_list(box, src=0)
        if box[0] is not None:
            try:
                eng.comm_init(world, rank, box[0])
            except Exception as e:
                err = f"ssde_comm_init_rank: {e}"
        else:
            err = err or "no communicator id from rank 0"
        flag = torch.tensor([1.0 if err else 0.0], dtype=torch.float64)
        dist.all_reduce(flag, op=dist.ReduceOp.MAX)
        if flag.item() > 0:
            if not err:                                   # this rank joined a communicator the others did not: start over without one
                eng.close()
                eng = capi.Engine(pb)
            host_reduce = err or "RCCL initialisation failed on another rank"
            print(f"[bench rank {rank}] RCCL not available ({host_reduce}); summing over ranks on the host (gloo)", file=sys.stderr, flush=True)

    thetas = {k: np.ascontiguousarray(theta_for(npar, d, q, k)) for k in range(-args.warmup - 1, args.steps)}   # built outside the timed region

    def reduce_on_host(val, grad):                        # the fallback collective: [value, gradient] summed over ranks by gloo
        buf = torch.from_numpy(np.concatenate([[val], grad]))
        dist.all_reduce(buf)
        return float(buf[0]), buf[1:].numpy()

    for k in range(args.warmup):
        v_, g_ = eng.eval(thetas[-1 - k], order=1)
        if host_reduce:
            reduce_on_host(v_, g_)
    if use_comm:
        dist.barrier()
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    main_ms = []
    ssde_eval = eng.bound_eval(order=1)                   # the C ABI call itself (preallocated outputs, no per-call conversions)
    # Every evaluation stamps its dominant kernel with HIP events of its own, on the stream it is launched on; the engine
    # keeps the last 64 pairs, so the durations of the timed steps are read AFTER the timed region (in batches of 64 for
    # longer runs) instead of paying an event query + a ctypes call between the steps.
    for k in range(args.steps):
        val, grad = ssde_eval(thetas[k])                  # ssde_eval: kernels, check, reduction, all-reduce, D2H
        if host_reduce:
            val, grad = reduce_on_host(val, grad)
        if (k + 1) % 64 == 0 and k + 1 < args.steps:
            main_ms.extend(eng.kernel_ms_history(64)[::-1])
    torch.cuda.synchronize(dev)
    if use_comm:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    main_ms.extend(eng.kernel_ms_history(args.steps - len(main_ms))[::-1])     # the remaining (<= 64) timed steps
    assert len(main_ms) == args.steps and all(m > 0 for m in main_ms), "a timed step has no kernel stamp"
    if use_comm:
        tmax = torch.tensor([elapsed], dtype=torch.float64)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())
    info = eng.info()
    check_max = info["window_check_max"]                  # over EVERY evaluation since create, kept by the engine
    assert np.isfinite(val) and np.all(np.isfinite(grad)), (val, grad)
    assert check_max <= capi.WINDOW_TOL, f"window hand-over check failed: {check_max}"
    assert info["n_memo_hits"] == 0, "a timed step was answered from the memo"

    # GPU span of a whole evaluation: the same evaluations again, asynchronously, between HIP events (N = 1 only)
    eval_ms = host_enq_ms = None
    if not use_comm:
        out = torch.zeros(2 + npar, dtype=torch.float64, device=dev)
        stream = torch.cuda.current_stream(dev)
        ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
        enq = []
        for k in range(args.steps):
            ev[k][0].record(stream)
            t_h = time.perf_counter()
            eng.eval_device(thetas[k], out.data_ptr(), order=1, stream=stream.cuda_stream)
            enq.append(time.perf_counter() - t_h)
            ev[k][1].record(stream)
        torch.cuda.synchronize(dev)
        eval_ms = float(np.mean([a.elapsed_time(b) for a, b in ev]))
        host_enq_ms = 1e3 * float(np.mean(enq))

    rows_per_gpu = info["n_rows"]
    total_rows = rows_per_gpu * world
    value = total_rows * args.steps / elapsed
    kern_ms = float(np.mean(main_ms))                                   # the dominant kernel alone
    # Bytes of the rows that launch scores.  `required`: what the resident layout has to read (the `times` stream is
    # not even stored on a globally regular grid: 16 of the 24 algorithmic B/row) -- the honest numerator of a
    # fraction of peak.  `algorithmic`: SURVEY 8(d)'s 24 B/row, kept for comparison; it can exceed the peak
    # precisely because 8 of those bytes are never moved.
    req_bytes = info["required_bytes_per_row"] * info["main_kernel_rows"]
    algo_bytes = info["algo_bytes_per_row"] * info["main_kernel_rows"]
    achieved = req_bytes / (kern_ms * 1e-3) / 1e9
    achieved_algo = algo_bytes / (kern_ms * 1e-3) / 1e9
    profiled = None
    pmc = os.path.join(ROOT, "profiles", "pmc_latest.json")
    if os.path.exists(pmc):
        try:
            pj = json.load(open(pmc))
            profiled = {"main_kernel_bytes": pj.get("main_kernel_bytes"), "evaluation_bytes": pj.get("hbm_bytes_per_launch"),
                        "source": "profiles/pmc_latest.json", "tag": pj.get("tag"),
                        "note": "PMC counters of an EARLIER rocprofv3 --pmc run of this command (committed profile), "
                                "not measured in this run"}
        except Exception:
            profiled = None
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
                   "rows_tiled": info["n_rows_tiled"], "groups": info["n_groups"], "clean_groups": info["n_clean_groups"],
                   "lanes_per_track": info["lanes_per_track"], "window_rows": info["window"],
                   "window_check": check_max, "window_retries": info["window_retries"],
                   "parallelism": f"tracks x{world}" + ("" if not use_comm else
                                                         (", in-engine ncclAllReduce of 2+p doubles" if not host_reduce else
                                                          f", HOST all-reduce over gloo after every evaluation (RCCL could not be brought up: {host_reduce})")),
                   "api": "ssde_eval (synchronous C ABI call)"},
        "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": achieved / HBM_PEAK_GBS, "traffic": None, "traffic_profiled": profiled,
                     "frac_of_measured_copy_6290": achieved / 6290.0,   # MI355X_MICROARCH.md: 6.29 TB/s measured copy
                     "required_bytes_per_row": info["required_bytes_per_row"],
                     "algo_bytes_per_row": info["algo_bytes_per_row"],
                     "achieved_algorithmic": achieved_algo, "frac_algorithmic": achieved_algo / HBM_PEAK_GBS,
                     "kernel": "iso_shared_kernel<stationary>" if info["uniform_dt"] else "iso_kernel",
                     "kernel_ms": kern_ms, "required_bytes_per_launch": req_bytes, "algo_bytes_per_launch": algo_bytes,
                     "rows_in_launch": info["main_kernel_rows"],
                     "whole_evaluation": None if eval_ms is None else {
                         "gpu_ms": eval_ms,
                         "achieved": info["required_bytes_per_row"] * rows_per_gpu / (eval_ms * 1e-3) / 1e9,
                         "frac": info["required_bytes_per_row"] * rows_per_gpu / (eval_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                         "host_enqueue_ms": host_enq_ms,
                         "note": "all kernels of one evaluation incl. the concurrent transient-window launch, the "
                                 "hand-over check and the reduction; from an untimed ssde_eval_device pass over the "
                                 "same evaluations between HIP events"}},
    }
    eng.close()
    del eng, pb
    if rank == 0 and world == 1 and not args.no_secondary and args.model == "CTCRW":
        # outside the timed region of `value`: the same batch shape on an irregular grid and with 5 % missing rows
        sec = []
        try:
            def irregular(ID, times, obs):
                gen = torch.Generator(device=dev)
                gen.manual_seed(5)
                return ID, torch.cumsum(0.5 + torch.rand(len(ID), device=dev, dtype=torch.float64, generator=gen), 0), obs

            def missing(ID, times, obs):
                gen = torch.Generator(device=dev)
                gen.manual_seed(7)
                na = torch.rand(len(ID), device=dev, generator=gen) < 0.05
                na[::T] = False                       # first rows stay observed (they initialise the state)
                obs[na] = float("nan")
                return ID, times, obs

            def absent(ID, times, obs):                # the same schedule, but the missing fixes are simply not in the data
                gen = torch.Generator(device=dev)
                gen.manual_seed(9)
                keep = torch.rand(len(ID), device=dev, generator=gen) >= 0.05
                keep[::T] = True
                return ID[keep].contiguous(), times[keep].contiguous(), obs[keep].contiguous()

            sec.append(secondary_workload(f"{M} CTCRW x {T}, irregular time grid (dt ~ U[0.5, 1.5] per row)", "CTCRW", M, T,
                                          dev, max(3, args.steps // 2), irregular))
            sec.append(secondary_workload(f"{M} CTCRW x {T}, regular grid, 5 % missing rows", "CTCRW", M, T, dev,
                                          max(3, args.steps // 2), missing))
            sec.append(secondary_workload(f"{M} CTCRW x {T} slots of a regular schedule, 5 % of the fixes absent from the data "
                                          f"(intervals of 1-4 steps; laid out on the lattice at create)", "CTCRW", M, T, dev,
                                          max(3, args.steps // 2), absent))
        except Exception as e:  # the secondary numbers must never take the bench line down
            sec.append({"workload": "failed", "error": str(e)})
        line["secondary"] = sec
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        try:
            line["cpu_baseline"] = cpu_baseline()
        except Exception as e:  # the baseline must never take the bench line down
            line["cpu_baseline"] = {"value": None, "unit": "track-timesteps/s", "cores": os.cpu_count(),
                                    "kind": "port", "sample": f"failed: {e}"}
    sys.stdout.flush()
    if rank == 0:
        os.write(json_fd, (json.dumps(line) + "\n").encode())
    os.close(json_fd)
    if use_comm:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
