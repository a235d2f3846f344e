#!/usr/bin/env python3
"""bench.py -- track-timesteps/s of one nllk + gradient evaluation (BASELINE.json metric).

A "step" is one evaluation of the hot path through the C ABI (order 1: value + full gradient, the hand-over check of the
time windows, the reduction, at N > 1 the in-engine ncclAllReduce of the 2 + p doubles over xGMI, and the result on the
host) over one resident batch of synthetic tracks, each step at a different parameter vector (the engine memoises the
last one).  Workloads (`--config`):

    c2p (default)  BASELINE's metric configuration, SURVEY 8(d) C2': 10^4 two-dimensional CTCRW tracks x 10^4 rows,
                   constant coefficients, sigma_obs / tau / nu free, mu fixed as in the vignette; synchronous ssde_eval
    c4             BASELINE config 4: 10^5 such tracks x 10^4 rows
    c5             BASELINE config 5: three sub-batches (BM_SSM, OU_SSM, CTCRW; 3 10^4 tracks each, ragged lengths
                   U[0.5 T, T], T = 10^4, 5 % of the rows missing -- half in column 0 only, half in every column),
                   one handle per model with its own parameter vector and its own communicator, the three
                   evaluated CONCURRENTLY (ssde_eval_device on three streams); throughput = all rows / wall time

`--scaling strong` (the default: the metric names ONE batch "on 1/2/4/8 GPUs") cuts the batch into whole-track shards,
rank r of N owning tracks [r M / N, (r + 1) M / N) -- the same batch at every N, because the simulator's numbers are a
function of (seed, global track index, row) and of nothing else (ssde_simulate, csrc/k_sim.hip); `--scaling weak` gives
every rank the whole configuration's track count (N times the work at N ranks).  At N > 1 the strong-scaling line also
carries, as `secondary[0]`, the weak-scaling reading of the same launch timed in the same run (weak_scaling_probe: every
rank a whole configuration's batch through the same engines and collective; BASELINE config 4's shape to within 25 %).

    python bench.py --gpus N --steps K --warmup W        (N > 1 without a launcher: starts its own N ranks)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

One process per GPU; the ranks' engines are joined by ssde_comm_init_rank (the ncclUniqueId travels over a gloo group,
which also carries the barriers and the max-over-ranks of the elapsed time: torch.distributed is plumbing here, the
collective of the data path is the engine's own).  Prints ONE JSON line on rank 0.

Timing: the K timed steps run with plain kernel launches (ssde_set_option(SSDE_OPT_KERNEL_STAMPS, 0): what a fitting
host runs); the dominant kernel's duration comes from a second, untimed pass over the SAME K parameter vectors with
every launch stamped by HIP events on its own stream (`roofline.kernel_ms`; `ms_per_step_stamped` is that pass's wall
time per step, for comparison)."""
import argparse
import ctypes as C
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
N_SIMDS, CLOCK_HZ = 1024, 2.4e9   # 256 CUs x 4 SIMDs; peak engine clock.  An fp64 VALU instruction occupies a SIMD's issue port
                                  # for 4 cycles (64 lanes at 16 lanes / cycle): the fp64-issue roof of a kernel that runs
                                  # V VALU wave-instructions per launch is V x 4 / (1024 x 2.4 GHz) seconds


def _valu_table():
    """VALU wave-instructions per row of every kernel family, from the committed SQ_INSTS_VALU passes (profiles/valu_per_row.json,
    written by tools/summarise_profiles.py from rocprofv3 --pmc runs of the same workloads) -- NOT measured in this run; the tag
    says which profile session."""
    try:
        return json.load(open(os.path.join(ROOT, "profiles", "valu_per_row.json")))
    except Exception:  # noqa: BLE001
        return {"tag": None, "kernels": {}}


def two_roofs(kernel_name, rows, bytes_required, kernel_ms, table=None):
    """Both roofs of one launch: HBM (bytes the resident layout must read / 8 TB/s) and fp64 issue (VALU wave-instructions x 4
    cycles / (1024 SIMDs x 2.4 GHz)); `bound` = the one that leaves less headroom."""
    table = table or _valu_table()
    fam = kernel_name.split("<")[0].split(" ")[0]
    ent = table.get("kernels", {}).get(kernel_name) or table.get("kernels", {}).get(fam)
    ks = kernel_ms * 1e-3
    frac_hbm = bytes_required / ks / 1e9 / HBM_PEAK_GBS if ks > 0 else None
    out = {"frac_hbm": frac_hbm, "frac_fp64_issue": None, "bound": "hbm", "valu_per_row": None, "valu_source": None}
    if ent and ks > 0:
        v = float(ent["valu_per_row"])
        out["valu_per_row"] = v
        out["valu_source"] = f"{ent.get('source')} (tag {table.get('tag')}; SQ_INSTS_VALU of a committed rocprofv3 --pmc pass, not measured in this run)"
        out["frac_fp64_issue"] = rows * v * 4.0 / (N_SIMDS * CLOCK_HZ) / ks
        if frac_hbm is None or out["frac_fp64_issue"] > frac_hbm:
            out["bound"] = "fp64_issue"
    return out

CONFIGS = {
    "c2p": {"tracks": 10_000, "rows": 10_000, "label": "SURVEY 8(d) C2' (BASELINE metric)"},
    "c4": {"tracks": 100_000, "rows": 10_000, "label": "SURVEY 8(d) C4 (BASELINE config 4)"},
    "c5": {"tracks": 30_000, "rows": 10_000, "label": "SURVEY 8(d) C5 (BASELINE config 5)"},
}
C5_MODELS = ("BM_SSM", "OU_SSM", "CTCRW")


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


def shard_range(M, world, rank, scaling):
    """Tracks [m0, m1) of the batch this rank owns.  strong: the M tracks of the configuration cut into whole-track
    shards; weak: every rank M tracks of an N M-track batch."""
    if scaling == "weak":
        return rank * M, (rank + 1) * M
    return rank * M // world, (rank + 1) * M // world


def track_lengths(M_total, T, seed):
    """C5: ragged lengths U[0.5 T, T] of EVERY track of the batch (a function of the seed only: any rank can slice it)."""
    rng = np.random.default_rng(seed)
    return rng.integers(T // 2, T + 1, size=M_total, dtype=np.int64)


def missing_rows(obs, row_offset, first_rows, frac=0.05):
    """C5: `frac` of the rows missing, half of them in column 0 only and half in every column (the reference tests
    column 0: nllk_ctcrw.hpp:214); a track's first row stays observed (it initialises the state).  Which rows is a
    hash of the GLOBAL row index, so the batch does not depend on how it is cut over ranks."""
    import torch
    n = obs.shape[0]
    P = 2147483647
    g = (torch.arange(n, device=obs.device, dtype=torch.int64) + int(row_offset)) % P
    x = (g * 48271 + 11) % P
    x = (x * 69621 + (g // 7) % P) % P
    x = (x * 16807 + 12345) % P
    u = x.to(torch.float64) / float(P)
    col0 = u < 0.5 * frac
    allc = (u >= 0.5 * frac) & (u < frac)
    col0[first_rows] = False
    allc[first_rows] = False
    obs[col0, 0] = float("nan")
    obs[allc, :] = float("nan")
    return int(col0.sum() + allc.sum())


class Handle:
    """One engine of the workload on this rank, with its parameter vectors and its result buffer."""

    def __init__(self, name, model, eng, pb, d, rows_local, rows_total, k_lo, k_hi):
        from smoothsde_amd import capi
        self.name, self.model, self.eng, self.pb, self.d = name, model, eng, pb, d
        self.q = capi.n_sde_par(model, d)
        self.npar = pb.n_par_full
        self.rows_local, self.rows_total = rows_local, rows_total
        self.thetas = {k: np.ascontiguousarray(theta_for(self.npar, d, self.q, k)) for k in range(k_lo, k_hi)}
        self.ptrs = {k: v.ctypes.data_as(C.POINTER(C.c_double)) for k, v in self.thetas.items()}
        self.kernel_ms = None
        self.out = None
        self.stream = None


def build_handles(args, dev, rank, world):
    """The configuration's engines on this rank (data simulated in HBM by the engine's own simulator kernel)."""
    import torch
    from smoothsde_amd import capi
    cfg = args.config
    M, T, d = args.tracks, args.rows, 2
    k_lo, k_hi = -args.warmup - 1, args.steps
    handles = []
    if cfg in ("c2p", "c4"):
        model = args.model
        m0, m1 = shard_range(M, world, rank, args.scaling)
        M_total = M * world if args.scaling == "weak" else M
        # SURVEY.md 8(d) C2' / C4: tau=2, nu=1, mu=0, sigma_obs=0.1, dt=1, time increasing over the whole batch
        ID, times, obs = capi.simulate_device(model, m1 - m0, T, d, mu=0.0, tau=2.0, nu=1.0, kappa=1.0, sigma=1.0,
                                              sigma_obs=0.1, seed=args.seed, track0=m0, device=dev)
        q = capi.n_sde_par(model, d)
        fixed = np.zeros(1 + q, dtype=np.uint8)
        fixed[1:1 + d] = 1  # fixpar = c("mu1","mu2") as in the vignette (smoothSDE.rmd:486-490)
        pb = capi.Problem.from_torch(model, ID, times, obs, par_fixed=fixed)
        eng = capi.Engine(pb)
        del ID, times, obs
        handles.append(Handle(model, model, eng, pb, d, (m1 - m0) * T, M_total * T, k_lo, k_hi))
    else:
        for i, model in enumerate(C5_MODELS):
            M_total = M * world if args.scaling == "weak" else M
            m0, m1 = shard_range(M, world, rank, args.scaling)
            lens = track_lengths(M_total, T, args.seed + 100 * (i + 1))
            row0_all = np.concatenate([[0], np.cumsum(lens)])
            # BM_SSM (mu = 0.1, sigma = 1), OU_SSM (mu = (5, -5), tau = 2, kappa = 1: smoothSDE.rmd:346-350), CTCRW as C2
            kw = {"BM_SSM": dict(mu=0.1, sigma=1.0), "OU_SSM": dict(mu=[5.0, -5.0], tau=2.0, kappa=1.0, z0=[5.0, -5.0]),
                  "CTCRW": dict(mu=0.0, tau=2.0, nu=1.0)}[model]
            ID, times, obs = capi.simulate_device(model, m1 - m0, T, d, sigma_obs=0.1, seed=args.seed + 100 * (i + 1), track0=m0,
                                                  lengths=lens[m0:m1], row_offset=int(row0_all[m0]), device=dev, **kw)
            first = torch.as_tensor(row0_all[m0:m1] - row0_all[m0], device=dev)
            missing_rows(obs, int(row0_all[m0]), first)
            pb = capi.Problem.from_torch(model, ID, times, obs)              # every parameter free (the full Kalman update path)
            eng = capi.Engine(pb)
            del ID, times, obs
            handles.append(Handle(model, model, eng, pb, d, int(row0_all[m1] - row0_all[m0]), int(row0_all[-1]), k_lo, k_hi))
    return handles


def join_ranks(handles, rank, world, dist):
    """ncclCommInitRank: the engines of all ranks into one communicator per handle that needs one (c5: ONE communicator, joined by
    the first handle -- the three handles' result vectors are summed by a single collective per step, main()).  If RCCL cannot
    be brought up on this node (an exception on ANY rank -- agreed on over the gloo group BEFORE anyone enters the next
    collective call, so that no rank is left alone inside ncclCommInitRank), the run still produces a line, with the sum over
    ranks done by the host over gloo after every evaluation and SAID SO in config.parallelism: a slower collective, the same
    per-rank engine.  (The engine itself has no such fallback: ssde_comm_init_rank fails loudly.)"""
    import torch
    from smoothsde_amd import capi

    def agreed(local_err):
        flag = torch.tensor([1.0 if local_err else 0.0], dtype=torch.float64)
        dist.all_reduce(flag, op=dist.ReduceOp.MAX)
        return flag.item() > 0

    err = ""
    for h in handles[:1]:
        try:
            if os.environ.get("SSDE_BENCH_FAKE_RCCL_FAILURE"):        # rehearsal of the fallback on a one-GPU box
                raise RuntimeError("faked for a rehearsal")
            box = [capi.comm_unique_id() if rank == 0 else None]
        except Exception as e:                            # rank 0 could not even make an id
            box, err = [None], f"ssde_comm_unique_id: {e}"
        dist.broadcast_object_list(box, src=0)
        if box[0] is None:
            err = err or "no communicator id from rank 0"
        if agreed(err):                                   # nobody calls ncclCommInitRank unless everybody will
            err = err or "RCCL initialisation failed on another rank"
            break
        try:
            h.eng.comm_init(world, rank, box[0])
        except Exception as e:
            err = f"ssde_comm_init_rank: {e}"
        if agreed(err):
            err = err or "RCCL initialisation failed on another rank"
            break
    if err:
        for h in handles:                                 # some engine may have joined a communicator others did not: start over without
            h.eng.close()
            h.eng = capi.Engine(h.pb)
        return err
    return None


def weak_scaling_probe(args, dev, rank, world, dist, steps):
    """N > 1, strong scaling (the default): the SAME launch once more as a weak-scaling batch -- every rank a whole configuration's
    worth of tracks (N x 10^4 x 10^4 rows in all: BASELINE config 4's shape, 10^5 tracks on 8 GPUs, to within 25 %) -- so that the
    driver's run shows both readings of "1/2/4/8 GPUs".  Every phase ends with an agreement over the gloo group: a rank that failed
    makes ALL ranks leave together (no rank is left alone in a collective), and the main line is printed either way."""
    import copy
    import torch
    from smoothsde_amd import capi

    def agreed(local_err):
        flag = torch.tensor([1.0 if local_err else 0.0], dtype=torch.float64)
        dist.all_reduce(flag, op=dist.ReduceOp.MAX)
        return flag.item() > 0

    a2 = copy.copy(args)
    a2.scaling, a2.steps, a2.warmup = "weak", steps, min(args.warmup, 3)
    hs, err = [], ""
    try:
        hs = build_handles(a2, dev, rank, world)
    except Exception as e:  # noqa: BLE001
        err = f"build: {e}"
    if agreed(err):
        for h in hs:
            h.eng.close()
        return {"workload": "weak-scaling probe", "error": err or "another rank could not build its batch"}
    host_reduce = join_ranks(hs, rank, world, dist)          # (agrees by itself; falls back to the host sum like the main run)
    h0 = hs[0]
    val_c, grad = C.c_double(), np.zeros(h0.npar)
    gp, vp = grad.ctypes.data_as(C.POINTER(C.c_double)), C.byref(val_c)
    err = ""
    try:
        for k in range(a2.warmup):
            h0.eng._check(h0.eng.lib.ssde_eval(h0.eng._h, h0.ptrs[-1 - k], h0.npar, 1, vp, gp))
    except Exception as e:  # noqa: BLE001
        err = f"warm-up: {e}"
    if agreed(err):
        h0.eng.close()
        return {"workload": "weak-scaling probe", "error": err or "another rank failed in its warm-up"}
    dist.barrier()
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    try:
        for k in range(steps):
            h0.eng._check(h0.eng.lib.ssde_eval(h0.eng._h, h0.ptrs[k], h0.npar, 1, vp, gp))
            if host_reduce:
                buf = torch.from_numpy(np.concatenate([[val_c.value], grad]))
                dist.all_reduce(buf)
    except Exception as e:  # noqa: BLE001  (an engine error is the same on every rank -- same parameters -- so nobody waits alone)
        err = f"step: {e}"
    torch.cuda.synchronize(dev)
    dist.barrier()
    el = torch.tensor([time.perf_counter() - t0], dtype=torch.float64)
    dist.all_reduce(el, op=dist.ReduceOp.MAX)
    inf = h0.eng.info()
    h0.eng.close()
    if agreed(err):
        return {"workload": "weak-scaling probe", "error": err or "another rank failed in a timed step"}
    rows = h0.rows_total
    return {"workload": f"weak scaling: {world} ranks x {args.tracks} {args.model} tracks x {args.rows} rows (every rank a whole configuration's batch), "
                        f"{'in-engine ncclAllReduce' if not host_reduce else 'HOST all-reduce over gloo'} of 2+p doubles per step",
            "scaling": "weak", "n_gpus": world, "value": rows * steps / float(el.item()), "unit": "track-timesteps/s", "steps": steps,
            "ms_per_step": 1e3 * float(el.item()) / steps, "rows_total": rows, "kernel": capi.KERNEL_NAMES.get(inf["kernel_id"], "?"),
            "comm_ranks_reported": inf["comm_ranks_reported"], "nllk_at_last_step": val_c.value}


def _roofs_of(inf, kern_ms):
    """kernel family + both roofs of a secondary workload's dominant launch (`frac` stays the HBM fraction, as before)"""
    from smoothsde_amd import capi
    kname = capi.KERNEL_NAMES.get(inf["kernel_id"], "?")
    if not kern_ms or kern_ms <= 0:
        return {"kernel": kname, "frac": None, "frac_hbm": None, "frac_fp64_issue": None, "bound": None, "rows_in_launch": inf["main_kernel_rows"]}
    tr = two_roofs(kname, inf["main_kernel_rows"], inf["required_bytes_per_row"] * inf["main_kernel_rows"], kern_ms)
    return {"kernel": kname, "frac": tr["frac_hbm"], "frac_hbm": tr["frac_hbm"], "frac_fp64_issue": tr["frac_fp64_issue"],
            "bound": tr["bound"], "valu_per_row": tr["valu_per_row"], "rows_in_launch": inf["main_kernel_rows"]}


def secondary_workload(name, model, M, T, dev, steps, mutate):
    """The same batch shape with what real data have -- an irregular time grid, missing rows -- so that the driver's
    run times the general per-lane kernel too (BASELINE's metric configuration is the engine's best case)."""
    import torch
    from smoothsde_amd import capi
    d = 2
    ID, times, obs = capi.simulate_device(model, M, T, d, mu=0.0, tau=2.0, nu=1.0, kappa=1.0, sigma=1.0, sigma_obs=0.1, seed=11, device=dev)
    ID, times, obs = mutate(ID, times, obs.contiguous())
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
            **_roofs_of(inf, kern),
            "window_check_max": chk, "window_retries": inf["window_retries"],
            "rows_tiled": inf["n_rows_tiled"], "groups": inf["n_groups"], "clean_groups": inf["n_clean_groups"],
            "quiet_window": inf["quiet_window"], "quiet_share": inf["quiet_share"]}


def _timed_engine(name, eng, theta, steps, dev, extra=None):
    """`steps` evaluations of an engine at distinct parameter vectors (theta(k)); the entry every secondary workload reports"""
    import torch
    from smoothsde_amd import capi
    for k in range(2):
        eng.eval(theta(-1 - k))
    ssde_eval = eng.bound_eval(order=1)
    ths = [np.ascontiguousarray(theta(k)) for k in range(steps)]
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    kms = []
    for k in range(steps):
        ssde_eval(ths[k])
        kms.append(eng.last_kernel_ms())
    torch.cuda.synchronize(dev)
    el = time.perf_counter() - t0
    inf = eng.info()
    eng.close()
    kern = float(np.mean(kms))
    out = {"workload": name, "value": inf["n_rows"] * steps / el, "unit": "track-timesteps/s", "steps": steps, "ms_per_step": 1e3 * el / steps,
           "kernel_ms": kern, "path": capi.PATH_NAMES[inf["path"]], "required_bytes_per_row": inf["required_bytes_per_row"],
           "algo_bytes_per_row": inf["algo_bytes_per_row"], **_roofs_of(inf, kern),
           "window_check_max": inf["window_check_max"], "window_retries": inf["window_retries"], "groups": inf["n_groups"]}
    out.update(extra or {})
    return out


def baseline_config_workloads(dev, steps, tracks=10_000, rows=10_000):
    """BASELINE.json's configurations 1, 2 and 3 at their named sizes (the metric's line is configuration 2 with 10^4 rows per track):
    C1 one elephant-like CTCRW track with tau and nu splines of a covariate (smoothSDE.rmd:476-490; the reference's own CPU-runnable
    case -- a latency measurement here: 3672 rows); C2 10^4 CTCRW tracks x 10^3 rows, constant coefficients; C3 10^4 OU tracks x 10^4
    rows with a 9-column spline-varying drift, the design block streamed (88 B/row) and -- the same model -- evaluated by the lanes
    from its B-spline table (ssde_ppbasis: 24 B/row resident)."""
    import torch
    from smoothsde_amd import capi
    from smoothsde_amd.synth import bspline_basis, bspline_ppbasis, second_difference_penalty, simulate
    out = []
    # C1
    ID1, t1, o1 = simulate("CTCRW", 1, 3672, 2, tau=1.0, nu=1.0, sigma_obs=0.05, z0=[572.34, 1675.42], seed=342)
    temp = 30 + 10 * np.sin(np.arange(3672) * 2 * np.pi / 24) + np.random.default_rng(342).normal(0, 2, 3672)
    B = bspline_basis((temp - temp.min()) / (temp.max() - temp.min()), 9)
    S9 = second_difference_penalty(9)
    eng = capi.Engine(capi.Problem("CTCRW", ID1, t1, o1, X_re=[None, None, B, B], S_list=[S9, S9],
                                   par_fixed=np.r_[0, 1, 1, 0, 0, 1, 1, np.zeros(18)].astype(np.uint8)))
    p1 = np.r_[np.log(0.05), 0, 0, 0, 0, 0, 0, 0.05 * np.sin(np.arange(18))]
    out.append(_timed_engine("BASELINE config 1: one elephant-like CTCRW track x 3672 rows, tau and nu splines of a covariate (18 columns)",
                             eng, lambda k: p1 + 1e-3 * np.sin(k + np.arange(len(p1))), max(steps, 10), dev, {"baseline_config": 1}))
    # C2
    T2 = max(16, rows // 10)
    ID, times, obs = capi.simulate_device("CTCRW", tracks, T2, 2, mu=0.0, tau=2.0, nu=1.0, kappa=1.0, sigma=1.0, sigma_obs=0.1, seed=1, device=dev)
    eng = capi.Engine(capi.Problem.from_torch("CTCRW", ID, times, obs, par_fixed=[0, 1, 1, 0, 0]))
    del ID, times, obs
    out.append(_timed_engine(f"BASELINE config 2: {tracks} CTCRW x {T2}, constant coefficients, regular grid", eng,
                             lambda k: theta_for(5, 2, 4, k), max(steps, 10), dev, {"baseline_config": 2}))
    # C3, streamed and table form
    ID, times, obs = simulate("OU", tracks, rows, 1, mu=1.0, tau=2.0, kappa=1.0, seed=2, backend="torch", device=dev)
    n = len(ID)
    x = torch.cumsum(torch.randn(n, device=dev, dtype=torch.float64) * 0.01, 0)
    x = (x - x.min()) / (x.max() - x.min())
    p3 = np.concatenate([[1.0, np.log(2.0), 0.0], [0.0], 0.05 * np.sin(np.arange(9))])
    th3 = lambda k: p3 + 1e-3 * np.sin(k + np.arange(len(p3)))
    basis = bspline_ppbasis(x, 9, centre=np.zeros(9))
    Bd = torch.stack([torch.cos((k + 1) * np.pi * x) for k in range(9)], dim=1)         # (a 9-column block built in HBM; tools/bench_configs.py)
    eng = capi.Engine(capi.Problem.from_torch("OU", ID, times, obs, X_re=[Bd, None, None], S_list=[second_difference_penalty(9)]))
    del Bd
    out.append(_timed_engine(f"BASELINE config 3: {tracks} OU x {rows}, spline-varying drift, 9 design columns streamed (88 B/row)", eng, th3,
                             steps, dev, {"baseline_config": 3}))
    eng = capi.Engine(capi.Problem.from_torch("OU", ID, times, obs, basis_re=[basis, None, None], S_list=[second_difference_penalty(9)]))
    del ID, times, obs, x
    out.append(_timed_engine("BASELINE config 3 with the design block evaluated by the lanes from its B-spline table (ssde_ppbasis)", eng, th3,
                             steps, dev, {"baseline_config": 3, "form": "table"}))
    return out


def row_varying_batch(M, T, dev, k_cols=9):
    """the batch of row_varying_workload, in HBM: (ID, times, obs, B, S, par_fixed) -- tests/test_gpu_whole_batch.py compares the very
    launch that is timed with the oracle"""
    import torch
    from smoothsde_amd import capi
    from smoothsde_amd.synth import second_difference_penalty
    ID, times, obs = capi.simulate_device("CTCRW", M, T, 2, tau=1.0, nu=1.0, sigma_obs=0.05, seed=342, device=dev)
    n = M * T
    gen = torch.Generator(device=dev)
    gen.manual_seed(3)
    i = torch.arange(n, device=dev, dtype=torch.float64)
    u = (0.5 + 0.4 * torch.sin(i * (2 * np.pi / 24)) + 0.03 * torch.randn(n, device=dev, dtype=torch.float64, generator=gen)).clamp_(0.0, 1.0)
    knots = torch.arange(k_cols, device=dev, dtype=torch.float64)
    B = (1.0 - (u[:, None] * (k_cols - 1) - knots[None, :]).abs()).clamp_(min=0.0)       # (n, k): a partition of unity
    del i, u
    S = second_difference_penalty(k_cols)
    fixed = np.r_[0, 1, 1, 0, 0, 1, 1, np.zeros(2 * k_cols)].astype(np.uint8)            # mu and the smoothing parameters held
    return ID, times, obs.contiguous(), B, S, fixed


def row_varying_theta(k, k_cols=9):
    npar = 7 + 2 * k_cols
    return np.ascontiguousarray(np.r_[np.log(0.05), 0, 0, 0, 0, 0, 0, 0.05 * np.sin(np.arange(2 * k_cols))] + 1e-3 * np.sin(k + np.arange(npar)))


def row_varying_workload(M, T, dev, steps, k_cols=9):
    """1e4 CTCRW tracks with tau AND nu smooth in a covariate (2 x 9 design columns streamed next to the observations): the
    batch-scale form of BASELINE's config 1 (nllk_ctcrw.hpp:143-156), on the lane = track kernel with the gradient by a reverse
    sweep (k_iso_adj.hip; SSDE_CV_ADJ=0: one filter tangent per design column, k_iso_colvar.hip).  Hat-function basis of a per-row
    covariate, built on the device."""
    import torch
    from smoothsde_amd import capi
    ID, times, obs, B, S, fixed = row_varying_batch(M, T, dev, k_cols)
    eng = capi.Engine(capi.Problem.from_torch("CTCRW", ID, times, obs, X_re=[None, None, B, B], S_list=[S, S], par_fixed=fixed))
    del ID, times, obs, B
    def theta(k):
        return row_varying_theta(k, k_cols)
    for k in range(2):
        eng.eval(theta(-1 - k))
    ssde_eval = eng.bound_eval(order=1)
    ths = [theta(k) for k in range(steps)]
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    kms = []
    for k in range(steps):
        ssde_eval(ths[k])
        kms.append(eng.last_kernel_ms())
    torch.cuda.synchronize(dev)
    el = time.perf_counter() - t0
    inf = eng.info()
    eng.close()
    kern = float(np.mean(kms))
    return {"workload": f"{M} CTCRW x {T}, tau and nu smooth in a covariate ({2 * k_cols} design columns streamed), regular grid",
            "value": inf["n_rows"] * steps / el, "unit": "track-timesteps/s", "steps": steps, "ms_per_step": 1e3 * el / steps,
            "kernel_ms": kern, "path": capi.PATH_NAMES[inf["path"]], "required_bytes_per_row": inf["required_bytes_per_row"],
            **_roofs_of(inf, kern),
            "window_check_max": inf["window_check_max"], "window_retries": inf["window_retries"], "groups": inf["n_groups"]}


def argos_workload(M, T, dev, steps):
    """The Argos model at the metric's size: CTCRW tracks with a 2 x 2 error ellipse on every fix (H_array, nllk_ctcrw.hpp:203-205)
    and one tau, one nu -- 4 x 4 covariance lanes, one wave per (64-track group, time window) (iso_full_kernel)."""
    import torch
    from smoothsde_amd import capi
    ID, times, obs = capi.simulate_device("CTCRW", M, T, 2, tau=2.0, nu=1.0, sigma_obs=0.1, seed=13, device=dev)
    n = M * T
    gen = torch.Generator(device=dev)
    gen.manual_seed(17)
    A = 0.05 * torch.randn(n, 2, 2, device=dev, dtype=torch.float64, generator=gen)
    Hn = A @ A.transpose(1, 2)
    Hn[:, 0, 0] += 0.0025
    Hn[:, 1, 1] += 0.0025
    H = Hn.permute(1, 2, 0)                                   # (d, d, n)
    del A
    fixed = np.array([1, 1, 1, 0, 0], dtype=np.uint8)         # (log_sigma_obs is not in the model; the drift held at zero)
    eng = capi.Engine(capi.Problem.from_torch("CTCRW", ID, times, obs.contiguous(), par_fixed=fixed, H=H))
    del ID, times, obs, H, Hn

    def theta(k):
        return np.ascontiguousarray(np.array([0.0, 0.0, 0.0, np.log(2.0), 0.0]) + 1e-3 * np.sin(k + np.arange(5)))
    for k in range(2):
        eng.eval(theta(-1 - k))
    ssde_eval = eng.bound_eval(order=1)
    ths = [theta(k) for k in range(steps)]
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    kms = []
    for k in range(steps):
        ssde_eval(ths[k])
        kms.append(eng.last_kernel_ms())
    torch.cuda.synchronize(dev)
    el = time.perf_counter() - t0
    inf = eng.info()
    eng.close()
    kern = float(np.mean(kms))
    return {"workload": f"{M} CTCRW x {T} with a per-row 2 x 2 measurement covariance (H_array) and constant tau, nu", "value": inf["n_rows"] * steps / el,
            "unit": "track-timesteps/s", "steps": steps, "ms_per_step": 1e3 * el / steps, "kernel_ms": kern, "path": capi.PATH_NAMES[inf["path"]],
            "required_bytes_per_row": inf["required_bytes_per_row"],
            **_roofs_of(inf, kern),
            "window_check_max": inf["window_check_max"], "window_retries": inf["window_retries"], "groups": inf["n_groups"]}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--config", choices=sorted(CONFIGS), default="c2p")
    ap.add_argument("--scaling", choices=("strong", "weak"), default="strong",
                    help="strong: the configuration's batch split over the ranks; weak: that many tracks PER rank")
    ap.add_argument("--tracks", type=int, default=None, help="tracks of the batch (c5: per sub-batch); default: the configuration's")
    ap.add_argument("--rows", type=int, default=None, help="rows per track (c5: of the longest track)")
    ap.add_argument("--model", default="CTCRW", help="c2p / c4 only")
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-secondary", action="store_true", help="skip the irregular-grid / missing-row workloads")
    args = ap.parse_args()
    args.tracks = args.tracks or CONFIGS[args.config]["tracks"]
    args.rows = args.rows or CONFIGS[args.config]["rows"]

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

    handles = build_handles(args, dev, rank, world)
    host_reduce = join_ranks(handles, rank, world, dist) if use_comm else None
    if host_reduce:
        print(f"[bench rank {rank}] RCCL not available ({host_reduce}); summing over ranks on the host (gloo)", file=sys.stderr, flush=True)
    concurrent = len(handles) > 1
    npmax = max(h.npar for h in handles)
    outs = torch.zeros(len(handles), 2 + npmax, dtype=torch.float64, device=dev)
    for i, h in enumerate(handles):
        h.out = outs[i]
        h.stream = torch.cuda.Stream(dev) if concurrent else torch.cuda.current_stream(dev)
        h.eng.set_option(capi.OPT_KERNEL_STAMPS, 0)       # the timed steps: plain launches

    def reduce_on_host(val, grad):                        # the fallback collective: [value, gradient] summed over ranks by gloo
        buf = torch.from_numpy(np.concatenate([[val], grad]))
        dist.all_reduce(buf)
        return float(buf[0]), buf[1:].numpy()

    last = {}
    one_collective = concurrent and use_comm and not host_reduce
    if one_collective:
        # the three handles' result vectors are ONE contiguous buffer: the first handle's communicator sums it in a single
        # ncclAllReduce per step (its own in-engine collective is deferred), instead of three collectives on three streams
        handles[0].eng.set_option(capi.OPT_COMM_DEFER, 1)

    if not concurrent:
        h0 = handles[0]
        f, hp, n = h0.eng.lib.ssde_eval, h0.eng._h, h0.npar
        val_c = C.c_double()
        grad = np.zeros(n)
        gp, vp = grad.ctypes.data_as(C.POINTER(C.c_double)), C.byref(val_c)

        def step(k):                                      # ssde_eval: kernels, check, reduction, all-reduce, result on the host
            st = f(hp, h0.ptrs[k], n, 1, vp, gp)
            if st != 0:
                h0.eng._check(st)
            if host_reduce:
                last["val"], last["grad"] = reduce_on_host(val_c.value, grad)
    else:
        def step(k):                                      # the three handles side by side, each on its own stream
            for h in handles:
                h.eng.eval_device(h.thetas[k], h.out.data_ptr(), order=1, stream=h.stream.cuda_stream)
            if one_collective:
                s0 = handles[0].stream
                for h in handles[1:]:
                    s0.wait_stream(h.stream)
                handles[0].eng.comm_allreduce(outs.data_ptr(), outs.numel(), stream=s0.cuda_stream)
            for h in handles:
                h.stream.synchronize()
            host = outs.cpu().numpy()
            for i, h in enumerate(handles):
                # (rare) a hand-over check failed: widen and repeat this handle -- the check value is the ranks' sum, so every
                # rank takes this branch together
                if not host[i, 1 + h.npar] <= capi.WINDOW_TOL:
                    for _ in range(4):
                        h.eng.widen_windows(4)
                        h.eng.eval_device(h.thetas[k], h.out.data_ptr(), order=1, stream=h.stream.cuda_stream)
                        if one_collective:
                            handles[0].eng.comm_allreduce(h.out.data_ptr(), h.out.numel(), stream=h.stream.cuda_stream)
                        h.stream.synchronize()
                        host[i] = h.out.cpu().numpy()
                        if host[i, 1 + h.npar] <= capi.WINDOW_TOL:
                            break
            if host_reduce:
                t = torch.from_numpy(host[:, :1 + npmax].copy())
                dist.all_reduce(t)
                host[:, :1 + npmax] = t.numpy()
            last["val"] = float(sum(host[i, 0] for i in range(len(handles))))
            last["grad"] = np.concatenate([host[i, 1:1 + h.npar] for i, h in enumerate(handles)])
            last["check"] = float(max(host[i, 1 + h.npar] for i, h in enumerate(handles)))

    for k in range(args.warmup):
        step(-1 - k)
    import gc
    gc.collect()
    gc.disable()                                      # (a generation-2 collection of the interpreter inside the timed region costs milliseconds: one 4.4 ms step seen in 100)
    if use_comm:
        dist.barrier()
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    t_step = [t0]
    for k in range(args.steps):
        step(k)
        t_step.append(time.perf_counter())            # (the synchronous step has returned its result: one clock read per step)
    torch.cuda.synchronize(dev)
    if use_comm:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    gc.enable()
    step_ms = 1e3 * np.diff(np.array(t_step))
    if use_comm:
        tmax = torch.tensor([elapsed], dtype=torch.float64)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())
    if not concurrent and not host_reduce:
        last["val"], last["grad"] = val_c.value, grad                        # (the C ABI wrote them in place)
    val, grad_last = last["val"], np.array(last["grad"])
    infos = [h.eng.info() for h in handles]
    check_max = max([i["window_check_max"] for i in infos] + [last.get("check", 0.0)])   # over EVERY evaluation since create
    assert np.isfinite(val) and np.all(np.isfinite(grad_last)), (val, grad_last)
    assert check_max <= capi.WINDOW_TOL, f"window hand-over check failed: {check_max}"
    assert all(i["n_memo_hits"] == 0 for i in infos), "a timed step was answered from the memo"

    # ---- second pass, untimed for `value`: the same K evaluations with every dominant kernel stamped by HIP events on its own
    # stream; the engine keeps the last 64 pairs, read in batches after the calls
    for h in handles:
        h.eng.set_option(capi.OPT_KERNEL_STAMPS, 1)
        h.eng.forget()                                    # (K = 1: the memo holds that very vector)
        h.ms = []
    torch.cuda.synchronize(dev)
    if use_comm:
        dist.barrier()
    stamped_s = 0.0
    phases = []                                           # per step: where this rank's synchronous evaluation spent its time
    for k in range(args.steps):
        t1 = time.perf_counter()
        step(k)                                           # the same parameter vectors
        stamped_s += time.perf_counter() - t1
        if not concurrent:
            try:
                phases.append(handles[0].eng.last_phase_ms())
            except Exception:  # noqa: BLE001  (a measurement hook must never take the line down)
                pass
        if (k + 1) % 64 == 0 or k + 1 == args.steps:
            cnt = (k % 64) + 1
            for h in handles:
                h.ms.extend(h.eng.kernel_ms_history(cnt)[::-1])
    torch.cuda.synchronize(dev)
    stamped_ms = 1e3 * stamped_s / args.steps
    if use_comm:
        dist.barrier()
    for h in handles:
        assert len(h.ms) == args.steps and all(m > 0 for m in h.ms), "a stamped step has no kernel stamp"
        h.kernel_ms = float(np.mean(h.ms))
    infos = [h.eng.info() for h in handles]

    # GPU span of a whole evaluation: the same evaluations again, asynchronously, between HIP events (N = 1 only)
    eval_ms = host_enq_ms = None
    if not use_comm and not concurrent:
        h0 = handles[0]
        stream = torch.cuda.current_stream(dev)
        ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
        enq = []
        for k in range(args.steps):
            ev[k][0].record(stream)
            t_h = time.perf_counter()
            h0.eng.eval_device(h0.thetas[k], h0.out.data_ptr(), order=1, stream=stream.cuda_stream)
            enq.append(time.perf_counter() - t_h)
            ev[k][1].record(stream)
        torch.cuda.synchronize(dev)
        eval_ms = float(np.mean([a.elapsed_time(b) for a, b in ev]))
        host_enq_ms = 1e3 * float(np.mean(enq))

    rows_rank = sum(h.rows_local for h in handles)
    rows_t = torch.tensor([float(rows_rank)], dtype=torch.float64)
    if use_comm:
        dist.all_reduce(rows_t)
    total_rows = int(rows_t.item())
    value = total_rows * args.steps / elapsed

    def roof(h, info):
        # Bytes of the rows that launch scores.  `required`: what the resident layout has to read (the `times` stream is
        # not even stored on a globally regular grid: 16 of the 24 algorithmic B/row) -- the honest numerator of a
        # fraction of peak.  `algorithmic`: SURVEY 8(d)'s 24 B/row, kept for comparison; it can exceed the peak
        # precisely because 8 of those bytes are never moved.
        req = info["required_bytes_per_row"] * info["main_kernel_rows"]
        algo = info["algo_bytes_per_row"] * info["main_kernel_rows"]
        ach = req / (h.kernel_ms * 1e-3) / 1e9
        ach_a = algo / (h.kernel_ms * 1e-3) / 1e9
        kname = capi.KERNEL_NAMES.get(info["kernel_id"], "?")
        tr = two_roofs(kname, info["main_kernel_rows"], req, h.kernel_ms, valu_table)
        return {"bound": tr["bound"], "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS, "traffic": None,
                "frac_hbm": tr["frac_hbm"], "frac_fp64_issue": tr["frac_fp64_issue"], "valu_per_row": tr["valu_per_row"],
                "valu_source": tr["valu_source"],
                "roofs_note": "achieved / peak / frac are the HBM roof (bytes the resident layout must read over 8 TB/s); frac_fp64_issue = "
                              "VALU wave-instructions x 4 cycles / (1024 SIMDs x 2.4 GHz) / kernel time; bound = the larger of the two",
                "frac_of_measured_copy_6290": ach / 6290.0,       # MI355X_MICROARCH.md: 6.29 TB/s measured copy
                "required_bytes_per_row": info["required_bytes_per_row"], "algo_bytes_per_row": info["algo_bytes_per_row"],
                "achieved_algorithmic": ach_a, "frac_algorithmic": ach_a / HBM_PEAK_GBS,
                "kernel": f"{kname} ({h.model})", "kernel_id": info["kernel_id"],
                "kernel_ms": h.kernel_ms, "required_bytes_per_launch": req, "algo_bytes_per_launch": algo,
                "rows_in_launch": info["main_kernel_rows"], "clean_groups": info["n_clean_groups"], "groups": info["n_groups"],
                "kernel_ms_source": "untimed second pass over the same parameter vectors, every launch stamped with HIP events on its stream"}

    valu_table = _valu_table()
    roofs = [roof(h, i) for h, i in zip(handles, infos)]
    dom = int(np.argmax([h.kernel_ms for h in handles]))
    roofline = dict(roofs[dom])
    if concurrent:
        roofline["handles"] = roofs          # (each kernel's own duration WHILE the others share the chip with it)
        tot_req = sum(i["required_bytes_per_row"] * i["main_kernel_rows"] for i in infos)
        roofline["concurrent"] = {"required_bytes_all_handles": tot_req, "ms_per_step": 1e3 * elapsed / args.steps,
                                  "achieved": tot_req / (elapsed / args.steps) / 1e9, "frac": tot_req / (elapsed / args.steps) / 1e9 / HBM_PEAK_GBS,
                                  "note": "the three handles' bytes over the wall time of one step: what the concurrently running kernels move together"}
    profiled = None
    pmc = os.path.join(ROOT, "profiles", "pmc_latest.json")
    if os.path.exists(pmc) and args.config == "c2p":
        try:
            pj = json.load(open(pmc))
            profiled = {"main_kernel_bytes": pj.get("main_kernel_bytes"), "evaluation_bytes": pj.get("hbm_bytes_per_launch"),
                        "source": "profiles/pmc_latest.json", "tag": pj.get("tag"),
                        "note": "PMC counters of an EARLIER rocprofv3 --pmc run of this command (committed profile), "
                                "not measured in this run"}
        except Exception:
            profiled = None
    roofline["traffic_profiled"] = profiled
    # `traffic`: HBM bytes per launch of the dominant kernel from the PMC counters (FETCH_SIZE x 2 + WRITE_SIZE, separate passes, as
    # MI355X_MICROARCH.md prescribes).  Counters cannot be read from inside this process: the figure is the one of the committed
    # rocprofv3 --pmc run of this very command (profiles/pmc_latest.json), given only when that run profiled the kernel family that
    # dominated here.
    pjm = json.load(open(pmc)) if profiled else {}
    if (profiled and profiled.get("main_kernel_bytes") and str(pjm.get("main_kernel", "")).find(roofline["kernel"].split(" ")[0]) >= 0
            and abs(float(pjm.get("algorithmic_bytes_per_launch", 0.0)) - float(roofline.get("algo_bytes_per_launch") or -1.0)) < 0.5):      # the same launch
        roofline["traffic"] = float(profiled["main_kernel_bytes"])
        roofline["traffic_over_required"] = roofline["traffic"] / roofline["required_bytes_per_launch"] if roofline.get("required_bytes_per_launch") else None
        roofline["traffic_source"] = f"profiles/pmc_latest.json (tag {profiled.get('tag')}): rocprofv3 --pmc of this command, an earlier run"
    info0 = infos[dom]
    roofline["whole_evaluation"] = None if eval_ms is None else {
        "gpu_ms": eval_ms,
        "achieved": info0["required_bytes_per_row"] * rows_rank / (eval_ms * 1e-3) / 1e9,
        "frac": info0["required_bytes_per_row"] * rows_rank / (eval_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
        "host_enqueue_ms": host_enq_ms,
        "note": "all kernels of one evaluation incl. the hand-over check and the reduction; from an untimed ssde_eval_device "
                "pass over the same evaluations between HIP events"}
    # ---- per rank: rows, and where the stamped pass's synchronous evaluations spent their time (ssde_last_phase_ms) ----------
    mine = {"rank": rank, "rows": int(rows_rank), "kernel_ms": float(np.mean([h.kernel_ms for h in handles])),
            "ms_per_step": float(np.mean(step_ms)), "comm_ranks_reported": int(max(i["comm_ranks_reported"] for i in infos))}
    if phases:
        for key, out_key in (("kernel", "kernel_stamp_ms"), ("gpu_pre", "gpu_pre_ms"), ("finalize", "finalize_ms"),
                             ("allreduce", "allreduce_wait_ms"), ("readback", "readback_ms"), ("host_enqueue", "host_enqueue_ms"),
                             ("host_total", "host_total_ms")):
            mine[out_key] = float(np.mean([ph[key] for ph in phases]))
    if use_comm:
        per_rank = [None] * world
        dist.all_gather_object(per_rank, mine)
    else:
        per_rank = [mine]

    M, T = args.tracks, args.rows
    if args.config == "c5":
        wl = (f"{CONFIGS['c5']['label']}: BM_SSM + OU_SSM + CTCRW sub-batches of {M} tracks each, ragged lengths U[{T // 2}, {T}], d=2, "
              f"sigma_obs=0.1, 5 % of the rows missing (column 0 only / every column), every parameter free, three handles evaluated "
              f"concurrently; {total_rows} rows in total")
    else:
        wl = (f"{CONFIGS[args.config]['label']}: {M} {args.model} tracks x {T} rows, d=2, constant coefficients, sigma_obs/tau/nu free, "
              f"mu fixed, dt=1")
    wl += (f"; {args.scaling} scaling: " + ("the batch split into whole-track shards over the ranks" if args.scaling == "strong"
                                             else "that many tracks on EVERY rank"))
    line = {
        "metric": "track-timesteps/s nllk+grad",
        "value": value, "unit": "track-timesteps/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": 1e3 * elapsed / args.steps, "ms_per_step_stamped": stamped_ms,
        "extra": {"ms_per_step_median": float(np.median(step_ms)), "ms_per_step_min": float(np.min(step_ms)),
                  "ms_per_step_p90": float(np.percentile(step_ms, 90)), "ms_per_step_max": float(np.max(step_ms)),
                  "note": "spread of THIS rank's timed steps (one clock read after every synchronous step); `value` is all rows over "
                          "the whole timed region, max over ranks"},
        "higher_is_better": True, "scaling": args.scaling, "vs_baseline": None,
        "dtype": "f64", "data": "synthetic",
        "config": {"workload": wl, "config": args.config, "tracks": M, "rows_per_track": T, "total_rows": total_rows,
                   "rows_this_rank": rows_rank, "seed": args.seed,
                   "nllk_at_last_step": val,                # the same number at every N under strong scaling (same batch, same theta)
                   "n_free_par": [i["n_free"] for i in infos] if concurrent else info0["n_free"],
                   "engine_path": capi.PATH_NAMES[info0["path"]], "engine_kernel": capi.KERNEL_NAMES.get(info0["kernel_id"], "?"),
                   "comm_ranks_requested": max(i["comm_ranks"] for i in infos) if use_comm else 0,
                   "comm_ranks_reported": max(i["comm_ranks_reported"] for i in infos),       # ncclCommCount of the joined communicator (0: none)
                   "uniform_dt": info0["uniform_dt"], "workgroups": info0["n_kernel_blocks"],
                   "rows_tiled": info0["n_rows_tiled"], "groups": info0["n_groups"], "clean_groups": info0["n_clean_groups"],
                   "lanes_per_track": info0["lanes_per_track"], "window_rows": info0["window"],
                   "window_check": check_max, "window_retries": sum(i["window_retries"] for i in infos),
                   "parallelism": f"tracks x{world}" + ("" if not use_comm else
                                                         ((", ONE ncclAllReduce of the three handles' 2+p doubles per step (ssde_comm_allreduce)" if concurrent
                                                           else ", in-engine ncclAllReduce of 2+p doubles")
                                                          if not host_reduce else
                                                          f", HOST all-reduce over gloo after every evaluation (RCCL could not be brought up: {host_reduce})")),
                   "api": "ssde_eval_device on one stream per handle + one read-back" if concurrent else "ssde_eval (synchronous C ABI call)",
                   "data_generator": "ssde_simulate (HIP, Philox4x32-10 keyed by seed / global track / row)"},
        "roofline": roofline,
        "per_rank": per_rank,
    }
    for h in handles:
        h.eng.close()
    del handles
    # N > 1 under strong scaling: the weak-scaling reading of the same launch beside it (SSDE_BENCH_WEAK_PROBE=1: also with one rank,
    # to rehearse it on a one-GPU box)
    if use_comm and args.scaling == "strong" and args.config in ("c2p", "c4") and not args.no_secondary and \
            (world > 1 or os.environ.get("SSDE_BENCH_WEAK_PROBE")):
        try:
            probe = weak_scaling_probe(args, dev, rank, world, dist, max(3, min(args.steps, 50)))
        except Exception as e:  # noqa: BLE001  (never the line)
            probe = {"workload": "weak-scaling probe", "error": str(e)}
        line["secondary"] = [probe]
    if rank == 0 and world == 1 and not args.no_secondary and args.config == "c2p" and args.model == "CTCRW":
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

            def missing_one(ID, times, obs):           # every track misses ONE row (quiet rows of the general kernel)
                gen = torch.Generator(device=dev)
                gen.manual_seed(8)
                rows = torch.randint(1, T, (M,), device=dev, generator=gen) + T * torch.arange(M, device=dev)
                obs[rows] = float("nan")
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
            sec.append(secondary_workload(f"{M} CTCRW x {T}, regular grid, one missing row in every track", "CTCRW", M, T, dev,
                                          max(3, args.steps // 2), missing_one))
            sec.append(secondary_workload(f"{M} CTCRW x {T} slots of a regular schedule, 5 % of the fixes absent from the data "
                                          f"(intervals of 1-4 steps; laid out on the lattice at create)", "CTCRW", M, T, dev,
                                          max(3, args.steps // 2), absent))
            sec.append(row_varying_workload(M, max(16, T // 10), dev, max(3, args.steps // 2)))
            sec.append(argos_workload(M, T, dev, max(3, args.steps // 2)))
            sec.extend(baseline_config_workloads(dev, max(3, args.steps // 2), M, T))
        except Exception as e:  # the secondary numbers must never take the bench line down
            sec.append({"workload": "failed", "error": str(e)})
        line["secondary"] = line.get("secondary", []) + sec
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
