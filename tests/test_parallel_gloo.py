"""CPU suite: the framework-level N > 1 route (smoothsde_amd/parallel.py: track sharding + one all-reduce + penalty
once) with world_size 2 over gloo.  The per-rank evaluator is the ORACLE here (there is no GPU and no CPU fallback in
the product).  The engine's own multi-GPU routes -- ssde_desc.n_devices and ssde_comm_init_rank, the HIP engine on every
shard / rank and the all-reduce inside ssde_eval -- are covered on the device by tests/test_gpu_parallel.py."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from smoothsde_amd.parallel import shard_rows

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_shard_rows_cover_whole_tracks():
    rng = np.random.default_rng(0)
    lens = rng.integers(1, 40, size=37)
    ID = np.repeat(np.arange(37), lens)
    for world in (1, 2, 3, 8):
        cuts = [shard_rows(ID, world, r) for r in range(world)]
        assert cuts[0][0] == 0 and cuts[-1][1] == len(ID)
        for (a, b), (c, d) in zip(cuts[:-1], cuts[1:]):
            assert b == c
        for lo, hi in cuts:
            if lo < hi and lo > 0:
                assert ID[lo] != ID[lo - 1]


def _worker(rank, world, port, q):
    for p in (ROOT, os.path.join(ROOT, "tests")):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from cases import problem_from_spec
    from golden_io import load_golden
    from oracle_lib import oracle_eval
    from smoothsde_amd import capi
    from smoothsde_amd.parallel import ShardedObjective
    rec = next(r for r in load_golden() if r["name"] == "CTCRW_d2_tv")
    lo, hi = shard_rows(rec["ID"], world, rank)
    sl = slice(lo, hi)
    spec = dict(rec, ID=rec["ID"][sl], times=rec["times"][sl], obs=rec["obs"][sl],
                X_fe=[None if x is None else x[sl] for x in rec["X_fe"]],
                X_re=[None if x is None else x[sl] for x in rec["X_re"]])
    pb = problem_from_spec(spec)
    full = problem_from_spec(rec)

    def local_eval(par):
        v, g = oracle_eval(pb, par, order=1, data_only=True)
        return torch.tensor(np.concatenate([[v], g, [0.0]]))

    def penalty(par):
        v_all, g_all = oracle_eval(full, par, order=1)
        v_dat, g_dat = oracle_eval(full, par, order=1, data_only=True)
        return v_all - v_dat, g_all - g_dat

    obj = ShardedObjective(local_eval, pb.n_par_full, penalty)
    val, grad = obj.eval(rec["par"])
    q.put((rank, val, grad))
    dist.destroy_process_group()


def test_two_rank_gloo_matches_single_process():
    from golden_io import load_golden
    rec = next(r for r in load_golden() if r["name"] == "CTCRW_d2_tv")
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, val, grad in res:
        assert abs(val - rec["expected"]["value"]) <= 1e-11 * abs(rec["expected"]["value"])
        assert np.max(np.abs(grad - rec["expected"]["grad"])) <= 1e-10 * np.max(np.abs(rec["expected"]["grad"]))


def test_sharded_objective_widens_and_relaxes():
    """Window policy of the sharded objective (single process, injected local evaluation): a failed hand-over check
    calls on_window_failure and re-evaluates; after `cooldown` evaluations accepted at the first try on_window_calm is
    called once per earlier widening; a failure doubles the cooldown."""
    import torch
    from smoothsde_amd.parallel import ShardedObjective
    state = dict(boost=1, calls=0, fail_next=False)

    def local_eval(par):
        state["calls"] += 1
        bad = state["fail_next"] and state["boost"] == 1
        return torch.tensor([1.0 + par[0], 2.0, 1e-3 if bad else 1e-14], dtype=torch.float64)

    def widen(factor=4):
        state["boost"] *= 4

    def relax():
        state["boost"] = max(1, state["boost"] // 2)

    obj = ShardedObjective(local_eval, 1, None, on_window_failure=widen, on_window_calm=relax, cooldown=5)
    state["fail_next"] = True
    v, g = obj.eval(np.array([0.5]))
    assert v == 1.5 and state["boost"] == 4 and state["calls"] == 2 and obj.cooldown == 10
    state["fail_next"] = False
    for k in range(9):
        obj.eval(np.array([0.5]))
    assert state["boost"] == 4
    obj.eval(np.array([0.5]))                       # the 10th clean evaluation: one halving back
    assert state["boost"] == 2 and obj._widened == 1
    for k in range(10):
        obj.eval(np.array([0.5]))
    assert state["boost"] == 1 and obj._widened == 0
    for k in range(30):
        obj.eval(np.array([0.5]))
    assert state["boost"] == 1                      # nothing more to undo


def test_sharded_objective_failed_check_is_never_returned_silently():
    """include/ssde.h: a result whose hand-over check exceeds the tolerance is invalid.  Without a handler the
    objective raises; a non-finite nllk ends the attempt without widening (an absurd line-search point must not tax
    every later evaluation); when widening does not help the last resort is one sequential window (factor 0)."""
    import torch
    from smoothsde_amd.parallel import ShardedObjective
    bad = lambda par: torch.tensor([1.0, 2.0, 1e-3], dtype=torch.float64)
    with pytest.raises(RuntimeError, match="hand-over"):
        ShardedObjective(bad, 1, None).eval(np.array([0.0]))
    calls = dict(widen=[], n=0)

    def nan_eval(par):
        calls["n"] += 1
        return torch.tensor([float("nan"), 0.0, float("inf")], dtype=torch.float64)

    obj = ShardedObjective(nan_eval, 1, None, on_window_failure=lambda f=4: calls["widen"].append(f))
    v, g = obj.eval(np.array([0.0]))
    assert np.isnan(v) and calls["n"] == 1 and calls["widen"] == [] and obj._widened == 0 and obj.cooldown == 32

    state = dict(seq=False)

    def stubborn(par):
        return torch.tensor([1.0, 2.0, 0.0 if state["seq"] else 1e-3], dtype=torch.float64)

    def widen(factor=4):
        calls["widen"].append(factor)
        if factor == 0:
            state["seq"] = True

    obj = ShardedObjective(stubborn, 1, None, on_window_failure=widen)
    v, g = obj.eval(np.array([0.0]))
    assert v == 1.0 and calls["widen"][-1] == 0 and calls["widen"][:-1] == [4] * (len(calls["widen"]) - 1)


def _max_worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    sys.path.insert(0, ROOT)
    from smoothsde_amd.parallel import ShardedObjective
    widened = []
    # both ranks just under the tolerance: a SUM compared with tol * world would also pass, a SUM compared with tol
    # would fail; rank 1 above it while rank 0 is at zero: the MAX must fail on BOTH ranks
    state = dict(k=0)

    def local_eval(par):
        chk = [9e-12, (0.0 if rank == 0 else 1.9e-11) if not widened else 1e-14][min(state["k"], 1)]
        state["k"] += 1
        return torch.tensor([1.0 + rank, 0.5, chk], dtype=torch.float64)

    obj = ShardedObjective(local_eval, 1, None, on_window_failure=lambda f=4: widened.append(f))
    v1, _ = obj.eval(np.array([0.0]))
    n_after_first = len(widened)
    v2, _ = obj.eval(np.array([0.0]))
    q.put((rank, v1, v2, n_after_first, len(widened)))
    dist.destroy_process_group()


def test_check_slot_is_reduced_with_max_over_ranks():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_max_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, v1, v2, n1, n2 in res:
        assert v1 == 3.0 and v2 == 3.0          # summed over the two ranks
        assert n1 == 0                           # 9e-12 on both ranks: passes (a plain sum would not)
        assert n2 == 1                           # 1.9e-11 on ONE rank: fails on both (sum vs tol * world would pass)
