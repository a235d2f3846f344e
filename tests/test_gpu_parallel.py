"""GPU suite: the engine over several shards / devices / ranks, through the C ABI.

* single-process multi-device handles (ssde_desc.n_devices): on a one-GPU box the same device is listed several times
  (the rehearsal mode: shards summed by a kernel); with >= 2 visible devices the RCCL route (ncclCommInitAll + one
  ncclAllReduce per device) runs as well and must give the one-device numbers;
* the multi-process route (ssde_comm_init_rank): a one-rank communicator exercises librccl's loading, ncclCommInitRank
  and the all-reduce on the evaluation's stream on any box; two ranks on two GPUs when the box has them;
* the memo of ssde_eval.

Tolerances: a sharded evaluation sums the same per-track terms in another order: value 1e-12 relative, gradient
1e-10 * max|g| (the single-engine result itself is compared with the oracle at 1e-10 / 1e-8 as everywhere else)."""
import os
import socket
import sys

import numpy as np
import pytest

from smoothsde_amd import capi
from smoothsde_amd.synth import bspline_basis, second_difference_penalty, simulate

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _oracle(pb, par):
    from oracle_lib import oracle_eval
    return oracle_eval(pb, np.asarray(par, dtype=float), order=1, threads=4)


def _ragged(model, n_tracks, lo, hi, d, seed, na=0.0, irregular=False):
    rng = np.random.default_rng(seed)
    lens = rng.integers(lo, hi, size=n_tracks)
    ID = np.repeat(np.arange(n_tracks), lens).astype(float)
    n = len(ID)
    times = np.cumsum(rng.uniform(0.5, 1.5, n)) if irregular else np.arange(1.0, n + 1)
    obs = np.cumsum(rng.standard_normal((n, d)) * 0.5, axis=0)
    if model in ("OU", "OU_SSM"):
        obs = 2.0 + rng.standard_normal((n, d))
    first = np.r_[True, ID[1:] != ID[:-1]]
    if na > 0:
        miss = (rng.random(n) < na) & ~first
        obs[miss] = np.nan
    return ID, times, obs


def _problems():
    out = []
    ID, t, o = _ragged("CTCRW", 333, 2, 400, 2, 1, na=0.05)
    out.append(("CTCRW ragged NA", capi.Problem("CTCRW", ID, t, o), [-0.7, 0.05, -0.02, 0.4, 0.1]))
    ID, t, o = simulate("CTCRW", 700, 600, 2, tau=2.0, nu=1.0, sigma_obs=0.1, seed=3)
    out.append(("CTCRW regular (shared-covariance path, time windows)", capi.Problem("CTCRW", ID, t, o, par_fixed=[0, 1, 1, 0, 0]),
                [np.log(0.1), 0.0, 0.0, np.log(2.0), 0.0]))
    ID, t, o = _ragged("OU_SSM", 150, 2, 300, 2, 4, irregular=True)
    out.append(("OU_SSM irregular", capi.Problem("OU_SSM", ID, t, o), [-1.0, 0.3, -0.2, 0.6, 0.1]))
    ID, t, o = _ragged("OU", 90, 2, 200, 1, 5)
    x = (np.sin(np.arange(len(ID)) * 0.05) + 1) / 2
    pb = capi.Problem("OU", ID, t, o, X_re=[bspline_basis(x, 6), None, None], S_list=[second_difference_penalty(6)])
    out.append(("OU direct with a smooth (penalty added once)", pb, np.r_[0.8, 0.5, 0.1, 0.3, 0.1 * np.cos(np.arange(6))]))
    ID, t, o = _ragged("CTCRW", 7, 150, 500, 2, 6)
    x = (np.sin(np.arange(len(ID)) * 0.03) + 1) / 2
    pb = capi.Problem("CTCRW", ID, t, o, X_re=[None, None, bspline_basis(x, 5), None], S_list=[second_difference_penalty(5)])
    out.append(("CTCRW row-varying tau", pb, np.r_[np.log(0.3), 0.0, 0.0, 0.1, -0.1, 0.2, 0.1 * np.sin(np.arange(5))]))
    return out


@pytest.mark.parametrize("n_shards", [2, 3])
def test_multi_device_handle_on_one_gpu_matches_single_engine(n_shards):
    for name, pb, par in _problems():
        par = np.asarray(par, dtype=float)
        e1 = capi.Engine(pb)
        v1, g1 = e1.eval(par)
        ov, og = _oracle(pb, par)
        assert abs(v1 - ov) <= 1e-10 * max(1.0, abs(ov)), name
        assert np.max(np.abs(g1 - og)) <= 1e-8 * np.max(np.abs(og)) + 1e-10, name
        em = capi.Engine(pb, devices=[0] * n_shards)
        inf = em.info()
        assert inf["n_devices"] == n_shards and inf["n_rows"] == pb.n and inf["n_tracks"] == pb.n_seg, (name, inf)
        vm, gm = em.eval(par)
        assert abs(vm - v1) <= 1e-12 * max(1.0, abs(v1)), (name, vm, v1)
        assert np.max(np.abs(gm - g1)) <= 1e-10 * np.max(np.abs(g1)) + 1e-12, (name, gm, g1)
        v0 = em.eval(par + 0.01, order=0)
        assert abs(v0 - e1.eval(par + 0.01, order=0)) <= 1e-12 * max(1.0, abs(v0)), name
        if pb.kalman:
            assert np.allclose(em.report(par), e1.report(par), rtol=1e-13, atol=1e-13), name
        e1.close()
        em.close()


def test_multi_device_window_retries_move_in_lockstep(monkeypatch):
    """A deliberately short warm-up (SSDE_WINDOW=2) fails the hand-over check on every shard: the parent widens all
    shards together and the repaired result is the sequential one."""
    ID, t, o = simulate("CTCRW", 256, 3000, 2, tau=2.0, nu=1.0, sigma_obs=0.1, seed=9)
    o[5::97] = np.nan
    o[::3000] = np.where(np.isnan(o[::3000]), 0.0, o[::3000])
    pb = capi.Problem("CTCRW", ID, t, o)
    par = np.array([np.log(0.1), 0.01, -0.01, np.log(2.0), 0.0])
    e1 = capi.Engine(pb)
    v1, g1 = e1.eval(par)
    e1.close()
    monkeypatch.setenv("SSDE_WINDOW", "2")
    em = capi.Engine(pb, devices=[0, 0])
    monkeypatch.delenv("SSDE_WINDOW")
    vm, gm = em.eval(par)
    inf = em.info()
    assert inf["window_retries"] >= 1 and inf["window_check"] <= capi.WINDOW_TOL
    assert abs(vm - v1) <= 1e-11 * abs(v1) and np.max(np.abs(gm - g1)) <= 1e-9 * np.max(np.abs(g1))
    em.close()


def test_more_devices_than_tracks_and_tiny_shards():
    ID = np.repeat(np.arange(3), [1, 5, 2]).astype(float)
    n = len(ID)
    rng = np.random.default_rng(0)
    pb = capi.Problem("BM_SSM", ID, np.arange(1.0, n + 1), rng.standard_normal((n, 1)).cumsum(axis=0))
    par = np.array([-1.0, 0.05, 0.2])
    e1 = capi.Engine(pb)
    em = capi.Engine(pb, devices=[0] * 6)
    assert em.info()["n_devices"] <= 3
    v1, g1 = e1.eval(par)
    vm, gm = em.eval(par)
    assert abs(vm - v1) <= 1e-12 * max(1, abs(v1)) and np.allclose(gm, g1, rtol=1e-11, atol=1e-13)
    e1.close()
    em.close()


def test_real_devices_rccl_all_reduce_in_one_process():
    import torch
    if torch.cuda.device_count() < 2:
        pytest.skip("one visible GPU: the RCCL route of a multi-device handle needs two")
    nd = min(torch.cuda.device_count(), 4)
    for name, pb, par in _problems()[:3]:
        par = np.asarray(par, dtype=float)
        e1 = capi.Engine(pb)
        v1, g1 = e1.eval(par)
        em = capi.Engine(pb, devices=list(range(nd)))
        vm, gm = em.eval(par)
        assert abs(vm - v1) <= 1e-12 * max(1.0, abs(v1)), (name, vm, v1)
        assert np.max(np.abs(gm - g1)) <= 1e-10 * np.max(np.abs(g1)) + 1e-12, name
        e1.close()
        em.close()


def test_one_rank_communicator_runs_the_collective():
    """ncclCommInitRank over a single rank + the all-reduce on the evaluation's stream: same numbers, bitwise."""
    ID, t, o = _ragged("CTCRW", 200, 2, 300, 2, 2, na=0.03)
    pb = capi.Problem("CTCRW", ID, t, o)
    par = np.array([-0.7, 0.05, -0.02, 0.4, 0.1])
    eng = capi.Engine(pb)
    v1, g1 = eng.eval(par)
    eng.comm_init(1, 0, capi.comm_unique_id())
    assert eng.info()["comm_ranks"] == 1
    v2, g2 = eng.eval(par)                          # comm_init dropped the memo: a fresh evaluation + the collective
    assert eng.info()["n_memo_hits"] == 0
    assert v1 == v2 and np.array_equal(g1, g2)
    import torch
    out = torch.zeros(2 + pb.n_par_full, dtype=torch.float64, device="cuda:0")
    s = torch.cuda.Stream()
    eng.eval_device(par, out.data_ptr(), order=1, stream=s.cuda_stream)
    s.synchronize()
    pv, pg = eng.penalty(par)
    res = out.cpu().numpy()
    assert res[0] + pv == v1 and np.array_equal(res[1:-1] + pg, g1)
    eng.close()


def test_one_rank_communicator_keeps_the_eseal_priors_and_the_agreed_decisions():
    """ESEAL_SSM's priors use the total row count and sigma(0) of the WHOLE data (nllk_e_seal_ssm.hpp:212-216):
    ssde_comm_init_rank sums the rows and takes rank 0's sigma(0) row over the communicator -- with one rank that must
    leave every number as it was."""
    from cases import eseal_spec, problem_from_spec
    spec = eseal_spec("x", 3, [300, 180, 240], variant="tv", na_rows=(7, 50, 51))
    pb = problem_from_spec(spec)
    par = np.asarray(spec["par"], dtype=float)
    eng = capi.Engine(pb)
    v1, g1 = eng.eval(par)
    p1 = eng.penalty(par)
    eng.comm_init(1, 0, capi.comm_unique_id())
    v2, g2 = eng.eval(par)
    p2 = eng.penalty(par)
    assert v1 == v2 and np.array_equal(g1, g2) and p1[0] == p2[0] and np.array_equal(p1[1], p2[1])
    eng.close()


def _hetero_rank_main(rank, world, port, q):
    """Rank 0: complete tracks on a regular grid (shared-covariance kernel: the gradient rides along with the value);
    rank 1: tracks with missing rows (general kernel: it does not).  fn(x); gr(x) must still be ONE consistent batch."""
    for p in (ROOT, os.path.join(ROOT, "tests")):
        if p not in sys.path:
            sys.path.insert(0, p)
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.cuda.set_device(rank)
    ID, t, o, lo, hi = _hetero_batch(rank)
    pb = capi.Problem("CTCRW", ID[lo:hi], t[lo:hi], o[lo:hi], device=rank)
    eng = capi.Engine(pb)
    box = [capi.comm_unique_id() if rank == 0 else None]
    dist.broadcast_object_list(box, src=0)
    eng.comm_init(world, rank, box[0])
    par = np.array([np.log(0.1), 0.01, -0.02, np.log(2.0), 0.05])
    v0 = eng.eval(par, order=0)                     # fn(x)
    v1, g1 = eng.eval(par, order=1)                 # gr(x) at the same x
    par2 = par + 0.01
    v2, g2 = eng.eval(par2, order=1)
    q.put((rank, v0, v1, g1, v2, g2, eng.info()["n_evals"]))
    dist.barrier()
    eng.close()
    dist.destroy_process_group()


def _hetero_batch(rank=None):
    ID, t, o = simulate("CTCRW", 256, 500, 2, tau=2.0, nu=1.0, sigma_obs=0.1, seed=5)
    half = 128 * 500
    o = o.copy()
    o[half + 7::53] = np.nan                        # missing rows in the second half of the tracks only
    o[::500] = np.where(np.isnan(o[::500]), 0.0, o[::500])
    lo, hi = (0, half) if rank == 0 else (half, len(ID))
    return ID, t, o, lo, hi


def test_two_ranks_with_different_kernels_agree_on_the_order_of_evaluations():
    import torch
    if torch.cuda.device_count() < 2:
        pytest.skip("one visible GPU: two ranks need two")
    import torch.multiprocessing as mp
    ID, t, o, _, _ = _hetero_batch()
    e1 = capi.Engine(capi.Problem("CTCRW", ID, t, o))
    par = np.array([np.log(0.1), 0.01, -0.02, np.log(2.0), 0.05])
    w1, h1 = e1.eval(par)
    w2, h2 = e1.eval(par + 0.01)
    e1.close()
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_hetero_rank_main, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=300) for _ in procs]
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    n_evals = set()
    for rank, v0, v1, g1, v2, g2, ne in res:
        assert v0 == v1 and abs(v1 - w1) <= 1e-12 * abs(w1) and np.max(np.abs(g1 - h1)) <= 1e-10 * np.max(np.abs(h1))
        assert abs(v2 - w2) <= 1e-12 * abs(w2) and np.max(np.abs(g2 - h2)) <= 1e-10 * np.max(np.abs(h2))
        n_evals.add(ne)
    assert len(n_evals) == 1                        # every rank entered the collective the same number of times


def _rank_main(rank, world, port, q):
    for p in (ROOT, os.path.join(ROOT, "tests")):
        if p not in sys.path:
            sys.path.insert(0, p)
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.cuda.set_device(rank)
    from smoothsde_amd.parallel import shard_rows
    name, pb_full, par = _problems()[0]
    lo, hi = shard_rows(pb_full.id, world, rank)
    pb = capi.Problem("CTCRW", pb_full.id[lo:hi], pb_full.times[lo:hi], pb_full.obs[lo:hi], device=rank)
    eng = capi.Engine(pb)
    box = [capi.comm_unique_id() if rank == 0 else None]
    dist.broadcast_object_list(box, src=0)
    eng.comm_init(world, rank, box[0])
    v, g = eng.eval(np.asarray(par, dtype=float))
    q.put((rank, v, g))
    dist.barrier()
    eng.close()
    dist.destroy_process_group()


def test_two_ranks_two_gpus_in_engine_all_reduce():
    import torch
    if torch.cuda.device_count() < 2:
        pytest.skip("one visible GPU: two ranks need two")
    import torch.multiprocessing as mp
    name, pb, par = _problems()[0]
    e1 = capi.Engine(pb)
    v1, g1 = e1.eval(np.asarray(par, dtype=float))
    e1.close()
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_rank_main, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=300) for _ in procs]
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    for rank, v, g in res:                           # every rank returns the batch value
        assert abs(v - v1) <= 1e-12 * abs(v1) and np.max(np.abs(g - g1)) <= 1e-10 * np.max(np.abs(g1))


def test_memo_fn_then_gr_is_one_evaluation():
    """optim's fn(x); gr(x) (R/sde.R:694-696) through the C ABI: one device evaluation where the gradient rides
    along (shared-covariance and direct kernels), none for a repeated call; a different x evaluates again."""
    ID, t, o = simulate("CTCRW", 128, 400, 2, tau=2.0, nu=1.0, sigma_obs=0.1, seed=3)
    pb = capi.Problem("CTCRW", ID, t, o, par_fixed=[0, 1, 1, 0, 0])
    eng = capi.Engine(pb)
    par = np.array([np.log(0.1), 0.0, 0.0, np.log(2.0), 0.0])
    n0 = eng.info()["n_evals"]
    v = eng.eval(par, order=0)                      # fn(x)
    v2, g = eng.eval(par, order=1)                  # gr(x): answered from the memo
    inf = eng.info()
    assert inf["n_evals"] == n0 + 1 and inf["n_memo_hits"] == 1 and v == v2
    assert eng.eval(par, order=0) == v and eng.info()["n_memo_hits"] == 2
    ov, og = _oracle(pb, par)
    assert abs(v - ov) <= 1e-10 * abs(ov) and np.max(np.abs(g - og)) <= 1e-8 * np.max(np.abs(og)) + 1e-10
    par2 = par.copy()
    par2[0] = np.nextafter(par2[0], 1.0)            # one bit off: not the memoised vector
    eng.eval(par2, order=1)
    assert eng.info()["n_evals"] == n0 + 2
    eng.close()
    # general kernel (missing rows): value-only first (cheaper there), the gradient call evaluates again
    o2 = o.copy()
    o2[7::53] = np.nan
    o2[::400] = o[::400]
    eng = capi.Engine(capi.Problem("CTCRW", ID, t, o2))
    par = np.array([np.log(0.1), 0.01, 0.0, np.log(2.0), 0.0])
    eng.eval(par, order=0)
    eng.eval(par, order=1)
    eng.eval(par, order=0)
    inf = eng.info()
    assert inf["n_memo_hits"] == 1 and inf["n_evals"] >= 2
    eng.close()


@pytest.mark.parametrize("seed", range(0, 64))
def test_multi_device_handle_on_random_problems(seed):
    """The fuzz generator of tests/test_gpu_fuzz.py (every model; ragged and one-row tracks, a0 / P0 / H_array, design
    columns and smooths, decaying columns, ESEAL's h / R, fixed parameters) through a three-shard handle: create_sharded
    has to cut every per-row array the descriptor can carry at the same track boundaries.  Against the single engine:
    value 1e-11, gradient 1e-9 of its largest entry (shards plan their own time windows), aest_all 1e-12."""
    from test_gpu_fuzz import random_problem
    pb, par = random_problem(seed)
    e1 = capi.Engine(pb)
    v1, g1 = e1.eval(par)
    em = capi.Engine(pb, devices=[0, 0, 0])
    vm, gm = em.eval(par)
    inf = em.info()
    ctx = (pb.model, pb.n, pb.n_seg, inf["n_devices"], inf["path"])
    assert inf["n_rows"] == pb.n and inf["n_tracks"] == pb.n_seg, ctx
    if not np.isfinite(v1):
        assert not np.isfinite(vm), ctx
    else:
        assert abs(vm - v1) <= 1e-11 * max(1.0, abs(v1)), (vm, v1, ctx)
        assert np.max(np.abs(gm - g1)) <= 1e-9 * max(np.max(np.abs(g1)), 1e-3), (gm, g1, ctx)
        assert np.all(gm[pb.par_fixed != 0] == 0.0)
        if pb.model in ("CTCRW", "OU_SSM", "BM_SSM"):
            a1, am = e1.report(par), em.report(par)
            assert np.allclose(am, a1, rtol=1e-12, atol=1e-12 * max(1.0, np.nanmax(np.abs(a1))), equal_nan=True), ctx
    e1.close()
    em.close()


def test_multi_device_descriptor_errors():
    import torch
    ID, t, o = simulate("CTCRW", 8, 30, 2, seed=1)
    pb = capi.Problem("CTCRW", ID, t, o)
    with pytest.raises(capi.EngineError, match="no such HIP device"):
        capi.Engine(pb, devices=[0, 4096])
    nd = torch.cuda.device_count()
    if nd >= 2:
        with pytest.raises(capi.EngineError, match="all different"):
            capi.Engine(pb, devices=[0, 0, 1])
    IDt, tt, ot = (torch.tensor(x, device="cuda:0") for x in (ID, t, o))
    pbd = capi.Problem.from_torch("CTCRW", IDt, tt, ot)
    with pytest.raises(capi.EngineError, match="host arrays"):
        capi.Engine(pbd, devices=[0, 0])
    # a parent is evaluated with ssde_eval; the asynchronous entry point says so instead of doing something else
    em = capi.Engine(pb, devices=[0, 0])
    out = torch.zeros(2 + pb.n_par_full, dtype=torch.float64, device="cuda:0")
    with pytest.raises(capi.EngineError, match="ssde_eval"):
        em.eval_device(np.zeros(pb.n_par_full), out.data_ptr())
    em.close()


def test_kernel_stamp_history_covers_every_evaluation():
    """every evaluation stamps its dominant kernel with an event pair of its own: the durations of the last n evaluations
    can be read after a timed region (bench.py) -- most recent first, equal to what ssde_last_kernel_ms said each time"""
    ID, t, o = simulate("CTCRW", 256, 500, 2, tau=2.0, nu=1.0, sigma_obs=0.1, seed=3)
    eng = capi.Engine(capi.Problem("CTCRW", ID, t, o, par_fixed=[0, 1, 1, 0, 0]))
    par = np.array([np.log(0.1), 0.0, 0.0, np.log(2.0), 0.0])
    seen = []
    for k in range(70):
        eng.eval(par + 1e-3 * k)
        seen.append(eng.last_kernel_ms())
    hist = eng.kernel_ms_history(70)
    assert np.all(hist[:64] > 0) and np.all(hist[64:] == 0)          # the ring holds 64 pairs
    if eng.info()["window_retries"] == 0:                            # (a window retry is an evaluation of its own in the history)
        assert np.allclose(hist[:64], seen[::-1][:64], rtol=1e-3, atol=1e-6)   # (the runtime may re-derive a pair's elapsed time with another clock calibration)
    else:
        assert set(seen[-32:]) <= set(hist[:64])
    eng.eval(par)                                                      # answered from... a new vector: one more stamp
    assert eng.kernel_ms_history(2)[1] == hist[0]
    eng.close()
