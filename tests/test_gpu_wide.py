"""GPU suite (-m gpu): responses wider than two columns (n_dim > 2).

The reference takes any n_dim (nllk_ctcrw.hpp:12-24 switches to a log-determinant beyond two dimensions; nllk_sde.hpp:77-84
loops over them); the engine evaluates such a response as pairs of columns behind one handle (csrc/ssde_engine_dist.hip).
Checked here against the oracle, which runs the reference's full n_dim-dimensional matrix recursion.

Tolerances as in test_gpu_parity.py: value 1e-10 relative, gradient 1e-8 * max|g| + 1e-10."""
import numpy as np
import pytest

from cases import make_spec, problem_from_spec
from oracle_lib import oracle_eval
from smoothsde_amd import capi
from smoothsde_amd.capi import na_real

pytestmark = pytest.mark.gpu

VT = 1e-10
LENGTHS = [40, 7, 23, 2, 61, 1, 30]


def _close(val, grad, eval_, egrad):
    assert abs(val - eval_) <= VT * max(1.0, abs(eval_)), (val, eval_)
    assert np.max(np.abs(grad - egrad)) <= 1e-8 * np.max(np.abs(egrad)) + 1e-10, (grad, egrad)


CASES = [
    ("CTCRW", 3, "const", {}), ("CTCRW", 4, "const", dict(irregular=False)), ("CTCRW", 3, "tv", {}), ("CTCRW", 3, "const", dict(fix_mu=True)),
    ("OU_SSM", 3, "const", {}), ("OU_SSM", 5, "tv2", {}), ("OU_SSM", 8, "const", dict(irregular=False)),
    ("BM_SSM", 3, "const", {}), ("BM_SSM", 4, "tv", {}),
    ("OU", 3, "const", {}), ("OU", 3, "tv2", {}), ("OU", 5, "tv", dict(decay=True)),
    ("BM", 4, "tv", {}), ("BM", 3, "const", dict(irregular=False)), ("CIR", 3, "const", {}), ("CIR", 4, "tv", {}),
]


@pytest.mark.parametrize("na_mode", [0, 1])
@pytest.mark.parametrize("model,d,variant,kw", CASES, ids=[f"{m}_d{d}_{v}{'_' + '_'.join(k) if k else ''}" for m, d, v, k in CASES])
def test_wide_response_matches_the_oracle(model, d, variant, kw, na_mode):
    spec = make_spec("wide", model, d, seed=100 + d, lengths=LENGTHS, variant=variant, na_rows=(5, 17, 50, 90),
                     na_mode=na_mode, **kw)
    pb = problem_from_spec(spec)
    eng = capi.Engine(pb)
    val, grad = eng.eval(spec["par"], order=1)
    oval, ograd = oracle_eval(pb, spec["par"], order=1)
    assert np.isfinite(oval)
    _close(val, grad, oval, ograd)
    assert eng.eval(spec["par"], order=0) == val
    info = eng.info()
    assert info["n_rows"] == pb.n and info["n_tracks"] == len(LENGTHS) and info["sdim"] == pb.sdim
    assert info["n_par_full"] == pb.n_par_full and info["n_devices"] == 1
    if model in ("CTCRW", "OU_SSM", "BM_SSM"):
        aest = eng.report(spec["par"])
        _, _, oaest = oracle_eval(pb, spec["par"], order=1, report=True)
        assert np.allclose(aest, oaest, rtol=1e-10, atol=1e-10, equal_nan=True)
    eng.close()


@pytest.mark.parametrize("model,d", [("CTCRW", 3), ("OU_SSM", 3), ("BM_SSM", 5)])
@pytest.mark.parametrize("na_mode", [0, 1])
def test_missing_rows_are_decided_on_column_zero_of_the_whole_response(model, d, na_mode):
    """nllk_ctcrw.hpp:214 tests obs(i, 0) only: a row whose column 0 is missing is skipped for every dimension, whatever
    the other columns hold; a NaN elsewhere in an OBSERVED row is a NaN innovation (the result is NaN)."""
    spec = make_spec("wide_na", model, d, seed=7, lengths=[30, 12, 25], na_mode=na_mode)
    na = na_real() if na_mode == 0 else float("nan")
    obs = spec["obs"]
    obs[4, 0] = na                       # column 0 only: the row is missing although columns 1.. hold numbers
    obs[9, :] = na
    obs[33, 0] = na; obs[33, d - 1] = na
    pb = problem_from_spec(spec)
    eng = capi.Engine(pb)
    val, grad = eng.eval(spec["par"], order=1)
    oval, ograd = oracle_eval(pb, spec["par"], order=1)
    assert np.isfinite(oval)
    _close(val, grad, oval, ograd)
    eng.close()
    # an NA in the last column of an observed row
    obs[20, d - 1] = na
    pb = problem_from_spec(spec)
    eng = capi.Engine(pb)
    val, grad = eng.eval(spec["par"], order=1)
    oval = oracle_eval(pb, spec["par"], order=0)
    assert np.isnan(oval) and np.isnan(val)
    eng.close()


def test_block_diagonal_P0_and_supplied_a0():
    d, model = 3, "CTCRW"
    spec = make_spec("wide_p0", model, d, seed=11, lengths=[25, 14, 31])
    rng = np.random.default_rng(3)
    P0 = np.zeros((6, 6))
    A = rng.standard_normal((4, 4)); P0[:4, :4] = A @ A.T + np.eye(4)       # columns (0, 1) coupled with each other: their own pair
    B = rng.standard_normal((2, 2)); P0[4:, 4:] = B @ B.T + np.eye(2)
    spec["P0"] = P0
    spec["a0"] = rng.standard_normal((3, 6))
    pb = problem_from_spec(spec)
    eng = capi.Engine(pb)
    val, grad = eng.eval(spec["par"], order=1)
    oval, ograd = oracle_eval(pb, spec["par"], order=1)
    _close(val, grad, oval, ograd)
    aest = eng.report(spec["par"])
    _, _, oaest = oracle_eval(pb, spec["par"], order=1, report=True)
    assert np.allclose(aest, oaest, rtol=1e-10, atol=1e-10, equal_nan=True)
    eng.close()


@pytest.mark.parametrize("model,d,variant", [("CTCRW", 3, "const"), ("CTCRW", 4, "const"), ("OU_SSM", 3, "tv"), ("OU_SSM", 4, "const"),
                                             ("BM_SSM", 3, "const"), ("BM_SSM", 4, "tv"), ("CTCRW", 3, "tv"),
                                             # round 5: five to eight columns (k_dense_wide.hip; the covariance of a lane in scratch memory)
                                             ("CTCRW", 5, "const"), ("OU_SSM", 6, "const"), ("BM_SSM", 7, "const"), ("CTCRW", 8, "const"),
                                             ("OU_SSM", 5, "tv"), ("BM_SSM", 8, "tv"), ("CTCRW", 6, "tv")])
@pytest.mark.parametrize("what", ["H", "P0", "both"])
def test_three_and_four_columns_that_couple_run_as_one_filter(model, d, variant, what):
    """A per-row measurement covariance with entries between ALL columns (H_array[,, i] a full d x d matrix), a P0 that couples the
    column pairs, or both: F is a full matrix and the reference evaluates it through atomic::logdet and F.inverse()
    (nllk_ctcrw.hpp:12-24, 203-205, 231-241).  Round 3 refused these; the response now runs as ONE filter over all columns on the
    lane = track general kernel (k_dense.hip: F by LU with partial pivoting).  Value, gradient and filtered states against the
    oracle's full-matrix recursion, missing rows included."""
    spec = make_spec("wide_coupled", model, d, seed=31 + d, lengths=[40, 7, 23, 2, 61, 1, 30], variant=variant, na_rows=(5, 17, 50, 90),
                     with_H=what in ("H", "both"))
    sd = capi.state_dim(model, d)
    if what in ("P0", "both"):
        rng = np.random.default_rng(7 + d)
        A = rng.standard_normal((sd, sd))
        spec["P0"] = A @ A.T / sd + np.eye(sd)                                  # every state component coupled with every other
    pb = problem_from_spec(spec)
    eng = capi.Engine(pb)
    inf = eng.info()
    assert inf["path"] == 2 and inf["n_devices"] == 1 and inf["sdim"] == sd, inf     # one dense engine, not a parent of column pairs
    val, grad = eng.eval(spec["par"], order=1)
    assert eng.info()["kernel_id"] == 14                                              # SSDE_KERNEL_DENSE
    oval, ograd = oracle_eval(pb, spec["par"], order=1)
    assert np.isfinite(oval)
    _close(val, grad, oval, ograd)
    assert eng.eval(spec["par"], order=0) == val
    aest = eng.report(spec["par"])
    _, _, oaest = oracle_eval(pb, spec["par"], order=1, report=True)
    assert np.allclose(aest, oaest, rtol=1e-9, atol=1e-9, equal_nan=True)
    eng.close()


def test_what_still_couples_the_pairs_is_refused_with_a_reason():
    """beyond eight columns, and from device-resident arrays, a coupling entry is still refused (and says what does run)"""
    spec = make_spec("wide_bad", "CTCRW", 9, seed=5, lengths=[20, 20])
    P0 = np.eye(18); P0[0, 4] = P0[4, 0] = 0.1
    with pytest.raises(capi.EngineError, match="P0 must not couple"):
        capi.Engine(problem_from_spec(dict(spec, P0=P0)))
    spec = make_spec("wide_bad", "OU_SSM", 9, seed=5, lengths=[20, 20], with_H=True)        # full 9 x 9 matrices
    with pytest.raises(capi.EngineError, match="H_array.*must not couple.*run as one filter"):
        capi.Engine(problem_from_spec(spec))


@pytest.mark.parametrize("model,d", [("CTCRW", 3), ("CTCRW", 4), ("OU_SSM", 4), ("BM_SSM", 5)])
def test_wide_response_with_a_block_diagonal_H_array(model, d):
    """per-row measurement covariances that couple the columns of a pair only (an error ellipse per pair, independent axis
    errors): F stays block-diagonal (nllk_ctcrw.hpp:223-234 with logdet, :20-22), every column pair runs with its block of
    every row; value, gradient and filtered states against the oracle, which takes the d x d matrices as they are"""
    spec = make_spec("wide_h", model, d, seed=17, lengths=[40, 23, 57], with_H=True)
    H = np.array(spec["H"], dtype=float)                                                   # (d, d, n)
    for i in range(d):
        for j in range(d):
            if i // 2 != j // 2:
                H[i, j, :] = 0.0
    spec["H"] = H
    pb = problem_from_spec(spec)
    eng = capi.Engine(pb)
    val, grad = eng.eval(spec["par"], order=1)
    oval, ograd = oracle_eval(pb, spec["par"], order=1)
    _close(val, grad, oval, ograd)
    aest = eng.report(spec["par"])
    _, _, oaest = oracle_eval(pb, spec["par"], order=1, report=True)
    assert np.allclose(aest, oaest, rtol=1e-9, atol=1e-9, equal_nan=True)
    eng.close()
    if not spec.get("X_re") and not spec.get("X_fe"):
        # the same batch from device-resident arrays: the blocks are cut by a kernel, a coupling entry found by one
        import torch
        dev = torch.device("cuda:0")
        tz = lambda x: torch.as_tensor(np.ascontiguousarray(x), device=dev)
        pbd = capi.Problem.from_torch(model, tz(spec["ID"]), tz(spec["times"]), tz(spec["obs"]), H=tz(H))
        engd = capi.Engine(pbd)
        vd, gd = engd.eval(spec["par"], order=1)
        engd.close()
        assert abs(vd - val) <= 1e-12 * abs(val) and np.max(np.abs(gd - grad)) <= 1e-10 * max(1.0, np.max(np.abs(grad)))
        # one entry between two pairs, found by a kernel: up to eight columns then run as ONE filter from the HBM arrays too (round 5:
        # the same numbers as from host arrays); beyond eight the refusal names the reason
        Hbad = H.copy()
        Hbad[0, d - 1, 5] = Hbad[d - 1, 0, 5] = 0.01
        pbb = capi.Problem.from_torch(model, tz(spec["ID"]), tz(spec["times"]), tz(spec["obs"]), H=tz(Hbad))
        if d <= 8 and model in ("CTCRW", "OU_SSM", "BM_SSM"):
            eb = capi.Engine(pbb)
            vb, gb = eb.eval(spec["par"], order=1)
            assert eb.info()["kernel_id"] == 14 and eb.info()["n_devices"] == 1
            eb.close()
            eh = capi.Engine(problem_from_spec(dict(spec, H=Hbad)))
            vh, gh = eh.eval(spec["par"], order=1)
            eh.close()
            assert abs(vb - vh) <= 1e-12 * max(1.0, abs(vh)) and np.max(np.abs(gb - gh)) <= 1e-11 * max(1.0, np.max(np.abs(gh)))
        else:
            with pytest.raises(capi.EngineError, match="must not couple"):
                capi.Engine(pbb)


@pytest.mark.parametrize("model,d", [("CTCRW", 3), ("OU", 4), ("OU_SSM", 5)])
def test_wide_response_over_several_engines_per_device_and_device_resident_data(model, d):
    """track shards x dimension parts (the one-GPU rehearsal of a multi-device handle), and construction from HBM arrays"""
    import torch
    spec = make_spec("wide_md", model, d, seed=21, lengths=[40, 7, 23, 18, 61, 30], na_rows=(5, 50))
    pb = problem_from_spec(spec)
    eng = capi.Engine(pb)
    val, grad = eng.eval(spec["par"], order=1)
    eng.close()
    eng2 = capi.Engine(pb, devices=[0, 0, 0])
    v2, g2 = eng2.eval(spec["par"], order=1)
    info = eng2.info()
    assert info["n_devices"] == 3 and info["n_rows"] == pb.n and info["n_tracks"] == 6
    assert abs(v2 - val) <= 1e-12 * max(1.0, abs(val)) and np.max(np.abs(g2 - grad)) <= 1e-11 * max(1.0, np.max(np.abs(grad)))
    if model != "OU":
        assert np.allclose(eng2.report(spec["par"]), capi.Engine(pb).report(spec["par"]), rtol=1e-12, atol=1e-12, equal_nan=True)
    eng2.close()
    dev = torch.device("cuda:0")
    pbd = capi.Problem.from_torch(model, torch.as_tensor(spec["ID"], device=dev), torch.as_tensor(spec["times"], device=dev),
                                  torch.as_tensor(spec["obs"], device=dev), na_mode=spec["na_mode"])
    eng3 = capi.Engine(pbd)
    v3, g3 = eng3.eval(spec["par"], order=1)
    assert v3 == val and np.array_equal(g3, grad)
    eng3.close()


def test_asynchronous_evaluation_and_rank_communicator_on_a_wide_response():
    """ssde_eval_device sums the dimension parts on the caller's stream; a one-rank communicator joined on top of them
    leaves the numbers alone"""
    import torch
    spec = make_spec("wide_async", "CTCRW", 3, seed=31, lengths=[50, 33, 64])
    pb = problem_from_spec(spec)
    eng = capi.Engine(pb)
    val, grad = eng.eval(spec["par"], order=1)
    out = torch.zeros(2 + pb.n_par_full, dtype=torch.float64, device="cuda:0")
    eng.eval_device(spec["par"], out.data_ptr(), order=1, stream=0)
    torch.cuda.synchronize()
    o = out.cpu().numpy()
    pen, pgrad = eng.penalty(spec["par"])
    assert abs(o[0] + pen - val) <= 1e-12 * max(1.0, abs(val))
    assert np.max(np.abs(o[1:-1] + pgrad - grad)) <= 1e-11 * max(1.0, np.max(np.abs(grad)))
    eng.comm_init(1, 0, capi.comm_unique_id())
    eng.forget()
    v2, g2 = eng.eval(spec["par"], order=1)
    assert v2 == val and np.array_equal(g2, grad)
    assert eng.info()["comm_ranks"] == 1
    eng.close()


@pytest.mark.parametrize("model", ["CTCRW", "OU_SSM", "BM_SSM"])
def test_wide_batch_on_the_shared_covariance_path_full_windows(model):
    """enough rows for time windows and the shared-covariance kernels in every dimension part; tracks sampled against the oracle"""
    from smoothsde_amd.synth import simulate
    M, T, d = 256, 1500, 3
    ID, times, obs = simulate(model, M, T, d, seed=4)
    pb = capi.Problem(model, ID, times, obs)
    par = np.array([-1.0, 0.1, -0.2, 0.3, 0.2, 0.1])[: pb.n_par_full] if model != "BM_SSM" else np.array([-1.0, 0.1, -0.2, 0.3, 0.1])
    eng = capi.Engine(pb)
    val, grad = eng.eval(par, order=1)
    info = eng.info()
    assert info["path"] == 1 and info["window_check"] <= 1e-11
    oval, ograd = oracle_eval(pb, par, order=1, threads=8)
    _close(val, grad, oval, ograd)
    eng.close()


def test_full_size_wide_batch_is_the_sum_of_its_column_pairs():
    """BASELINE's batch shape (1e4 tracks x 1e4 rows) with a 3-column response: a size-independent property -- without missing
    rows the likelihood is additive over the dimensions, so the wide handle must return what two separate handles over
    columns (0, 1) and (2) return, summed, with the gradient entries landing on the right parameters."""
    import torch
    from smoothsde_amd.synth import simulate
    dev = torch.device("cuda:0")
    M, T = 10_000, 10_000
    ID, times, obs = simulate("CTCRW", M, T, 3, tau=2.0, nu=1.0, sigma_obs=0.1, seed=5, backend="torch", device=dev)
    par = np.array([np.log(0.1), 0.01, -0.02, 0.03, np.log(2.0), 0.05])          # sigma_obs, mu1..3, tau, nu
    eng = capi.Engine(capi.Problem.from_torch("CTCRW", ID, times, obs))
    val, grad = eng.eval(par, order=1)
    info = eng.info()
    assert info["n_rows"] == M * T and info["window_check"] <= 1e-11 and info["required_bytes_per_row"] == 24.0
    eng.close()
    e01 = capi.Engine(capi.Problem.from_torch("CTCRW", ID, times, obs[:, :2].contiguous()))
    v01, g01 = e01.eval(par[[0, 1, 2, 4, 5]], order=1)
    e01.close()
    e2 = capi.Engine(capi.Problem.from_torch("CTCRW", ID, times, obs[:, 2:].contiguous()))
    v2, g2 = e2.eval(par[[0, 3, 4, 5]], order=1)
    e2.close()
    want = np.zeros(6)
    want[[0, 1, 2, 4, 5]] += g01
    want[[0, 3, 4, 5]] += g2
    assert abs(val - (v01 + v2)) <= 1e-12 * abs(val)
    assert np.max(np.abs(grad - want)) <= 1e-11 * np.max(np.abs(want))


@pytest.mark.parametrize("model,d", [("CTCRW", 3), ("OU_SSM", 4), ("CTCRW", 5)])
def test_wide_coupled_response_on_long_tracks_against_the_stabilised_oracle(model, d):
    """a CONSTANT measurement covariance with entries between all neighbouring columns, tracks of 400 rows: the reference's full-matrix
    update amplifies rounding there (the literal oracle is percents away after 400 rows), the engine's dense step keeps P
    symmetric -- value and gradient against the oracle in arbiter mode, and the literal value is checked to HAVE left"""
    from oracle_lib import keep_P_symmetric
    from smoothsde_amd.synth import simulate
    M, T = 12, 400
    ID, times, obs = simulate(model, M, T, d, tau=2.0, nu=1.0, kappa=1.0, sigma=1.0, sigma_obs=0.1, seed=13)
    n = len(ID)
    H1 = np.diag(np.where(np.arange(d) % 2 == 0, 0.005, 0.004)) + 0.002 * (np.eye(d, k=1) + np.eye(d, k=-1))
    H = np.ascontiguousarray(np.transpose(np.tile(H1, (n, 1, 1)), (1, 2, 0)))
    pb = capi.Problem(model, ID, times, obs, H=H)
    par = np.zeros(pb.n_par_full)
    par[1 + d] = np.log(2.0) if model != "BM_SSM" else 0.0
    eng = capi.Engine(pb)
    val, grad = eng.eval(par)
    assert eng.info()["kernel_id"] == 14 and eng.info()["sdim"] == capi.state_dim(model, d)
    eng.close()
    lit, _ = oracle_eval(pb, par, order=1, threads=8)
    keep_P_symmetric(True)
    try:
        oval, ograd = oracle_eval(pb, par, order=1, threads=8)
    finally:
        keep_P_symmetric(False)
    assert abs(val - oval) <= 1e-10 * abs(oval), (val, oval, lit)
    free = pb.par_fixed == 0
    assert np.max(np.abs(grad - ograd)[free]) <= 1e-8 * np.max(np.abs(ograd[free])), (grad, ograd)
    if model == "CTCRW":
        assert abs(lit - oval) >= 1e-6 * abs(oval), (lit, oval)


@pytest.mark.parametrize("model,d,what", [("CTCRW", 3, "H"), ("OU_SSM", 5, "both"), ("BM_SSM", 4, "P0")])
def test_coupled_wide_response_over_track_shards(model, d, what):
    """several devices (rehearsed as two / three shards on the one GPU): whole-track shards as ever, every shard runs all columns as ONE
    filter -- the sum is the single engine's value, gradient and filtered states"""
    spec = make_spec("wide_shards", model, d, seed=41 + d, lengths=[40, 7, 23, 2, 61, 1, 30, 12], variant="const", na_rows=(5, 17, 50, 90),
                     with_H=what in ("H", "both"))
    sd = capi.state_dim(model, d)
    if what in ("P0", "both"):
        A = np.random.default_rng(7 + d).standard_normal((sd, sd))
        spec["P0"] = A @ A.T / sd + np.eye(sd)
    pb = problem_from_spec(spec)
    one = capi.Engine(pb)
    v1, g1 = one.eval(spec["par"])
    a1 = one.report(spec["par"])
    one.close()
    for devs in ([0, 0], [0, 0, 0]):
        eng = capi.Engine(pb, devices=devs)
        inf = eng.info()
        assert inf["n_devices"] == len(devs) and inf["kernel_id"] in (0, 14), inf
        v, g = eng.eval(spec["par"])
        assert eng.info()["kernel_id"] == 14
        assert abs(v - v1) <= 1e-12 * max(1.0, abs(v1)) and np.max(np.abs(g - g1)) <= 1e-11 * np.max(np.abs(g1)), (devs, v, v1)
        assert np.allclose(eng.report(spec["par"]), a1, rtol=1e-12, atol=1e-12, equal_nan=True)
        eng.close()
    oval, ograd = oracle_eval(pb, spec["par"], order=1)
    _close(v1, g1, oval, ograd)
