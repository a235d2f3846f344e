"""GPU suite: shared-covariance Kalman lanes with a ROW-VARYING DRIFT (csrc/k_iso_drift.hip) -- mu smooth in covariates,
tau / nu / kappa / sigma and sigma_obs constant, regular grid, complete tracks -- against the oracle, the golden vectors
and the lane = direction path on the same problems.  Reference: nllk_ctcrw.hpp:143-149, 211-212, 238;
nllk_ou_ssm.hpp:113-124, 174-207; nllk_bm_ssm.hpp:80-86, 138-169.

Tolerances (fp64): value 1e-10 * max(1,|v|); gradient 1e-8 * max|g| + 1e-10 (north-star bar: 1e-8)."""
import numpy as np
import pytest

from cases import problem_from_spec
from golden_io import load_golden
from smoothsde_amd import capi
from smoothsde_amd.synth import bspline_basis, second_difference_penalty, simulate

pytestmark = pytest.mark.gpu
GOLD = load_golden()
PATH_ISO, PATH_TV = 1, 3


@pytest.fixture(autouse=True)
def _take_the_register_lanes_from_32_tracks(monkeypatch):
    """The engine sends smooth-drift batches to these kernels by their ROWS (from 1.5 10^5 on, 5 10^5 for CTCRW: the measured
    crossover against the lane = direction path, tools/sweep_dispatch.py); the cases below are smaller so that the oracle stays
    quick, and ask for the kernels by track count, as round 3's rule did."""
    monkeypatch.setenv("SSDE_DRIFT_MIN_TRACKS", "32")


def _oracle(pb, par, **kw):
    from oracle_lib import oracle_eval
    return oracle_eval(pb, np.asarray(par, dtype=float), order=1, threads=8, **kw)


def _close(val, grad, oval, ograd):
    assert abs(val - oval) <= 1e-10 * max(1.0, abs(oval)), (val, oval)
    assert np.max(np.abs(grad - ograd)) <= 1e-8 * np.max(np.abs(ograd)) + 1e-10, (grad, ograd)


def _is_drift(eng):
    inf = eng.info()
    return inf["path"] == PATH_ISO and inf["const_coeff"] == 0


DRIFT_GOLD = [r for r in GOLD if "drift" in r["name"]]


@pytest.mark.parametrize("rec", DRIFT_GOLD, ids=[r["name"] for r in DRIFT_GOLD])
def test_golden_drift_cases_take_the_shared_covariance_path(rec, monkeypatch):
    pb = problem_from_spec(rec)
    eng = capi.Engine(pb)
    assert _is_drift(eng)
    val, grad = eng.eval(rec["par"], order=1)
    _close(val, grad, rec["expected"]["value"], rec["expected"]["grad"])
    assert eng.eval(rec["par"], order=0) == val
    aest = eng.report(rec["par"])
    assert np.allclose(aest, rec["expected"]["aest_all"], rtol=1e-10, atol=1e-10)
    # the lane = direction kernels on the same problem
    monkeypatch.setenv("SSDE_NO_DRIFT", "1")
    engt = capi.Engine(problem_from_spec(rec))
    assert engt.info()["path"] == PATH_TV
    vt, gt = engt.eval(rec["par"], order=1)
    assert engt.info()["kernel_id"] == 15 and eng.info()["kernel_id"] in (9, 10)        # tv_filter_kernel against the drift lanes
    _close(val, grad, vt, gt)
    eng.close(); engt.close()


def _batch(model, d, M, T, ks, seed, fe_slope=False, ragged=False, dt=1.0):
    """M tracks x T rows; mu_a gets a ks[a]-column spline of a per-row covariate (ks[a] = 0: constant mu_a)."""
    kw = dict(CTCRW=dict(tau=1.5, nu=0.8), OU_SSM=dict(mu=2.0, tau=2.0, kappa=1.0, z0=2.0), BM_SSM=dict(sigma=0.7))[model]
    ID, t, o = simulate(model, M, T, d, sigma_obs=0.1, dt=dt, seed=seed, **kw)
    if ragged:
        rng = np.random.default_rng(seed)
        lens = rng.integers(T // 3, T + 1, size=M)
        keep = np.concatenate([np.arange(T) < L for L in lens])
        ID, o = ID[keep], o[keep]
        t = dt * np.arange(1.0, len(ID) + 1)
    n = len(ID)
    x = np.clip(0.5 + 0.4 * np.sin(np.arange(n) * 2 * np.pi / 37) + 0.05 * np.random.default_rng(seed + 1).standard_normal(n), 0, 1)
    q = capi.n_sde_par(model, d)
    X_fe, X_re, S = [None] * q, [None] * q, []
    if fe_slope:
        X_fe[0] = np.column_stack([np.ones(n), x])
    for a in range(d):
        if ks[a]:
            X_re[a] = bspline_basis(np.clip(x ** (1 + a), 0, 1), ks[a])
            S.append(second_difference_penalty(ks[a]))
    pb = capi.Problem(model, ID, t, o, X_fe=X_fe, X_re=X_re if S else None, S_list=S or None)
    rng = np.random.default_rng(seed + 2)
    par = []
    for nm in pb.par_names():
        if nm == "log_sigma_obs":
            par.append(np.log(0.12))
        elif nm.startswith("log_lambda"):
            par.append(0.3)
        elif nm.startswith("coeff_re"):
            par.append(0.2 * rng.standard_normal())
        else:
            par.append(0.1 * rng.standard_normal() + (2.0 if model == "OU_SSM" and nm.startswith("coeff_fe[0]") else 0.0))
    par = np.array(par)
    par[pb.off_fe + pb.fe_off[d]] = np.log(2.0 if model != "BM_SSM" else 0.7)
    return pb, par


@pytest.mark.parametrize("model,d,ks,fe", [("OU_SSM", 1, (9,), False), ("CTCRW", 2, (9, 9), False), ("CTCRW", 1, (5,), True),
                                            ("BM_SSM", 2, (0, 6), False), ("OU_SSM", 2, (12, 10), True), ("CTCRW", 2, (4, 0), False)])
def test_long_tracks_with_time_windows_vs_oracle(model, d, ks, fe):
    pb, par = _batch(model, d, 96, 1500, ks, seed=11, fe_slope=fe)
    eng = capi.Engine(pb)
    assert _is_drift(eng)
    val, grad = eng.eval(par)
    inf = eng.info()
    assert inf["lanes_per_track"] > 1 and inf["window"] > 0 and inf["window_check"] <= 1e-11      # several verified windows
    oval, ograd = _oracle(pb, par)
    _close(val, grad, oval, ograd)
    # bitwise repeatable, and the value-only call agrees
    eng.forget()
    v2, g2 = eng.eval(par)
    assert v2 == val and np.array_equal(g2, grad)
    eng.close()


def test_ragged_tracks_fixed_parameters_and_a_short_step():
    pb, par = _batch("CTCRW", 2, 150, 400, (7, 5), seed=5, ragged=True, dt=0.25)
    fixed = np.zeros(pb.n_par_full, dtype=np.uint8)
    fixed[[0, pb.off_fe + pb.fe_off[2], pb.off_re + 3]] = 1                    # sigma_obs, tau and one spline coefficient held
    pb = capi.Problem("CTCRW", pb.id, pb.times, pb.obs, X_fe=pb.X_fe, X_re=pb.X_re, S_list=pb.S_list, par_fixed=fixed)
    eng = capi.Engine(pb)
    assert _is_drift(eng)
    val, grad = eng.eval(par)
    oval, ograd = _oracle(pb, par)
    _close(val, grad, oval, ograd)
    assert grad[0] == 0.0 and grad[pb.off_re + 3] == 0.0
    aest = eng.report(par)
    _, _, oaest = _oracle(pb, par, report=True)
    assert np.max(np.abs(aest - oaest)) <= 1e-9 * max(1.0, np.max(np.abs(oaest)))
    eng.close()


@pytest.mark.parametrize("model,d,ks", [("OU_SSM", 1, (6,)), ("CTCRW", 2, (5, 4)), ("BM_SSM", 2, (0, 7)), ("CTCRW", 1, (9,))])
@pytest.mark.parametrize("what", ["missing", "irregular", "both"])
def test_missing_rows_and_irregular_grids_keep_the_register_path(model, d, ks, what):
    """The lanes then carry their own covariance (nllk_ctcrw.hpp:214-217: a row whose first column is NA is a prediction step):
    general step + column recursions with the lane's gains, against the oracle."""
    pb, par = _batch(model, d, 96, 700, ks, seed=21)
    o, t = pb.obs.copy(), pb.times.copy()
    rng = np.random.default_rng(4)
    if what in ("missing", "both"):
        na = rng.random(len(t)) < 0.05
        na[pb.seg_start] = False
        o[na, 0] = np.nan                                       # column 0 decides (the other column may hold a number)
        o[na & (rng.random(len(t)) < 0.5)] = np.nan
    if what in ("irregular", "both"):
        t = np.cumsum(rng.uniform(0.4, 1.6, len(t)))
    pb2 = capi.Problem(model, pb.id, t, o, X_fe=pb.X_fe, X_re=pb.X_re, S_list=pb.S_list)
    eng = capi.Engine(pb2)
    assert _is_drift(eng) and eng.info()["uniform_dt"] == (0 if what != "missing" else 1)
    val, grad = eng.eval(par)
    inf = eng.info()
    assert inf["lanes_per_track"] > 1 and inf["window_check"] <= 1e-11
    _close(val, grad, *_oracle(pb2, par))
    aest = eng.report(par)
    _, _, oaest = _oracle(pb2, par, report=True)
    assert np.allclose(aest, oaest, rtol=1e-9, atol=1e-9)
    eng.close()


def test_general_and_shared_lanes_agree_on_a_complete_regular_batch(monkeypatch):
    pb, par = _batch("CTCRW", 2, 128, 600, (6, 6), seed=23)
    e1 = capi.Engine(pb)
    v1, g1 = e1.eval(par)
    monkeypatch.setenv("SSDE_NO_SHARED", "1")
    e2 = capi.Engine(pb)
    assert _is_drift(e2)
    v2, g2 = e2.eval(par)
    assert e1.info()["kernel_id"] == 9 and e2.info()["kernel_id"] == 10                 # iso_drift_kernel / iso_drift_general_kernel
    assert abs(v1 - v2) <= 1e-11 * abs(v1) and np.max(np.abs(g1 - g2)) <= 1e-9 * np.max(np.abs(g1))
    e1.close(); e2.close()


def test_few_tracks_stay_on_the_lane_direction_path(monkeypatch):
    pb1, par1 = _batch("CTCRW", 2, 3, 600, (5, 5), seed=9)
    eng = capi.Engine(pb1)
    assert eng.info()["path"] == PATH_TV
    eng.close()
    # ... and SSDE_NO_DRIFT_GENERAL sends an incomplete batch back there too
    pb, par = _batch("OU_SSM", 1, 64, 300, (6,), seed=7)
    o = pb.obs.copy()
    o[1234] = np.nan
    monkeypatch.setenv("SSDE_NO_DRIFT_GENERAL", "1")
    eng = capi.Engine(capi.Problem("OU_SSM", pb.id, pb.times, o, X_re=pb.X_re, S_list=pb.S_list))
    assert eng.info()["path"] == PATH_TV
    val, grad = eng.eval(par)
    _close(val, grad, *_oracle(capi.Problem("OU_SSM", pb.id, pb.times, o, X_re=pb.X_re, S_list=pb.S_list), par))
    eng.close()


def test_smooths_on_other_parameters_do_not_take_the_drift_path(monkeypatch):
    monkeypatch.delenv("SSDE_DRIFT_MIN_TRACKS")                 # the engine's own rule (a batch this small: the lane = direction path)
    pb, par = _batch("CTCRW", 1, 64, 200, (5,), seed=3)
    n = pb.n
    B = bspline_basis(np.clip(np.linspace(0, 1, n), 0, 1), 4)
    pb2 = capi.Problem("CTCRW", pb.id, pb.times, pb.obs, X_re=[pb.X_re[0], B, None],
                       S_list=[second_difference_penalty(5), second_difference_penalty(4)])
    eng = capi.Engine(pb2)
    assert eng.info()["path"] == PATH_TV
    eng.close()


def test_sharded_handle_and_one_rank_communicator_on_the_drift_path():
    pb, par = _batch("OU_SSM", 1, 130, 500, (9,), seed=13)
    e1 = capi.Engine(pb)
    v1, g1 = e1.eval(par)
    em = capi.Engine(pb, devices=[0, 0, 0])
    vm, gm = em.eval(par)
    assert abs(vm - v1) <= 1e-11 * abs(v1) and np.max(np.abs(gm - g1)) <= 1e-9 * np.max(np.abs(g1))
    e1.comm_init(1, 0, capi.comm_unique_id())
    v2, g2 = e1.eval(par + 0.0)
    assert v2 == v1 and np.array_equal(g2, g1)
    e1.close(); em.close()


def test_forced_short_warm_up_is_caught_and_repaired(monkeypatch):
    pb, par = _batch("CTCRW", 1, 64, 2000, (6,), seed=17)
    monkeypatch.setenv("SSDE_WINDOW", "2")
    eng = capi.Engine(pb)
    assert _is_drift(eng)
    val, grad = eng.eval(par)
    inf = eng.info()
    assert inf["window_retries"] >= 1 and inf["window_check"] <= 1e-11
    _close(val, grad, *_oracle(pb, par))
    eng.close()


def test_a_response_wider_than_two_columns_runs_the_drift_kernels_as_column_pairs():
    """n_dim = 3: the parts (columns 0-1, column 2) each take the register path with their own mu columns."""
    ID, t, o = simulate("CTCRW", 70, 300, 3, tau=1.5, nu=0.8, sigma_obs=0.1, seed=31)
    n = len(ID)
    x = np.clip(0.5 + 0.4 * np.sin(np.arange(n) * 2 * np.pi / 41), 0, 1)
    X_re = [bspline_basis(x, 5), None, bspline_basis(x ** 2, 4), None, None]
    pb = capi.Problem("CTCRW", ID, t, o, X_re=X_re, S_list=[second_difference_penalty(5), second_difference_penalty(4)])
    rng = np.random.default_rng(2)
    par = np.r_[np.log(0.12), 0.05, -0.03, 0.02, np.log(1.5), np.log(0.8), 0.3, -0.2, 0.2 * rng.standard_normal(9)]
    eng = capi.Engine(pb)
    val, grad = eng.eval(par)
    _close(val, grad, *_oracle(pb, par))
    eng.close()


# ---- drift blocks given as functions of a covariate (ssde_ppbasis): evaluated by the lanes from the table (k_iso_drift_pp.hip) ----
def _table_batch(model, d, M, T, ks, seed, what="clean"):
    from smoothsde_amd.synth import bspline_ppbasis
    pb0, par = _batch(model, d, M, T, ks, seed)
    n = pb0.n
    x = np.clip(0.5 + 0.4 * np.sin(np.arange(n) * 2 * np.pi / 37) + 0.05 * np.random.default_rng(seed + 1).standard_normal(n), 0, 1)
    q = capi.n_sde_par(model, d)
    basis = [None] * q
    for a in range(d):
        if ks[a]:
            basis[a] = bspline_ppbasis(np.clip(x ** (1 + a), 0, 1), ks[a])
    o, t = pb0.obs.copy(), pb0.times.copy()
    rng = np.random.default_rng(4)
    if what in ("missing", "both"):
        na = rng.random(n) < 0.05
        na[pb0.seg_start] = False
        o[na, 0] = np.nan
    if what in ("irregular", "both"):
        t = np.cumsum(rng.uniform(0.4, 1.6, n))
    pb = capi.Problem(model, pb0.id, t, o, S_list=pb0.S_list, basis_re=basis)
    return pb, par


@pytest.mark.parametrize("model,d,ks", [("OU_SSM", 1, (9,)), ("CTCRW", 2, (6, 6)), ("BM_SSM", 2, (0, 6)), ("CTCRW", 1, (5,)), ("OU_SSM", 2, (12, 10))])
@pytest.mark.parametrize("what", ["clean", "missing", "irregular"])
def test_drift_blocks_given_as_tables_are_evaluated_by_the_lanes(model, d, ks, what, monkeypatch):
    """basis_re on the drift of a state-space model: the tiles carry the block's COVARIATE (8 B/row per block instead of 8 K),
    the lanes evaluate the row's K columns from the piecewise-cubic table in LDS -- value, gradient, REPORT(aest_all) and the
    exact drift Hessian against the oracle (which reads the dense block the table stands for) and against the streamed form."""
    pb, par = _table_batch(model, d, 96, 700, ks, seed=31, what=what)
    nb = sum(1 for k in ks if k)
    if model == "CTCRW" or what != "clean":
        # where the table form is slower than the streamed one (CTCRW; the lanes that carry their own covariance) ssde_create keeps
        # the columns: the kernels exist all the same, SSDE_DRIFT_PP_ALL=1 runs them
        e0 = capi.Engine(pb)
        assert _is_drift(e0) and e0.info()["required_bytes_per_row"] == 8.0 * (d + sum(ks) + (1 if what == "irregular" else 0))
        e0.close()
        monkeypatch.setenv("SSDE_DRIFT_PP_ALL", "1")
    eng = capi.Engine(pb)
    assert _is_drift(eng)
    inf = eng.info()
    assert inf["required_bytes_per_row"] == 8.0 * (d + nb + (1 if what == "irregular" else 0)), inf["required_bytes_per_row"]
    val, grad = eng.eval(par)
    assert eng.info()["window_check"] <= 1e-11
    _close(val, grad, *_oracle(pb, par))
    aest = eng.report(par)
    _, _, oaest = _oracle(pb, par, report=True)
    assert np.allclose(aest, oaest, rtol=1e-9, atol=1e-9)
    monkeypatch.setenv("SSDE_NO_DRIFT_PP", "1")
    e2 = capi.Engine(pb)
    assert _is_drift(e2) and e2.info()["required_bytes_per_row"] == 8.0 * (d + sum(ks) + (1 if what == "irregular" else 0))
    v2, g2 = e2.eval(par)
    assert abs(val - v2) <= 1e-12 * max(1.0, abs(val)) and np.max(np.abs(grad - g2)) <= 1e-10 * np.max(np.abs(grad))
    if what == "clean":                                    # the exact Hessian over the drift coefficients (shared-covariance lanes)
        idx = list(range(pb.off_re, pb.off_re + pb.n_re)) + [pb.off_fe + pb.fe_off[a] for a in range(d)]
        H1, H2 = eng.hess(par, idx), e2.hess(par, idx)
        assert np.max(np.abs(H1 - H2)) <= 1e-10 * np.max(np.abs(H2))
    assert eng.info()["hbm_bytes"] < e2.info()["hbm_bytes"]
    eng.close(); e2.close()


def test_table_drift_on_long_tracks_with_time_windows_and_device_resident_covariates():
    import torch
    from smoothsde_amd.synth import bspline_ppbasis
    pb, par = _table_batch("OU_SSM", 1, 128, 4000, (9,), seed=5)
    eng = capi.Engine(pb)
    val, grad = eng.eval(par)
    inf = eng.info()
    assert _is_drift(eng) and inf["lanes_per_track"] > 1 and inf["window"] > 0 and inf["window_check"] <= 1e-11
    _close(val, grad, *_oracle(pb, par))
    # the same batch handed over as HBM tensors (SSDE_FLAG_DEVICE_DATA): the covariate never exists as a block of columns
    dev = "cuda:0"
    bd = bspline_ppbasis(torch.tensor(np.asarray(pb.basis_re[0].x), device=dev), 9, centre=_centre_of(pb.basis_re[0]))
    pbd = capi.Problem.from_torch("OU_SSM", torch.tensor(pb.id, device=dev), torch.tensor(pb.times, device=dev), torch.tensor(pb.obs, device=dev),
                                  basis_re=[bd, None, None], S_list=pb.S_list)
    ed = capi.Engine(pbd)
    vd, gd = ed.eval(par)
    assert ed.info()["required_bytes_per_row"] == 16.0
    assert vd == val and np.array_equal(gd, grad)
    eng.close(); ed.close()


def _centre_of(basis):
    """the column means a bspline_ppbasis subtracted (its table's constant terms against the uncentred table's)"""
    from smoothsde_amd.synth import bspline_ppbasis
    raw = bspline_ppbasis(np.asarray(basis.x), basis.n_cols, centre=np.zeros(basis.n_cols))
    return raw.coef[0, :, 0] - basis.coef[0, :, 0]
