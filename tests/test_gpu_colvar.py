"""GPU suite: lane = track Kalman lanes with ROW-VARYING tau / nu (kappa, sigma) -- one filter tangent per design column, the
columns dealt to the four waves of a workgroup, rows staged once through LDS (csrc/k_iso_colvar.hip) -- against the oracle, the
golden vectors and the lane = direction path (k_tv.hip) on the same problems.
Reference: nllk_ctcrw.hpp:143-156, 206-241; nllk_ou_ssm.hpp:113-124, 174-207; nllk_bm_ssm.hpp:98-108, 138-169.

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


@pytest.fixture(autouse=True, params=["tangents", "adjoint"])
def _take_the_lane_track_path_from_32_tracks(request, monkeypatch):
    """The engine sends row-varying tau / nu batches to this kernel by their rows (rows x lanes per track >= 4.5 10^6: the measured
    crossover against the lane = direction path, tools/sweep_dispatch.py); the cases below are smaller so that the oracle stays
    quick, and ask for the kernel by track count.
    Every case runs twice: on the forward-tangent kernels this file was written for (SSDE_CV_ADJ=0: the eight-wave pipeline,
    iso_few_kernel, iso_full_kernel) and as the engine dispatches it (the reverse sweep of k_iso_adj.hip wherever it applies)."""
    monkeypatch.setenv("SSDE_DRIFT_MIN_TRACKS", "32")
    monkeypatch.setenv("SSDE_CV_ADJ", "0" if request.param == "tangents" else "1")


def _wpw(eng):
    """waves per (group, window) of the kernel that ran: eight in the pipeline, one in the one-wave kernels"""
    return 8 if eng.info()["kernel_id"] == 11 else 1


def _oracle(pb, par, **kw):
    from oracle_lib import oracle_eval
    return oracle_eval(pb, np.asarray(par, dtype=float), order=1, threads=8, **kw)


def _close(val, grad, oval, ograd):
    assert abs(val - oval) <= 1e-10 * max(1.0, abs(oval)), (val, oval)
    assert np.max(np.abs(grad - ograd)) <= 1e-8 * np.max(np.abs(ograd)) + 1e-10, (grad, ograd, np.abs(grad - ograd))


def _is_colvar(eng):
    inf = eng.info()
    return inf["path"] == PATH_ISO and inf["const_coeff"] == 0


# golden cases whose design columns sit in the rows of tau / nu (kappa) only: X_fe = [1, x] on par[d], a spline on par[d + 1]
CV_GOLD = [r for r in GOLD if r["name"] in ("CTCRW_d1_tv", "CTCRW_d2_tv", "OU_SSM_d1_tv", "OU_SSM_d2_tv")]


@pytest.mark.parametrize("rec", CV_GOLD, ids=[r["name"] for r in CV_GOLD])
def test_golden_row_varying_cases_on_the_lane_track_path(rec, monkeypatch):
    monkeypatch.setenv("SSDE_DRIFT_MIN_TRACKS", "1")
    pb = problem_from_spec(rec)
    eng = capi.Engine(pb)
    assert _is_colvar(eng)
    val, grad = eng.eval(rec["par"], order=1)
    _close(val, grad, rec["expected"]["value"], rec["expected"]["grad"])
    assert abs(eng.eval(rec["par"], order=0) - val) <= 1e-12 * max(1.0, abs(val))
    aest = eng.report(rec["par"])
    assert np.allclose(aest, rec["expected"]["aest_all"], rtol=1e-10, atol=1e-10)
    eng.close()


def _batch(model, d, M, T, k1, k2, seed, fe_slope=False, ragged=False, dt=1.0, same_basis=False):
    """M tracks x T rows; par[d] (log tau / log sigma) gets a k1-column spline of a per-row covariate, par[d + 1] (log nu /
    log kappa) a k2-column one (0: constant); fe_slope adds a fixed-effect slope to par[d]."""
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
        X_fe[d] = np.column_stack([np.ones(n), x])
    if k1:
        X_re[d] = bspline_basis(x, k1)
        S.append(second_difference_penalty(k1))
    if k2 and q > d + 1:
        X_re[d + 1] = bspline_basis(x if same_basis else np.clip(x ** 2, 0, 1), k2)
        S.append(second_difference_penalty(k2))
    pb = capi.Problem(model, ID, t, o, X_fe=X_fe, X_re=X_re, S_list=S)
    rng = np.random.default_rng(seed + 2)
    par = []
    for nm in pb.par_names():
        if nm == "log_sigma_obs":
            par.append(np.log(0.12))
        elif nm.startswith("log_lambda"):
            par.append(0.3)
        elif nm.startswith("coeff_re"):
            par.append(0.15 * rng.standard_normal())
        else:
            par.append(0.1 * rng.standard_normal() + (2.0 if model == "OU_SSM" and nm.startswith("coeff_fe[0]") else 0.0))
    par = np.array(par)
    par[pb.off_fe + pb.fe_off[d]] = np.log(2.0 if model != "BM_SSM" else 0.7)
    return pb, par


@pytest.mark.parametrize("model,d,k1,k2,fe", [("CTCRW", 2, 9, 9, False), ("CTCRW", 1, 5, 0, True), ("CTCRW", 2, 0, 7, False),
                                               ("OU_SSM", 1, 9, 6, False), ("OU_SSM", 2, 4, 12, True), ("BM_SSM", 2, 8, 0, False),
                                               ("BM_SSM", 1, 5, 0, True), ("CTCRW", 2, 11, 11, True)])
def test_long_tracks_with_time_windows_vs_oracle(model, d, k1, k2, fe):
    pb, par = _batch(model, d, 96, 1500, k1, k2, seed=11, fe_slope=fe)
    eng = capi.Engine(pb)
    assert _is_colvar(eng)
    val, grad = eng.eval(par)
    inf = eng.info()
    assert inf["lanes_per_track"] > 1 and inf["window"] > 0 and inf["window_check"] <= 1e-11      # several verified windows
    oval, ograd = _oracle(pb, par)
    _close(val, grad, oval, ograd)
    # bitwise repeatable, and the value-only call agrees
    eng.forget()
    v2, g2 = eng.eval(par)
    assert v2 == val and np.array_equal(g2, grad)
    assert abs(eng.eval(par, order=0) - val) <= 1e-12 * max(1.0, abs(val))
    eng.close()


def test_ragged_tracks_fixed_parameters_and_a_short_step():
    pb, par = _batch("CTCRW", 2, 150, 400, 7, 5, seed=5, ragged=True, dt=0.25)
    fixed = np.zeros(pb.n_par_full, dtype=np.uint8)
    fixed[[0, pb.off_fe + pb.fe_off[0], pb.off_fe + pb.fe_off[3], pb.off_re + 3]] = 1     # sigma_obs, mu_1, the nu intercept and one spline coefficient held
    pb = capi.Problem("CTCRW", pb.id, pb.times, pb.obs, X_fe=pb.X_fe, X_re=pb.X_re, S_list=pb.S_list, par_fixed=fixed)
    eng = capi.Engine(pb)
    assert _is_colvar(eng)
    val, grad = eng.eval(par)
    oval, ograd = _oracle(pb, par)
    _close(val, grad, oval, ograd)
    assert grad[0] == 0.0 and grad[pb.off_re + 3] == 0.0 and grad[pb.off_fe + pb.fe_off[0]] == 0.0
    aest = eng.report(par)
    _, _, oaest = _oracle(pb, par, report=True)
    assert np.max(np.abs(aest - oaest)) <= 1e-9 * max(1.0, np.max(np.abs(oaest)))
    eng.close()


@pytest.mark.parametrize("model,d,k1,k2", [("OU_SSM", 1, 6, 4), ("CTCRW", 2, 5, 4), ("BM_SSM", 2, 7, 0), ("CTCRW", 1, 9, 0)])
@pytest.mark.parametrize("what", ["missing", "irregular", "both"])
def test_missing_rows_and_irregular_grids(model, d, k1, k2, what):
    """nllk_ctcrw.hpp:214-217: a row whose first column is NA is a prediction step; the interval is the row's own."""
    pb, par = _batch(model, d, 96, 700, k1, k2, seed=21)
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
    assert _is_colvar(eng)
    val, grad = eng.eval(par)
    inf = eng.info()
    assert inf["window_check"] <= 1e-11
    _close(val, grad, *_oracle(pb2, par))
    aest = eng.report(par)
    _, _, oaest = _oracle(pb2, par, report=True)
    assert np.allclose(aest, oaest, rtol=1e-9, atol=1e-9)
    eng.close()


def test_lane_track_and_lane_direction_paths_agree(monkeypatch):
    pb, par = _batch("CTCRW", 2, 128, 600, 6, 6, seed=23, same_basis=True)
    e1 = capi.Engine(pb)
    assert _is_colvar(e1)
    v1, g1 = e1.eval(par)
    monkeypatch.setenv("SSDE_NO_COLVAR", "1")
    e2 = capi.Engine(pb)
    assert e2.info()["path"] == PATH_TV
    v2, g2 = e2.eval(par)
    assert abs(v1 - v2) <= 1e-10 * abs(v1) and np.max(np.abs(g1 - g2)) <= 1e-8 * np.max(np.abs(g1))
    e1.close(); e2.close()


def test_a_smooth_shared_by_tau_and_nu_is_streamed_once(monkeypatch):
    """The same design block in both formulas (tau ~ s(x), nu ~ s(x)): the engine finds the equal columns at create (host arrays
    and device-resident ones) and keeps one tile channel per distinct column."""
    import torch
    pb, par = _batch("CTCRW", 2, 96, 700, 7, 7, seed=29, same_basis=True)
    eng = capi.Engine(pb)
    assert _is_colvar(eng)
    inf = eng.info()
    assert inf["algo_bytes_per_row"] == 8.0 * (3 + 14) and inf["required_bytes_per_row"] == 8.0 * (2 + 7)      # regular grid: no dt channel
    val, grad = eng.eval(par)
    _close(val, grad, *_oracle(pb, par))
    # device-resident columns (distinct tensors holding the same numbers), and 14 + 14 columns that only fit because they are shared
    dev = torch.device("cuda:0")
    B = torch.as_tensor(pb.X_re[2], device=dev)
    pbd = capi.Problem.from_torch("CTCRW", torch.as_tensor(pb.id, device=dev), torch.as_tensor(pb.times, device=dev),
                                  torch.as_tensor(pb.obs, device=dev), X_re=[None, None, B, B.clone()], S_list=pb.S_list)
    engd = capi.Engine(pbd)
    assert _is_colvar(engd) and engd.info()["required_bytes_per_row"] == 8.0 * (2 + 7)
    vd, gd = engd.eval(par)
    assert abs(vd - val) <= 1e-12 * abs(val) and np.max(np.abs(gd - grad)) <= 1e-10 * np.max(np.abs(grad))
    pb14, par14 = _batch("CTCRW", 2, 64, 300, 14, 14, seed=31, same_basis=True)
    e14 = capi.Engine(pb14)
    assert _is_colvar(e14)
    _close(*e14.eval(par14), *_oracle(pb14, par14))
    monkeypatch.setenv("SSDE_CV_NO_SHARE", "1")
    e14b = capi.Engine(pb14)
    assert e14b.info()["path"] == PATH_TV                       # 28 distinct channels do not fit
    eng.close(); engd.close(); e14.close(); e14b.close()


def test_window_plan_follows_the_range_the_predictors_actually_reach():
    """The design-based bound on log tau_i, log nu_i (sum of |coefficient| x column range) is loose for a partition-of-unity
    basis; from the second evaluation on the plan uses the range the previous launch saw.  A parameter jump that leaves it is
    caught by the hand-over check and repaired by the retry."""
    pb, par = _batch("CTCRW", 2, 96, 1500, 9, 9, seed=41)
    big = par.copy()
    big[pb.off_re:] = 1.2 * np.sin(np.arange(pb.n_par_full - pb.off_re))          # sum |coef| ~ 7 per block, max |coef| 1.2
    eng = capi.Engine(pb)
    assert _is_colvar(eng)
    v1, g1 = eng.eval(big)
    w_first = eng.info()["window"]
    v2, g2 = eng.eval(big + 1e-6)
    inf = eng.info()
    assert inf["window"] > 0 and inf["lanes_per_track"] > _wpw(eng) and inf["window_check"] <= 1e-11
    assert w_first == 0 or inf["window"] <= w_first                # the measured range never asks for a longer warm-up than the bound
    _close(v1, g1, *_oracle(pb, big))
    _close(v2, g2, *_oracle(pb, big + 1e-6))
    # a jump to slow-forgetting parameters (small tau x large nu corner moves): still right, whatever the plan had to become
    jump = big.copy()
    jump[pb.off_fe + pb.fe_off[2]] += 2.5
    jump[0] += 1.5
    v3, g3 = eng.eval(jump)
    assert eng.info()["window_check"] <= 1e-11
    _close(v3, g3, *_oracle(pb, jump))
    eng.close()


def test_a_block_given_as_a_basis_function_is_materialised_and_streamed():
    """ssde_ppbasis (include/ssde.h): the design block of tau as a piecewise-cubic function of a covariate; the Kalman families
    materialise it at create, and the lane = track path takes the columns like any others."""
    from smoothsde_amd.synth import bspline_ppbasis
    pb0, _ = _batch("CTCRW", 2, 80, 500, 0, 6, seed=51)
    x = np.clip(0.5 + 0.45 * np.sin(np.arange(pb0.n) * 0.013), 0, 1)
    basis = [None, None, bspline_ppbasis(x, 7), None]
    S = [second_difference_penalty(7), second_difference_penalty(6)]
    pb = capi.Problem("CTCRW", pb0.id, pb0.times, pb0.obs, X_re=[None, None, None, pb0.X_re[3]], S_list=S, basis_re=basis)
    dense = capi.Problem("CTCRW", pb0.id, pb0.times, pb0.obs, X_re=[None, None, basis[2].dense(), pb0.X_re[3]], S_list=S)
    rng = np.random.default_rng(2)
    par = np.r_[np.log(0.12), 0.05, -0.03, np.log(2.0), np.log(0.8), 0.3, 0.2, 0.15 * rng.standard_normal(13)]
    eng = capi.Engine(pb)
    assert _is_colvar(eng)
    val, grad = eng.eval(par)
    _close(val, grad, *_oracle(dense, par))
    eng.close()


@pytest.mark.parametrize("seed", range(10))
def test_fuzz_against_the_oracle(seed):
    """Random model / dimension / block sizes / grid / missing rows / fixed parameters / track lengths."""
    rng = np.random.default_rng(1000 + seed)
    model = ["CTCRW", "OU_SSM", "BM_SSM"][rng.integers(3)]
    d = int(rng.integers(1, 3))
    k1 = int(rng.choice([0, 3, 4, 6, 9, 12]))
    k2 = int(rng.choice([0, 3, 5, 8, 11])) if model != "BM_SSM" else 0
    if k1 == 0 and k2 == 0:
        k1 = 3
    fe = bool(rng.integers(2)) and k1 + k2 < 22
    pb, par = _batch(model, d, int(rng.integers(33, 140)), int(rng.integers(60, 900)), k1, k2, seed=2000 + seed, fe_slope=fe,
                     ragged=bool(rng.integers(2)), dt=float(rng.choice([0.25, 1.0, 3.0])), same_basis=bool(rng.integers(2)))
    o, t = pb.obs.copy(), pb.times.copy()
    if rng.integers(2):
        na = rng.random(len(t)) < 0.03
        na[pb.seg_start] = False
        o[na, 0] = np.nan
    if rng.integers(2):
        t = np.cumsum(rng.uniform(0.5, 1.5, len(t)))
    fixed = (rng.random(pb.n_par_full) < 0.15).astype(np.uint8)
    fixed[pb.off_fe:pb.off_fe + d] |= np.uint8(rng.integers(2))   # mu held or free
    pb2 = capi.Problem(model, pb.id, t, o, X_fe=pb.X_fe, X_re=pb.X_re, S_list=pb.S_list, par_fixed=fixed)
    par = par + 0.1 * rng.standard_normal(len(par))
    eng = capi.Engine(pb2)
    assert _is_colvar(eng)
    val, grad = eng.eval(par)
    assert eng.info()["window_check"] <= 1e-11
    _close(val, grad, *_oracle(pb2, par))
    assert np.all(grad[fixed.astype(bool)] == 0.0)
    eng.close()


def _with_h(pb, seed, scale=0.1, floor=0.01):
    """per-row 2 x 2 error ellipses (Argos-like): H_array[,,i] = A_i A_i' + floor I.  (Of the size of the simulated measurement
    noise: with H two orders of magnitude below the state covariance the reference's literal update P' = T P L' + Q is itself
    unstable over hundreds of rows -- the oracle, the lane = direction lanes and these lanes then each give a different number.)"""
    n = pb.n
    A = np.random.default_rng(seed).standard_normal((n, 2, 2)) * scale
    return np.einsum("nij,nkj->ikn", A, A) + floor * np.eye(2)[:, :, None]


H_GOLD = [r for r in GOLD if r["name"] == "CTCRW_d2_tv_H_P0"]


@pytest.mark.parametrize("rec", H_GOLD, ids=[r["name"] for r in H_GOLD])
def test_golden_case_with_h_array_and_a_general_p0_on_the_full_covariance_lanes(rec, monkeypatch):
    monkeypatch.setenv("SSDE_DRIFT_MIN_TRACKS", "1")
    pb = problem_from_spec(rec)
    eng = capi.Engine(pb)
    assert _is_colvar(eng)
    val, grad = eng.eval(rec["par"], order=1)
    _close(val, grad, rec["expected"]["value"], rec["expected"]["grad"])
    aest = eng.report(rec["par"])
    assert np.allclose(aest, rec["expected"]["aest_all"], rtol=1e-10, atol=1e-10)
    eng.close()


@pytest.mark.parametrize("k1,k2,what", [(9, 9, "plain"), (5, 0, "missing"), (0, 7, "irregular"), (6, 6, "both")])
def test_per_row_measurement_covariance_vs_oracle(k1, k2, what, monkeypatch):
    """H_array (nllk_ctcrw.hpp:203-205): 4 x 4 covariance lanes, one tangent of 14 doubles per design column; drift intercepts free."""
    pb, par = _batch("CTCRW", 2, 96, 900, k1, k2, seed=61, same_basis=(what == "plain"))
    o, t = pb.obs.copy(), pb.times.copy()
    rng = np.random.default_rng(6)
    if what in ("missing", "both"):
        na = rng.random(len(t)) < 0.05
        na[pb.seg_start] = False
        o[na, 0] = np.nan
    if what in ("irregular", "both"):
        t = np.cumsum(rng.uniform(0.4, 1.6, len(t)))
    P0 = None
    if what == "both":
        A = rng.standard_normal((4, 4))
        P0 = A @ A.T + np.eye(4)
    pb2 = capi.Problem("CTCRW", pb.id, t, o, X_fe=pb.X_fe, X_re=pb.X_re, S_list=pb.S_list, H=_with_h(pb, 7), P0=P0)
    eng = capi.Engine(pb2)
    assert _is_colvar(eng)
    val, grad = eng.eval(par)
    inf = eng.info()
    assert inf["window_check"] <= 1e-11
    _close(val, grad, *_oracle(pb2, par))
    assert grad[0] == 0.0                                       # log sigma_obs is not in the model when H_array is given
    aest = eng.report(par)
    _, _, oaest = _oracle(pb2, par, report=True)
    assert np.allclose(aest, oaest, rtol=1e-9, atol=1e-9)
    # the lane = direction full-covariance lanes on the same problem
    monkeypatch.setenv("SSDE_NO_COLVAR", "1")
    e2 = capi.Engine(pb2)
    assert e2.info()["path"] == PATH_TV
    v2, g2 = e2.eval(par)
    assert abs(val - v2) <= 1e-10 * abs(val) and np.max(np.abs(grad - g2)) <= 1e-8 * np.max(np.abs(grad))
    eng.close(); e2.close()


@pytest.mark.parametrize("irregular,fix", [(False, ()), (True, ()), (True, (1, 2)), (False, (3,)), (True, (2, 4))])
def test_constant_coefficients_with_error_ellipses_take_the_full_covariance_lanes(irregular, fix, monkeypatch):
    """Tracks with per-row measurement covariances and ONE tau, ONE nu (the usual Argos model): no design column at all; one
    wave per (group, window) runs the 4 x 4 filter and its (at most) four tangents (iso_full_kernel)."""
    ID, t, o = simulate("CTCRW", 80, 700, 2, tau=1.5, nu=0.8, sigma_obs=0.1, seed=71)
    o = o.copy()
    rng = np.random.default_rng(8)
    na = rng.random(len(t)) < 0.03
    na[::700] = False
    o[na, 0] = np.nan
    if irregular:
        t = np.cumsum(rng.uniform(0.4, 1.6, len(t)))
    fixed = np.zeros(5, dtype=np.uint8)
    fixed[list(fix)] = 1
    pb0 = capi.Problem("CTCRW", ID, t, o)
    A = rng.standard_normal((4, 4))
    pb = capi.Problem("CTCRW", ID, t, o, H=_with_h(pb0, 9), par_fixed=fixed, P0=(A @ A.T + np.eye(4)) if irregular else None)
    par = np.array([0.0, 0.05, -0.03, np.log(1.7), np.log(0.7)])
    eng = capi.Engine(pb)
    inf = eng.info()
    assert inf["path"] == PATH_ISO and inf["const_coeff"] == 1 and inf["uniform_dt"] == (0 if irregular else 1)
    val, grad = eng.eval(par)
    inf = eng.info()
    assert inf["lanes_per_track"] > 1 and inf["window_check"] <= 1e-11
    _close(val, grad, *_oracle(pb, par))
    assert np.all(grad[fixed.astype(bool)] == 0.0) and grad[0] == 0.0
    assert abs(eng.eval(par, order=0) - val) <= 1e-12 * max(1.0, abs(val))
    aest = eng.report(par)
    _, _, oaest = _oracle(pb, par, report=True)
    assert np.allclose(aest, oaest, rtol=1e-9, atol=1e-9)
    monkeypatch.setenv("SSDE_NO_COLVAR", "1")
    e2 = capi.Engine(pb)
    assert e2.info()["path"] == PATH_TV
    v2, g2 = e2.eval(par)
    assert abs(val - v2) <= 1e-10 * abs(val) and np.max(np.abs(grad - g2)) <= 1e-8 * np.max(np.abs(grad))
    eng.close(); e2.close()


H2_GOLD = [r for r in GOLD if r["name"] in ("CTCRW_d2_H", "OU_SSM_d2_H", "BM_SSM_d2_H")]


@pytest.mark.parametrize("rec", H2_GOLD, ids=[r["name"] for r in H2_GOLD])
def test_golden_two_column_cases_with_h_array(rec, monkeypatch):
    """Constant coefficients with per-row 2 x 2 measurement covariances: one wave per (group, window) runs the full-covariance
    filter and its tangents (iso_full_kernel: 4 x 4 for CTCRW, 2 x 2 for OU_SSM / BM_SSM)."""
    monkeypatch.setenv("SSDE_DRIFT_MIN_TRACKS", "1")
    pb = problem_from_spec(rec)
    eng = capi.Engine(pb)
    assert eng.info()["path"] == PATH_ISO
    val, grad = eng.eval(rec["par"], order=1)
    _close(val, grad, rec["expected"]["value"], rec["expected"]["grad"])
    aest = eng.report(rec["par"])
    assert np.allclose(aest, rec["expected"]["aest_all"], rtol=1e-10, atol=1e-10)
    eng.close()


@pytest.mark.parametrize("model,k1,k2,irregular", [("OU_SSM", 6, 5, False), ("OU_SSM", 0, 0, True), ("BM_SSM", 7, 0, True), ("BM_SSM", 0, 0, False),
                                                   ("OU_SSM", 0, 8, True)])
def test_scalar_models_with_error_ellipses_vs_oracle(model, k1, k2, irregular, monkeypatch):
    """OU_SSM / BM_SSM, d = 2, H_array (nllk_ou_ssm.hpp:171-172, nllk_bm_ssm.hpp:135-136): full 2 x 2 covariance lanes, through
    the pipeline (row-varying coefficients) or one wave per (group, window) (constant coefficients)."""
    const = k1 == 0 and k2 == 0
    pb, par = _batch(model, 2, 96, 700, k1 or (0 if k2 else 3), k2, seed=91)
    rng = np.random.default_rng(5)
    o, t = pb.obs.copy(), pb.times.copy()
    na = rng.random(pb.n) < 0.03
    na[pb.seg_start] = False
    o[na, 0] = np.nan
    if irregular:
        t = np.cumsum(rng.uniform(0.4, 1.6, len(t)))
    H = _with_h(pb, 11)
    if const:
        pb2 = capi.Problem(model, pb.id, t, o, H=H)
        par = par[:pb2.n_par_full].copy()
    else:
        pb2 = capi.Problem(model, pb.id, t, o, X_fe=pb.X_fe, X_re=pb.X_re, S_list=pb.S_list, H=H)
    eng = capi.Engine(pb2)
    inf = eng.info()
    assert inf["path"] == PATH_ISO and inf["const_coeff"] == (1 if const else 0)
    val, grad = eng.eval(par)
    assert eng.info()["window_check"] <= 1e-11
    _close(val, grad, *_oracle(pb2, par))
    assert grad[0] == 0.0
    aest = eng.report(par)
    _, _, oaest = _oracle(pb2, par, report=True)
    assert np.allclose(aest, oaest, rtol=1e-9, atol=1e-9)
    monkeypatch.setenv("SSDE_NO_COLVAR", "1")
    e2 = capi.Engine(pb2)
    assert e2.info()["path"] == PATH_TV
    v2, g2 = e2.eval(par)
    assert abs(val - v2) <= 1e-10 * max(1.0, abs(val)) and np.max(np.abs(grad - g2)) <= 1e-8 * np.max(np.abs(grad))
    eng.close(); e2.close()


def test_device_resident_h_array_gives_the_same_numbers_as_the_host_array():
    import torch
    ID, t, o = simulate("CTCRW", 70, 300, 2, tau=1.5, nu=0.8, sigma_obs=0.1, seed=77)
    pb0 = capi.Problem("CTCRW", ID, t, o)
    H = _with_h(pb0, 19)
    pbh = capi.Problem("CTCRW", ID, t, o, H=H)
    dev = torch.device("cuda:0")
    pbd = capi.Problem.from_torch("CTCRW", torch.as_tensor(ID, device=dev), torch.as_tensor(t, device=dev), torch.as_tensor(o, device=dev),
                                  H=torch.as_tensor(H, device=dev))
    assert pbd.par_fixed[0] == 1
    par = np.array([0.0, 0.05, -0.03, np.log(1.7), np.log(0.7)])
    eh, ed = capi.Engine(pbh), capi.Engine(pbd)
    vh, gh = eh.eval(par)
    vd, gd = ed.eval(par)
    assert vh == vd and np.array_equal(gh, gd)
    eh.close(); ed.close()


H1_GOLD = [r for r in GOLD if r["name"] in ("CTCRW_d1_H", "OU_SSM_d1_H", "BM_SSM_d1_H")]


@pytest.mark.parametrize("rec", H1_GOLD, ids=[r["name"] for r in H1_GOLD])
def test_golden_one_column_cases_with_h_array(rec, monkeypatch):
    """One response column: H_array[,,i] is the row's measurement variance; the isotropic lanes run with h = H_i."""
    monkeypatch.setenv("SSDE_DRIFT_MIN_TRACKS", "1")
    pb = problem_from_spec(rec)
    eng = capi.Engine(pb)
    assert eng.info()["path"] == PATH_ISO
    val, grad = eng.eval(rec["par"], order=1)
    _close(val, grad, rec["expected"]["value"], rec["expected"]["grad"])
    eng.close()


@pytest.mark.parametrize("model,k1,k2", [("CTCRW", 6, 5), ("OU_SSM", 0, 7), ("BM_SSM", 5, 0), ("CTCRW", 0, 0), ("OU_SSM", 0, 0)])
def test_one_column_with_per_row_measurement_variance_vs_oracle(model, k1, k2):
    pb, par = _batch(model, 1, 96, 700, k1 or 3, k2, seed=81)
    rng = np.random.default_rng(5)
    o = pb.obs.copy()
    na = rng.random(pb.n) < 0.03
    na[pb.seg_start] = False
    o[na, 0] = np.nan
    H = (0.01 + 0.02 * rng.random(pb.n)).reshape(1, 1, -1)
    if k1 == 0 and k2 == 0:                                      # constant coefficients
        pb2 = capi.Problem(model, pb.id, pb.times, o, H=H)
        par = par[:pb2.n_par_full].copy()
    else:
        pb2 = capi.Problem(model, pb.id, pb.times, o, X_fe=pb.X_fe, X_re=pb.X_re, S_list=pb.S_list, H=H)
    eng = capi.Engine(pb2)
    assert eng.info()["path"] == PATH_ISO
    val, grad = eng.eval(par)
    assert eng.info()["window_check"] <= 1e-11
    _close(val, grad, *_oracle(pb2, par))
    assert grad[0] == 0.0
    eng.close()


@pytest.mark.parametrize("model,d,kmu,k1,k2,with_h", [("CTCRW", 2, (5, 4), 6, 0, False), ("CTCRW", 1, (6,), 0, 5, False), ("OU_SSM", 2, (0, 7), 5, 4, False),
                                                     ("BM_SSM", 2, (4, 3), 6, 0, False), ("OU_SSM", 1, (5,), 4, 0, False), ("CTCRW", 2, (4, 0), 5, 5, True),
                                                     ("OU_SSM", 2, (5, 5), 0, 0, True), ("CTCRW", 2, (0, 6), 0, 0, True)])
def test_mixed_designs_drift_columns_next_to_tau_nu_columns(model, d, kmu, k1, k2, with_h, monkeypatch):
    """mu_a smooth in a covariate AND tau / nu (kappa, sigma) smooth: the drift's design columns are columns of kinds of their
    own, mu_a(i) reaches the filter wave with the row's transition.  With H_array also when only the drift is smooth."""
    pb, par0 = _batch(model, d, 96, 700, k1, k2, seed=101)
    n = pb.n
    rng = np.random.default_rng(12)
    xm = np.clip(0.5 + 0.4 * np.cos(np.arange(n) * 2 * np.pi / 53), 0, 1)
    q = capi.n_sde_par(model, d)
    X_re = list(pb.X_re) if pb.X_re is not None else [None] * q
    S = []
    for a in range(d):
        if kmu[a]:
            X_re[a] = bspline_basis(np.clip(xm ** (1 + a), 0, 1), kmu[a])
    for j in range(q):
        if X_re[j] is not None:
            S.append(second_difference_penalty(X_re[j].shape[1]))
    o = pb.obs.copy()
    na = rng.random(n) < 0.03
    na[pb.seg_start] = False
    o[na, 0] = np.nan
    H = _with_h(pb, 13) if with_h else None
    pb2 = capi.Problem(model, pb.id, pb.times, o, X_re=X_re, S_list=S, H=H)
    par = []
    for nm in pb2.par_names():
        if nm == "log_sigma_obs":
            par.append(np.log(0.12))
        elif nm.startswith("log_lambda"):
            par.append(0.3)
        elif nm.startswith("coeff_re"):
            par.append(0.15 * rng.standard_normal())
        else:
            par.append(0.1 * rng.standard_normal() + (2.0 if model == "OU_SSM" and nm.startswith("coeff_fe[0]") else 0.0))
    par = np.array(par)
    par[pb2.off_fe + pb2.fe_off[d]] = np.log(2.0 if model != "BM_SSM" else 0.7)
    eng = capi.Engine(pb2)
    assert _is_colvar(eng)
    val, grad = eng.eval(par)
    assert eng.info()["window_check"] <= 1e-11
    _close(val, grad, *_oracle(pb2, par))
    aest = eng.report(par)
    _, _, oaest = _oracle(pb2, par, report=True)
    assert np.allclose(aest, oaest, rtol=1e-9, atol=1e-9)
    monkeypatch.setenv("SSDE_NO_COLVAR", "1")
    e2 = capi.Engine(pb2)
    assert e2.info()["path"] == PATH_TV
    v2, g2 = e2.eval(par)
    assert abs(val - v2) <= 1e-10 * max(1.0, abs(val)) and np.max(np.abs(grad - g2)) <= 1e-8 * np.max(np.abs(grad))
    eng.close(); e2.close()


@pytest.mark.parametrize("model,d,which", [("CTCRW", 1, 0), ("CTCRW", 2, 1), ("OU_SSM", 2, 0), ("BM_SSM", 1, 0)])
def test_drift_with_a_fixed_effect_design_of_its_own_next_to_smooth_tau_nu(model, d, which):
    """mu_a ~ 1 + x as a FIXED-effect design (X_fe[a] = [1, x]): that drift has no intercept slot -- its column of ones is a
    streamed column -- and the constant part of mu_a handed to the filter wave is zero (round 4: it was the coefficient of the
    ones column a second time; found by the fuzz once small batches reached these kernels)."""
    pb, par0 = _batch(model, d, 96, 700, 5, 0, seed=77)
    n = pb.n
    q = capi.n_sde_par(model, d)
    x = np.clip(0.5 + 0.4 * np.sin(np.arange(n) * 2 * np.pi / 41), 0, 1)
    X_fe = [None] * q
    X_fe[which] = np.column_stack([np.ones(n), x])
    pb2 = capi.Problem(model, pb.id, pb.times, pb.obs, X_fe=X_fe, X_re=pb.X_re, S_list=pb.S_list)
    rng = np.random.default_rng(5)
    par = 0.1 * rng.standard_normal(pb2.n_par_full)
    par[0] = np.log(0.12)
    for a in range(d):
        par[pb2.off_fe + pb2.fe_off[a]] = (2.0 if model == "OU_SSM" else 0.4) + 0.1 * a
    par[pb2.off_fe + pb2.fe_off[which] + 1] = 0.3
    par[pb2.off_fe + pb2.fe_off[d]] = np.log(2.0 if model != "BM_SSM" else 0.7)
    eng = capi.Engine(pb2)
    assert _is_colvar(eng)
    val, grad = eng.eval(par)
    _close(val, grad, *_oracle(pb2, par))
    eng.close()


def test_bench_size_properties(monkeypatch):
    """At the size tools/bench_colvar.py and bench.py time (10^4 tracks x 10^3 rows, 18 design columns), where the oracle is out of
    reach: the windowed evaluation agrees with the sequential one (one window per track) and with the lane = direction path, the
    value-only call with the gradient call, a repeat is bitwise the same, and the gradient is the derivative of the value along a
    random direction (central difference)."""
    import torch
    dev = torch.device("cuda:0")
    M, T, k = 10_000, 1_000, 9
    ID, times, obs = capi.simulate_device("CTCRW", M, T, 2, tau=1.0, nu=1.0, sigma_obs=0.05, seed=342, device=dev)
    n = M * T
    i = torch.arange(n, device=dev, dtype=torch.float64)
    u = (0.5 + 0.4 * torch.sin(i * (2 * np.pi / 24))).clamp_(0.0, 1.0)
    B = (1.0 - (u[:, None] * (k - 1) - torch.arange(k, device=dev, dtype=torch.float64)[None, :]).abs()).clamp_(min=0.0)
    B2 = (1.0 - ((u ** 2)[:, None] * (k - 1) - torch.arange(k, device=dev, dtype=torch.float64)[None, :]).abs()).clamp_(min=0.0)
    S = second_difference_penalty(k)
    pb = capi.Problem.from_torch("CTCRW", ID, times, obs.contiguous(), X_re=[None, None, B, B2], S_list=[S, S])
    rng = np.random.default_rng(3)
    par = np.r_[np.log(0.05), 0.01, -0.02, 0.1, -0.1, 0.2, 0.3, 0.1 * rng.standard_normal(2 * k)]
    eng = capi.Engine(pb)
    assert _is_colvar(eng)
    val, grad = eng.eval(par)
    inf = eng.info()
    assert inf["lanes_per_track"] > _wpw(eng) and inf["window_check"] <= 1e-11
    v2, g2 = eng.eval(par + 0.0)
    assert v2 == val and np.array_equal(g2, grad)
    assert abs(eng.eval(par, order=0) - val) <= 1e-12 * abs(val)
    dvec = rng.standard_normal(len(par))
    eps = 1e-6
    fd = (eng.eval(par + eps * dvec, order=0) - eng.eval(par - eps * dvec, order=0)) / (2 * eps)
    assert abs(fd - grad @ dvec) <= 1e-6 * abs(grad @ dvec) + 1e-3
    monkeypatch.setenv("SSDE_CHUNKS", "1")
    e1 = capi.Engine(pb)
    v1, g1 = e1.eval(par)
    assert e1.info()["lanes_per_track"] == _wpw(e1)                         # one window per track (x eight waves in the pipeline)
    assert abs(v1 - val) <= 1e-11 * abs(val) and np.max(np.abs(g1 - grad)) <= 1e-9 * np.max(np.abs(grad))
    monkeypatch.delenv("SSDE_CHUNKS")
    monkeypatch.setenv("SSDE_NO_COLVAR", "1")
    e2 = capi.Engine(pb)
    assert e2.info()["path"] == PATH_TV
    vt, gt = e2.eval(par)
    assert abs(vt - val) <= 1e-10 * abs(val) and np.max(np.abs(gt - grad)) <= 1e-8 * np.max(np.abs(grad))
    eng.close(); e1.close(); e2.close()


def test_argos_batch_size_properties(monkeypatch):
    """10^4 tracks x 2000 rows with error ellipses and constant tau / nu (iso_full_kernel), out of the oracle's reach: windows vs one
    window per track, vs the lane = direction full-covariance lanes, value-only, repeat, directional derivative."""
    import torch
    dev = torch.device("cuda:0")
    M, T = 10_000, 2_000
    ID, times, obs = capi.simulate_device("CTCRW", M, T, 2, tau=2.0, nu=1.0, sigma_obs=0.1, seed=13, device=dev)
    gen = torch.Generator(device=dev)
    gen.manual_seed(17)
    A = 0.1 * torch.randn(M * T, 2, 2, device=dev, dtype=torch.float64, generator=gen)
    Hn = A @ A.transpose(1, 2)
    Hn[:, 0, 0] += 0.01
    Hn[:, 1, 1] += 0.01
    pb = capi.Problem.from_torch("CTCRW", ID, times, obs.contiguous(), H=Hn.permute(1, 2, 0))
    del A
    par = np.array([0.0, 0.02, -0.01, np.log(2.2), np.log(0.9)])
    eng = capi.Engine(pb)
    inf = eng.info()
    assert inf["path"] == PATH_ISO and inf["const_coeff"] == 1
    val, grad = eng.eval(par)
    inf = eng.info()
    assert inf["lanes_per_track"] > 1 and inf["window_check"] <= 1e-11
    v2, g2 = eng.eval(par + 0.0)
    assert v2 == val and np.array_equal(g2, grad) and grad[0] == 0.0
    assert abs(eng.eval(par, order=0) - val) <= 1e-12 * abs(val)
    dvec = np.array([0.0, 0.3, -0.5, 0.7, 0.4])
    eps = 1e-6
    fd = (eng.eval(par + eps * dvec, order=0) - eng.eval(par - eps * dvec, order=0)) / (2 * eps)
    assert abs(fd - grad @ dvec) <= 1e-6 * abs(grad @ dvec) + 1e-2
    monkeypatch.setenv("SSDE_CHUNKS", "1")
    e1 = capi.Engine(pb)
    v1, g1 = e1.eval(par)
    assert e1.info()["lanes_per_track"] == 1
    assert abs(v1 - val) <= 1e-11 * abs(val) and np.max(np.abs(g1 - grad)) <= 1e-9 * np.max(np.abs(grad))
    monkeypatch.delenv("SSDE_CHUNKS")
    monkeypatch.setenv("SSDE_NO_COLVAR", "1")
    e2 = capi.Engine(pb)
    assert e2.info()["path"] == PATH_TV
    vt, gt = e2.eval(par)
    assert abs(vt - val) <= 1e-10 * abs(val) and np.max(np.abs(gt - grad)) <= 1e-8 * np.max(np.abs(grad))
    eng.close(); e1.close(); e2.close()


@pytest.mark.parametrize("model,d,which,what", [("CTCRW", 2, "p1", "plain"), ("CTCRW", 1, "p2", "missing"), ("OU_SSM", 2, "both", "irregular"),
                                                ("BM_SSM", 2, "p1", "both"), ("OU_SSM", 1, "p2", "plain"), ("CTCRW", 2, "both", "both"),
                                                ("OU_SSM", 2, "wide", "both"), ("CTCRW", 1, "wide", "missing"), ("BM_SSM", 1, "wide", "plain")])
def test_linear_covariate_effects_run_on_one_wave_per_window(model, d, which, what, monkeypatch):
    """tau ~ 1 + x and / or nu ~ 1 + x (fixed-effect slopes, no smooth): at most four streamed columns and four tangents besides
    the directions the filter carries -- iso_few_kernel; against the oracle and, A/B, the eight-wave pipeline."""
    ID, t, o = simulate(model, 96, 900, d, sigma_obs=0.1, seed=111,
                        **dict(CTCRW=dict(tau=1.5, nu=0.8), OU_SSM=dict(mu=2.0, tau=2.0, kappa=1.0, z0=2.0), BM_SSM=dict(sigma=0.7))[model])
    n = len(ID)
    rng = np.random.default_rng(14)
    x = np.clip(0.5 + 0.4 * np.sin(np.arange(n) * 2 * np.pi / 37) + 0.05 * rng.standard_normal(n), 0, 1)
    q = capi.n_sde_par(model, d)
    X_fe = [None] * q
    if which in ("p1", "both"):
        X_fe[d] = np.column_stack([np.ones(n), x])
    if which in ("p2", "both") and q > d + 1:
        X_fe[d + 1] = np.column_stack([np.ones(n), x ** 2])
    if which == "wide":                                          # five to seven streamed columns and tangents: the wide instantiation
        X_fe[d] = np.column_stack([np.ones(n), x, x ** 2, np.sin(3 * x)])
        if q > d + 1:
            X_fe[d + 1] = np.column_stack([np.ones(n), x ** 2, np.cos(2 * x)])
    if all(v is None for v in X_fe):
        X_fe[d] = np.column_stack([np.ones(n), x])
    o = o.copy()
    if what in ("missing", "both"):
        na = rng.random(n) < 0.04
        na[::900] = False
        o[na, 0] = np.nan
    if what in ("irregular", "both"):
        t = np.cumsum(rng.uniform(0.4, 1.6, n))
    pb = capi.Problem(model, ID, t, o, X_fe=X_fe)
    par = 0.1 * rng.standard_normal(pb.n_par_full)
    par[0] = np.log(0.12)
    if model == "OU_SSM":
        par[pb.off_fe:pb.off_fe + d] += 2.0
    par[pb.off_fe + pb.fe_off[d]] = np.log(2.0 if model != "BM_SSM" else 0.7)
    eng = capi.Engine(pb)
    assert _is_colvar(eng)
    val, grad = eng.eval(par)
    inf = eng.info()
    assert inf["lanes_per_track"] > 1 and inf["window_check"] <= 1e-11
    n_win = inf["lanes_per_track"]                                              # (windows x ONE wave)
    _close(val, grad, *_oracle(pb, par))
    assert abs(eng.eval(par, order=0) - val) <= 1e-12 * max(1.0, abs(val))
    aest = eng.report(par)
    _, _, oaest = _oracle(pb, par, report=True)
    assert np.allclose(aest, oaest, rtol=1e-9, atol=1e-9)
    monkeypatch.setenv("SSDE_CV_NO_FEW", "1")
    e2 = capi.Engine(pb)
    v2, g2 = e2.eval(par)
    if e2.info()["kernel_id"] == 11:
        assert e2.info()["lanes_per_track"] % 8 == 0 and e2.info()["lanes_per_track"] != n_win      # the pipeline: eight waves per window
    else:
        assert e2.info()["kernel_id"] == 17                         # ... or the reverse sweep, which takes what the pipeline would
    assert abs(val - v2) <= 1e-11 * max(1.0, abs(val)) and np.max(np.abs(grad - g2)) <= 1e-9 * np.max(np.abs(grad))
    eng.close(); e2.close()


def test_a_response_wider_than_two_columns_runs_this_kernel_as_column_pairs():
    """n_dim = 3: the parts (columns 0-1, column 2) each take the lane = track kernel; the gradient entries of the
    shared tau / nu coefficients are summed over the parts."""
    ID, t, o = simulate("CTCRW", 70, 300, 3, tau=1.5, nu=0.8, sigma_obs=0.1, seed=31)
    n = len(ID)
    x = np.clip(0.5 + 0.4 * np.sin(np.arange(n) * 2 * np.pi / 41), 0, 1)
    B = bspline_basis(x, 5)
    pb = capi.Problem("CTCRW", ID, t, o, X_re=[None, None, None, B, bspline_basis(x ** 2, 4)],
                      S_list=[second_difference_penalty(5), second_difference_penalty(4)])
    rng = np.random.default_rng(2)
    par = np.r_[np.log(0.12), 0.05, -0.03, 0.02, np.log(1.5), np.log(0.8), 0.3, -0.2, 0.2 * rng.standard_normal(9)]
    eng = capi.Engine(pb)
    val, grad = eng.eval(par)
    _close(val, grad, *_oracle(pb, par))
    eng.close()


def test_few_tracks_and_mixed_designs_stay_on_the_lane_direction_path(monkeypatch):
    pb1, _ = _batch("CTCRW", 2, 3, 600, 5, 5, seed=9)
    eng = capi.Engine(pb1)
    assert eng.info()["path"] == PATH_TV
    eng.close()
    monkeypatch.delenv("SSDE_DRIFT_MIN_TRACKS")                 # the engine's own rule: rows x lanes per track >= 4.5 10^6 (round 4: rows decide, not tracks)
    for M, T, want in ((100, 100, PATH_TV), (400, 100, PATH_TV), (400, 2000, PATH_ISO), (64, 12000, PATH_ISO)):
        pbm, _ = _batch("CTCRW", 1, M, T, 9, 0, seed=9)
        eng = capi.Engine(pbm)
        assert eng.info()["path"] == want
        eng.close()
    monkeypatch.setenv("SSDE_DRIFT_MIN_TRACKS", "32")
    # a smooth in the drift AND in tau with SSDE_CV_NO_MU_COLS: back to the lane = direction path
    pb, _ = _batch("CTCRW", 1, 64, 200, 5, 0, seed=3)
    B = bspline_basis(np.clip(np.linspace(0, 1, pb.n), 0, 1), 4)
    pb2 = capi.Problem("CTCRW", pb.id, pb.times, pb.obs, X_re=[B, pb.X_re[1], None],
                       S_list=[second_difference_penalty(4), second_difference_penalty(5)])
    monkeypatch.setenv("SSDE_CV_NO_MU_COLS", "1")
    eng = capi.Engine(pb2)
    assert eng.info()["path"] == PATH_TV
    eng.close()
    monkeypatch.delenv("SSDE_CV_NO_MU_COLS")


def test_sharded_handle_and_one_rank_communicator():
    pb, par = _batch("OU_SSM", 1, 130, 500, 9, 5, seed=13)
    e1 = capi.Engine(pb)
    v1, g1 = e1.eval(par)
    e2 = capi.Engine(pb, devices=[0, 0])
    v2, g2 = e2.eval(par)
    assert abs(v1 - v2) <= 1e-11 * abs(v1) and np.max(np.abs(g1 - g2)) <= 1e-9 * np.max(np.abs(g1))
    e1.comm_init(1, 0, capi.comm_unique_id())
    v3, g3 = e1.eval(par)
    assert abs(v1 - v3) <= 1e-12 * abs(v1) and np.max(np.abs(g1 - g3)) <= 1e-10 * np.max(np.abs(g1))
    e1.close(); e2.close()
    # ... and with error ellipses: the full-covariance lanes, row-varying and constant coefficients
    for k1 in (5, 0):
        pbh, parh = _batch("CTCRW", 2, 130, 400, k1, 0, seed=15)
        if k1 == 0:
            pbh = capi.Problem("CTCRW", pbh.id, pbh.times, pbh.obs, H=_with_h(pbh, 3))
            parh = np.array([0.0, 0.05, -0.03, np.log(1.7), np.log(0.7)])
        else:
            pbh = capi.Problem("CTCRW", pbh.id, pbh.times, pbh.obs, X_re=pbh.X_re, S_list=pbh.S_list, H=_with_h(pbh, 3))
        ea, eb = capi.Engine(pbh), capi.Engine(pbh, devices=[0, 0, 0])
        assert ea.info()["path"] == PATH_ISO
        va, ga = ea.eval(parh)
        vb, gb = eb.eval(parh)
        assert abs(va - vb) <= 1e-11 * abs(va) and np.max(np.abs(ga - gb)) <= 1e-9 * np.max(np.abs(ga))
        ea.close(); eb.close()


# ---- a measurement covariance that COUPLES the response columns, on long tracks: every engine path against the arbiter ----------------------
@pytest.mark.parametrize("path", ["iso_full", "lane=direction dense lanes", "dense_kernel", "reverse sweep, full lanes", "tangent pipeline, full lanes"])
def test_coupling_H_on_long_tracks_every_path_against_the_stabilised_oracle(path, monkeypatch):
    """Argos error ellipses (H_array with off-diagonal entries) on tracks of 1500 rows.  The reference's recursion keeps P as a full
    matrix and amplifies the antisymmetric rounding residue there (tests/test_oracle_golden.py::test_reference_form_loses_...): the
    LITERAL oracle is 1e-3 or worse from the likelihood after a few hundred rows, so the comparison is made with the oracle in
    arbiter mode (P kept symmetric: equal to the binary128 evaluation and to the joint Gaussian where those can be computed).  Every
    engine path that takes such an H -- the 4 x 4 covariance lanes (iso_full_kernel), the lane = direction dense lanes and the
    lane = track dense kernel (ssde_dense.hpp, P symmetrised since round 5), the full-covariance lanes of the reverse sweep and of
    the tangent pipeline (row-varying tau / nu) -- must give THAT value: 1e-10 / 1e-8."""
    from oracle_lib import keep_P_symmetric, oracle_eval
    from smoothsde_amd.synth import bspline_basis, second_difference_penalty, simulate
    M, T = 96, 1500
    ID, times, obs = simulate("CTCRW", M, T, 2, tau=2.0, nu=1.0, sigma_obs=0.1, seed=13)
    n = len(ID)
    rng = np.random.default_rng(17)
    A = 0.05 * rng.standard_normal((n, 2, 2))
    H = np.ascontiguousarray(np.transpose(np.einsum("nij,nkj->nik", A, A) + 0.0025 * np.eye(2), (1, 2, 0)))
    kw, par = dict(par_fixed=np.array([1, 1, 1, 0, 0], dtype=np.uint8)), np.array([0.0, 0.0, 0.0, np.log(2.0), 0.1])
    flags = 0
    if path == "lane=direction dense lanes":
        monkeypatch.setenv("SSDE_NO_COLVAR", "1")
    elif path == "dense_kernel":
        flags = capi.FLAG_FORCE_DENSE
    elif path in ("reverse sweep, full lanes", "tangent pipeline, full lanes"):
        u = 0.5 + 0.4 * np.sin(np.arange(n) * 2 * np.pi / 24)
        B = bspline_basis(u, 5)
        S = second_difference_penalty(5)
        kw = dict(X_re=[None, None, B, B], S_list=[S, S], par_fixed=np.r_[1, 1, 1, 0, 0, 1, 1, np.zeros(10)].astype(np.uint8))
        par = np.r_[0.0, 0.0, 0.0, np.log(2.0), 0.1, 0.0, 0.0, 0.05 * np.sin(np.arange(10))]
        monkeypatch.setenv("SSDE_CV_ADJ", "2" if path.startswith("reverse") else "0")
        monkeypatch.setenv("SSDE_DRIFT_MIN_TRACKS", "1")
    pb = capi.Problem("CTCRW", ID, times, obs, H=H, flags=flags, **kw)
    eng = capi.Engine(pb)
    val, grad = eng.eval(par)
    inf = eng.info()
    want = {"iso_full": 13, "lane=direction dense lanes": 16, "dense_kernel": 14, "reverse sweep, full lanes": 17, "tangent pipeline, full lanes": 11}[path]
    assert inf["kernel_id"] == want, (path, capi.KERNEL_NAMES.get(inf["kernel_id"]))
    assert inf["window_check"] <= capi.WINDOW_TOL
    eng.close()
    lit, _ = oracle_eval(pb, par, order=1, threads=8)
    keep_P_symmetric(True)
    try:
        oval, ograd = oracle_eval(pb, par, order=1, threads=8)
    finally:
        keep_P_symmetric(False)
    assert abs(val - oval) <= 1e-10 * abs(oval), (path, val, oval, lit)
    assert np.max(np.abs(grad - ograd)) <= 1e-8 * np.max(np.abs(ograd)), (path, grad, ograd)
    assert abs(lit - oval) >= 1e-8 * abs(oval), (lit, oval)          # (the literal recursion has long left: that is why the arbiter is used)


@pytest.mark.parametrize("no_tv", [False, True])
def test_general_P0_on_long_ctcrw_tracks_against_the_stabilised_oracle(no_tv, monkeypatch):
    """the other trigger of DESIGN 5c: a P0 with entries BETWEEN the response dimensions (H = sigma_obs^2 I).  As long as the cross-dimension
    entries of P are exact zeros nothing seeds the unstable antisymmetric mode of the reference's full-matrix update; a general P0 makes
    them non-zero and the literal recursion leaves the likelihood by percents within 600 rows (CTCRW only: T is a multiple of the identity
    for OU_SSM / BM_SSM).  The engine's dense lanes keep P symmetric: value and gradient against the oracle in arbiter mode."""
    from oracle_lib import keep_P_symmetric, oracle_eval
    from smoothsde_amd.synth import simulate
    if no_tv:
        monkeypatch.setenv("SSDE_NO_TV", "1")
    ID, times, obs = simulate("CTCRW", 6, 600, 2, tau=2.0, nu=1.0, sigma_obs=0.1, seed=13)
    A = np.random.default_rng(3).standard_normal((4, 4))
    pb = capi.Problem("CTCRW", ID, times, obs, P0=A @ A.T + np.eye(4))
    par = np.array([np.log(0.07), 0.02, -0.01, np.log(2.0), 0.1])
    eng = capi.Engine(pb)
    val, grad = eng.eval(par)
    assert eng.info()["kernel_id"] in ((14,) if no_tv else (14, 16)), capi.KERNEL_NAMES.get(eng.info()["kernel_id"])
    eng.close()
    lit, _ = oracle_eval(pb, par, order=1, threads=4)
    keep_P_symmetric(True)
    try:
        oval, ograd = oracle_eval(pb, par, order=1, threads=4)
    finally:
        keep_P_symmetric(False)
    assert abs(val - oval) <= 1e-10 * abs(oval), (val, oval, lit)
    assert np.max(np.abs(grad - ograd)) <= 1e-8 * np.max(np.abs(ograd)), (grad, ograd)
    assert abs(lit - oval) >= 1e-6 * abs(oval), (lit, oval)
