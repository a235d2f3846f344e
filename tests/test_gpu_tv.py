"""GPU suite: the row-varying-coefficient isotropic Kalman path (csrc/k_tv.hip: per-row records, lane =
gradient direction, verified time windows) against the oracle, the dense path and the golden vectors.

Tolerances (fp64): value 1e-10 * max(1,|v|); gradient 1e-8 * max|g| + 1e-10 (north-star bar: 1e-8)."""
import numpy as np
import pytest

from cases import problem_from_spec
from golden_io import load_golden
from smoothsde_amd import capi
from smoothsde_amd.synth import bspline_basis, second_difference_penalty, simulate

pytestmark = pytest.mark.gpu

GOLD = load_golden()
PATH_TV = 3


def _oracle(pb, par, **kw):
    from oracle_lib import oracle_eval
    return oracle_eval(pb, np.asarray(par, dtype=float), order=1, threads=8, **kw)


def _close(val, grad, oval, ograd):
    assert abs(val - oval) <= 1e-10 * max(1.0, abs(oval)), (val, oval)
    assert np.max(np.abs(grad - ograd)) <= 1e-8 * np.max(np.abs(ograd)) + 1e-10, (grad, ograd)


def _covariate(n, seed):
    rng = np.random.default_rng(seed)
    x = 0.5 + 0.4 * np.sin(np.arange(n) * 2 * np.pi / 24) + 0.05 * rng.standard_normal(n)
    return np.clip(x, 0, 1)


def elephant_like(T=3000, seed=342, k=9, na_frac=0.0):
    """one 2-D CTCRW track, tau and nu smooth in a covariate (vignette model, smoothSDE.rmd:476-490), mu fixed"""
    ID, t, o = simulate("CTCRW", 1, T, 2, tau=1.0, nu=1.0, sigma_obs=0.05, z0=[572.34, 1675.42], seed=seed)
    if na_frac:
        rng = np.random.default_rng(seed + 1)
        na = rng.random(T) < na_frac
        na[0] = False
        o[na] = np.nan
    B = bspline_basis(_covariate(T, seed), k)
    S = second_difference_penalty(k)
    fixed = np.r_[0, 1, 1, 0, 0, 1, 1, np.zeros(2 * k)].astype(np.uint8)
    pb = capi.Problem("CTCRW", ID, t, o, X_re=[None, None, B, B], S_list=[S, S], par_fixed=fixed)
    par = np.r_[np.log(0.05), 0, 0, 0.1, -0.1, 0.3, -0.2, 0.05 * np.sin(np.arange(2 * k))]
    return pb, par


TV_GOLD = [r for r in GOLD if r["model"] in ("CTCRW", "OU_SSM", "BM_SSM") and r.get("X_fe") is not None
           and r.get("H") is None and r.get("P0") is None and "drift" not in r["name"]]      # (drift-only cases: test_gpu_drift.py)


@pytest.mark.parametrize("rec", TV_GOLD, ids=[r["name"] for r in TV_GOLD])
def test_golden_tv_cases_take_the_tv_path(rec):
    pb = problem_from_spec(rec)
    eng = capi.Engine(pb)
    assert eng.info()["path"] == PATH_TV
    val, grad = eng.eval(rec["par"], order=1)
    _close(val, grad, rec["expected"]["value"], rec["expected"]["grad"])
    assert eng.eval(rec["par"], order=0) == val
    # the dense kernel on the same problem
    engd = capi.Engine(problem_from_spec(rec, flags=capi.FLAG_FORCE_DENSE))
    assert engd.info()["path"] == 2
    vd, gd = engd.eval(rec["par"], order=1)
    _close(val, grad, vd, gd)
    eng.close(); engd.close()


@pytest.mark.parametrize("na_frac", [0.0, 0.05])
def test_single_long_track_uses_verified_windows(na_frac):
    pb, par = elephant_like(3000, na_frac=na_frac)
    eng = capi.Engine(pb)
    val, grad = eng.eval(par)
    info = eng.info()
    assert info["path"] == PATH_TV
    assert info["window"] > 0 and info["lanes_per_track"] > 32, info    # several time windows were used
    assert info["window_check"] <= capi.WINDOW_TOL
    oval, ograd = _oracle(pb, par)
    _close(val, grad, oval, ograd)
    # repeated and value-only evaluations are bitwise identical / consistent
    eng.forget()
    v2, g2 = eng.eval(par)
    assert v2 == val and np.array_equal(g2, grad)
    eng.forget()
    assert abs(eng.eval(par, order=0) - val) <= 1e-12 * abs(val)
    # a different parameter vector (the plan of the previous evaluation is reused)
    par2 = par + 0.05 * np.cos(np.arange(len(par)))
    v3, g3 = eng.eval(par2)
    o3, og3 = _oracle(pb, par2)
    _close(v3, g3, o3, og3)
    eng.close()


def test_too_short_warm_up_is_caught_and_repaired(monkeypatch):
    monkeypatch.setenv("SSDE_WINDOW", "2")
    pb, par = elephant_like(2000)
    par = par.copy()
    par[0] = np.log(2.0)            # large observation error: slow forgetting
    eng = capi.Engine(pb)
    val, grad = eng.eval(par)
    info = eng.info()
    assert info["window_retries"] >= 1, info
    oval, ograd = _oracle(pb, par)
    _close(val, grad, oval, ograd)
    eng.close()


@pytest.mark.parametrize("model,d", [("CTCRW", 1), ("OU_SSM", 1), ("OU_SSM", 2), ("BM_SSM", 1), ("BM_SSM", 2), ("CTCRW", 2)])
def test_ragged_multi_track_batches(model, d):
    """37 ragged tracks with NA rows; mu_1 gets a smooth, par[d] an intercept + slope, everything free:
    a few directions per track, so several tracks share a wavefront"""
    rng = np.random.default_rng(17)
    ID, times, obs = simulate(model, 37, 300, d, mu=0.2 if model != "OU_SSM" else 3.0, seed=8)
    keep = np.ones(len(ID), bool)
    for m in range(37):
        cut = rng.integers(0, 250)
        if cut:
            keep[m * 300 + 300 - cut: m * 300 + 300] = False
    ID, obs = ID[keep], obs[keep]
    n = len(ID)
    times = np.cumsum(rng.uniform(0.5, 1.5, size=n))
    first = np.r_[True, ID[1:] != ID[:-1]]
    obs[(rng.random(n) < 0.04) & ~first] = np.nan
    q = capi.n_sde_par(model, d)
    x = _covariate(n, 3)
    X_fe, X_re = [None] * q, [None] * q
    X_fe[d] = np.column_stack([np.ones(n), x])
    X_re[0] = bspline_basis(x, 5)
    pb = capi.Problem(model, ID, times, obs, X_fe=X_fe, X_re=X_re, S_list=[second_difference_penalty(5)])
    par = 0.1 * np.sin(1.0 + np.arange(pb.n_par_full))
    par[0] = -1.0
    if model == "OU_SSM":
        par[pb.off_fe:pb.off_fe + d] = 3.0
    eng = capi.Engine(pb)
    assert eng.info()["path"] == PATH_TV
    val, grad = eng.eval(par)
    oval, ograd = _oracle(pb, par)
    _close(val, grad, oval, ograd)
    eng.close()


def test_more_than_64_directions():
    """two 40-column smooths + intercepts + sigma_obs = 84 free directions: two direction blocks"""
    T = 700
    ID, t, o = simulate("CTCRW", 2, T, 2, tau=1.5, nu=0.8, sigma_obs=0.1, seed=5)
    n = len(ID)
    B1, B2 = bspline_basis(_covariate(n, 1), 40), bspline_basis(_covariate(n, 2) ** 2, 40)
    S = second_difference_penalty(40)
    pb = capi.Problem("CTCRW", ID, t, o, X_re=[None, None, B1, B2], S_list=[S, S])
    par = np.r_[np.log(0.1), 0.05, -0.05, 0.3, -0.1, 0.2, 0.1, 0.03 * np.sin(np.arange(80))]
    eng = capi.Engine(pb)
    info0 = eng.info()
    assert info0["path"] == PATH_TV
    val, grad = eng.eval(par)
    oval, ograd = _oracle(pb, par)
    _close(val, grad, oval, ograd)
    eng.close()


def test_custom_a0_and_report():
    pb0, par = elephant_like(400)
    a0 = np.array([[572.0, 0.1, 1675.0, -0.1]])
    pb = capi.Problem("CTCRW", pb0.id, pb0.times, pb0.obs, X_re=pb0.X_re, S_list=pb0.S_list,
                      par_fixed=pb0.par_fixed, a0=a0)
    eng = capi.Engine(pb)
    val, grad = eng.eval(par)
    oval, ograd, oaest = _oracle(pb, par, report=True)
    _close(val, grad, oval, ograd)
    aest = eng.report(par)
    assert np.allclose(aest, oaest, rtol=1e-10, atol=1e-10, equal_nan=True)
    eng.forget()
    v2, g2 = eng.eval(par)       # evaluating after a report gives the same answer
    assert v2 == val and np.array_equal(g2, grad)
    eng.close()


def test_device_resident_inputs():
    import torch
    ID, times, obs = simulate("OU_SSM", 5, 500, 1, mu=3.0, seed=3, backend="torch", device="cuda:0")
    n = len(ID)
    x = torch.as_tensor(_covariate(n, 4), device="cuda:0")
    B = torch.stack([torch.cos((k + 1) * np.pi * x) for k in range(6)], dim=1)
    S = second_difference_penalty(6)
    pbd = capi.Problem.from_torch("OU_SSM", ID, times, obs, X_re=[B, None, None], S_list=[S])
    pbh = capi.Problem("OU_SSM", ID.cpu().numpy(), times.cpu().numpy(), obs.cpu().numpy(),
                       X_re=[B.cpu().numpy(), None, None], S_list=[S])
    par = np.r_[-1.0, 3.0, 0.4, 0.1, 0.2, 0.1 * np.sin(np.arange(6))]
    ed, eh = capi.Engine(pbd), capi.Engine(pbh)
    assert ed.info()["path"] == PATH_TV
    vd, gd = ed.eval(par)
    vh, gh = eh.eval(par)
    assert vd == vh and np.array_equal(gd, gh)
    oval, ograd = _oracle(pbh, par)
    _close(vd, gd, oval, ograd)
    ed.close(); eh.close()


# ---- full-covariance lanes: per-row H_array and / or a P0 that is not block-identical ---------------------------
HP_GOLD = [r for r in GOLD if r["model"] in ("CTCRW", "OU_SSM", "BM_SSM") and (r.get("H") is not None or r.get("P0") is not None)]


@pytest.mark.parametrize("rec", HP_GOLD, ids=[r["name"] for r in HP_GOLD])
def test_golden_H_and_P0_cases_take_the_tv_path(rec):
    eng = capi.Engine(problem_from_spec(rec))
    # (a one-dimensional model's P0 is trivially block-identical: the constant-coefficient register path takes it)
    iso_ok = rec.get("H") is None and rec["n_dim"] == 1 and rec.get("X_fe") is None
    assert eng.info()["path"] == (1 if iso_ok else PATH_TV)
    val, grad = eng.eval(rec["par"], order=1)
    _close(val, grad, rec["expected"]["value"], rec["expected"]["grad"])
    assert abs(eng.eval(rec["par"], order=0) - val) <= 1e-12 * max(1.0, abs(val))
    aest = eng.report(rec["par"])
    assert np.allclose(aest, rec["expected"]["aest_all"], rtol=1e-10, atol=1e-10, equal_nan=True)
    eng.close()


def _random_H(n, d, seed):
    rng = np.random.default_rng(seed)
    A = rng.standard_normal((n, d, d)) * 0.15
    return np.einsum("nij,nkj->ikn", A, A) + 0.02 * np.eye(d)[:, :, None]


@pytest.mark.parametrize("model,d", [("CTCRW", 2), ("CTCRW", 1), ("OU_SSM", 2), ("BM_SSM", 1)])
def test_H_array_long_tracks_use_verified_windows(model, d):
    """3 tracks x 1500 rows, per-row error ellipses (H_array), tau (or sigma) smooth in a covariate, NA rows"""
    ID, times, obs = simulate(model, 3, 1500, d, mu=0.0 if model != "OU_SSM" else 3.0, sigma_obs=0.15, seed=21)
    n = len(ID)
    rng = np.random.default_rng(4)
    first = np.r_[True, ID[1:] != ID[:-1]]
    obs[(rng.random(n) < 0.03) & ~first] = np.nan
    q = capi.n_sde_par(model, d)
    X_re = [None] * q
    X_re[d] = bspline_basis(_covariate(n, 6), 7)
    pb = capi.Problem(model, ID, times, obs, X_re=X_re, S_list=[second_difference_penalty(7)], H=_random_H(n, d, 2))
    par = 0.1 * np.cos(np.arange(pb.n_par_full))
    if model == "OU_SSM":
        par[pb.off_fe:pb.off_fe + d] = 3.0
    eng = capi.Engine(pb)
    val, grad = eng.eval(par)
    info = eng.info()
    assert info["path"] == PATH_TV and info["window"] > 0, info
    assert info["window_check"] <= capi.WINDOW_TOL
    oval, ograd = _oracle(pb, par)
    _close(val, grad, oval, ograd)
    assert grad[0] == 0.0                      # log_sigma_obs is mapped when H is supplied (R/sde.R:565, 595)
    eng.close()


def test_constant_coefficients_with_custom_P0():
    """constant coefficients but a full P0: no streamed column at all, still the lane = direction kernel"""
    ID, times, obs = simulate("CTCRW", 5, 400, 2, seed=12)
    A = np.random.default_rng(1).standard_normal((4, 4))
    pb = capi.Problem("CTCRW", ID, times, obs, P0=A @ A.T + np.eye(4))
    par = np.array([-1.2, 0.1, -0.1, 0.4, 0.1])
    eng = capi.Engine(pb)
    assert eng.info()["path"] == PATH_TV
    val, grad = eng.eval(par)
    oval, ograd = _oracle(pb, par)
    _close(val, grad, oval, ograd)
    eng.close()


# ---- ESEAL_SSM (nllk_e_seal_ssm.hpp): scalar lipid-mass lanes --------------------------------------------------------
def _eseal_problem(lengths=(300, 180, 240), seed=3, smooth=True, par_fixed=None):
    from cases import eseal_spec
    spec = eseal_spec("x", seed, list(lengths), variant="tv" if smooth else "const", na_rows=(7, 50, 51))
    pb = problem_from_spec(spec, par_fixed=par_fixed)
    return pb, spec["par"]


def test_eseal_long_tracks_vs_oracle():
    pb, par = _eseal_problem()
    assert pb.par_names()[:3] == ["log_tau", "a1", "log_a2"] and pb.off_fe == 3
    eng = capi.Engine(pb)
    info = eng.info()
    assert info["path"] == PATH_TV and info["sdim"] == 2
    val, grad = eng.eval(par)
    oval, ograd = _oracle(pb, par)
    _close(val, grad, oval, ograd)
    assert eng.eval(par, order=0) == val
    # the priors (inverse gamma on sigma(0)^2 and tau^2, nllk_e_seal_ssm.hpp:212-216) live in ssde_penalty
    pen, gpen = eng.penalty(par)
    dval, dgrad = _oracle(pb, par, data_only=True)
    assert abs((val - pen) - dval) <= 1e-10 * abs(dval)
    assert np.max(np.abs((grad - gpen) - dgrad)) <= 1e-8 * np.max(np.abs(dgrad)) + 1e-10
    eng.close()


def test_eseal_fixed_ssm_parameters_and_errors():
    fixed = np.zeros(11, dtype=np.uint8)
    fixed[[1, 2]] = 1                        # a1, log_a2 held at their values (map), as Schick et al. fix them
    pb, par = _eseal_problem(par_fixed=fixed)
    eng = capi.Engine(pb)
    val, grad = eng.eval(par)
    oval, ograd = _oracle(pb, par)
    _close(val, grad, oval, ograd)
    assert grad[1] == 0.0 and grad[2] == 0.0
    with pytest.raises(capi.EngineError):
        eng.report(par)                      # the ESEAL template has no REPORT(aest_all)
    eng.close()
    with pytest.raises(ValueError):
        capi.Problem("ESEAL_SSM", pb.id, pb.times, pb.obs, a0=pb.a0)          # h and R missing
    bad = pb.a0.copy(); bad[0, 0] = 2.0
    pb_bad = capi.Problem("ESEAL_SSM", pb.id, pb.times, pb.obs, a0=bad, eseal_h=pb.eseal_h, eseal_R=pb.eseal_R)
    with pytest.raises(capi.EngineError):
        capi.Engine(pb_bad)


def test_one_animal_with_error_ellipses_on_a_long_track_against_the_stabilised_oracle():
    """BASELINE config 1's model with an error ellipse on every fix, ONE track of 6000 rows: the lane = direction dense lanes
    (ssde_dense.hpp, P kept symmetric) against the oracle in arbiter mode -- the literal recursion is percents away at this length
     -- and the exact second derivatives of the same lanes against differences of the gradient."""
    from oracle_lib import keep_P_symmetric, oracle_eval
    from smoothsde_amd.synth import bspline_basis, second_difference_penalty, simulate
    ID, times, obs = simulate("CTCRW", 1, 6000, 2, tau=2.0, nu=1.0, sigma_obs=0.1, seed=13)
    n = len(ID)
    A = 0.05 * np.random.default_rng(17).standard_normal((n, 2, 2))
    H = np.ascontiguousarray(np.transpose(np.einsum("nij,nkj->nik", A, A) + 0.0025 * np.eye(2) + np.array([[0, .001], [.001, 0]]), (1, 2, 0)))
    B = bspline_basis(0.5 + 0.4 * np.sin(np.arange(n) * 2 * np.pi / 24), 4)
    S = second_difference_penalty(4)
    pb = capi.Problem("CTCRW", ID, times, obs, H=H, X_re=[None, None, B, B], S_list=[S, S], par_fixed=np.r_[1, 1, 1, 0, 0, 1, 1, np.zeros(8)].astype(np.uint8))
    par = np.r_[0.0, 0.0, 0.0, np.log(2.0), 0.1, 0.0, 0.0, 0.05 * np.sin(np.arange(8))]
    eng = capi.Engine(pb)
    val, grad = eng.eval(par)
    inf = eng.info()
    assert inf["kernel_id"] == 16 and inf["lanes_per_track"] > 64 and inf["window_check"] <= capi.WINDOW_TOL, inf
    idx = [3, 4, 7, 10, 11, 14]
    Hs = eng.hess(par, idx)
    Hfd = np.zeros_like(Hs)
    for j, k in enumerate(idx):
        pp, pm = par.copy(), par.copy()
        pp[k] += 1e-5
        pm[k] -= 1e-5
        Hfd[:, j] = (eng.eval(pp)[1][idx] - eng.eval(pm)[1][idx]) / 2e-5
    eng.close()
    lit = oracle_eval(pb, par, order=0)
    keep_P_symmetric(True)
    try:
        oval, ograd = oracle_eval(pb, par, order=1)
    finally:
        keep_P_symmetric(False)
    assert abs(val - oval) <= 1e-10 * abs(oval) and np.max(np.abs(grad - ograd)) <= 1e-8 * np.max(np.abs(ograd)), (val, oval)
    assert abs(lit - oval) >= 1e-4 * abs(oval), (lit, oval)
    assert np.max(np.abs(Hs - Hfd)) <= 1e-6 * np.max(np.abs(Hfd))
