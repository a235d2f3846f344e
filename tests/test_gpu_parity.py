"""GPU suite (-m gpu): the HIP engine, called through the C ABI, against the committed golden
vectors, against the oracle on seeded inputs, and through size-independent properties.

Tolerances (fp64): value |d| <= 1e-10 * max(1,|v|) against oracle/golden (north-star bar vs the
reference: 1e-8); gradient |d| <= 1e-8 * max|g| + 1e-10."""
import numpy as np
import pytest

from cases import problem_from_spec
from golden_io import load_golden
from smoothsde_amd import capi
from smoothsde_amd.synth import simulate

pytestmark = pytest.mark.gpu

GOLD = load_golden()
VT = 1e-10


def _close(val, grad, eval_, egrad):
    assert abs(val - eval_) <= VT * max(1.0, abs(eval_)), (val, eval_)
    assert np.max(np.abs(grad - egrad)) <= 1e-8 * np.max(np.abs(egrad)) + 1e-10, (grad, egrad)


@pytest.mark.parametrize("rec", GOLD, ids=[r["name"] for r in GOLD])
def test_golden(rec):
    pb = problem_from_spec(rec)
    eng = capi.Engine(pb)
    val, grad = eng.eval(rec["par"], order=1)
    _close(val, grad, rec["expected"]["value"], rec["expected"]["grad"])
    v0 = eng.eval(rec["par"], order=0)
    assert v0 == val
    eng.close()


KALMAN_CONST = [r for r in GOLD if r["model"] in ("CTCRW", "OU_SSM", "BM_SSM") and r.get("X_fe") is None
                and r.get("H") is None and r.get("P0") is None]


@pytest.mark.parametrize("rec", KALMAN_CONST, ids=[r["name"] for r in KALMAN_CONST])
@pytest.mark.parametrize("flags", [capi.FLAG_FORCE_DENSE, capi.FLAG_NO_UNIFORM_DT])
def test_paths_agree(rec, flags):
    """isotropic register path == dense path == non-hoisted path on the same inputs"""
    pb = problem_from_spec(rec, flags=flags)
    eng = capi.Engine(pb)
    info = eng.info()
    assert info["path"] == (2 if flags == capi.FLAG_FORCE_DENSE else 1)
    val, grad = eng.eval(rec["par"], order=1)
    _close(val, grad, rec["expected"]["value"], rec["expected"]["grad"])
    eng.close()


@pytest.mark.parametrize("split", ["fused", "split", "1,2,4,8"])
def test_direction_split_variants(split, monkeypatch):
    monkeypatch.setenv("SSDE_ISO_SPLIT", split)
    for name in ("CTCRW_d2_const", "OU_SSM_d2_const", "BM_SSM_d1_const", "CTCRW_d1_const_regular_fixmu"):
        rec = next(r for r in GOLD if r["name"] == name)
        eng = capi.Engine(problem_from_spec(rec))
        val, grad = eng.eval(rec["par"], order=1)
        _close(val, grad, rec["expected"]["value"], rec["expected"]["grad"])
        eng.close()


@pytest.mark.parametrize("split", ["split", "1,2,4,8", "3,12"])
@pytest.mark.parametrize("model,par", [
    ("CTCRW", [-1.0, 0.1, -0.1, 0.5, 0.2]), ("OU_SSM", [-1.0, 0.3, -0.2, 0.6, 0.1]), ("BM_SSM", [-1.0, 0.05, 0.0, 0.2])])
@pytest.mark.parametrize("grid", ["missing", "irregular"])
def test_direction_parts_on_windowed_batches_vs_oracle(split, model, par, grid, monkeypatch):
    """iso_kernel (SSDE_KERNEL_ISO_SPLIT): the gradient directions of one (group, window) split over several waves, each recomputing
    the primal -- a testing path of the general lanes (SSDE_ISO_SPLIT), here on batches long enough for several verified windows,
    with missing rows or on an irregular grid, against the oracle."""
    monkeypatch.setenv("SSDE_ISO_SPLIT", split)
    rng = np.random.default_rng(23)
    ID, times, obs = simulate(model, 70, 2500, 2, seed=31)
    if grid == "missing":
        na = rng.random(len(ID)) < 0.03
        na[::2500] = False
        obs[na] = np.nan
    else:
        times = np.cumsum(rng.uniform(0.5, 1.5, size=len(ID)))
    pb = capi.Problem(model, ID, times, obs)
    eng = capi.Engine(pb)
    val, grad = eng.eval(np.array(par))
    inf = eng.info()
    assert inf["kernel_id"] == 8 and inf["window"] > 0 and inf["window_check"] <= capi.WINDOW_TOL
    _close(val, grad, *_oracle(pb, np.array(par)))
    eng.close()


KALMAN = [r for r in GOLD if r["model"] in ("CTCRW", "OU_SSM", "BM_SSM")]


@pytest.mark.parametrize("rec", KALMAN, ids=[r["name"] for r in KALMAN])
def test_report_aest_all(rec):
    pb = problem_from_spec(rec)
    eng = capi.Engine(pb)
    aest = eng.report(rec["par"])
    exp = rec["expected"]["aest_all"]
    assert aest.shape == exp.shape
    assert np.allclose(aest, exp, rtol=1e-10, atol=1e-10, equal_nan=True)
    eng.close()


def _oracle(pb, par, **kw):
    from oracle_lib import oracle_eval
    return oracle_eval(pb, par, order=1, threads=8, **kw)


@pytest.mark.parametrize("model,par", [
    ("CTCRW", [-1.0, 0.1, -0.1, 0.5, 0.2]), ("OU_SSM", [-1.0, 0.3, -0.2, 0.6, 0.1]), ("BM_SSM", [-1.0, 0.05, 0.0, 0.2])])
@pytest.mark.parametrize("row_varying", [False, True])
def test_report_aest_all_ragged_batch(model, par, row_varying):
    """REPORT(aest_all) (nllk_ctcrw.hpp:192-194, 246, 249) on 300 ragged tracks of 2-600 rows with missing rows, on
    an irregular grid, with constant coefficients and with a covariate on par[d]: every row of the n x sdim matrix
    against the oracle (rows the template never writes stay 0 in both)."""
    rng = np.random.default_rng(13)
    lens = rng.integers(2, 600, size=300)
    lens[::17] = 1                      # one-row tracks: initialised, never stepped, still reported (their a0)
    ID = np.repeat(np.arange(300), lens).astype(float)
    n = len(ID)
    _, _, obs = simulate(model, 1, n, 2, seed=6)
    times = np.cumsum(rng.uniform(0.5, 1.5, size=n))
    first = np.r_[True, ID[1:] != ID[:-1]]
    obs[(rng.random(n) < 0.05) & ~first] = np.nan
    q = capi.n_sde_par(model, 2)
    X_fe = [None] * q
    par = list(par)
    if row_varying:
        X_fe[2] = np.column_stack([np.ones(n), np.sin(np.arange(n) * 0.02)])
        par = par[:4] + [0.3] + par[4:]
    kw = {}
    if row_varying:                     # ... and with a user-supplied a0 (one row per ID segment, R/sde.R:574)
        kw["a0"] = rng.standard_normal((300, capi.state_dim(model, 2)))
    pb = capi.Problem(model, ID, times, obs, X_fe=X_fe, **kw)
    eng = capi.Engine(pb)
    aest = eng.report(np.array(par))
    _, _, oaest = _oracle(pb, np.array(par), report=True)
    assert aest.shape == oaest.shape == (n, pb.sdim)
    scale = np.nanmax(np.abs(oaest))
    assert np.allclose(aest, oaest, rtol=1e-10, atol=1e-10 * scale, equal_nan=True)
    eng.close()


@pytest.mark.parametrize("model,par", [
    ("CTCRW", [-1.0, 0.1, -0.1, 0.5, 0.2]), ("OU_SSM", [-1.0, 0.3, -0.2, 0.6, 0.1]), ("BM_SSM", [-1.0, 0.05, 0.0, 0.2])])
def test_medium_batch_vs_oracle(model, par):
    """3000 ragged tracks (47 wavefronts, several TILE_U blocks), 5 % NA rows, irregular time grid"""
    rng = np.random.default_rng(11)
    ID, times, obs = simulate(model, 3000, 40, 2, seed=5)
    keep = np.ones(len(ID), bool)
    for m in range(3000):  # ragged: drop a random tail of every track
        cut = rng.integers(0, 25)
        if cut:
            keep[m * 40 + 40 - cut: m * 40 + 40] = False
    ID, obs = ID[keep], obs[keep]
    times = np.cumsum(rng.uniform(0.5, 1.5, size=len(ID)))
    na = rng.random(len(ID)) < 0.05
    first = np.r_[True, ID[1:] != ID[:-1]]
    obs[na & ~first] = np.nan
    pb = capi.Problem(model, ID, times, obs)
    eng = capi.Engine(pb)
    val, grad = eng.eval(np.array(par), order=1)
    oval, ograd = _oracle(pb, np.array(par))
    _close(val, grad, oval, ograd)
    eng.close()


def test_device_resident_inputs_and_determinism():
    import torch
    ID, times, obs = simulate("CTCRW", 700, 64, 2, seed=3, backend="torch", device="cuda:0")
    pbd = capi.Problem.from_torch("CTCRW", ID, times, obs)
    pbh = capi.Problem("CTCRW", ID.cpu().numpy(), times.cpu().numpy(), obs.cpu().numpy())
    par = np.array([-0.5, 0.0, 0.1, 0.3, -0.2])
    ed, eh = capi.Engine(pbd), capi.Engine(pbh)
    vd, gd = ed.eval(par)
    vh, gh = eh.eval(par)
    assert vd == vh and np.array_equal(gd, gh)          # same tiles, same arithmetic: bitwise
    for _ in range(3):                                   # deterministic reduction order
        ed.forget()                                      # (a repeated par would be answered from the memo)
        v2, g2 = ed.eval(par)
        assert v2 == vd and np.array_equal(g2, gd)
    oval, ograd = _oracle(pbh, par)
    _close(vd, gd, oval, ograd)
    assert ed.info()["uniform_dt"] == 1
    ed.close(); eh.close()


def test_additivity_over_track_shards_full_size_tracks():
    """Size-independent property at the benchmark's track length (T = 1e4): the nllk and gradient
    of a batch equal the sums over two disjoint track shards (what the multi-GPU path relies on)."""
    import torch
    ID, times, obs = simulate("CTCRW", 256, 10_000, 2, seed=9, backend="torch", device="cuda:0")
    par = np.array([-2.0, 0.0, 0.0, 0.6, 0.1])
    n = ID.numel()
    cut = 100 * 10_000
    full = capi.Engine(capi.Problem.from_torch("CTCRW", ID, times, obs))
    a = capi.Engine(capi.Problem.from_torch("CTCRW", ID[:cut], times[:cut], obs[:cut]))
    b = capi.Engine(capi.Problem.from_torch("CTCRW", ID[cut:], times[cut:], obs[cut:]))
    vf, gf = full.eval(par)
    va, ga = a.eval(par)
    vb, gb = b.eval(par)
    assert abs(vf - (va + vb)) <= 1e-12 * abs(vf)
    assert np.max(np.abs(gf - (ga + gb))) <= 1e-11 * np.max(np.abs(gf))
    # and a 16-track slice of it against the oracle at full track length
    sl = 16 * 10_000
    pbh = capi.Problem("CTCRW", ID[:sl].cpu().numpy(), times[:sl].cpu().numpy(), obs[:sl].cpu().numpy())
    e = capi.Engine(pbh)
    v, g = e.eval(par)
    oval, ograd = _oracle(pbh, par)
    _close(v, g, oval, ograd)
    for x in (full, a, b, e):
        x.close()


def _full_size_check(model, par, fixed, na_frac, sim_kw, M=10_000, T=10_000):
    import torch
    ID, times, obs = simulate(model, M, T, 2, seed=1, backend="torch", device="cuda:0", **sim_kw)
    if na_frac > 0:
        gen = torch.Generator(device=ID.device)
        gen.manual_seed(7)
        na = torch.rand(ID.numel(), device=ID.device, generator=gen) < na_frac
        na[::T] = False
        obs[na] = float("nan")
    fixed = np.asarray(fixed, dtype=np.uint8)
    par = np.asarray(par, dtype=float)
    full = capi.Engine(capi.Problem.from_torch(model, ID, times, obs, par_fixed=fixed))
    vf, gf = full.eval(par)
    info = full.info()
    assert info["n_rows"] == M * T and info["window_check"] <= capi.WINDOW_TOL
    full.forget()                                        # (a repeated par would be answered from the memo)
    v2, g2 = full.eval(par)
    assert v2 == vf and np.array_equal(g2, gf)
    # additivity over two track shards
    cut = (37 * M // 100) * T
    va = 0.0
    ga = np.zeros_like(gf)
    for sl in (slice(0, cut), slice(cut, M * T)):
        e = capi.Engine(capi.Problem.from_torch(model, ID[sl], times[sl], obs[sl], par_fixed=fixed))
        v, g = e.eval(par)
        va += v
        ga += g
        e.close()
    assert abs(vf - va) <= 1e-12 * abs(vf), (vf, va)
    # The variance-direction entries are differences of two sums that are each ~10^3 times larger (sum dF/F against
    # sum u^2 dF/F^2): the rounding of those sums (1e-13 relative, and it depends on how many rows one accumulator
    # sees, i.e. on the window plan, which differs between the batch and its shards) shows as ~1e-10 in the entries.
    assert np.max(np.abs(gf - ga)) <= 1e-9 * np.max(np.abs(gf)), (gf, ga)
    # gradient = derivative of the value
    rng = np.random.default_rng(0)
    free = np.flatnonzero(fixed == 0)
    dirn = np.zeros_like(par)
    dirn[free] = rng.standard_normal(len(free))
    h = 1e-5
    fd = (full.eval(par + h * dirn, order=0) - full.eval(par - h * dirn, order=0)) / (2 * h)
    # what a central difference can resolve at 10^8 rows: rounding eps |f| / h ~ 1e-3 and truncation h^2 f''' / 6 ~ 1e-3
    # against entries of 10^3-10^4, i.e. six digits of the gradient
    assert abs(fd - gf @ dirn) <= 1e-6 * np.max(np.abs(gf)), (fd, gf @ dirn)
    assert np.all(gf[fixed != 0] == 0.0)
    full.close()
    # a random sample of whole tracks against the oracle
    pick = np.sort(rng.choice(M, size=24, replace=False))
    rows = torch.cat([torch.arange(k * T, (k + 1) * T, device=ID.device) for k in pick])
    pbh = capi.Problem(model, ID[rows].cpu().numpy(), times[rows].cpu().numpy(), obs[rows].cpu().numpy(), par_fixed=fixed)
    e = capi.Engine(pbh)
    v, g = e.eval(par)
    oval, ograd = _oracle(pbh, par)
    _close(v, g, oval, ograd)
    e.close()
    return info


def test_full_bench_size_properties():
    """BASELINE.json's configuration itself (10^4 CTCRW tracks x 10^4 rows, d = 2, mu fixed; 10^8 rows, built in HBM
    like bench.py does) under size-independent properties (the whole batch against the oracle: tests/test_gpu_whole_batch.py,
    ~50 s of the literal oracle on 16 threads):
      * additivity: the batch equals the sum of its two halves (disjoint track shards) in value and gradient;
      * the gradient is the derivative of the value (central difference along a random direction, 1e-7 relative);
      * repeated evaluations are bitwise identical; the window hand-over check passes;
      * a random sample of 24 whole tracks, evaluated on its own, matches the oracle at the usual tolerance."""
    info = _full_size_check("CTCRW", [np.log(0.1), 0.0, 0.0, np.log(2.0), 0.05], [0, 1, 1, 0, 0], 0.0,
                            dict(mu=0.0, tau=2.0, nu=1.0, sigma_obs=0.1))
    assert info["uniform_dt"] == 1 and info["window"] > 0


@pytest.mark.parametrize("model,par,sim_kw", [
    ("BM_SSM", [np.log(0.1), 0.1, 0.1, 0.0], dict(mu=0.1, sigma=1.0, sigma_obs=0.1)),
    ("OU_SSM", [np.log(0.1), 5.0, -5.0, np.log(2.0), 0.0], dict(mu=[5.0, -5.0], tau=2.0, kappa=1.0, sigma_obs=0.1)),
    ("CTCRW", [np.log(0.1), 0.0, 0.0, np.log(2.0), 0.0], dict(mu=0.0, tau=2.0, nu=1.0, sigma_obs=0.1))])
def test_full_size_properties_missing_rows(model, par, sim_kw):
    """The pieces of SURVEY 8(d) C5 at full size (10^4 tracks x 10^4 rows, 5 % missing rows, every parameter free):
    the general register kernel (two waves per SIMD for the scalar-covariance models) under the same properties."""
    _full_size_check(model, par, [0] * len(par), 0.05, sim_kw)


def test_full_size_properties_streamed_design_config():
    """SURVEY 8(d) C3 at full size (10^4 OU tracks x 10^4 rows, d = 1, a 9-column design block streamed for mu,
    88 B/row, 10^8 rows): additivity over two track shards, gradient = derivative of the value, determinism, and
    a random sample of 16 whole tracks against the oracle."""
    import torch
    from smoothsde_amd.synth import second_difference_penalty
    M, T, K = 10_000, 10_000, 9
    ID, times, obs = simulate("OU", M, T, 1, mu=1.0, tau=2.0, kappa=1.0, seed=2, backend="torch", device="cuda:0")
    n = ID.numel()
    x = torch.cumsum(torch.randn(n, device=ID.device, dtype=torch.float64) * 0.01, 0)
    x = (x - x.min()) / (x.max() - x.min())
    B = torch.stack([torch.cos((k + 1) * np.pi * x) for k in range(K)], dim=1)
    S = [second_difference_penalty(K)]
    par = np.concatenate([[1.0, np.log(2.0), 0.0], [0.3], 0.05 * np.sin(np.arange(K))])

    def engine(sl):
        return capi.Engine(capi.Problem.from_torch("OU", ID[sl], times[sl], obs[sl], X_re=[B[sl], None, None], S_list=S))

    full = engine(slice(0, n))
    vf, gf = full.eval(par)
    full.forget()                                        # (a repeated par would be answered from the memo)
    v2, g2 = full.eval(par)
    assert v2 == vf and np.array_equal(g2, gf)
    pen, gpen = full.penalty(par)
    cut = 4_100 * T
    va, ga = -pen, -gpen            # every engine adds the (parameter-only) penalty once: count it once
    for sl in (slice(0, cut), slice(cut, n)):
        e = engine(sl)
        v, g = e.eval(par)
        va += v
        ga = ga + g
        e.close()
    assert abs(vf - va) <= 1e-12 * abs(vf), (vf, va)
    assert np.max(np.abs(gf - ga)) <= 1e-11 * np.max(np.abs(gf)), (gf, ga)
    rng = np.random.default_rng(1)
    dirn = rng.standard_normal(len(par))
    h = 1e-5
    fd = (full.eval(par + h * dirn, order=0) - full.eval(par - h * dirn, order=0)) / (2 * h)
    assert abs(fd - gf @ dirn) <= 1e-6 * np.max(np.abs(gf)), (fd, gf @ dirn)      # (six digits: see _full_size_check)
    full.close()
    pick = np.sort(rng.choice(M, size=16, replace=False))
    rows = torch.cat([torch.arange(k * T, (k + 1) * T, device=ID.device) for k in pick])
    pbh = capi.Problem("OU", ID[rows].cpu().numpy(), times[rows].cpu().numpy(), obs[rows].cpu().numpy(),
                       X_re=[B[rows].cpu().numpy(), None, None], S_list=S)
    e = capi.Engine(pbh)
    v, g = e.eval(par)
    oval, ograd = _oracle(pbh, par)
    _close(v, g, oval, ograd)
    e.close()


def test_direct_large_vs_oracle():
    from smoothsde_amd.synth import bspline_basis, second_difference_penalty
    ID, times, obs = simulate("OU", 500, 400, 1, mu=1.0, seed=21)
    n = len(ID)
    x = (np.sin(np.arange(n) * 0.01) + 1) / 2
    B = bspline_basis(x, 9)
    pb = capi.Problem("OU", ID, times, obs, X_re=[B, None, None], S_list=[second_difference_penalty(9)])
    par = np.concatenate([[0.9, 0.5, 0.1], [0.2], 0.05 * np.sin(np.arange(9))])
    eng = capi.Engine(pb)
    val, grad = eng.eval(par)
    oval, ograd = _oracle(pb, par)
    _close(val, grad, oval, ograd)
    assert eng.info()["algo_bytes_per_row"] == 88.0
    eng.close()


def test_error_paths():
    rec = GOLD[1]
    pb = problem_from_spec(rec)
    eng = capi.Engine(pb)
    with pytest.raises(ValueError):
        eng.eval(np.zeros(pb.n_par_full + 1))
    # non-finite nllk is returned, not raised
    bad = rec["par"].copy()
    bad[0] = 800.0
    v = eng.eval(bad, order=0)
    assert not np.isfinite(v) or v > 0
    eng.close()


# ---- time windows (k_iso.hip) ------------------------------------------------------------------------
def _long_tracks(model="CTCRW", M=130, T=4000, seed=13):
    ID, times, obs = simulate(model, M, T, 2, seed=seed)
    return ID, times, obs


@pytest.mark.parametrize("model,par", [("CTCRW", [-1.5, 0.0, 0.0, 0.5, 0.0]), ("OU_SSM", [-1.0, 0.3, -0.2, 0.6, 0.1]),
                                       ("BM_SSM", [-1.0, 0.05, 0.0, 0.2])])
def test_time_windows_match_sequential_and_oracle(model, par, monkeypatch):
    ID, times, obs = _long_tracks(model)
    par = np.array(par)
    pb = capi.Problem(model, ID, times, obs)
    eng = capi.Engine(pb)
    v, g = eng.eval(par)
    info = eng.info()
    assert info["lanes_per_track"] > 1 and info["window"] > 0 and info["window_check"] <= capi.WINDOW_TOL
    monkeypatch.setenv("SSDE_CHUNKS", "1")
    seq = capi.Engine(pb)
    vs, gs = seq.eval(par)
    assert seq.info()["lanes_per_track"] == 1
    assert abs(v - vs) <= 1e-12 * abs(vs)
    assert np.max(np.abs(g - gs)) <= 1e-10 * np.max(np.abs(gs))
    oval, ograd = _oracle(pb, par)
    _close(v, g, oval, ograd)
    eng.close(); seq.close()


@pytest.mark.parametrize("model", ["CTCRW", "OU_SSM", "BM_SSM"])
def test_fused_finalising_work_is_bitwise_the_two_launch_form(model, monkeypatch):
    """SSDE_FUSED_FINALIZE=1: the hand-over checks by the second wave to arrive at a boundary and the fixed-order sums by the last wave
    of iso_shared_kernel itself, instead of the dependent iso_finalize_kernel launch (the default: the fused form measured slower,
    profiles/r05_fused_finalize_ab.txt) -- value, gradient and check value must be the same bits, evaluation after evaluation."""
    ID, times, obs = simulate(model, 700, 3000, 2, seed=17)
    pb = capi.Problem(model, ID, times, obs, par_fixed=[0, 1, 1, 0, 0][:1 + capi.n_sde_par(model, 2)])
    q = capi.n_sde_par(model, 2)
    pars = [np.r_[-1.0, 0.0, 0.0, 0.4, 0.1][:1 + q] + 0.01 * np.sin(k + np.arange(1 + q)) for k in range(4)]
    two = capi.Engine(pb)
    monkeypatch.setenv("SSDE_FUSED_FINALIZE", "1")
    fused = capi.Engine(pb)
    for p in pars + pars[:1]:
        v2, g2 = two.eval(p)
        vf, gf = fused.eval(p)
        assert two.info()["kernel_id"] == 3 and fused.info()["kernel_id"] == 3 and two.info()["lanes_per_track"] > 1
        assert vf == v2 and np.array_equal(gf, g2)
        assert fused.info()["window_check"] == two.info()["window_check"] <= capi.WINDOW_TOL
        two.forget(); fused.forget()
    _close(vf, gf, *_oracle(pb, pars[0]))
    two.close(); fused.close()


def test_short_overlap_is_detected_and_repaired(monkeypatch):
    """A deliberately useless warm-up (4 rows) must fail the hand-over check and be repaired by
    ssde_eval (longer overlap, finally one sequential window) -- never returned silently."""
    ID, times, obs = _long_tracks()
    par = np.array([0.5, 0.0, 0.0, 0.5, 0.0])      # sigma_obs = 1.6: slow forgetting
    pb = capi.Problem("CTCRW", ID, times, obs)
    monkeypatch.setenv("SSDE_WINDOW", "4")
    eng = capi.Engine(pb)
    v, g = eng.eval(par)
    info = eng.info()
    assert info["window_retries"] >= 1
    assert info["window_check"] <= capi.WINDOW_TOL or info["lanes_per_track"] == 1
    oval, ograd = _oracle(pb, par)
    _close(v, g, oval, ograd)
    eng.close()


@pytest.mark.parametrize("row_varying", [False, True])
def test_widened_plan_recovers(row_varying):
    """A warm-up widened after a failed hand-over check (here: through ssde_widen_windows) is not for life: after 32
    evaluations accepted at the first try it is halved on probation, and again, until the planned length is back --
    with every evaluation still checked.  Values do not move by more than rounding."""
    from smoothsde_amd.synth import bspline_basis, second_difference_penalty
    if row_varying:
        ID, times, obs = simulate("CTCRW", 1, 6000, 2, tau=1.0, nu=1.0, sigma_obs=0.05, seed=11)
        B = bspline_basis((np.sin(np.arange(6000) * 0.01) + 1) / 2, 6)
        pb = capi.Problem("CTCRW", ID, times, obs, X_re=[None, None, B, None], S_list=[second_difference_penalty(6)])
        par = np.zeros(pb.n_par_full)
        par[0] = np.log(0.05)
    else:
        ID, times, obs = simulate("CTCRW", 70, 6000, 2, tau=2.0, nu=1.0, sigma_obs=0.1, seed=11)
        pb = capi.Problem("CTCRW", ID, times, obs)
        par = np.array([np.log(0.1), 0.0, 0.0, np.log(2.0), 0.0])
    eng = capi.Engine(pb)
    v0, g0 = eng.eval(par)
    eng.forget()
    eng.eval(par)
    w0 = eng.info()["window"]
    assert w0 > 0
    eng.widen_windows(4)
    eng.eval(par)
    assert eng.info()["window"] >= 3 * w0
    seen = set()
    for k in range(80):
        eng.forget()                          # the policy counts evaluations that ran, not answers from the memo
        v, g = eng.eval(par)
        seen.add(eng.info()["window"])
        assert abs(v - v0) <= 1e-12 * abs(v0) and np.max(np.abs(g - g0)) <= 1e-9 * np.max(np.abs(g0))
    assert eng.info()["window"] == w0, sorted(seen)
    assert eng.info()["window_check"] <= capi.WINDOW_TOL
    eng.close()


def test_no_forgetting_falls_back_to_sequential():
    ID, times, obs = _long_tracks(M=70, T=2000)
    par = np.array([6.0, 0.0, 0.0, 0.5, 0.0])      # sigma_obs = 400: the filter barely updates
    pb = capi.Problem("CTCRW", ID, times, obs)
    eng = capi.Engine(pb)
    v, g = eng.eval(par)
    oval, ograd = _oracle(pb, par)
    _close(v, g, oval, ograd)
    eng.close()


def test_windows_with_missing_rows_and_irregular_times():
    rng = np.random.default_rng(5)
    ID, times, obs = _long_tracks(M=100, T=3000)
    times = np.cumsum(rng.uniform(0.2, 1.8, size=len(ID)))
    first = np.r_[True, ID[1:] != ID[:-1]]
    na = (rng.random(len(ID)) < 0.2) & ~first
    obs[na] = np.nan
    # a long gap of missing rows in every track: no forgetting across it
    for m in range(100):
        obs[m * 3000 + 1400: m * 3000 + 1460] = np.nan
    par = np.array([-1.0, 0.02, -0.01, 0.4, 0.1])
    pb = capi.Problem("CTCRW", ID, times, obs)
    eng = capi.Engine(pb)
    v, g = eng.eval(par)
    assert eng.info()["window_check"] <= capi.WINDOW_TOL or eng.info()["lanes_per_track"] == 1
    oval, ograd = _oracle(pb, par)
    _close(v, g, oval, ograd)
    eng.close()


# ---- shared-covariance path (regular grid: covariance half evaluated once per evaluation) ---------------
@pytest.mark.parametrize("model,par", [("CTCRW", [-1.2, 0.03, -0.02, 0.5, 0.1]), ("OU_SSM", [-1.0, 0.3, -0.2, 0.6, 0.1]),
                                       ("BM_SSM", [-1.0, 0.05, 0.0, 0.2])])
@pytest.mark.parametrize("d", [1, 2])
def test_shared_covariance_path_mixed_groups(model, par, d, monkeypatch):
    """Regular grid; 300 ragged tracks of which the first 100 contain missing rows: their wavefront
    groups take the general kernel, the NaN-free groups the shared-gain kernel."""
    rng = np.random.default_rng(3)
    ID, times, obs = simulate(model, 300, 900, d, seed=31)
    keep = np.ones(len(ID), bool)
    for m in range(300):
        cut = rng.integers(0, 400)
        if cut:
            keep[m * 900 + 900 - cut: m * 900 + 900] = False
    ID, obs = ID[keep], obs[keep]
    times = np.arange(1, len(ID) + 1, dtype=float) * 0.5
    first = np.r_[True, ID[1:] != ID[:-1]]
    na = (rng.random(len(ID)) < 0.03) & ~first & (ID < 100)
    obs[na] = np.nan
    par = np.array(par)
    if d == 1:
        par = np.delete(par, 2)
    pb = capi.Problem(model, ID, times, obs)
    eng = capi.Engine(pb)
    assert eng.info()["uniform_dt"] == 1
    v, g = eng.eval(par)
    oval, ograd = _oracle(pb, par)
    _close(v, g, oval, ograd)
    monkeypatch.setenv("SSDE_NO_SHARED", "1")
    gen = capi.Engine(pb)
    v2, g2 = gen.eval(par)
    assert eng.info()["kernel_id"] in (3, 7) and gen.info()["kernel_id"] in (5, 6)           # iso_shared_kernel (alone or beside a general kernel) / the general lanes
    assert abs(v - v2) <= 1e-12 * abs(v2)
    assert np.max(np.abs(g - g2)) <= 1e-10 * np.max(np.abs(g2))
    eng.close(); gen.close()


DIRECT = [r for r in GOLD if r["model"] in ("BM", "OU", "BM_t")]


@pytest.mark.parametrize("rec", DIRECT, ids=[r["name"] for r in DIRECT])
def test_generic_direct_kernel(rec, monkeypatch):
    """the runtime-routed direct kernel (the fallback for three or more streamed parameters) on every
    direct-family golden case, incl. BM_t (Student-t increments, tr_dens.hpp:38-44)"""
    monkeypatch.setenv("SSDE_NO_DIRECT_FAST", "1")
    eng = capi.Engine(problem_from_spec(rec))
    val, grad = eng.eval(rec["par"], order=1)
    assert eng.info()["kernel_id"] == 1                            # direct_kernel
    _close(val, grad, rec["expected"]["value"], rec["expected"]["grad"])
    eng.close()


def test_bm_t_medium_batch_vs_oracle():
    """2000 ragged BM_t tracks with NA rows and a smooth on log sigma, df = 4"""
    from smoothsde_amd.synth import bspline_basis, second_difference_penalty
    rng = np.random.default_rng(5)
    ID, times, obs = simulate("BM", 2000, 30, 1, mu=0.2, sigma=0.7, seed=9)
    obs[rng.random(len(ID)) < 0.03] = np.nan
    n = len(ID)
    x = (np.sin(np.arange(n) * 0.01) + 1) / 2
    pb = capi.Problem("BM_t", ID, times, obs, X_re=[None, bspline_basis(x, 6)], S_list=[second_difference_penalty(6)],
                      other_data=4.0)
    par = np.r_[0.15, -0.3, 0.2, 0.1 * np.sin(np.arange(6))]
    eng = capi.Engine(pb)
    val, grad = eng.eval(par)
    oval, ograd = _oracle(pb, par)
    _close(val, grad, oval, ograd)
    eng.close()
    with pytest.raises(ValueError):
        capi.Problem("BM_t", ID, times, obs)                 # no degrees of freedom


def test_more_groups_than_simds_split_transient_and_stationary(monkeypatch):
    """33 000 tracks = 516 wavefront groups: no time windows are needed to fill the chip, but every track is still
    split into the covariance transient and ONE stationary window so that the bulk runs the lean stationary
    kernel; against the plain sequential filter (SSDE_CHUNKS=1), which the other tests pin to the oracle"""
    import torch
    ID, times, obs = simulate("CTCRW", 33000, 400, 2, tau=2.0, nu=1.0, sigma_obs=0.1, seed=6, backend="torch", device="cuda:0")
    pb = capi.Problem.from_torch("CTCRW", ID, times, obs, par_fixed=[0, 1, 1, 0, 0])
    par = np.array([np.log(0.1), 0.0, 0.0, np.log(2.0), 0.05])
    eng = capi.Engine(pb)
    v, g = eng.eval(par)
    info = eng.info()
    assert info["lanes_per_track"] == 2 and info["window"] > 0 and info["window_check"] <= capi.WINDOW_TOL, info
    monkeypatch.setenv("SSDE_CHUNKS", "1")
    seq = capi.Engine(pb)
    vs, gs = seq.eval(par)
    assert seq.info()["lanes_per_track"] == 1
    assert abs(v - vs) <= 1e-12 * abs(vs)
    assert np.max(np.abs(g - gs)) <= 1e-10 * np.max(np.abs(gs))
    eng.close(); seq.close()
    del ID, times, obs
    torch.cuda.empty_cache()


def test_decaying_columns_medium_batch_vs_oracle():
    """decaying response model (nllk_sde.hpp:47-58): 500 ragged OU tracks, a 6-column smooth on mu whose columns decay
    with t_decay at two different rates, plus a non-decaying slope on log tau"""
    from smoothsde_amd.synth import bspline_basis, second_difference_penalty
    rng = np.random.default_rng(15)
    ID, times, obs = simulate("OU", 500, 40, 1, mu=1.0, tau=2.0, kappa=1.0, seed=19)
    n = len(ID)
    x = (np.sin(np.arange(n) * 0.02) + 1) / 2
    X_fe = [None, np.column_stack([np.ones(n), x]), None]
    pb = capi.Problem("OU", ID, times, obs, X_fe=X_fe, X_re=[bspline_basis(x, 6), None, None],
                      S_list=[second_difference_penalty(6)], t_decay=rng.uniform(0, 3, size=3 * n),
                      col_decay=[0, 1, 2, 3, 5], ind_decay=[0, 0, 0, 1, 1])
    assert pb.n_decay == 2 and pb.n_par_full == 4 + 1 + 2 + 6
    par = np.r_[1.0, 0.6, -0.2, 0.1, 0.3, -0.4, 0.2, 0.4 * np.sin(np.arange(6))]
    eng = capi.Engine(pb)
    val, grad = eng.eval(par)
    oval, ograd = _oracle(pb, par)
    _close(val, grad, oval, ograd)
    assert grad[pb.off_decay] != 0.0 and grad[pb.off_decay + 1] != 0.0
    eng.close()
    with pytest.raises(ValueError):
        capi.Problem("OU_SSM", ID, times, obs, X_re=[bspline_basis(x, 6), None, None], S_list=[second_difference_penalty(6)],
                     t_decay=np.zeros(3 * n), col_decay=[0], ind_decay=[0])


@pytest.mark.parametrize("ls,ln,expect_windows", [(1.0, -2.0, True), (1.5, -2.0, True), (2.0, -2.0, False)])
def test_stationary_lanes_with_slow_forgetting(ls, ln, expect_windows):
    """closed-loop spectral radius 0.92 / 0.95 (long warm-ups, transfer-function lanes close to their conditioning
    limit) and 0.97 (beyond it: the engine must stay on the sequential filter) -- all against the oracle"""
    ID, times, obs = simulate("CTCRW", 64, 12000, 2, tau=2.0, nu=np.exp(ln), sigma_obs=np.exp(ls), seed=31)
    pb = capi.Problem("CTCRW", ID, times, obs, par_fixed=[0, 1, 1, 0, 0])
    par = np.array([ls, 0.0, 0.0, np.log(2.0), ln])
    eng = capi.Engine(pb)
    v, g = eng.eval(par)
    info = eng.info()
    assert (info["window"] > 0) == expect_windows, info
    assert info["window_check"] <= capi.WINDOW_TOL
    oval, ograd = _oracle(pb, par)
    _close(v, g, oval, ograd)
    eng.close()


def test_cir_medium_batch_vs_oracle():
    """Cox-Ingersoll-Ross (tr_dens.hpp:53-67): 1500 ragged positive tracks, smooth on log mu, NA entries; the
    Bessel arguments range over several orders of magnitude (sigma small on part of the covariate range)"""
    from smoothsde_amd.synth import bspline_basis, second_difference_penalty
    rng = np.random.default_rng(25)
    lens = rng.integers(2, 40, size=1500)
    ID = np.repeat(np.arange(1500), lens).astype(float)
    n = len(ID)
    times = np.cumsum(rng.uniform(0.3, 1.5, size=n))
    obs = np.exp(0.3 * np.cumsum(rng.standard_normal((n, 1)) * 0.4, axis=0) % 2.0)
    obs[rng.random(n) < 0.03] = np.nan
    x = (np.sin(np.arange(n) * 0.01) + 1) / 2
    X_fe = [None, None, np.column_stack([np.ones(n), x])]
    pb = capi.Problem("CIR", ID, times, obs, X_fe=X_fe, X_re=[bspline_basis(x, 5), None, None],
                      S_list=[second_difference_penalty(5)])
    par = np.r_[0.2, -0.5, -0.4, -1.6, 0.3, 0.2 * np.sin(np.arange(5))]      # sigma from 0.67 down to 0.14
    eng = capi.Engine(pb)
    val, grad = eng.eval(par)
    oval, ograd = _oracle(pb, par)
    _close(val, grad, oval, ograd)
    eng.close()


def test_cir_weak_diffusion_large_bessel_arguments():
    """sigma = 0.03-0.05: Bessel arguments of 10^3-10^5 and orders of 10^3-10^4 (an unscaled besselI overflows at
    700); generic and fast direct kernels against the oracle, and a finite result where a capped series would not be"""
    rng = np.random.default_rng(78)
    ID, times, _ = simulate("BM", 300, 60, 1, seed=78)
    n = len(ID)
    obs = np.exp(1.0 + 0.04 * np.cumsum(rng.standard_normal((n, 1)), axis=0) % 1.5)
    for X_fe, par in (([None, None, None], np.array([1.0, -0.3, np.log(0.05)])),
                      ([None, None, np.column_stack([np.ones(n), np.linspace(0, 1, n)])],
                       np.array([1.2, 0.1, np.log(0.05), np.log(0.6)]))):
        pb = capi.Problem("CIR", ID, times, obs, X_fe=X_fe)
        eng = capi.Engine(pb)
        val, grad = eng.eval(par)
        oval, ograd = _oracle(pb, par)
        assert np.isfinite(oval)
        _close(val, grad, oval, ograd)
        eng.close()


# ---- random-effect blocks given as piecewise-cubic functions of a covariate (ssde_ppbasis) ---------------------------
def _pp_problem(model, seed, fe_slope=False, on_second=False, **kw):
    from smoothsde_amd.synth import bspline_ppbasis, second_difference_penalty
    rng = np.random.default_rng(seed)
    d = 1 if model in ("BM_t",) else 2
    sim = {"OU": "OU", "BM": "BM", "BM_t": "BM", "CTCRW": "CTCRW"}[model]
    ID, times, obs = simulate(sim, 40, 120, d, mu=1.0 if model == "OU" else 0.1, seed=seed)
    n = len(ID)
    x = np.clip(0.5 + 0.45 * np.sin(np.arange(n) * 0.013) + 0.03 * rng.standard_normal(n), 0, 1)
    q = capi.n_sde_par(model, d)
    basis = [None] * q
    basis[0] = bspline_ppbasis(x, 7)
    S = [second_difference_penalty(7)]
    if on_second:
        basis[d] = bspline_ppbasis(np.clip(x ** 2, 0, 1), 5)
        S.append(second_difference_penalty(5))
    X_fe = None
    if fe_slope:
        X_fe = [None] * q
        X_fe[0] = np.column_stack([np.ones(n), x])          # streamed FE column on the same parameter: no fast route
    pb = capi.Problem(model, ID, times, obs, X_fe=X_fe, S_list=S, basis_re=basis, **kw)
    dense = capi.Problem(model, ID, times, obs, X_fe=X_fe, X_re=[None if b is None else b.dense() for b in basis], S_list=S, **kw)
    par = 0.15 * rng.standard_normal(pb.n_par_full)
    if model == "OU":
        par[pb.off_fe:pb.off_fe + d] += 1.0
    if pb.lead_names:
        par[0] = -1.0
    return pb, dense, par


@pytest.mark.parametrize("model,kw", [("OU", {}), ("BM", {}), ("BM_t", {"other_data": 4.0}), ("OU", {"on_second": True}),
                                      ("OU", {"fe_slope": True}), ("CTCRW", {})])
def test_basis_tables_match_streamed_blocks_and_oracle(model, kw):
    """the block evaluated on the fly from the table (fast direct kernel), materialised from it (generic / Kalman
    paths) and streamed as a dense matrix must agree with each other and with the oracle"""
    extra = {k: v for k, v in kw.items() if k == "other_data"}
    flags = {k: v for k, v in kw.items() if k != "other_data"}
    pb, dense, par = _pp_problem(model, 41, **flags, **extra)
    e1, e2 = capi.Engine(pb), capi.Engine(dense)
    v1, g1 = e1.eval(par)
    v2, g2 = e2.eval(par)
    oval, ograd = _oracle(dense, par)
    _close(v1, g1, oval, ograd)
    _close(v2, g2, oval, ograd)
    assert abs(v1 - v2) <= 1e-12 * abs(v2)
    if model != "CTCRW" and not flags.get("fe_slope"):
        # on-the-fly evaluation: the design block is not resident (only x, the table and the data are)
        assert e1.info()["hbm_bytes"] < e2.info()["hbm_bytes"] - 0.8 * 7 * pb.n * 8
    assert e1.info()["algo_bytes_per_row"] == e2.info()["algo_bytes_per_row"]
    e1.close(); e2.close()


def test_basis_table_next_to_a_block_too_wide_for_the_fast_kernel():
    """mu as a basis table while tau carries 25 streamed smooth columns (more than the fast direct kernel
    holds per parameter): the table must be materialised for the generic kernel, never scored as an intercept."""
    from smoothsde_amd.synth import bspline_basis, bspline_ppbasis, second_difference_penalty
    rng = np.random.default_rng(5)
    ID, times, obs = simulate("OU", 30, 150, 1, mu=1.0, seed=5)
    n = len(ID)
    x = np.clip(0.5 + 0.45 * np.sin(np.arange(n) * 0.011) + 0.03 * rng.standard_normal(n), 0, 1)
    basis = [bspline_ppbasis(x, 7), None, None]
    wide = bspline_basis(np.clip(x ** 1.5, 0, 1), 25)
    S = [second_difference_penalty(7), second_difference_penalty(25)]
    pb = capi.Problem("OU", ID, times, obs, X_re=[None, wide, None], S_list=S, basis_re=basis)
    dense = capi.Problem("OU", ID, times, obs, X_re=[basis[0].dense(), wide, None], S_list=S)
    par = 0.05 * rng.standard_normal(pb.n_par_full)
    par[0] = 1.0
    e1, e2 = capi.Engine(pb), capi.Engine(dense)
    v1, g1 = e1.eval(par)
    v2, g2 = e2.eval(par)
    oval, ograd = _oracle(dense, par)
    _close(v1, g1, oval, ograd)
    _close(v2, g2, oval, ograd)
    e1.close(); e2.close()
