"""CPU suite: the per-lane arithmetic the HIP kernels inline (csrc/ssde_math.hpp), built for the
host by tests/hostsim, against the oracle.  Tolerances: value 1e-11 rel, gradient
1e-9 * max|g| + 1e-11 (the engine's stated bar against the reference is 1e-8)."""
import numpy as np
import pytest

from cases import problem_from_spec
from golden_io import load_golden
from hostsim_lib import kalman_adj, kalman_adj_full, kalman_iso, kalman_tv, load
from oracle_lib import oracle_eval

GOLD = {r["name"]: r for r in load_golden()}
ISO = [n for n in GOLD if n.endswith("_const") or n.endswith("_const_regular_fixmu") or n == "elephant6_ctcrw"]
ISO = [n for n in ISO if GOLD[n]["model"] in ("CTCRW", "OU_SSM", "BM_SSM") and GOLD[n]["n_dim"] <= 2]   # lanes hold d <= 2


@pytest.mark.parametrize("name", ISO)
def test_iso_lane_math_matches_golden(name):
    rec = GOLD[name]
    pb = problem_from_spec(rec)
    val, grad = kalman_iso(pb, rec["par"], mask=15)
    exp = rec["expected"]
    assert abs(val - exp["value"]) <= 1e-11 * max(1.0, abs(exp["value"]))
    g = grad.copy()
    g[pb.par_fixed != 0] = 0.0
    assert np.max(np.abs(g - exp["grad"])) <= 1e-9 * np.max(np.abs(exp["grad"])) + 1e-11


TV = [n for n in GOLD if GOLD[n]["model"] in ("CTCRW", "OU_SSM", "BM_SSM") and GOLD[n].get("H") is None
      and GOLD[n].get("P0") is None and GOLD[n]["n_dim"] <= 2]


@pytest.mark.parametrize("name", TV)
def test_tv_lane_math_matches_oracle(name):
    """csrc/ssde_tv.hpp (records + one generic tangent per lane) against the oracle's data term, on every
    golden Kalman case without H_array / custom P0 (constant-coefficient cases are the special case of
    intercept-only directions)."""
    rec = GOLD[name]
    pb = problem_from_spec(rec)
    par = np.asarray(rec["par"], dtype=np.float64)
    val, grad = kalman_tv(pb, par)
    oval, ograd = oracle_eval(pb, par, order=1, data_only=True)
    grad[pb.par_fixed != 0] = 0.0
    assert abs(val - oval) <= 1e-11 * max(1.0, abs(oval))
    assert np.max(np.abs(grad - ograd)) <= 1e-9 * np.max(np.abs(ograd)) + 1e-11


@pytest.mark.parametrize("name", TV)
def test_adjoint_lane_math_matches_oracle(name):
    """csrc/ssde_adj.hpp (the reverse sweep k_iso_adj.hip runs lane = track: a record per row forwards, the transposed step
    backwards, the coefficient gradient as X' g) against the oracle's data term and against the forward tangents, on every
    golden Kalman case without H_array / custom P0."""
    rec = GOLD[name]
    pb = problem_from_spec(rec)
    par = np.asarray(rec["par"], dtype=np.float64)
    val, grad = kalman_adj(pb, par)
    oval, ograd = oracle_eval(pb, par, order=1, data_only=True)
    grad[pb.par_fixed != 0] = 0.0
    assert abs(val - oval) <= 1e-11 * max(1.0, abs(oval))
    assert np.max(np.abs(grad - ograd)) <= 1e-9 * np.max(np.abs(ograd)) + 1e-11
    tval, tgrad = kalman_tv(pb, par)
    tgrad[pb.par_fixed != 0] = 0.0
    assert abs(val - tval) <= 1e-12 * max(1.0, abs(tval))
    assert np.max(np.abs(grad - tgrad)) <= 1e-10 * np.max(np.abs(tgrad)) + 1e-12


FULL = [n for n in GOLD if GOLD[n]["model"] in ("CTCRW", "OU_SSM", "BM_SSM") and GOLD[n]["n_dim"] == 2]


@pytest.mark.parametrize("name", FULL)
def test_adjoint_full_covariance_lane_math_matches_oracle(name):
    """csrc/ssde_adj.hpp, AdjFull: the reverse sweep with a full covariance -- per-row H_array (nllk_ctcrw.hpp:203-205), a P0 that is
    not block-identical, and (the special case) sigma_obs^2 I with the default P0 -- on every golden Kalman case with two response columns."""
    rec = GOLD[name]
    pb = problem_from_spec(rec)
    par = np.asarray(rec["par"], dtype=np.float64)
    val, grad = kalman_adj_full(pb, par)
    oval, ograd = oracle_eval(pb, par, order=1, data_only=True)
    grad[pb.par_fixed != 0] = 0.0
    if pb.H is not None:
        grad[0] = 0.0
    assert abs(val - oval) <= 1e-11 * max(1.0, abs(oval)), (val, oval)
    assert np.max(np.abs(grad - ograd)) <= 1e-9 * np.max(np.abs(ograd)) + 1e-11, (grad, ograd)


@pytest.mark.parametrize("mask", [0, 1, 2, 4, 8, 5, 10])
def test_direction_masks_are_consistent(mask):
    rec = GOLD["CTCRW_d2_const"]
    pb = problem_from_spec(rec)
    vfull, gfull = kalman_iso(pb, rec["par"], mask=15)
    v, g = kalman_iso(pb, rec["par"], mask=mask)
    assert v == vfull
    want = np.zeros_like(gfull)
    if mask & 1: want[0] = gfull[0]
    if mask & 2: want[1:3] = gfull[1:3]
    if mask & 4: want[3] = gfull[3]
    if mask & 8: want[4] = gfull[4]
    assert np.array_equal(g, want)


def test_direct_transition_gradients():
    import ctypes as C
    lib = load()
    dp = C.POINTER(C.c_double)
    rng = np.random.default_rng(7)

    def f(model, z0, z1, dt, m, a, b, g=None):
        g = np.zeros(3) if g is None else g
        return lib.hostsim_direct(model, z0, z1, dt, m, a, b, g.ctypes.data_as(dp))

    for model in (0, 1):
        for _ in range(20):
            z0, z1, dt = rng.normal(), rng.normal(), rng.uniform(0.2, 3)
            mu, p1, p2 = rng.normal(), rng.uniform(-1, 1), rng.uniform(-1, 1)
            g = np.zeros(3)
            f(model, z0, z1, dt, mu, p1, p2, g)
            h = 1e-6
            fd = [(f(model, z0, z1, dt, mu + h, p1, p2) - f(model, z0, z1, dt, mu - h, p1, p2)) / (2 * h),
                  (f(model, z0, z1, dt, mu, p1 + h, p2) - f(model, z0, z1, dt, mu, p1 - h, p2)) / (2 * h),
                  (f(model, z0, z1, dt, mu, p1, p2 + h) - f(model, z0, z1, dt, mu, p1, p2 - h)) / (2 * h)]
            n = 2 if model == 0 else 3
            assert np.allclose(g[:n], fd[:n], rtol=1e-6, atol=1e-7)


def test_log_bessel_i_matches_mpmath():
    """log I_q(x) and its derivatives in x and in the order q (ssde_math.hpp, the function the CIR kernels call),
    compiled for the host, against mpmath at 40 digits -- from the small arguments where the series starts at k = 0
    up to arguments far beyond the overflow of an unscaled besselI (x ~ 700) and the orders q ~ 1/sigma^2 of a
    weakly diffusive CIR process.  Tolerances: 2e-14 relative on the value, 1e-11 relative on the derivatives."""
    import ctypes as C
    mp = pytest.importorskip("mpmath")
    lib = load()
    dp = C.POINTER(C.c_double)
    lib.hostsim_log_bessel_i.argtypes = [C.c_double, C.c_double, dp]
    lib.hostsim_log_bessel_i.restype = None
    mp.mp.dps = 40
    grid = [(0.01, -0.5), (0.3, 0.0), (2.5, 0.7), (9.0, 3.2), (35.0, -0.9), (80.0, 12.0), (700.0, 1.5), (3830.0, 2000.0),
            (3830.0, 0.25), (2.0e4, 150.0), (1.0e5, 3.0), (1.0e5, 4.0e4), (50.0, 5000.0), (1.0e6, 10.0)]
    for x, q in grid:
        out = np.zeros(3)
        lib.hostsim_log_bessel_i(x, q, out.ctypes.data_as(dp))
        f = lambda a, b: mp.log(mp.besseli(b, a, maxterms=10 ** 7))
        xf, qf = mp.mpf(x), mp.mpf(q)
        val = f(xf, qf)
        dx = mp.diff(lambda a: f(a, qf), xf)
        dq = mp.diff(lambda b: f(xf, b), qf)
        assert abs(out[0] - float(val)) <= 2e-14 * max(1.0, abs(float(val))), (x, q, out[0], float(val))
        assert abs(out[1] - float(dx)) <= 1e-11 * max(1.0, abs(float(dx))), (x, q, out[1], float(dx))
        assert abs(out[2] - float(dq)) <= 1e-11 * max(1.0, abs(float(dq))), (x, q, out[2], float(dq))


def test_log_bessel_i_second_derivatives_match_mpmath():
    """log_bessel_i2 (ssde_math.hpp): the five derivatives the exact CIR Hessian needs -- moments of k and psi(k + q + 1) under the
    series' weights -- against mpmath at 40 digits.  l_xx is a difference of two O(1/x) terms that leaves O(1/x^2): its relative
    tolerance grows with x (1e-10 * max(1, x / 100))."""
    import ctypes as C
    mp = pytest.importorskip("mpmath")
    lib = load()
    dp = C.POINTER(C.c_double)
    lib.hostsim_log_bessel_i2.argtypes = [C.c_double, C.c_double, dp]
    lib.hostsim_log_bessel_i2.restype = None
    lib.hostsim_log_bessel_i.argtypes = [C.c_double, C.c_double, dp]
    lib.hostsim_log_bessel_i.restype = None
    mp.mp.dps = 40
    f = lambda a, b: mp.log(mp.besseli(b, a, maxterms=10 ** 7))
    for x, q in [(0.01, -0.5), (0.3, 0.0), (2.5, 0.7), (9.0, 3.2), (35.0, -0.7), (80.0, 12.0), (700.0, 1.5), (3830.0, 2000.0),
                 (3830.0, 0.25), (2.0e4, 150.0), (50.0, 5000.0)]:
        out, o1 = np.zeros(6), np.zeros(3)
        lib.hostsim_log_bessel_i2(x, q, out.ctypes.data_as(dp))
        lib.hostsim_log_bessel_i(x, q, o1.ctypes.data_as(dp))
        assert abs(out[0] - o1[0]) <= 4e-15 * max(1.0, abs(o1[0])) and np.allclose(out[1:3], o1[1:3], rtol=1e-12, atol=1e-13)
        xf, qf = mp.mpf(x), mp.mpf(q)
        for k, (nx, nq) in enumerate([(2, 0), (1, 1), (0, 2)]):
            ref = float(mp.diff(f, (xf, qf), (nx, nq)))
            tol = 1e-10 * max(1.0, x / 100.0) if k == 0 else 1e-10
            assert abs(out[3 + k] - ref) <= tol * max(abs(ref), 1e-12 if k else 0.0) + 1e-300, (x, q, k, out[3 + k], ref)

