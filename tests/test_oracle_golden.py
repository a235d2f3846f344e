"""CPU suite: the oracle against the committed golden vectors and against the independent
restatement (dense joint Gaussian / torch.distributions + autograd)."""
import numpy as np
import pytest

from cases import all_specs, problem_from_spec
from golden_io import load_golden
from oracle_lib import oracle_eval

GOLD = load_golden()
VALUE_RTOL = 1e-12   # oracle is deterministic: re-evaluation must reproduce the fixture
GRAD_TOL = lambda g: 1e-10 * np.max(np.abs(g)) + 1e-12  # noqa: E731


@pytest.mark.parametrize("rec", GOLD, ids=[r["name"] for r in GOLD])
def test_oracle_reproduces_golden(rec):
    pb = problem_from_spec(rec)
    val, grad = oracle_eval(pb, rec["par"], order=1)
    exp = rec["expected"]
    assert abs(val - exp["value"]) <= VALUE_RTOL * max(1.0, abs(exp["value"]))
    assert np.max(np.abs(grad - exp["grad"])) <= GRAD_TOL(exp["grad"])
    # independent restatement stored next to it (value 1e-10 rel, gradient 1e-8 rel + 1e-10)
    assert abs(val - exp["indep_value"]) <= 1e-10 * max(1.0, abs(exp["indep_value"]))
    assert np.max(np.abs(grad - exp["indep_grad"])) <= 1e-8 * np.max(np.abs(exp["indep_grad"])) + 1e-10


def test_golden_inputs_match_generator():
    """The fixture inputs are exactly what tests/cases.py generates (bitwise, NA payloads included)."""
    specs = {s["name"]: s for s in all_specs()}
    assert set(specs) == {r["name"] for r in GOLD}
    for rec in GOLD:
        s = specs[rec["name"]]
        for key in ("ID", "times", "obs", "par"):
            a, b = np.asarray(s[key], dtype=np.float64), np.asarray(rec[key], dtype=np.float64)
            assert a.shape == b.shape and np.array_equal(a.view(np.uint64), b.view(np.uint64)), (rec["name"], key)


@pytest.mark.parametrize("name", ["CTCRW_d2_const", "OU_SSM_d1_tv", "BM_SSM_d2_H", "OU_d2_tv", "CTCRW_d2_tv2_RNA"])
def test_independent_restatement_live(name):
    from refimpl import ref_eval
    rec = next(r for r in GOLD if r["name"] == name)
    pb = problem_from_spec(rec)
    val, grad = oracle_eval(pb, rec["par"], order=1)
    rval, rgrad = ref_eval(pb, rec["par"])
    assert abs(val - rval) <= 1e-10 * max(1.0, abs(rval))
    assert np.max(np.abs(grad - rgrad)) <= 1e-8 * np.max(np.abs(rgrad)) + 1e-10


@pytest.mark.parametrize("model,d,variant", [("CTCRW", 3, "const"), ("OU_SSM", 4, "tv"), ("CTCRW", 5, "const"), ("BM_SSM", 6, "tv"),
                                             ("OU_SSM", 7, "const"), ("CTCRW", 8, "const")])
def test_wide_coupled_responses_against_the_independent_restatement(model, d, variant):
    """three to eight response columns with a full per-row measurement covariance AND a full P0 (F a full d x d matrix: the
    reference's atomic::logdet / F.inverse() branch, nllk_ctcrw.hpp:12-24, 231-241): the oracle's LU recursion against the
    joint Gaussian of every scored observation of a track (tests/refimpl.py), values and autograd gradients"""
    from cases import make_spec
    from refimpl import ref_eval
    from smoothsde_amd import capi
    spec = make_spec("wide_pin", model, d, seed=70 + d, lengths=[9, 6, 2, 12], variant=variant, na_rows=(3, 11), with_H=True)
    sd = capi.state_dim(model, d)
    A = np.random.default_rng(d).standard_normal((sd, sd))
    spec["P0"] = A @ A.T / sd + np.eye(sd)
    pb = problem_from_spec(spec)
    val, grad = oracle_eval(pb, spec["par"], order=1)
    rval, rgrad = ref_eval(pb, spec["par"])
    assert np.isfinite(val)
    assert abs(val - rval) <= 1e-10 * max(1.0, abs(rval))
    assert np.max(np.abs(grad - rgrad)) <= 1e-8 * np.max(np.abs(rgrad)) + 1e-10


def test_reference_form_loses_the_likelihood_when_H_couples_the_columns():
    """A finding of round 5, kept as a test.  The reference propagates P as a FULL matrix, P <- T P (T - K Z)' + Q (nllk_ctcrw.hpp:240-241,
    Q8).  With a measurement covariance that couples the response columns (H_array with off-diagonal entries: Argos error ellipses) the
    antisymmetric part that rounding leaves in P is AMPLIFIED by that recursion (~1.2 per row here): the literal restatement in double
    drifts away from its own evaluation in binary128 -- 1e-10 after 100 rows, 1e-3 after 200 -- while the joint Gaussian of the track
    (tests/refimpl.py, no recursion) agrees with binary128 to 1e-11, and so does the restatement once P is kept symmetric (the identity
    in exact arithmetic; oracle/ssde_oracle.hpp: keep_P_symmetric, the ARBITER mode the GPU tests use on long tracks with such an H).
    With a diagonal H nothing of the kind happens."""
    from oracle_lib import keep_P_symmetric, oracle_eval_quad
    from refimpl import ref_eval
    from smoothsde_amd import capi
    from smoothsde_amd.synth import simulate
    fixed = np.array([1, 1, 1, 0, 0], dtype=np.uint8)
    theta = np.array([0.0, 0.0, 0.0, np.log(2.0), 0.0])
    drift = {}
    for T in (100, 200):
        ID, times, obs = simulate("CTCRW", 2, T, 2, tau=2.0, nu=1.0, sigma_obs=0.1, seed=13)
        n = len(ID)
        for label, H2 in (("coupling", [[0.005, 0.002], [0.002, 0.004]]), ("diagonal", [[0.005, 0.0], [0.0, 0.004]])):
            H = np.ascontiguousarray(np.transpose(np.tile(np.array(H2), (n, 1, 1)), (1, 2, 0)))
            pb = capi.Problem("CTCRW", ID, times, obs, par_fixed=fixed, H=H)
            lit = oracle_eval(pb, theta, order=0)
            q = oracle_eval_quad(pb, theta, order=0)
            keep_P_symmetric(True)
            try:
                sym, sym_grad = oracle_eval(pb, theta, order=1)
            finally:
                keep_P_symmetric(False)
            rel = lambda a: abs(a - q) / abs(q)     # noqa: E731
            drift[(T, label)] = rel(lit)
            assert rel(sym) <= 1e-11, (T, label, rel(sym))                      # the stabilised recursion = binary128
            if T == 100:
                rv, rg = ref_eval(pb, theta)                                    # ... = the joint Gaussian, value and gradient
                assert rel(rv) <= 1e-11, (label, rel(rv))
                assert np.max(np.abs(sym_grad - rg)) <= 1e-8 * np.max(np.abs(rg)), (label, sym_grad, rg)
    assert drift[(100, "diagonal")] <= 1e-13 and drift[(200, "diagonal")] <= 1e-13, drift
    assert drift[(100, "coupling")] >= 1e-12 and drift[(200, "coupling")] >= 1e-5, drift      # (measured 4e-11 and 3e-3)
    # The unstable mode is the CROSS-DIMENSION antisymmetric part of P: while those entries are exact zeros (H = sigma^2 I or diagonal,
    # block-diagonal P0) nothing seeds it.  A P0 with entries between the dimensions seeds it as well as a coupling H does -- and only
    # CTCRW has it: T is a multiple of the identity for OU_SSM / BM_SSM.
    A = np.random.default_rng(3).standard_normal((4, 4))
    for model, sd, unstable in (("CTCRW", 4, True), ("OU_SSM", 2, False)):
        ID, times, obs = simulate(model, 2, 600, 2, tau=2.0, nu=1.0, kappa=1.0, sigma_obs=0.1, seed=13)
        pb = capi.Problem(model, ID, times, obs, P0=(A @ A.T + np.eye(4))[:sd, :sd])
        par = np.zeros(pb.n_par_full)
        par[0], par[3] = np.log(0.07), np.log(2.0)
        lit = oracle_eval(pb, par, order=0)
        keep_P_symmetric(True)
        try:
            sym = oracle_eval(pb, par, order=0)
        finally:
            keep_P_symmetric(False)
        assert (abs(lit - sym) >= 1e-6 * abs(sym)) if unstable else (abs(lit - sym) <= 1e-13 * abs(sym)), (model, lit, sym)


def test_unstable_fixtures_for_a_machine_with_tmb_are_what_the_generator_writes():
    """tests/golden/unstable_cases.json (the hand-off of DESIGN 5c to a machine with R + TMB: tools/tmb_oracle.R reads it like
    cases.json) holds what tests/golden/gen_unstable.py computes: literal double, binary128 and arbiter values of four small cases"""
    import json
    import os
    from golden_io import dec
    from oracle_lib import keep_P_symmetric
    from smoothsde_amd import capi
    recs = [dec(r) for r in json.load(open(os.path.join(os.path.dirname(__file__), "golden", "unstable_cases.json")))]
    assert [r["name"] for r in recs] == ["unstable_H_coupling", "stable_H_diagonal", "unstable_P0_coupling", "stable_P0_default"]
    for r in recs:
        pb = capi.Problem("CTCRW", r["ID"], r["times"], r["obs"], par_fixed=r["par_fixed"], H=r["H"], P0=r["P0"])
        e = r["expected"]
        lit = oracle_eval(pb, r["par"], order=0)
        keep_P_symmetric(True)
        try:
            arb = oracle_eval(pb, r["par"], order=0)
        finally:
            keep_P_symmetric(False)
        assert abs(arb - e["arbiter_value"]) <= 1e-12 * abs(arb) and abs(arb - e["binary128_value"]) <= 1e-11 * abs(arb), r["name"]
        gap = abs(lit - e["binary128_value"]) / abs(e["binary128_value"])
        assert (gap >= 1e-5) if r["name"].startswith("unstable") else (gap <= 1e-13), (r["name"], gap)


def test_cir_weak_diffusion_against_mpmath_restatement():
    """CIR with sigma = 0.05: Bessel arguments of 10^3-10^4 and orders ~10^3, where the reference's unscaled
    besselI has long overflowed.  The oracle's series (summed outwards from its largest term) against the
    independent restatement whose log I_q and derivatives come from mpmath."""
    from refimpl import ref_eval
    from smoothsde_amd import capi
    rng = np.random.default_rng(77)
    n = 14
    ID = np.repeat([0.0, 1.0], [8, 6])
    times = np.cumsum(rng.uniform(0.5, 1.5, size=n))
    obs = np.exp(1.0 + 0.05 * np.cumsum(rng.standard_normal((n, 1)), axis=0))
    pb = capi.Problem("CIR", ID, times, obs)
    par = np.array([1.0, -0.3, np.log(0.05)])
    val, grad = oracle_eval(pb, par, order=1)
    rval, rgrad = ref_eval(pb, par)
    assert abs(val - rval) <= 1e-10 * max(1.0, abs(rval))
    assert np.max(np.abs(grad - rgrad)) <= 1e-8 * np.max(np.abs(rgrad)) + 1e-10


def test_oracle_threads_and_data_only():
    rec = next(r for r in GOLD if r["name"] == "CTCRW_d2_tv")
    pb = problem_from_spec(rec)
    v1, g1 = oracle_eval(pb, rec["par"], order=1, threads=1)
    v3, g3 = oracle_eval(pb, rec["par"], order=1, threads=3)
    assert abs(v1 - v3) <= 1e-12 * abs(v1) and np.allclose(g1, g3, rtol=1e-11, atol=1e-12)
    vd, gd = oracle_eval(pb, rec["par"], order=1, data_only=True)
    assert vd != v1  # the penalty is non-zero in this case


def test_oracle_finite_difference_gradient():
    rec = next(r for r in GOLD if r["name"] == "CTCRW_d2_const")
    pb = problem_from_spec(rec)
    par = rec["par"]
    _, g = oracle_eval(pb, par, order=1)
    for k in range(pb.n_par_full):
        h = 1e-6
        e = np.zeros_like(par)
        e[k] = h
        fd = (oracle_eval(pb, par + e, order=0) - oracle_eval(pb, par - e, order=0)) / (2 * h)
        assert abs(fd - g[k]) <= 1e-6 * max(1.0, abs(g[k]))


def test_first_observation_never_scored():
    """Quirk Q1: changing a track's first observation only moves a0."""
    rec = next(r for r in GOLD if r["name"] == "BM_SSM_d1_const")
    pb = problem_from_spec(rec)
    a0 = np.zeros((pb.n_seg, pb.sdim))
    a0[:, 0] = pb.obs[pb.seg_start, 0]
    obs2 = rec["obs"].copy()
    obs2[pb.seg_start, 0] += 123.0
    pb2 = problem_from_spec(dict(rec, obs=obs2), a0=a0)
    assert abs(oracle_eval(pb, rec["par"], 0) - oracle_eval(pb2, rec["par"], 0)) < 1e-12


def test_cpu_fast_matches_the_oracle():
    """bench.py's CPU baseline (oracle/cpu_fast.cpp: the engine's hand-derived step compiled for the host, threads over
    tracks) against the oracle: it is a timing baseline, not a checker, and is itself checked here."""
    import numpy as np
    from oracle_lib import cpu_fast_eval, oracle_eval
    from smoothsde_amd import capi
    from smoothsde_amd.synth import simulate
    for model, par, fixed in (("CTCRW", [-1.0, 0.05, -0.02, 0.4, 0.1], [0, 0, 0, 0, 0]), ("CTCRW", [-2.3, 0.0, 0.0, 0.7, 0.0], [0, 1, 1, 0, 0]),
                              ("OU_SSM", [-1.0, 0.3, -0.2, 0.6, 0.1], [0, 0, 0, 0, 0]), ("BM_SSM", [-1.0, 0.05, 0.0, 0.2], [0, 0, 0, 0])):
        for irregular in (False, True):
            ID, times, obs = simulate(model, 37, 120, 2, seed=3)
            if irregular:
                times = np.cumsum(np.random.default_rng(1).uniform(0.5, 1.5, len(ID)))
            obs[11::29] = np.nan
            obs[::120] = np.where(np.isnan(obs[::120]), 0.0, obs[::120])
            pb = capi.Problem(model, ID, times, obs, par_fixed=fixed)
            v, g = cpu_fast_eval(pb, np.array(par), threads=3)
            ov, og = oracle_eval(pb, np.array(par), order=1, data_only=True)
            assert abs(v - ov) <= 1e-11 * abs(ov), (model, irregular, v, ov)
            assert np.max(np.abs(g - og)) <= 1e-9 * np.max(np.abs(og)), (model, irregular, g, og)
