"""GPU suite (-m gpu): time grids that are a regular schedule with fixes MISSING FROM THE DATA (not NA-padded).

The reference evaluates such data with one transition per interval, makeT/makeQ/makeB(k * step) (nllk_ctcrw.hpp:45-91, 206-241).
The engine recognises the lattice at ssde_create, lays the tracks out on it with NA rows where the data have none, and runs
the regular-grid kernels (csrc/ssde_engine.hip: lattice_pad): the k-step transition is the k-fold product of the one-step
transition and a missing row is exactly one prediction step.  Checked here against the oracle, which evaluates the caller's
irregular rows directly.  Tolerances as everywhere: value 1e-10 relative, gradient 1e-8 * max|g| + 1e-10."""
import numpy as np
import pytest

from oracle_lib import oracle_eval
from smoothsde_amd import capi
from smoothsde_amd.capi import na_real

pytestmark = pytest.mark.gpu


def _close(val, grad, eval_, egrad):
    assert abs(val - eval_) <= 1e-10 * max(1.0, abs(eval_)), (val, eval_)
    assert np.max(np.abs(grad - egrad)) <= 1e-8 * np.max(np.abs(egrad)) + 1e-10, (grad, egrad)


def lattice_tracks(model, d, lengths, step, drop, seed, max_gap=3, na_frac=0.0, na_mode=1):
    """tracks on a lattice of `step`, each row kept with probability 1 - drop (runs of at most max_gap absent fixes)"""
    rng = np.random.default_rng(seed)
    ID, times, obs = [], [], []
    t0 = 3.0
    for k, T in enumerate(lengths):
        keep = rng.random(T) >= drop
        keep[0] = keep[-1] = True
        gap = 0
        for i in range(T):                      # no run of absent fixes longer than max_gap
            gap = 0 if keep[i] else gap + 1
            if gap > max_gap:
                keep[i] = True
                gap = 0
        idx = np.flatnonzero(keep)
        tt = t0 + step * idx                    # (step = 0.1: multiples of the step to rounding, not bitwise)
        t0 = tt[-1] + 7.25                      # the interval to the next track is NOT on the lattice (but exact in binary)
        z = np.cumsum(rng.standard_normal((T, d)) * (1.0 if model == "CTCRW" else 0.3), axis=0)[idx]
        if model == "OU_SSM":
            z = 3.0 + rng.standard_normal((len(idx), d))
        ID += [float(k)] * len(idx)
        times += list(tt)
        obs.append(z)
    ID, times, obs = np.array(ID), np.array(times), np.vstack(obs)
    if na_frac > 0:
        first = np.r_[True, ID[1:] != ID[:-1]]
        na = (rng.random(len(ID)) < na_frac) & ~first
        obs[na, :] = na_real() if na_mode == 0 else np.nan
    return ID, times, obs


def _par(model, d, rng):
    q = capi.n_sde_par(model, d)
    p = [rng.uniform(-1.5, -0.3)] + list(rng.uniform(-0.3, 0.3, size=d) + (3.0 if model == "OU_SSM" else 0.0))
    return np.array(p + list(rng.uniform(-0.2, 0.6, size=q - d)))


@pytest.mark.parametrize("model", ["CTCRW", "OU_SSM", "BM_SSM"])
@pytest.mark.parametrize("d", [1, 2])
@pytest.mark.parametrize("step,na_frac,na_mode", [(0.5, 0.0, 1), (0.1, 0.0, 1), (1.0, 0.05, 0), (0.25, 0.04, 1)])
def test_lattice_with_absent_fixes_matches_the_oracle_on_the_callers_rows(model, d, step, na_frac, na_mode, monkeypatch):
    ID, times, obs = lattice_tracks(model, d, [60, 9, 33, 2, 48, 1, 27], step, 0.12, seed=11 + d, na_frac=na_frac, na_mode=na_mode)
    pb = capi.Problem(model, ID, times, obs, na_mode=na_mode)
    par = _par(model, d, np.random.default_rng(5))
    oval, ograd, oaest = oracle_eval(pb, par, order=1, report=True)
    eng = capi.Engine(pb)
    info = eng.info()
    assert info["path"] == 1 and info["uniform_dt"] == 1, info            # laid out on the lattice: the hoisted transition
    assert info["n_rows"] == pb.n and pb.n < info["n_rows_tiled"] <= 1.35 * pb.n + 64
    val, grad = eng.eval(par, order=1)
    assert eng.info()["kernel_id"] in (3, 5, 6, 7)                          # a regular-grid kernel (shared / hoisted transition / quiet rows / mixed)
    _close(val, grad, oval, ograd)
    assert np.allclose(eng.report(par), oaest, rtol=1e-9, atol=1e-9, equal_nan=True)
    eng.close()
    monkeypatch.setenv("SSDE_NO_LATTICE", "1")                             # the same rows through the irregular-grid kernel
    eng = capi.Engine(pb)
    assert eng.info()["uniform_dt"] == 0
    v2, g2 = eng.eval(par, order=1)
    assert eng.info()["kernel_id"] == 4                                     # iso_mask_kernel, per-row transition
    _close(v2, g2, oval, ograd)
    eng.close()


def test_supplied_a0_fixed_mu_and_device_resident_rows():
    import torch
    ID, times, obs = lattice_tracks("CTCRW", 2, [80, 45, 63], 0.5, 0.1, seed=3)
    rng = np.random.default_rng(2)
    a0 = rng.standard_normal((3, 4))
    fixed = np.array([0, 1, 1, 0, 0], dtype=np.uint8)
    pb = capi.Problem("CTCRW", ID, times, obs, a0=a0, par_fixed=fixed)
    par = _par("CTCRW", 2, rng)
    eng = capi.Engine(pb)
    val, grad = eng.eval(par, order=1)
    oval, ograd = oracle_eval(pb, par, order=1)
    _close(val, grad, oval, ograd)
    eng.close()
    dev = torch.device("cuda:0")
    pbh = capi.Problem("CTCRW", ID, times, obs, par_fixed=fixed)
    eh = capi.Engine(pbh)
    vh, gh = eh.eval(par, order=1)
    pbd = capi.Problem.from_torch("CTCRW", torch.as_tensor(ID, device=dev), torch.as_tensor(times, device=dev),
                                  torch.as_tensor(obs, device=dev), par_fixed=fixed)
    ed = capi.Engine(pbd)
    assert ed.info()["uniform_dt"] == 1
    vd, gd = ed.eval(par, order=1)
    assert vd == vh and np.array_equal(gd, gh)
    assert np.array_equal(ed.report(par), eh.report(par))
    eh.close(); ed.close()


def test_batch_with_gap_free_groups_on_the_shared_path_and_windows():
    """320 tracks x ~900 rows: the tracks of four 64-track groups have every fix, the fifth group's tracks miss 3 %: time
    windows, the shared-covariance kernel for the gap-free groups, the general kernel for the others, one result"""
    rng = np.random.default_rng(8)
    parts = [lattice_tracks("CTCRW", 2, [900] * 256, 1.0, 0.0, seed=1), lattice_tracks("CTCRW", 2, [880] * 64, 1.0, 0.03, seed=2)]
    ID = np.concatenate([parts[0][0], parts[1][0] + 1000])
    times = np.concatenate([parts[0][1], parts[1][1] + parts[0][1][-1] + 11.0])
    obs = np.vstack([parts[0][2], parts[1][2]])
    pb = capi.Problem("CTCRW", ID, times, obs)
    par = np.array([-1.0, 0.05, -0.05, 0.4, 0.1])
    eng = capi.Engine(pb)
    info = eng.info()
    assert info["uniform_dt"] == 1 and info["path"] == 1
    val, grad = eng.eval(par, order=1)
    info = eng.info()
    assert info["lanes_per_track"] > 1 and info["window_check"] <= 1e-11
    oval, ograd = oracle_eval(pb, par, order=1, threads=8)
    _close(val, grad, oval, ograd)
    eng.close()


@pytest.mark.parametrize("kind", ["continuous", "too_sparse", "wide_gap", "off_lattice"])
def test_what_is_not_a_cheap_lattice_keeps_the_irregular_kernel(kind):
    rng = np.random.default_rng(4)
    if kind == "continuous":
        ID = np.repeat(np.arange(3.0), 50); times = np.cumsum(rng.uniform(0.4, 1.7, size=150))
    elif kind == "too_sparse":                    # half the fixes absent: more lattice rows than the kernels' break-even
        ID, times, _ = lattice_tracks("CTCRW", 1, [100, 100, 100], 1.0, 0.5, seed=1)
    elif kind == "wide_gap":                      # one interval of 40 steps
        ID = np.repeat(np.arange(2.0), 60); t = np.arange(120.0); t[30:] += 40.0; times = t
    else:                                         # one interval 1.5 steps long
        ID = np.repeat(np.arange(2.0), 60); t = np.arange(120.0); t[30:] += 0.5; times = t
    obs = np.cumsum(rng.standard_normal((len(ID), 1)), axis=0)
    pb = capi.Problem("CTCRW", ID, times, obs)
    par = np.array([-1.0, 0.0, 0.3, 0.1])
    eng = capi.Engine(pb)
    assert eng.info()["uniform_dt"] == 0
    val, grad = eng.eval(par, order=1)
    oval, ograd = oracle_eval(pb, par, order=1)
    _close(val, grad, oval, ograd)
    eng.close()


_LLO, _LHI = (int(v) for v in __import__("os").environ.get("SSDE_FUZZ_LATTICE_SEEDS", "0:48").split(":"))


@pytest.mark.parametrize("seed", range(_LLO, _LHI))
def test_random_lattice_problems(seed):
    rng = np.random.default_rng(1000 + seed)
    model = ["CTCRW", "OU_SSM", "BM_SSM"][seed % 3]
    d = 1 + (seed // 3) % 2
    n_tracks = int(rng.integers(1, 9)) if rng.random() < 0.6 else int(rng.integers(60, 140))
    lengths = rng.integers(1, 40, size=n_tracks) if rng.random() < 0.5 else rng.integers(100, 500, size=n_tracks)
    ID, times, obs = lattice_tracks(model, d, list(lengths), float(rng.choice([1.0, 0.5, 0.1, 2.5, 1 / 3])), float(rng.choice([0.02, 0.1, 0.2])),
                                    seed=seed, max_gap=int(rng.integers(1, 6)), na_frac=float(rng.choice([0.0, 0.05])), na_mode=seed % 2)
    if len(ID) < 2:
        pytest.skip("degenerate draw")
    pb = capi.Problem(model, ID, times, obs, na_mode=seed % 2)
    par = _par(model, d, rng)
    fixed = (rng.random(pb.n_par_full) < 0.2).astype(np.uint8)
    if fixed.all():
        fixed[0] = 0
    pb.par_fixed = fixed
    eng = capi.Engine(pb)
    val, grad = eng.eval(par, order=1)
    oval, ograd, oaest = oracle_eval(pb, par, order=1, report=True, threads=4)
    ctx = (model, d, pb.n, pb.n_seg, eng.info()["uniform_dt"], eng.info()["window"])
    assert abs(val - oval) <= 1e-10 * max(1.0, abs(oval)), (ctx, val, oval)
    assert np.max(np.abs(grad - ograd)) <= 1e-8 * np.max(np.abs(ograd)) + 1e-10, (ctx, grad, ograd)
    assert np.allclose(eng.report(par), oaest, rtol=1e-9, atol=1e-9, equal_nan=True), ctx
    eng.close()


def test_lattice_composes_with_wide_responses_and_track_shards():
    """every engine behind a handle (dimension parts, track shards) lays its own rows out on the lattice"""
    ID, times, obs = lattice_tracks("CTCRW", 3, [70, 31, 52, 44, 66, 23], 0.5, 0.1, seed=6, na_frac=0.03)
    pb = capi.Problem("CTCRW", ID, times, obs)
    par = np.array([-0.9, 0.02, -0.03, 0.04, 0.3, 0.1])
    oval, ograd, oaest = oracle_eval(pb, par, order=1, report=True)
    for devices in (None, [0, 0, 0]):
        eng = capi.Engine(pb, devices=devices)
        assert eng.info()["uniform_dt"] == 1
        val, grad = eng.eval(par, order=1)
        _close(val, grad, oval, ograd)
        assert np.allclose(eng.report(par), oaest, rtol=1e-9, atol=1e-9, equal_nan=True)
        eng.close()


@pytest.mark.parametrize("model", ["CTCRW", "OU_SSM"])
def test_tracks_with_missing_rows_get_wavefronts_of_their_own(model, monkeypatch):
    """a third of the tracks (every third one) have missing rows: dealt to wavefronts of their own, the others keep the
    shared-covariance kernel; same numbers as with the caller's order of tracks (SSDE_NO_REGROUP=1) and as the oracle"""
    from smoothsde_amd.synth import simulate
    M, T, d = 384, 700, 2
    ID, times, obs = simulate(model, M, T, d, seed=9)
    rng = np.random.default_rng(1)
    for k in range(0, M, 3):
        rows = k * T + rng.integers(1, T, size=5)
        obs[rows, :] = np.nan
    pb = capi.Problem(model, ID, times, obs)
    par = _par(model, d, rng)
    eng = capi.Engine(pb)
    info = eng.info()
    assert info["n_groups"] == 6 and info["n_clean_groups"] == 4          # 256 complete tracks = four wavefronts of their own
    val, grad = eng.eval(par, order=1)
    aest = eng.report(par)
    eng.close()
    monkeypatch.setenv("SSDE_NO_REGROUP", "1")
    e2 = capi.Engine(pb)
    assert e2.info()["n_clean_groups"] == 0                               # every third track: no wavefront without one
    v2, g2 = e2.eval(par, order=1)
    assert abs(v2 - val) <= 1e-12 * abs(val) and np.max(np.abs(g2 - grad)) <= 1e-10 * max(1.0, np.max(np.abs(grad)))
    assert np.allclose(e2.report(par), aest, rtol=1e-12, atol=1e-12, equal_nan=True)
    e2.close()
    oval, ograd = oracle_eval(pb, par, order=1, threads=8)
    _close(val, grad, oval, ograd)


@pytest.mark.parametrize("model", ["CTCRW", "OU_SSM"])
def test_decimal_step_is_regular_to_the_last_bits(model, monkeypatch):
    """times = 0.1 * (0, 1, 2, ...): the intervals differ in their last bits.  Within 1e-12 of one step the grid is taken as
    regular (hoisted transition); the numbers stay within the parity tolerance of the oracle, which uses every row's own dt"""
    rng = np.random.default_rng(3)
    lens = [120, 64, 200]
    ID = np.repeat(np.arange(3.0), lens)
    times = 0.1 * np.arange(len(ID))
    assert len(np.unique(np.diff(times))) > 1                     # not bitwise regular
    obs = np.cumsum(rng.standard_normal((len(ID), 2)) * 0.3, axis=0) + (3.0 if model == "OU_SSM" else 0.0)
    pb = capi.Problem(model, ID, times, obs)
    par = _par(model, 2, rng)
    eng = capi.Engine(pb)
    assert eng.info()["uniform_dt"] == 1
    val, grad = eng.eval(par, order=1)
    oval, ograd = oracle_eval(pb, par, order=1)
    _close(val, grad, oval, ograd)
    eng.close()
    monkeypatch.setenv("SSDE_GRID_RTOL", "0")                     # strict: the per-row transition
    eng = capi.Engine(pb)
    assert eng.info()["uniform_dt"] == 0
    v2, g2 = eng.eval(par, order=1)
    _close(v2, g2, oval, ograd)
    eng.close()


def test_a_tracks_first_interval_does_not_decide_regularity():
    """the first interval of a track is never used (a0 is the prediction for the second row as it stands): a grid that is
    regular from the second row on gets the hoisted transition whatever the first stamps are"""
    rng = np.random.default_rng(12)
    ID = np.repeat(np.arange(4.0), 80)
    times = np.arange(320.0)
    times[::80] -= rng.uniform(0.1, 5.0, size=4)          # every track's first stamp is somewhere earlier
    obs = np.cumsum(rng.standard_normal((320, 2)), axis=0)
    pb = capi.Problem("CTCRW", ID, times, obs)
    par = _par("CTCRW", 2, rng)
    eng = capi.Engine(pb)
    assert eng.info()["uniform_dt"] == 1
    val, grad = eng.eval(par, order=1)
    oval, ograd, oaest = oracle_eval(pb, par, order=1, report=True)
    _close(val, grad, oval, ograd)
    assert np.allclose(eng.report(par), oaest, rtol=1e-9, atol=1e-9, equal_nan=True)
    eng.close()
