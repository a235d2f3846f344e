"""GPU suite (-m gpu): QUIET ROWS of the general kernel (csrc/k_iso.hip).

A regular-grid batch in which every wavefront holds a few missing rows has no group for the shared-covariance kernel.  The
general lanes drop their covariance wherever it is stationary on all 64 lanes of the wavefront -- no missing observation in the
block or in the `quiet_window` rows before it, past the transient of P0 -- and run the mean half with the stationary gains there
(nllk_ctcrw.hpp:214-241: a row without an observation is a prediction step; every other row sees the same K and F once P has
forgotten the last such step).  Tracks are dealt to wavefronts by WHERE they miss rows, so that the lanes of a wavefront leave the
stationary regime together.  Checked against the oracle and against the same engine with SSDE_NO_QUIET=1 (every row on the
lane's own covariance)."""
import numpy as np
import pytest

from oracle_lib import oracle_eval
from smoothsde_amd import capi
from smoothsde_amd.capi import na_real
from smoothsde_amd.synth import simulate

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def _flags_whatever_the_share(monkeypatch):
    """the batches here are small (a few wavefronts, whose missing rows spread over much of the track): keep the block flags
    whatever share of the blocks qualifies; the tests of the decision itself remove the variable"""
    monkeypatch.setenv("SSDE_QUIET_ALWAYS", "1")


def _par(model, d, rng):
    q = capi.n_sde_par(model, d)
    p = [rng.uniform(-1.5, -0.3)] + list(rng.uniform(-0.3, 0.3, size=d) + (3.0 if model == "OU_SSM" else 0.0))
    return np.array(p + list(rng.uniform(-0.2, 0.6, size=q - d)))


def _close(val, grad, eval_, egrad):
    assert abs(val - eval_) <= 1e-10 * max(1.0, abs(eval_)), (val, eval_)
    assert np.max(np.abs(grad - egrad)) <= 1e-8 * np.max(np.abs(egrad)) + 1e-10, (grad, egrad)


def _batch(model, M, T, d, seed, n_na=1, col0_only=False, first_rows=False):
    """every track misses n_na rows (uniform positions; `first_rows`: the track's second row and its last one as well)"""
    ID, times, obs = simulate(model, M, T, d, seed=seed)
    rng = np.random.default_rng(seed + 100)
    for k in range(M):
        rows = list(rng.integers(1, T, size=n_na))
        if first_rows and k % 7 == 0:
            rows += [1, T - 1]
        for r in rows:
            if col0_only:
                obs[k * T + r, 0] = na_real()
            else:
                obs[k * T + r, :] = np.nan
    return ID, times, obs


@pytest.mark.parametrize("model", ["CTCRW", "OU_SSM", "BM_SSM"])
@pytest.mark.parametrize("d", [1, 2])
def test_one_missing_row_per_track_against_the_oracle(model, d, monkeypatch):
    M, T = 256, 900
    ID, times, obs = _batch(model, M, T, d, seed=3 + d, first_rows=True)
    pb = capi.Problem(model, ID, times, obs)
    rng = np.random.default_rng(5)
    par = _par(model, d, rng)
    eng = capi.Engine(pb)
    inf = eng.info()
    assert inf["n_clean_groups"] == 0 and inf["quiet_share"] > 0.3, inf
    val, grad = eng.eval(par, order=1)
    inf = eng.info()
    assert inf["quiet_window"] > 0 and inf["window_retries"] == 0, inf
    v0 = eng.eval(par, order=0)
    v0 = v0[0] if isinstance(v0, tuple) else v0
    assert abs(v0 - val) <= 1e-12 * abs(val)
    eng.close()
    oval, ograd = oracle_eval(pb, par, order=1, threads=8)
    _close(val, grad, oval, ograd)
    monkeypatch.setenv("SSDE_NO_QUIET", "1")
    e2 = capi.Engine(pb)
    v2, g2 = e2.eval(par, order=1)
    assert e2.info()["quiet_window"] == 0 and e2.info()["kernel_id"] in (4, 5, 7)       # the general lanes without quiet rows (7: next to the shared kernel)
    e2.close()
    assert abs(v2 - val) <= 1e-11 * abs(val) and np.max(np.abs(g2 - grad)) <= 1e-9 * max(1.0, np.max(np.abs(grad)))


@pytest.mark.parametrize("model", ["CTCRW", "OU_SSM"])
def test_windows_and_one_window_agree(model, monkeypatch):
    """the same batch as one sequential window per track (SSDE_CHUNKS=1) and with the planned windows: the hand-over check
    passes (quiet lanes dump the stationary covariance) and the sums agree"""
    M, T, d = 192, 2400, 2
    ID, times, obs = _batch(model, M, T, d, seed=11, n_na=2)
    pb = capi.Problem(model, ID, times, obs)
    par = _par(model, d, np.random.default_rng(8))
    eng = capi.Engine(pb)
    val, grad = eng.eval(par, order=1)
    inf = eng.info()
    assert inf["quiet_window"] > 0 and inf["lanes_per_track"] > 1 and inf["window_retries"] == 0 and inf["window_check"] < 1e-11, inf
    eng.close()
    monkeypatch.setenv("SSDE_CHUNKS", "1")
    e1 = capi.Engine(pb)
    v1, g1 = e1.eval(par, order=1)
    assert e1.info()["lanes_per_track"] == 1
    e1.close()
    assert abs(v1 - val) <= 1e-11 * abs(val) and np.max(np.abs(g1 - grad)) <= 1e-9 * max(1.0, np.max(np.abs(grad)))
    oval, ograd = oracle_eval(pb, par, order=1, threads=8)
    _close(val, grad, oval, ograd)


def test_missing_first_column_only_and_fixed_parameters():
    """NA_real_ in the first response column only is a missing row (nllk_ctcrw.hpp:214); fixed parameters change the direction
    mask of the kernel"""
    M, T, d = 128, 800, 2
    ID, times, obs = _batch("CTCRW", M, T, d, seed=21, col0_only=True)
    for fixed in ([], [0], [3], [0, 4]):
        pf = np.zeros(1 + d + 2, dtype=np.uint8)
        pf[fixed] = 1
        pb = capi.Problem("CTCRW", ID, times, obs, par_fixed=pf)
        par = _par("CTCRW", d, np.random.default_rng(2))
        eng = capi.Engine(pb)
        val, grad = eng.eval(par, order=1)
        assert eng.info()["quiet_window"] > 0
        eng.close()
        oval, ograd = oracle_eval(pb, par, order=1, threads=8)
        _close(val, grad, oval, ograd)


def test_mixed_batch_keeps_its_complete_groups_on_the_shared_kernel():
    """half of the tracks complete (shared-covariance kernel, plan of its own), the others with one missing row each"""
    M, T, d = 512, 1600, 2
    ID, times, obs = simulate("CTCRW", M, T, d, seed=4)
    rng = np.random.default_rng(9)
    for k in range(0, M, 2):
        obs[k * T + rng.integers(1, T), :] = np.nan
    pb = capi.Problem("CTCRW", ID, times, obs)
    par = _par("CTCRW", d, rng)
    eng = capi.Engine(pb)
    inf = eng.info()
    assert inf["n_clean_groups"] == 4 and inf["n_groups"] == 8
    val, grad = eng.eval(par, order=1)
    assert eng.info()["quiet_window"] > 0 and eng.info()["window_retries"] == 0
    eng.close()
    oval, ograd = oracle_eval(pb, par, order=1, threads=8)
    _close(val, grad, oval, ograd)


@pytest.mark.parametrize("model", ["CTCRW", "OU_SSM"])
def test_the_share_of_quiet_blocks_decides(model, monkeypatch):
    """one missing row per track of 2000 rows, four wavefronts: two thirds of the blocks qualify -- quiet rows for every model;
    three per track: few do -- the kernels with quiet rows want a quarter (CTCRW) or a fifth of them"""
    monkeypatch.delenv("SSDE_QUIET_ALWAYS")
    rng = np.random.default_rng(4)
    for n_na in (1, 3):
        ID, times, obs = _batch(model, 256, 2000, 2, seed=31, n_na=n_na)
        pb = capi.Problem(model, ID, times, obs)
        par = _par(model, 2, rng)
        eng = capi.Engine(pb)
        val, grad = eng.eval(par, order=1)
        inf = eng.info()
        eng.close()
        want = inf["quiet_share"] >= (0.25 if model == "CTCRW" else 0.2)
        assert (inf["quiet_window"] > 0) == want and (want or n_na == 3), (n_na, inf["quiet_share"], inf["quiet_window"])
        oval, ograd = oracle_eval(pb, par, order=1, threads=8)
        _close(val, grad, oval, ograd)


def test_dense_missing_rows_stay_on_the_lanes_own_covariance(monkeypatch):
    """5 % of the rows missing: no block of a wavefront qualifies, the handle does not carry the flags"""
    monkeypatch.delenv("SSDE_QUIET_ALWAYS")
    M, T, d = 128, 600, 2
    ID, times, obs = simulate("CTCRW", M, T, d, seed=6)
    rng = np.random.default_rng(3)
    first = np.r_[True, ID[1:] != ID[:-1]]
    obs[(rng.random(len(ID)) < 0.05) & ~first, :] = np.nan
    pb = capi.Problem("CTCRW", ID, times, obs)
    eng = capi.Engine(pb)
    par = _par("CTCRW", d, rng)
    val, grad = eng.eval(par, order=1)
    inf = eng.info()
    assert inf["quiet_share"] < 0.2 and inf["quiet_window"] == 0
    eng.close()
    oval, ograd = oracle_eval(pb, par, order=1, threads=8)
    _close(val, grad, oval, ograd)


def test_ragged_tracks_and_parameter_moves():
    """ragged lengths (padding rows of the shorter lanes are not missing rows; tracks are dealt by length first, so few blocks
    qualify) and a sequence of parameter vectors on one handle (the memory of the covariance follows the plan's warm-up)"""
    rng = np.random.default_rng(17)
    M, d = 200, 2
    lens = rng.integers(500, 1000, size=M)
    ID = np.repeat(np.arange(float(M)), lens)
    times = np.arange(len(ID), dtype=float)
    obs = np.cumsum(rng.standard_normal((len(ID), d)), axis=0)
    starts = np.r_[0, np.cumsum(lens)[:-1]]
    for k in range(M):
        obs[starts[k] + rng.integers(1, lens[k]), :] = np.nan
    pb = capi.Problem("CTCRW", ID, times, obs)
    eng = capi.Engine(pb)
    for i in range(4):
        par = _par("CTCRW", d, rng) + np.r_[0.0, 0.0, 0.0, 0.4 * i, -0.3 * i]
        val, grad = eng.eval(par, order=1)
        oval, ograd = oracle_eval(pb, par, order=1, threads=8)
        _close(val, grad, oval, ograd)
    assert eng.info()["quiet_window"] > 0
    eng.close()


@pytest.mark.parametrize("model", ["CTCRW", "OU_SSM"])
def test_a_schedule_with_one_fix_absent_per_track(model):
    """the missing fixes are simply not in the data (an interval of two steps): laid out on the lattice at create, the absent
    fix becomes a row without an observation -- and everything W rows past it a quiet row"""
    rng = np.random.default_rng(23)
    M, T, d = 128, 900, 2
    ID, times, obs = simulate(model, M, T, d, seed=13)
    keep = np.ones(len(ID), dtype=bool)
    for k in range(M):
        keep[k * T + rng.integers(2, T - 2)] = False
    ID, times, obs = ID[keep], times[keep], obs[keep]
    pb = capi.Problem(model, ID, times, obs)
    eng = capi.Engine(pb)
    inf = eng.info()
    assert inf["uniform_dt"] == 1 and inf["n_rows_tiled"] == M * T and inf["quiet_share"] > 0.2, inf
    par = _par(model, d, rng)
    val, grad = eng.eval(par, order=1)
    assert eng.info()["quiet_window"] > 0
    aest = eng.report(par)
    eng.close()
    oval, ograd, oaest = oracle_eval(pb, par, order=1, threads=8, report=True)
    _close(val, grad, oval, ograd)
    assert np.allclose(aest, oaest, rtol=1e-9, atol=1e-9, equal_nan=True)


def test_asynchronous_evaluations_and_a_one_rank_communicator():
    """ssde_eval_device on a caller's stream (no gain table is uploaded for a handle without complete wavefronts: nothing of
    the previous evaluation is in flight when the next one plans), then the same handle behind a rank communicator"""
    import torch
    M, T, d = 192, 1000, 2
    ID, times, obs = _batch("CTCRW", M, T, d, seed=41)
    pb = capi.Problem("CTCRW", ID, times, obs)
    eng = capi.Engine(pb)
    rng = np.random.default_rng(6)
    pars = [_par("CTCRW", d, rng) for _ in range(4)]
    outs = [torch.zeros(2 + pb.n_par_full, dtype=torch.float64, device="cuda:0") for _ in pars]
    st = torch.cuda.Stream()
    with torch.cuda.stream(st):
        for p, o in zip(pars, outs):
            eng.eval_device(p, o.data_ptr(), order=1, stream=st.cuda_stream)
    st.synchronize()
    assert eng.info()["quiet_window"] > 0
    for p, o in zip(pars, outs):
        v, g = eng.eval(p, order=1)
        pen = eng.penalty(p)[0]
        assert abs(o[0].item() + pen - v) <= 1e-12 * abs(v) and np.max(np.abs(o[1:-1].cpu().numpy() - g)) <= 1e-10 * max(1.0, np.max(np.abs(g)))
        assert o[-1].item() < 1e-11
    eng.close()


@pytest.mark.parametrize("model", ["CTCRW", "OU_SSM"])
def test_a_memory_too_short_is_detected_at_the_switch_and_repaired(model, monkeypatch):
    """SSDE_QUIET_WINDOW=8: the lanes would return to quiet rows three blocks after a missing row, long before their covariance
    has forgotten it (the windows keep their planned warm-up: their hand-over checks have nothing to report).  The switch compares the stationary lanes' state and the stationary covariance with the general lane's, the
    evaluation reports the disagreement with the hand-over checks' and is repeated with a longer memory -- never returned
    silently"""
    M, T, d = 128, 1200, 2
    ID, times, obs = _batch(model, M, T, d, seed=51)
    pb = capi.Problem(model, ID, times, obs)
    par = np.array([0.5, 0.0, 0.0, 0.5, 0.0]) + (np.array([0, 3.0, 3.0, 0, 0]) if model == "OU_SSM" else 0.0)   # sigma_obs = 1.6: slow forgetting
    monkeypatch.setenv("SSDE_QUIET_WINDOW", "8")
    eng = capi.Engine(pb)
    val, grad = eng.eval(par, order=1)
    inf = eng.info()
    eng.close()
    assert inf["window_retries"] >= 1, inf
    oval, ograd = oracle_eval(pb, par, order=1, threads=8)
    _close(val, grad, oval, ograd)


@pytest.mark.parametrize("model", ["CTCRW", "OU_SSM"])
def test_a_memory_too_short_is_repaired_on_a_one_window_plan_too(model, monkeypatch):
    """SSDE_CHUNKS=1: one window, so no hand-over exists -- but quiet rows still run (the plan keeps a usable warm-up length) and
    their switch check is the only thing that can notice a memory that is too short.  ssde_eval used to accept any check value
    on a one-window plan; it must retry with a longer memory like everywhere else."""
    M, T, d = 128, 1200, 2
    ID, times, obs = _batch(model, M, T, d, seed=52)
    pb = capi.Problem(model, ID, times, obs)
    par = np.array([0.5, 0.0, 0.0, 0.5, 0.0]) + (np.array([0, 3.0, 3.0, 0, 0]) if model == "OU_SSM" else 0.0)
    monkeypatch.setenv("SSDE_QUIET_WINDOW", "8")
    monkeypatch.setenv("SSDE_CHUNKS", "1")
    eng = capi.Engine(pb)
    val, grad = eng.eval(par, order=1)
    inf = eng.info()
    eng.close()
    assert inf["lanes_per_track"] == 1, inf
    assert inf["window_retries"] >= 1, inf
    oval, ograd = oracle_eval(pb, par, order=1, threads=8)
    _close(val, grad, oval, ograd)


@pytest.mark.timeout(300)
@pytest.mark.parametrize("seed", range(24))
def test_random_sparse_patterns(seed, monkeypatch):
    """random model, width, length, number and place of the missing rows (runs of them, rows next to block and window
    boundaries, the second and the last row of a track), random share of complete tracks, a forced number of windows now and
    then: value and gradient against the oracle, with and without the dealing by position"""
    rng = np.random.default_rng(1000 + seed)
    model = ["CTCRW", "OU_SSM", "BM_SSM"][seed % 3]
    d = int(rng.integers(1, 3))
    M = int(rng.choice([64, 130, 200]))
    T = int(rng.integers(500, 1400))
    ID, times, obs = simulate(model, M, T, d, seed=200 + seed)
    for k in range(M):
        if rng.random() < 0.25:
            continue                                   # a complete track
        n_na = int(rng.integers(1, 4))
        for _ in range(n_na):
            r = int(rng.choice([1, T - 1, rng.integers(1, T), 16 * rng.integers(1, T // 16), 8 * rng.integers(1, T // 8) - 1]))
            run = int(rng.choice([1, 1, 2, 9]))
            obs[k * T + r: k * T + min(T, r + run), :] = np.nan
    if seed % 4 == 1:
        monkeypatch.setenv("SSDE_CHUNKS", str(int(rng.integers(2, 9))))
    if seed % 2 == 1:
        monkeypatch.setenv("SSDE_NO_NA_SORT", "1")
    pb = capi.Problem(model, ID, times, obs)
    par = _par(model, d, rng)
    if seed % 5 == 2:
        par[0] += 1.2                                  # a larger sigma_obs: slower forgetting, longer memories and warm-ups
    eng = capi.Engine(pb)
    val, grad = eng.eval(par, order=1)
    inf = eng.info()
    eng.close()
    assert inf["window_check_max"] <= 1e-11 or inf["window_retries"] >= 1, inf
    oval, ograd = oracle_eval(pb, par, order=1, threads=8)
    _close(val, grad, oval, ograd)
