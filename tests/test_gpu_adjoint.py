"""GPU suite: the REVERSE sweep of the row-varying tau / nu / drift lanes (csrc/k_iso_adj.hip, csrc/ssde_adj.hpp) -- one wave per
(64-track group, time window), a forward pass that leaves checkpoints, blocks of rows forwards again and then backwards --
against the oracle, against the forward-tangent kernels (k_iso_colvar.hip / k_iso_onewave.hip, SSDE_CV_ADJ=0) on the same
batches, and its own hand-over check (the adjoint at a window boundary, from both sides).
Reference: nllk_ctcrw.hpp:143-156, 203-241; nllk_ou_ssm.hpp:113-124, 171-207; nllk_bm_ssm.hpp:98-108, 135-169; the gradient is
TMB's reverse sweep over the tape of that loop (R/sde.R:656-658).

Tolerances (fp64): value 1e-10 * max(1,|v|); gradient 1e-8 * max|g| + 1e-10 (north-star bar: 1e-8)."""
import numpy as np
import pytest

from smoothsde_amd import capi
from smoothsde_amd.synth import bspline_basis, second_difference_penalty
from test_gpu_colvar import _batch, _close, _is_colvar, _oracle

pytestmark = pytest.mark.gpu
K_COLVAR, K_FEW, K_ADJ = 11, 12, 17


@pytest.fixture(autouse=True)
def _take_the_lane_track_path_from_32_tracks(monkeypatch):
    monkeypatch.setenv("SSDE_DRIFT_MIN_TRACKS", "32")
    monkeypatch.setenv("SSDE_CV_ADJ", "2")                      # the few-column shapes too


SHAPES = [("CTCRW", 2, 9, 9, False, True), ("CTCRW", 2, 9, 9, False, False), ("CTCRW", 1, 5, 0, True, False), ("CTCRW", 2, 0, 7, False, False),
          ("OU_SSM", 1, 9, 6, False, False), ("OU_SSM", 2, 4, 12, True, False), ("BM_SSM", 2, 8, 0, False, False),
          ("BM_SSM", 1, 5, 0, True, False), ("CTCRW", 2, 0, 0, True, False), ("OU_SSM", 2, 7, 7, False, True)]


@pytest.mark.parametrize("model,d,k1,k2,fe,same", SHAPES)
def test_reverse_sweep_vs_oracle_and_forward_tangents(model, d, k1, k2, fe, same, monkeypatch):
    """Several verified windows per track; the same batch on the forward-tangent kernels must give the same numbers."""
    pb, par = _batch(model, d, 96, 1500, k1, k2, seed=13, fe_slope=fe, same_basis=same)
    eng = capi.Engine(pb)
    assert _is_colvar(eng)
    val, grad = eng.eval(par)
    inf = eng.info()
    assert inf["kernel_id"] == K_ADJ
    assert inf["lanes_per_track"] > 1 and inf["window"] > 0 and inf["window_check"] <= 1e-11
    _close(val, grad, *_oracle(pb, par))
    # bitwise repeatable once the plan is (the first evaluation plans its windows from the design's bound on the predictors,
    # the later ones from the range the previous launch saw: a different number of windows, the same numbers to ~1e-16)
    eng.forget()
    v2, g2 = eng.eval(par)
    eng.forget()
    v3, g3 = eng.eval(par)
    assert v3 == v2 and np.array_equal(g3, g2)
    assert abs(v2 - val) <= 1e-13 * max(1.0, abs(val)) and np.max(np.abs(g2 - grad)) <= 1e-12 * np.max(np.abs(grad))
    assert abs(eng.eval(par, order=0) - val) <= 1e-12 * max(1.0, abs(val))
    monkeypatch.setenv("SSDE_CV_ADJ", "0")
    fwd = capi.Engine(pb)
    vf, gf = fwd.eval(par)
    assert fwd.info()["kernel_id"] in (K_COLVAR, K_FEW)
    assert abs(vf - val) <= 1e-11 * max(1.0, abs(val)) and np.max(np.abs(gf - grad)) <= 1e-9 * np.max(np.abs(grad)) + 1e-11
    eng.close(); fwd.close()


@pytest.mark.parametrize("model,d,k1,k2", [("OU_SSM", 1, 6, 4), ("CTCRW", 2, 5, 4), ("BM_SSM", 2, 7, 0), ("CTCRW", 1, 9, 0)])
@pytest.mark.parametrize("what", ["missing", "irregular", "both", "ragged"])
def test_missing_rows_irregular_grids_and_ragged_tracks(model, d, k1, k2, what):
    """A row whose first column is NA is a prediction step (nllk_ctcrw.hpp:214-217): its adjoint has no update half; the
    interval is the row's own; a lane whose track ends inside a window starts its backward recursion there."""
    pb, par = _batch(model, d, 96, 700, k1, k2, seed=21, ragged=(what == "ragged"))
    o, t = pb.obs.copy(), pb.times.copy()
    rng = np.random.default_rng(4)
    if what in ("missing", "both"):
        na = rng.random(len(t)) < 0.05
        na[pb.seg_start] = False
        o[na, 0] = np.nan
        o[na & (rng.random(len(t)) < 0.5)] = np.nan
    if what in ("irregular", "both"):
        t = np.cumsum(rng.uniform(0.4, 1.6, len(t)))
    pb2 = capi.Problem(model, pb.id, t, o, X_fe=pb.X_fe, X_re=pb.X_re, S_list=pb.S_list)
    eng = capi.Engine(pb2)
    assert _is_colvar(eng)
    val, grad = eng.eval(par)
    inf = eng.info()
    assert inf["kernel_id"] == K_ADJ and inf["window_check"] <= 1e-11
    _close(val, grad, *_oracle(pb2, par))
    eng.close()


def test_fixed_parameters_are_left_out():
    pb, par = _batch("CTCRW", 2, 150, 400, 7, 5, seed=5, ragged=True, dt=0.25)
    fixed = np.zeros(pb.n_par_full, dtype=np.uint8)
    fixed[[0, pb.off_fe + pb.fe_off[0], pb.off_fe + pb.fe_off[3], pb.off_re + 3]] = 1
    pb = capi.Problem("CTCRW", pb.id, pb.times, pb.obs, X_fe=pb.X_fe, X_re=pb.X_re, S_list=pb.S_list, par_fixed=fixed)
    eng = capi.Engine(pb)
    val, grad = eng.eval(par)
    assert eng.info()["kernel_id"] == K_ADJ
    _close(val, grad, *_oracle(pb, par))
    assert grad[0] == 0.0 and grad[pb.off_re + 3] == 0.0 and grad[pb.off_fe + pb.fe_off[0]] == 0.0
    eng.close()


def _mixed(model, d, M, T, seed):
    """mu_a smooth AND tau smooth (mixed design): the drift's columns are kinds of their own."""
    pb, par = _batch(model, d, M, T, 5, 0, seed=seed)
    n = pb.n
    x = np.clip(0.5 + 0.4 * np.cos(np.arange(n) * 2 * np.pi / 53), 0, 1)
    q = capi.n_sde_par(model, d)
    X_re, S = [None] * q, []
    X_re[0] = bspline_basis(x, 4); S.append(second_difference_penalty(4))
    X_re[d] = pb.X_re[d]; S.append(second_difference_penalty(5))
    pbm = capi.Problem(model, pb.id, pb.times, pb.obs, X_re=X_re, S_list=S)
    rng = np.random.default_rng(seed + 7)
    parm = 0.1 * rng.standard_normal(pbm.n_par_full)
    parm[0] = np.log(0.12)
    parm[pbm.off_fe + pbm.fe_off[d]] = np.log(2.0 if model != "BM_SSM" else 0.7)
    return pbm, parm


@pytest.mark.parametrize("model,d", [("CTCRW", 2), ("OU_SSM", 1), ("BM_SSM", 2), ("CTCRW", 1), ("OU_SSM", 2)])
def test_drift_columns_next_to_those_of_tau(model, d):
    pb, par = _mixed(model, d, 96, 900, seed=71)
    eng = capi.Engine(pb)
    assert _is_colvar(eng)
    val, grad = eng.eval(par)
    inf = eng.info()
    assert inf["kernel_id"] == K_ADJ and inf["window_check"] <= 1e-11
    _close(val, grad, *_oracle(pb, par))
    eng.close()


def test_short_backward_tail_is_detected_and_repaired(monkeypatch):
    """The adjoint forgets like the filter: a window that is not the last walks `window` rows past its end and starts its backward
    recursion there from zero.  With a deliberately useless tail (4 rows; the forward warm-up is left alone) the adjoint a window
    arrives with at a boundary disagrees with the one the next window computed -- the hand-over check must notice and the retry
    must repair it; never returned silently."""
    pb, par = _batch("CTCRW", 2, 96, 1500, 6, 6, seed=17)
    par[0] = 0.3                                                # sigma_obs = 1.35: slow forgetting
    monkeypatch.setenv("SSDE_ADJ_TAIL", "4")
    eng = capi.Engine(pb)
    val, grad = eng.eval(par)
    inf = eng.info()
    assert inf["kernel_id"] == K_ADJ
    assert inf["window_retries"] >= 1
    assert inf["window_check"] <= capi.WINDOW_TOL or inf["lanes_per_track"] == 1
    _close(val, grad, *_oracle(pb, par))
    eng.close()


def test_one_response_column_with_h_array():
    """d = 1 with H_array: the row's measurement variance is a tile channel; no log sigma_obs direction."""
    pb, par = _batch("OU_SSM", 1, 96, 800, 6, 5, seed=33)
    H = (0.01 + 0.02 * np.random.default_rng(3).random(pb.n)).reshape(1, 1, -1)
    pb2 = capi.Problem("OU_SSM", pb.id, pb.times, pb.obs, X_fe=pb.X_fe, X_re=pb.X_re, S_list=pb.S_list, H=H)
    eng = capi.Engine(pb2)
    assert _is_colvar(eng)
    val, grad = eng.eval(par)
    assert eng.info()["kernel_id"] == K_ADJ
    _close(val, grad, *_oracle(pb2, par))
    eng.close()


@pytest.mark.parametrize("model,k1,k2,what", [("CTCRW", 9, 9, "plain"), ("CTCRW", 5, 0, "missing"), ("CTCRW", 0, 7, "irregular"), ("CTCRW", 6, 6, "both"),
                                              ("OU_SSM", 6, 5, "plain"), ("BM_SSM", 7, 0, "missing"), ("OU_SSM", 4, 4, "both")])
def test_error_ellipses_on_every_fix_full_covariance_lanes(model, k1, k2, what):
    """Two response columns with a per-row measurement covariance (H_array, nllk_ctcrw.hpp:203-205): the dimensions couple, the
    covariance is a full 4 x 4 (CTCRW) / 2 x 2 (OU_SSM, BM_SSM) and the reverse sweep runs AdjFull (ssde_adj.hpp) -- 14 / 5 doubles of
    state and of adjoint; missing rows, irregular grids, a general P0; the drift intercepts free."""
    from test_gpu_colvar import _with_h
    pb, par = _batch(model, 2, 96, 900, k1, k2, seed=61, same_basis=(what == "plain"))
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
        sd = 4 if model == "CTCRW" else 2
        A = rng.standard_normal((sd, sd))
        P0 = A @ A.T + np.eye(sd)
    pb2 = capi.Problem(model, pb.id, t, o, X_fe=pb.X_fe, X_re=pb.X_re, S_list=pb.S_list, H=_with_h(pb, 7), P0=P0)
    eng = capi.Engine(pb2)
    assert _is_colvar(eng)
    val, grad = eng.eval(par)
    inf = eng.info()
    assert inf["kernel_id"] == K_ADJ and inf["lanes_per_track"] > 1 and inf["window_check"] <= 1e-11
    _close(val, grad, *_oracle(pb2, par))
    assert grad[0] == 0.0                                        # log_sigma_obs is not in the model
    assert abs(eng.eval(par, order=0) - val) <= 1e-12 * max(1.0, abs(val))
    eng.close()


_ALO, _AHI = (int(v) for v in __import__("os").environ.get("SSDE_FUZZ_ADJ_SEEDS", "0:12").split(":"))      # a one-off hunt widens the range


@pytest.mark.parametrize("seed", range(_ALO, _AHI))
def test_fuzz_against_the_oracle(seed):
    rng = np.random.default_rng(7000 + seed)
    model = ["CTCRW", "OU_SSM", "BM_SSM"][rng.integers(3)]
    d = int(rng.integers(1, 3))
    k1 = int(rng.choice([0, 3, 4, 6, 9, 10]))
    k2 = int(rng.choice([0, 3, 5, 8])) if model != "BM_SSM" else 0       # (at most 18 streamed columns: the widest instantiation)
    if k1 == 0 and k2 == 0:
        k1 = 3
    fe = bool(rng.integers(2)) and k1 + k2 < 17
    pb, par = _batch(model, d, int(rng.integers(33, 140)), int(rng.integers(60, 900)), k1, k2, seed=8000 + seed, fe_slope=fe,
                     ragged=bool(rng.integers(2)), dt=float(rng.choice([0.25, 1.0, 3.0])), same_basis=bool(rng.integers(2)))
    o, t = pb.obs.copy(), pb.times.copy()
    if rng.integers(2):
        na = rng.random(len(t)) < 0.03
        na[pb.seg_start] = False
        o[na, 0] = np.nan
    if rng.integers(2):
        t = np.cumsum(rng.uniform(0.5, 1.5, len(t)))
    fixed = (rng.random(pb.n_par_full) < 0.15).astype(np.uint8)
    fixed[pb.off_fe:pb.off_fe + d] |= np.uint8(rng.integers(2))
    pb2 = capi.Problem(model, pb.id, t, o, X_fe=pb.X_fe, X_re=pb.X_re, S_list=pb.S_list, par_fixed=fixed)
    par = par + 0.1 * rng.standard_normal(len(par))
    eng = capi.Engine(pb2)
    assert _is_colvar(eng)
    val, grad = eng.eval(par)
    inf = eng.info()
    assert inf["kernel_id"] == K_ADJ and inf["window_check"] <= 1e-11
    _close(val, grad, *_oracle(pb2, par))
    assert np.all(grad[fixed.astype(bool)] == 0.0)
    eng.close()


_HLO, _HHI = (int(v) for v in __import__("os").environ.get("SSDE_FUZZ_ADJ_H_SEEDS", "0:8").split(":"))


@pytest.mark.parametrize("seed", range(_HLO, _HHI))
def test_fuzz_with_error_ellipses_against_the_oracle_in_arbiter_mode(seed, monkeypatch):
    """row-varying tau / nu with a per-row measurement covariance (one response column: a variance; two: an ellipse, with or without
    entries between the columns), both gradient forms.  The oracle runs in ARBITER mode (P kept symmetric): where H couples the columns
    of a CTCRW the literal recursion drifts away within a few hundred rows; everywhere else the two modes are the same number."""
    from oracle_lib import keep_P_symmetric, oracle_eval
    rng = np.random.default_rng(9000 + seed)
    model = ["CTCRW", "OU_SSM", "BM_SSM"][rng.integers(3)]
    d = int(rng.integers(1, 3))
    k1 = int(rng.choice([3, 4, 6, 9]))
    k2 = int(rng.choice([0, 3, 5])) if model != "BM_SSM" else 0
    pb, par = _batch(model, d, int(rng.integers(33, 100)), int(rng.integers(100, 700)), k1, k2, seed=9500 + seed, ragged=bool(rng.integers(2)),
                     same_basis=bool(rng.integers(2)))
    n = len(pb.times)
    A = 0.1 * rng.standard_normal((n, d, d))
    Hn = np.einsum("nij,nkj->nik", A, A) + 0.01 * np.eye(d)
    if d == 2 and rng.integers(2):
        Hn = Hn * np.eye(2)                                        # independent axis errors
    H = np.ascontiguousarray(np.transpose(Hn, (1, 2, 0)))
    o = pb.obs.copy()
    if rng.integers(2):
        na = rng.random(n) < 0.03
        na[pb.seg_start] = False
        o[na, 0] = np.nan
    fixed = np.zeros(pb.n_par_full, dtype=np.uint8)
    fixed[0] = 1
    fixed[pb.off_fe:pb.off_fe + d] = np.uint8(rng.integers(2))
    pb2 = capi.Problem(model, pb.id, pb.times, o, X_fe=pb.X_fe, X_re=pb.X_re, S_list=pb.S_list, par_fixed=fixed, H=H)
    par = par + 0.05 * rng.standard_normal(len(par))
    keep_P_symmetric(True)
    try:
        oval, ograd = oracle_eval(pb2, par, order=1, threads=8)
    finally:
        keep_P_symmetric(False)
    for mode in ("2", "0"):
        monkeypatch.setenv("SSDE_CV_ADJ", mode)
        eng = capi.Engine(pb2)
        val, grad = eng.eval(par)
        inf = eng.info()
        ctx = (model, d, k1, k2, n, capi.KERNEL_NAMES.get(inf["kernel_id"]), inf["lanes_per_track"])
        assert inf["window_check"] <= 1e-11, ctx
        assert abs(val - oval) <= 1e-10 * max(1.0, abs(oval)), (ctx, val, oval)
        assert np.max(np.abs(grad - ograd)) <= 1e-8 * np.max(np.abs(ograd)) + 1e-10, (ctx, grad, ograd)
        eng.close()


def test_queued_asynchronous_evaluations_of_the_reverse_sweep():
    """ssde_eval_device (what a sharded host uses): eight evaluations at different parameter vectors queued on one stream without a
    synchronisation between them, each into its own HBM buffer -- every one must be the number the synchronous call returns (the
    checkpoints, range words and partial sums of the launches are reused from one evaluation to the next; the parameter slots are a ring)"""
    import torch
    pb, par = _batch("CTCRW", 2, 128, 800, 6, 5, seed=21)
    eng = capi.Engine(pb)
    eng.eval(par)                                                  # (plans the windows from the measured ranges once)
    pars = [par + 0.02 * np.sin(k + np.arange(len(par))) for k in range(8)]
    outs = torch.zeros(8, 2 + pb.n_par_full, dtype=torch.float64, device="cuda:0")
    st = torch.cuda.Stream()
    for k in range(8):
        eng.eval_device(pars[k], outs[k].data_ptr(), order=1, stream=st.cuda_stream)
    st.synchronize()
    host = outs.cpu().numpy()
    assert eng.info()["kernel_id"] == K_ADJ
    for k in range(8):
        assert host[k, -1] <= capi.WINDOW_TOL
        v, g = eng.eval(pars[k])
        pen, pg = eng.penalty(pars[k])
        assert abs(host[k, 0] + pen - v) <= 1e-12 * max(1.0, abs(v)), (k, host[k, 0] + pen, v)
        assert np.max(np.abs(host[k, 1:-1] + pg - g)) <= 1e-11 * np.max(np.abs(g)), k
    eng.close()
