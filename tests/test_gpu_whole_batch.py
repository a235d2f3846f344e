"""GPU suite: the WHOLE batch that bench.py times, against the oracle -- value and gradient of the very launch, not of a sample.

The other full-size tests (test_gpu_parity.py::_full_size_check, test_gpu_configs.py) check size-independent properties and a
sample of tracks evaluated as an engine of their own, i.e. under a different window plan.  The literal oracle (oracle/ssde_oracle.hpp,
threads over tracks) walks ~2 10^6 rows a second on the GPU box's 16 cores: 5 s for BASELINE config 2 (10^7 rows), ~50 s for the
headline batch (10^8 rows) -- affordable once per run, so the launch that is timed is also the launch that is compared.
The batches are built exactly as bench.py builds them (same simulator, seed and parameter vectors).
Reference: nllk_ctcrw.hpp:195-247 (the loop), :143-156 (the predictor); nllk_sde.hpp:61-84 + tr_dens.hpp:45-52 (config 3).

Tolerances (fp64): value 1e-10 * |v|; gradient 1e-8 * max|g| (north-star bar: 1e-8)."""
import os
import time

import numpy as np
import pytest

from smoothsde_amd import capi

pytestmark = pytest.mark.gpu
K_ISO_SHARED, K_DIRECT_FAST = 3, 2
THREADS = min(16, os.cpu_count() or 8)          # (the GPU box's CPU share for one GPU)


def _oracle(pb, par):
    from oracle_lib import oracle_eval
    t0 = time.perf_counter()
    out = oracle_eval(pb, np.asarray(par, dtype=float), order=1, threads=THREADS)
    return out, time.perf_counter() - t0


def _bench_batch(M, T):
    """bench.py: build_handles(), configuration c2p (its tracks / rows overridden for config 2)"""
    import torch
    import bench
    dev = torch.device("cuda:0")
    ID, times, obs = capi.simulate_device("CTCRW", M, T, 2, mu=0.0, tau=2.0, nu=1.0, kappa=1.0, sigma=1.0, sigma_obs=0.1, seed=1,
                                          track0=0, device=dev)
    fixed = np.zeros(5, dtype=np.uint8)
    fixed[1:3] = 1
    pb = capi.Problem.from_torch("CTCRW", ID, times, obs, par_fixed=fixed)
    eng = capi.Engine(pb)
    host = capi.Problem("CTCRW", ID.cpu().numpy(), times.cpu().numpy(), obs.cpu().numpy(), par_fixed=fixed)
    del ID, times, obs
    return eng, host, bench.theta_for(5, 2, 4, 0)


def _compare(eng, host, theta, kernel_id, rows):
    val, grad = eng.eval(theta)
    inf = eng.info()
    assert inf["n_rows"] == rows and inf["kernel_id"] == kernel_id and inf["window_check"] <= capi.WINDOW_TOL
    (oval, ograd), secs = _oracle(host, theta)
    print("\noracle: %d rows in %.1f s on %d threads (%.2e rows/s)" % (rows, secs, THREADS, rows / secs))
    assert abs(val - oval) <= 1e-10 * abs(oval), (val, oval)
    assert np.max(np.abs(grad - ograd)) <= 1e-8 * np.max(np.abs(ograd)), (grad, ograd)
    assert np.all(grad[host.par_fixed != 0] == 0.0)
    return inf


def test_config_c2_whole_batch_vs_oracle():
    """BASELINE config 2: 10^4 CTCRW tracks x 10^3 rows, constant coefficients, one MI355X -- all 10^7 rows against the oracle."""
    eng, host, theta = _bench_batch(10_000, 1_000)
    _compare(eng, host, theta, K_ISO_SHARED, 10_000_000)
    eng.close()


def test_headline_whole_batch_vs_oracle():
    """BASELINE's metric configuration (SURVEY 8(d) C2'): 10^4 CTCRW tracks x 10^4 rows -- all 10^8 rows of the launch bench.py
    times, at the parameter vector of its first timed step."""
    eng, host, theta = _bench_batch(10_000, 10_000)
    inf = _compare(eng, host, theta, K_ISO_SHARED, 100_000_000)
    assert inf["uniform_dt"] == 1 and inf["lanes_per_track"] > 1          # the windowed launch, as timed
    eng.close()


def test_config_c3_whole_batch_vs_oracle():
    """BASELINE config 3: 10^4 OU tracks x 10^4 rows with a 9-column spline-varying drift streamed (88 B/row): all 10^8 rows (the
    direct path has no recursion: the oracle walks 2 10^7 rows a second)."""
    import torch
    from smoothsde_amd.synth import second_difference_penalty, simulate
    M, T, K = 10_000, 10_000, 9
    ID, times, obs = simulate("OU", M, T, 1, mu=1.0, tau=2.0, kappa=1.0, seed=2, backend="torch", device="cuda:0")
    n = ID.numel()
    x = torch.cumsum(torch.randn(n, device=ID.device, dtype=torch.float64) * 0.01, 0)
    x = (x - x.min()) / (x.max() - x.min())
    B = torch.stack([torch.cos((k + 1) * np.pi * x) for k in range(K)], dim=1)
    S = [second_difference_penalty(K)]
    par = np.concatenate([[1.0, np.log(2.0), 0.0], [0.3], 0.05 * np.sin(np.arange(K))])
    eng = capi.Engine(capi.Problem.from_torch("OU", ID, times, obs, X_re=[B, None, None], S_list=S))
    host = capi.Problem("OU", ID.cpu().numpy(), times.cpu().numpy(), obs.cpu().numpy(), X_re=[B.cpu().numpy(), None, None], S_list=S)
    val, grad = eng.eval(par)
    inf = eng.info()
    assert inf["n_rows"] == n and inf["kernel_id"] == K_DIRECT_FAST
    (oval, ograd), secs = _oracle(host, par)
    print("\noracle: %d rows in %.1f s (%.2e rows/s)" % (n, secs, n / secs))
    assert abs(val - oval) <= 1e-10 * abs(oval), (val, oval)
    assert np.max(np.abs(grad - ograd)) <= 1e-8 * np.max(np.abs(ograd)), (grad, ograd)
    eng.close()


def test_row_varying_whole_batch_vs_oracle():
    """the row-varying launch bench.py times as `secondary` (10^4 CTCRW tracks x 10^3 rows, tau and nu smooth in a covariate, 2 x 9
    design columns streamed; nllk_ctcrw.hpp:143-156 feeding :195-247): all 10^7 rows, value and all 21 free gradient entries of the
    reverse sweep (k_iso_adj.hip) against the oracle's forward duals, at the parameter vector of the first timed step"""
    import torch
    import bench
    dev = torch.device("cuda:0")
    M, T, K = 10_000, 1_000, 9
    ID, times, obs, B, S, fixed = bench.row_varying_batch(M, T, dev, K)
    eng = capi.Engine(capi.Problem.from_torch("CTCRW", ID, times, obs, X_re=[None, None, B, B], S_list=[S, S], par_fixed=fixed))
    Bh = B.cpu().numpy()
    host = capi.Problem("CTCRW", ID.cpu().numpy(), times.cpu().numpy(), obs.cpu().numpy(), X_re=[None, None, Bh, Bh], S_list=[S, S], par_fixed=fixed)
    del ID, times, obs, B
    theta = bench.row_varying_theta(0, K)
    val, grad = eng.eval(theta)
    inf = eng.info()
    assert inf["n_rows"] == M * T and inf["kernel_id"] == 17 and inf["lanes_per_track"] > 1 and inf["window_check"] <= capi.WINDOW_TOL, inf
    (oval, ograd), secs = _oracle(host, theta)
    print("\noracle: %d rows x 21 directions in %.1f s on %d threads" % (M * T, secs, THREADS))
    assert abs(val - oval) <= 1e-10 * abs(oval), (val, oval)
    assert np.max(np.abs(grad - ograd)) <= 1e-8 * np.max(np.abs(ograd)), (grad, ograd)
    assert np.all(grad[fixed != 0] == 0.0)
    eng.close()

