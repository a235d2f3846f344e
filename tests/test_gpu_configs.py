"""GPU suite: the BASELINE.json configurations the other files do not run at their named sizes.

* C2  -- 10^4 CTCRW tracks x 10^3 rows, constant coefficients, regular grid;
* C4  -- one GPU's shard of 10^5 tracks x 10^4 rows over 8 GPUs: 12 500 tracks x 10^4 rows;
* C5  -- the mixed batch as specified: BM_SSM, OU_SSM and CTCRW handles with observation error, ragged lengths
         T ~ U[0.5 T, T], 5 % missing rows (half of them NA in column 0 only, half in every column: the reference tests
         column 0, nllk_ctcrw.hpp:214), one GPU's share (3 750 tracks per model), the three handles evaluated
         CONCURRENTLY on one GPU (three streams).
Parity is asserted here through size-independent properties (additivity over track shards, gradient = derivative of the
value, determinism) plus a random sample of whole tracks against the oracle at the usual tolerance (value 1e-10, gradient
1e-8); the WHOLE batches of C2 (10^7 rows) and of the headline configuration (10^8 rows) against the oracle -- seconds to a
minute of the literal oracle on the box's 16 threads -- are tests/test_gpu_whole_batch.py."""
import numpy as np
import pytest

from smoothsde_amd import capi
from smoothsde_amd.synth import simulate
from test_gpu_parity import _close, _full_size_check, _oracle

pytestmark = pytest.mark.gpu


def test_config_c2_ten_thousand_tracks_of_one_thousand_rows():
    info = _full_size_check("CTCRW", [np.log(0.1), 0.0, 0.0, np.log(2.0), 0.0], [0, 1, 1, 0, 0], 0.0,
                            dict(mu=0.0, tau=2.0, nu=1.0, sigma_obs=0.1), M=10_000, T=1_000)
    assert info["n_rows"] == 10_000_000 and info["uniform_dt"] == 1 and info["required_bytes_per_row"] == 16.0
    # ... and at theta_0 = (0, 0, 0), the other point SURVEY 8(d) names
    _full_size_check("CTCRW", [0.0, 0.0, 0.0, 0.0, 0.0], [0, 1, 1, 0, 0], 0.0,
                     dict(mu=0.0, tau=2.0, nu=1.0, sigma_obs=0.1), M=2_000, T=1_000)


def test_config_c4_one_gpu_shard():
    info = _full_size_check("CTCRW", [np.log(0.1), 0.0, 0.0, np.log(2.0), 0.0], [0, 1, 1, 0, 0], 0.0,
                            dict(mu=0.0, tau=2.0, nu=1.0, sigma_obs=0.1), M=12_500, T=10_000)
    assert info["n_rows"] == 125_000_000 and info["n_tracks"] == 12_500


def _c5_batch(model, M, T, seed, sim_kw):
    """ragged lengths U[0.5 T, T], 5 % missing rows: NA in column 0 only or in both columns"""
    import torch
    dev = "cuda:0"
    ID, times, obs = simulate(model, M, T, 2, seed=seed, backend="torch", device=dev, **sim_kw)
    gen = torch.Generator(device=dev)
    gen.manual_seed(100 + seed)
    lens = torch.randint(T // 2, T + 1, (M,), device=dev, generator=gen)
    pos = torch.arange(T, device=dev).repeat(M)
    keep = pos < lens.repeat_interleave(T)
    u = torch.rand(M * T, device=dev, generator=gen)
    obs[(u < 0.025) & (pos > 0), 0] = float("nan")                       # column 0 only
    both = (u >= 0.025) & (u < 0.05) & (pos > 0)
    obs[both] = float("nan")                                             # every column
    ID, obs = ID[keep].contiguous(), obs[keep].contiguous()
    times = torch.arange(1, ID.numel() + 1, dtype=torch.float64, device=dev)
    return ID, times, obs, lens.cpu().numpy()


def test_config_c5_mixed_batch_three_handles_concurrently():
    import torch
    M, T = 3_750, 10_000
    specs = [("BM_SSM", [np.log(0.1), 0.1, 0.1, 0.0], dict(mu=0.1, sigma=1.0, sigma_obs=0.1)),
             ("OU_SSM", [np.log(0.1), 5.0, -5.0, np.log(2.0), 0.0], dict(mu=[5.0, -5.0], tau=2.0, kappa=1.0, sigma_obs=0.1)),
             ("CTCRW", [np.log(0.1), 0.0, 0.0, np.log(2.0), 0.0], dict(mu=0.0, tau=2.0, nu=1.0, sigma_obs=0.1))]
    engines, data = [], []
    for k, (model, par, kw) in enumerate(specs):
        ID, times, obs, lens = _c5_batch(model, M, T, 4 + k, kw)
        data.append((ID, times, obs, lens))
        engines.append(capi.Engine(capi.Problem.from_torch(model, ID, times, obs)))
    # sequential, synchronous: the reference values of this test
    seq = [e.eval(np.asarray(p, dtype=float)) for e, (_, p, _) in zip(engines, specs)]
    for e in engines:
        assert e.info()["window_check"] <= capi.WINDOW_TOL
    # the three handles concurrently: one stream each, nothing synchronises until all three are enqueued
    streams = [torch.cuda.Stream() for _ in engines]
    outs = [torch.zeros(2 + e.n_par_full, dtype=torch.float64, device="cuda:0") for e in engines]
    for rep in range(2):
        for e, s, o, (_, p, _) in zip(engines, streams, outs, specs):
            e.eval_device(np.asarray(p, dtype=float), o.data_ptr(), order=1, stream=s.cuda_stream)
        torch.cuda.synchronize()
        for e, o, (v, g), (_, p, _) in zip(engines, outs, seq, specs):
            r = o.cpu().numpy()
            pv, pg = e.penalty(np.asarray(p, dtype=float))
            assert r[-1] <= capi.WINDOW_TOL
            assert r[0] + pv == v and np.array_equal(r[1:-1] + pg, g)       # concurrency changes nothing, bitwise
    rng = np.random.default_rng(5)
    for e, (ID, times, obs, lens), (v, g), (model, par, _) in zip(engines, data, seq, specs):
        par = np.asarray(par, dtype=float)
        n = ID.numel()
        assert e.info()["n_rows"] == n and e.info()["n_tracks"] == M
        # gradient = derivative of the value
        dirn = rng.standard_normal(len(par))
        h = 1e-5
        fd = (e.eval(par + h * dirn, order=0) - e.eval(par - h * dirn, order=0)) / (2 * h)
        assert abs(fd - g @ dirn) <= 1e-6 * np.max(np.abs(g)), (model, fd, g @ dirn)
        # additivity over two track shards
        starts = np.concatenate([[0], np.cumsum(lens)])
        cut = int(starts[1400])
        va, ga = 0.0, np.zeros_like(g)
        for sl in (slice(0, cut), slice(cut, n)):
            es = capi.Engine(capi.Problem.from_torch(model, ID[sl], times[sl], obs[sl]))
            vs, gs = es.eval(par)
            va, ga = va + vs, ga + gs
            es.close()
        assert abs(v - va) <= 1e-12 * abs(v) and np.max(np.abs(g - ga)) <= 1e-9 * np.max(np.abs(g)), model
        # a random sample of whole tracks against the oracle
        pick = np.sort(rng.choice(M, size=12, replace=False))
        rows = torch.cat([torch.arange(int(starts[k]), int(starts[k + 1]), device=ID.device) for k in pick])
        pbh = capi.Problem(model, ID[rows].cpu().numpy(), times[rows].cpu().numpy(), obs[rows].cpu().numpy())
        eh = capi.Engine(pbh)
        vh, gh = eh.eval(par)
        ov, og = _oracle(pbh, par)
        _close(vh, gh, ov, og)
        eh.close()
    for e in engines:
        e.close()
