"""ssde_simulate (the HIP simulator, csrc/k_sim.hip) against its numpy restatement (tests/sim_ref.py), and the
restatement against the published known-answer vectors of Philox4x32-10.  Reference behaviour simulated:
/root/reference/R/sde.R:1434-1478 (exact transitions), /root/reference/R/utility.R:188-196 (CTCRW_cov)."""
import numpy as np
import pytest

from sim_ref import normal_pair, philox4x32_10, simulate_ref


def _kat(c, k):
    return [int(x) for x in philox4x32_10(*[np.uint64(v) for v in c], *k)]


def test_philox_known_answers():
    # Random123 kat_vectors, philox4x32 with 10 rounds
    assert _kat((0, 0, 0, 0), (0, 0)) == [0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8]
    assert _kat((0xffffffff,) * 4, (0xffffffff, 0xffffffff)) == [0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd]
    assert _kat((0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344), (0xa4093822, 0x299f31d0)) == [0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1]


def test_restated_normals_have_unit_moments():
    n1, n2 = normal_pair(11, np.arange(400_000), 5, 1)
    for x in (n1, n2):
        assert abs(x.mean()) < 6e-3 and abs(x.std() - 1.0) < 5e-3
        assert abs(np.mean(x ** 4) - 3.0) < 0.06
    assert abs(np.corrcoef(n1, n2)[0, 1]) < 6e-3


def test_restated_batch_does_not_depend_on_how_it_is_cut():
    _, whole, _ = simulate_ref("CTCRW", 6, 20, 2, seed=5)
    _, a, _ = simulate_ref("CTCRW", 2, 20, 2, seed=5, track0=0)
    _, b, _ = simulate_ref("CTCRW", 4, 20, 2, seed=5, track0=2)
    assert np.array_equal(whole, np.concatenate([a, b]))


def test_restated_increment_variances_match_the_transition_densities():
    # BM: Var(dz) = sigma^2 dt (R/sde.R:1437);  OU: stationary variance kappa (R/sde.R:1445)
    _, obs, _ = simulate_ref("BM", 400, 200, 1, mu=0.3, sigma=0.7, dt=0.5, seed=2)
    dz = np.diff(obs.reshape(400, 200), axis=1)
    assert abs(dz.mean() - 0.15) < 0.01 and abs(dz.var() - 0.245) < 0.01
    _, obs, _ = simulate_ref("OU", 400, 400, 1, mu=2.0, tau=3.0, kappa=1.5, seed=3, z0=2.0)
    z = obs.reshape(400, 400)[:, 100:]
    assert abs(z.mean() - 2.0) < 0.05 and abs(z.var() - 1.5) < 0.08


@pytest.mark.gpu
@pytest.mark.parametrize("model,d", [("CTCRW", 2), ("CTCRW", 1), ("OU_SSM", 2), ("BM_SSM", 1), ("OU", 1), ("BM", 2), ("CTCRW", 3)])
def test_device_batch_matches_the_restatement(model, d):
    import torch
    from smoothsde_amd import capi
    kw = dict(mu=[0.2, -0.1, 0.05][:d], tau=1.7, nu=0.8, kappa=1.3, sigma=0.9, sigma_obs=0.07, dt=0.5, z0=[3.0, -2.0, 1.0][:d], seed=1234567890123)
    ID, times, obs = capi.simulate_device(model, 150, 77, d, track0=40, **kw)
    rID, robs, n = simulate_ref(model, 150, 77, d, track0=40, **kw)
    torch.cuda.synchronize()
    assert obs.shape == (n, d)
    assert np.array_equal(ID.cpu().numpy(), rID)
    assert np.array_equal(times.cpu().numpy(), (40 * 77 + np.arange(1, n + 1)) * 0.5)
    assert np.max(np.abs(obs.cpu().numpy() - robs)) < 1e-9
    if model in ("OU", "BM"):      # the direct families carry no observation error: row 0 is z0 itself
        assert np.array_equal(obs.cpu().numpy()[::77], np.tile(kw["z0"], (150, 1)))


@pytest.mark.gpu
def test_device_shards_are_the_same_batch_and_ragged_tracks_are_prefixes():
    import torch
    from smoothsde_amd import capi
    _, _, whole = capi.simulate_device("CTCRW", 300, 64, 2, seed=9)
    _, _, a = capi.simulate_device("CTCRW", 100, 64, 2, seed=9, track0=0)
    _, tb, b = capi.simulate_device("CTCRW", 200, 64, 2, seed=9, track0=100)
    assert torch.equal(whole, torch.cat([a, b]))
    assert float(tb[0]) == 100 * 64 + 1                  # times continue where the first shard's end
    lengths = 1 + (np.arange(300) * 7) % 64
    ID, times, rag = capi.simulate_device("CTCRW", 300, 64, 2, seed=9, lengths=lengths)
    w = whole.cpu().numpy().reshape(300, 64, 2)
    ref = np.concatenate([w[m, :lengths[m]] for m in range(300)])
    assert np.array_equal(rag.cpu().numpy(), ref)
    assert np.array_equal(ID.cpu().numpy(), np.repeat(np.arange(300.0), lengths))
    assert np.array_equal(times.cpu().numpy(), np.arange(1, len(ref) + 1, dtype=np.float64))


@pytest.mark.gpu
def test_simulated_batch_is_what_the_engine_expects_and_the_truth_is_near_the_optimum():
    # a batch simulated at theta*: the per-row gradient of the nllk at theta* is O(1 / sqrt(rows)) (the score has mean zero)
    from smoothsde_amd import capi
    ID, times, obs = capi.simulate_device("CTCRW", 2000, 500, 2, tau=2.0, nu=1.0, sigma_obs=0.1, seed=3)
    eng = capi.Engine(capi.Problem.from_torch("CTCRW", ID, times, obs, par_fixed=[0, 1, 1, 0, 0]))
    par = np.array([np.log(0.1), 0.0, 0.0, np.log(2.0), 0.0])
    v, g = eng.eval(par)
    v2, _ = eng.eval(par + np.array([0.3, 0, 0, -0.4, 0.3]))
    inf = eng.info()
    eng.close()
    assert inf["uniform_dt"] == 1 and inf["n_tracks"] == 2000
    assert np.isfinite(v) and np.max(np.abs(g)) / (2000 * 499) < 5e-3
    assert v2 > v + 1000.0                # ... and a wrong theta is clearly worse


@pytest.mark.gpu
def test_simulate_refuses_what_the_reference_refuses():
    from smoothsde_amd import capi
    with pytest.raises(capi.EngineError, match="not implemented"):
        capi.simulate_device("CIR", 4, 8, 1)
    with pytest.raises(ValueError):
        capi.simulate_device("nope", 4, 8, 1)
    # parameters that would put a NaN / Inf Cholesky factor into HBM are refused, not simulated
    for model, kw in (("CTCRW", {"tau": 0.0}), ("CTCRW", {"nu": -1.0}), ("CTCRW", {"nu": 0.0}), ("OU", {"kappa": -0.5}),
                      ("OU_SSM", {"tau": -2.0}), ("BM", {"sigma": float("nan")}), ("BM_SSM", {"sigma_obs": -0.1})):
        with pytest.raises(capi.EngineError, match="are required"):
            capi.simulate_device(model, 4, 8, 1, **kw)
