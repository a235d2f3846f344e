"""numpy restatement of ssde_simulate (smoothsde_amd/csrc/k_sim.hip) -- test infrastructure.

Same arithmetic as the kernel: Philox4x32-10 counters (row, track_lo, track_hi, stream) keyed by the seed, two 53-bit
uniforms, Box-Muller, then the reference's exact transitions (/root/reference/R/sde.R:1434-1478, CTCRW_cov of
/root/reference/R/utility.R:188-196) and the observation error of the state-space families.  Agreement with the
device is to the last bits of log / sincos (the tests assert 1e-9 absolute on O(100)-row tracks)."""
import math

import numpy as np

M0, M1, W0, W1 = 0xD2511F53, 0xCD9E8D57, 0x9E3779B9, 0xBB67AE85
MASK = np.uint64(0xFFFFFFFF)


def philox4x32_10(c0, c1, c2, c3, k0, k1):
    """Vectorised over numpy arrays of uint64 holding 32-bit values."""
    c0, c1, c2, c3 = (np.asarray(c, dtype=np.uint64) & MASK for c in (c0, c1, c2, c3))
    c0, c1, c2, c3 = np.broadcast_arrays(c0, c1, c2, c3)
    k0, k1 = np.uint64(k0 & 0xFFFFFFFF), np.uint64(k1 & 0xFFFFFFFF)
    for _ in range(10):
        p0, p1 = np.uint64(M0) * c0, np.uint64(M1) * c2
        n0 = ((p1 >> np.uint64(32)) ^ c1 ^ k0) & MASK
        n1 = p1 & MASK
        n2 = ((p0 >> np.uint64(32)) ^ c3 ^ k1) & MASK
        n3 = p0 & MASK
        c0, c1, c2, c3 = n0, n1, n2, n3
        k0 = (k0 + np.uint64(W0)) & MASK
        k1 = (k1 + np.uint64(W1)) & MASK
    return c0, c1, c2, c3


def normal_pair(seed, track, row, stream):
    track = np.asarray(track, dtype=np.uint64)
    o0, o1, o2, o3 = philox4x32_10(np.asarray(row, dtype=np.uint64), track & MASK, track >> np.uint64(32),
                                   np.asarray(stream, dtype=np.uint64), seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF)
    u1 = ((((o0 << np.uint64(32)) | o1) >> np.uint64(11)).astype(np.float64) + 0.5) * 2.0 ** -53
    u2 = ((((o2 << np.uint64(32)) | o3) >> np.uint64(11)).astype(np.float64) + 0.5) * 2.0 ** -53
    r = np.sqrt(-2.0 * np.log(u1))
    return r * np.cos(2.0 * np.pi * u2), r * np.sin(2.0 * np.pi * u2)


def simulate_ref(model, n_tracks, n_steps, n_dim=2, *, mu=0.0, tau=2.0, nu=1.0, kappa=1.0, sigma=1.0, sigma_obs=0.1,
                 dt=1.0, z0=0.0, seed=1, track0=0, lengths=None):
    """(ID, times, obs) of tracks [track0, track0 + n_tracks) as numpy arrays, obs of shape (n, d)."""
    M, T, d = int(n_tracks), int(n_steps), int(n_dim)
    mu = np.broadcast_to(np.asarray(mu, dtype=np.float64), (d,))
    z0 = np.broadcast_to(np.asarray(z0, dtype=np.float64), (d,))
    with_error = model in ("CTCRW", "OU_SSM", "BM_SSM")
    so = sigma_obs if with_error else 0.0
    track = np.arange(track0, track0 + M, dtype=np.uint64)
    out = np.empty((M, T, d))
    for a in range(d):
        z = np.full(M, z0[a])
        v = np.zeros(M)
        if model == "CTCRW":
            beta, sig = 1.0 / tau, 2.0 * nu / math.sqrt(tau * math.pi)
            e, e2 = math.exp(-beta * dt), math.exp(-2.0 * beta * dt)
            qvv = sig * sig / (2.0 * beta) * (1.0 - e2)
            qzz = (sig / beta) * (sig / beta) * (dt + (1.0 - e2) / (2.0 * beta) - 2.0 * (1.0 - e) / beta)
            qvz = sig * sig / (2.0 * beta * beta) * (1.0 - 2.0 * e + e2)
            l11 = math.sqrt(qvv)
            l21 = qvz / l11
            l22 = math.sqrt(max(qzz - l21 * l21, 0.0))
            ib1e = (1.0 - e) / beta
        elif model in ("OU", "OU_SSM"):
            e = math.exp(-dt / tau)
            sd = math.sqrt(kappa * (1.0 - math.exp(-2.0 * dt / tau)))
        elif model in ("BM", "BM_SSM"):
            sd = sigma * math.sqrt(dt)
        else:
            raise ValueError(model)
        for t in range(T):
            if model == "CTCRW":
                no = normal_pair(seed, track, t, 2 * a + 1)[0] if so > 0 else 0.0
                if t > 0:
                    na, nb = normal_pair(seed, track, t, 2 * a)
                    z = z + mu[a] * dt + (v - mu[a]) * ib1e + l21 * na + l22 * nb
                    v = e * v + (1.0 - e) * mu[a] + l11 * na
            else:
                na, no = normal_pair(seed, track, t, 2 * a)
                if t > 0:
                    z = e * z + (1.0 - e) * mu[a] + sd * na if model in ("OU", "OU_SSM") else z + mu[a] * dt + sd * na
            out[:, t, a] = z + so * no
    if lengths is None:
        n = M * T
        obs = out.reshape(n, d)
        ID = np.repeat(np.arange(track0, track0 + M, dtype=np.float64), T)
    else:
        lengths = np.asarray(lengths, dtype=np.int64)
        obs = np.concatenate([out[m, :lengths[m]] for m in range(M)], axis=0)
        ID = np.repeat(np.arange(track0, track0 + M, dtype=np.float64), lengths)
        n = len(ID)
    return ID, obs, n
