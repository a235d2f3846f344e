"""Small seeded parity cases (SURVEY.md 8(c) fixture matrix).

A case is a plain dict of arrays (JSON-serialisable via tests/golden/gen_golden.py) plus
one parameter vector.  `problem_from_spec` turns it into a smoothsde_amd.capi.Problem.
"""
from __future__ import annotations

import numpy as np

from smoothsde_amd.capi import Problem, n_sde_par, state_dim, na_real
from smoothsde_amd.synth import bspline_basis, second_difference_penalty

# the six rows of the vignette's elephant track that survive offline
# (/root/reference/vignettes/smoothSDE.pdf, output of smoothSDE.rmd:470)
ELEPHANT_ROWS = np.array([
    [572.3427, 1675.424, 33, 9703],
    [572.5443, 1675.392, 32, 9704],
    [572.6159, 1675.339, 31, 9705],
    [572.7745, 1675.101, 31, 9706],
    [572.8844, 1675.065, 31, 9707],
    [573.6659, 1674.322, 30, 9708],
])


def _tracks(rng, model, d, lengths, irregular=True, scale=1.0):
    """Random-walk-ish observations; the likelihood is evaluated, not fitted, so any
    smooth-ish data will do."""
    ID, times, obs = [], [], []
    t0 = 0.0
    for k, T in enumerate(lengths):
        if isinstance(irregular, str) and irregular == "lattice":    # a schedule of 0.5 with fixes absent (runs of 1-2)
            dts = 0.5 * rng.choice([1, 1, 1, 1, 2, 3], size=T)
        else:
            dts = rng.uniform(0.3, 2.0, size=T) if irregular else np.ones(T)
        tt = t0 + np.cumsum(dts)
        t0 = tt[-1] + 5.0
        steps = rng.standard_normal((T, d)) * scale
        if model in ("CTCRW",):
            vel = np.cumsum(rng.standard_normal((T, d)) * 0.3, axis=0)
            z = np.cumsum(vel * dts[:, None], axis=0) + 0.1 * steps
        elif model in ("OU", "OU_SSM"):
            z = 3.0 + steps
        elif model == "CIR":
            z = np.exp(0.4 * np.cumsum(steps * np.sqrt(dts[:, None]) * 0.5, axis=0))     # positive
        else:
            z = np.cumsum(steps, axis=0)
        ID += [float(k)] * T
        times += list(tt)
        obs.append(z + (0.0 if model == "CIR" else 10.0 * k))
    return np.array(ID), np.array(times), np.vstack(obs)


def _par_const(rng, model, d, kalman):
    q = n_sde_par(model, d)
    p = []
    if kalman:
        p.append(rng.uniform(-1.5, 0.0))  # log_sigma_obs
    if model == "CIR":
        return np.array(list(rng.uniform(-0.3, 0.5, size=d)) + [rng.uniform(-1.2, 0.0), rng.uniform(-1.2, -0.3)])  # log mu, log beta, log sigma
    p += list(rng.uniform(-0.5, 0.5, size=d) + (3.0 if model in ("OU", "OU_SSM") else 0.0))  # mu
    p += list(rng.uniform(-0.3, 0.7, size=q - d))  # log-scale parameters
    return np.array(p)


def make_spec(name, model, d, *, seed, lengths, variant="const", na_rows=(), irregular=True,
              fix_mu=False, with_H=False, with_P0=False, na_mode=1, other_data=None, decay=False):
    rng = np.random.default_rng(seed)
    kalman = model in ("CTCRW", "OU_SSM", "BM_SSM")
    ID, times, obs = _tracks(rng, model, d, lengths, irregular)
    n = len(ID)
    q = n_sde_par(model, d)
    sdim = state_dim(model, d)
    spec = dict(name=name, model=model, n_dim=d, ID=ID, times=times, obs=obs, X_fe=None, X_re=None,
                S_list=None, a0=None, P0=None, H=None, par_fixed=None, na_mode=na_mode, include_penalty=1)
    if other_data is not None:
        spec["other_data"] = np.atleast_1d(np.asarray(other_data, dtype=np.float64))
    na = na_real() if na_mode == 0 else float("nan")
    for r in na_rows:
        if kalman:
            obs[r, :] = na
        else:
            obs[r, rng.integers(0, d)] = na
    if variant == "const":
        par = _par_const(rng, model, d, kalman)
    else:
        # time-varying: covariate x in [0,1]; parameter d (tau / sigma) gets intercept + linear x;
        # the last parameter (or mu_1 for BM) gets a 4-column spline block with a penalty;
        # the rest stay intercept-only.
        x = (np.sin(np.linspace(0, 7, n)) + 1) / 2 + 0.05 * rng.standard_normal(n)
        x = np.clip(x, 0, 1)
        X_fe = [None] * q
        X_re = [None] * q
        X_fe[d] = np.column_stack([np.ones(n), x])
        jr = q - 1 if model not in ("BM", "BM_SSM", "BM_t") else 0
        B = bspline_basis(x, n_basis=4)
        X_re[jr] = B
        S_list = [second_difference_penalty(4)]
        if variant == "tv2":  # a second smooth, on mu_0 (two penalty blocks, two parameters with RE)
            j2 = 0 if jr != 0 else d
            X_re[j2] = bspline_basis(np.clip(x ** 2, 0, 1), n_basis=5)
            S_list = ([second_difference_penalty(5)] + S_list) if j2 < jr else (S_list + [second_difference_penalty(5)])
        spec.update(X_fe=X_fe, X_re=X_re, S_list=S_list)
        ncol_fe = [1 if X_fe[j] is None else X_fe[j].shape[1] for j in range(q)]
        p = []
        if kalman:
            p.append(rng.uniform(-1.5, 0.0))
        for j in range(q):
            base = rng.uniform(-0.3, 0.5) + (3.0 if (j < d and model in ("OU", "OU_SSM")) else 0.0)
            p += [base] + list(rng.uniform(-0.4, 0.4, size=ncol_fe[j] - 1))
        p += list(rng.uniform(-0.5, 1.0, size=len(S_list)))  # log_lambda
        if decay:
            # decaying response (nllk_sde.hpp:47-58): every smooth decays with its own rate, except that the
            # last column of the first smooth is left alone (mixed blocks must work too)
            n_re_tot = sum(s.shape[0] for s in S_list)
            cols, inds, start = [], [], 0
            for k, s_ in enumerate(S_list):
                for c in range(s_.shape[0]):
                    if not (k == 0 and c == s_.shape[0] - 1):
                        cols.append(start + c); inds.append(k)
                start += s_.shape[0]
            spec.update(t_decay=rng.uniform(0.0, 2.5, size=q * n), col_decay=np.array(cols, dtype=np.int32),
                        ind_decay=np.array(inds, dtype=np.int32))
            p += list(rng.uniform(-1.0, 0.3, size=len(S_list)))  # log_decay
        p += list(rng.uniform(-0.3, 0.3, size=sum(s.shape[0] for s in S_list)))  # coeff_re
        par = np.array(p)
    if with_H:
        A = rng.standard_normal((n, d, d)) * 0.2
        H = np.einsum("nij,nkj->ikn", A, A) + 0.05 * np.eye(d)[:, :, None]
        spec["H"] = H
    if with_P0:
        A = rng.standard_normal((sdim, sdim))
        spec["P0"] = A @ A.T + np.eye(sdim)
    if fix_mu:
        pb = problem_from_spec(dict(spec, par=par))
        fixed = np.zeros(pb.n_par_full, dtype=np.uint8)
        for a in range(d):
            fixed[pb.off_fe + pb.fe_off[a]] = 1
        spec["par_fixed"] = fixed
    spec["par"] = par
    return spec


def drift_spec(name, model, d, *, seed, n_tracks=34, lo=6, hi=15, fe_slope=False, smooth_dims=(0,), fix=()):
    """Row-varying DRIFT only (mu smooth in a covariate; tau, nu / kappa / sigma, sigma_obs constant), many short tracks on
    a regular grid without missing rows: the configuration the shared-covariance kernel with streamed drift columns takes
    (k_iso_drift.hip; nllk_ctcrw.hpp:143-149, 211-212, nllk_ou_ssm.hpp:113-124)."""
    rng = np.random.default_rng(seed)
    lengths = list(rng.integers(lo, hi, size=n_tracks))
    ID, times, obs = _tracks(rng, model, d, lengths, irregular=False)
    n = len(ID)
    q = n_sde_par(model, d)
    x = np.clip((np.sin(np.linspace(0, 9, n)) + 1) / 2 + 0.05 * rng.standard_normal(n), 0, 1)
    X_fe, X_re, S_list = [None] * q, [None] * q, []
    if fe_slope:
        X_fe[0] = np.column_stack([np.ones(n), x])
    for a in smooth_dims:
        k = 4 + a
        X_re[a] = bspline_basis(np.clip(x ** (1 + a), 0, 1), n_basis=k)
        S_list.append(second_difference_penalty(k))
    spec = dict(name=name, model=model, n_dim=d, ID=ID, times=times, obs=obs, X_fe=X_fe, X_re=X_re if S_list else None,
                S_list=S_list or None, a0=None, P0=None, H=None, par_fixed=None, na_mode=1, include_penalty=1)
    p = [rng.uniform(-1.5, 0.0)]
    for j in range(q):
        base = rng.uniform(-0.3, 0.5) + (3.0 if (j < d and model == "OU_SSM") else 0.0)
        p += [base] + ([rng.uniform(-0.4, 0.4)] if (j == 0 and fe_slope) else [])
    p += list(rng.uniform(-0.5, 1.0, size=len(S_list)))
    p += list(rng.uniform(-0.3, 0.3, size=sum(s_.shape[0] for s_ in S_list)))
    spec["par"] = np.array(p)
    if fix:
        pb = problem_from_spec(spec)
        fixed = np.zeros(pb.n_par_full, dtype=np.uint8)
        fixed[list(fix)] = 1
        spec["par_fixed"] = fixed
    return spec


def elephant_spec():
    """6-row CTCRW micro-fixture at the vignette's initial parameters
    (par0 = c(0, 0, 1, 1), fixpar = c("mu1", "mu2"): smoothSDE.rmd:476-490)."""
    rows = ELEPHANT_ROWS
    spec = dict(name="elephant6_ctcrw", model="CTCRW", n_dim=2, ID=np.ones(6), times=rows[:, 3].copy(),
                obs=rows[:, :2].copy(), X_fe=None, X_re=None, S_list=None, a0=None, P0=None, H=None,
                na_mode=1, include_penalty=1)
    fixed = np.zeros(5, dtype=np.uint8)
    fixed[1:3] = 1
    spec["par_fixed"] = fixed
    spec["par"] = np.array([0.0, 0.0, 0.0, 0.0, 0.0])  # log_sigma_obs, mu1, mu2, log tau, log nu
    return spec


def eseal_spec(name, seed, lengths, variant="const", na_rows=()):
    """Elephant-seal body-condition model (nllk_e_seal_ssm.hpp): drift rate observations, lipid mass state."""
    rng = np.random.default_rng(seed)
    ID, times, L0s, obs, hh, RR = [], [], [], [], [], []
    t0 = 0.0
    for k, T in enumerate(lengths):
        dts = rng.uniform(0.7, 1.5, size=T)
        tt = t0 + np.cumsum(dts)
        t0 = tt[-1] + 5.0
        L = 30.0 + 5.0 * k + np.cumsum(0.3 * dts + 0.4 * np.sqrt(dts) * rng.standard_normal(T))
        R = rng.uniform(150.0, 250.0, size=T)
        h = rng.integers(3, 25, size=T).astype(float)
        y = -0.578 + 1.214 * L / R + rng.standard_normal(T) / np.sqrt(h)
        ID += [float(k)] * T; times += list(tt); L0s.append(L[0]); obs += list(y); hh += list(h); RR += list(R)
    n = len(ID)
    obs = np.array(obs)[:, None]
    for r in na_rows:
        obs[r, 0] = float("nan")
    spec = dict(name=name, model="ESEAL_SSM", n_dim=1, ID=np.array(ID), times=np.array(times), obs=obs, X_fe=None,
                X_re=None, S_list=None, a0=np.column_stack([np.ones(len(lengths)), L0s]), P0=None, H=None,
                par_fixed=None, na_mode=1, include_penalty=1, eseal_h=np.array(hh), eseal_R=np.array(RR))
    p = [rng.uniform(-0.3, 0.3), -0.578 + rng.uniform(-0.1, 0.1), np.log(1.214) + rng.uniform(-0.1, 0.1)]
    if variant == "const":
        p += [rng.uniform(0.1, 0.5), rng.uniform(-1.2, -0.5)]
    else:
        x = np.clip((np.sin(np.linspace(0, 5, n)) + 1) / 2 + 0.05 * rng.standard_normal(n), 0, 1)
        spec.update(X_fe=[np.column_stack([np.ones(n), x]), None], X_re=[None, bspline_basis(x, n_basis=4)],
                    S_list=[second_difference_penalty(4)])
        p += [rng.uniform(0.1, 0.5), rng.uniform(-0.3, 0.3), rng.uniform(-1.2, -0.5)]   # mu: intercept + slope, log sigma
        p += [rng.uniform(-0.5, 1.0)]                                                      # log_lambda
        p += list(rng.uniform(-0.3, 0.3, size=4))                                          # coeff_re
    spec["par"] = np.array(p)
    return spec


SPEC_CALLS = [0]     # problems built from golden records so far (the kernel-coverage ledger of tests/conftest.py reads it)


def problem_from_spec(spec, **over) -> Problem:
    SPEC_CALLS[0] += 1
    kw = dict(a0=spec.get("a0"), P0=spec.get("P0"), H=spec.get("H"), par_fixed=spec.get("par_fixed"),
              include_penalty=spec.get("include_penalty", 1), na_mode=spec.get("na_mode", 1),
              other_data=spec.get("other_data"), t_decay=spec.get("t_decay"), col_decay=spec.get("col_decay"),
              ind_decay=spec.get("ind_decay"), eseal_h=spec.get("eseal_h"), eseal_R=spec.get("eseal_R"))
    kw.update(over)
    return Problem(spec["model"], spec["ID"], spec["times"], spec["obs"], spec.get("X_fe"), spec.get("X_re"),
                   spec.get("S_list"), **kw)


def all_specs():
    specs = [elephant_spec()]
    seed = 100
    for model in ("CTCRW", "OU_SSM", "BM_SSM"):
        for d in (1, 2):
            seed += 1
            specs.append(make_spec(f"{model}_d{d}_const", model, d, seed=seed, lengths=[9, 2, 14, 6, 11],
                                   na_rows=(3, 16, 17, 30)))
            seed += 1
            specs.append(make_spec(f"{model}_d{d}_const_regular_fixmu", model, d, seed=seed,
                                   lengths=[12, 12, 12], irregular=False, fix_mu=True))
            seed += 1
            specs.append(make_spec(f"{model}_d{d}_tv", model, d, seed=seed, lengths=[13, 8, 10],
                                   variant="tv", na_rows=(5,)))
            seed += 1
            specs.append(make_spec(f"{model}_d{d}_H", model, d, seed=seed, lengths=[10, 7], with_H=True,
                                   na_rows=(4,)))
            seed += 1
            specs.append(make_spec(f"{model}_d{d}_P0", model, d, seed=seed, lengths=[8, 9], with_P0=True))
    specs.append(make_spec("CTCRW_d2_tv2_RNA", "CTCRW", 2, seed=171, lengths=[15, 9], variant="tv2",
                           na_rows=(6, 7), na_mode=0))
    specs.append(make_spec("CTCRW_d2_tv_H_P0", "CTCRW", 2, seed=172, lengths=[11, 12], variant="tv",
                           with_H=True, with_P0=True))
    for model in ("OU", "BM"):
        for d in (1, 2):
            seed += 1
            specs.append(make_spec(f"{model}_d{d}_const", model, d, seed=seed, lengths=[9, 2, 14, 6],
                                   na_rows=(3, 12)))
            seed += 1
            specs.append(make_spec(f"{model}_d{d}_tv", model, d, seed=seed, lengths=[16, 11],
                                   variant="tv", na_rows=(5,)))
    specs.append(make_spec("OU_d1_tv2", "OU", 1, seed=181, lengths=[20, 13], variant="tv2"))
    # BM with Student-t increments (tr_dens.hpp:38-44), degrees of freedom in other_data
    specs.append(make_spec("BM_t_d1_const", "BM_t", 1, seed=191, lengths=[9, 2, 14, 6], na_rows=(3, 12), other_data=5.0))
    specs.append(make_spec("BM_t_d1_tv", "BM_t", 1, seed=192, lengths=[16, 11], variant="tv", na_rows=(5,), other_data=3.5))
    # decaying random-effect columns (nllk_sde.hpp:47-58): one rate, two rates
    specs.append(make_spec("OU_d1_decay", "OU", 1, seed=201, lengths=[18, 12], variant="tv", na_rows=(7,), decay=True))
    specs.append(make_spec("BM_d2_decay2", "BM", 2, seed=202, lengths=[15, 14], variant="tv2", decay=True))
    # elephant-seal state-space model (nllk_e_seal_ssm.hpp)
    specs.append(eseal_spec("ESEAL_const", 211, [14, 9, 11], na_rows=(4, 20)))
    specs.append(eseal_spec("ESEAL_tv", 212, [16, 12], variant="tv", na_rows=(6,)))
    # Cox-Ingersoll-Ross (tr_dens.hpp:53-67)
    specs.append(make_spec("CIR_d1_const", "CIR", 1, seed=221, lengths=[9, 2, 14, 6], na_rows=(3, 12)))
    specs.append(make_spec("CIR_d2_tv", "CIR", 2, seed=222, lengths=[14, 10], variant="tv", na_rows=(5,)))
    # responses wider than two columns (nllk_ctcrw.hpp:12-24: log-determinant beyond n_dim = 2; nllk_sde.hpp:77-84)
    specs.append(make_spec("CTCRW_d3_const", "CTCRW", 3, seed=231, lengths=[9, 2, 14, 6, 11], na_rows=(3, 16, 17, 30)))
    specs.append(make_spec("CTCRW_d4_const_regular_RNA", "CTCRW", 4, seed=232, lengths=[12, 12, 12], irregular=False, na_rows=(7,), na_mode=0))
    specs.append(make_spec("OU_SSM_d3_tv", "OU_SSM", 3, seed=233, lengths=[13, 8, 10], variant="tv", na_rows=(5,)))
    specs.append(make_spec("BM_SSM_d5_const", "BM_SSM", 5, seed=234, lengths=[10, 7, 9]))
    specs.append(make_spec("OU_d3_tv2", "OU", 3, seed=235, lengths=[16, 11], variant="tv2", na_rows=(5,)))
    specs.append(make_spec("BM_d4_const", "BM", 4, seed=236, lengths=[9, 2, 14, 6], na_rows=(3, 12)))
    # a regular schedule with fixes absent from the data: intervals of 1-3 steps (laid out on the lattice by the engine)
    specs.append(make_spec("CTCRW_d2_lattice", "CTCRW", 2, seed=241, lengths=[19, 3, 24, 12], irregular="lattice", na_rows=(5, 30)))
    specs.append(make_spec("OU_SSM_d1_lattice_fixmu", "OU_SSM", 1, seed=242, lengths=[22, 15], irregular="lattice", fix_mu=True))
    specs.append(make_spec("BM_SSM_d3_lattice", "BM_SSM", 3, seed=243, lengths=[14, 18, 9], irregular="lattice", na_rows=(7,), na_mode=0))
    # row-varying drift on the shared-covariance path (>= 32 complete tracks on a regular grid: k_iso_drift.hip)
    specs.append(drift_spec("OU_SSM_d1_drift", "OU_SSM", 1, seed=251))
    specs.append(drift_spec("CTCRW_d2_drift", "CTCRW", 2, seed=252, smooth_dims=(0, 1)))
    specs.append(drift_spec("BM_SSM_d2_drift_fixsig", "BM_SSM", 2, seed=253, smooth_dims=(1,), fix=(0,)))
    specs.append(drift_spec("CTCRW_d1_drift_fe", "CTCRW", 1, seed=254, fe_slope=True, smooth_dims=()))
    return specs
