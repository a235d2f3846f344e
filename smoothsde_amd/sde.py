"""Host-side mirror of the reference's `SDE` R6 class for the nllk/gradient path.

The reference host is R (`SDE$new`, `$setup()`, `$fit()`, `logLik.SDE`:
/root/reference/R/sde.R:45-182, 491-720, R/utility.R:115-123).  R is not in this image, so
this module restates that surface in Python with the same names, argument meaning and error
behaviour; the R glue that a maintainer would drop into the package is R_glue/ (see
INTEGRATION.md).  What changes underneath: `$setup()` builds an `Engine` (HIP, via the C ABI)
instead of calling `TMB::MakeADFun`, and returns an object with the same `par`, `fn`, `gr`
members that `$fit()` hands to a BFGS optimiser (the reference uses `optim(method = "BFGS")`,
R/sde.R:694-697).

Out of scope here, as in SURVEY.md: mgcv smooth construction (stays in R; a B-spline stand-in
is provided for `s(x, k=)` and an identity-penalised indicator block for `s(ID, bs="re")`),
sdreport, plotting, posterior simulation.  The Laplace approximation over `coeff_re`
(`random = "coeff_re"`, R/sde.R:522) is provided by `smoothsde_amd.laplace` on top of the GPU
gradient (finite-difference Hessian / outer gradient instead of TMB's AD).
"""
from __future__ import annotations

import re
import time
import warnings
from typing import Dict, List, Optional, Sequence

import numpy as np

from . import capi
from .synth import bspline_basis, second_difference_penalty

PAR_NAMES = {
    "BM": lambda d: _mu_names(d) + ["sigma"],
    "BM_t": lambda d: ["mu", "sigma"],                   # R/sde.R:61
    "BM_SSM": lambda d: _mu_names(d) + ["sigma"],
    "ESEAL_SSM": lambda d: ["mu", "sigma"],              # R/sde.R:70
    "OU": lambda d: _mu_names(d) + ["tau", "kappa"],
    "OU_SSM": lambda d: _mu_names(d) + ["tau", "kappa"],
    "CIR": lambda d: _mu_names(d) + ["beta", "sigma"],   # R/sde.R:66-67 (every parameter on the log scale)
    "CTCRW": lambda d: _mu_names(d) + ["tau", "nu"],
}


def _mu_names(d):  # c(mu = lapply(1:n_dim, ...)) names: "mu" for one dimension, "mu1", "mu2", ... otherwise
    return ["mu"] if d == 1 else [f"mu{i + 1}" for i in range(d)]


class Design:
    """Pre-built design blocks of one SDE parameter (what mgcv::gam(fit=FALSE) gives the reference,
    R/sde.R:396-424): X_fe (n x nsdf, first column the intercept), X_re (n x k) and the penalty
    blocks S (one per smooth, sizes adding up to k)."""

    def __init__(self, X_fe=None, X_re=None, S: Optional[Sequence] = None, names_fe=None, names_re=None):
        self.X_fe = None if X_fe is None else np.asarray(X_fe, dtype=np.float64)
        self.X_re = None if X_re is None else np.asarray(X_re, dtype=np.float64)
        self.S = [] if S is None else [np.asarray(s, dtype=np.float64) for s in S]
        self.names_fe, self.names_re = names_fe, names_re


def _parse_formula(form: str, data: Dict[str, np.ndarray], n: int) -> Design:
    """Tiny stand-in for mgcv's formula interface: `~ 1`, `~ x1 + x2`, `s(x, k = K)`, `s(ID, bs = "re")`."""
    rhs = form.strip()
    if rhs.startswith("~"):
        rhs = rhs[1:]
    terms = [t.strip() for t in re.split(r"\+(?![^()]*\))", rhs) if t.strip()]
    X_fe, names_fe = [np.ones(n)], ["(Intercept)"]
    X_re, S, names_re = [], [], []
    for t in terms:
        if t == "1":
            continue
        m = re.fullmatch(r"s\(\s*([A-Za-z_][\w.]*)\s*(?:,(.*))?\)", t)
        if m:
            var, opts = m.group(1), (m.group(2) or "")
            if var not in data:
                raise KeyError(f"variable '{var}' not found in 'data'")
            if re.search(r"bs\s*=\s*[\"']re[\"']", opts):
                codes, inv = np.unique(np.asarray(data[var]), return_inverse=True)
                Z = np.zeros((n, len(codes)))
                Z[np.arange(n), inv] = 1.0
                X_re.append(Z)
                S.append(np.eye(len(codes)))
                names_re += [f"s({var}).{i + 1}" for i in range(len(codes))]
            else:
                km = re.search(r"k\s*=\s*(\d+)", opts)
                k = int(km.group(1)) if km else 10
                x = np.asarray(data[var], dtype=np.float64)
                lo, hi = np.nanmin(x), np.nanmax(x)
                xs = (x - lo) / (hi - lo) if hi > lo else np.zeros_like(x)
                # k B-splines with the sum-to-zero constraint absorbed as mgcv does (k - 1 columns): without it the
                # columns add up to the intercept and the marginal likelihood is flat along that direction
                B = bspline_basis(xs, n_basis=k)
                Sk = second_difference_penalty(k)
                Q, _ = np.linalg.qr(B.sum(axis=0)[:, None], mode="complete")
                Z = Q[:, 1:]                                   # null space of the constraint 1'B c = 0
                Sz = Z.T @ Sk @ Z
                if re.search(r"bs\s*=\s*[\"'](ts|cs)[\"']", opts):
                    # shrinkage smooths (the vignette's bs = "ts"): the penalty's null space gets a small eigenvalue
                    w, V = np.linalg.eigh(0.5 * (Sz + Sz.T))
                    pos = w > 1e-8 * w[-1]
                    w = np.where(pos, w, 0.1 * w[pos].min())
                    Sz = (V * w) @ V.T
                X_re.append(B @ Z)
                S.append(Sz)
                names_re += [f"s({var}).{i + 1}" for i in range(k - 1)]
        else:
            if t not in data:
                raise KeyError(f"variable '{t}' not found in 'data'")
            X_fe.append(np.asarray(data[t], dtype=np.float64))
            names_fe.append(t)
    return Design(np.column_stack(X_fe), np.column_stack(X_re) if X_re else None, S, names_fe, names_re)


class TmbObj:
    """The members of a `MakeADFun` object that the reference uses: `par`, `fn`, `gr` over the FREE
    parameters (fixed ones are held at their values, TMB's `map`).  `fn`/`gr` arrive as separate
    calls with the same x (R/sde.R:694-697): one GPU evaluation serves both."""

    def __init__(self, engine, par_full: np.ndarray, free: np.ndarray):
        self.engine = engine
        self.par_full = np.array(par_full, dtype=np.float64)
        self.free = free
        self.par = self.par_full[free].copy()
        self._last_x = None
        self._last = None
        self.n_eval = 0

    def _eval(self, x):
        x = np.asarray(x, dtype=np.float64)
        if self._last_x is None or not np.array_equal(x, self._last_x):
            full = self.par_full.copy()
            full[self.free] = x
            val, grad = self.engine.eval(full, order=1)
            self._last_x, self._last = x.copy(), (val, grad[self.free])
            self.n_eval += 1
        return self._last

    def fn(self, x=None):
        return self._eval(self.par if x is None else x)[0]

    def gr(self, x=None):
        return self._eval(self.par if x is None else x)[1]

    def he(self, x=None):
        """Hessian over the free parameters by central differences of the GPU gradient (`tmb_obj$he`,
        used by edf_conditional, R/sde.R:1363)."""
        from .report import fd_hessian
        return fd_hessian(self.gr, self.par if x is None else np.asarray(x, dtype=np.float64))


class SDE:
    def __init__(self, formulas=None, data=None, type=None, response=None, par0=None, fixpar=None, other_data=None):
        if data is None or type is None or response is None:
            raise TypeError("SDE(formulas, data, type, response, ...) needs data, type and response")
        self.type_ = type
        self.response_ = [response] if isinstance(response, str) else list(response)
        self.fixpar_ = None if fixpar is None else list(fixpar)
        self.other_data_ = other_data or {}
        if hasattr(data, "to_dict") and hasattr(data, "columns"):
            data = {c: data[c].to_numpy() for c in data.columns}
        data = dict(data)
        if any(r not in data for r in self.response_):
            raise ValueError("'response' not found in 'data'")          # R/sde.R:51-52
        if type in capi.UNSUPPORTED_MODELS:
            raise NotImplementedError(f"SDE type {type!r} is outside this engine's scope")
        if type not in PAR_NAMES:
            raise ValueError("Unknown SDE type")
        n_dim = len(self.response_)
        names = PAR_NAMES[type](n_dim)
        if formulas is None:
            formulas = {k: "~1" for k in names}                         # R/sde.R:93-94
        elif len(formulas) != len(names):
            raise ValueError(f"'formulas' should be a list of length {len(names)} for the model {type}, "
                             f"with components {', '.join(names)}")    # R/sde.R:95-100
        elif list(formulas.keys()) != names:
            raise ValueError(f"'formulas' should be a list with components {', '.join(names)}")
        if self.fixpar_:
            for f in self.fixpar_:
                if not (isinstance(formulas[f], str) and formulas[f].replace(" ", "") == "~1"):
                    raise ValueError("formulas should be ~1 for fixed parameters")   # R/sde.R:106-108
        self.formulas_ = dict(formulas)
        n = len(np.asarray(data[self.response_[0]]))
        if "ID" not in data:
            warnings.warn("No ID column found in 'data', assuming same ID for all observations")  # R/sde.R:112-115
            data["ID"] = np.ones(n)
        if "time" not in data:
            raise ValueError("'data' should have a time column")        # R/sde.R:121-123
        self.data_ = data
        self.n_ = n
        self.names_ = names
        self.make_mat()
        self.coeff_fe_ = np.zeros(int(np.sum(self.terms_["ncol_fe"])))   # R/sde.R:138-140
        self.coeff_re_ = np.zeros(int(np.sum(self.terms_["ncol_re_par"])))
        self.lambda_vals_ = np.ones(len(self.terms_["ncol_re"]))
        if par0 is not None:
            if len(par0) != len(names):
                raise ValueError(f"'par0' should be of length {len(names)} with one entry for each SDE parameter "
                                 f"({', '.join(names)})")               # R/sde.R:147-151
            i0 = np.concatenate([[0], np.cumsum(self.terms_["ncol_fe"])[:-1]]).astype(int)
            for i, nm in enumerate(names):                               # link: identity for mu*, log otherwise
                identity = nm.startswith("mu") and type != "CIR"          # CIR: log link for mu too (R/sde.R:66)
                self.coeff_fe_[i0[i]] = par0[i] if identity else np.log(par0[i])
        self.tmb_obj_ = None
        self.tmb_obj_joint_ = None
        self.tmb_rep_ = None
        self.out_ = None
        self.engine_ = None

    # -- accessors (R/sde.R:188-360) --------------------------------------------------------------
    def formulas(self): return self.formulas_
    def data(self): return self.data_
    def type(self): return self.type_
    def response(self): return self.response_
    def fixpar(self): return self.fixpar_
    def coeff_fe(self): return self.coeff_fe_
    def coeff_re(self): return self.coeff_re_
    def lambda_(self): return self.lambda_vals_    # `lambda` is a Python keyword
    def terms(self): return self.terms_
    def mats(self): return self.mats_
    def out(self): return self.out_
    def tmb_obj(self): return self.tmb_obj_
    def tmb_obj_joint(self): return self.tmb_obj_joint_
    def tmb_rep(self): return self.tmb_rep_

    def obs(self):
        return np.column_stack([np.asarray(self.data_[r], dtype=np.float64) for r in self.response_])

    def ind_fixcoeff(self):
        """Indices (in coeff_fe) of the coefficients of fixed SDE parameters (R/sde.R ind_fixcoeff)."""
        if not self.fixpar_:
            return np.zeros(0, dtype=int)
        i0 = np.concatenate([[0], np.cumsum(self.terms_["ncol_fe"])[:-1]]).astype(int)
        return np.array([i0[self.names_.index(f)] for f in self.fixpar_], dtype=int)

    # -- design matrices (R/sde.R:378-455) -------------------------------------------------------------
    def make_mat(self):
        designs: List[Design] = []
        for nm in self.names_:
            f = self.formulas_[nm]
            designs.append(f if isinstance(f, Design) else _parse_formula(f, self.data_, self.n_))
        ncol_fe = [1 if d.X_fe is None else d.X_fe.shape[1] for d in designs]
        ncol_re_par = [0 if d.X_re is None else d.X_re.shape[1] for d in designs]
        ncol_re = [s.shape[0] for d in designs for s in d.S]
        self.designs_ = designs
        self.terms_ = dict(ncol_fe=np.array(ncol_fe), ncol_re=np.array(ncol_re, dtype=int),
                           ncol_re_par=np.array(ncol_re_par))
        self.mats_ = dict(X_list_fe=[d.X_fe for d in designs], X_list_re=[d.X_re for d in designs],
                          S_list=[s for d in designs for s in d.S])
        return self.mats_

    # -- TMB setup counterpart (R/sde.R:491-670) -----------------------------------------------------------
    def _problem(self, include_penalty=1, fix_lambda=True, **over):
        X_fe = []
        for d in self.designs_:
            # an intercept-only block is passed as "no column" (broadcast), SURVEY 7.3-5
            X_fe.append(None if (d.X_fe is None or (d.X_fe.shape[1] == 1 and np.all(d.X_fe == 1.0))) else d.X_fe)
        X_re = [d.X_re for d in self.designs_]
        kw = dict(a0=None, P0=self.other_data_.get("P0"), H=self.other_data_.get("H"), include_penalty=include_penalty,
                  other_data=self.other_data_.get("df") if self.type_ == "BM_t" else None)   # R/sde.R:539-541
        if self.type_ == "ESEAL_SSM":                          # R/sde.R:599-614: a0 = (1, first dep_fat of every track)
            ids = np.asarray(self.data_["ID"])
            first = np.r_[True, ids[1:] != ids[:-1]]
            kw.update(a0=np.column_stack([np.ones(first.sum()), np.asarray(self.data_["dep_fat"], dtype=float)[first]]),
                      P0=np.diag([0.0, 10.0]), eseal_h=self.data_["h"], eseal_R=self.data_["R"])
        if self.other_data_.get("t_decay") is not None:        # decaying response model, R/sde.R:635-644 (1-based in R)
            kw.update(t_decay=self.other_data_["t_decay"],
                      col_decay=np.asarray(self.other_data_["col_decay"], dtype=int) - 1,
                      ind_decay=np.asarray(self.other_data_["ind_decay"], dtype=int) - 1)
        kw.update(over)
        pb = capi.Problem(self.type_, self.data_["ID"], self.data_["time"], self.obs(), X_fe, X_re,
                          self.mats_["S_list"], **kw)
        fixed = pb.par_fixed.copy()
        for k in self.ind_fixcoeff():                      # map$coeff_fe with NA for fixed coefficients
            fixed[pb.off_fe + k] = 1
        if fix_lambda:                                     # joint fit: smoothing parameters held at their values
            fixed[pb.off_lambda:pb.off_lambda + pb.n_smooth] = 1
        pb.par_fixed = fixed
        return pb

    def _par_full(self, pb):
        p = np.zeros(pb.n_par_full)
        if pb.lead_names == ["log_sigma_obs"]:
            p[0] = 0.0                                      # log_sigma_obs = 0 (R/sde.R:560, 590)
        elif self.type_ == "ESEAL_SSM":
            p[0:3] = [np.log(1.0), -0.578, np.log(1.214)]   # ssm_par, R/sde.R:606-608
        p[pb.off_fe:pb.off_fe + pb.n_fe] = self.coeff_fe_
        p[pb.off_lambda:pb.off_lambda + pb.n_smooth] = np.log(self.lambda_vals_)
        if pb.n_decay:
            p[pb.off_decay:pb.off_decay + pb.n_decay] = np.log(getattr(self, "rho_", np.ones(pb.n_decay)))   # R/sde.R:177
        p[pb.off_re:pb.off_re + pb.n_re] = self.coeff_re_
        return p

    def setup(self, silent=True, map=None, laplace=None):
        """laplace=None: integrate coeff_re out (Laplace approximation, like `random = "coeff_re"` in the
        reference, R/sde.R:522) whenever there are random effects; laplace=False: joint penalised
        likelihood with the smoothing parameters held fixed."""
        has_re = len(self.mats_["S_list"]) > 0
        self.laplace_ = has_re if laplace is None else (bool(laplace) and has_re)
        pb = self._problem(include_penalty=1, fix_lambda=not self.laplace_)
        self.problem_ = pb
        if self.laplace_:                                  # exact H_uu / H_u,theta wherever the batch is evaluated (include/ssde.h)
            pb.flags |= capi.FLAG_EXACT_HESS
        self.engine_ = capi.Engine(pb)
        self.joint_obj_ = TmbObj(self.engine_, self._par_full(pb), pb.free_index())   # all free parameters, fixed + random
        self.tmb_obj_ = self.joint_obj_
        if self.laplace_:
            from .laplace import EngineLaplaceObjective
            free = set(pb.free_index().tolist())
            idx_r = [k for k in range(pb.off_re, pb.off_re + pb.n_re) if k in free]
            idx_o = [k for k in sorted(free) if k not in set(idx_r)]
            # fn / gr = ssde_laplace_eval: the entry point an R or C host uses for random = "coeff_re"
            self.tmb_obj_ = EngineLaplaceObjective(self.engine_, self._par_full(pb), idx_o, idx_r)
        # joint object "excluding penalty" (R/sde.R:663-669): include_penalty = 0 is honoured by the
        # direct families only (Q7); the Kalman families share the same engine
        if pb.kalman or pb.n_smooth == 0:
            self.tmb_obj_joint_ = self.joint_obj_
        else:
            pbj = self._problem(include_penalty=0, fix_lambda=not self.laplace_)
            self.engine_joint_ = capi.Engine(pbj)
            self.tmb_obj_joint_ = TmbObj(self.engine_joint_, self._par_full(pbj), pbj.free_index())
        return self.tmb_obj_

    def fit(self, silent=True, map=None, maxiter=200, optimizer="scipy", staged=True):
        """optimizer = "scipy": scipy's BFGS; "optim": R's optim(method = "BFGS") restated (smoothsde_amd/optim.py, the
        reference's own call, R/sde.R:694-697 -- pure backtracking from a unit step along -g, which on n >> 10^3 rows
        first jumps to absurd log-scale values exactly as it does in R).
        staged: with random effects, first fit the fixed effects alone (coeff_re = 0, log_lambda at its start) and
        start the Laplace fit from there.  The reference starts the Laplace fit directly from par0 with
        log_sigma_obs = 0; with a finite-difference Laplace layer that start is outside the region where the inner
        problem is well conditioned."""
        if self.tmb_obj_ is None:
            self.setup(silent=silent, map=map)
        if self.problem_.n_smooth > 0 and not self.laplace_:
            warnings.warn("random effects present: optimising the joint penalised likelihood with the smoothing "
                          "parameters fixed (setup(laplace=False)); the default integrates them out (Laplace)")
        obj = self.tmb_obj_
        t0 = time.perf_counter()

        def _minimise(fn, gr, x0):
            if optimizer == "scipy":
                from scipy.optimize import minimize
                r = minimize(fn, x0, jac=gr, method="BFGS", options=dict(maxiter=maxiter))
                return dict(par=r.x, value=r.fun, counts=(r.nfev, r.njev), convergence=int(not r.success), message=r.message)
            from .optim import optim_bfgs
            out = optim_bfgs(fn, gr, x0, maxit=maxiter)
            out["message"] = None
            return out

        if self.laplace_ and staged:
            pbk = self.problem_
            not_lam = np.array([not (pbk.off_lambda <= k < pbk.off_lambda + pbk.n_smooth) for k in obj.io])
            if not_lam.any():
                u0 = np.zeros(len(obj.ir))
                memo = {}

                def _j1(x):
                    key = x.tobytes()
                    if key not in memo:
                        memo.clear()
                        th = obj.par.copy()
                        th[not_lam] = x
                        v, g = obj.joint(obj._full(th, u0))
                        memo[key] = (v, g[obj.io][not_lam])
                    return memo[key]
                r1 = _minimise(lambda x: _j1(np.asarray(x, dtype=np.float64))[0],
                               lambda x: _j1(np.asarray(x, dtype=np.float64))[1], obj.par[not_lam].copy())
                if np.isfinite(r1["value"]):
                    obj.par[not_lam] = r1["par"]
                    obj.u_hat = u0.copy()
        if optimizer not in ("scipy", "optim"):
            raise ValueError("optimizer must be 'scipy' or 'optim'")
        self.out_ = _minimise(obj.fn, obj.gr, obj.par)
        self.out_["systime"] = time.perf_counter() - t0

        xhat = np.asarray(self.out_["par"], dtype=np.float64)
        full = obj.par_full.copy()
        if self.laplace_:
            full[obj.io] = xhat
            obj.fn(xhat)
            full[obj.ir] = obj.u_hat
        else:
            full[obj.free] = xhat
        pb = self.problem_
        self.par_full_ = full
        self.tmb_rep_ = None
        self.log_sigma_obs_ = full[0] if pb.lead_names == ["log_sigma_obs"] else None
        self.coeff_fe_ = full[pb.off_fe:pb.off_fe + pb.n_fe].copy()        # R/sde.R:707-713
        self.coeff_re_ = full[pb.off_re:pb.off_re + pb.n_re].copy()
        if pb.n_smooth:
            self.lambda_vals_ = np.exp(full[pb.off_lambda:pb.off_lambda + pb.n_smooth])   # R/sde.R:712
        if pb.n_decay:
            self.rho_ = np.exp(full[pb.off_decay:pb.off_decay + pb.n_decay])             # R/sde.R:715-718
        return self.out_

    def _exact_hess(self, par_full, idx):
        """tmb_obj_joint$he counterpart where the engine has exact second derivatives (ssde_hess: BM / OU); None elsewhere."""
        try:
            return self.engine_.hess(par_full, idx)
        except capi.EngineError:
            return None

    def report(self):
        """sdreport counterpart (R/sde.R:702-704): estimates, cov.fixed and the joint precision of
        (fixed, random) built from finite differences of the GPU gradient (smoothsde_amd/report.py)."""
        from .report import sdreport
        pb = self.problem_
        full = self.par_full_ if getattr(self, "par_full_", None) is not None else self.tmb_obj_.par_full
        if self.laplace_:
            lap = self.tmb_obj_
            u_keep = lap.u_hat.copy()

            def marg(theta):
                lap.u_hat = u_keep.copy()
                return lap.fn(theta, update_warm_start=False)
            rep = sdreport(pb, lambda p: self.engine_.eval(p, order=1), full, lap.io, lap.ir, marginal_fn=marg, joint_hess=self._exact_hess)
            lap.u_hat = u_keep
        else:
            free = pb.free_index()
            ir = np.array([k for k in free if pb.off_re <= k < pb.off_re + pb.n_re], dtype=int)
            io = np.array([k for k in free if k not in set(ir.tolist())], dtype=int)
            rep = sdreport(pb, lambda p: self.engine_.eval(p, order=1), full, io, ir, marginal_fn=None, joint_hess=self._exact_hess)
        self.tmb_rep_ = rep
        return rep

    def post_coeff(self, n_post, seed=None):
        """Posterior draws of all coefficients from MVN(par_all, jointCov) (R/sde.R:871-915): a dict with one
        (n_post x k) matrix per parameter block."""
        rep = self.tmb_rep_ if self.tmb_rep_ is not None else self.report()
        cov = np.linalg.inv(rep.jointPrecision) if rep.jointPrecision is not None else rep.cov_fixed   # prec_to_cov
        cov = 0.5 * (cov + cov.T)
        draws = np.random.default_rng(seed).multivariate_normal(rep.par_all(), cov, size=int(n_post), method="eigh")
        names = rep.names_all()
        out = {nm: draws[:, [i for i, k in enumerate(names) if k == nm]] for nm in dict.fromkeys(names)}
        out.setdefault("coeff_re", np.zeros((int(n_post), 0)))          # R/sde.R:907-910
        return out

    def _joint_free_x(self):
        obj = self.tmb_obj_joint_
        x = self.par_full_[obj.free] if getattr(self, "par_full_", None) is not None else obj.par
        return obj, x

    def edf_conditional(self):
        """Effective degrees of freedom (R/sde.R:1360-1375): fixed-effect count + tr(H_re V_re), H the Hessian of
        the joint objective `tmb_obj_joint` (excluding the penalty for the direct families; the Kalman families
        ignore that flag, Q7), V the joint covariance."""
        n_lambda = self.problem_.n_smooth if getattr(self, "laplace_", False) else 0
        edf = float(len(self.tmb_obj_.par) - n_lambda)
        rep = self.tmb_rep_ if self.tmb_rep_ is not None else self.report()
        if rep.jointPrecision is not None:
            pb = self.problem_
            jobj, x = self._joint_free_x()
            H = jobj.he(x)
            ind_h = [i for i, k in enumerate(jobj.free) if pb.off_re <= k < pb.off_re + pb.n_re]
            V = np.linalg.inv(rep.jointPrecision)
            ind_v = [i for i, nm in enumerate(rep.names_all()) if nm == "coeff_re"]
            edf += float(np.trace(H[np.ix_(ind_h, ind_h)] @ V[np.ix_(ind_v, ind_v)]))
        return edf

    def logLik(self):
        """- tmb_obj_joint$fn(par_all) with attributes df = edf_conditional() and nobs (R/utility.R:115-123)."""
        obj, x = self._joint_free_x()
        return dict(value=-obj.fn(x), df=self.edf_conditional(), nobs=self.n_)

    def AIC_conditional(self):
        ll = self.logLik()
        return -2.0 * ll["value"] + 2.0 * ll["df"]                       # R/sde.R:1318-1328

    def AIC_marginal(self):
        n_lambda = 0 if not getattr(self, "laplace_", False) else self.problem_.n_smooth
        return 2.0 * self.out_["value"] + 2.0 * (len(self.out_["par"]) - n_lambda)   # R/sde.R:1340-1349

    def par(self, t=None):
        """SDE parameters on the natural scale for every row (R/sde.R:749-856, new_data = NULL)."""
        out = {}
        fe_off = np.concatenate([[0], np.cumsum(self.terms_["ncol_fe"])]).astype(int)
        re_off = np.concatenate([[0], np.cumsum(self.terms_["ncol_re_par"])]).astype(int)
        for j, (nm, d) in enumerate(zip(self.names_, self.designs_)):
            cf = self.coeff_fe_[fe_off[j]:fe_off[j + 1]]
            lp = (np.full(self.n_, cf[0]) if d.X_fe is None else d.X_fe @ cf)
            if d.X_re is not None:
                lp = lp + d.X_re @ self.coeff_re_[re_off[j]:re_off[j + 1]]
            out[nm] = lp if (nm.startswith("mu") and self.type_ != "CIR") else np.exp(lp)
        return out
