"""CPU suite: the Laplace layer (smoothsde_amd/laplace.py) on an oracle-backed joint objective, against an exact
Laplace value computed independently (torch autograd Hessian of the dense restatement, tight inner solve)."""
import numpy as np
import torch
from scipy.optimize import minimize

from cases import problem_from_spec
from golden_io import load_golden
from oracle_lib import oracle_eval
from refimpl import direct_nllk, kalman_dense_nllk, penalty
from smoothsde_amd.laplace import LaplaceObjective

GOLD = {r["name"]: r for r in load_golden()}


def _exact_laplace(pb, par, idx_r):
    idx_r = list(idx_r)

    def joint_t(u, p0):
        p = p0.clone()
        p[idx_r] = u
        return (kalman_dense_nllk(pb, p) if pb.kalman else direct_nllk(pb, p)) + penalty(pb, p)

    p0 = torch.tensor(par)

    def f(u):
        ut = torch.tensor(u, requires_grad=True)
        v = joint_t(ut, p0)
        (g,) = torch.autograd.grad(v, ut)
        return float(v.detach()), g.numpy()

    res = minimize(f, par[idx_r], jac=True, method="BFGS", options=dict(gtol=1e-10))
    res = minimize(f, res.x, jac=True, method="BFGS", options=dict(gtol=1e-12))
    H = torch.autograd.functional.hessian(lambda u: joint_t(u, p0), torch.tensor(res.x)).numpy()
    sign, ld = np.linalg.slogdet(H)
    return res.fun + 0.5 * ld - 0.5 * len(idx_r) * np.log(2 * np.pi), res.x


def _check(name):
    rec = GOLD[name]
    pb = problem_from_spec(rec)
    par = rec["par"].copy()
    idx_r = list(range(pb.off_re, pb.off_re + pb.n_re))
    idx_o = [k for k in range(pb.n_par_full) if k not in idx_r]
    lap = LaplaceObjective(lambda p: oracle_eval(pb, p, order=1), par, idx_o, idx_r)
    f = lap.fn(par[idx_o])
    f_exact, u_exact = _exact_laplace(pb, par, idx_r)
    assert abs(f - f_exact) <= 1e-6 * max(1.0, abs(f_exact)), (f, f_exact)
    assert np.max(np.abs(lap.u_hat - u_exact)) <= 1e-5
    # outer gradient: finite differences of the marginal itself, checked by a coarser difference
    g = lap.gr(par[idx_o])
    k = 1
    e = np.zeros(len(idx_o)); e[k] = 1e-3
    fd = (lap.fn(par[idx_o] + e) - lap.fn(par[idx_o] - e)) / 2e-3
    assert abs(g[k] - fd) <= 1e-3 * max(1.0, abs(fd))


def test_laplace_direct_ou():
    _check("OU_d1_tv")


def test_laplace_kalman_bm_ssm():
    _check("BM_SSM_d1_tv")
