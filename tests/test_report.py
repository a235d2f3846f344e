"""CPU suite: the sdreport replacement (smoothsde_amd/report.py).  Known-answer test on an exactly Gaussian joint
objective (the Laplace approximation is exact there, so jointPrecision must reproduce the generating precision),
then an oracle-backed joint objective checked against torch-autograd Hessians of the dense restatement."""
import numpy as np
import torch

from cases import problem_from_spec
from golden_io import load_golden
from oracle_lib import oracle_eval
from refimpl import direct_nllk, kalman_dense_nllk, penalty
from smoothsde_amd.laplace import LaplaceObjective
from smoothsde_amd.report import fd_hessian, fd_hessian_fn, sdreport

GOLD = {r["name"]: r for r in load_golden()}


class _Layout:
    """minimal stand-in for capi.Problem's layout fields"""
    def __init__(self, n_fe, n_smooth, n_re, kalman=False):
        self.kalman = kalman
        self.off_fe = 1 if kalman else 0
        self.n_fe, self.n_smooth, self.n_re = n_fe, n_smooth, n_re
        self.off_lambda = self.off_fe + n_fe
        self.off_re = self.off_lambda + n_smooth


def test_gaussian_joint_reproduces_its_precision():
    rng = np.random.default_rng(3)
    nf, nr = 3, 5
    A = rng.normal(size=(nf + nr, nf + nr))
    M = A @ A.T + (nf + nr) * np.eye(nf + nr)
    z0 = rng.normal(size=nf + nr)

    def joint(p):
        d = p - z0
        return 0.5 * d @ M @ d, M @ d

    io, ir = np.arange(nf), np.arange(nf, nf + nr)
    lap = LaplaceObjective(joint, z0 + 0.1, io, ir)
    theta = z0[io].copy()
    lap.fn(theta)
    full = z0.copy()
    rep = sdreport(_Layout(nf, 0, nr), joint, full, io, ir, marginal_fn=lambda t: lap.fn(t, update_warm_start=False))
    assert np.allclose(rep.jointPrecision, M, rtol=1e-5, atol=1e-5)
    schur = M[:nf, :nf] - M[:nf, nf:] @ np.linalg.solve(M[nf:, nf:], M[nf:, :nf])
    assert np.allclose(rep.hessian_fixed, schur, rtol=1e-4, atol=1e-4)
    assert np.allclose(rep.cov_fixed, np.linalg.inv(M)[:nf, :nf], rtol=1e-4, atol=1e-5)
    # the marginal covariance of the fixed block of the joint equals cov.fixed
    assert np.allclose(np.linalg.inv(rep.jointPrecision)[:nf, :nf], rep.cov_fixed, rtol=1e-4, atol=1e-6)
    assert rep.names_all() == ["coeff_fe"] * nf + ["coeff_re"] * nr
    est = rep.as_list("Estimate")
    assert np.allclose(est["coeff_fe"], z0[:nf]) and np.allclose(est["coeff_re"], z0[nf:])
    se = rep.as_list("Std. Error")
    assert np.allclose(se["coeff_re"], np.sqrt(np.diag(np.linalg.inv(M)))[nf:], rtol=1e-4)


def test_fd_hessians():
    rng = np.random.default_rng(1)
    A = rng.normal(size=(4, 4)); M = A @ A.T + np.eye(4)
    f = lambda x: 0.5 * x @ M @ x + np.sum(np.sin(x))
    g = lambda x: M @ x + np.cos(x)
    x = rng.normal(size=4)
    H = M - np.diag(np.sin(x))
    assert np.allclose(fd_hessian(g, x), H, rtol=1e-7, atol=1e-7)
    assert np.allclose(fd_hessian_fn(f, x, 1e-3), H, rtol=1e-4, atol=1e-4)


def _check_oracle_backed(name):
    rec = GOLD[name]
    pb = problem_from_spec(rec)
    par = rec["par"].copy()
    free = pb.free_index()
    ir = np.array([k for k in free if pb.off_re <= k < pb.off_re + pb.n_re])
    io = np.array([k for k in free if k not in set(ir.tolist()) and not (pb.off_lambda <= k < pb.off_lambda + pb.n_smooth)])
    rep = sdreport(pb, lambda p: oracle_eval(pb, p, order=1), par, io, ir, marginal_fn=None)
    # exact joint Hessian by autograd of the dense restatement
    idx = np.concatenate([io, ir])
    p0 = torch.tensor(par)

    def joint_t(x):
        p = p0.clone()
        p[list(idx)] = x
        return (kalman_dense_nllk(pb, p) if pb.kalman else direct_nllk(pb, p)) + penalty(pb, p)

    H = torch.autograd.functional.hessian(joint_t, torch.tensor(par[idx])).numpy()
    nf = len(io)
    scale = np.max(np.abs(H))
    assert np.allclose(rep.jointPrecision[nf:, nf:], H[nf:, nf:], rtol=1e-5, atol=1e-6 * scale)
    assert np.allclose(rep.jointPrecision[:nf, nf:], H[:nf, nf:], rtol=1e-5, atol=1e-6 * scale)
    schur = H[:nf, :nf] - H[:nf, nf:] @ np.linalg.solve(H[nf:, nf:], H[nf:, :nf])
    assert np.allclose(rep.hessian_fixed, schur, rtol=1e-4, atol=1e-5 * scale)
    assert rep.names_random == ["coeff_re"] * len(ir)
    assert set(rep.names_fixed) <= {"log_sigma_obs", "coeff_fe", "log_lambda"}


def test_oracle_backed_direct_ou():
    _check_oracle_backed("OU_d1_tv")


def test_oracle_backed_kalman_ctcrw():
    _check_oracle_backed("CTCRW_d1_tv")
