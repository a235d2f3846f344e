"""Loader for the CPU oracle (oracle/liboracle.so) -- test infrastructure only.

Used by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg as the CHECKER of
the HIP path.  Nothing in smoothsde_amd/ imports this module.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

from smoothsde_amd.capi import Problem, SsdeDesc

_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
# SSDE_ORACLE_LIBDIR: load the checker's libraries from another directory (tools/sanitize_cpu.sh: ASan / UBSan builds)
_ORACLE_DIR = os.environ.get("SSDE_ORACLE_LIBDIR") or os.path.join(_ROOT, "oracle")
_LIB = None
_dp = C.POINTER(C.c_double)


CALLS = [0]          # evaluations of the oracle so far (tests/conftest.py: the kernel-coverage ledger reads it around every GPU test)


def build_oracle():
    subprocess.run(["make", "-s", "-C", _ORACLE_DIR], check=True)


def load_oracle():
    global _LIB
    if _LIB is None:
        path = os.path.join(_ORACLE_DIR, "liboracle.so")
        if not os.path.exists(path):
            build_oracle()
        lib = C.CDLL(path)
        lib.oracle_eval.argtypes = [C.POINTER(SsdeDesc), _dp, C.c_int, _dp, _dp, _dp, C.c_int]
        lib.oracle_eval.restype = C.c_int
        lib.oracle_eval_data.argtypes = [C.POINTER(SsdeDesc), _dp, C.c_int, _dp, _dp, C.c_int]
        lib.oracle_eval_data.restype = C.c_int
        lib.oracle_n_par_full.argtypes = [C.POINTER(SsdeDesc)]
        lib.oracle_n_par_full.restype = C.c_int
        _LIB = lib
    return _LIB


_QLIB = None


def keep_P_symmetric(on: bool):
    """ARBITER mode of both oracle libraries (oracle/ssde_oracle.hpp: keep_P_symmetric): the update ends with P <- (P + P') / 2.  Not the
    reference's arithmetic -- the value its model defines where the literal recursion is roundoff (a coupling H_array on long tracks)."""
    global _QLIB
    load_oracle().ssde_oracle_keep_P_symmetric(int(bool(on)))
    _load_quad().ssde_oracle_keep_P_symmetric(int(bool(on)))


def _load_quad():
    global _QLIB
    if _QLIB is None:
        path = os.path.join(_ORACLE_DIR, "liboracle_quad.so")
        if not os.path.exists(path):
            build_oracle()
        _QLIB = C.CDLL(path)
        _QLIB.oracle_eval_quad.argtypes = [C.POINTER(SsdeDesc), _dp, C.c_int, _dp, _dp, C.c_double]
        _QLIB.oracle_eval_quad.restype = C.c_int
    return _QLIB


def oracle_eval_quad(problem: Problem, par, order: int = 1, fd_step: float = 1e-10):
    """The same restated templates evaluated in IEEE binary128 (oracle/oracle_quad.cpp): value rounded to double once,
    gradient by central differences of the binary128 function.  Slow (single thread, software quad arithmetic):
    the arbiter for cases where the double-precision oracle and the engine disagree."""
    _load_quad()
    d = problem.desc()
    par = np.ascontiguousarray(par, dtype=np.float64)
    val = C.c_double()
    grad = np.zeros(problem.n_par_full)
    st = _QLIB.oracle_eval_quad(C.byref(d), par.ctypes.data_as(_dp), order, C.byref(val), grad.ctypes.data_as(_dp), fd_step)
    assert st == 0
    return (val.value, grad) if order >= 1 else val.value


_FLIB = None


def cpu_fast_eval(problem: Problem, par, threads: int = 1):
    """bench.py's CPU baseline (oracle/cpu_fast.cpp: analytic gradient, threads over tracks; constant-coefficient
    isotropic Kalman families only): data-term nllk and gradient over the full parameter vector."""
    global _FLIB
    from smoothsde_amd.capi import MODEL_CODES
    if _FLIB is None:
        path = os.path.join(_ORACLE_DIR, "libcpu_fast.so")
        if not os.path.exists(path):
            build_oracle()
        _FLIB = C.CDLL(path)
        _lp = C.POINTER(C.c_int64)
        _FLIB.cpu_fast_kalman.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, C.c_int64, C.c_int64, _lp, _lp, _dp, _dp, _dp, _dp,
                                          C.c_int, C.c_double, C.c_int, _dp]
        _FLIB.cpu_fast_kalman.restype = C.c_int
    pb = problem
    assert pb.model in ("CTCRW", "OU_SSM", "BM_SSM") and pb.n_re == 0 and all(x is None for x in pb.X_fe) and pb.H is None
    d = pb.n_dim
    par = np.ascontiguousarray(par, dtype=np.float64)
    theta = np.zeros(3 + d)
    theta[:len(par)] = par
    free = pb.par_fixed == 0
    mask = (1 if free[0] else 0) | (2 if free[1:1 + d].any() else 0) | (4 if free[1 + d] else 0) | \
           (8 if len(par) > 2 + d and free[2 + d] else 0)
    row0 = np.ascontiguousarray(pb.seg_start, dtype=np.int64)
    nrows = np.diff(np.append(pb.seg_start, pb.n)).astype(np.int64)
    p0 = np.array([1.0, 0.0, 10.0]) if pb.model == "CTCRW" else np.array([10.0, 0.0, 0.0])
    assert pb.P0 is None and pb.a0 is None
    inner = np.ones(pb.n - 1, dtype=bool)
    inner[pb.seg_start[1:] - 1] = False                    # intervals that span to the next track are never used
    dts = np.diff(pb.times)[inner]
    uni = len(dts) > 0 and np.all(dts == dts[0])
    out = np.zeros(4 + d)
    _lp = C.POINTER(C.c_int64)
    st = _FLIB.cpu_fast_kalman(MODEL_CODES[pb.model], d, mask, int(pb.na_mode == 1), pb.n, pb.n_seg, row0.ctypes.data_as(_lp),
                               nrows.ctypes.data_as(_lp), pb.times.ctypes.data_as(_dp), pb.obs.ctypes.data_as(_dp),
                               theta.ctypes.data_as(_dp), p0.ctypes.data_as(_dp), int(uni), float(dts[0]) if uni else 0.0,
                               threads, out.ctypes.data_as(_dp))
    if st != 0:
        raise RuntimeError(f"cpu_fast_kalman returned {st} (2 = this host CPU has no FMA)")
    grad = np.zeros(pb.n_par_full)
    grad[:len(par)] = out[1:1 + len(par)]
    grad[pb.par_fixed != 0] = 0.0
    return out[0], grad


def oracle_eval(problem: Problem, par, order: int = 1, threads: int = 1, report: bool = False,
                data_only: bool = False):
    """nllk (+ penalty unless data_only), gradient over the full parameter vector, and
    optionally aest_all (n x sdim)."""
    CALLS[0] += 1
    lib = load_oracle()
    d = problem.desc()
    par = np.ascontiguousarray(par, dtype=np.float64)
    assert lib.oracle_n_par_full(C.byref(d)) == problem.n_par_full == par.size
    val = C.c_double()
    grad = np.zeros(problem.n_par_full)
    aest = None
    if data_only:
        st = lib.oracle_eval_data(C.byref(d), par.ctypes.data_as(_dp), order, C.byref(val),
                                  grad.ctypes.data_as(_dp), threads)
    else:
        ap = None
        if report:
            aest = np.zeros((problem.n, problem.sdim), order="F")
            ap = aest.ctypes.data_as(_dp)
        st = lib.oracle_eval(C.byref(d), par.ctypes.data_as(_dp), order, C.byref(val),
                             grad.ctypes.data_as(_dp), ap, threads)
    assert st == 0
    out = [val.value]
    if order >= 1:
        out.append(grad)
    if report:
        out.append(aest)
    return out[0] if len(out) == 1 else tuple(out)
