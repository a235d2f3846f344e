"""ctypes loader of the test-only host build of the kernel arithmetic (tests/hostsim)."""
import ctypes as C
import os
import subprocess

import numpy as np

_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "hostsim")
_LIB = None
_dp = C.POINTER(C.c_double)
_lp = C.POINTER(C.c_int64)


def load():
    global _LIB
    if _LIB is None:
        alt = os.environ.get("SSDE_ORACLE_LIBDIR")           # tools/sanitize_cpu.sh: the sanitizer build
        if not alt:
            subprocess.run(["make", "-s", "-C", _DIR], check=True)
        lib = C.CDLL(os.path.join(alt or _DIR, "libhostsim.so"))
        lib.hostsim_kalman_iso.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, C.c_int64, C.c_int64, _lp, _lp,
                                           _dp, _dp, _dp, _dp, _dp]
        lib.hostsim_kalman_iso.restype = C.c_int
        lib.hostsim_direct.argtypes = [C.c_int] + [C.c_double] * 6 + [_dp]
        lib.hostsim_direct.restype = C.c_double
        _ipt = C.POINTER(C.c_int)
        lib.hostsim_kalman_tv.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int64, C.c_int64, _lp, _lp, _dp, _dp, _dp,
                                          C.c_int, C.c_int, _ipt, _ipt, _dp, C.c_double, _dp, _dp, _dp]
        lib.hostsim_kalman_tv.restype = C.c_int
        lib.hostsim_kalman_adj.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int64, C.c_int64, _lp, _lp, _dp, _dp, _dp,
                                           C.c_int, C.c_double, _dp, _dp, _dp, _dp]
        lib.hostsim_kalman_adj.restype = C.c_int
        lib.hostsim_kalman_adj_full.argtypes = [C.c_int, C.c_int, C.c_int64, C.c_int64, _lp, _lp, _dp, _dp, _dp,
                                                C.c_int, C.c_double, _dp, _dp, _dp, _dp, _dp]
        lib.hostsim_kalman_adj_full.restype = C.c_int
        _LIB = lib
    return _LIB


def kalman_adj_full(pb, par):
    """The reverse sweep on the FULL-covariance lanes (csrc/ssde_adj.hpp: AdjFull -- two response columns, per-row H_array or
    sigma_obs^2 I, any P0): nllk (data term) and gradient, the coefficient gradient formed here as X' G."""
    from smoothsde_amd.capi import MODEL_CODES
    lib = load()
    d, q, n = pb.n_dim, pb.q, pb.n
    assert d == 2
    par = np.asarray(par, dtype=np.float64)
    parmat = np.zeros((n, q), order="F")
    blocks = []
    for j in range(q):
        for src, off, nc in ((pb.X_fe[j], pb.off_fe + pb.fe_off[j], pb.ncol_fe[j]),
                             (pb.X_re[j], pb.off_re + pb.re_off[j], pb.ncol_re[j])):
            for c in range(nc):
                col = np.ones(n) if src is None else src[:, c]
                parmat[:, j] += col * par[off + c]
                blocks.append((j, off + c, col))
    row0 = np.ascontiguousarray(pb.seg_start, dtype=np.int64)
    nrows = np.diff(np.append(pb.seg_start, n)).astype(np.int64)
    sd = 4 if pb.model == "CTCRW" else 2
    if pb.P0 is None:
        P0 = np.diag([1.0, 10.0, 1.0, 10.0]) if pb.model == "CTCRW" else np.diag([10.0, 10.0])
    else:
        P0 = np.asarray(pb.P0, dtype=float)
    p0f = np.asfortranarray(P0).ravel(order="F")
    a0 = None if pb.a0 is None else np.ascontiguousarray(pb.a0)
    harr = None if pb.H is None else np.ascontiguousarray(np.transpose(np.asarray(pb.H, dtype=float), (2, 1, 0)).reshape(n, 4))   # [i][col][row]: column-major d x d per row
    out = np.zeros(2)
    G = np.zeros((n, q), order="F")
    st = lib.hostsim_kalman_adj_full(MODEL_CODES[pb.model], int(pb.na_mode == 1), n, pb.n_seg, row0.ctypes.data_as(_lp),
                                     nrows.ctypes.data_as(_lp), pb.times.ctypes.data_as(_dp), pb.obs.ctypes.data_as(_dp),
                                     parmat.ctypes.data_as(_dp), q, float(par[0]), p0f.ctypes.data_as(_dp),
                                     None if a0 is None else a0.ctypes.data_as(_dp), None if harr is None else harr.ctypes.data_as(_dp),
                                     out.ctypes.data_as(_dp), G.ctypes.data_as(_dp))
    assert st == 0
    grad = np.zeros(pb.n_par_full)
    grad[0] = out[1]
    for j, pidx, col in blocks:
        grad[pidx] += float(col @ G[:, j])
    return out[0], grad


def kalman_adj(pb, par):
    """The same nllk (data term) + gradient by the REVERSE sweep of csrc/ssde_adj.hpp: one forward pass leaving a record
    per row, one backward pass giving d nllk / d par_mat(i, j); the coefficient gradient is X' G, formed here."""
    from smoothsde_amd.capi import MODEL_CODES
    lib = load()
    d, q, n = pb.n_dim, pb.q, pb.n
    par = np.asarray(par, dtype=np.float64)
    parmat = np.zeros((n, q), order="F")
    blocks = []
    for j in range(q):
        for src, off, nc in ((pb.X_fe[j], pb.off_fe + pb.fe_off[j], pb.ncol_fe[j]),
                             (pb.X_re[j], pb.off_re + pb.re_off[j], pb.ncol_re[j])):
            for c in range(nc):
                col = np.ones(n) if src is None else src[:, c]
                parmat[:, j] += col * par[off + c]
                blocks.append((j, off + c, col))
    row0 = np.ascontiguousarray(pb.seg_start, dtype=np.int64)
    nrows = np.diff(np.append(pb.seg_start, n)).astype(np.int64)
    if pb.model == "CTCRW":
        p0 = np.array([1.0, 0.0, 10.0]) if pb.P0 is None else np.array([pb.P0[0, 0], pb.P0[0, 1], pb.P0[1, 1]])
    else:
        p0 = np.array([10.0, 0, 0]) if pb.P0 is None else np.array([pb.P0[0, 0], 0, 0])
    a0 = None if pb.a0 is None else np.ascontiguousarray(pb.a0)
    out = np.zeros(2)
    G = np.zeros((n, q), order="F")
    st = lib.hostsim_kalman_adj(MODEL_CODES[pb.model], d, int(pb.na_mode == 1), n, pb.n_seg, row0.ctypes.data_as(_lp),
                                nrows.ctypes.data_as(_lp), pb.times.ctypes.data_as(_dp), pb.obs.ctypes.data_as(_dp),
                                parmat.ctypes.data_as(_dp), q, float(par[0]), p0.ctypes.data_as(_dp),
                                None if a0 is None else a0.ctypes.data_as(_dp), out.ctypes.data_as(_dp), G.ctypes.data_as(_dp))
    assert st == 0
    grad = np.zeros(pb.n_par_full)
    grad[0] = out[1]
    for j, pidx, col in blocks:
        grad[pidx] += float(col @ G[:, j])
    return out[0], grad


def kalman_tv(pb, par):
    """Row-varying-coefficient isotropic Kalman nllk (data term) + gradient over the full parameter
    vector by the arithmetic of csrc/ssde_tv.hpp: linear predictors formed here, one lane per direction."""
    from smoothsde_amd.capi import MODEL_CODES
    lib = load()
    d, q, n = pb.n_dim, pb.q, pb.n
    par = np.asarray(par, dtype=np.float64)
    parmat = np.zeros((n, q), order="F")
    kinds, dims, pidx, wcols = [1], [0], [0], [np.ones(n)]          # log_sigma_obs
    for j in range(q):
        kind, dim = (2, j) if j < d else (3 + (j - d), 0)
        for src, off, nc in ((pb.X_fe[j], pb.off_fe + pb.fe_off[j], pb.ncol_fe[j]),
                             (pb.X_re[j], pb.off_re + pb.re_off[j], pb.ncol_re[j])):
            for c in range(nc):
                col = np.ones(n) if src is None else src[:, c]
                parmat[:, j] += col * par[off + c]
                kinds.append(kind); dims.append(dim); pidx.append(off + c); wcols.append(col)
    nd = len(kinds)
    wmat = np.ascontiguousarray(np.column_stack(wcols))
    row0 = np.ascontiguousarray(pb.seg_start, dtype=np.int64)
    nrows = np.diff(np.append(pb.seg_start, n)).astype(np.int64)
    if pb.model == "CTCRW":
        p0 = np.array([1.0, 0.0, 10.0]) if pb.P0 is None else np.array([pb.P0[0, 0], pb.P0[0, 1], pb.P0[1, 1]])
    else:
        p0 = np.array([10.0, 0, 0]) if pb.P0 is None else np.array([pb.P0[0, 0], 0, 0])
    a0 = None if pb.a0 is None else np.ascontiguousarray(pb.a0)
    ki, di = np.asarray(kinds, dtype=np.int32), np.asarray(dims, dtype=np.int32)
    out = np.zeros(1 + nd)
    _ipt = C.POINTER(C.c_int)
    st = lib.hostsim_kalman_tv(MODEL_CODES[pb.model], d, int(pb.na_mode == 1), n, pb.n_seg, row0.ctypes.data_as(_lp),
                               nrows.ctypes.data_as(_lp), pb.times.ctypes.data_as(_dp), pb.obs.ctypes.data_as(_dp),
                               parmat.ctypes.data_as(_dp), q, nd, ki.ctypes.data_as(_ipt), di.ctypes.data_as(_ipt),
                               wmat.ctypes.data_as(_dp), float(par[0]), p0.ctypes.data_as(_dp),
                               None if a0 is None else a0.ctypes.data_as(_dp), out.ctypes.data_as(_dp))
    assert st == 0
    grad = np.zeros(pb.n_par_full)
    grad[np.asarray(pidx)] = out[1:]
    return out[0], grad


def kalman_iso(pb, par, mask):
    """Constant-coefficient isotropic Kalman nllk + gradient by the kernel arithmetic."""
    from smoothsde_amd.capi import MODEL_CODES
    lib = load()
    d = pb.n_dim
    row0 = np.ascontiguousarray(pb.seg_start, dtype=np.int64)
    nrows = np.diff(np.append(pb.seg_start, pb.n)).astype(np.int64)
    theta = np.zeros(3 + d)
    theta[:len(par)] = par
    if pb.model == "CTCRW":
        p0 = np.array([1.0, 0.0, 10.0]) if pb.P0 is None else np.array([pb.P0[0, 0], pb.P0[0, 1], pb.P0[1, 1]])
    else:
        p0 = np.array([10.0, 0, 0]) if pb.P0 is None else np.array([pb.P0[0, 0], 0, 0])
    out = np.zeros(4 + d)
    st = lib.hostsim_kalman_iso(MODEL_CODES[pb.model], d, mask, int(pb.na_mode == 1), pb.n, pb.n_seg,
                                row0.ctypes.data_as(_lp), nrows.ctypes.data_as(_lp),
                                pb.times.ctypes.data_as(_dp), pb.obs.ctypes.data_as(_dp),
                                theta.ctypes.data_as(_dp), p0.ctypes.data_as(_dp), out.ctypes.data_as(_dp))
    assert st == 0
    return out[0], out[1:1 + pb.n_par_full]
