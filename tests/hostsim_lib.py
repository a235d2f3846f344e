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
        subprocess.run(["make", "-s", "-C", _DIR], check=True)
        lib = C.CDLL(os.path.join(_DIR, "libhostsim.so"))
        lib.hostsim_kalman_iso.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, C.c_int64, C.c_int64, _lp, _lp,
                                           _dp, _dp, _dp, _dp, _dp]
        lib.hostsim_kalman_iso.restype = C.c_int
        lib.hostsim_direct.argtypes = [C.c_int] + [C.c_double] * 6 + [_dp]
        lib.hostsim_direct.restype = C.c_double
        _LIB = lib
    return _LIB


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
