"""ctypes mirror of include/ssde.h and the loader of the HIP engine library.

The product path is `libssde_hip.so` (hand-written HIP, gfx950).  There is NO CPU
fallback: if the library is missing, or no GPU is visible when an engine is created,
this module raises.  (The CPU oracle under oracle/ is test infrastructure and is never
imported from here.)

`Problem` carries the arguments of the reference's `tmb_dat` list
(/root/reference/R/sde.R:528-536, 542-598) as numpy arrays and exposes them as an
`ssde_desc`.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional, Sequence

import numpy as np

ABI_VERSION = 10

MODEL_CODES = {"BM": 0, "OU": 1, "BM_SSM": 2, "OU_SSM": 3, "CTCRW": 4, "BM_t": 5, "ESEAL_SSM": 6, "CIR": 7}
KALMAN_MODELS = ("BM_SSM", "OU_SSM", "CTCRW")
# types the reference dispatches (src/smoothSDE.cpp:12-27) that this engine does not cover
UNSUPPORTED_MODELS = ()

NA_R_ONLY, NA_ANY_NAN = 0, 1
PATH_NAMES = {0: "direct", 1: "isotropic-register", 2: "dense", 3: "isotropic-row-varying"}
FLAG_DEVICE_DATA, FLAG_FORCE_DENSE, FLAG_NO_UNIFORM_DT, FLAG_EXACT_HESS = 0x1, 0x2, 0x4, 0x8

_dp = C.POINTER(C.c_double)
_ip = C.POINTER(C.c_int32)


class SsdePPBasis(C.Structure):
    _fields_ = [("x", C.c_void_p), ("n_knots", C.c_int32), ("n_cols", C.c_int32), ("knots", C.c_void_p),
                ("coef", C.c_void_p)]


class SsdeDesc(C.Structure):
    _fields_ = [
        ("abi_version", C.c_int32), ("model", C.c_int32), ("n_dim", C.c_int32), ("n_par", C.c_int32),
        ("n", C.c_int64),
        ("id", C.c_void_p), ("times", C.c_void_p), ("obs", C.c_void_p),
        ("ncol_fe", _ip), ("x_fe", C.POINTER(C.c_void_p)),
        ("ncol_re", _ip), ("x_re", C.POINTER(C.c_void_p)),
        ("n_smooth", C.c_int32), ("smooth_ncol", _ip), ("s_blocks", C.c_void_p),
        ("include_penalty", C.c_int32),
        ("n_seg", C.c_int64), ("a0", C.c_void_p), ("p0", C.c_void_p), ("h_array", C.c_void_p),
        ("par_fixed", C.c_void_p), ("na_mode", C.c_int32), ("device", C.c_int32),
        ("flags", C.c_uint32), ("reserved", C.c_uint32),
        ("other_data", C.c_void_p), ("n_other_data", C.c_int32),
        ("n_decay", C.c_int32), ("t_decay", C.c_void_p), ("n_decay_cols", C.c_int32), ("reserved3", C.c_int32),
        ("col_decay", C.c_void_p), ("ind_decay", C.c_void_p),
        ("eseal_h", C.c_void_p), ("eseal_R", C.c_void_p),
        ("basis_re", C.POINTER(C.POINTER(SsdePPBasis))),
        ("n_devices", C.c_int32), ("reserved4", C.c_int32), ("devices", _ip),
    ]


class SsdeLaplaceOpts(C.Structure):
    _fields_ = [("hess_step", C.c_double), ("fd_step", C.c_double), ("newton_tol", C.c_double),
                ("max_newton", C.c_int32), ("reserved", C.c_int32)]


class SsdeSimDesc(C.Structure):
    _fields_ = [("abi_version", C.c_int32), ("model", C.c_int32), ("n_dim", C.c_int32), ("n_steps", C.c_int32),
                ("track0", C.c_int64), ("n_tracks", C.c_int64), ("row0", C.c_void_p), ("n_rows", C.c_int64),
                ("row_offset", C.c_int64), ("mu", C.c_double * 8), ("z0", C.c_double * 8),
                ("tau", C.c_double), ("nu", C.c_double), ("kappa", C.c_double), ("sigma", C.c_double),
                ("sigma_obs", C.c_double), ("dt", C.c_double), ("seed", C.c_uint64), ("device", C.c_int32),
                ("reserved", C.c_int32)]


OPT_KERNEL_STAMPS, OPT_COMM_DEFER = 1, 2
# ssde_info_t.kernel_id (include/ssde.h: SSDE_KERNEL_*): the kernel family that ran the rows of the last evaluation
KERNEL_NAMES = {0: "none", 1: "direct_kernel", 2: "direct_fast_kernel", 3: "iso_shared_kernel", 4: "iso_mask_kernel",
                5: "iso_mask_kernel<uniform grid>", 6: "iso_quiet_kernel", 7: "iso_shared_kernel + general kernel (mixed batch)",
                8: "iso_kernel (direction parts)", 9: "iso_drift_kernel", 10: "iso_drift_general_kernel", 11: "iso_colvar_kernel",
                12: "iso_few_kernel", 13: "iso_full_kernel", 14: "dense_kernel", 15: "tv_filter_kernel", 16: "tv_filter_kernel<dense lanes>",
                17: "iso_adj_kernel"}
PHASE_NAMES = ("host_total", "host_enqueue", "gpu_pre", "kernel", "finalize", "allreduce", "readback", "reserved")


class SsdeInfo(C.Structure):
    _fields_ = [
        ("n_par_full", C.c_int32), ("n_free", C.c_int32), ("sdim", C.c_int32), ("path", C.c_int32),
        ("const_coeff", C.c_int32), ("uniform_dt", C.c_int32),
        ("n_tracks", C.c_int64), ("n_rows", C.c_int64), ("n_steps", C.c_int64), ("hbm_bytes", C.c_int64),
        ("algo_bytes_per_row", C.c_double), ("n_kernel_blocks", C.c_int32), ("lanes_per_track", C.c_int32),
        ("window", C.c_int32), ("window_retries", C.c_int32), ("window_check", C.c_double),
        ("main_kernel_ms", C.c_double), ("main_kernel_rows", C.c_int64),
        ("required_bytes_per_row", C.c_double), ("n_evals", C.c_int64), ("n_memo_hits", C.c_int64),
        ("n_devices", C.c_int32), ("comm_ranks", C.c_int32), ("window_check_max", C.c_double),
        ("n_rows_tiled", C.c_int64), ("n_groups", C.c_int32), ("n_clean_groups", C.c_int32),
        ("quiet_window", C.c_int32), ("kernel_id", C.c_int32), ("quiet_share", C.c_double),
        ("comm_ranks_reported", C.c_int32), ("exact_hess_scope", C.c_int32),
    ]

    def as_dict(self):
        return {f: getattr(self, f) for f, _ in self._fields_}


def na_real() -> float:
    """R's NA_real_: the NaN whose low word is 1954 (what R_IsNA tests)."""
    return float(np.array([0x7FF00000000007A2], dtype=np.uint64).view(np.float64)[0])


def _f64(a, order="F"):
    return np.require(np.asarray(a, dtype=np.float64), requirements=["ALIGNED"] + (["F"] if order == "F" else ["C"]))


def n_sde_par(model: str, n_dim: int) -> int:
    return n_dim + 1 if model in ("BM", "BM_SSM", "BM_t", "ESEAL_SSM") else n_dim + 2


def state_dim(model: str, n_dim: int) -> int:
    if model == "CTCRW":
        return 2 * n_dim
    if model == "ESEAL_SSM":
        return 2
    return n_dim if model in KALMAN_MODELS else 0


class Problem:
    """Host-side container for one model's data, in the reference's argument shapes.

    ID      : (n,) any dtype; only neighbour inequality matters (nllk_ctcrw.hpp:196)
    times   : (n,)
    obs     : (n, d)
    X_fe    : list of q arrays (n, ncol_fe[j]) or None (= intercept-only)
    X_re    : list of q arrays (n, ncol_re[j]) or None
    S_list  : list of penalty blocks (one per smooth), smooth order = column order of X_re
    """

    def __init__(self, model: str, ID, times, obs, X_fe: Optional[Sequence] = None,
                 X_re: Optional[Sequence] = None, S_list: Optional[Sequence] = None,
                 a0=None, P0=None, H=None, par_fixed=None, include_penalty: int = 1,
                 na_mode: int = NA_ANY_NAN, device: int = -1, flags: int = 0, other_data=None,
                 t_decay=None, col_decay=None, ind_decay=None, eseal_h=None, eseal_R=None, basis_re=None):
        if model in UNSUPPORTED_MODELS:
            raise NotImplementedError(f"SDE type {model!r} is outside this engine's scope")
        if model not in MODEL_CODES:
            raise ValueError("Unknown SDE type")  # src/smoothSDE.cpp:25
        self.model = model
        obs = np.asarray(obs, dtype=np.float64)
        if obs.ndim == 1:
            obs = obs[:, None]
        self.n, self.n_dim = obs.shape
        self.q = n_sde_par(model, self.n_dim)
        self.sdim = state_dim(model, self.n_dim)
        self.obs = _f64(obs)
        ID = np.asarray(ID)
        if ID.dtype.kind not in "fiu":
            _, ID = np.unique(ID, return_inverse=True)
        self.id = _f64(ID)
        self.times = _f64(times)
        if self.id.shape != (self.n,) or self.times.shape != (self.n,):
            raise ValueError("ID, times and obs must have the same number of rows")

        # basis_re[j]: a synth.PPBasis -- the random-effect block of parameter j as a piecewise-cubic function of one
        # covariate (include/ssde.h: ssde_ppbasis).  The dense block it stands for is kept in X_re[j] as well: the
        # oracle and the generic kernels use it, the engine's fast direct kernel evaluates the table instead.
        self.basis_re = [None] * self.q if basis_re is None else list(basis_re)
        if any(b is not None for b in self.basis_re):
            X_re = [None] * self.q if X_re is None else list(X_re)
            for j, b in enumerate(self.basis_re):
                if b is not None:
                    X_re[j] = b.dense()
        self.X_fe, self.X_re = [], []
        ncol_fe, ncol_re = [], []
        for j in range(self.q):
            xf = None if X_fe is None else X_fe[j]
            if xf is None:
                self.X_fe.append(None)
                ncol_fe.append(1)
            else:
                xf = _f64(np.asarray(xf, dtype=np.float64).reshape(self.n, -1))
                self.X_fe.append(xf)
                ncol_fe.append(xf.shape[1])
            xr = None if X_re is None else X_re[j]
            if xr is None or np.asarray(xr).size == 0:
                self.X_re.append(None)
                ncol_re.append(0)
            else:
                xr = _f64(np.asarray(xr, dtype=np.float64).reshape(self.n, -1))
                self.X_re.append(xr)
                ncol_re.append(xr.shape[1])
        self.ncol_fe = np.asarray(ncol_fe, dtype=np.int32)
        self.ncol_re = np.asarray(ncol_re, dtype=np.int32)
        self.n_fe, self.n_re = int(self.ncol_fe.sum()), int(self.ncol_re.sum())

        S_list = [] if S_list is None else [np.asarray(s, dtype=np.float64) for s in S_list]
        self.S_list = S_list
        self.smooth_ncol = np.asarray([s.shape[0] for s in S_list], dtype=np.int32)
        if int(self.smooth_ncol.sum()) != self.n_re:
            raise ValueError("penalty blocks do not match the random-effect columns")
        self.s_blocks = _f64(np.concatenate([s.flatten(order="F") for s in S_list])) if S_list else None
        self.n_smooth = len(S_list)
        self.include_penalty = int(include_penalty)

        self.kalman = model in KALMAN_MODELS or model == "ESEAL_SSM"     # Kalman-style penalty (no constants)
        # leading scalar parameters of the template: log_sigma_obs (nllk_ctcrw.hpp:135) or log_tau, a1, log_a2
        # (nllk_e_seal_ssm.hpp:114-116)
        self.lead_names = ["log_tau", "a1", "log_a2"] if model == "ESEAL_SSM" else (["log_sigma_obs"] if self.kalman else [])
        self.eseal_h = self.eseal_R = None
        if model == "ESEAL_SSM":
            if self.n_dim != 1:
                raise ValueError("ESEAL_SSM takes one response variable")
            if eseal_h is None or eseal_R is None or a0 is None:
                raise ValueError("ESEAL_SSM needs h (daily drift dives), R (non-lipid tissue mass) and a0 (R/sde.R:599-614)")
            self.eseal_h, self.eseal_R = _f64(eseal_h), _f64(eseal_R)
            if self.eseal_h.shape != (self.n,) or self.eseal_R.shape != (self.n,):
                raise ValueError("h and R must have one entry per row")
        first = np.ones(self.n, dtype=bool)
        first[1:] = self.id[1:] != self.id[:-1]
        self.seg_start = np.flatnonzero(first)
        self.n_seg = len(self.seg_start)
        self.a0 = None if a0 is None else _f64(np.asarray(a0, dtype=np.float64).reshape(self.n_seg, self.sdim))
        self.P0 = None if P0 is None else _f64(np.asarray(P0, dtype=np.float64).reshape(self.sdim, self.sdim))
        if H is not None:
            H = np.asarray(H, dtype=np.float64)
            if H.shape != (self.n_dim, self.n_dim, self.n):
                raise ValueError("H must be an array of shape (d, d, n)")
            H = _f64(H)
        self.H = H

        # decaying random-effect columns (direct families; nllk_sde.hpp:30-32, 47-58; R/sde.R:163-177):
        # col_decay / ind_decay are 0-based here (the R objects are 1-based)
        self.n_decay, self.t_decay, self.col_decay, self.ind_decay = 0, None, None, None
        self.decay_of_col = np.full(self.n_re, -1, dtype=int)
        if t_decay is not None:
            if self.kalman:
                raise ValueError("decaying terms are a feature of the direct families (BM, BM_t, OU)")
            self.t_decay = _f64(np.asarray(t_decay, dtype=np.float64).ravel())
            if self.t_decay.shape != (self.q * self.n,):
                raise ValueError("'t_decay' should be of length (number of parameters) x (number of data)")  # R/sde.R:170-173
            self.col_decay = np.ascontiguousarray(col_decay, dtype=np.int32)
            self.ind_decay = np.ascontiguousarray(ind_decay, dtype=np.int32)
            if self.col_decay.shape != self.ind_decay.shape:
                raise ValueError("Check length of 'ind_decay' and 'col_decay'")                               # R/sde.R:174-176
            if len(self.col_decay) and (self.col_decay.min() < 0 or self.col_decay.max() >= self.n_re):
                raise ValueError(f"'col_decay' should be between 0 and {self.n_re - 1}")                      # R/sde.R:637-640
            self.n_decay = int(self.ind_decay.max()) + 1 if len(self.ind_decay) else 0
            self.decay_of_col[self.col_decay] = self.ind_decay

        # full parameter vector layout (include/ssde.h)
        o = len(self.lead_names)
        self.off_sigobs = 0 if self.lead_names == ["log_sigma_obs"] else None
        self.off_fe = o
        o += self.n_fe
        self.off_lambda = o
        o += self.n_smooth
        self.off_decay = o
        o += self.n_decay
        self.off_re = o
        o += self.n_re
        self.n_par_full = o
        self.fe_off = np.concatenate([[0], np.cumsum(self.ncol_fe)[:-1]]).astype(int)
        self.re_off = np.concatenate([[0], np.cumsum(self.ncol_re)[:-1]]).astype(int)

        fixed = np.zeros(self.n_par_full, dtype=np.uint8)
        if par_fixed is not None:
            fixed[:] = np.asarray(par_fixed, dtype=np.uint8)
        if self.off_sigobs is not None and self.H is not None:
            fixed[0] = 1  # map log_sigma_obs = NA when H is supplied (R/sde.R:565, 595)
        self.par_fixed = fixed
        self.na_mode, self.device, self.flags = int(na_mode), int(device), int(flags)
        # DATA_VECTOR(other_data): the degrees of freedom of BM_t (R/sde.R:539-541)
        self.other_data = None if other_data is None else _f64(np.atleast_1d(np.asarray(other_data, dtype=np.float64)))
        if model == "BM_t":
            if self.n_dim != 1:
                raise ValueError("BM_t takes one response variable")
            if self.other_data is None or not (self.other_data[0] > 2):
                raise ValueError("BM_t needs other_data = df (degrees of freedom > 2)")
        self._keep = []

    @classmethod
    def from_torch(cls, model: str, ID, times, obs, par_fixed=None, na_mode: int = NA_ANY_NAN, flags: int = 0,
                   X_re=None, S_list=None, basis_re=None, H=None):
        """Problem whose data already live in HBM (torch CUDA tensors, fp64): the engine re-tiles /
        copies them on the device instead of uploading (SSDE_FLAG_DEVICE_DATA).  Fixed effects are
        intercept-only; `X_re[j]` may be an (n, k) CUDA tensor of streamed design columns for SDE
        parameter j, with the penalty blocks in `S_list` (numpy); `H`: per-row measurement covariances as a (d, d, n)
        CUDA tensor (H_array, Kalman families)."""
        import torch
        assert ID.is_cuda and times.is_cuda and obs.is_cuda
        self = cls.__new__(cls)
        if model not in MODEL_CODES:
            raise ValueError("Unknown SDE type")
        if model in ("ESEAL_SSM", "BM_t", "CIR"):
            raise NotImplementedError("device-resident construction covers BM, OU, BM_SSM, OU_SSM, CTCRW")
        self.model = model
        if obs.dim() == 1:
            obs = obs[:, None]
        self.n, self.n_dim = int(obs.shape[0]), int(obs.shape[1])
        self.q = n_sde_par(model, self.n_dim)
        self.sdim = state_dim(model, self.n_dim)
        self._t_id = ID.to(torch.float64).contiguous()
        self._t_times = times.to(torch.float64).contiguous()
        self._t_obs = obs.to(torch.float64).t().contiguous()     # (d, n) row-major == (n, d) column-major
        self.id = self.times = self.obs = None
        self.X_fe, self.X_re = [None] * self.q, [None] * self.q
        self.ncol_fe = np.ones(self.q, dtype=np.int32)
        self.ncol_re = np.zeros(self.q, dtype=np.int32)
        self.n_fe, self.n_re = self.q, 0
        self.S_list, self.smooth_ncol, self.s_blocks, self.n_smooth = [], np.zeros(0, dtype=np.int32), None, 0
        self._t_xre = [None] * self.q
        # basis_re[j]: synth.PPBasis whose covariate x is a CUDA tensor -- no n x K block exists anywhere
        self.basis_re = [None] * self.q if basis_re is None else list(basis_re)
        if X_re is None and any(b is not None for b in self.basis_re):
            X_re = [None] * self.q
        if X_re is not None:
            for j in range(self.q):
                if self.basis_re[j] is not None:
                    self.ncol_re[j] = self.basis_re[j].n_cols
                elif X_re[j] is not None:
                    xt = X_re[j].to(torch.float64)
                    self._t_xre[j] = xt.t().contiguous()          # (k, n) row-major == (n, k) column-major
                    self.ncol_re[j] = xt.shape[1]
            self.n_re = int(self.ncol_re.sum())
            self.S_list = [np.asarray(s_, dtype=np.float64) for s_ in (S_list or [])]
            self.smooth_ncol = np.asarray([s_.shape[0] for s_ in self.S_list], dtype=np.int32)
            self.s_blocks = _f64(np.concatenate([s_.flatten(order="F") for s_ in self.S_list])) if self.S_list else None
            self.n_smooth = len(self.S_list)
            if int(self.smooth_ncol.sum()) != self.n_re:
                raise ValueError("penalty blocks do not match the random-effect columns")
        self.include_penalty = 1
        self.kalman = model in KALMAN_MODELS
        self.lead_names = ["log_sigma_obs"] if self.kalman else []
        self.eseal_h = self.eseal_R = None
        first = torch.ones(self.n, dtype=torch.bool, device=ID.device)
        first[1:] = self._t_id[1:] != self._t_id[:-1]
        self.seg_start = first.nonzero().flatten().cpu().numpy()
        self.n_seg = len(self.seg_start)
        self.a0 = self.P0 = self.H = None
        o = 0
        self.off_sigobs = None
        if self.kalman:
            self.off_sigobs, o = 0, 1
        self.off_fe = o
        o += self.n_fe
        self.off_lambda = o
        o += self.n_smooth
        self.off_decay, self.n_decay, self.t_decay, self.col_decay, self.ind_decay = o, 0, None, None, None
        self.off_re = o
        o += self.n_re
        self.n_par_full = o
        self.decay_of_col = np.full(self.n_re, -1, dtype=int)
        self.fe_off = np.arange(self.q)
        self.re_off = np.concatenate([[0], np.cumsum(self.ncol_re)[:-1]]).astype(int)
        fixed = np.zeros(self.n_par_full, dtype=np.uint8)
        if par_fixed is not None:
            fixed[:] = np.asarray(par_fixed, dtype=np.uint8)
        self._t_h = None
        if H is not None:
            if not self.kalman or tuple(H.shape) != (self.n_dim, self.n_dim, self.n):
                raise ValueError("H must be a (d, d, n) tensor, Kalman families only")
            self._t_h = H.to(torch.float64).permute(2, 1, 0).contiguous()     # memory order of the column-major d x d x n array
            fixed[0] = 1                                                       # log_sigma_obs is not in the model then (R/sde.R:565, 595)
        self.par_fixed = fixed
        self.na_mode, self.device = int(na_mode), int(ID.device.index or 0)
        self.flags = int(flags) | FLAG_DEVICE_DATA
        self.other_data = None
        self._keep = []
        return self

    # -- parameter helpers ---------------------------------------------------------------
    def par_names(self):
        names = list(self.lead_names)
        for j in range(self.q):
            names += [f"coeff_fe[{j}][{c}]" for c in range(self.ncol_fe[j])]
        names += [f"log_lambda[{s}]" for s in range(self.n_smooth)]
        names += [f"log_decay[{k}]" for k in range(self.n_decay)]
        for j in range(self.q):
            names += [f"coeff_re[{j}][{c}]" for c in range(self.ncol_re[j])]
        return names

    def free_index(self):
        return np.flatnonzero(self.par_fixed == 0)

    # -- ctypes view ------------------------------------------------------------------------
    def desc(self) -> SsdeDesc:
        d = SsdeDesc()
        keep = self._keep = []

        def ptr(a):
            if a is None:
                return None
            keep.append(a)
            return a.ctypes.data

        d.abi_version = ABI_VERSION
        d.model = MODEL_CODES[self.model]
        d.n_dim, d.n_par, d.n = self.n_dim, self.q, self.n
        if getattr(self, "_t_id", None) is not None:
            d.id, d.times, d.obs = self._t_id.data_ptr(), self._t_times.data_ptr(), self._t_obs.data_ptr()
        else:
            d.id, d.times, d.obs = ptr(self.id), ptr(self.times), ptr(self.obs)
        d.ncol_fe = self.ncol_fe.ctypes.data_as(_ip)
        d.ncol_re = self.ncol_re.ctypes.data_as(_ip)
        xfe = (C.c_void_p * self.q)(*[ptr(x) for x in self.X_fe])
        if getattr(self, "_t_xre", None) is not None:
            xre = (C.c_void_p * self.q)(*[None if x is None else x.data_ptr() for x in self._t_xre])
        else:
            xre = (C.c_void_p * self.q)(*[ptr(x) for x in self.X_re])
        keep += [xfe, xre]
        d.x_fe, d.x_re = xfe, xre
        d.n_smooth = self.n_smooth
        d.smooth_ncol = self.smooth_ncol.ctypes.data_as(_ip) if self.n_smooth else None
        d.s_blocks = ptr(self.s_blocks)
        d.include_penalty = self.include_penalty
        d.n_seg = self.n_seg
        d.a0, d.p0, d.h_array = ptr(self.a0), ptr(self.P0), ptr(self.H)
        if getattr(self, "_t_h", None) is not None:
            d.h_array = self._t_h.data_ptr()
        d.par_fixed = ptr(self.par_fixed)
        d.na_mode, d.device, d.flags = self.na_mode, self.device, self.flags
        d.other_data = ptr(getattr(self, "other_data", None))
        d.n_other_data = 0 if getattr(self, "other_data", None) is None else len(self.other_data)
        d.n_decay = self.n_decay
        if self.n_decay > 0:
            d.t_decay, d.n_decay_cols = ptr(self.t_decay), len(self.col_decay)
            d.col_decay, d.ind_decay = ptr(self.col_decay), ptr(self.ind_decay)
        d.eseal_h, d.eseal_R = ptr(getattr(self, "eseal_h", None)), ptr(getattr(self, "eseal_R", None))
        bl = getattr(self, "basis_re", None)
        if bl is not None and any(b is not None for b in bl):
            arr = (C.POINTER(SsdePPBasis) * self.q)()
            for j, b in enumerate(bl):
                if b is None:
                    continue
                sb = SsdePPBasis()
                if hasattr(b.x, "data_ptr"):
                    sb.x = b.x.data_ptr()
                    keep.append(b.x)
                else:
                    sb.x = ptr(_f64(b.x))
                sb.n_knots, sb.n_cols = len(b.knots), b.n_cols
                sb.knots, sb.coef = ptr(b.knots), ptr(b.coef)
                keep.append(sb)
                arr[j] = C.pointer(sb)
            keep.append(arr)
            d.basis_re = arr
        return d


# ---------------------------------------------------------------------------------------------
_LIB = None


def lib_path() -> str:
    override = os.environ.get("SSDE_LIB")  # kernel-tuning builds (same ABI); the default is the in-tree library
    if override:
        return override
    return os.path.join(os.path.dirname(os.path.abspath(__file__)), "lib", "libssde_hip.so")


def load_library():
    """Load libssde_hip.so; raise loudly if the HIP extension has not been built."""
    global _LIB
    if _LIB is not None:
        return _LIB
    path = lib_path()
    if not os.path.exists(path):
        raise RuntimeError(
            f"HIP engine library not found at {path}: build it with "
            "`python -c 'import __graft_entry__ as g; g.build()'` (hipcc, gfx950). "
            "There is no CPU fallback.")
    # One HIP runtime per process: PyTorch-ROCm wheels bundle their own libamdhip64 / libhsa-runtime64
    # under an unversioned file name, so if this library pulled in /opt/rocm's copy first, torch
    # would later load a SECOND runtime that finds no device.  Importing torch first makes the
    # loader resolve our DT_NEEDED libamdhip64.so.7 to the runtime torch already mapped (same
    # soname).  Hosts without torch (the R shim) simply use /opt/rocm's runtime.
    try:
        import torch  # noqa: F401
    except Exception:
        pass
    lib = C.CDLL(path)
    lib.ssde_create.argtypes = [C.POINTER(SsdeDesc), C.POINTER(C.c_void_p)]
    lib.ssde_create.restype = C.c_int
    lib.ssde_eval.argtypes = [C.c_void_p, _dp, C.c_int32, C.c_int32, _dp, _dp]
    lib.ssde_eval.restype = C.c_int
    lib.ssde_eval_device.argtypes = [C.c_void_p, _dp, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p]
    lib.ssde_eval_device.restype = C.c_int
    lib.ssde_penalty.argtypes = [C.c_void_p, _dp, C.c_int32, _dp, _dp]
    lib.ssde_penalty.restype = C.c_int
    lib.ssde_report.argtypes = [C.c_void_p, _dp, C.c_int32, _dp]
    lib.ssde_report.restype = C.c_int
    lib.ssde_widen_windows.argtypes = [C.c_void_p, C.c_int32]
    lib.ssde_widen_windows.restype = C.c_int
    lib.ssde_relax_windows.argtypes = [C.c_void_p]
    lib.ssde_relax_windows.restype = C.c_int
    lib.ssde_info.argtypes = [C.c_void_p, C.POINTER(SsdeInfo)]
    lib.ssde_info.restype = C.c_int
    lib.ssde_destroy.argtypes = [C.c_void_p]
    lib.ssde_destroy.restype = None
    lib.ssde_last_error.argtypes = [C.c_void_p]
    lib.ssde_last_error.restype = C.c_char_p
    lib.ssde_abi_version.argtypes = []
    lib.ssde_abi_version.restype = C.c_int
    lib.ssde_laplace_eval.argtypes = [C.c_void_p, _dp, C.c_int32, C.c_int32, _dp, _dp, _dp, C.POINTER(SsdeLaplaceOpts)]
    lib.ssde_laplace_eval.restype = C.c_int
    lib.ssde_last_kernel_ms.argtypes = [C.c_void_p]
    lib.ssde_last_kernel_ms.restype = C.c_double
    lib.ssde_kernel_ms_history.argtypes = [C.c_void_p, _dp, C.c_int32]
    lib.ssde_kernel_ms_history.restype = C.c_int
    lib.ssde_forget.argtypes = [C.c_void_p]
    lib.ssde_forget.restype = C.c_int
    lib.ssde_comm_unique_id.argtypes = [C.c_void_p]
    lib.ssde_comm_unique_id.restype = C.c_int
    lib.ssde_comm_init_rank.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.c_void_p]
    lib.ssde_comm_init_rank.restype = C.c_int
    lib.ssde_simulate.argtypes = [C.POINTER(SsdeSimDesc), C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    lib.ssde_simulate.restype = C.c_int
    lib.ssde_hess.argtypes = [C.c_void_p, _dp, C.c_int32, _ip, C.c_int32, _dp]
    lib.ssde_hess.restype = C.c_int
    lib.ssde_set_option.argtypes = [C.c_void_p, C.c_int32, C.c_int64]
    lib.ssde_set_option.restype = C.c_int
    lib.ssde_last_phase_ms.argtypes = [C.c_void_p, _dp]
    lib.ssde_last_phase_ms.restype = C.c_int
    lib.ssde_comm_allreduce.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p]
    lib.ssde_comm_allreduce.restype = C.c_int
    if lib.ssde_abi_version() != ABI_VERSION:
        raise RuntimeError("libssde_hip.so ABI version mismatch")
    _LIB = lib
    return lib


class EngineError(RuntimeError):
    """a non-zero status from the C ABI; `.status` holds it (include/ssde.h: SSDE_ERR_*)"""
    status = None


WINDOW_TOL = 1e-11  # largest tolerated relative hand-over disagreement between time windows

EXPORTED_SYMBOLS = ("ssde_create", "ssde_eval", "ssde_eval_device", "ssde_penalty", "ssde_report", "ssde_widen_windows", "ssde_relax_windows",
                    "ssde_info", "ssde_destroy", "ssde_last_error", "ssde_abi_version", "ssde_comm_unique_id", "ssde_comm_init_rank", "ssde_forget", "ssde_laplace_eval", "ssde_last_kernel_ms", "ssde_kernel_ms_history",
                    "ssde_simulate", "ssde_set_option", "ssde_hess", "ssde_last_phase_ms", "ssde_comm_allreduce")

COMM_ID_BYTES = 128


def comm_unique_id() -> bytes:
    """An ncclUniqueId (rank 0 makes it, the host ships it to the other ranks, every rank passes it to
    Engine.comm_init)."""
    lib = load_library()
    buf = C.create_string_buffer(COMM_ID_BYTES)
    st = lib.ssde_comm_unique_id(buf)
    if st != 0:
        msg = lib.ssde_last_error(None)
        raise EngineError(f"ssde_comm_unique_id failed ({st}): {msg.decode() if msg else ''}")
    return buf.raw


def simulate_device(model: str, n_tracks: int, n_steps: int, n_dim: int = 2, *, mu=0.0, tau=2.0, nu=1.0, kappa=1.0,
                    sigma=1.0, sigma_obs=0.1, dt: float = 1.0, z0=0.0, seed: int = 1, track0: int = 0, row_offset=None,
                    lengths=None, device=None, want_id_times: bool = True):
    """ssde_simulate: tracks [track0, track0 + n_tracks) of the batch `seed` names, generated in HBM by the HIP
    simulator (exact transitions of R/sde.R:1434-1478 + observation error; counter-based, so a shard of a batch is the
    same numbers whoever generates it).  Returns torch CUDA tensors (ID, times, obs) with obs of shape (n, d) (a view of
    the column-major buffer the engine takes as it is).  `lengths`: per-track row counts (ragged batch, <= n_steps).
    Raises without a GPU: there is no CPU fallback."""
    import torch
    lib = load_library()
    if model not in MODEL_CODES:
        raise ValueError("Unknown SDE type")
    dev = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
    M, T, d = int(n_tracks), int(n_steps), int(n_dim)
    row0 = None
    if lengths is not None:
        ln = torch.as_tensor(lengths, dtype=torch.int64, device=dev)
        if ln.numel() != M or int(ln.max()) > T or int(ln.min()) < 1:
            raise ValueError("lengths: one entry per track, 1 <= length <= n_steps")
        row0 = torch.zeros(M + 1, dtype=torch.int64, device=dev)
        row0[1:] = torch.cumsum(ln, 0)
        n = int(row0[-1])
    else:
        n = M * T
    sd = SsdeSimDesc()
    sd.abi_version, sd.model, sd.n_dim, sd.n_steps = ABI_VERSION, MODEL_CODES[model], d, T
    sd.track0, sd.n_tracks, sd.n_rows = int(track0), M, n
    sd.row0 = None if row0 is None else row0.data_ptr()
    sd.row_offset = int(track0) * T if row_offset is None else int(row_offset)
    for k, v in enumerate(np.broadcast_to(np.asarray(mu, dtype=np.float64), (d,))):
        sd.mu[k] = float(v)
    for k, v in enumerate(np.broadcast_to(np.asarray(z0, dtype=np.float64), (d,))):
        sd.z0[k] = float(v)
    sd.tau, sd.nu, sd.kappa, sd.sigma, sd.sigma_obs, sd.dt = float(tau), float(nu), float(kappa), float(sigma), float(sigma_obs), float(dt)
    sd.seed, sd.device = int(seed) & 0xFFFFFFFFFFFFFFFF, int(dev.index or 0)
    obs = torch.empty((d, n), dtype=torch.float64, device=dev)
    ID = torch.empty(n, dtype=torch.float64, device=dev) if want_id_times else None
    times = torch.empty(n, dtype=torch.float64, device=dev) if want_id_times else None
    stream = torch.cuda.current_stream(dev).cuda_stream
    st = lib.ssde_simulate(C.byref(sd), None if ID is None else ID.data_ptr(), None if times is None else times.data_ptr(),
                           obs.data_ptr(), stream)
    if st != 0:
        msg = lib.ssde_last_error(None)
        raise EngineError(f"ssde_simulate failed ({st}): {msg.decode() if msg else ''}")
    return ID, times, obs.t()


class Engine:
    """One created engine (= one `MakeADFun` object of the reference): on one GPU, or -- `devices=[...]` -- sharded
    by whole tracks over several GPUs of this process (ssde_desc.n_devices; one RCCL all-reduce per evaluation)."""

    def __init__(self, problem: Problem, devices: Optional[Sequence[int]] = None):
        self.lib = load_library()
        self.problem = problem
        self._h = C.c_void_p()
        d = problem.desc()
        if devices is not None and len(devices) > 1:
            self._devices = np.ascontiguousarray(devices, dtype=np.int32)
            d.n_devices = len(self._devices)
            d.devices = self._devices.ctypes.data_as(_ip)
        st = self.lib.ssde_create(C.byref(d), C.byref(self._h))
        if st != 0:
            msg = self.lib.ssde_last_error(None)
            raise EngineError(f"ssde_create failed ({st}): {msg.decode() if msg else ''}")
        self.n_par_full = problem.n_par_full

    def _check(self, st):
        if st != 0:
            msg = self.lib.ssde_last_error(self._h)
            err = EngineError(f"ssde call failed ({st}): {msg.decode() if msg else ''}")
            err.status = int(st)
            raise err

    def info(self) -> dict:
        inf = SsdeInfo()
        self._check(self.lib.ssde_info(self._h, C.byref(inf)))
        return inf.as_dict()

    def eval(self, par, order: int = 1):
        par = np.ascontiguousarray(par, dtype=np.float64)
        if par.shape != (self.n_par_full,):
            raise ValueError(f"par must have length {self.n_par_full}")
        val = C.c_double()
        grad = np.zeros(self.n_par_full)
        self._check(self.lib.ssde_eval(self._h, par.ctypes.data_as(_dp), self.n_par_full, order,
                                       C.byref(val), grad.ctypes.data_as(_dp)))
        return (val.value, grad) if order >= 1 else val.value

    def comm_init(self, n_ranks: int, rank: int, unique_id: bytes):
        """Join this rank's engine with the other ranks' (one process per GPU): every later eval / eval_device returns
        the RCCL all-reduced batch result.  Collective: every rank calls it with the same id."""
        assert len(unique_id) == COMM_ID_BYTES
        buf = C.create_string_buffer(unique_id, COMM_ID_BYTES)
        self._check(self.lib.ssde_comm_init_rank(self._h, n_ranks, rank, buf))

    def laplace_eval(self, par, order: int = 1, want_hessian: bool = False, hess_step: float = 0.0, fd_step: float = 0.0,
                     newton_tol: float = 0.0, max_newton: int = 0):
        """Marginal nllk (coeff_re integrated out by the Laplace approximation, ssde_laplace_eval) at the outer entries
        of `par`, warm-started at its coeff_re entries.  Returns (value, grad, par_with_u_hat[, H_uu])."""
        p = np.array(par, dtype=np.float64)
        if p.shape != (self.n_par_full,):
            raise ValueError(f"par must have length {self.n_par_full}")
        val = C.c_double()
        grad = np.zeros(self.n_par_full)
        pb = self.problem
        nu = int(np.sum(pb.par_fixed[pb.off_re:pb.off_re + pb.n_re] == 0))
        H = np.zeros((nu, nu), order="F") if want_hessian else None
        opts = SsdeLaplaceOpts(hess_step, fd_step, newton_tol, max_newton, 0)
        self._check(self.lib.ssde_laplace_eval(self._h, p.ctypes.data_as(_dp), self.n_par_full, order, C.byref(val),
                                               grad.ctypes.data_as(_dp), None if H is None else H.ctypes.data_as(_dp),
                                               C.byref(opts)))
        out = (val.value, grad, p)
        return out + (H,) if want_hessian else out

    def bound_eval(self, order: int = 1):
        """ssde_eval with preallocated buffers and pre-resolved ctypes objects: call(par_array) -> (value, grad_view).
        For timing loops: `Engine.eval` spends ~5 us per call on argument conversion and a fresh gradient array."""
        val = C.c_double()
        grad = np.zeros(self.n_par_full)
        gp, vp, f, h, n = grad.ctypes.data_as(_dp), C.byref(val), self.lib.ssde_eval, self._h, self.n_par_full

        def call(par):
            st = f(h, par.ctypes.data_as(_dp), n, order, vp, gp)
            if st != 0:
                self._check(st)
            return val.value, grad

        return call

    def last_kernel_ms(self) -> float:
        return float(self.lib.ssde_last_kernel_ms(self._h))

    def kernel_ms_history(self, n: int) -> np.ndarray:
        """Dominant-kernel durations of the last n (<= 64) evaluations, most recent first; 0 where there is no stamp."""
        out = np.zeros(int(n))
        self._check(self.lib.ssde_kernel_ms_history(self._h, out.ctypes.data_as(_dp), int(n)))
        return out

    def last_phase_ms(self) -> dict:
        """Where the last stamped synchronous eval spent its time (ssde_last_phase_ms), in ms, by PHASE_NAMES."""
        out = np.zeros(8)
        self._check(self.lib.ssde_last_phase_ms(self._h, out.ctypes.data_as(_dp)))
        return dict(zip(PHASE_NAMES[:7], out[:7].tolist()))

    def comm_allreduce(self, buf_ptr: int, count: int, stream: int = 0):
        """Sum `count` doubles at the HBM address buf_ptr over the ranks of this handle's communicator (enqueue only)."""
        self._check(self.lib.ssde_comm_allreduce(self._h, C.c_void_p(buf_ptr), int(count), C.c_void_p(stream)))

    def hess(self, par, idx):
        """ssde_hess: exact second derivatives of the joint penalised nllk over the full-parameter indices `idx` -- what
        info()["exact_hess_scope"] says: 3 every free entry (state-space models on the lane = direction path -- row-varying
        coefficients, per-row H_array, ESEAL_SSM -- or created with FLAG_EXACT_HESS; track shards and communicator ranks summed),
        2 the coefficients of the direct families BM / OU / BM_t / CIR, decaying columns and log_decay included (+ log_lambda),
        1 the drift coefficients of a smooth-drift state-space
        batch (+ log_lambda), 0 nothing: EngineError with status 2, difference the gradient.  Returns an (len(idx), len(idx)) array."""
        par = np.ascontiguousarray(par, dtype=np.float64)
        ix = np.ascontiguousarray(idx, dtype=np.int32)
        H = np.zeros((len(ix), len(ix)), order="F")
        self._check(self.lib.ssde_hess(self._h, par.ctypes.data_as(_dp), self.n_par_full, ix.ctypes.data_as(_ip), len(ix),
                                       H.ctypes.data_as(_dp)))
        return H

    def set_option(self, option: int, value: int):
        self._check(self.lib.ssde_set_option(self._h, option, value))

    def forget(self):
        """Drop the memoised last result: the next eval runs on the device even at the same par."""
        self._check(self.lib.ssde_forget(self._h))

    def widen_windows(self, factor: int = 4):
        self._check(self.lib.ssde_widen_windows(self._h, factor))

    def relax_windows(self):
        self._check(self.lib.ssde_relax_windows(self._h))

    def eval_device(self, par, out_ptr: int, order: int = 1, stream: int = 0):
        """Asynchronous evaluation of the data term into an HBM buffer of 2+n_par_full doubles:
        [nllk, grad..., window_check]; the caller rejects the result if window_check > WINDOW_TOL."""
        par = np.ascontiguousarray(par, dtype=np.float64)
        self._check(self.lib.ssde_eval_device(self._h, par.ctypes.data_as(_dp), self.n_par_full, order,
                                              C.c_void_p(out_ptr), C.c_void_p(stream)))

    def penalty(self, par, want_grad: bool = True):
        par = np.ascontiguousarray(par, dtype=np.float64)
        val = C.c_double()
        grad = np.zeros(self.n_par_full)
        self._check(self.lib.ssde_penalty(self._h, par.ctypes.data_as(_dp), self.n_par_full, C.byref(val),
                                          grad.ctypes.data_as(_dp) if want_grad else None))
        return val.value, grad

    def report(self, par):
        par = np.ascontiguousarray(par, dtype=np.float64)
        out = np.zeros((self.problem.n, self.problem.sdim), order="F")
        self._check(self.lib.ssde_report(self._h, par.ctypes.data_as(_dp), self.n_par_full,
                                         out.ctypes.data_as(_dp)))
        return out

    def close(self):
        if self._h:
            self.lib.ssde_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
