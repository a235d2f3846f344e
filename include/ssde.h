/*
 * ssde.h -- C ABI of the MI355X-native smoothSDE negative-log-likelihood engine.
 *
 * This is the drop-in boundary for ONE path of the reference package
 * (TheoMichelot/smoothSDE): the evaluation of the penalised negative
 * log-likelihood and its parameter gradient, which the reference reaches as
 *
 *     R: tmb_obj$fn(x) / tmb_obj$gr(x)                 (R/sde.R:694-697)
 *       -> .Call("EvalADFunObject", ptr, theta, ctrl)  (src/init.c:8)
 *         -> objective_function<Type>::operator()      (src/smoothSDE.cpp:9-28)
 *           -> nllk_ctcrw / nllk_ou_ssm / nllk_bm_ssm / nllk_sde
 *                                                      (src/nllk/ headers)
 *
 * and which TMB::MakeADFun sets up from `tmb_dat` / `tmb_par` / `map`
 * (R/sde.R:528-536, 542-598, 621-632, 656-669;
 *  .Call("MakeADFunObject", data, parameters, reportenv, control), src/init.c:6).
 *
 * Entry points and what they replace:
 *
 *   ssde_create        <- MakeADFunObject       (src/init.c:6;  R/sde.R:656-658)
 *   ssde_eval          <- EvalADFunObject       (src/init.c:8;  order 0 = fn, 1 = gr)
 *   ssde_eval_device   <- (new) same evaluation, asynchronous, result left in HBM so the
 *                         caller can all-reduce it over RCCL before reading it
 *   ssde_report        <- REPORT(aest_all)      (src/nllk/nllk_ctcrw.hpp:249,
 *                                                nllk_ou_ssm.hpp:215, nllk_bm_ssm.hpp:177)
 *   ssde_penalty       <- smoothing penalty     (nllk_ctcrw.hpp:254-280, nllk_sde.hpp:89-124)
 *   ssde_info          <- InfoADFunObject       (src/init.c:7)
 *   ssde_forget        <- (new) drops the memo of ssde_eval
 *   ssde_last_kernel_ms <- (new) measurement hook: duration of the last dominant kernel launch
 *   ssde_kernel_ms_history <- (new) the same for the last n evaluations, read after a timed region
 *   ssde_comm_unique_id / ssde_comm_init_rank
 *                      <- (new) one process per GPU: joins the handles of all ranks into one RCCL communicator, after
 *                         which ssde_eval / ssde_eval_device return the all-reduced batch result on every rank
 *   ssde_laplace_eval  <- what `random = "coeff_re"` makes TMB's fn/gr do (R/sde.R:510-525, 656-658): the Laplace
 *                         approximation of the marginal likelihood over the random-effect coefficients
 *   ssde_hess          <- MakeADHessObject2 / tmb_obj_joint$he(x) (src/init.c:13; R/sde.R:1363): second derivatives of the
 *                         joint penalised nllk -- exact for the Gaussian direct families (BM, OU)
 *   ssde_simulate      <- SDE$simulate          (R/sde.R:1393-1500; CTCRW_cov, R/utility.R:188-196): exact-transition
 *                         simulation of a batch of tracks, written straight into HBM in the reference's long format
 *   ssde_set_option    <- (new) per-handle switches of the measurement hooks
 *   ssde_last_phase_ms <- (new) measurement hook: where the last stamped synchronous evaluation spent its time (kernel,
 *                         finalising launch, all-reduce wait, host side) -- what a multi-GPU run's per-rank account is made of
 *   ssde_comm_allreduce <- (new) sums a caller's HBM buffer over the handle's communicator: hosts that drive several
 *                         handles at once issue ONE collective for all of them (SSDE_OPT_COMM_DEFER)
 *   ssde_destroy       <- external-pointer finalizer of the ADFun object
 *   ssde_last_error    <- Rf_error text         (src/smoothSDE.cpp:25)
 *
 * Plain C types only: no R, torch or HIP types appear in any signature.  A HIP
 * stream crosses the boundary as `void*` (NULL = the null stream).
 *
 * Layout conventions are those of the R objects the reference passes
 * (column-major matrices; `obs` is n x d as.matrix(self$obs()), R/sde.R:531).
 *
 * PARAMETER VECTOR.  `par` is the FULL parameter vector in TMB template order,
 * fixed ("mapped") entries included:
 *   Kalman families (BM_SSM, OU_SSM, CTCRW; nllk_ctcrw.hpp:135-140):
 *       [ log_sigma_obs | coeff_fe (sum ncol_fe) | log_lambda (n_smooth) | coeff_re (sum ncol_re) ]
 *   ESEAL_SSM (nllk_e_seal_ssm.hpp:114-124):
 *       [ log_tau | a1 | log_a2 | coeff_fe | log_lambda | coeff_re ]
 *   direct families (BM, BM_t, OU; nllk_sde.hpp:42-45):
 *       [ coeff_fe | log_lambda | log_decay (only with decaying columns, see n_decay) | coeff_re ]
 *   coeff_fe / coeff_re are ordered parameter-by-parameter exactly like the columns
 *   of the reference's block-diagonal X_fe / X_re (R/sde.R:443-447).
 * `par_fixed[k] != 0` marks an entry that TMB's `map` would hold fixed
 * (R/sde.R:514-515, 565, 595, 627-631): no derivative is computed for it and
 * its gradient slot is returned as 0.  The R shim scatters theta into the full
 * vector and gathers the free gradient entries.
 */
#ifndef SSDE_H
#define SSDE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SSDE_ABI_VERSION 10

/* model codes: DATA_STRING(type) of src/smoothSDE.cpp:12-27 */
enum {
    SSDE_MODEL_BM     = 0, /* "BM"      -> nllk_sde + tr_dens BM branch  (tr_dens.hpp:32-37) */
    SSDE_MODEL_OU     = 1, /* "OU"      -> nllk_sde + tr_dens OU branch  (tr_dens.hpp:45-52) */
    SSDE_MODEL_BM_SSM = 2, /* "BM_SSM"  -> nllk_bm_ssm                                    */
    SSDE_MODEL_OU_SSM = 3, /* "OU_SSM"  -> nllk_ou_ssm                                    */
    SSDE_MODEL_CTCRW  = 4, /* "CTCRW"   -> nllk_ctcrw                                     */
    SSDE_MODEL_BM_T   = 5, /* "BM_t"    -> nllk_sde + tr_dens BM_t branch (tr_dens.hpp:38-44): one response,
                              par = (mu, log sigma), degrees of freedom in other_data[0] (R/sde.R:539-541) */
    SSDE_MODEL_ESEAL_SSM = 6, /* "ESEAL_SSM" -> nllk_eseal_ssm (nllk_e_seal_ssm.hpp:83-250): one response, par = (mu, log sigma),
                              state (1, lipid mass), Z_i = (a1, a2 / R_i), H_i = tau^2 / h_i; needs a0, eseal_h, eseal_R */
    SSDE_MODEL_CIR    = 7  /* "CIR"     -> nllk_sde + tr_dens CIR branch (tr_dens.hpp:53-67): par = (log mu_a, log beta, log sigma),
                              positive observations; log I_q(x) is formed directly (no overflow at large x) */
};

/* Which kernel family ran the last evaluation's rows (ssde_info_t.kernel_id).  `path` says which engine path a handle is on;
 * this says which of that path's kernels the dispatch picked for the batch at hand (kernel map). */
enum {
    SSDE_KERNEL_NONE          = 0,  /* nothing evaluated yet */
    SSDE_KERNEL_DIRECT        = 1,  /* direct_kernel: BM / BM_t / OU / CIR transition densities, any slot layout (k_direct.hip) */
    SSDE_KERNEL_DIRECT_FAST   = 2,  /* direct_fast_kernel: at most two parameters with streamed / table-evaluated columns */
    SSDE_KERNEL_ISO_SHARED    = 3,  /* iso_shared_kernel: regular grid, complete tracks -- shared covariance, stationary lanes */
    SSDE_KERNEL_ISO_MASK      = 4,  /* iso_mask_kernel: the lane's own covariance, irregular grid (k_iso.hip) */
    SSDE_KERNEL_ISO_MASK_UNI  = 5,  /* ... regular grid with missing rows (transition hoisted) */
    SSDE_KERNEL_ISO_QUIET     = 6,  /* iso_quiet_kernel: ... with quiet rows (stationary lanes between missing rows) */
    SSDE_KERNEL_ISO_MIXED     = 7,  /* iso_shared_kernel and a general kernel side by side (complete and incomplete wavefronts) */
    SSDE_KERNEL_ISO_SPLIT     = 8,  /* iso_kernel: gradient directions split over several waves per (group, window) */
    SSDE_KERNEL_ISO_DRIFT     = 9,  /* iso_drift_kernel: shared covariance + streamed row-varying drift (k_iso_drift.hip) */
    SSDE_KERNEL_ISO_DRIFT_GEN = 10, /* iso_drift_general_kernel: the same model with the lane's own covariance */
    SSDE_KERNEL_ISO_COLVAR    = 11, /* iso_colvar_kernel: row-varying tau / nu, eight-wave pipeline (k_iso_colvar.hip) */
    SSDE_KERNEL_ISO_FEW       = 12, /* iso_few_kernel: ... few design columns, one wave per (group, window) */
    SSDE_KERNEL_ISO_FULL      = 13, /* iso_full_kernel: per-row H_array, constant tau / nu (4 x 4 covariance lanes) */
    SSDE_KERNEL_DENSE         = 14, /* dense_kernel: general Kalman step, lane = track (k_dense.hip) */
    SSDE_KERNEL_TV            = 15, /* tv_filter_kernel: row-varying coefficients, lane = gradient direction (k_tv.hip) */
    SSDE_KERNEL_TV_DENSE      = 16, /* ... with full-covariance lanes (k_tv_dense.hip; ESEAL_SSM too) */
    SSDE_KERNEL_ISO_ADJ       = 17  /* iso_adj_kernel: row-varying tau / nu / drift, gradient by a reverse sweep (k_iso_adj.hip) */
};

/* status codes (0 = ok).  HIP runtime failures are reported as SSDE_ERR_HIP with
 * the hipError_t text in ssde_last_error(). */
enum {
    SSDE_OK            = 0,
    SSDE_ERR_ARG       = 1, /* malformed descriptor / argument                         */
    SSDE_ERR_MODEL     = 2, /* "Unknown SDE type" (src/smoothSDE.cpp:25) / unsupported */
    SSDE_ERR_HIP       = 3, /* HIP runtime error                                       */
    SSDE_ERR_NODEVICE  = 4, /* no gfx950 device visible: there is NO CPU fallback      */
    SSDE_ERR_ALLOC     = 5
};

/* how a missing observation is recognised (Q5 of SURVEY.md Appendix A) */
enum {
    SSDE_NA_R_ONLY = 0, /* R_IsNA(): NaN whose low word is 1954 (nllk_ctcrw.hpp:214, tr_dens.hpp:31) */
    SSDE_NA_ANY_NAN = 1 /* any NaN counts as missing (non-R hosts)                                    */
};

/* ssde_desc.flags */
#define SSDE_FLAG_DEVICE_DATA   0x1u /* id/times/obs/x_fe/x_re/h_array pointers are HBM pointers on `device` */
#define SSDE_FLAG_FORCE_DENSE   0x2u /* disable the isotropic register path (testing the dense path)         */
#define SSDE_FLAG_NO_UNIFORM_DT 0x4u /* disable hoisting of the transition matrices on a regular time grid    */
#define SSDE_FLAG_EXACT_HESS    0x8u /* the caller will ask for exact second derivatives (ssde_hess, ssde_laplace_eval): a state-space
                                        batch (H = sigma_obs^2 I) that ssde_create puts on the lane = track register kernels -- constant
                                        coefficients, a smooth drift, row-varying tau / nu: first-order kernels -- gets a second,
                                        lane = direction copy of its rows for the second-order pass (k_tv_hess.hip): ~0.64 KB of HBM
                                        per row on top of the tiles, kept only when that is at most a third of the device memory still
                                        free (ssde_info.exact_hess_scope tells).  Without the flag (or the copy) such a handle answers
                                        SSDE_ERR_MODEL and the Laplace layer differences the gradient.  Every track shard of a
                                        multi-device handle keeps its own copy; the ranks of a communicator sum theirs. */

/* A random-effect design block given as a FUNCTION of one covariate instead of as n x K numbers: a piecewise-cubic
 * table (regression-spline bases -- mgcv "cr", "cs", "bs", "ps" -- are exactly that; thin-plate bases are not and
 * keep being streamed):
 *     X[i, k] = sum_{m=0..3} coef[(iv * n_cols + k) * 4 + m] * t^m,   t = x[i] - knots[iv],
 *     iv = the interval knots[iv] <= x[i] < knots[iv+1] (x outside the knot range uses the first / last interval).
 * The direct families then read 8 B/row (x) where the streamed block costs 8 K B/row. */
typedef struct ssde_ppbasis {
    const double *x;              /* [n] covariate (host, or HBM with SSDE_FLAG_DEVICE_DATA) */
    int32_t  n_knots;             /* breakpoints; n_knots - 1 intervals */
    int32_t  n_cols;              /* K == ncol_re[j] */
    const double *knots;          /* [n_knots] increasing (host) */
    const double *coef;           /* [(n_knots - 1) * n_cols * 4] (host) */
} ssde_ppbasis;

typedef struct ssde_desc {
    int32_t  abi_version;     /* SSDE_ABI_VERSION */
    int32_t  model;           /* SSDE_MODEL_* */
    int32_t  n_dim;           /* d = ncol(obs).  Every kernel is sized for d <= 2; a wider response is evaluated as the pairs of
                                 columns (0,1), (2,3), ... behind the one handle (the likelihood is a sum over dimensions:
                                 nllk_ctcrw.hpp:49-89 are block-diagonal in the dimension, nllk_sde.hpp:77-84 loops over it),
                                 which needs a P0 and an H_array without entries between different pairs.  BOUND: a measurement
                                 covariance or P0 that couples the pairs is accepted for d <= 8 (the whole response as one filter, F by
                                 LU -- k_dense.hip, k_dense_wide.hip -- on one device, host or device-resident arrays, or on every
                                 whole-track shard of several); d >= 9 with coupling is SSDE_ERR_MODEL */
    int32_t  n_par;           /* q = SDE parameters per row: d+1 (BM, BM_t, BM_SSM), d+2 (OU, OU_SSM, CTCRW) */
    int64_t  n;               /* rows of the long-format data (all tracks concatenated) */
    const double *id;         /* [n] track codes (TMB passes the factor as doubles); only
                                 id[i] != id[i-1] is ever used (nllk_ctcrw.hpp:196)     */
    const double *times;      /* [n]   DATA_VECTOR(times).  Any increasing stamps; recognised at create (the numbers are the
                                 reference's either way): a regular grid (transition hoisted), and a regular
                                 schedule whose failed fixes are absent from the data -- intervals that are small whole
                                 multiples of one step -- which is laid out on its lattice and run as a regular grid with
                                 missing rows */
    const double *obs;        /* [n*d] DATA_MATRIX(obs), column-major */

    /* design blocks, one per SDE parameter j = 0..q-1: the X_list_fe[[j]] /
     * X_list_re[[j]] of R/sde.R:412-417 (the diagonal blocks of X_fe / X_re).
     * x_fe[j] == NULL with ncol_fe[j] == 1 means an intercept-only column of ones. */
    const int32_t *ncol_fe;       /* [q] */
    const double *const *x_fe;    /* [q] each n x ncol_fe[j], column-major, or NULL */
    const int32_t *ncol_re;       /* [q] number of random-effect columns of parameter j (may be 0) */
    const double *const *x_re;    /* [q] each n x ncol_re[j], column-major, or NULL when 0 columns  */

    /* smoothing penalty: DATA_IVECTOR(ncol_re) + DATA_SPARSE_MATRIX(S) of the reference */
    int32_t  n_smooth;            /* number of penalty blocks; 0 = no random effects (R/sde.R:511-518) */
    const int32_t *smooth_ncol;   /* [n_smooth] columns of each block (reference `ncol_re`)           */
    const double *s_blocks;       /* the diagonal blocks of S, each smooth_ncol[s]^2 column-major,
                                     concatenated */
    int32_t  include_penalty;     /* DATA_INTEGER(include_penalty): read by nllk_sde only (Q7) */

    /* Kalman families only */
    int64_t  n_seg;               /* rows of a0 = number of ID segments (R/sde.R:547,574) */
    const double *a0;             /* [n_seg x sdim] column-major, or NULL = built from obs as R/sde.R:549,576-580 */
    const double *p0;             /* [sdim x sdim] column-major, or NULL = default of R/sde.R:554,584 */
    const double *h_array;        /* [d x d x n] DATA_ARRAY(H_array), or NULL = sigma_obs^2 I (R/sde.R:563-568,593-598) */

    const uint8_t *par_fixed;     /* [n_par_full] or NULL (= nothing fixed) */
    int32_t  na_mode;             /* SSDE_NA_* */
    int32_t  device;              /* HIP device ordinal, -1 = current device */
    uint32_t flags;               /* SSDE_FLAG_* */
    uint32_t reserved;
    const double *other_data;     /* DATA_VECTOR(other_data) (nllk_sde.hpp:29): [0] = degrees of freedom for BM_t; NULL otherwise */
    int32_t  n_other_data;
    /* decaying random-effect columns, direct families only (nllk_sde.hpp:30-32, 47-58; R/sde.R:163-177, 635-648):
     * column col_decay[c] of X_re (0-based, counted over the concatenated blocks of all SDE parameters, i.e. the
     * index inside coeff_re) is multiplied row by row by exp(-exp(log_decay[ind_decay[c]]) * t_decay).
     * The parameter vector then carries log_decay: [ coeff_fe | log_lambda | log_decay (n_decay) | coeff_re ]. */
    int32_t  n_decay;             /* number of decay rates, 0 = feature unused (the reference's dummy log_decay is dropped) */
    const double *t_decay;        /* [q * n], entry j*n + i belongs to row i of SDE parameter j (the rows of X_re) */
    int32_t  n_decay_cols;
    int32_t  reserved3;
    const int32_t *col_decay;     /* [n_decay_cols] */
    const int32_t *ind_decay;     /* [n_decay_cols] 0-based index into log_decay */
    /* ESEAL_SSM only: DATA_VECTOR(h) (daily drift dives) and DATA_VECTOR(R) (non-lipid tissue mass),
     * nllk_e_seal_ssm.hpp:100-101, R/sde.R:611-614 */
    const double *eseal_h;        /* [n] */
    const double *eseal_R;        /* [n] */
    /* [q] or NULL.  basis_re[j] != NULL: the random-effect block of SDE parameter j is evaluated from this table
     * (x_re[j] is then ignored by the engine and may be NULL). */
    const ssde_ppbasis *const *basis_re;
    /* Single-process multi-GPU (the reference's host is ONE R process, R/sde.R:656-669): n_devices > 1 shards whole
     * tracks (contiguous ID segments, balanced by rows) over devices[0..n_devices), one engine per device; every
     * ssde_eval then runs all shards concurrently and sums their 2 + p doubles with one ncclAllReduce per device
     * inside ncclGroupStart/End (communicators from ncclCommInitAll).  Host arrays only (no SSDE_FLAG_DEVICE_DATA).
     * n_devices <= 1 or devices == NULL: the single device `device`.  A device listed more than once is accepted for
     * rehearsal on a one-GPU machine: its shards are then summed by a small kernel instead of RCCL (which refuses
     * two ranks on one device). */
    int32_t  n_devices;
    int32_t  reserved4;
    const int32_t *devices;       /* [n_devices] HIP device ordinals */
} ssde_desc;

typedef struct ssde_handle ssde_handle;

/* facts about a created engine (InfoADFunObject counterpart) */
typedef struct ssde_info_t {
    int32_t n_par_full;     /* length of `par` / `grad` */
    int32_t n_free;         /* entries with par_fixed == 0 */
    int32_t sdim;           /* Kalman state dimension (0 for direct families) */
    int32_t path;           /* 0 direct, 1 isotropic register Kalman (constant coefficients), 2 dense Kalman,
                               3 isotropic Kalman with row-varying coefficients */
    int32_t const_coeff;    /* 1 if every SDE parameter is intercept-only */
    int32_t uniform_dt;     /* 1 if one dt is shared by every scored interval (transition matrices hoisted) */
    int64_t n_tracks;       /* ID segments */
    int64_t n_rows;         /* n */
    int64_t n_steps;        /* rows that are not the first row of a segment */
    int64_t hbm_bytes;      /* resident bytes of the tiled streams */
    double  algo_bytes_per_row; /* 8*(d + 1 + d*d*[H_array] + K_row) */
    int32_t n_kernel_blocks;/* workgroups of the main kernel (last evaluation) */
    int32_t lanes_per_track;/* direction parts x time windows (register path), direction blocks (dense),
                               direction lanes x time windows (row-varying path) */
    int32_t window;         /* warm-up rows of a time window in the last evaluation (0 = sequential) */
    int32_t window_retries; /* evaluations repeated because the window hand-over check failed */
    double  window_check;   /* largest relative hand-over disagreement of the last ssde_eval (<= 1e-11 when the windows agree; a value up
                               to 1e-8 is accepted only as a ROUNDING FLOOR: one that stayed flat while the warm-up was quadrupled twice --
                               very precise fixes -- and is reported here as it is) */
    double  main_kernel_ms; /* HIP-event duration of the dominant kernel launch of the last evaluation */
    int64_t main_kernel_rows;/* rows scored by that launch (the rest belong to the small concurrent launches) */
    double  required_bytes_per_row; /* bytes per row the engine's resident layout really has to read: algo_bytes_per_row
                                       minus the 8 B/row of `times` when the grid is globally regular and the dt channel
                                       is not even stored (24 -> 16 for 2-D constant-coefficient models) */
    int64_t n_evals;        /* evaluations that ran on the device (window retries included) */
    int64_t n_memo_hits;    /* ssde_eval calls answered from the memoised last result (same par, bitwise) */
    int32_t n_devices;      /* shards of a single-process multi-GPU handle (1 otherwise) */
    int32_t comm_ranks;     /* ranks of the RCCL communicator joined with ssde_comm_init_rank (1 = none) */
    double  window_check_max; /* largest ACCEPTED hand-over disagreement over every ssde_eval since ssde_create */
    /* how the engine laid the batch out (Kalman register path; 0 elsewhere) -- ABI 7 */
    int64_t n_rows_tiled;   /* rows of the resident layout: n_rows, or more when a schedule with absent fixes was laid out
                               on its lattice */
    int32_t n_groups;       /* 64-track wavefront groups */
    int32_t n_clean_groups; /* ... of which every track has every row: the shared-covariance kernel's share on a regular grid */
    int32_t quiet_window;   /* > 0: the general kernel of the last evaluation ran rows that lie this many rows past the last missing
                               observation of their wavefront with the stationary gains; 0: no such rows */
    int32_t kernel_id;      /* SSDE_KERNEL_*: the kernel family that ran the rows of the last evaluation -- ABI 10 */
    double  quiet_share;    /* share of the blocks of wavefronts with missing rows that qualify, at a nominal 128-row memory
                               (found at ssde_create; 0 when the layout has no such wavefront or the grid is irregular) */
    int32_t comm_ranks_reported; /* what ncclCommCount says about the joined communicator (0 = none joined); comm_ranks above
                               echoes the caller's argument -- ABI 10 */
    int32_t exact_hess_scope; /* what ssde_hess is exact over on this handle: 0 nothing (difference the gradient), 1 the drift coefficients
                                 (+ log_lambda), 2 every coefficient of a direct family (+ log_lambda), 3 every free entry (was reserved: 0) */
} ssde_info_t;

/* Create an engine: validates the descriptor, finds the ID segments, uploads the
 * data ONCE and re-tiles it in HBM (tracks -> wavefront lanes, time-major) so that
 * every later evaluation streams it with coalesced loads.  The caller keeps
 * ownership of every host array; nothing is aliased after return. */
int ssde_create(const ssde_desc *desc, ssde_handle **out);

/* fn(par) / gr(par).  order 0: *value only.  order 1: *value and grad[n_par_full].
 * value = nllk including the smoothing penalty exactly as the reference adds it.
 * A non-finite nllk is RETURNED (status SSDE_OK), not raised.
 * The last result is memoised on the bit pattern of `par`: optim's fn(x); gr(x) pair (R/sde.R:694-696) costs ONE
 * device evaluation on the paths where the gradient rides along with the value at no extra memory traffic (shared-
 * covariance and direct kernels: an order-0 call evaluates order 1 there), and a repeated call costs none. */
int ssde_eval(ssde_handle *h, const double *par, int32_t n_par_full, int32_t order,
              double *value, double *grad);

/* Same evaluation WITHOUT the penalty and without synchronising: enqueues on
 * `stream` and leaves [nllk_data, grad..., window_check] (2 + n_par_full doubles) in
 * the HBM buffer `out_dev`.  Used by multi-GPU hosts: shard tracks, all-reduce
 * out_dev, then add ssde_penalty once.  window_check is the largest relative
 * disagreement between overlapping time windows of the register path (0 when the
 * evaluation ran as one sequential window): the caller must treat the result as
 * invalid when it exceeds 1e-11 (ssde_eval re-evaluates with a longer overlap by
 * itself; asynchronous callers do the same after ssde_widen_windows). */
int ssde_eval_device(ssde_handle *h, const double *par, int32_t n_par_full, int32_t order,
                     double *out_dev, void *stream);

/* Smoothing penalty and its gradient (host arithmetic, parameter-only). grad may be NULL;
 * otherwise it is ADDED into grad[n_par_full]. */
int ssde_penalty(ssde_handle *h, const double *par, int32_t n_par_full, double *value, double *grad);

/* One-step-ahead predicted states for every row: aest_all [n x sdim] column-major. */
int ssde_report(ssde_handle *h, const double *par, int32_t n_par_full, double *aest_all);

/* Multiply the warm-up overlap of the time windows by `factor` for all later evaluations
 * (factor <= 0: force one sequential window). */
int ssde_widen_windows(ssde_handle *h, int32_t factor);

/* Halve that multiplier again (never below 1).  ssde_eval does this by itself after 32 evaluations accepted at the
 * first try; asynchronous callers that re-evaluate after ssde_widen_windows use this for the same policy.  Every
 * evaluation is checked, so a narrower overlap that does not hold shows in window_check. */
int ssde_relax_windows(ssde_handle *h);

int ssde_info(const ssde_handle *h, ssde_info_t *info);

/* HIP-event duration (ms) of the dominant kernel launch of the last evaluation (== ssde_info().main_kernel_ms, without
 * filling the rest of the structure: cheap enough to be read after every timed step of a benchmark); 0 if unknown. */
double ssde_last_kernel_ms(const ssde_handle *h);

/* The same for the last n evaluations at once: ms[0] = the last one, ms[1] the one before, ... (0 where no stamp exists:
 * more than 64 evaluations ago, or an evaluation replayed from a hipGraph).  Every evaluation stamps its dominant kernel
 * with an event pair of its own, so a caller can time K <= 64 evaluations back to back and read their kernel durations
 * AFTER the timed region (bench.py) instead of querying events between them. */
int ssde_kernel_ms_history(const ssde_handle *h, double *ms, int32_t n);

/* Where the last SYNCHRONOUS ssde_eval spent its time, in milliseconds (single-device handle; valid after an evaluation made with
 * SSDE_OPT_KERNEL_STAMPS on, zeros otherwise).  GPU-side phases come from HIP events on the evaluation's stream:
 *   ms[0] host_total   wall time of the call
 *   ms[1] host_enqueue ... of which: planning, gain table, launches (the host side until everything was enqueued)
 *   ms[2] gpu_pre      first operation of the evaluation on the stream -> the dominant kernel's begin (uploads, launch gap)
 *   ms[3] kernel       the dominant kernel (== ssde_last_kernel_ms)
 *   ms[4] finalize     its end -> the end of the finalising launch (hand-over checks + fixed-order sums)
 *   ms[5] allreduce    ... -> the end of the ncclAllReduce behind it (0 without a communicator): wire time AND the wait for
 *                      the slowest rank
 *   ms[6] readback     host_total minus everything above (the blocking copy and what the runtime adds around it)
 *   ms[7] reserved (0)
 * Returns SSDE_ERR_ARG for a multi-device parent.  A measurement hook: bench.py gathers it per rank. */
int ssde_last_phase_ms(ssde_handle *h, double ms[8]);

/* Drop the memoised last result: the next ssde_eval runs on the device whatever its argument (determinism checks). */
int ssde_forget(ssde_handle *h);

/* fn / gr of the MARGINAL negative log-likelihood: coeff_re integrated out by the Laplace approximation, which is
 * what tmb_obj$fn / tmb_obj$gr are when SDE$setup passes random = "coeff_re" (R/sde.R:510-525, 656-658):
 *     f(theta) = g(theta, u^) + 1/2 log det H_uu(theta, u^) - n_u/2 log(2 pi),   u^ = argmin_u g(theta, u),
 * g = ssde_eval's joint penalised nllk, u = the coeff_re entries that are not fixed, theta = every other free entry
 * (log_sigma_obs, coeff_fe, log_lambda, log_decay).
 *   par     [n_par_full] IN/OUT: outer parameters and the starting values of coeff_re (warm start); on return the
 *           coeff_re entries hold u^ (TMB's par.random / env$last.par)
 *   order   0: *value;  1: also grad[n_par_full] = df/dtheta (zeros at coeff_re and at fixed entries)
 *   hess_uu NULL or [n_u x n_u] column-major: H_uu at u^ (the random-effect block of sdreport's jointPrecision)
 * Direct families BM / OU / BM_t / CIR: H_uu and H_u,theta are EXACT (ssde_hess), only 1/2 d log det H_uu / dtheta is a central
 * difference -- of exact Hessians, along the implicit-function tangent of u^.  Elsewhere H_uu comes from central
 * differences of the device gradient.  The inner problem is solved by Newton iterations; the gradient is the exact
 * dg/dtheta at u^ plus that log-determinant term (ssde_laplace.hip).  A joint nllk that has no minimum in u gives
 * *value = +Inf (a rejected step; `par` is then left untouched). */
typedef struct ssde_laplace_opts {
    double hess_step;      /* relative step of the Hessian differences   (<= 0: 1e-4) */
    double fd_step;        /* relative step of the log-determinant term  (<= 0: 1e-4) */
    double newton_tol;     /* inner convergence: largest step component  (<= 0: 1e-8) */
    int32_t max_newton;    /* inner iterations                           (<= 0: 30)   */
    int32_t reserved;
} ssde_laplace_opts;
int ssde_laplace_eval(ssde_handle *h, double *par, int32_t n_par_full, int32_t order, double *value, double *grad,
                      double *hess_uu, const ssde_laplace_opts *opts);

/* Second derivatives of the joint penalised nllk (ssde_eval's value) over the parameter entries idx[0 .. n_idx):
 * hess [n_idx x n_idx] column-major.  EXACT -- no differencing -- for the direct families "BM", "OU", "BM_t" and "CIR"
 * (tr_dens.hpp:32-67) with resident design columns (decaying ones and log_decay included): the SDE parameters are linear in coeff_fe /
 * coeff_re, so the data term's Hessian is X' D X with a closed-form per-row D, and the penalty is exp(log_lambda) times a
 * quadratic form (nllk_sde.hpp:91-124); and for the state-space families CTCRW / OU_SSM / BM_SSM on handles whose rows the
 * lane = direction filter holds (row-varying coefficients, ESEAL_SSM, or any handle created with SSDE_FLAG_EXACT_HESS):
 * second-order forward mode through the filter.  ssde_info.exact_hess_scope says which entries a handle covers.  idx may name coeff_fe,
 * coeff_re and log_lambda entries, fixed or free.  Where the scope does not cover idx (exact_hess_scope 0 or 1) the
 * call returns SSDE_ERR_MODEL: difference ssde_eval's gradient there, as ssde_laplace_eval does.  Works on single-device,
 * multi-device and communicator handles (the shards' / ranks' Hessians are summed). */
int ssde_hess(ssde_handle *h, const double *par, int32_t n_par_full, const int32_t *idx, int32_t n_idx, double *hess);

/* One process per GPU (torchrun / mpirun style hosts).  Rank 0 calls ssde_comm_unique_id (128 bytes, an
 * ncclUniqueId), the host ships it to the other ranks by its own means, then EVERY rank calls ssde_comm_init_rank
 * on its own single-device handle (collective call).  From then on ssde_eval / ssde_eval_device sum
 * [nllk_data, grad..., window_check] over the ranks with one ncclAllReduce (RCCL over xGMI) on the evaluation's
 * stream before anything is read back, so every rank returns the batch value; the penalty is added once, after
 * the sum.  Window retries are decided on the reduced check value, i.e. identically on every rank. */
#define SSDE_COMM_ID_BYTES 128
int ssde_comm_unique_id(void *id128);
int ssde_comm_init_rank(ssde_handle *h, int32_t n_ranks, int32_t rank, const void *id128);

/* Per-handle options.
 *   SSDE_OPT_KERNEL_STAMPS (default 1): every evaluation stamps its dominant kernel with a HIP event pair
 *       (ssde_last_kernel_ms / ssde_kernel_ms_history).  0 = plain launches: a few microseconds less per evaluation,
 *       which is what a fitting host wants; measurement passes switch it back on.
 *   SSDE_OPT_COMM_DEFER (default 0): 1 = ssde_eval_device leaves this rank's PARTIAL [nllk, grad..., check] in the caller's
 *       buffer and the caller sums it over the ranks itself (ssde_comm_allreduce) -- a host that evaluates several handles
 *       side by side packs their result vectors and issues one collective for all of them instead of one per handle on
 *       as many streams.  ssde_eval ignores it. */
enum { SSDE_OPT_KERNEL_STAMPS = 1, SSDE_OPT_COMM_DEFER = 2 };
int ssde_set_option(ssde_handle *h, int32_t option, int64_t value);

/* Sum buf_dev[0 .. count) (doubles, HBM) over the ranks of the communicator `h` joined, in place, on `stream` (enqueue only:
 * one ncclAllReduce).  SSDE_ERR_ARG if the handle has joined none. */
int ssde_comm_allreduce(ssde_handle *h, double *buf_dev, int64_t count, void *stream);

/* Synthetic track batches in HBM: the exact-transition simulator of SDE$simulate (R/sde.R:1434-1478: BM increments,
 * OU transition density, CTCRW joint (velocity, position) transition with CTCRW_cov of R/utility.R:188-196), one
 * wavefront lane per track, plus N(0, sigma_obs^2) observation error for the state-space families (BM_SSM, OU_SSM,
 * CTCRW).  Every normal deviate is a pure function of (seed, GLOBAL track index, row, dimension) -- Philox4x32-10
 * counters + Box-Muller --, so tracks [track0, track0 + n_tracks) of a batch are the same numbers whichever process
 * generates them: ranks of a multi-GPU run each generate their own shard of ONE batch.
 *   obs_dev   [n_rows x n_dim] column-major (HBM), the reference's DATA_MATRIX(obs) layout
 *   id_dev    NULL or [n_rows]: the global track index of every row (as doubles, like TMB's factor codes)
 *   times_dev NULL or [n_rows]: (row_offset + i + 1) * dt -- increasing over the whole batch (inst/example.R:17)
 * Rows: track m (local index) owns rows [m * n_steps, (m + 1) * n_steps), or [row0[m], row0[m + 1]) when the HBM
 * array row0 [n_tracks + 1] is given (ragged tracks; n_steps is then the longest track).  Enqueues on `stream` and
 * returns; ssde_last_error(NULL) holds the message of a failure. */
typedef struct ssde_sim_desc {
    int32_t  abi_version;     /* SSDE_ABI_VERSION */
    int32_t  model;           /* SSDE_MODEL_BM, _OU, _BM_SSM, _OU_SSM, _CTCRW (R/sde.R:1434-1478; the others: SSDE_ERR_MODEL) */
    int32_t  n_dim;           /* response columns, 1..8 */
    int32_t  n_steps;         /* rows per track (the longest track when row0 is given) */
    int64_t  track0;          /* global index of the first track generated by this call */
    int64_t  n_tracks;        /* tracks generated by this call */
    const int64_t *row0;      /* NULL, or HBM [n_tracks + 1] first local row of every track */
    int64_t  n_rows;          /* rows of the local long format (column stride of obs_dev) */
    int64_t  row_offset;      /* global index of local row 0 (only `times` uses it) */
    double   mu[8], z0[8];    /* drift / long-term mean and first value per dimension (simulate's z0, R/sde.R:1388) */
    double   tau, nu;         /* CTCRW (beta = 1 / tau, sigma = 2 nu / sqrt(pi tau): R/sde.R:1455-1458); OU: tau */
    double   kappa;           /* OU variance parameter (R/sde.R:1445) */
    double   sigma;           /* BM diffusion (R/sde.R:1437) */
    double   sigma_obs;       /* sd of the observation error (state-space families; 0 = none) */
    double   dt;              /* the regular time step */
    uint64_t seed;
    int32_t  device;          /* HIP device ordinal, -1 = current device */
    int32_t  reserved;
} ssde_sim_desc;
int ssde_simulate(const ssde_sim_desc *desc, double *id_dev, double *times_dev, double *obs_dev, void *stream);

void ssde_destroy(ssde_handle *h);

/* Message of the last failure on this handle (or of the last failed ssde_create when h == NULL). */
const char *ssde_last_error(const ssde_handle *h);

int ssde_abi_version(void);

#ifdef __cplusplus
}
#endif
#endif /* SSDE_H */
