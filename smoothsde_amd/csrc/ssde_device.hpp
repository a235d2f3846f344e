// ssde_device.hpp -- kernel argument blocks and launch prototypes shared by the engine
// (ssde_engine.hip) and the kernel translation units (k_*.hip).  gfx950 only.
//
// HBM layout ("tiles"): tracks are sorted by length, packed 64 to a wavefront ("group"),
// and every group stores its rows time-major:
//
//     tiles[ group_off[g] + (s * C + c) * 64 + lane ]      s = step (row 1.. of the track),
//                                                           c = channel, lane = track in group
//
// so that one wave-wide load of channel c at step s is 512 contiguous bytes, and a block of
// U steps is U*C*512 contiguous bytes.  Channels: 0 = dt (interval AFTER the row,
// nllk_ctcrw.hpp:126-129,206), 1..D = obs columns, then D*D H_array entries (if supplied),
// then the streamed design columns.  A group's length is padded to a multiple of TILE_U
// steps and the buffer ends with one spare block, so prefetching a block ahead never
// leaves the allocation.
#ifndef SSDE_DEVICE_HPP
#define SSDE_DEVICE_HPP

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "ssde_math.hpp"

namespace ssde {

constexpr int WAVE = 64;
constexpr int WG_WAVES = 4;       // waves per workgroup of the register Kalman kernels: the four waves of a
                                 // workgroup land on the four SIMDs of one CU, which is what balances the SIMDs
                                 // (single-wave workgroups were observed to be packed unevenly); the waves are
                                 // independent work items and never synchronise
constexpr int TILE_U = 4;        // steps per prefetch block of the general register kernel
constexpr int WIN_ALIGN = 16;    // time-window starts / lengths / warm-ups are multiples of this many rows
#ifndef SSDE_SHARED_U
#define SSDE_SHARED_U 8
#endif
constexpr int SHARED_U = SSDE_SHARED_U;     // steps per prefetch block of the shared-covariance kernel (divides WIN_ALIGN)
constexpr int TILE_SPARE = 64;   // spare rows after the last group, so prefetching ahead stays in bounds
constexpr int NACC_MAX = 8;      // 1 + 3 + D accumulators of the constant-coefficient kernels
constexpr int GAIN_ROW = 16;     // doubles per row of the shared gain table (128-B rows for scalar loads)
constexpr int NSTATE_MAX = 32;   // state + sensitivity components dumped at a window hand-over
constexpr int MAX_PARTS = 4;     // direction split: at most one part per direction bit
constexpr int MAX_PAR = 320;     // parameters passed by value in the kernel argument block
constexpr int MAX_COLS = 96;     // streamed design columns (dense / direct kernels)
constexpr int MAX_Q = 4;         // SDE parameters per row (d + 2, d <= 2)
constexpr int HESS_T = 8;         // exact Hessians: coefficient pairs are cut into HESS_T x HESS_T tiles
constexpr int DRIFT_KMAX = 24;   // streamed drift columns of the shared-covariance kernel with a row-varying drift (k_iso_drift.hip)

struct TileView {
    const double* tiles;
    const int64_t* group_off;    // [n_groups] offset of the group's first step, in doubles
    const int32_t* group_len;    // [n_groups] steps, padded to a multiple of TILE_U
    const int32_t* lane_nsteps;  // [n_groups*64] scored rows of the lane's track (T_m - 1), 0 = empty lane
    const double* a0;            // [n_groups][sdim][64] initial state per lane
    int n_groups;
    int C;                       // channels per step
    int c_obs;                   // channel of the first obs column: 1, or 0 when the dt channel is left out (globally
                                 // regular time grid: nobody would read it, and holes in the stream cost bandwidth)
    double dt_all;               // the one interval of a globally regular grid (c_obs == 0)
};

// ---- constant-coefficient isotropic Kalman kernels (k_iso.hip) -----------------------------
// ---- a smooth drift whose design blocks are FUNCTIONS of a covariate (ssde_ppbasis, include/ssde.h): the tiles carry the covariate
// (one channel per block: 8 B/row) and the lanes evaluate the block's K columns from the piecewise-cubic table in LDS, instead of
// streaming K columns (8 K B/row).  At most two blocks (mu_1, mu_2); block b's columns are slots k0[b] .. k0[b] + kcols[b] - 1.
constexpr int PPD_LDS = 1024;              // doubles per block in LDS: (nk - 1) * (K * 4 + 2) table entries (padded, ppd_stage) + nk knots
struct PpDrift {
    int nb;                                // 0: the columns are streamed
    int kcols[2], k0[2], nk[2], uniform[2];
    double x0[2], inv_h[2];                // equally spaced knots: interval by multiplication
    const double* tab[2];                  // HBM [(nk - 1) * kcols * 4]
    const double* knots[2];                // HBM [nk]
};
#ifdef __HIPCC__
// tables and knots into LDS, once per workgroup (every thread of the block calls this, before any wave leaves).  An interval's
// K x 4 coefficients are followed by two doubles of padding: the lanes of a wave sit in DIFFERENT intervals and read 16 bytes each --
// with a stride that is an odd number of 16-byte slots, eight consecutive intervals land in eight different bank groups (the unpadded
// stride, 32 K bytes, lands intervals iv and iv + 4 in the same banks).
__device__ __forceinline__ int ppd_stride(int K) { return K * 4 + 2; }
__device__ __forceinline__ void ppd_stage(const PpDrift& P, double* lds /* [2 * PPD_LDS] */) {
    for (int b = 0; b < P.nb; b++) {
        const int K4 = P.kcols[b] * 4, KS = ppd_stride(P.kcols[b]), ni = P.nk[b] - 1;
        for (int k = threadIdx.x; k < ni * K4; k += blockDim.x) lds[b * PPD_LDS + (k / K4) * KS + (k % K4)] = P.tab[b][k];
        for (int k = threadIdx.x; k < P.nk[b]; k += blockDim.x) lds[b * PPD_LDS + ni * KS + k] = P.knots[b][k];
    }
    __syncthreads();
}
// this lane's interval and offset in block b for covariate value x: returns the LDS index of the interval's first entry MINUS the
// block's first slot (so that slot k's four coefficients sit at [ret + 4 k]), t = x - knots[iv]
__device__ __forceinline__ int ppd_locate(const PpDrift& P, const double* lds, int b, double x, double& t) {
    const int nk = P.nk[b], K = P.kcols[b], KS = ppd_stride(K);
    const double* knots = lds + b * PPD_LDS + (nk - 1) * KS;
    int iv;
    if (P.uniform[b]) {
        iv = (int)floor((x - P.x0[b]) * P.inv_h[b]);
        iv = iv < 0 ? 0 : (iv > nk - 2 ? nk - 2 : iv);
        if (iv > 0 && x < knots[iv]) iv--;                   // (the multiplication may land one interval off at a breakpoint)
        else if (iv < nk - 2 && x >= knots[iv + 1]) iv++;
    } else {
        iv = 0;
        for (int k = 1; k < nk - 1; k++) iv += (x >= knots[k]) ? 1 : 0;
    }
    t = x - knots[iv];
    return b * PPD_LDS + iv * KS - P.k0[b] * 4;
}
// the same without LDS, straight from the tables in HBM (ssde_report of such a handle: once per fit)
__device__ __forceinline__ double ppd_value_hbm(const PpDrift& P, int k, const double* xs /* the row's covariates, stride WAVE */) {
    const int b = (P.nb > 1 && k >= P.k0[1]) ? 1 : 0;
    const double x = xs[b * WAVE];
    const int nk = P.nk[b], K = P.kcols[b];
    int iv = 0;
    for (int q = 1; q < nk - 1; q++) iv += (x >= P.knots[b][q]) ? 1 : 0;
    const double t = x - P.knots[b][iv];
    const double* q = P.tab[b] + ((int64_t)iv * K + (k - P.k0[b])) * 4;
    return fma(fma(fma(q[3], t, q[2]), t, q[1]), t, q[0]);
}
// slot k's value: the cubic of its block's interval (uniform k)
__device__ __forceinline__ double ppd_value(const double* lds, int at, int k, double t) {
    const double2* q = (const double2*)(lds + at + 4 * k);
    const double2 lo = q[0], hi = q[1];
    return fma(fma(fma(hi.y, t, hi.x), t, lo.y), t, lo.x);
}
#endif

struct IsoArgs {
    TileView tv;
    double* partials;            // [n_parts * n_chunks][NACC][n_groups]
    double* bnd;                 // [n_parts * n_chunks][n_groups][2][NSTATE_MAX][64] window hand-over states
    double* chk;                 // [n_parts][n_chunks - 1][n_groups] largest relative hand-over disagreement
    // shared-covariance path (regular time grid, no missing rows in the group): the covariance
    // half of the filter is evaluated once per evaluation into `gain` (ssde_engine.hip) and the
    // lanes run the mean half only
    const double* gain;          // [gain_last + 1][GAIN_ROW] or NULL; rows beyond gain_last repeat the last one
    int gain_last;
    double gain_stat[GAIN_ROW];  // the stationary row (== row gain_last)
    // every wave-uniform constant of the stationary update, precomputed on the host so that the
    // kernel receives them as scalar (SGPR) operands instead of recomputing them into VGPRs
    //   CTCRW : 0 iF 1 k1 2 k2 3 c1=1-k1 4 t12 5 e 6 dt12 7 de 8 cb1 9 cb2 | 10+j hd_j | 13+j dk1_j | 16+j dk2_j
    //           | 19+a cx_a | 21+a cv_a | 23+a bmu_a
    //   OU/BM : 0 iF 1 k 2 c=t-k 3 t 4 b 5 dt_ | 10+j hd_j | 13+j dk_j | 19+a cmu_a | 21+a dbmu_a
    //   CTCRW transfer-function lanes (TfCtcrw): 26 -d1 27 -d2 28 d2 | 29+a mu_a dt | 31+j pi0_j | 34+j pi1_j | 37+j pi2_j
    //           | 40+j d d1_j | 43+j d d2_j | 46 dx/dmu 47 dv/dmu
    double statc[48];
    const int32_t* group_flags;  // [n_groups] bit 0: every track of the group is NaN-free
    int group_mode;              // 0: this launch handles every group; 1: only groups WITHOUT bit 0; 2: only groups WITH bit 0
    int n_chunks;                // time windows per track group (1 = plain sequential filter)
    int window;                  // warm-up rows of a window, multiple of WIN_ALIGN
    int t0;                      // > 0: window 0 is the covariance transient [0, t0) (shared-covariance path)
    int t0_delta;                // stationary rows the transient window is worth (window 1 is shortened by it), multiple of WIN_ALIGN
    // Mixed batch (some track groups on the shared-covariance kernel, the others on the general kernel): the two launches
    // have window plans of their own -- the shared kernel wants few long windows (every window costs it a warm-up), the
    // general launch enough of them to fill two waves per SIMD.  The launches get their plan in the fields above; the
    // finalize launch, which checks the hand-overs of both, reads the general launch's plan here (dual != 0).
    int dual;
    int n_chunks_d, window_d, t0_d, t0_delta_d;
    const int32_t* dirty_groups; // [n_dirty_groups] the groups of the general launch (dual != 0: their hand-over checks are enumerated through it)
    int n_dirty_groups;
    int use_group_list;          // this launch's grid enumerates dirty_groups[] instead of every group (the general launch of a mixed batch)
    int n_parts;
    int part_mask[MAX_PARTS];    // DIR_* bits handled by each part
    int any_nan;
    int uniform_dt;              // 1: transition hoisted (ctr / str valid), dt channel not read
    double h;                    // sigma_obs^2
    double mu[2];
    double p0[3];                // CTCRW: p11,p12,p22; OU/BM: p
    double tau, beta, sigma;     // CTCRW (nllk_ctcrw.hpp:152-156); OU: tau, kappa(in sigma); BM: sigma
    CtcrwTrans ctr;
    ScalTrans str;
    // hand-over dumps of the shared-covariance kernels are compact (state + the sensitivities of the wanted
    // directions only, no covariance part): components per lane, 0 = no group uses that layout
    int nstate_clean;
    int all_clean;               // every group dumps the compact layout (no group on the general kernel): the hand-over check need not read group_flags
    int derive;                  // windows >= 1 of the general kernel derive one variance direction from log sigma_obs (k_iso.hip)
    double* chk_out;             // &out[n_out]: zeroed by the main kernel, raised by the finalize kernel's checks
    double* wave_clock;          // debugging (SSDE_WAVE_CLOCK=file at create): [work item][4] = start, end (wall_clock64, 100 MHz), HW_ID, rows
    int deep_prefetch;           // shared-covariance kernel, d = 2: three-block rotation (two blocks in flight) instead of the ping-pong pair
    int stream_nt;               // non-temporal loads of the tile stream (a batch far larger than the Infinity Cache); 0: ordinary loads -- the
                                 // batch fits that cache and is still there at the next evaluation (k_iso_shared.inc: load_obs_block)
    int bnd_stride;              // components per hand-over dump in `bnd` (NSTATE_MAX, or more with drift columns)
    // Row-varying DRIFT on the shared-covariance path (k_iso_drift.hip): mu_a(i) = mu[a] + sum_k coef_k X_k(i) over the
    // streamed design columns that feed dimension a (nllk_ctcrw.hpp:143-149, 211-212; nllk_ou_ssm.hpp:113-124); tau,
    // nu / kappa / sigma and sigma_obs constant.  Tile channels c_col .. c_col + drift_k - 1 hold the columns.
    int drift_k;                 // streamed columns (0 = constant drift: the other kernels)
    int c_col;                   // tile channel of column 0
    double coefA[DRIFT_KMAX];    // coefficient of column k if it feeds dimension 0, else 0
    double coefB[DRIFT_KMAX];    // ... dimension 1
    unsigned drift_dim1;         // bit k: column k feeds dimension 1
    PpDrift pp;                  // pp.nb > 0: tile channels c_col .. c_col + pp.nb - 1 hold the blocks' covariates instead of the columns
    // Row-varying tau / nu (kappa, sigma) on lane = track lanes (k_iso_colvar.hip): the linear predictors are
    // p1(i) = cv_eta0[0] + sum_k coefA[k] X_k(i) and p2(i) = cv_eta0[1] + sum_k coefB[k] X_k(i) over the drift_k streamed columns
    double cv_eta0[2];
    double coefC[DRIFT_KMAX];    // ... and a row-varying drift next to them: column k's coefficient in mu_1, mu_2 (cv_mu_cols != 0)
    double coefD[DRIFT_KMAX];
    int cv_mu_cols;
    double* cv_ranges;           // [workgroup][4]: min / max of p1, min / max of p2 over the workgroup's rows, or NULL
    int cv_full;                 // 4 x 4 covariance lanes (CTCRW, d = 2): per-row H_array and / or a P0 that is not block-identical
    int cv_has_h;                // ... the tiles hold H_array[,,i] in the d^2 channels after the observations
    double cv_p0[16];            // ... P0, column-major
    // The finalising work INSIDE the main launch (iso_shared_kernel; fused_finalize_wave below): the second wave to arrive at a window
    // boundary checks that hand-over, the last wave of the launch forms the sums and publishes -- no dependent second launch.
    int fused;                   // 0: launch_iso_finalize follows
    int fuse_items;              // work items (waves that run a window) of this launch
    unsigned* fuse_arrive;       // [n_chunks - 1][n_groups] arrivals at a boundary (zero between launches)
    unsigned* fuse_done;         // one word: work items that have finished (zero between launches)
    // ... the same models with the gradient by a reverse sweep (k_iso_adj.hip): the state entering every CB-th row of a window
    double* adj_ckpt;            // [work item][adj_ckpt_stride]
    int64_t adj_ckpt_stride;     // doubles per work item: checkpoints of its window x state doubles x 64
    int adj_diag;                // testing (SSDE_ADJ_DIAG): bit 0 = every row load from the window's first rows (what the kernel takes without HBM)
    int adj_tail;                // rows a window that is not the last walks past its end before its backward recursion starts (= window;
                                 // testing: SSDE_ADJ_TAIL, deliberately short)
    // Quiet rows of the general kernel (regular grid; k_iso.hip): a block of ISO_U rows with no missing observation in it or in
    // the quiet_w blocks before it (any lane), past the covariance transient, has the STATIONARY covariance on every lane --
    // the lanes run the mean half with the stationary gains (statc) there and take up their own covariance again, from the
    // stationary values, at the next missing row.
    const unsigned long long* nan_bits;   // [n_groups][nan_words] bit b: a lane of the group misses an observation in rows [ISO_U b, ISO_U (b + 1)), or NULL
    int nan_words;
    int quiet_w;                 // blocks a lane's covariance takes to forget a missing row (0: no quiet rows in this launch)
    int quiet_b0;                // first block past the covariance transient of the initial P0
    double quiet_p[3 + 3 * 3];   // stationary P (CTCRW: p11, p12, p22; OU / BM: p) | its sensitivities, direction-major
    double* quiet_flag;          // one word: the largest disagreement the switches to quiet rows found (raised by the main kernel, folded into
                                 // out[n_out] and cleared by the finalize launch), or NULL
    double quiet_ld;             // log F at the stationary covariance
    double quiet_gld[3];         // ... dF / F per covariance direction
};
// One part of a k_iso_colvar.hip launch: the design columns whose coefficient gradients one wave of the workgroups carries
// (device table, CV_WAVES entries)
constexpr int CV_CMAX = 32;      // channels of a row the kernel can stage: dt, y, H_array entries (d = 2), DRIFT_KMAX columns
constexpr int CV_WAVES = 8;      // waves of a workgroup = parts: two per SIMD
constexpr int CV_KC = 4;         // columns per part (register budget of a wave at two per SIMD)
struct CvPart {
    int32_t n_col;               // columns of this part
    int32_t with_mu;             // the part also carries the drift-intercept direction
    int32_t with_sig;            // ... the log sigma_obs direction
    int32_t chan[CV_KC];         // tile channel of the slot's design column, -1 = a column of ones (an intercept)
    int32_t type[CV_KC];         // 1: the column feeds par[d] (log tau / log sigma), 2: par[d + 1] (log nu / log kappa), 3, 4: mu_1, mu_2
};
// partials [n_parts * n_chunks][2 + CV_KC + d][n_groups]: value | the part's columns | mu_1 .. mu_d | log sigma_obs;
// a.n_parts == CV_WAVES; a.part_mask[0] == 0: the value only (no tangents); kc: the widest part's column count
hipError_t launch_iso_colvar(int model, int d, const IsoArgs& a, const CvPart* parts, int kc, hipStream_t s);
int colvar_nstate(int model, int d, int kc, bool full);
hipError_t launch_colvar_range_reduce(const double* wg, int n_wg, double* out_pinned /* 4 doubles, host-visible */, hipStream_t s);
hipError_t launch_cols_differ(const double* a, const double* b, int64_t n, int* differ /* device, zeroed */, hipStream_t s);
// row-varying tau / nu with at most 2 CV_FEW_K streamed columns and 2 CV_KC tangents: one wave per (group, window) (iso_few_kernel);
// kc = CV_KC or 2 CV_KC tangent slots (parts[0], parts[1])
constexpr int CV_FEW_K = 4;
hipError_t launch_iso_few(int model, int d, const IsoArgs& a, const CvPart* parts, int kc, hipStream_t s);
// constant tau / nu with per-row H_array (CTCRW, d = 2): one wave per (group, window), the tangents of parts[0] (columns of ones)
hipError_t launch_iso_full(int model, const IsoArgs& a, const CvPart* parts, hipStream_t s);
// row-varying tau / nu (and drift), gradient by a reverse sweep: one wave per (group, window), two passes (k_iso_adj.hip)
hipError_t launch_iso_adj(int model, int d, const IsoArgs& a, hipStream_t s);
int adj_ks(int k);                                   // streamed-column capacity of the instantiation that takes k columns, -1: none
int adj_nk(int model, int d, bool mu);               // accumulators per streamed column
int adj_nacc(int model, int d, int k, bool mu);      // [value | log sigma_obs | mu_a | par[d] | par[d + 1] | column 0's kinds | ...]
int adj_nstate(int model, int d, bool full);         // doubles of a hand-over record: the state and its adjoint
int adj_ckpt_rows(int model, int d, bool full);      // rows between checkpoints
int adj_items(int n_groups, int n_chunks);           // work items (waves) of a launch
hipError_t launch_colvar_h_stats(const TileView& tv, int c_h, int d, double* out /* [n_groups][2]: max diag(H), max |H01 - H10| */, hipStream_t s);
hipError_t launch_colvar_ranges(const TileView& tv, int c_col, int K, double* out /* [n_groups][K][2] */, hipStream_t s);
// components of a compact hand-over dump (shared-covariance kernels): state, one block per wanted covariance
// direction, one block for mu
__host__ __device__ constexpr inline int shared_nstate(int sd, int mask, bool has_p2) {
    return (1 + ((mask & DIR_SIG) ? 1 : 0) + ((mask & DIR_P1) ? 1 : 0) + (((mask & DIR_P2) && has_p2) ? 1 : 0) +
            ((mask & DIR_MU) ? 1 : 0)) * sd;
}
hipError_t launch_iso(int model, int d, const IsoArgs& a, bool any_dirty, hipStream_t s);
// blocks of iso_block_rows() rows in which some lane of a group misses an observation: bits [n_groups][nwords] (k_iso.hip)
int iso_block_rows(int model);
hipError_t launch_nan_blocks(const TileView& tv, int d, int block_rows, unsigned long long* bits, int nwords, hipStream_t s);
// ev0 / ev1 (may be NULL): stamped with the kernel's own begin / end
struct ReduceArgs;
hipError_t launch_iso_shared(int model, int d, const IsoArgs& a, const ReduceArgs& r, hipStream_t s, hipEvent_t ev0, hipEvent_t ev1);
// shared-covariance lanes with a streamed row-varying drift (k_iso_drift.hip); partials [n_chunks][4 + d + drift_k][n_groups]
hipError_t launch_iso_drift(int model, int d, const IsoArgs& a, hipStream_t s, hipEvent_t ev0, hipEvent_t ev1);
int drift_nstate(int model, int d, int k);
// the same model with the lane's own covariance (missing rows / irregular grid): general step + column recursions
hipError_t launch_iso_drift_general(int model, int d, const IsoArgs& a, hipStream_t s);
int drift_general_nstate(int model, int d, int k);
// exact Hessian over the drift coefficients, shared-covariance case (k_iso_drift.hip: iso_drift_hess_kernel)
constexpr int DRIFT_HESS_MAXN = DRIFT_KMAX + 2;     // streamed columns + one intercept per dimension
struct DriftHessArgs {
    int n;                         // slots
    const int16_t* chan;           // device [n]: tile channel of the slot's column, or -1: an intercept (a column of ones)
    const int16_t* tile_i;         // device [n_tiles]
    const int16_t* tile_j;
    double* partials;              // [n_tiles][HESS_T^2][n_chunks][n_groups]
    double* hess;                  // [n x n] column-major
};
hipError_t launch_iso_drift_hess(int model, const IsoArgs& a, const DriftHessArgs& hx, int n_tiles, hipStream_t s);
hipError_t launch_window_check(int model, int d, const IsoArgs& a, hipStream_t s);
struct ReduceArgs;
// the hand-over checks and the final sums of an isotropic evaluation in ONE launch (the checks raise out[n_out])
hipError_t launch_iso_finalize(int model, int d, const IsoArgs& a, const ReduceArgs& r, hipStream_t s);
void fill_stat_consts(int model, int d, IsoArgs& a);
int iso_nstate(int model, int d);

// ---- final deterministic reduction (k_reduce.hip) --------------------------------------------
// out[0]      = sum over the first n_value_parts parts (the time windows of direction part 0) and g
//               of partials[part][acc 0][g]        (every direction part recomputes the nllk)
// out[slot]   = sum over the (part, k >= 1) pairs with map[(part / chunks_per_part)*(nacc-1) + k-1] == slot
// out[n_out]  = max of chk[0..n_chk)  (window hand-over check; 0 when there is nothing to check)
// One workgroup per output slot, fixed summation order: bitwise reproducible.
struct ReduceArgs {
    const double* partials;       // [n_parts][nacc][n_blocks]
    int n_parts, nacc, n_blocks;
    int n_value_parts, chunks_per_part;
    int n_out;                    // 1 + n_par_full
    int kfast;                    // partials laid out [part][block][acc] instead of [part][acc][block] (the kernel whose last wave forms the sums
                                  // itself reads all accumulators of an entry with one wide load: fused_finalize_wave)
    const double* chk;
    int n_chk;
    double add[4];                // data-independent terms of the shared-covariance path, added to
    int16_t add_slot[4];          // out[add_slot[i]] (slot -1 = unused)
    int16_t map[MAX_PAR + 16];    // -> output slot (1 + full-par index) or -1
    double* out;                  // n_out + 1 doubles
    // Publication of the result to the host (synchronous ssde_eval): the LAST workgroup of the reducing launch to finish copies
    // out[0 .. n_out] into host-visible pinned memory and then stores the evaluation's sequence number, which the host
    // spins on -- no read-back copy, no synchronisation call (tools/microbench_latency.hip: 5 us less than a stream
    // synchronisation, 7 us less than an asynchronous copy + synchronisation).  pub == NULL: not wanted.
    double* pub;                  // pinned, n_out + 1 doubles
    unsigned long long* pub_flag; // pinned
    unsigned long long pub_seq;
    unsigned int* pub_count;      // device, zero between launches: workgroups of this launch that have finished
    int pub_blocks;               // workgroups of this launch
};
hipError_t launch_reduce(const ReduceArgs& a, hipStream_t s);

// ---- ingest (k_ingest.hip): long format -> tiles ------------------------------------------------
struct IngestArgs {
    const double* times;
    const double* obs;           // n x d column-major
    const double* h_array;       // d x d x n or NULL
    const double* const* cols;   // device array of ncols column pointers (each length n) or NULL
    int ncols;
    int d;
    int64_t n;
    const int64_t* lane_row0;    // [n_groups*64], -1 = empty lane
    const int32_t* lane_nsteps;
    const int64_t* group_off;
    const int32_t* group_len;
    int n_groups, C, c_obs;
    double* tiles;
    double* a0;                  // [n_groups][sdim][64] (written from obs when a0_src == NULL)
    const double* a0_src;        // caller-supplied a0 [n_seg x sdim] column-major or NULL
    const int64_t* lane_seg;     // [n_groups*64] segment index of the lane's track (for a0_src)
    int64_t n_seg;
    int sdim, model;
    double* dt_minmax;           // [n_groups * ychunks * 3]: min / max of the intervals used INSIDE tracks, NaN seen
    int ychunks;
    double last_dt;              // dtimes(n-1): 1 (nllk_ctcrw.hpp:126-129), or the interval to the next shard's first row
};
int ingest_ychunks(int n_groups);
hipError_t launch_ingest(const IngestArgs& a, hipStream_t s);

// per-row flags of the direct families: bit i set = row i is scored (ID(i-1) == ID(i))
hipError_t launch_scored_mask(const double* id, int64_t n, uint32_t* mask, hipStream_t s);
// first-row flags for segment discovery on device data
hipError_t launch_first_flags(const double* id, int64_t n, uint8_t* flags, hipStream_t s);
hipError_t launch_seg_nan(const double* obs, int64_t n, int d, const int64_t* starts, int64_t n_seg, int* flags, hipStream_t s);
hipError_t launch_used_dt_minmax(const double* id, const double* times, int64_t n, double* out, int n_blocks, hipStream_t s);   // k_lattice.hip
hipError_t lattice_positions(const double* id, const double* times, int64_t n, double delta, double rtol, int64_t* inc, int64_t* pos,
                             int64_t* rep, int* bad_dev, int64_t* n_lattice, int* bad_host, hipStream_t s);
hipError_t launch_gather_i64(const int64_t* src, const int64_t* idx, int64_t m, int64_t* dst, hipStream_t s);
hipError_t launch_lattice_scatter(const int64_t* pos, const double* id, const double* times, const double* obs, int64_t n, int d,
                                  int64_t np, double delta, double* times_p, double* obs_p, hipStream_t s);
hipError_t launch_lattice_gather(const int64_t* pos, const double* src, int64_t n, int64_t np, int ncol, double* dst, hipStream_t s);
hipError_t launch_h_couples(const double* H, int64_t n, int D, int* couples, hipStream_t s);
hipError_t launch_h_block(const double* H, int64_t n, int D, int dlo, int cnt, double* out, hipStream_t s);
hipError_t launch_na_follow(const double* id, const double* lead, double* col, int64_t n, int any_nan, int* poison, hipStream_t s);

// ---- general parameter description for the dense / direct kernels -------------------------------
// Every coefficient of the linear predictor (nllk_ctcrw.hpp:143-149) is a "slot": SDE parameter
// j it feeds, the streamed design column that multiplies it (or none = intercept column of
// ones), and its index in the full parameter vector.  Slots are ordered parameter by parameter,
// fixed-effect columns first.  The table lives in HBM (uploaded once) and is read with scalar
// loads; the parameter vector is uploaded per evaluation.
struct SlotTable {
    int n_slots;
    int q;
    int16_t par_j[MAX_COLS];   // SDE parameter fed by the slot
    int16_t col[MAX_COLS];     // streamed column index, or -1 for an intercept
    int16_t pidx[MAX_COLS];    // index into the full parameter vector
    int16_t is_free[MAX_COLS]; // 0 = held fixed (TMB map): no gradient wanted
    int16_t decay[MAX_COLS];   // decaying column: index into log_decay, else -1 (direct families)
};
constexpr int MAX_DECAY = 4;   // decay rates (direct families)

// ---- direct families (k_direct.hip) ---------------------------------------------------------------
struct DirectArgs {
    const double* times;
    const double* obs;           // n x d column-major (engine-owned copy)
    const double* const* cols;   // device array of streamed column pointers (each length n)
    const uint32_t* scored;      // bit i: row i is scored
    int64_t n;
    int d, model, any_nan;
    const SlotTable* slots;      // device
    const double* par;           // device, full parameter vector
    int n_slots;                 // == slots->n_slots (host copy, selects the kernel variant)
    int n_blocks;
    double* partials;            // [1][1 + n_slots][n_blocks]
    double tdf, tconst;          // BM_t: degrees of freedom and the normalising constant of dt(., df)
    const double* t_decay;       // [q * n] or NULL: decaying columns are scaled by exp(-exp(log_decay) * t_decay) (nllk_sde.hpp:47-57)
    int n_decay, off_decay;      // decay rates and where log_decay sits in the parameter vector
};
hipError_t launch_direct(const DirectArgs& a, hipStream_t s);

// Fast direct kernel: at most two SDE parameters have streamed design columns (the common case: one or
// two smooth terms); their column blocks are contiguous in the engine-owned column buffer, so a column
// is base + c * n, and the slot -> parameter routing is static inside the kernel.
constexpr int DIRECT_KCAP = 24;   // streamed columns per parameter the fast kernel can hold
constexpr int PP_LDS = 1536;      // doubles of LDS per basis-evaluated group: (n_knots - 1) * K * 4 coefficients + the knots
// a design block evaluated from a piecewise-cubic table (include/ssde.h: ssde_ppbasis) instead of streamed
struct PPRef {
    const double* x;              // [n] covariate in HBM, NULL = the group is streamed
    const double* knots;          // [nk] (HBM)
    const double* tab;            // [(nk - 1) * K * 4] (HBM)
    int nk;
    int uniform;                  // equally spaced knots: interval by multiplication
    double k0, inv_h;
};
hipError_t launch_pp_materialise(const PPRef& pp, int K, int64_t n, double* dst, int64_t stride, hipStream_t s);
struct DirectFastArgs {
    const double* times;
    const double* obs;
    const uint32_t* scored;
    int64_t n;
    int d, model, any_nan;
    int n_blocks;
    double* partials;             // [1 + n_acc][n_blocks], accumulators: q intercept slots, then ncA, then ncB
    double base[MAX_Q];           // intercept part of every parameter (working scale)
    int has_icpt[MAX_Q];          // parameter has an intercept coefficient (its gradient is wanted)
    int ja, jb;                   // parameters with streamed columns (-1 = none)
    int ncA, ncB;
    int64_t col_stride;           // doubles between consecutive streamed columns
    const double* colA;           // first streamed column of parameter ja
    const double* colB;
    double coefA[DIRECT_KCAP], coefB[DIRECT_KCAP];
    int uniform_dt;               // every scored interval equals dt_uniform
    double dt_uniform;
    double tdf, tconst;           // BM_t
    PPRef ppA, ppB;               // groups evaluated on the fly from a basis table (x != NULL)
};
hipError_t launch_direct_fast(const DirectFastArgs& a, hipStream_t s);
// exact Hessian of the data term over coefficients of the linear predictor, BM / OU (k_direct_hess.hip)
struct DirectHessArgs {
    const double* times;
    const double* obs;
    const double* const* cols;
    const uint32_t* scored;
    int64_t n;
    int d, model, any_nan;
    double tdf;                  // BM_t: degrees of freedom (other_data(0), tr_dens.hpp:40)
    const SlotTable* slots;      // device
    const double* par;           // device, full parameter vector
    int n_slots;
    const double* t_decay;       // [q * n] or NULL: decaying columns (nllk_sde.hpp:47-57), as DirectArgs
    int n_decay, off_decay;
    int nu;                      // wanted unknowns
    const int16_t* uslot;        // device [nu]: slot of every wanted coefficient; -1 - m: log_decay_m (n_decay > 0 only)
    const int16_t* tile_i;       // device [n_tiles]: tile (ti, tj), ti <= tj
    const int16_t* tile_j;
    double* partials;            // [n_tiles][HESS_T * HESS_T][n_blocks]
    double* hess;                // [nu x nu] column-major
};
hipError_t launch_direct_hess(const DirectHessArgs& a, int n_tiles, int n_blocks, hipStream_t s);
hipError_t launch_dt_minmax(const double* times, const uint32_t* scored, int64_t n, double* out2_per_block, int n_blocks,
                            hipStream_t s);

// ---- dense / time-varying Kalman (k_dense.hip) ----------------------------------------------------
constexpr int DENSE_NT = 2;      // tangent directions per lane
constexpr int DENSE_MAXD = 8;    // widest response run as ONE filter (coupling H_array / P0); wider ones only pair by pair
struct DenseDir {                // one gradient direction
    int16_t kind;                // 0 none, 1 log_sigma_obs, 2 coefficient slot
    int16_t slot;                // slot index (kind 2)
    int16_t pidx;                // index into the full parameter vector
    int16_t pad;
};
struct DenseArgs {
    TileView tv;
    int model, d, any_nan, has_h;
    const SlotTable* slots;      // device
    const double* par;           // device
    int n_slots;
    double p0[256];              // sdim x sdim column-major (sdim <= 16: CTCRW with eight response columns)
    int n_dirblocks;
    const DenseDir* dirs;        // device [n_dirblocks * DENSE_NT]
    double* partials;            // [n_dirblocks][1 + DENSE_NT][n_groups]
    double* report;              // optional aest_all
    const int64_t* lane_row0;
    int64_t n;
    double last_dt;              // dtimes(n-1), see IngestArgs
    PpDrift pp;                  // pp.nb > 0: the tiles hold covariates, the slots' columns come out of the blocks' tables (REPORT of such a handle)
};
hipError_t launch_dense(const DenseArgs& a, bool want_grad, hipStream_t s);

// ---- row-varying-coefficient isotropic Kalman (k_tv.hip, ssde_tv.hpp) -----------------------------
// Long-format data (no tiles): a row-parallel pre-pass writes one 128-byte record per row, the
// serial recursion runs one WAVE per (pack of tracks, time window, direction block) with
// lane = (track of the pack, gradient direction).
constexpr int TV_U = 4;           // rows per prefetch block (divides WIN_ALIGN)
constexpr int TV_NSTATE = 40;     // doubles per lane dumped at a window hand-over (dense CTCRW, d = 2: 2 (4 + 16))
constexpr int TV_LEAN_BLOCKS = 64, TV_LEAN_ITEMS = 2048;   // the lean replay (TvArgs.par0_w): pre-pass blocks / work items at most
constexpr int TV_STATS = 8;       // per block: min/max of dt, par[d], par[d+1], largest diag(H) over the rows the filter propagates
struct TvItem { int32_t pack, c, nc, b; };        // work item of one wave: track pack, window c of nc, direction block
struct TvDir { int16_t kind, dim, pidx, slot; };  // TVK_* kind, dimension (TVK_MU), full-par index, coefficient slot
struct TvArgs {
    const double* times;         // [n]
    const double* obs;           // [n x d] column-major
    const double* colbuf;        // streamed design columns, column c at colbuf + c * col_stride
    int64_t col_stride;
    const uint32_t* scored;      // bit i: ID(i-1) == ID(i)
    int64_t n;
    int d, model, any_nan;
    const SlotTable* slots;      // device
    int n_slots;
    const double* par;           // device, full parameter vector
    double* rec;                 // [n][TV_RS] per-evaluation row records
    const double* wdir;          // [n][ndp] d par_row / d coefficient of every direction (built once)
    int ndp;                     // directions padded to a multiple of the lanes per track
    int lpt_shift;               // lanes per track = 1 << lpt_shift; tracks per wave = 64 >> lpt_shift
    const TvDir* dirs;           // [ndp]
    const int64_t* trk_row0;     // [n_tracks] first row of the track (tracks sorted by length, longest first)
    const int32_t* trk_ns;       // [n_tracks] rows - 1
    const double* a0;            // [n_tracks][sdim]
    int64_t n_tracks;
    const TvItem* items;
    int n_items;
    int window;                  // warm-up rows of a time window
    double h;                    // sigma_obs^2
    int h_from_par;              // 1: the kernels take sigma_obs^2 = exp(2 par[0]) themselves (hipGraph replay)
    double p0[3];
    int dense;                   // 1: full-covariance lanes (per-row H_array and / or a P0 that is not block-identical)
    int has_h;
    const double* h_array;       // [d x d x n] or NULL
    double p0f[16];              // sdim x sdim column-major (dense lanes)
    const double* eseal_h;       // ESEAL_SSM: daily drift dives h_i and non-lipid tissue mass R_i (nllk_e_seal_ssm.hpp:100-101)
    const double* eseal_R;
    double* bnd;                 // [n_items][2][TV_NSTATE][64]
    double* gval;                // [n_items][64] per-lane nllk
    double* gdir;                // [n_items][64] per-lane d nllk / d direction
    double* stats;               // [stats_blocks][TV_STATS]
    int stats_blocks;
    double* report;              // optional aest_all [n x sdim]
    int n_out;                   // 1 + n_par_full
    double last_dt;              // dtimes(n-1), see IngestArgs
    int16_t dir_of_par[MAX_PAR]; // full-par index -> direction, -1 = no gradient (fixed, or not in the data term)
    double* out;                 // n_out + 1 doubles
    // The lean form of a replayed evaluation (few rows: C1, one animal -- every graph node costs ~4 us there): `par`, `stats` and `out`
    // are PINNED HOST memory the kernels read / write directly (no copy nodes); sigma_obs reaches the filter through a device word the
    // pre-pass fills, and the hand-over checks land per item in pinned memory, the host takes their maximum.
    double* par0_w;              // device, 1 double: the pre-pass writes par[0] here (NULL: the filter reads par[0] itself)
    double* chk_items;           // pinned, [n_items]: each item's hand-over check (NULL: atomicMax into out[n_out])
};
hipError_t launch_tv_weights(const TvArgs& a, hipStream_t s);
hipError_t launch_tv_a0(const TvArgs& a, const double* a0_src, const int64_t* trk_seg, int64_t n_seg, int sdim,
                        double* a0_dst, hipStream_t s);
hipError_t launch_tv_prepare(const TvArgs& a, hipStream_t s);
hipError_t launch_tv_filter(const TvArgs& a, bool want_grad, hipStream_t s);
hipError_t launch_tv_finalize(const TvArgs& a, hipStream_t s);

// ---- exact second derivatives on the lane = direction path (k_tv_hess.hip, ssde_hdual.hpp) -----------------------------
// one wavefront lane per coefficient PAIR, the primal recursion in hyper-dual arithmetic; isotropic lanes only
constexpr int HESS_RS = 12;       // doubles per row record: dt | the row's linear predictors p_0 .. p_3 | y_0 y_1 | 0 | H_array[,,i] (d x d, column-major; full-covariance lanes)
constexpr int HESS_NSTATE = 80;   // doubles per lane dumped at a window hand-over: (2 D + 3) hyper-dual numbers on the isotropic lanes, sd + sd^2 = 20 on the
                                  // full-covariance ones (CTCRW, d = 2)
struct TvHessArgs {
    const double* times;         // [n]
    const double* obs;           // [n x d] column-major
    const double* colbuf;        // streamed design columns
    int64_t col_stride;
    int64_t n;
    int d, model, any_nan;
    const SlotTable* slots;      // device
    int n_slots;
    const double* par;           // device, full parameter vector
    double* rec;                 // [n][HESS_RS]
    const double* wdir;          // [n][ndp]: d par_row / d coefficient of every direction (TvArgs.wdir)
    int ndp;
    const TvDir* dirs;           // [ndp]
    const int64_t* trk_row0;     // [n_tracks]
    const int32_t* trk_ns;
    const double* a0;            // [n_tracks][sdim]
    const TvItem* items;         // (track, window c of nc, block of 64 pairs), ordered track / pair block / window
    int n_items;
    int window;                  // warm-up rows
    const int16_t* pair_a;       // [n_pairs] direction of the pair's first coefficient (index into dirs / wdir)
    const int16_t* pair_b;
    int n_pairs, n_pb;           // pairs, blocks of 64 pairs
    double p0[3];
    int dense, has_h;            // full-covariance lanes (per-row H_array and / or a P0 that is not block-identical: ssde_dense.hpp in hyper-dual arithmetic)
    const double* h_array;       // [n][d x d] (has_h)
    const double* eseal_h;       // ESEAL_SSM: [n] h_i and R_i (nllk_e_seal_ssm.hpp:43-59)
    const double* eseal_R;
    double p0_full[16];          // sd x sd, column-major
    double last_dt;
    double* bnd;                 // [n_items][2][HESS_NSTATE][64]
    double* part;                // [n_items][4][64]: ab | a | b | value part of every lane's likelihood
    double* out;                 // [4][n_pb][64] sums over (track, window) + [1] largest hand-over disagreement
};
hipError_t launch_tv_hess(const TvHessArgs& a, hipStream_t s);

// ---- device helpers -------------------------------------------------------------------------------
// Window geometry shared by the kernels, the hand-over check and the engine.
//   t0 == 0 : n_chunks equal windows over [0, L)
//   t0 >  0 : window 0 is the covariance transient [0, t0) (no warm-up: it starts from the true initial
//             state) and windows 1..n_chunks-1 split [t0, L) equally
// Rows [s_acc, s_end) are scored, rows [s_begin, s_acc) warm up.  All bounds are multiples of WIN_ALIGN
// (except L itself).
//             delta = how many stationary rows the transient window is worth (it runs on the wave of window 1)
__host__ __device__ inline void window_bounds(int L, int n_chunks, int window, int t0, int c, int& s_begin, int& s_acc,
                                              int& s_end, int delta_rows = -1) {
    if (n_chunks <= 1) { s_begin = 0; s_acc = 0; s_end = L; return; }
    // Equal windows are equal up to ONE alignment unit: `units` units of WIN_ALIGN rows are dealt to `nw` windows, the first
    // units % nw of them get one more.  (Round 3 rounded every window's length up instead: one rank's share of a strong-scaled
    // batch -- 42 windows over 10^4 rows -- ran 256-row windows where 241 were needed, its last two windows were empty and every
    // other wave walked 5 % more rows than its share; profiles/r04_a_wave_clock_share8.txt.)
    if (t0 > 0) {
        if (c == 0) { s_begin = 0; s_acc = 0; s_end = L < t0 ? L : t0; return; }
        // the wave that owns window 1 also runs window 0 first (and window 0's rows cost more): window 1
        // is shortened by delta so that all waves finish together
        const int delta = delta_rows >= 0 ? delta_rows : (3 * t0 / 2 + WIN_ALIGN - 1) / WIN_ALIGN * WIN_ALIGN;
        const int rest = L > t0 ? L - t0 : 0;
        const int nw = n_chunks - 1;
        const int units = (rest + delta + WIN_ALIGN - 1) / WIN_ALIGN, per = units / nw, extra = units % nw;
        const int cl = (per + (extra ? 1 : 0)) * WIN_ALIGN;         // the longest window
        if (cl <= delta + WIN_ALIGN) {
            // Short tracks (C2: 10^3 rows): the transient is worth more than a whole window.  Its wave gets a token
            // window 1 and windows 2.. share the rest EQUALLY (the geometry above would leave the last ones empty).
            const int w1 = rest < 2 * WIN_ALIGN ? rest : 2 * WIN_ALIGN;
            if (c == 1) { s_acc = t0; s_end = n_chunks > 2 ? t0 + w1 : L; }
            else {
                const int nw2 = n_chunks - 2;
                const int u2 = (rest - w1 + WIN_ALIGN - 1) / WIN_ALIGN, per2 = u2 / nw2, ex2 = u2 % nw2;
                const int k = c - 2;
                s_acc = t0 + w1 + (k * per2 + (k < ex2 ? k : ex2)) * WIN_ALIGN;
                s_end = t0 + w1 + ((k + 1) * per2 + (k + 1 < ex2 ? k + 1 : ex2)) * WIN_ALIGN;
            }
            if (s_acc > L) s_acc = L;
            if (s_end > L) s_end = L;
        } else {
            const int e0 = t0 - delta + ((c - 1) * per + (c - 1 < extra ? c - 1 : extra)) * WIN_ALIGN;   // end of window c - 1
            const int e1 = t0 - delta + (c * per + (c < extra ? c : extra)) * WIN_ALIGN;
            s_acc = (c == 1) ? t0 : e0; if (s_acc > L) s_acc = L;
            s_end = e1; if (s_end > L) s_end = L; if (s_end < s_acc) s_end = s_acc;
        }
    } else {
        const int units = (L + WIN_ALIGN - 1) / WIN_ALIGN, per = units / n_chunks, extra = units % n_chunks;
        s_acc = (c * per + (c < extra ? c : extra)) * WIN_ALIGN; if (s_acc > L) s_acc = L;
        s_end = ((c + 1) * per + (c + 1 < extra ? c + 1 : extra)) * WIN_ALIGN; if (s_end > L) s_end = L;
    }
    s_begin = s_acc - window; if (s_begin < 0) s_begin = 0;
}

#if defined(__HIPCC__)
// Every workgroup of a reducing launch calls this when it is done (ALL its threads).  `dep` is what thread 0 got back from
// the device-scope atomic that carried the block's result (publish_store / publish_max below): the count is made to
// depend on it, so the result has been performed at the device's coherence point before the block is counted -- without
// a release fence, i.e. without the L2 write-back + invalidate that ~800 workgroups would each pay (measured: that
// version was 6-10 us SLOWER than a blocking read-back copy).  The last workgroup to arrive publishes (ReduceArgs.pub).
__device__ __forceinline__ void publish_if_last(const ReduceArgs& R, unsigned long long dep) {
    if (!R.pub) return;
    if (threadIdx.x >= WAVE) return;               // wave 0 (thread 0 carried the block's result)
    unsigned old = 0;
    if (threadIdx.x == 0) {
        // A data dependency on the atomic's RETURN value, one the compiler cannot fold away (`1 + (dep & 0)` was folded: the
        // exchange became a non-returning atomic and the add followed it with no s_waitcnt in between -- ADVICE r03): the
        // operand of the add is produced by instructions that read the returned register, after an explicit wait for it.
        const unsigned lo = (unsigned)dep;
        unsigned one;
        asm volatile("s_waitcnt vmcnt(0)\n\tv_and_b32 %0, 0, %1\n\tv_add_u32 %0, 1, %0" : "=v"(one) : "v"(lo) : "memory");
        old = __hip_atomic_fetch_add(R.pub_count, one, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    old = __shfl(old, 0, WAVE);
    if (old != (unsigned)R.pub_blocks - 1u) return;
    for (int k = threadIdx.x; k <= R.n_out; k += WAVE)       // one coalesced round trip, past this XCD's L2
        R.pub[k] = __hip_atomic_load(&R.out[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __threadfence_system();                        // the wave's stores have reached the host before the flag does
    if (threadIdx.x == 0) {
        __hip_atomic_store(R.pub_count, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // ready for the next launch
        __hip_atomic_store(R.pub_flag, R.pub_seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}
// a block's result slot, written where every other workgroup of the device will see it; returns the dependency token
__device__ __forceinline__ unsigned long long publish_store(double* p, double v) {
    return (unsigned long long)__hip_atomic_exchange((unsigned long long*)p, (unsigned long long)__double_as_longlong(v), __ATOMIC_RELAXED,
                                                     __HIP_MEMORY_SCOPE_AGENT);
}

// Workgroup id -> (track group, window, part).  Workgroups are dealt round-robin over the 8
// XCDs, so ids that are equal mod 8 share an XCD (and its L2): the parts of one (group, window)
// get such ids because they stream the same rows.
// nc = number of windows the GRID enumerates (the shared-covariance kernel enumerates windows 1.. only)
__device__ __forceinline__ bool decode_block(const IsoArgs& A, int nc, int& g, int& part, int& chunk) {
    const int id = blockIdx.x * WG_WAVES + (threadIdx.x >> 6);   // one work item per WAVE
    const int np = A.n_parts;
    const int hi = id >> 3;  // ((g/8) * nc + chunk) * np + part
    part = hi % np;
    chunk = (hi / np) % nc;
    g = (hi / (np * nc)) * 8 + (id & 7);
    if (A.use_group_list) {
        if (g >= A.n_dirty_groups) return false;
        g = A.dirty_groups[g];
        return true;
    }
    return g < A.tv.n_groups;
}

__device__ __forceinline__ bool group_selected(const IsoArgs& A, int g) {
    if (A.group_mode == 0) return true;
    const bool clean = (A.group_flags[g] & 1) != 0;
    return A.group_mode == 2 ? clean : !clean;
}


__device__ __forceinline__ double wave_sum(double x) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) x += __shfl_xor(x, o, 64);
    return x;
}

// one output slot by one 256-thread workgroup: 0 = nllk, 1.. = gradient entries, n_out = hand-over check
__device__ __forceinline__ unsigned long long reduce_slot(const ReduceArgs& A, int slot, double* sh) {
    const int tid = threadIdx.x;
    double acc = 0.0;
    if (slot == A.n_out) {
        for (int b = tid; b < A.n_chk; b += 256) acc = fmax(acc, A.chk[b] == A.chk[b] ? A.chk[b] : INFINITY);
        sh[tid] = acc;
        __syncthreads();
        for (int o = 128; o > 0; o >>= 1) {
            if (tid < o) sh[tid] = fmax(sh[tid], sh[tid + o]);
            __syncthreads();
        }
        return tid == 0 ? publish_store(&A.out[slot], sh[0]) : 0ull;
    }
    // a run = `chunks` parts that feed the same slot, n_blocks entries each, `stride` doubles apart; walked as one
    // flat index range with four independent loads in flight per thread (fixed order: bitwise reproducible)
    const int64_t stride = (int64_t)A.nacc * A.n_blocks;
    // accumulator k of the parts [p0, p0 + chunks): entry i = c n_blocks + b in the order the sums have always been formed
    auto sum_run = [&](int p0, int k, int chunks) {
        const double* base = A.kfast ? A.partials + (int64_t)p0 * stride + k : A.partials + ((int64_t)p0 * A.nacc + k) * A.n_blocks;
        const int total = chunks * A.n_blocks;
        for (int i0 = tid; i0 < total; i0 += 4 * 256) {
            double v[4];
#pragma unroll
            for (int u = 0; u < 4; u++) {
                const int i = i0 + u * 256;
                const int c = i / A.n_blocks, b = i - c * A.n_blocks;
                v[u] = i < total ? (A.kfast ? base[(int64_t)i * A.nacc] : base[c * stride + b]) : 0.0;
            }
            acc += (v[0] + v[1]) + (v[2] + v[3]);
        }
    };
    if (slot == 0) {
        sum_run(0, 0, A.n_value_parts);                                              // accumulator 0
    } else {
        const int nk = A.nacc - 1, cpp = A.chunks_per_part;
        for (int pg = 0; pg * cpp < A.n_parts; pg++)
            for (int k = 1; k < A.nacc; k++)
                if (A.map[pg * nk + (k - 1)] == slot)
                    sum_run(pg * cpp, k, cpp);
    }
    sh[tid] = acc;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (tid < o) sh[tid] += sh[tid + o];
        __syncthreads();
    }
    if (tid == 0) {
        double r = sh[0];
        for (int i = 0; i < 4; i++)
            if (A.add_slot[i] == slot) r += A.add[i];
        return publish_store(&A.out[slot], r);
    }
    return 0ull;
}
// ---- the finalising work fused into the main launch -----------------------------------------------------------------------------------
// iso_finalize_kernel is a dependent launch of ~10 us (+ the gap before it) that re-reads what the main kernel's waves have just
// written.  Fused: a wave that has run its window(s) announces itself at the boundaries it shares with its neighbours; the SECOND
// wave to arrive at a boundary compares the two hand-over records (the arithmetic of window_check_block, one wave instead of four:
// maxima, order-free); the LAST wave of the launch forms the fixed-order sums -- reduce_slot's order exactly, its 256 threads as four
// virtual threads per lane, so the result is bitwise the two-launch one -- and publishes them.
// The records and partial sums cross XCDs (each has its own L2): they are written and read with DEVICE-scope accesses (dev_store /
// dev_load: past the local L2), a wave waits for its stores (s_waitcnt vmcnt(0)) before it announces them with a relaxed device-scope
// atomic -- and no fence anywhere: a release / acquire fence here is a write-back / invalidate of the XCD's whole L2, a thousand of
// them per launch cost 0.1-0.15 ms (measured: 0.477 against 0.328 ms on the headline batch).
__device__ __forceinline__ void dev_store(double* p, double v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ double dev_load(const double* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void dev_stores_done() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

__device__ __forceinline__ double window_check_wave(const IsoArgs& A, int nstate, int g, int c) {
    const int lane = threadIdx.x & 63;
    const TileView& tv = A.tv;
    const int L = tv.group_len[g];
    const int ns = tv.lane_nsteps[g * WAVE + lane];
    int sb_, s_next, se_;
    window_bounds(L, A.n_chunks, A.window, A.t0, c + 1, sb_, s_next, se_, A.t0_delta);
    const bool valid = (ns > s_next) && (s_next < L);
    const double* out_c = A.bnd + (((int64_t)c * tv.n_groups + g) * 2 + 1) * A.bnd_stride * WAVE + lane;
    const double* in_n = A.bnd + (((int64_t)(c + 1) * tv.n_groups + g) * 2 + 0) * A.bnd_stride * WAVE + lane;
    double worst = 0.0;
    for (int k0 = 0; k0 < nstate; k0 += 8) {
        double av[8], bv[8];
#pragma unroll
        for (int i = 0; i < 8; i++) {
            const bool on = k0 + i < nstate;
            av[i] = on ? dev_load(out_c + (k0 + i) * WAVE) : 0.0;
            bv[i] = on ? dev_load(in_n + (k0 + i) * WAVE) : 0.0;
        }
#pragma unroll
        for (int i = 0; i < 8; i++) {
            if (k0 + i >= nstate) break;
            const double a = valid ? av[i] : 0.0, b = valid ? bv[i] : 0.0;
            double err = fabs(a - b), sc = fmax(fabs(a), fabs(b));
            if (valid && !(err == err)) err = INFINITY;
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) {
                err = fmax(err, __shfl_xor(err, o, 64));
                sc = fmax(sc, __shfl_xor(sc, o, 64));
            }
            if (err > 0.0) worst = fmax(worst, err / sc);
        }
    }
    return worst;
}
// reduce_slot for ALL NACC accumulators of a one-part launch by ONE wave, bitwise: lane l carries the virtual threads l, l + 64,
// l + 128, l + 192 of the 256-thread form; the partial sums are laid out [window][group][accumulator] (ReduceArgs.kfast), so an
// entry's accumulators arrive with one wide load and every load of the pass is in flight before the first sum
template <int NACC>
__device__ __forceinline__ void reduce_all_wave(const ReduceArgs& A, double (&r)[NACC]) {
    const int lane = threadIdx.x & 63;
    double acc[NACC][4];
#pragma unroll
    for (int k = 0; k < NACC; k++) acc[k][0] = acc[k][1] = acc[k][2] = acc[k][3] = 0.0;
    const int total = A.n_value_parts * A.n_blocks;
#pragma unroll
    for (int j = 0; j < 4; j++) {
        const int tid = lane + 64 * j;
        for (int i0 = tid; i0 < total; i0 += 4 * 256) {
            double v[4][NACC];
#pragma unroll
            for (int u = 0; u < 4; u++) {
                const int i = i0 + u * 256;
                const double* e = A.partials + (int64_t)(i < total ? i : 0) * NACC;
#pragma unroll
                for (int k = 0; k < NACC; k++) v[u][k] = i < total ? e[k] : 0.0;
            }
#pragma unroll
            for (int k = 0; k < NACC; k++) acc[k][j] += (v[0][k] + v[1][k]) + (v[2][k] + v[3][k]);
        }
    }
#pragma unroll
    for (int k = 0; k < NACC; k++) {
        acc[k][0] += acc[k][2]; acc[k][1] += acc[k][3];             // the tree: o = 128,
        double x = acc[k][0] + acc[k][1];                            // 64,
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) x += __shfl_down(x, o, 64);   // 32 .. 1 (lane l < o reads lane l + o: sh[tid] += sh[tid + o])
        r[k] = x;                                                    // (lane 0's is the sum)
    }
}
// every lane of a wave that ran windows c_lo .. c_hi of group g calls this once, after its last store (dev_store: records, partial sums)
template <int NACC>
__device__ __forceinline__ void fused_finalize_wave(const IsoArgs& A, const ReduceArgs& R, int nstate, int g, int c_lo, int c_hi) {
    const int lane = threadIdx.x & 63;
    const int G = A.tv.n_groups;
    dev_stores_done();                                               // this wave's records and partial sums have left it
    double worst = 0.0;
    for (int b = c_lo; b < c_hi; b++) worst = fmax(worst, window_check_wave(A, nstate, g, b));      // boundaries between this wave's own windows
    for (int side = 0; side < 2; side++) {
        const int b = side == 0 ? c_lo - 1 : c_hi;                   // the boundary before the first window, after the last one
        if (b < 0 || b >= A.n_chunks - 1) continue;
        unsigned old = 0;
        if (lane == 0) old = __hip_atomic_fetch_add(&A.fuse_arrive[(int64_t)b * G + g], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        old = __builtin_amdgcn_readfirstlane(old);
        if (old != 1u) continue;                                     // the first to arrive: the neighbour will check
        worst = fmax(worst, window_check_wave(A, nstate, g, b));
        if (lane == 0) __hip_atomic_store(&A.fuse_arrive[(int64_t)b * G + g], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // ready for the next launch
    }
    unsigned done = 0;
    if (lane == 0) {
        if (worst > 0.0 || worst != worst)
            atomicMax((unsigned long long*)A.chk_out, (unsigned long long)__double_as_longlong(worst == worst ? worst : INFINITY));
        dev_stores_done();                                           // (the atomicMax has returned: it precedes the count)
        done = __hip_atomic_fetch_add(A.fuse_done, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    done = __builtin_amdgcn_readfirstlane(done);
    if (done != (unsigned)A.fuse_items - 1u) return;
    // the last wave of the launch.  One acquire (an invalidate of this XCD's L2 -- once per launch): the partial sums, written through
    // by the other waves, are then read with ordinary wide loads
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    double r[NACC];
    reduce_all_wave<NACC>(R, r);
    if (lane == 0) {
        for (int slot = 0; slot < R.n_out; slot++) {                 // (a slot nothing feeds -- a fixed parameter -- is zero, as reduce_slot leaves it)
            double v = slot == 0 ? r[0] : 0.0;
#pragma unroll
            for (int k = 1; k < NACC; k++)
                if (R.map[k - 1] == slot) v = r[k];
            for (int i = 0; i < 4; i++)
                if (R.add_slot[i] == slot) v += R.add[i];
            dev_store(&R.out[slot], v);
        }
        // the check word has a slot of its own (nobody zeroes out[] between launches): moved, and cleared for the next launch
        const unsigned long long w = __hip_atomic_exchange((unsigned long long*)A.chk_out, 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        dev_store(&R.out[R.n_out], __longlong_as_double((long long)w));
        __hip_atomic_store(A.fuse_done, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (R.pub) {
        dev_stores_done();
        for (int k = lane; k <= R.n_out; k += WAVE) R.pub[k] = dev_load(&R.out[k]);
        __threadfence_system();                                      // the wave's stores have reached the host before the flag does
        if (lane == 0) __hip_atomic_store(R.pub_flag, R.pub_seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}
#endif

}  // namespace ssde
#endif
