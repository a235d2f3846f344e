// k_tv_hess.hip -- exact second derivatives of the isotropic Kalman likelihoods with ROW-VARYING SDE parameters (gfx950): the
// counterpart of tmb_obj_joint$he(x) (MakeADHessObject2, src/init.c:13; R/sde.R:1363) and of the H_uu / H_u,theta blocks TMB's
// Laplace approximation takes from CppAD (random = "coeff_re", R/sde.R:510-525, 656-658) for the models the reference exists for:
// tau ~ s(covariate), nu ~ s(covariate), mu ~ s(covariate) in a CTCRW / OU_SSM / BM_SSM (nllk_ctcrw.hpp:143-156 feeding :195-247).
//
// Forward over forward, one wavefront lane per coefficient PAIR (a, b): the lane runs the PRIMAL recursion in hyper-dual
// arithmetic (ssde_hdual.hpp) with the row's linear predictors seeded by the two coefficients' design-matrix entries, and the
// mixed part of the accumulated likelihood is d^2 nllk / d coef_a d coef_b.  Nothing is differenced.  Work item of a wave:
// (track, time window, block of 64 pairs); long tracks are cut into windows with a warm-up and a hand-over check of every
// component of the hyper-dual state, exactly as the gradient lanes are (k_tv_filter.hpp) -- a handful of animals is a problem
// of latency, and a window's rows are what a wave's serial chain is made of.
//
//   hess_prepare_kernel   row-parallel: the linear predictors of every row (A2) -> an 8-double record [dt | p_0 .. p_3 | y_0 y_1]
//   hess_filter_kernel    the recursion; per lane and row ~40 hyper-dual products + the row's transition in the same arithmetic
//   hess_finish_kernel    hand-over checks (largest relative disagreement -> out[n_pairs]) and the fixed-order sums per pair
#include <algorithm>

#include "ssde_device.hpp"
#include "ssde_hdual.hpp"
#include "ssde_tv.hpp"
#include "ssde_dense.hpp"

namespace ssde {

namespace {

template <int MODEL, int D>
__global__ __launch_bounds__(256) void hess_prepare_kernel(const TvHessArgs A) {
    constexpr int Q = (MODEL == M_BM_SSM) ? D + 1 : D + 2;
    const SlotTable* __restrict__ T = A.slots;
    __shared__ double s_coef[MAX_COLS];
    __shared__ int s_col[MAX_COLS], s_pj[MAX_COLS];
    if ((int)threadIdx.x < A.n_slots) {
        const int k = threadIdx.x;
        s_col[k] = T->col[k]; s_pj[k] = T->par_j[k]; s_coef[k] = A.par[T->pidx[k]];
    }
    __syncthreads();
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < A.n; i += (int64_t)gridDim.x * 256) {
        double par[Q];
#pragma unroll
        for (int j = 0; j < Q; j++) par[j] = 0.0;
        for (int k = 0; k < A.n_slots; k++) {                        // par_vec = X_fe coeff_fe + X_re coeff_re (nllk_ctcrw.hpp:143-149)
            const double x = s_col[k] >= 0 ? A.colbuf[(int64_t)s_col[k] * A.col_stride + i] : 1.0;
            const double t = x * s_coef[k];
#pragma unroll
            for (int j = 0; j < Q; j++) par[j] += (s_pj[k] == j) ? t : 0.0;
        }
        double* r = A.rec + i * HESS_RS;
        r[0] = (i + 1 < A.n) ? A.times[i + 1] - A.times[i] : A.last_dt;   // the interval AFTER the row (nllk_ctcrw.hpp:126-129)
#pragma unroll
        for (int j = 0; j < 4; j++) r[1 + j] = j < Q ? par[j] : 0.0;
#pragma unroll
        for (int a = 0; a < 2; a++) r[5 + a] = a < D ? A.obs[i + (int64_t)a * A.n] : 0.0;
        r[7] = 0.0;
#pragma unroll
        for (int k = 0; k < 4; k++) r[8 + k] = (A.has_h && k < D * D) ? A.h_array[i * (D * D) + k] : 0.0;   // H_array[,,i]
    }
}

// ESEAL_SSM (nllk_e_seal_ssm.hpp:104-118, 136-137): record [dt | mu | log sigma | a1 | log a2 | y | . | . | R_i | h_i]
__global__ __launch_bounds__(256) void hess_prepare_eseal_kernel(const TvHessArgs A) {
    const SlotTable* __restrict__ T = A.slots;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < A.n; i += (int64_t)gridDim.x * 256) {
        double par[2] = {0.0, 0.0};
        for (int k = 0; k < A.n_slots; k++) {
            const int col = T->col[k], j = T->par_j[k];
            const double t = ((col >= 0) ? A.colbuf[(int64_t)col * A.col_stride + i] : 1.0) * A.par[T->pidx[k]];
            par[0] += (j == 0) ? t : 0.0; par[1] += (j == 1) ? t : 0.0;
        }
        double* r = A.rec + i * HESS_RS;
        r[0] = (i + 1 < A.n) ? A.times[i + 1] - A.times[i] : A.last_dt;
        r[1] = par[0]; r[2] = par[1]; r[3] = A.par[1]; r[4] = A.par[2];
        r[5] = A.obs[i]; r[6] = r[7] = 0.0;
        r[8] = A.eseal_R[i]; r[9] = A.eseal_h[i]; r[10] = r[11] = 0.0;
    }
}

// the lane's two directions as seeds of the row's predictors
struct PairSeed {
    int kind_a, dim_a, kind_b, dim_b;
};

template <int MODEL, int D>
struct HessLane;

template <int D>
struct HessLane<M_CTCRW, D> {
    static constexpr int NSTATE = 4 * (2 * D + 3);
    IsoCtcrwState<HD, D> S;
    __device__ __forceinline__ void init(const double* a0, const double* p0) {
        for (int a = 0; a < D; a++) { S.x[a] = HD(a0[2 * a]); S.v[a] = HD(a0[2 * a + 1]); }
        S.p11 = HD(p0[0]); S.p12 = HD(p0[1]); S.p22 = HD(p0[2]);
        S.nll = HD(0.0);
    }
    __device__ __forceinline__ void warm_init(const double* y, const double* p0) {
        double a0[2 * D];
        for (int a = 0; a < D; a++) { a0[2 * a] = (y[a] == y[a]) ? y[a] : 0.0; a0[2 * a + 1] = 0.0; }
        init(a0, p0);
    }
    __device__ __forceinline__ void step(const double* r, const HD& h, const PairSeed& sd, double wa, double wb, int any_nan) {
        HD p[D + 2];
#pragma unroll
        for (int j = 0; j < D + 2; j++) {
            const bool ma = (j < D) ? (sd.kind_a == TVK_MU && sd.dim_a == j) : (j == D ? sd.kind_a == TVK_P1 : sd.kind_a == TVK_P2);
            const bool mb = (j < D) ? (sd.kind_b == TVK_MU && sd.dim_b == j) : (j == D ? sd.kind_b == TVK_P1 : sd.kind_b == TVK_P2);
            p[j] = HD(r[1 + j], ma ? wa : 0.0, mb ? wb : 0.0, 0.0);
        }
        CtcrwTr<HD> tr;
        ctcrw_trans_g<HD>(r[0], p[D], p[D + 1], tr);
        iso_ctcrw_row<HD, D>(S, tr, h, p, r + 5, any_nan);
    }
    __device__ __forceinline__ void dump(double* o) const {
        int k = 0;
        auto put = [&](const HD& z) { o[k++] = z.v; o[k++] = z.a; o[k++] = z.b; o[k++] = z.ab; };
        for (int a = 0; a < D; a++) { put(S.x[a]); put(S.v[a]); }
        put(S.p11); put(S.p12); put(S.p22);
    }
};

template <int MODEL, int D>
struct HessLane {                                  // OU_SSM / BM_SSM
    static constexpr int NSTATE = 4 * (D + 1);
    static constexpr int Q = (MODEL == M_BM_SSM) ? D + 1 : D + 2;
    IsoScalState<HD, D> S;
    __device__ __forceinline__ void init(const double* a0, const double* p0) {
        for (int a = 0; a < D; a++) S.x[a] = HD(a0[a]);
        S.p = HD(p0[0]);
        S.nll = HD(0.0);
    }
    __device__ __forceinline__ void warm_init(const double* y, const double* p0) {
        double a0[D];
        for (int a = 0; a < D; a++) a0[a] = (y[a] == y[a]) ? y[a] : 0.0;
        init(a0, p0);
    }
    __device__ __forceinline__ void step(const double* r, const HD& h, const PairSeed& sd, double wa, double wb, int any_nan) {
        HD p[Q];
#pragma unroll
        for (int j = 0; j < Q; j++) {
            const bool ma = (j < D) ? (sd.kind_a == TVK_MU && sd.dim_a == j) : (j == D ? sd.kind_a == TVK_P1 : sd.kind_a == TVK_P2);
            const bool mb = (j < D) ? (sd.kind_b == TVK_MU && sd.dim_b == j) : (j == D ? sd.kind_b == TVK_P1 : sd.kind_b == TVK_P2);
            p[j] = HD(r[1 + j], ma ? wa : 0.0, mb ? wb : 0.0, 0.0);
        }
        ScalTr<HD> tr;
        if (MODEL == M_OU_SSM) ou_trans_g<HD>(r[0], p[D], p[Q - 1], tr);
        else bm_trans_g<HD>(r[0], p[D], tr);
        iso_scal_row<HD, D>(S, tr, h, p, r + 5, any_nan);
    }
    __device__ __forceinline__ void dump(double* o) const {
        int k = 0;
        auto put = [&](const HD& z) { o[k++] = z.v; o[k++] = z.a; o[k++] = z.b; o[k++] = z.ab; };
        for (int a = 0; a < D; a++) put(S.x[a]);
        put(S.p);
    }
};

// Full-covariance lanes (per-row H_array, nllk_ctcrw.hpp:203-205, and / or a P0 that is not block-identical, R/sde.R:552-557,
// 582-587): the general step of ssde_dense.hpp -- the text the first-order lanes run in DualN -- in hyper-dual arithmetic.  sd + sd^2
// hyper-dual numbers of state (80 doubles for CTCRW with two response columns): the kernel spills; a Hessian is asked for once
// per outer iteration of a fit.
template <int MODEL, int D>
struct HessLaneDense {
    static constexpr int SD = DenseDims<MODEL, D>::SD;
    static constexpr int NSTATE = 4 * (SD + SD * SD);
    static constexpr int Q = DenseDims<MODEL, D>::Q;
    struct State { HD a[SD]; HD P[SD][SD]; HD nll; } S;
    bool has_h;
    __device__ __forceinline__ void init(const double* a0, const double* p0f) {
        for (int i = 0; i < SD; i++) {
            S.a[i] = HD(a0[i]);
            for (int j = 0; j < SD; j++) S.P[i][j] = HD(p0f[i + j * SD]);
        }
        S.nll = HD(0.0);
    }
    __device__ __forceinline__ void warm_init(const double* y, const double* p0f) {
        double a0[SD];
        for (int c = 0; c < SD; c++) a0[c] = 0.0;
        for (int a = 0; a < D; a++) a0[DenseDims<MODEL, D>::z(a)] = (y[a] == y[a]) ? y[a] : 0.0;
        init(a0, p0f);
    }
    __device__ __forceinline__ void step(const double* r, const HD& h, const PairSeed& sd, double wa, double wb, int any_nan) {
        HD p[Q];
        for (int j = 0; j < Q; j++) {
            const bool ma = (j < D) ? (sd.kind_a == TVK_MU && sd.dim_a == j) : (j == D ? sd.kind_a == TVK_P1 : sd.kind_a == TVK_P2);
            const bool mb = (j < D) ? (sd.kind_b == TVK_MU && sd.dim_b == j) : (j == D ? sd.kind_b == TVK_P1 : sd.kind_b == TVK_P2);
            p[j] = HD(r[1 + j], ma ? wa : 0.0, mb ? wb : 0.0, 0.0);
        }
        HD H[D][D];
        for (int i = 0; i < D; i++)
            for (int j = 0; j < D; j++) H[i][j] = has_h ? HD(r[8 + i + j * D]) : (i == j ? h : HD(0.0));       // H_array[,,i] holds no parameter
        double y[D];
        for (int a = 0; a < D; a++) y[a] = r[5 + a];
        dense_step_g<MODEL, D, HD>(S, p, H, r[0], y, is_na(y[0], any_nan));
    }
    __device__ __forceinline__ void dump(double* o) const {
        int k = 0;
        auto put = [&](const HD& z) { o[k++] = z.v; o[k++] = z.a; o[k++] = z.b; o[k++] = z.ab; };
        for (int i = 0; i < SD; i++) put(S.a[i]);
        for (int i = 0; i < SD; i++)
            for (int j = 0; j < SD; j++) put(S.P[i][j]);
    }
};
// ESEAL_SSM: the scalar lipid-mass filter of ssde_tv.hpp (TvEsealOps: the first state component is the constant 1) in hyper-dual
// arithmetic.  y_i = a1 + z_i L + N(0, H_i), z_i = a2 / R_i, H_i = tau^2 / h_i; L' = L + mu_i dt_i + N(0, sigma_i^2 dt_i)
// (nllk_e_seal_ssm.hpp:139-207).  Directions: log tau (the kernel's h = tau^2 carries its seeds), a1, log a2, coefficients of mu / log sigma.
struct HessLaneEseal {
    static constexpr int NSTATE = 8;
    struct State { HD x, p, nll; } S;
    bool has_h;
    __device__ __forceinline__ void init(const double* a0 /* (1, L0) */, const double* p0f /* 2 x 2 */) {
        S.x = HD(a0[1]); S.p = HD(p0f[3]); S.nll = HD(0.0);
    }
    __device__ __forceinline__ void warm_init(const double*, const double* p0f) { S.x = HD(0.0); S.p = HD(p0f[3]); S.nll = HD(0.0); }
    __device__ __forceinline__ void step(const double* r, const HD& h, const PairSeed& sd, double wa, double wb, int any_nan) {
        const HD mu(r[1], sd.kind_a == TVK_MU ? wa : 0.0, sd.kind_b == TVK_MU ? wb : 0.0, 0.0);
        const HD ls(r[2], sd.kind_a == TVK_P1 ? wa : 0.0, sd.kind_b == TVK_P1 ? wb : 0.0, 0.0);
        const HD a1(r[3], sd.kind_a == TVK_A1 ? 1.0 : 0.0, sd.kind_b == TVK_A1 ? 1.0 : 0.0, 0.0);
        const HD la2(r[4], sd.kind_a == TVK_A2 ? 1.0 : 0.0, sd.kind_b == TVK_A2 ? 1.0 : 0.0, 0.0);
        const double dt = r[0];
        const HD z = dexp(la2) * (1.0 / r[8]);                         // makeZ :43-48
        const HD H = h * (1.0 / r[9]);                                 // makeH :55-59
        const HD drift = mu * dt;                                      // makeT :16-23
        const HD q = dexp(2.0 * ls) * dt;                              // makeQ :30-35
        const bool na = is_na(r[5], any_nan);                          // :175
        const HD F = z * z * S.p + H;                                  // :184
        if (!na && !(F.v <= 0.0)) {                                    // :188 (a NaN takes the update branch)
            const HD u = r[5] - a1 - z * S.x;                          // :182
            S.nll = S.nll + (dlog(F) + u * u / F) * 0.5;
            const HD k = S.p * z / F;                                  // :197
            S.x = S.x + drift + k * u;                                 // :199
            S.p = S.p * (H / F) + q;                                   // :201-202: p - k z p = p H / F
        } else {
            S.x = S.x + drift;                                         // :176
            S.p = S.p + q;                                             // :177
        }
    }
    __device__ __forceinline__ void dump(double* o) const {
        o[0] = S.x.v; o[1] = S.x.a; o[2] = S.x.b; o[3] = S.x.ab; o[4] = S.p.v; o[5] = S.p.a; o[6] = S.p.b; o[7] = S.p.ab;
    }
};

template <int MODEL, int D, bool DENSE>
struct HessLaneSel { typedef HessLane<MODEL, D> type; };
template <>
struct HessLaneSel<M_ESEAL, 1, true> { typedef HessLaneEseal type; };
template <int MODEL, int D>
struct HessLaneSel<MODEL, D, true> { typedef HessLaneDense<MODEL, D> type; };

constexpr int HESS_U = 2;          // rows per prefetch block of the lane's two weights

// One wave = one (track, window, block of 64 pairs).  The row record is the same for every lane (one track per wave): it is
// read through a wave-uniform pointer -- scalar loads, scalar registers -- and only the two design-matrix entries of the
// lane's pair are vector loads, prefetched one block of rows ahead.  The row loop is NOT unrolled beyond that block: a row is
// ~10^3 instructions of hyper-dual arithmetic, and the registers are better spent on the state than on a second copy of it.
template <int MODEL, int D, bool DENSE>
__global__ __launch_bounds__(WG_WAVES * WAVE, 1) void hess_filter_kernel(const TvHessArgs A) {
    typedef typename HessLaneSel<MODEL, D, DENSE>::type Lane;
    constexpr int SD = (MODEL == M_CTCRW || MODEL == M_ESEAL) ? 2 * D : D;
    const int item = blockIdx.x * WG_WAVES + __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));      // one work item per WAVE, no barriers
    if (item >= A.n_items) return;
    const int lane = threadIdx.x & 63;
    const TvItem it = A.items[item];
    const int64_t trk = it.pack;
    const int64_t row0 = A.trk_row0[trk];
    const int ns = A.trk_ns[trk];
    int s_begin, s_acc, s_end;
    window_bounds(ns, it.nc, A.window, 0, it.c, s_begin, s_acc, s_end);
    const int pr = it.b * WAVE + lane;
    const bool live = pr < A.n_pairs;
    const int ka = A.pair_a[live ? pr : 0], kb = A.pair_b[live ? pr : 0];
    const TvDir da = A.dirs[ka], db = A.dirs[kb];
    PairSeed sd;
    sd.kind_a = da.kind; sd.dim_a = da.dim; sd.kind_b = db.kind; sd.dim_b = db.dim;
    // sigma_obs^2 = exp(2 log_sigma_obs) (nllk_ctcrw.hpp:136, 167): d / d log_sigma_obs = 2 h, second derivative 4 h
    const double h0 = exp(2.0 * A.par[0]);
    const double sa = (da.kind == TVK_SIG) ? 2.0 : 0.0, sb = (db.kind == TVK_SIG) ? 2.0 : 0.0;
    const HD h(h0, sa * h0, sb * h0, sa * sb * h0);
    const int64_t imax = A.n - 1;
    double p0[DENSE ? 16 : 3];                                  // (the full-covariance lanes: P0 as given, sd x sd)
#pragma unroll
    for (int q = 0; q < (DENSE ? 16 : 3); q++) p0[q] = DENSE ? A.p0_full[q] : A.p0[q < 3 ? q : 0];
    const double* wpa = A.wdir + ka;
    const double* wpb = A.wdir + kb;
    auto row_of = [&](int s) { const int64_t i = row0 + 1 + s; return i < imax ? i : imax; };     // (look-ahead rows stay inside the buffers)

    Lane S;
    if constexpr (DENSE) S.has_h = A.has_h != 0;
    if (s_begin == 0) {
        double a0[SD];
#pragma unroll
        for (int c = 0; c < SD; c++) a0[c] = A.a0[trk * SD + c];
        S.init(a0, p0);
    } else {
        S.warm_init(A.rec + row_of(s_begin) * HESS_RS + 5, p0);
    }
    double wa[HESS_U], wb[HESS_U];
#pragma unroll
    for (int u = 0; u < HESS_U; u++) { const int64_t i = row_of(s_begin + u); wa[u] = wpa[i * A.ndp]; wb[u] = wpb[i * A.ndp]; }
#pragma unroll 1
    for (int s0 = s_begin; s0 < s_end; s0 += HESS_U) {
        double na[HESS_U], nb[HESS_U];
#pragma unroll
        for (int u = 0; u < HESS_U; u++) { const int64_t i = row_of(s0 + HESS_U + u); na[u] = wpa[i * A.ndp]; nb[u] = wpb[i * A.ndp]; }
#pragma unroll
        for (int u = 0; u < HESS_U; u++) {
            const int s = s0 + u;
            if (s >= s_end) break;
            if (s == s_acc && s_acc > s_begin) {
                double st[Lane::NSTATE];
                S.dump(st);
                double* o = A.bnd + ((int64_t)item * 2 + 0) * HESS_NSTATE * WAVE + lane;
#pragma unroll
                for (int q = 0; q < Lane::NSTATE; q++) o[q * WAVE] = st[q];
                S.S.nll = HD(0.0);
            }
            S.step(A.rec + row_of(s) * HESS_RS, h, sd, wa[u], wb[u], A.any_nan);
        }
#pragma unroll
        for (int u = 0; u < HESS_U; u++) { wa[u] = na[u]; wb[u] = nb[u]; }
    }
    if (it.c + 1 < it.nc) {
        double st[Lane::NSTATE];
        S.dump(st);
        double* o = A.bnd + ((int64_t)item * 2 + 1) * HESS_NSTATE * WAVE + lane;
#pragma unroll
        for (int q = 0; q < Lane::NSTATE; q++) o[q * WAVE] = st[q];
    }
    const bool empty = s_acc >= s_end || !live;
    double* o = A.part + (int64_t)item * 4 * WAVE + lane;
    o[0] = empty ? 0.0 : S.S.nll.ab;
    o[WAVE] = empty ? 0.0 : S.S.nll.a;
    o[2 * WAVE] = empty ? 0.0 : S.S.nll.b;
    o[3 * WAVE] = empty ? 0.0 : S.S.nll.v;
}

// blocks [0, n_items): hand-over check of one item against the next window's warm-up dump; blocks [n_items, n_items + n_pb): the sums
// of one block of 64 pairs over every (track, window), in item order (fixed: bitwise reproducible)
__global__ __launch_bounds__(WAVE) void hess_finish_kernel(const TvHessArgs A, int nstate) {
    const int lane = threadIdx.x;
    if ((int)blockIdx.x < A.n_items) {
        const int item = blockIdx.x;
        const TvItem it = A.items[item];
        if (it.c + 1 >= it.nc) return;
        const int ns = A.trk_ns[it.pack];
        int sb_, s_next, se_;
        window_bounds(ns, it.nc, A.window, 0, it.c + 1, sb_, s_next, se_);
        if (!(s_next < ns)) return;
        const double* out_c = A.bnd + ((int64_t)item * 2 + 1) * HESS_NSTATE * WAVE + lane;
        const double* in_n = A.bnd + ((int64_t)(item + 1) * 2 + 0) * HESS_NSTATE * WAVE + lane;     // (items are ordered track, pair block, window)
        const bool live = it.b * WAVE + lane < A.n_pairs;
        double worst = 0.0;
        for (int k = 0; k < nstate; k++) {
            const double a = live ? out_c[k * WAVE] : 0.0, b = live ? in_n[k * WAVE] : 0.0;
            double err = fabs(a - b), sc = fmax(fabs(a), fabs(b));
            if (live && !(err == err)) err = INFINITY;
            // scale of a component: its largest magnitude over the lanes' (every pair's) copies of it, and over its group of four parts
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) { err = fmax(err, __shfl_xor(err, o, 64)); sc = fmax(sc, __shfl_xor(sc, o, 64)); }
            if (err > 0.0) worst = fmax(worst, err / sc);
        }
        if (lane == 0 && worst > 0.0)
            atomicMax((unsigned long long*)(A.out + 4 * A.n_pb * WAVE), (unsigned long long)__double_as_longlong(worst == worst ? worst : INFINITY));
        return;
    }
    const int pb = blockIdx.x - A.n_items;
    double acc[4] = {0.0, 0.0, 0.0, 0.0};
    for (int i = 0; i < A.n_items; i++) {
        if (A.items[i].b != pb) continue;
        const double* p = A.part + (int64_t)i * 4 * WAVE + lane;
#pragma unroll
        for (int q = 0; q < 4; q++) acc[q] += p[q * WAVE];
    }
#pragma unroll
    for (int q = 0; q < 4; q++) A.out[((int64_t)q * A.n_pb + pb) * WAVE + lane] = acc[q];
}

}  // namespace

hipError_t launch_tv_hess(const TvHessArgs& a, hipStream_t s) {
    if (a.n_items == 0) return hipSuccess;
    const hipError_t e0 = hipMemsetAsync(a.out + 4 * a.n_pb * WAVE, 0, 8, s);
    if (e0 != hipSuccess) return e0;
    const unsigned pblocks = (unsigned)std::min<int64_t>((a.n + 255) / 256, 4096);
    dim3 grid((a.n_items + WG_WAVES - 1) / WG_WAVES), block(WG_WAVES * WAVE);
    int nstate = 0;
#define SSDE_HESS_ONE(MODEL, D)                                                                              \
    if (a.model == MODEL && a.d == D) {                                                                      \
        hipLaunchKernelGGL((hess_prepare_kernel<MODEL, D>), dim3(pblocks), dim3(256), 0, s, a);              \
        if (a.dense) {                                                                                       \
            hipLaunchKernelGGL((hess_filter_kernel<MODEL, D, true>), grid, block, 0, s, a);                  \
            nstate = HessLaneDense<MODEL, D>::NSTATE;                                                        \
        } else {                                                                                             \
            hipLaunchKernelGGL((hess_filter_kernel<MODEL, D, false>), grid, block, 0, s, a);                 \
            nstate = HessLane<MODEL, D>::NSTATE;                                                             \
        }                                                                                                    \
    }
    SSDE_HESS_ONE(M_CTCRW, 1) SSDE_HESS_ONE(M_CTCRW, 2) SSDE_HESS_ONE(M_OU_SSM, 1) SSDE_HESS_ONE(M_OU_SSM, 2)
    SSDE_HESS_ONE(M_BM_SSM, 1) SSDE_HESS_ONE(M_BM_SSM, 2)
#undef SSDE_HESS_ONE
    if (a.model == M_ESEAL && a.d == 1) {
        hipLaunchKernelGGL(hess_prepare_eseal_kernel, dim3(pblocks), dim3(256), 0, s, a);
        hipLaunchKernelGGL((hess_filter_kernel<M_ESEAL, 1, true>), grid, block, 0, s, a);
        nstate = HessLaneEseal::NSTATE;
    }
    if (nstate == 0) return hipErrorInvalidValue;
    hipLaunchKernelGGL(hess_finish_kernel, dim3(a.n_items + a.n_pb), dim3(WAVE), 0, s, a, nstate);
    return hipGetLastError();
}

}  // namespace ssde
