// k_iso_drift.hip -- shared-covariance Kalman lanes with a ROW-VARYING DRIFT for gfx950 (CTCRW, OU_SSM, BM_SSM).
//
// The model the reference exists for, in its state-space form: mu smooth in covariates, every other parameter constant --
//     par_mat.row(i) = X_fe coeff_fe + X_re coeff_re          (nllk_ctcrw.hpp:143-149, nllk_ou_ssm.hpp:113-119)
//     a <- T a + K u + B mu_i                                  (nllk_ctcrw.hpp:211-212, 238; nllk_ou_ssm.hpp:174-204)
// with design columns only in the rows of mu_1 .. mu_d.  tau, nu / kappa / sigma and sigma_obs are constants, the grid is
// regular and no row is missing: the covariance half of the filter (P, F, K and their sensitivities) is then as
// data-independent as with a constant drift, so the engine's gain table (ssde_engine.hip: build_gain_table) serves it
// unchanged, and the lanes run the mean half -- ssde_math.hpp's ctcrw_mean_step / scal_mean_step with THIS ROW's mu --
// plus one linear recursion per streamed column for d nllk / d coefficient:
//     mu_a(i) = mu_a0 + sum_k coef_k X_k(i)                    (columns k that feed dimension a)
//     d x / d coef_k:  mx_k <- (1 - k1) mx_k + t12 mv_k + b1 X_k(i),   mv_k <- e mv_k - k2 mx_k + b2 X_k(i)      (CTCRW)
//                      mx_k <- (t - k) mx_k + b X_k(i)                                                          (OU / BM)
//     d nllk / d coef_k = - sum_i F_i^-1 u_a(i) mx_k(i)
// (the recursion of mx_k is the same whichever dimension the column feeds -- the gains are isotropic -- only the
// innovation it is paired with differs).  5 fp64 FMAs per column and row against the 8 bytes the column costs to stream:
// the kernel is bound by HBM, 8 (d + K) bytes per row.
//
// Layout: the tiles of ssde_device.hpp with the design columns as further channels; lane = track, one wave per
// (64-track group, time window); observations and columns prefetched one block of rows ahead in two ping-pong register
// blocks; the covariance transient's gains staged through LDS exactly as in k_iso_shared.hip; hand-over dumps hold the
// state, the covariance directions, the intercept direction and the column sensitivities (IsoArgs.bnd_stride wide).
#include <hip/hip_ext.h>

#include <type_traits>

#include "ssde_device.hpp"

namespace ssde {

constexpr int DRIFT_SLAB_ROWS = 64;
constexpr int DRIFT_MASK = DIR_SIG | DIR_MU | DIR_P1 | DIR_P2;

template <int MODEL>
struct DriftModel;
template <>
struct DriftModel<M_CTCRW> {
    static constexpr bool CT = true, HAS_P2 = true;
    template <int D> using Mean = CtcrwMean<D, DRIFT_MASK>;
    typedef CtcrwGain Gain;
    typedef CtcrwTrans Trans;
};
template <>
struct DriftModel<M_OU_SSM> {
    static constexpr bool CT = false, HAS_P2 = true;
    template <int D> using Mean = ScalMean<D, DRIFT_MASK>;
    typedef ScalGain Gain;
    typedef ScalTrans Trans;
};
template <>
struct DriftModel<M_BM_SSM> {
    static constexpr bool CT = false, HAS_P2 = false;
    template <int D> using Mean = ScalMean<D, DRIFT_MASK>;
    typedef ScalGain Gain;
    typedef ScalTrans Trans;
};

// components of a hand-over dump: state, covariance directions, intercept direction, column sensitivities
__host__ __device__ constexpr inline int drift_nstate_c(bool ct, bool has_p2, int d, int kp) {
    return (ct ? 2 * d : d) * (2 + 2 + (has_p2 ? 1 : 0)) + kp * (ct ? 2 : 1);
}
int drift_nstate(int model, int d, int k) {
    const int kp = (k + 3) / 4 * 4;
    return drift_nstate_c(model == M_CTCRW, model != M_BM_SSM, d, kp);
}

template <int MODEL, int D, int KP>
struct DriftLane {
    typedef DriftModel<MODEL> DM;
    static constexpr bool CT = DM::CT, HAS_P2 = DM::HAS_P2;
    static constexpr int SD = CT ? 2 * D : D;
    static constexpr int NSTATE = drift_nstate_c(CT, HAS_P2, D, KP);
    typename DM::template Mean<D> M;
    double cx[KP], cv[CT ? KP : 1], gk[KP];

    __device__ __forceinline__ void init(const double* a0) {
        M.init(a0);
#pragma unroll
        for (int k = 0; k < KP; k++) { cx[k] = 0.0; gk[k] = 0.0; if (CT) cv[k] = 0.0; }
    }
    __device__ __forceinline__ void reset_acc() {
        M.reset_acc();
#pragma unroll
        for (int k = 0; k < KP; k++) gk[k] = 0.0;
    }
    // one row: y[D] observations, X[KP] design columns
    __device__ __forceinline__ void step(const IsoArgs& A, const typename DM::Trans& tr, const typename DM::Gain& G, const double* y,
                                         const double* X) {
        const bool scored = G.iF != 0.0;
        double mu[D], u[D];
#pragma unroll
        for (int a = 0; a < D; a++) { mu[a] = A.mu[a]; u[a] = scored ? y[a] - M.x[a] : 0.0; }
#pragma unroll
        for (int k = 0; k < KP; k++) {
            mu[0] = fma(A.coefA[k], X[k], mu[0]);
            if (D > 1) mu[D - 1] = fma(A.coefB[k], X[k], mu[D - 1]);
        }
        // column sensitivities (before the state moves: they pair with THIS row's innovation)
#pragma unroll
        for (int k = 0; k < KP; k++) {
            const double uk = (D > 1 && ((A.drift_dim1 >> k) & 1u)) ? u[D - 1] : u[0];
            const double mx = cx[k];
            gk[k] = fma(-G.iF * mx, uk, gk[k]);
            if constexpr (CT) {
                const double mv = cv[k];
                const double du = scored ? -mx : 0.0;
                cx[k] = fma(G.bm * tr.b1, X[k], fma(G.k1, du, fma(tr.t12, mv, mx)));
                cv[k] = fma(G.bm * tr.b2, X[k], fma(G.k2, du, tr.e * mv));
            } else {
                cx[k] = fma(tr.b, X[k], G.c * mx);
            }
        }
        if constexpr (CT) ctcrw_mean_step<D, DRIFT_MASK>(M, tr, G, mu, y, scored);
        else scal_mean_step<D, DRIFT_MASK, HAS_P2>(M, tr, G, mu, y, scored);
    }
    __device__ __forceinline__ void dump(double* o) const {
        int n = 0;
#pragma unroll
        for (int a = 0; a < D; a++) { o[n++] = M.x[a]; if constexpr (CT) o[n++] = M.v[a]; }
#pragma unroll
        for (int j = 0; j < NDIRP; j++) {
            if (j == 2 && !HAS_P2) continue;
#pragma unroll
            for (int a = 0; a < D; a++) { o[n++] = M.tx[j][a]; if constexpr (CT) o[n++] = M.tv[j][a]; }
        }
#pragma unroll
        for (int a = 0; a < D; a++) { o[n++] = M.mx[a]; if constexpr (CT) o[n++] = M.mv[a]; }
#pragma unroll
        for (int k = 0; k < KP; k++) { o[n++] = cx[k]; if constexpr (CT) o[n++] = cv[k]; }
    }
};

template <class Gain>
__device__ __forceinline__ Gain gain_from_row(const double* r);
template <>
__device__ __forceinline__ CtcrwGain gain_from_row<CtcrwGain>(const double* r) {
    CtcrwGain G;
    G.iF = r[0]; G.k1 = r[1]; G.k2 = r[2]; G.bm = r[3];
#pragma unroll
    for (int j = 0; j < NDIRP; j++) { G.diF[j] = r[4 + j]; G.dk1[j] = r[7 + j]; G.dk2[j] = r[10 + j]; }
    return G;
}
template <>
__device__ __forceinline__ ScalGain gain_from_row<ScalGain>(const double* r) {
    ScalGain G;
    G.iF = r[0]; G.k = r[1]; G.c = r[2];
#pragma unroll
    for (int j = 0; j < NDIRP; j++) { G.diF[j] = r[4 + j]; G.dk[j] = r[7 + j]; }
    return G;
}

// NG = column slots / 4.  Slots beyond the batch's drift_k columns re-read column 0 (a cache hit, no HBM traffic) and feed
// sensitivities nobody reads: the address is selected, never branched on (a guarded load would make every later load wait).
template <int MODEL, int D, int NG>
__device__ __forceinline__ void run_lane_drift(const IsoArgs& A, int g, int chunk) {
    constexpr int KP = 4 * NG;
    constexpr int U = (NG <= 3) ? 4 : 2;          // rows per prefetch block (divides SHARED_U and WIN_ALIGN)
    static_assert(3 * U <= TILE_SPARE && SHARED_U % U == 0, "prefetch block");
    typedef DriftLane<MODEL, D, KP> Lane;
    typedef DriftModel<MODEL> DM;
    constexpr int SD = Lane::SD;
    const int lane = threadIdx.x & 63;
    const TileView& tv = A.tv;
    const int C = tv.C, c_obs = tv.c_obs, K = A.drift_k, c_col = A.c_col;
    const int nacc = 4 + D + K;
    const double* base = tv.tiles + tv.group_off[g] + lane;
    const int L = tv.group_len[g];
    const int ns = tv.lane_nsteps[g * WAVE + lane];
    int s_begin, s_acc, s_end;
    window_bounds(L, A.n_chunks, A.window, A.t0, chunk, s_begin, s_acc, s_end, A.t0_delta);
    int s_stat = (A.gain_last + SHARED_U - 1) / SHARED_U * SHARED_U;
    if (A.gain_stat[0] == 0.0) s_stat = INT32_MAX;          // degenerate ("never scored"): stay on the table

    __shared__ double gain_slab[WG_WAVES][DRIFT_SLAB_ROWS * GAIN_ROW];
    double* slab = gain_slab[threadIdx.x >> 6];
    int slab_row0 = -1;

    int chan[KP];                                             // tile channel of every column slot
#pragma unroll
    for (int k = 0; k < KP; k++) chan[k] = c_col + (k < K ? k : 0);

    typename DM::Trans tr;
    if constexpr (DM::CT) tr = A.ctr; else tr = A.str;
    const typename DM::Gain Gs = gain_from_row<typename DM::Gain>(A.gain_stat);

    Lane S;
    {
        double a0[SD];
        if (s_begin == 0) {
#pragma unroll
            for (int c = 0; c < SD; c++) a0[c] = tv.a0[((int64_t)g * SD + c) * WAVE + lane];
        } else {
#pragma unroll
            for (int a = 0; a < D; a++) {
                const double y0 = base[((int64_t)s_begin * C + c_obs + a) * WAVE];
                if constexpr (DM::CT) { a0[2 * a] = (y0 == y0) ? y0 : 0.0; a0[2 * a + 1] = 0.0; }
                else a0[a] = (y0 == y0) ? y0 : 0.0;
            }
        }
        S.init(a0);
    }

    double bufA[U][D + KP], bufB[U][D + KP];
    auto load = [&](double (&dst)[U][D + KP], int s0) {
        const double* p = base + (int64_t)s0 * C * WAVE;
#pragma unroll
        for (int u = 0; u < U; u++) {
#pragma unroll
            for (int a = 0; a < D; a++) dst[u][a] = __builtin_nontemporal_load(&p[(u * C + c_obs + a) * WAVE]);
#pragma unroll
            for (int k = 0; k < KP; k++) dst[u][D + k] = __builtin_nontemporal_load(&p[(u * C + chan[k]) * WAVE]);
        }
    };
    auto block = [&](const double (&blk)[U][D + KP], int s0) {
        if (s0 == s_acc && s_acc > s_begin) {
            // end of the warm-up: publish the state for the hand-over check, start scoring from zero
            double st[Lane::NSTATE];
            S.dump(st);
            double* o = A.bnd + (((int64_t)chunk * tv.n_groups + g) * 2 + 0) * A.bnd_stride * WAVE + lane;
#pragma unroll
            for (int k = 0; k < Lane::NSTATE; k++) o[k * WAVE] = st[k];
            S.reset_acc();
        }
        if (s0 < s_stat) {
            // covariance transient: this row's gains from the table, staged DRIFT_SLAB_ROWS rows at a time into LDS
            if (slab_row0 < 0 || s0 >= slab_row0 + DRIFT_SLAB_ROWS) {
                slab_row0 = s0;
                const int glast = A.gain_last;
                const double* __restrict__ gain = A.gain;
#pragma unroll 4
                for (int r = 0; r < DRIFT_SLAB_ROWS; r += WAVE / GAIN_ROW) {
                    const int rr = r + lane / GAIN_ROW;
                    slab[rr * GAIN_ROW + (lane % GAIN_ROW)] = gain[(int64_t)min(s0 + rr, glast) * GAIN_ROW + (lane % GAIN_ROW)];
                }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            }
#pragma unroll
            for (int u = 0; u < U; u++)
                if (s0 + u < ns) {
                    const typename DM::Gain G = gain_from_row<typename DM::Gain>(slab + (s0 + u - slab_row0) * GAIN_ROW);
                    S.step(A, tr, G, &blk[u][0], &blk[u][D]);
                }
        } else {
#pragma unroll
            for (int u = 0; u < U; u++)
                if (s0 + u < ns) S.step(A, tr, Gs, &blk[u][0], &blk[u][D]);
        }
    };
    load(bufA, s_begin);
    for (int s0 = s_begin; s0 < s_end; s0 += 2 * U) {
        load(bufB, s0 + U);                                  // TILE_SPARE keeps the look-ahead inside the allocation
        block(bufA, s0);
        load(bufA, s0 + 2 * U);
        if (s0 + U < s_end) block(bufB, s0 + U);
    }
    if (A.n_chunks > 1 && chunk + 1 < A.n_chunks) {
        double st[Lane::NSTATE];
        S.dump(st);
        double* o = A.bnd + (((int64_t)chunk * tv.n_groups + g) * 2 + 1) * A.bnd_stride * WAVE + lane;
#pragma unroll
        for (int k = 0; k < Lane::NSTATE; k++) o[k * WAVE] = st[k];
    }
    // accumulators: [value, sigma_obs, mu intercepts.., par d, par d+1, columns..]; the data-independent log-determinant
    // terms are added by the finalize launch
    double out[4 + D];
    {
        const double z[NDIRP] = {0.0, 0.0, 0.0};
        if constexpr (DM::CT) ctcrw_finish_parts<D, DRIFT_MASK>(0.0, z, S.M, out);
        else scal_finish_parts<D, DRIFT_MASK>(0.0, z, S.M, out);
    }
    const bool empty = s_acc >= s_end;
#pragma unroll
    for (int k = 0; k < 4 + D; k++) {
        const double t = wave_sum(empty ? 0.0 : out[k]);
        if (lane == 0) A.partials[((int64_t)chunk * nacc + k) * tv.n_groups + g] = t;
    }
#pragma unroll
    for (int k = 0; k < KP; k++) {
        const double t = wave_sum(empty ? 0.0 : S.gk[k]);
        if (lane == 0 && k < K) A.partials[((int64_t)chunk * nacc + 4 + D + k) * tv.n_groups + g] = t;
    }
}

template <int MODEL, int D, int NG>
__global__ __launch_bounds__(WG_WAVES * WAVE, 1) void iso_drift_kernel(const IsoArgs A) {
    if (blockIdx.x == 0 && threadIdx.x == 0 && A.chk_out) *A.chk_out = 0.0;   // raised by the finalize launch
    int g, part, chunk;
    if (!decode_block(A, A.n_chunks, g, part, chunk)) return;
    run_lane_drift<MODEL, D, NG>(A, g, chunk);
}

// =====================================================================================================================
// The same drift model where the covariance half is NOT shared: missing rows and / or an irregular time grid.  Lane = track
// with the lane's own covariance and covariance sensitivities -- the fused general step of ssde_math.hpp (ctcrw_step /
// scal_cov_step + scal_mean_step, what k_iso.hip runs for constant coefficients) fed THIS ROW's mu -- plus the column
// recursions with the lane's own gains: k1 = (p11 + t12 p12) / F, k2 = e p12 / F (CTCRW), c = t h / F (OU / BM), all zero /
// one on a row that is not scored (nllk_ctcrw.hpp:214-217, 226-228).  Bound by fp64 issue like the general kernel it extends
// (~180 + 8 K instructions per row), one wave per SIMD; still an order of magnitude above the lane = direction path at
// batch scale, which is where a batch with ANY missing row used to land.
template <int MODEL, int D, int KP>
struct DriftGenLane {
    typedef DriftModel<MODEL> DM;
    static constexpr bool CT = DM::CT, HAS_P2 = DM::HAS_P2;
    static constexpr int SD = CT ? 2 * D : D;
    typedef typename std::conditional<CT, CtcrwLane<D, DRIFT_MASK>, ScalLane<D, DRIFT_MASK>>::type Lane;
    static constexpr int NBASE = Lane::NSTATE;
    static constexpr int NSTATE = NBASE + KP * (CT ? 2 : 1);
    Lane L;
    double cx[KP], cv[CT ? KP : 1], gk[KP];

    __device__ __forceinline__ void init(const double* a0, const IsoArgs& A) {
        if constexpr (CT) L.init(a0, A.p0[0], A.p0[1], A.p0[2]); else L.init(a0, A.p0[0]);
#pragma unroll
        for (int k = 0; k < KP; k++) { cx[k] = 0.0; gk[k] = 0.0; if (CT) cv[k] = 0.0; }
    }
    __device__ __forceinline__ void reset_acc() {
        L.reset_acc();
#pragma unroll
        for (int k = 0; k < KP; k++) gk[k] = 0.0;
    }
    __device__ __forceinline__ void step(const IsoArgs& A, const typename DM::Trans& tr, const double* y, const double* X) {
        const bool na = is_na(y[0], A.any_nan);
        const double h = A.h;
        double mu[D];
#pragma unroll
        for (int a = 0; a < D; a++) mu[a] = A.mu[a];
#pragma unroll
        for (int k = 0; k < KP; k++) {
            mu[0] = fma(A.coefA[k], X[k], mu[0]);
            if (D > 1) mu[D - 1] = fma(A.coefB[k], X[k], mu[D - 1]);
        }
        if constexpr (CT) {
            // the lane's gains, exactly as ctcrw_step forms them (the compiler merges the two)
            const double F = L.C.p11 + h;
            const double detF = (D == 1) ? F : F * F;
            const bool upd = !na && !(detF <= 0.0);
            const double updf = upd ? 1.0 : 0.0;
            const double iF = rcp(upd ? F : 1.0) * updf;
            const double bm = (na || upd) ? 1.0 : 0.0;
            const double kf1 = L.C.p11 * iF, kf2 = L.C.p12 * iF;
            const double k1 = fma(tr.t12, kf2, kf1), k2 = tr.e * kf2, c1 = 1.0 - k1;
            const double b1 = bm * tr.b1, b2 = bm * tr.b2;
#pragma unroll
            for (int k = 0; k < KP; k++) {
                const double yk = (D > 1 && ((A.drift_dim1 >> k) & 1u)) ? y[D - 1] : y[0];
                const double xk = (D > 1 && ((A.drift_dim1 >> k) & 1u)) ? L.M.x[D - 1] : L.M.x[0];
                const double uk = upd ? yk - xk : 0.0;
                const double mx = cx[k], mv = cv[k];
                gk[k] = fma(-iF * mx, uk, gk[k]);
                cx[k] = fma(b1, X[k], fma(tr.t12, mv, c1 * mx));
                cv[k] = fma(b2, X[k], fma(tr.e, mv, -k2 * mx));
            }
            ctcrw_step<D, DRIFT_MASK>(L, tr, h, mu, y, na);
        } else {
            ScalGain G;
            const double x0 = L.M.x[0], x1 = L.M.x[D - 1];
            scal_cov_step<D, DRIFT_MASK, HAS_P2>(L.C, tr, h, na, G);
            const bool scored = G.iF != 0.0;
#pragma unroll
            for (int k = 0; k < KP; k++) {
                const bool d1 = D > 1 && ((A.drift_dim1 >> k) & 1u);
                const double uk = scored ? (d1 ? y[D - 1] - x1 : y[0] - x0) : 0.0;
                const double mx = cx[k];
                gk[k] = fma(-G.iF * mx, uk, gk[k]);
                cx[k] = fma(tr.b, X[k], G.c * mx);
            }
            scal_mean_step<D, DRIFT_MASK, HAS_P2>(L.M, tr, G, mu, y, scored);
        }
    }
    __device__ __forceinline__ void dump_to(double* o) const {      // o[k * WAVE]
        double st[NBASE];
        L.dump(st);
#pragma unroll
        for (int k = 0; k < NBASE; k++) o[k * WAVE] = st[k];
        int n = NBASE;
#pragma unroll
        for (int k = 0; k < KP; k++) { o[(n++) * WAVE] = cx[k]; if constexpr (CT) o[(n++) * WAVE] = cv[k]; }
    }
};

int drift_general_nstate(int model, int d, int k) {
    const int kp = (k + 3) / 4 * 4;
    return iso_nstate(model, d) + kp * (model == M_CTCRW ? 2 : 1);
}

template <int MODEL, int D, int NG, bool UNI>
__device__ __forceinline__ void run_lane_drift_general(const IsoArgs& A, int g, int chunk) {
    constexpr int KP = 4 * NG;
    constexpr int U = (NG <= 1) ? 4 : 2;            // fp64-issue-bound: a short look-ahead is enough, and registers are what this kernel lacks
    typedef DriftGenLane<MODEL, D, KP> Lane;
    typedef DriftModel<MODEL> DM;
    constexpr int SD = Lane::SD;
    constexpr int W = 1 + D + KP;                              // register block row: [dt | y | columns]
    const int lane = threadIdx.x & 63;
    const TileView& tv = A.tv;
    const int C = tv.C, c_obs = tv.c_obs, K = A.drift_k, c_col = A.c_col;
    const int nacc = 4 + D + K;
    const double* base = tv.tiles + tv.group_off[g] + lane;
    const int L = tv.group_len[g];
    const int ns = tv.lane_nsteps[g * WAVE + lane];
    int s_begin, s_acc, s_end;
    window_bounds(L, A.n_chunks, A.window, 0, chunk, s_begin, s_acc, s_end, 0);
    int chan[KP];
#pragma unroll
    for (int k = 0; k < KP; k++) chan[k] = c_col + (k < K ? k : 0);

    double bufA[U][W], bufB[U][W];
    auto load = [&](double (&dst)[U][W], int s0) {
        const double* p = base + (int64_t)s0 * C * WAVE;
#pragma unroll
        for (int u = 0; u < U; u++) {
            dst[u][0] = 0.0;
            if (!UNI) dst[u][0] = p[(u * C) * WAVE];           // the dt channel exists whenever the grid is not regular
#pragma unroll
            for (int a = 0; a < D; a++) dst[u][1 + a] = p[(u * C + c_obs + a) * WAVE];
#pragma unroll
            for (int k = 0; k < KP; k++) dst[u][1 + D + k] = p[(u * C + chan[k]) * WAVE];
        }
    };
    load(bufA, s_begin);
    Lane S;
    {
        double a0[SD];
        if (s_begin == 0) {
#pragma unroll
            for (int c = 0; c < SD; c++) a0[c] = tv.a0[((int64_t)g * SD + c) * WAVE + lane];
        } else {
#pragma unroll
            for (int a = 0; a < D; a++) {
                const double y0 = bufA[0][1 + a];
                if constexpr (DM::CT) { a0[2 * a] = (y0 == y0) ? y0 : 0.0; a0[2 * a + 1] = 0.0; }
                else a0[a] = (y0 == y0) ? y0 : 0.0;
            }
        }
        S.init(a0, A);
    }
    typename DM::Trans tr_uni;
    if constexpr (DM::CT) tr_uni = A.ctr; else tr_uni = A.str;
    auto block = [&](const double (&blk)[U][W], int s0) {
        if (s0 == s_acc && s_acc > s_begin) {
            S.dump_to(A.bnd + (((int64_t)chunk * tv.n_groups + g) * 2 + 0) * A.bnd_stride * WAVE + lane);
            S.reset_acc();
        }
#pragma unroll
        for (int u = 0; u < U; u++)
            if (s0 + u < ns) {
                if constexpr (UNI) S.step(A, tr_uni, &blk[u][1], &blk[u][1 + D]);
                else {
                    typename DM::Trans tr;
                    if constexpr (DM::CT) ctcrw_trans(blk[u][0], A.tau, A.beta, A.sigma, tr);
                    else if constexpr (MODEL == M_OU_SSM) ou_trans(blk[u][0], A.tau, A.sigma, tr);
                    else bm_trans(blk[u][0], A.sigma, tr);
                    S.step(A, tr, &blk[u][1], &blk[u][1 + D]);
                }
            }
    };
    for (int s0 = s_begin; s0 < s_end; s0 += 2 * U) {
        load(bufB, s0 + U);
        block(bufA, s0);
        load(bufA, s0 + 2 * U);
        if (s0 + U < s_end) block(bufB, s0 + U);
    }
    if (A.n_chunks > 1 && chunk + 1 < A.n_chunks)
        S.dump_to(A.bnd + (((int64_t)chunk * tv.n_groups + g) * 2 + 1) * A.bnd_stride * WAVE + lane);
    double out[4 + D];
    if constexpr (DM::CT) ctcrw_finish<D, DRIFT_MASK>(S.L, out); else scal_finish<D, DRIFT_MASK>(S.L, out);
    const bool empty = s_acc >= s_end;
#pragma unroll
    for (int k = 0; k < 4 + D; k++) {
        const double t = wave_sum(empty ? 0.0 : out[k]);
        if (lane == 0) A.partials[((int64_t)chunk * nacc + k) * tv.n_groups + g] = t;
    }
#pragma unroll
    for (int k = 0; k < KP; k++) {
        const double t = wave_sum(empty ? 0.0 : S.gk[k]);
        if (lane == 0 && k < K) A.partials[((int64_t)chunk * nacc + 4 + D + k) * tv.n_groups + g] = t;
    }
}

template <int MODEL, int D, int NG, bool UNI>
__global__ __launch_bounds__(WG_WAVES * WAVE, 1) void iso_drift_general_kernel(const IsoArgs A) {
    if (blockIdx.x == 0 && threadIdx.x == 0 && A.chk_out) *A.chk_out = 0.0;
    int g, part, chunk;
    if (!decode_block(A, A.n_chunks, g, part, chunk)) return;
    run_lane_drift_general<MODEL, D, NG, UNI>(A, g, chunk);
}

template <int MODEL, int D>
static hipError_t launch_ng_general(const IsoArgs& a, dim3 grid, hipStream_t s) {
    dim3 block(WG_WAVES * WAVE);
    const int ng = (a.drift_k + 3) / 4;
    const bool uni = a.uniform_dt != 0;
    switch (ng) {
#define SSDE_CASE(N) case N: if (uni) hipLaunchKernelGGL((iso_drift_general_kernel<MODEL, D, N, true>), grid, block, 0, s, a); \
                             else hipLaunchKernelGGL((iso_drift_general_kernel<MODEL, D, N, false>), grid, block, 0, s, a); break;
        SSDE_CASE(1) SSDE_CASE(2) SSDE_CASE(3) SSDE_CASE(4) SSDE_CASE(5) SSDE_CASE(6)
#undef SSDE_CASE
        default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

hipError_t launch_iso_drift_general(int model, int d, const IsoArgs& a, hipStream_t s) {
    if (a.n_parts != 1 || a.drift_k < 1 || a.drift_k > DRIFT_KMAX) return hipErrorInvalidValue;
    const int g8 = (a.tv.n_groups + 7) / 8;
    dim3 grid((g8 * 8 * a.n_chunks + WG_WAVES - 1) / WG_WAVES);
    if (grid.x == 0) return hipSuccess;
    if (model == M_CTCRW && d == 1) return launch_ng_general<M_CTCRW, 1>(a, grid, s);
    if (model == M_CTCRW && d == 2) return launch_ng_general<M_CTCRW, 2>(a, grid, s);
    if (model == M_OU_SSM && d == 1) return launch_ng_general<M_OU_SSM, 1>(a, grid, s);
    if (model == M_OU_SSM && d == 2) return launch_ng_general<M_OU_SSM, 2>(a, grid, s);
    if (model == M_BM_SSM && d == 1) return launch_ng_general<M_BM_SSM, 1>(a, grid, s);
    if (model == M_BM_SSM && d == 2) return launch_ng_general<M_BM_SSM, 2>(a, grid, s);
    return hipErrorInvalidValue;
}

// =====================================================================================================================
// EXACT Hessian of the data term over the drift coefficients (shared-covariance case).  The innovation is LINEAR in them --
// u_a(i) = y_a(i) - x_a(i), d x_a / d beta_k = mx_k(i), no second derivative -- and the covariance half does not see them, so
//     d2 nllk / d beta_k d beta_l = sum_i F_i^-1 mx_k(i) mx_l(i)        (k, l feeding the same dimension; 0 otherwise)
// exactly: the Gauss-Newton form IS the Hessian, and it does not even depend on the observations.  This is the H_uu block of
// the Laplace approximation for a smooth drift (random = "coeff_re", R/sde.R:510-525), which TMB gets from second-order AD.
// One pass over the design columns: the coefficient pairs are cut into HESS_T x HESS_T tiles (blockIdx.y), a wave carries the
// column recursions of its tile's two groups and HESS_T^2 accumulators through its time window (same windows, same warm-up,
// same gains as the evaluation: the plan was verified by the evaluation at these parameters that precedes the call).
// A slot is a streamed column (chan >= 0) or the intercept of a dimension (chan < 0: a column of ones).
template <int MODEL>
__global__ __launch_bounds__(WG_WAVES * WAVE, 1) void iso_drift_hess_kernel(const IsoArgs A, const DriftHessArgs H) {
    typedef DriftModel<MODEL> DM;
    constexpr bool CT = DM::CT;
    constexpr int U = 4;
    int g, part, chunk;
    if (!decode_block(A, A.n_chunks, g, part, chunk)) return;
    const int tile = blockIdx.y, ti = H.tile_i[tile], tj = H.tile_j[tile];
    const int lane = threadIdx.x & 63;
    const TileView& tv = A.tv;
    const int C = tv.C;
    const double* base = tv.tiles + tv.group_off[g] + lane;
    const int L = tv.group_len[g];
    const int ns = tv.lane_nsteps[g * WAVE + lane];
    int s_begin, s_acc, s_end;
    window_bounds(L, A.n_chunks, A.window, A.t0, chunk, s_begin, s_acc, s_end, A.t0_delta);
    int s_stat = (A.gain_last + SHARED_U - 1) / SHARED_U * SHARED_U;
    if (A.gain_stat[0] == 0.0) s_stat = INT32_MAX;
    __shared__ double gain_slab[WG_WAVES][DRIFT_SLAB_ROWS * GAIN_ROW];
    double* slab = gain_slab[threadIdx.x >> 6];
    int slab_row0 = -1;
    // the tile's slots: 2 x HESS_T (group ti, group tj); a slot beyond the list repeats slot 0 and is dropped by the host
    int chan[2 * HESS_T];
#pragma unroll
    for (int k = 0; k < 2 * HESS_T; k++) {
        const int idx = (k < HESS_T ? ti : tj) * HESS_T + (k % HESS_T);
        chan[k] = H.chan[idx < H.n ? idx : 0];
    }
    typename DM::Trans tr;
    if constexpr (CT) tr = A.ctr; else tr = A.str;
    const typename DM::Gain Gs = gain_from_row<typename DM::Gain>(A.gain_stat);
    double cx[2 * HESS_T], cv[CT ? 2 * HESS_T : 1], acc[HESS_T][HESS_T];
#pragma unroll
    for (int k = 0; k < 2 * HESS_T; k++) { cx[k] = 0.0; if (CT) cv[k] = 0.0; }
#pragma unroll
    for (int a = 0; a < HESS_T; a++)
#pragma unroll
        for (int b = 0; b < HESS_T; b++) acc[a][b] = 0.0;
    double bufA[U][2 * HESS_T], bufB[U][2 * HESS_T];
    auto load = [&](double (&dst)[U][2 * HESS_T], int s0) {
        const double* p = base + (int64_t)s0 * C * WAVE;
#pragma unroll
        for (int u = 0; u < U; u++)
#pragma unroll
            for (int k = 0; k < 2 * HESS_T; k++) dst[u][k] = chan[k] >= 0 ? p[(u * C + chan[k]) * WAVE] : 1.0;      // (uniform select)
    };
    auto row = [&](const typename DM::Gain& G, const double* X, bool score) {
        const double w = score ? G.iF : 0.0;
#pragma unroll
        for (int a = 0; a < HESS_T; a++) {
            const double wa = w * cx[a];
#pragma unroll
            for (int b = 0; b < HESS_T; b++) acc[a][b] = fma(wa, cx[HESS_T + b], acc[a][b]);
        }
        const bool scored = G.iF != 0.0;
#pragma unroll
        for (int k = 0; k < 2 * HESS_T; k++) {
            const double mx = cx[k];
            if constexpr (CT) {
                const double mv = cv[k], du = scored ? -mx : 0.0;
                cx[k] = fma(G.bm * tr.b1, X[k], fma(G.k1, du, fma(tr.t12, mv, mx)));
                cv[k] = fma(G.bm * tr.b2, X[k], fma(G.k2, du, tr.e * mv));
            } else {
                cx[k] = fma(tr.b, X[k], G.c * mx);
            }
        }
    };
    auto block = [&](const double (&blk)[U][2 * HESS_T], int s0) {
        if (s0 < s_stat) {
            if (slab_row0 < 0 || s0 >= slab_row0 + DRIFT_SLAB_ROWS) {
                slab_row0 = s0;
                const int glast = A.gain_last;
                const double* __restrict__ gain = A.gain;
#pragma unroll 4
                for (int r = 0; r < DRIFT_SLAB_ROWS; r += WAVE / GAIN_ROW) {
                    const int rr = r + lane / GAIN_ROW;
                    slab[rr * GAIN_ROW + (lane % GAIN_ROW)] = gain[(int64_t)min(s0 + rr, glast) * GAIN_ROW + (lane % GAIN_ROW)];
                }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            }
#pragma unroll
            for (int u = 0; u < U; u++)
                if (s0 + u < ns) row(gain_from_row<typename DM::Gain>(slab + (s0 + u - slab_row0) * GAIN_ROW), &blk[u][0], s0 + u >= s_acc);
        } else {
#pragma unroll
            for (int u = 0; u < U; u++)
                if (s0 + u < ns) row(Gs, &blk[u][0], s0 + u >= s_acc);
        }
    };
    load(bufA, s_begin);
    for (int s0 = s_begin; s0 < s_end; s0 += 2 * U) {
        load(bufB, s0 + U);
        block(bufA, s0);
        load(bufA, s0 + 2 * U);
        if (s0 + U < s_end) block(bufB, s0 + U);
    }
#pragma unroll
    for (int a = 0; a < HESS_T; a++)
#pragma unroll
        for (int b = 0; b < HESS_T; b++) {
            const double t = wave_sum(acc[a][b]);
            if (lane == 0) H.partials[(((int64_t)tile * HESS_T * HESS_T + a * HESS_T + b) * A.n_chunks + chunk) * tv.n_groups + g] = t;
        }
}

// H[k + l n] (and its mirror) = the tile's partials summed over windows and groups in a fixed order
__global__ __launch_bounds__(64) void iso_drift_hess_reduce_kernel(const DriftHessArgs H, int n_items) {
    const int tile = blockIdx.x, e = blockIdx.y, a = e / HESS_T, b = e % HESS_T;
    const int k = H.tile_i[tile] * HESS_T + a, l = H.tile_j[tile] * HESS_T + b;
    if (k >= H.n || l >= H.n) return;
    const double* p = H.partials + ((int64_t)tile * HESS_T * HESS_T + e) * n_items;
    double s = 0.0;
    for (int i = threadIdx.x; i < n_items; i += 64) s += p[i];
    s = wave_sum(s);
    if (threadIdx.x == 0) { H.hess[k + (int64_t)l * H.n] = s; H.hess[l + (int64_t)k * H.n] = s; }
}

hipError_t launch_iso_drift_hess(int model, const IsoArgs& a, const DriftHessArgs& hx, int n_tiles, hipStream_t s) {
    const int g8 = (a.tv.n_groups + 7) / 8;
    dim3 grid((g8 * 8 * a.n_chunks + WG_WAVES - 1) / WG_WAVES, n_tiles), block(WG_WAVES * WAVE);
    if (grid.x == 0) return hipSuccess;
    if (model == M_CTCRW) hipLaunchKernelGGL((iso_drift_hess_kernel<M_CTCRW>), grid, block, 0, s, a, hx);
    else if (model == M_OU_SSM) hipLaunchKernelGGL((iso_drift_hess_kernel<M_OU_SSM>), grid, block, 0, s, a, hx);
    else if (model == M_BM_SSM) hipLaunchKernelGGL((iso_drift_hess_kernel<M_BM_SSM>), grid, block, 0, s, a, hx);
    else return hipErrorInvalidValue;
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(iso_drift_hess_reduce_kernel, dim3(n_tiles, HESS_T * HESS_T), dim3(64), 0, s, hx, a.n_chunks * a.tv.n_groups);
    return hipGetLastError();
}

template <int MODEL, int D>
static hipError_t launch_ng(const IsoArgs& a, dim3 grid, hipStream_t s, hipEvent_t ev0, hipEvent_t ev1) {
    dim3 block(WG_WAVES * WAVE);
    const int ng = (a.drift_k + 3) / 4;
    switch (ng) {
#define SSDE_CASE(N) case N: if (ev0) hipExtLaunchKernelGGL((iso_drift_kernel<MODEL, D, N>), grid, block, 0, s, ev0, ev1, 0, a); \
                             else hipLaunchKernelGGL((iso_drift_kernel<MODEL, D, N>), grid, block, 0, s, a); break;
        SSDE_CASE(1) SSDE_CASE(2) SSDE_CASE(3) SSDE_CASE(4) SSDE_CASE(5) SSDE_CASE(6)
#undef SSDE_CASE
        default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

hipError_t launch_iso_drift(int model, int d, const IsoArgs& a, hipStream_t s, hipEvent_t ev0, hipEvent_t ev1) {
    if (a.n_parts != 1 || a.drift_k < 1 || a.drift_k > DRIFT_KMAX) return hipErrorInvalidValue;
    const int g8 = (a.tv.n_groups + 7) / 8;
    dim3 grid((g8 * 8 * a.n_chunks + WG_WAVES - 1) / WG_WAVES);
    if (grid.x == 0) return hipSuccess;
    if (model == M_CTCRW && d == 1) return launch_ng<M_CTCRW, 1>(a, grid, s, ev0, ev1);
    if (model == M_CTCRW && d == 2) return launch_ng<M_CTCRW, 2>(a, grid, s, ev0, ev1);
    if (model == M_OU_SSM && d == 1) return launch_ng<M_OU_SSM, 1>(a, grid, s, ev0, ev1);
    if (model == M_OU_SSM && d == 2) return launch_ng<M_OU_SSM, 2>(a, grid, s, ev0, ev1);
    if (model == M_BM_SSM && d == 1) return launch_ng<M_BM_SSM, 1>(a, grid, s, ev0, ev1);
    if (model == M_BM_SSM && d == 2) return launch_ng<M_BM_SSM, 2>(a, grid, s, ev0, ev1);
    return hipErrorInvalidValue;
}

}  // namespace ssde
