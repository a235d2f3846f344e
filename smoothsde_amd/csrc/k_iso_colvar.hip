// k_iso_colvar.hip -- lane = track Kalman lanes with ROW-VARYING tau / nu (kappa, sigma) for gfx950 (CTCRW, OU_SSM, BM_SSM).
//
// The batch-scale form of the model the reference exists for: the SDE parameters smooth in covariates,
//     par_mat.row(i) = X_fe coeff_fe + X_re coeff_re ;  tau_i = exp(par_mat(i, d)), nu_i = exp(par_mat(i, d + 1))
// (nllk_ctcrw.hpp:143-156, nllk_ou_ssm.hpp:113-124, nllk_bm_ssm.hpp:98-108), filtered by the same loop
// (nllk_ctcrw.hpp:206-241).  The lane = direction kernels (k_tv.hip) spend a wave-row per track-row whatever the batch:
// right for one animal, an order of magnitude off the chip's fp64 rate for 10^4 of them.  Here a lane is a TRACK, as in the
// constant-coefficient kernels, and the gradient with respect to the coefficient of design column k comes from the
// TANGENT of the filter in that direction: the linearised step is the same for every column, only the seed differs --
//     (dP, da)_k  <-  Lin_i (dP, da)_k  +  X_k(i) * seed_type(k)(i)
// where Lin_i is the Jacobian of row i's update + prediction with respect to (P, a) and seed_t the derivative of
// (T, Q, B, H) with respect to log tau (t = 1), log nu / kappa (t = 2) or log sigma_obs (t = 0, a column of ones) at THIS
// row's parameters.  3 + 2 d doubles of state and ~40 fp64 instructions per column and row (CTCRW, d = 2).
//
// Work split: the columns are dealt to the four waves of a workgroup (PARTS of at most CV_KC columns: a wave's register
// budget); every part recomputes the primal filter (the exp's of the row's tau / nu, T, Q, the gains) and carries its own
// columns, sorted by type so that the unrolled column loop takes uniform branches.  One workgroup per (64-track group,
// time window); windows, warm-up and the verified hand-over as in k_iso.hip.
// Layout: the tiles of ssde_device.hpp with the design columns as further channels (as k_iso_drift.hip).
#include <type_traits>

#include "ssde_device.hpp"

namespace ssde {

// ---- the per-row linearisation and the column recursion, CTCRW ------------------------------------------------------
template <int D, int KC>
struct CvCtcrw {
    static constexpr int SD = 2 * D;
    static constexpr int NCOL = 3 + 2 * D;
    static constexpr int NBASE = SD + 3 + 2;
    static constexpr int NSTATE = NBASE + KC * NCOL;
    double x[D], v[D], p11, p12, p22;
    LogAcc ld;
    double accq;
    double mx, mv, gmu[D];
    double d11[KC], d12[KC], d22[KC], tx[KC][D], tv[KC][D], g[KC];

    __device__ __forceinline__ void init(const double* a0, const double* p0) {
#pragma unroll
        for (int a = 0; a < D; a++) { x[a] = a0[2 * a]; v[a] = a0[2 * a + 1]; gmu[a] = 0.0; }
        p11 = p0[0]; p12 = p0[1]; p22 = p0[2];
        ld.init(); accq = 0.0; mx = mv = 0.0;
#pragma unroll
        for (int k = 0; k < KC; k++) {
            d11[k] = d12[k] = d22[k] = g[k] = 0.0;
#pragma unroll
            for (int a = 0; a < D; a++) tx[k][a] = tv[k][a] = 0.0;
        }
    }
    __device__ __forceinline__ void reset_acc() {
        ld.init(); accq = 0.0;
#pragma unroll
        for (int a = 0; a < D; a++) gmu[a] = 0.0;
#pragma unroll
        for (int k = 0; k < KC; k++) g[k] = 0.0;
    }
    // One row: score y (unless NA), the column tangents, then the prediction over the row's interval (ctcrw_step's
    // arrangement: filtered-form covariance update, Joseph-form sensitivities; ssde_math.hpp)
    __device__ __forceinline__ void step(const CvPart& P, const CtcrwTrans& tr, double h, const double* mu, const double* y,
                                         bool na, const double* X) {
        const double F = p11 + h;
        const double detF = (D == 1) ? F : F * F;                  // nllk_ctcrw.hpp:16-19, 223
        const bool upd = !na && !(detF <= 0.0);                    // :214, 226
        const double updf = upd ? 1.0 : 0.0;
        const double Fe = upd ? F : 1.0;
        const double iF = rcp(Fe) * updf;
        ld.mul(Fe);
        const double bm = (na || upd) ? 1.0 : 0.0;                 // Q3 (:226-228)
        const double e = tr.e, t12 = tr.t12, e2 = tr.e2;
        const double a = fma(h, iF, 1.0 - updf), a2 = a * a, aiF = a * iF;
        const double kf1 = p11 * iF, kf2 = p12 * iF;
        const double f11 = p11 * a, f12 = p12 * a, f22 = fma(-p12, kf2, p22);
        const double m = fma(t12, f22, f12);
        const double k1 = fma(t12, kf2, kf1), k2 = e * kf2, c1 = 1.0 - k1;
        double u[D], mue[D], wv[D];
        double su2 = 0.0;
#pragma unroll
        for (int a_ = 0; a_ < D; a_++) {
            const double ys = upd ? y[a_] : x[a_];
            u[a_] = ys - x[a_];
            su2 = fma(u[a_], u[a_], su2);
            mue[a_] = bm * mu[a_];
            wv[a_] = v[a_] - mue[a_];
        }
        accq = fma(iF, su2, accq);
        const double gF = fma(-0.5 * iF * iF, su2, 0.5 * (double)D * iF);     // d nllk / d F of this row
        // ---- seeds ------------------------------------------------------------------------------------------------
        // type 0, log sigma_obs: dh = 2 h enters F, the filtered covariance (k k' dh) and the gain (-k dh / F)
        const double h2 = 2.0 * h;
        double s0_11, s0_12, s0_22, s0_k1, s0_k2;
        {
            const double q1 = kf1 * h2, q2 = kf2 * h2;
            const double g11 = kf1 * q1, g12 = kf2 * q1, g22 = kf2 * q2;
            const double dkf1 = -q1 * iF, dkf2 = -q2 * iF;
            const double dm = fma(t12, g22, g12);
            s0_11 = fma(t12, g12 + dm, g11); s0_12 = e * dm; s0_22 = e2 * g22;
            s0_k1 = fma(t12, dkf2, dkf1); s0_k2 = e * dkf2;
        }
        // type 1, log tau: T, B and Q move
        const double s1_11 = fma(tr.dt12x2, m, tr.dq11), s1_12 = fma(tr.dt12e, f22, fma(tr.de, m, tr.dq12)),
                     s1_22 = fma(tr.edex2, f22, tr.dq22);
        double s1_x[D], s1_v[D];
#pragma unroll
        for (int a_ = 0; a_ < D; a_++) {
            const double w = fma(kf2, u[a_], wv[a_]);              // d k u + d(T a + B mu): (dt12, de) (kf2 u + v - mu)
            s1_x[a_] = tr.dt12 * w; s1_v[a_] = tr.de * w;
        }
        // type 2, log nu: Q only (dQ = 2 Q)
        const double s2_11 = 2.0 * tr.q11, s2_12 = 2.0 * tr.q12, s2_22 = 2.0 * tr.q22;
        // ---- columns ----------------------------------------------------------------------------------------------
        auto col = [&](int k, auto type) {
            constexpr int T = decltype(type)::value;
            const double Xk = X[k];
            const double c11 = d11[k], c12 = d12[k], c22 = d22[k];
            const double dF = (T == 0) ? fma(Xk, h2, c11) : c11;
            double sud = 0.0;
#pragma unroll
            for (int a_ = 0; a_ < D; a_++) sud = fma(u[a_], tx[k][a_], sud);
            g[k] = fma(gF, dF, fma(-iF, sud, g[k]));
            const double w = fma(-kf2, c11, c12);
            const double g11 = a2 * c11, g12 = a * w, g22 = fma(-kf2, c12 + w, c22);
            const double dkf1 = c11 * aiF, dkf2 = w * iF;
            const double dm = fma(t12, g22, g12);
            double n11 = fma(t12, g12 + dm, g11), n12 = e * dm, n22 = e2 * g22;
            double dk1 = fma(t12, dkf2, dkf1), dk2 = e * dkf2;
            if (T == 0) { n11 = fma(Xk, s0_11, n11); n12 = fma(Xk, s0_12, n12); n22 = fma(Xk, s0_22, n22);
                          dk1 = fma(Xk, s0_k1, dk1); dk2 = fma(Xk, s0_k2, dk2); }
            if (T == 1) { n11 = fma(Xk, s1_11, n11); n12 = fma(Xk, s1_12, n12); n22 = fma(Xk, s1_22, n22); }
            if (T == 2) { n11 = fma(Xk, s2_11, n11); n12 = fma(Xk, s2_12, n12); n22 = fma(Xk, s2_22, n22); }
            d11[k] = n11; d12[k] = n12; d22[k] = n22;
#pragma unroll
            for (int a_ = 0; a_ < D; a_++) {
                const double txk = tx[k][a_], tvk = tv[k][a_];
                double nx = fma(dk1, u[a_], fma(t12, tvk, c1 * txk));
                double nv = fma(dk2, u[a_], fma(e, tvk, -k2 * txk));
                if (T == 1) { nx = fma(Xk, s1_x[a_], nx); nv = fma(Xk, s1_v[a_], nv); }
                tx[k][a_] = nx; tv[k][a_] = nv;
            }
        };
#pragma unroll
        for (int k = 0; k < KC; k++) {
            if (k < P.n_col) {                                     // (wave-uniform branches: the part's columns are sorted by type)
                if (k < P.n0) col(k, std::integral_constant<int, 0>());
                else if (k < P.n01) col(k, std::integral_constant<int, 1>());
                else col(k, std::integral_constant<int, 2>());
            }
        }
        if (P.with_mu) {                                           // d / d mu_a: one data-independent chain for every dimension
            const double imx = iF * mx;
            const double nx = fma(bm, tr.b1, fma(t12, mv, c1 * mx)), nv = fma(bm, tr.b2, fma(e, mv, -k2 * mx));
#pragma unroll
            for (int a_ = 0; a_ < D; a_++) gmu[a_] = fma(-imx, u[a_], gmu[a_]);
            mx = nx; mv = nv;
        }
        // ---- primal -------------------------------------------------------------------------------------------------
#pragma unroll
        for (int a_ = 0; a_ < D; a_++) {                           // a = T a + K u + B mu (:238)
            const double nx = fma(tr.b1, mue[a_], fma(k1, u[a_], fma(t12, v[a_], x[a_])));
            const double nv = fma(tr.b2, mue[a_], fma(k2, u[a_], e * v[a_]));
            x[a_] = nx; v[a_] = nv;
        }
        p11 = fma(t12, f12 + m, f11) + tr.q11;                     // P = T P~ T' + Q (:240-241)
        p12 = fma(e, m, tr.q12);
        p22 = fma(e2, f22, tr.q22);
    }
    __device__ __forceinline__ void dump_to(double* o) const {      // o[k * WAVE]
        int n = 0;
#pragma unroll
        for (int a = 0; a < D; a++) { o[(n++) * WAVE] = x[a]; o[(n++) * WAVE] = v[a]; }
        o[(n++) * WAVE] = p11; o[(n++) * WAVE] = p12; o[(n++) * WAVE] = p22;
        o[(n++) * WAVE] = mx; o[(n++) * WAVE] = mv;
#pragma unroll
        for (int k = 0; k < KC; k++) {
            o[(n++) * WAVE] = d11[k]; o[(n++) * WAVE] = d12[k]; o[(n++) * WAVE] = d22[k];
#pragma unroll
            for (int a = 0; a < D; a++) { o[(n++) * WAVE] = tx[k][a]; o[(n++) * WAVE] = tv[k][a]; }
        }
    }
    __device__ __forceinline__ double value() const { return 0.5 * ((double)D * ld.value() + accq); }
    // this row's transition from the linear predictors p1 = log tau, p2 = log nu (nllk_ctcrw.hpp:152-156)
    static __device__ __forceinline__ void trans(double dt, double p1, double p2, CtcrwTrans& tr) {
        const double tau = exp(p1), nu = exp(p2);
        const double beta = rcp(tau);
        ctcrw_trans(dt, tau, beta, 2.0 * nu / sqrt(M_PI * tau), tr);
    }
};

// ---- OU_SSM / BM_SSM: scalar covariance --------------------------------------------------------------------------------
template <int D, int KC, bool HAS_P2>
struct CvScal {
    static constexpr int SD = D;
    static constexpr int NCOL = 1 + D;
    static constexpr int NBASE = SD + 1 + 1;
    static constexpr int NSTATE = NBASE + KC * NCOL;
    double x[D], p;
    LogAcc ld;
    double accq;
    double mx, gmu[D];
    double dp[KC], tx[KC][D], g[KC];

    __device__ __forceinline__ void init(const double* a0, const double* p0) {
#pragma unroll
        for (int a = 0; a < D; a++) { x[a] = a0[a]; gmu[a] = 0.0; }
        p = p0[0];
        ld.init(); accq = 0.0; mx = 0.0;
#pragma unroll
        for (int k = 0; k < KC; k++) {
            dp[k] = g[k] = 0.0;
#pragma unroll
            for (int a = 0; a < D; a++) tx[k][a] = 0.0;
        }
    }
    __device__ __forceinline__ void reset_acc() {
        ld.init(); accq = 0.0;
#pragma unroll
        for (int a = 0; a < D; a++) gmu[a] = 0.0;
#pragma unroll
        for (int k = 0; k < KC; k++) g[k] = 0.0;
    }
    // scal_cov_step + scal_mean_step (ssde_math.hpp) with the direction loops replaced by the column loop
    __device__ __forceinline__ void step(const CvPart& P, const ScalTrans& tr, double h, const double* mu, const double* y,
                                         bool na, const double* X) {
        const double F = p + h;
        const bool upd = !na && !(fabs(F) <= 0.0);                 // nllk_ou_ssm.hpp:190-195, nllk_bm_ssm.hpp:152-157
        const double updf = upd ? 1.0 : 0.0;
        const double Fe = upd ? F : 1.0;
        const double iF = rcp(Fe) * updf;
        ld.mul(Fe);
        const double t = HAS_P2 ? tr.t : 1.0, dt_ = HAS_P2 ? tr.dt_ : 0.0;
        const double a = fma(h, iF, 1.0 - updf), b = p * iF;
        const double c = t * a, k = t * b, tc = t * c;
        const double tiF = t * iF, ca = tiF * a, tca = tc * a, cp = c * p;
        double u[D];
        double su2 = 0.0;
#pragma unroll
        for (int a_ = 0; a_ < D; a_++) {
            const double ys = upd ? y[a_] : x[a_];
            u[a_] = ys - x[a_];
            su2 = fma(u[a_], u[a_], su2);
        }
        accq = fma(iF, su2, accq);
        const double gF = fma(-0.5 * iF * iF, su2, 0.5 * (double)D * iF);
        const double h2 = 2.0 * h, bh = b * h2;
        const double s0_k = -tiF * bh, s0_p = k * t * bh;           // type 0: d sigma_obs
        const double s1_k = HAS_P2 ? dt_ * b : 0.0;                 // type 1: log tau (OU) / log sigma (BM)
        const double s1_p = HAS_P2 ? fma(2.0 * dt_, cp, tr.dq) : tr.dq;
        double s1_x[D];
#pragma unroll
        for (int a_ = 0; a_ < D; a_++) s1_x[a_] = HAS_P2 ? fma(s1_k, u[a_], fma(tr.dt_, x[a_], tr.db * mu[a_])) : 0.0;
        const double s2_p = tr.q;                                   // type 2: log kappa (OU)
        auto col = [&](int k_, auto type) {
            constexpr int T = decltype(type)::value;
            const double Xk = X[k_];
            const double cdp = dp[k_];
            const double dF = (T == 0) ? fma(Xk, h2, cdp) : cdp;
            double sud = 0.0;
#pragma unroll
            for (int a_ = 0; a_ < D; a_++) sud = fma(u[a_], tx[k_][a_], sud);
            g[k_] = fma(gF, dF, fma(-iF, sud, g[k_]));
            double dk = ca * cdp, np_ = tca * cdp;
            if (T == 0) { dk = fma(Xk, s0_k, dk); np_ = fma(Xk, s0_p, np_); }
            if (T == 1) np_ = fma(Xk, s1_p, np_);
            if (T == 2) np_ = fma(Xk, s2_p, np_);
            dp[k_] = np_;
#pragma unroll
            for (int a_ = 0; a_ < D; a_++) {
                double nx = fma(dk, u[a_], c * tx[k_][a_]);
                if (T == 1 && HAS_P2) nx = fma(Xk, s1_x[a_], nx);
                tx[k_][a_] = nx;
            }
        };
#pragma unroll
        for (int k_ = 0; k_ < KC; k_++) {
            if (k_ < P.n_col) {
                if (k_ < P.n0) col(k_, std::integral_constant<int, 0>());
                else if (k_ < P.n01) col(k_, std::integral_constant<int, 1>());
                else col(k_, std::integral_constant<int, 2>());
            }
        }
        if (P.with_mu) {
            const double imx = iF * mx;
#pragma unroll
            for (int a_ = 0; a_ < D; a_++) gmu[a_] = fma(-imx, u[a_], gmu[a_]);
            mx = fma(c, mx, tr.b);
        }
#pragma unroll
        for (int a_ = 0; a_ < D; a_++) x[a_] = fma(tr.b, mu[a_], fma(k, u[a_], HAS_P2 ? tr.t * x[a_] : x[a_]));
        p = fma(tc, p, tr.q);
    }
    __device__ __forceinline__ void dump_to(double* o) const {
        int n = 0;
#pragma unroll
        for (int a = 0; a < D; a++) o[(n++) * WAVE] = x[a];
        o[(n++) * WAVE] = p;
        o[(n++) * WAVE] = mx;
#pragma unroll
        for (int k = 0; k < KC; k++) {
            o[(n++) * WAVE] = dp[k];
#pragma unroll
            for (int a = 0; a < D; a++) o[(n++) * WAVE] = tx[k][a];
        }
    }
    __device__ __forceinline__ double value() const { return 0.5 * ((double)D * ld.value() + accq); }
    static __device__ __forceinline__ void trans(double dt, double p1, double p2, ScalTrans& tr) {
        if constexpr (HAS_P2) ou_trans(dt, exp(p1), exp(p2), tr);     // nllk_ou_ssm.hpp:121-124
        else bm_trans(dt, exp(p1), tr);                               // nllk_bm_ssm.hpp:106-108
    }
};

template <int MODEL, int D, int KC>
struct CvModel;
template <int D, int KC>
struct CvModel<M_CTCRW, D, KC> { typedef CvCtcrw<D, KC> Lane; typedef CtcrwTrans Trans; };
template <int D, int KC>
struct CvModel<M_OU_SSM, D, KC> { typedef CvScal<D, KC, true> Lane; typedef ScalTrans Trans; };
template <int D, int KC>
struct CvModel<M_BM_SSM, D, KC> { typedef CvScal<D, KC, false> Lane; typedef ScalTrans Trans; };

int colvar_nstate(int model, int d, int kc) {
    return model == M_CTCRW ? 2 * d + 5 + kc * (3 + 2 * d) : d + 2 + kc * (1 + d);
}

// ---- the kernel ------------------------------------------------------------------------------------------------------------
// One WORKGROUP per (64-track group, time window); its four waves are the four PARTS (the columns dealt to them); the rows
// are staged ONCE through LDS for all four: wave w loads the channels c = w, w + 4, ... of the row two ahead into registers,
// stores them to the ring slot the row after next will be read from, and one barrier per row keeps the four in step.  HBM is
// read once per row (8 (1 + d + K) bytes), a wave holds a quarter of the row in flight instead of all of it, and the linear
// predictors come from a run-time loop over LDS (no register array as wide as the design matrix).
// accumulators of a part: [value | column 0 .. CV_KC-1 | mu_1 .. mu_d]
constexpr int CV_CMAX = 1 + 2 + DRIFT_KMAX;                   // channels of a staged row: dt, y, the streamed columns
constexpr int CV_LD = (CV_CMAX + WG_WAVES - 1) / WG_WAVES;   // channels a wave loads per row

template <int MODEL, int D>
__global__ __launch_bounds__(WG_WAVES * WAVE, 1) void iso_colvar_kernel(const IsoArgs A, const CvPart* parts) {
    constexpr int KC = CV_KC;
    typedef typename CvModel<MODEL, D, KC>::Lane Lane;
    typedef typename CvModel<MODEL, D, KC>::Trans Trans;
    constexpr int SD = Lane::SD;
    __shared__ double raw[2][CV_CMAX * WAVE];
    __shared__ double coef[DRIFT_KMAX][2];
    if (blockIdx.x == 0 && threadIdx.x == 0 && A.chk_out) *A.chk_out = 0.0;
    const int lane = threadIdx.x & 63, part = threadIdx.x >> 6;
    const TileView& tv = A.tv;
    const int G = tv.n_groups;
    const int g = blockIdx.x % G, chunk = blockIdx.x / G;      // (groups are sorted longest first: the long ones start first)
    const int C = tv.C, c_obs = tv.c_obs, K = A.drift_k, c_col = A.c_col;
    constexpr int nacc = 1 + CV_KC + D;
    if (threadIdx.x == 0) {
#pragma unroll
        for (int k = 0; k < DRIFT_KMAX; k++) { coef[k][0] = A.coefA[k]; coef[k][1] = A.coefB[k]; }
    }
    CvPart P;                                                  // (wave-uniform: scalar loads)
    P.n_col = parts[part].n_col; P.n0 = parts[part].n0; P.n01 = parts[part].n01; P.with_mu = parts[part].with_mu;
    if (!A.part_mask[0]) { P.n_col = 0; P.n0 = 0; P.n01 = 0; P.with_mu = 0; }      // value only: part 0 runs the primal filter alone
    const bool active = part == 0 || P.n_col > 0 || P.with_mu; // (a part without work still stages its share of the rows)
    int chan[KC];
    bool ones[KC];
#pragma unroll
    for (int k = 0; k < KC; k++) {
        const int ch = parts[part].chan[k];
        ones[k] = !(k < P.n_col && ch >= 0);
        chan[k] = ones[k] ? c_col : ch;
    }
    const double* base = tv.tiles + tv.group_off[g] + lane;
    const int L = tv.group_len[g];
    const int ns = tv.lane_nsteps[g * WAVE + lane];
    int s_begin, s_acc, s_end;
    window_bounds(L, A.n_chunks, A.window, 0, chunk, s_begin, s_acc, s_end, 0);
    const int pc = part * A.n_chunks + chunk;

    double set0[CV_LD], set1[CV_LD];
    auto ld = [&](double (&dst)[CV_LD], int s) {               // this wave's channels of row s: HBM -> registers
        const double* p = base + (int64_t)s * C * WAVE;
#pragma unroll
        for (int i = 0; i < CV_LD; i++) {
            const int c = part + WG_WAVES * i;
            dst[i] = 0.0;
            if (c < C) dst[i] = p[c * WAVE];
        }
    };
    auto st = [&](const double (&src)[CV_LD], int slot) {      // registers -> the ring
#pragma unroll
        for (int i = 0; i < CV_LD; i++) {
            const int c = part + WG_WAVES * i;
            if (c < C) raw[slot][c * WAVE + lane] = src[i];
        }
    };
    ld(set0, s_begin);
    ld(set1, s_begin + 1);
    Lane S;
    {
        double a0[SD];
        if (s_begin == 0) {
#pragma unroll
            for (int c = 0; c < SD; c++) a0[c] = tv.a0[((int64_t)g * SD + c) * WAVE + lane];
        } else {
#pragma unroll
            for (int a = 0; a < D; a++) {                      // a window past the first starts from its first observation
                const double y0 = base[((int64_t)s_begin * C + c_obs + a) * WAVE];
                if constexpr (MODEL == M_CTCRW) { a0[2 * a] = (y0 == y0) ? y0 : 0.0; a0[2 * a + 1] = 0.0; }
                else a0[a] = (y0 == y0) ? y0 : 0.0;
            }
        }
        S.init(a0, A.p0);
    }
    double mu[D];
#pragma unroll
    for (int a = 0; a < D; a++) mu[a] = A.mu[a];
    const double h = A.h;
    auto row = [&](int slot, int s) {
        if (!active) return;
        if (s == s_acc && s_acc > s_begin) {
            S.dump_to(A.bnd + (((int64_t)pc * G + g) * 2 + 0) * A.bnd_stride * WAVE + lane);
            S.reset_acc();
        }
        if (s < ns) {
            const double* r = &raw[slot][lane];
            double p1 = A.cv_eta0[0], p2 = A.cv_eta0[1];
#pragma unroll 4
            for (int k = 0; k < K; k++) {
                const double xk = r[(c_col + k) * WAVE];
                p1 = fma(coef[k][0], xk, p1);
                if (MODEL != M_BM_SSM) p2 = fma(coef[k][1], xk, p2);
            }
            double dt = tv.dt_all;
            if (c_obs) dt = r[0];
            double y[D];
#pragma unroll
            for (int a = 0; a < D; a++) y[a] = r[(c_obs + a) * WAVE];
            Trans tr;
            Lane::trans(dt, p1, p2, tr);
            double X[KC];
#pragma unroll
            for (int k = 0; k < KC; k++) { const double xl = r[chan[k] * WAVE]; X[k] = ones[k] ? 1.0 : xl; }
            S.step(P, tr, h, mu, y, is_na(y[0], A.any_nan), X);
        }
    };
    st(set0, 0);                                               // (window bounds are multiples of WIN_ALIGN: row s lives in slot s & 1)
    ld(set0, s_begin + 2);
    __syncthreads();
    for (int s = s_begin; s < s_end; s += 2) {
        st(set1, 1);                                           // row s + 1 (slot 1 was last read for row s - 1, before the barrier)
        ld(set1, s + 3);
        row(0, s);
        __syncthreads();
        st(set0, 0);                                           // row s + 2
        ld(set0, s + 4);
        row(1, s + 1);
        __syncthreads();
    }
    if (active && A.n_chunks > 1 && chunk + 1 < A.n_chunks)
        S.dump_to(A.bnd + (((int64_t)pc * G + g) * 2 + 1) * A.bnd_stride * WAVE + lane);
    const bool empty = s_acc >= s_end || !active;
    {
        const double t = wave_sum(empty ? 0.0 : S.value());
        if (lane == 0) A.partials[((int64_t)pc * nacc + 0) * G + g] = t;
    }
#pragma unroll
    for (int k = 0; k < CV_KC; k++) {
        const double t = wave_sum(empty ? 0.0 : S.g[k]);
        if (lane == 0) A.partials[((int64_t)pc * nacc + 1 + k) * G + g] = t;
    }
#pragma unroll
    for (int a = 0; a < D; a++) {
        const double t = wave_sum(empty ? 0.0 : S.gmu[a]);
        if (lane == 0) A.partials[((int64_t)pc * nacc + 1 + CV_KC + a) * G + g] = t;
    }
}

// range of every streamed column over the rows of a group (create time: the window planner bounds the linear predictors with it)
__global__ __launch_bounds__(WG_WAVES * WAVE) void colvar_ranges_kernel(TileView tv, int c_col, int K, double* out /* [n_groups][K][2] */) {
    __shared__ double sh[WG_WAVES][2];
    const int g = blockIdx.x, lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const double* base = tv.tiles + tv.group_off[g] + lane;
    const int ns = tv.lane_nsteps[g * WAVE + lane];
    for (int k = 0; k < K; k++) {
        double lo = INFINITY, hi = -INFINITY;
        for (int s = wv; s < ns; s += WG_WAVES) {
            const double x = base[((int64_t)s * tv.C + c_col + k) * WAVE];
            lo = fmin(lo, x); hi = fmax(hi, x);
            if (x != x) { lo = -INFINITY; hi = INFINITY; }
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) { lo = fmin(lo, __shfl_xor(lo, o, 64)); hi = fmax(hi, __shfl_xor(hi, o, 64)); }
        if (lane == 0) { sh[wv][0] = lo; sh[wv][1] = hi; }
        __syncthreads();
        if (threadIdx.x == 0) {
            for (int w = 1; w < WG_WAVES; w++) { lo = fmin(lo, sh[w][0]); hi = fmax(hi, sh[w][1]); }
            out[((int64_t)g * K + k) * 2] = lo; out[((int64_t)g * K + k) * 2 + 1] = hi;
        }
        __syncthreads();
    }
}
hipError_t launch_colvar_ranges(const TileView& tv, int c_col, int K, double* out, hipStream_t s) {
    if (tv.n_groups == 0 || K == 0) return hipSuccess;
    hipLaunchKernelGGL(colvar_ranges_kernel, dim3(tv.n_groups), dim3(WG_WAVES * WAVE), 0, s, tv, c_col, K, out);
    return hipGetLastError();
}

// a.n_parts == WG_WAVES parts (one per wave of a workgroup), a.drift_k streamed columns (1 .. DRIFT_KMAX)
hipError_t launch_iso_colvar(int model, int d, const IsoArgs& a, const CvPart* parts, hipStream_t s) {
    if (a.n_parts != WG_WAVES || a.drift_k < 1 || a.drift_k > DRIFT_KMAX || a.tv.C > CV_CMAX) return hipErrorInvalidValue;
    dim3 grid(a.tv.n_groups * a.n_chunks), block(WG_WAVES * WAVE);
    if (grid.x == 0) return hipSuccess;
#define SSDE_CASE(M_, D_) if (model == M_ && d == D_) { hipLaunchKernelGGL((iso_colvar_kernel<M_, D_>), grid, block, 0, s, a, parts); return hipGetLastError(); }
    SSDE_CASE(M_CTCRW, 1) SSDE_CASE(M_CTCRW, 2) SSDE_CASE(M_OU_SSM, 1) SSDE_CASE(M_OU_SSM, 2) SSDE_CASE(M_BM_SSM, 1) SSDE_CASE(M_BM_SSM, 2)
#undef SSDE_CASE
    return hipErrorInvalidValue;
}

}  // namespace ssde
