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
// (T, Q, B, H) with respect to log tau (t = 1), log nu / kappa (t = 2) or log sigma_obs (t = 0) at THIS row's parameters.
// 3 + 2 d doubles of state and ~45 fp64 instructions per column and row (CTCRW, d = 2).
//
// Work split: the columns are dealt to the four waves of a workgroup (PARTS of at most CV_KC columns: a wave's register
// budget); the waves that carry columns recompute the primal filter (the gains: ~140 instructions) next to them.  The
// column loop is straight-line code: what a column feeds is a pair of 0/1 factors on its value, not a branch (measured: a
// uniform branch per column and type cost 2100 of 4600 cycles per row -- every column a basic block of its own, nothing to
// overlap).  The log sigma_obs direction and the drift-intercept direction are single slots behind one uniform branch each.
// One workgroup per (64-track group, time window); windows, warm-up and the verified hand-over as in k_iso.hip.
// Layout: the tiles of ssde_device.hpp with the design columns as further channels (as k_iso_drift.hip).
#include <type_traits>

#include "ssde_device.hpp"

namespace ssde {

// ---- the per-row linearisation and the column recursion, CTCRW ------------------------------------------------------
template <int D, int KC>
struct CvCtcrw {
    static constexpr int SD = 2 * D;
    static constexpr int NCOL = 3 + 2 * D;
    static constexpr int NSTATE = SD + 3 + 2 + (KC + 1) * NCOL;
    double x[D], v[D], p11, p12, p22;
    LogAcc ld;
    double accq;
    double mx, mv, gmu[D];
    double d11[KC + 1], d12[KC + 1], d22[KC + 1], tx[KC + 1][D], tv[KC + 1][D], g[KC + 1];      // slot KC: log sigma_obs

    __device__ __forceinline__ void init(const double* a0, const double* p0) {
#pragma unroll
        for (int a = 0; a < D; a++) { x[a] = a0[2 * a]; v[a] = a0[2 * a + 1]; gmu[a] = 0.0; }
        p11 = p0[0]; p12 = p0[1]; p22 = p0[2];
        ld.init(); accq = 0.0; mx = mv = 0.0;
#pragma unroll
        for (int k = 0; k <= KC; k++) {
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
        for (int k = 0; k <= KC; k++) g[k] = 0.0;
    }
    // One row: score y (unless NA), the column tangents, then the prediction over the row's interval (ctcrw_step's
    // arrangement: filtered-form covariance update, Joseph-form sensitivities; ssde_math.hpp).
    // X1[k] / X2[k]: the column's value if it feeds log tau / log nu, else 0.
    __device__ __forceinline__ void step(const CtcrwTrans& tr, double h, const double* mu, const double* y, bool na,
                                         const double* X1, const double* X2, bool with_sig, bool with_mu) {
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
        double u[D], mue[D];
        double su2 = 0.0;
#pragma unroll
        for (int a_ = 0; a_ < D; a_++) {
            const double ys = upd ? y[a_] : x[a_];
            u[a_] = ys - x[a_];
            su2 = fma(u[a_], u[a_], su2);
            mue[a_] = bm * mu[a_];
        }
        accq = fma(iF, su2, accq);
        const double gF = fma(-0.5 * iF * iF, su2, 0.5 * (double)D * iF);     // d nllk / d F of this row
        // ---- seeds ------------------------------------------------------------------------------------------------
        // log tau: T, B and Q move
        const double s1_11 = fma(tr.dt12x2, m, tr.dq11), s1_12 = fma(tr.dt12e, f22, fma(tr.de, m, tr.dq12)),
                     s1_22 = fma(tr.edex2, f22, tr.dq22);
        double s1_x[D], s1_v[D];
#pragma unroll
        for (int a_ = 0; a_ < D; a_++) {
            const double w = fma(kf2, u[a_], v[a_] - mue[a_]);     // d k u + d(T a + B mu): (dt12, de) (kf2 u + v - mu)
            s1_x[a_] = tr.dt12 * w; s1_v[a_] = tr.de * w;
        }
        // log nu: Q only (dQ = 2 Q)
        const double s2_11 = 2.0 * tr.q11, s2_12 = 2.0 * tr.q12, s2_22 = 2.0 * tr.q22;
        // the part of a tangent's step that does not depend on what it is a tangent of: returns the gain tangent
        auto lin = [&](int k, double dF, double& n11, double& n12, double& n22, double& dk1, double& dk2) {
            const double c11 = d11[k], c12 = d12[k], c22 = d22[k];
            double sud = 0.0;
#pragma unroll
            for (int a_ = 0; a_ < D; a_++) sud = fma(u[a_], tx[k][a_], sud);
            g[k] = fma(gF, dF, fma(-iF, sud, g[k]));
            const double w = fma(-kf2, c11, c12);
            const double g11 = a2 * c11, g12 = a * w, g22 = fma(-kf2, c12 + w, c22);
            const double dkf1 = c11 * aiF, dkf2 = w * iF;
            const double dm = fma(t12, g22, g12);
            n11 = fma(t12, g12 + dm, g11); n12 = e * dm; n22 = e2 * g22;
            dk1 = fma(t12, dkf2, dkf1); dk2 = e * dkf2;
        };
        // ---- columns (straight-line: the two seed sets are weighted by the column's 0/1 factors) -----------------------
#pragma unroll
        for (int k = 0; k < KC; k++) {
            double n11, n12, n22, dk1, dk2;
            lin(k, d11[k], n11, n12, n22, dk1, dk2);
            const double x1 = X1[k], x2 = X2[k];
            d11[k] = fma(x2, s2_11, fma(x1, s1_11, n11));
            d12[k] = fma(x2, s2_12, fma(x1, s1_12, n12));
            d22[k] = fma(x2, s2_22, fma(x1, s1_22, n22));
#pragma unroll
            for (int a_ = 0; a_ < D; a_++) {
                const double txk = tx[k][a_], tvk = tv[k][a_];
                tx[k][a_] = fma(x1, s1_x[a_], fma(dk1, u[a_], fma(t12, tvk, c1 * txk)));
                tv[k][a_] = fma(x1, s1_v[a_], fma(dk2, u[a_], fma(e, tvk, -k2 * txk)));
            }
        }
        if (with_sig) {
            // log sigma_obs: dh = 2 h enters F, the filtered covariance (k k' dh) and the gain (-k dh / F)
            const double h2 = 2.0 * h;
            double n11, n12, n22, dk1, dk2;
            lin(KC, d11[KC] + h2, n11, n12, n22, dk1, dk2);
            const double q1 = kf1 * h2, q2 = kf2 * h2;
            const double g11 = kf1 * q1, g12 = kf2 * q1, g22 = kf2 * q2;
            const double dkf1 = -q1 * iF, dkf2 = -q2 * iF;
            const double dm = fma(t12, g22, g12);
            d11[KC] = n11 + fma(t12, g12 + dm, g11); d12[KC] = fma(e, dm, n12); d22[KC] = fma(e2, g22, n22);
            dk1 += fma(t12, dkf2, dkf1); dk2 = fma(e, dkf2, dk2);
#pragma unroll
            for (int a_ = 0; a_ < D; a_++) {
                const double txk = tx[KC][a_], tvk = tv[KC][a_];
                tx[KC][a_] = fma(dk1, u[a_], fma(t12, tvk, c1 * txk));
                tv[KC][a_] = fma(dk2, u[a_], fma(e, tvk, -k2 * txk));
            }
        }
        if (with_mu) {                                             // d / d mu_a: one data-independent chain for every dimension
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
        for (int k = 0; k <= KC; k++) {
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
    static constexpr int NSTATE = SD + 1 + 1 + (KC + 1) * NCOL;
    double x[D], p;
    LogAcc ld;
    double accq;
    double mx, gmu[D];
    double dp[KC + 1], tx[KC + 1][D], g[KC + 1];               // slot KC: log sigma_obs

    __device__ __forceinline__ void init(const double* a0, const double* p0) {
#pragma unroll
        for (int a = 0; a < D; a++) { x[a] = a0[a]; gmu[a] = 0.0; }
        p = p0[0];
        ld.init(); accq = 0.0; mx = 0.0;
#pragma unroll
        for (int k = 0; k <= KC; k++) {
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
        for (int k = 0; k <= KC; k++) g[k] = 0.0;
    }
    // scal_cov_step + scal_mean_step (ssde_math.hpp) with the direction loops replaced by the column loop
    __device__ __forceinline__ void step(const ScalTrans& tr, double h, const double* mu, const double* y, bool na,
                                         const double* X1, const double* X2, bool with_sig, bool with_mu) {
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
        const double s1_k = HAS_P2 ? dt_ * b : 0.0;                 // log tau (OU) / log sigma (BM)
        const double s1_p = HAS_P2 ? fma(2.0 * dt_, cp, tr.dq) : tr.dq;
        double s1_x[D];
#pragma unroll
        for (int a_ = 0; a_ < D; a_++) s1_x[a_] = HAS_P2 ? fma(s1_k, u[a_], fma(tr.dt_, x[a_], tr.db * mu[a_])) : 0.0;
        const double s2_p = tr.q;                                   // log kappa (OU)
        auto lin = [&](int k_, double dF, double& np_, double& dk) {
            double sud = 0.0;
#pragma unroll
            for (int a_ = 0; a_ < D; a_++) sud = fma(u[a_], tx[k_][a_], sud);
            g[k_] = fma(gF, dF, fma(-iF, sud, g[k_]));
            dk = ca * dp[k_]; np_ = tca * dp[k_];
        };
#pragma unroll
        for (int k_ = 0; k_ < KC; k_++) {
            double np_, dk;
            lin(k_, dp[k_], np_, dk);
            const double x1 = X1[k_], x2 = X2[k_];
            dp[k_] = HAS_P2 ? fma(x2, s2_p, fma(x1, s1_p, np_)) : fma(x1, s1_p, np_);
#pragma unroll
            for (int a_ = 0; a_ < D; a_++) {
                const double nx = fma(dk, u[a_], c * tx[k_][a_]);
                tx[k_][a_] = HAS_P2 ? fma(x1, s1_x[a_], nx) : nx;
            }
        }
        if (with_sig) {
            const double h2 = 2.0 * h, bh = b * h2;
            double np_, dk;
            lin(KC, dp[KC] + h2, np_, dk);
            dk = fma(-tiF, bh, dk);
            dp[KC] = fma(k * t, bh, np_);
#pragma unroll
            for (int a_ = 0; a_ < D; a_++) tx[KC][a_] = fma(dk, u[a_], c * tx[KC][a_]);
        }
        if (with_mu) {
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
        for (int k = 0; k <= KC; k++) {
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

// components of a part's hand-over dump with kc column slots
int colvar_nstate(int model, int d, int kc) {
    return model == M_CTCRW ? 2 * d + 5 + (kc + 1) * (3 + 2 * d) : d + 2 + (kc + 1) * (1 + d);
}

// ---- the kernel ------------------------------------------------------------------------------------------------------------
// One WORKGROUP per (64-track group, time window); its four waves are the four PARTS (the columns dealt to them), in step row
// by row (one barrier per row), and they share three things through LDS:
//   * the rows, staged ONCE for all four: wave w loads the channels c = w, w + 4, ... of a row into registers two rows ahead
//     and stores them to the ring slot the next row is read from -- HBM is read once per row (8 (1 + d + K) bytes), and a
//     wave holds a quarter of a row in flight instead of all of it;
//   * the linear predictors p1 = log tau_i, p2 = log nu_i: every wave sums ITS channels' terms while they are still in
//     registers (coefficients in scalar registers) and stores two partial sums per row;
//   * the row's transition: the LAST wave adds the partial sums of the NEXT row, takes the exp's and builds T, Q, B and
//     their log tau derivatives once (makeT/Q/B_ctcrw: nllk_ctcrw.hpp:45-91 through ctcrw_trans) into a two-slot ring; the
//     engine deals that wave fewer columns.
// accumulators of a part: [value | column 0 .. CV_KC-1 | mu_1 .. mu_d | log sigma_obs]
constexpr int CV_LD = (1 + 2 + DRIFT_KMAX + WG_WAVES - 1) / WG_WAVES;   // channels a wave loads per row (dt, y, the streamed columns)
constexpr int CV_CMAX = CV_LD * WG_WAVES;                               // channels of a staged row
constexpr int CV_PRODUCER = WG_WAVES - 1;                               // the wave that builds the transitions

__device__ __forceinline__ double uniform_double(double x) {   // a wave-uniform value into scalar registers
    const int lo = __builtin_amdgcn_readfirstlane(__double2loint(x)), hi = __builtin_amdgcn_readfirstlane(__double2hiint(x));
    return __hiloint2double(hi, lo);
}

template <int MODEL> struct CvTransIO;
template <> struct CvTransIO<M_CTCRW> {
    static constexpr int N = 12;
    static __device__ __forceinline__ void put(double* o, const CtcrwTrans& t) {      // o[j * WAVE]
        o[0 * WAVE] = t.e; o[1 * WAVE] = t.t12; o[2 * WAVE] = t.b1; o[3 * WAVE] = t.b2; o[4 * WAVE] = t.q11; o[5 * WAVE] = t.q12;
        o[6 * WAVE] = t.q22; o[7 * WAVE] = t.de; o[8 * WAVE] = t.dt12; o[9 * WAVE] = t.dq11; o[10 * WAVE] = t.dq12; o[11 * WAVE] = t.dq22;
    }
    static __device__ __forceinline__ void get(const double* o, CtcrwTrans& t) {
        t.e = o[0 * WAVE]; t.t12 = o[1 * WAVE]; t.b1 = o[2 * WAVE]; t.b2 = o[3 * WAVE]; t.q11 = o[4 * WAVE]; t.q12 = o[5 * WAVE];
        t.q22 = o[6 * WAVE]; t.de = o[7 * WAVE]; t.dt12 = o[8 * WAVE]; t.dq11 = o[9 * WAVE]; t.dq12 = o[10 * WAVE]; t.dq22 = o[11 * WAVE];
        t.e2 = t.e * t.e; t.dt12x2 = 2.0 * t.dt12; t.dt12e = t.dt12 * t.e; t.edex2 = 2.0 * t.e * t.de;      // as ctcrw_trans forms them
    }
};
template <int MODEL> struct CvTransIO {                       // OU_SSM, BM_SSM
    static constexpr int N = 6;
    static __device__ __forceinline__ void put(double* o, const ScalTrans& t) {
        o[0 * WAVE] = t.t; o[1 * WAVE] = t.b; o[2 * WAVE] = t.q; o[3 * WAVE] = t.dt_; o[4 * WAVE] = t.db; o[5 * WAVE] = t.dq;
    }
    static __device__ __forceinline__ void get(const double* o, ScalTrans& t) {
        t.t = o[0 * WAVE]; t.b = o[1 * WAVE]; t.q = o[2 * WAVE]; t.dt_ = o[3 * WAVE]; t.db = o[4 * WAVE]; t.dq = o[5 * WAVE];
    }
};

// KC: column slots per wave (the widest part's count, rounded up to even; the engine picks the instantiation)
template <int MODEL, int D, int KC>
__global__ __launch_bounds__(WG_WAVES * WAVE, 1) void iso_colvar_kernel(const IsoArgs A, const CvPart* parts) {
    typedef typename CvModel<MODEL, D, KC>::Lane Lane;
    typedef typename CvModel<MODEL, D, KC>::Trans Trans;
    typedef CvTransIO<MODEL> TIO;
    constexpr int SD = Lane::SD;
    __shared__ double raw[2][CV_CMAX * WAVE];                  // the staged rows
    __shared__ double eta[2][(2 * WG_WAVES + 1) * WAVE];       // per row: the four waves' partial sums of p1, p2, and the interval
    __shared__ double trs[2][TIO::N * WAVE];                   // per row: the transition
    __shared__ double coef[DRIFT_KMAX][2];
    if (blockIdx.x == 0 && threadIdx.x == 0 && A.chk_out) *A.chk_out = 0.0;
    const int lane = threadIdx.x & 63, part = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // (a scalar: the part tables are read with scalar loads)
    const TileView& tv = A.tv;
    const int G = tv.n_groups;
    const int g = blockIdx.x % G, chunk = blockIdx.x / G;      // (groups are sorted longest first: the long ones start first)
    const int C = tv.C, c_obs = tv.c_obs, K = A.drift_k, c_col = A.c_col;
    constexpr int nacc = 2 + CV_KC + D;
    if (threadIdx.x == 0) {
#pragma unroll
        for (int k = 0; k < DRIFT_KMAX; k++) { coef[k][0] = A.coefA[k]; coef[k][1] = A.coefB[k]; }
    }
    __syncthreads();
    // the channels this wave stages (past the last one: a design column again, with coefficient 0) and their coefficients
    int lch[CV_LD];
    unsigned col_bits = 0;                                     // bit i: the wave's i-th channel is a design column
    double cA[CV_LD], cB[CV_LD];
#pragma unroll
    for (int i = 0; i < CV_LD; i++) {
        const int c = part + WG_WAVES * i, k = c - c_col;
        const bool on = k >= 0 && k < K;
        if (on) col_bits |= 1u << i;
        lch[i] = c < C ? c : c_col;
        cA[i] = uniform_double(on ? coef[on ? k : 0][0] : 0.0);
        cB[i] = uniform_double(on ? coef[on ? k : 0][1] : 0.0);
    }
    const bool grad = A.part_mask[0] != 0;                     // (0: the value only -- part 0 runs the primal filter alone)
    const int n_col = grad ? parts[part].n_col : 0;
    const bool with_mu = grad && parts[part].with_mu, with_sig = grad && parts[part].with_sig;
    const bool active = part == 0 || n_col > 0 || with_mu || with_sig;     // (a part without work still stages its share of the rows)
    int chan[KC];
    unsigned ones_bits = 0, t1_bits = 0, t2_bits = 0;           // per slot: a column of ones / feeds par[d] / feeds par[d + 1]
#pragma unroll
    for (int k = 0; k < KC; k++) {
        const bool on = k < n_col;
        const int ch = on ? parts[part].chan[k] : -2, ty = on ? parts[part].type[k] : 0;
        chan[k] = ch >= 0 ? ch : c_col;                        // (an unused slot reads a design column and weighs it with 0)
        if (ch == -1) ones_bits |= 1u << k;
        if (ty == 1) t1_bits |= 1u << k;
        if (ty == 2) t2_bits |= 1u << k;
    }
    const double* base = tv.tiles + tv.group_off[g] + lane;
    const int L = tv.group_len[g];
    const int ns = tv.lane_nsteps[g * WAVE + lane];
    int s_begin, s_acc, s_end;
    window_bounds(L, A.n_chunks, A.window, 0, chunk, s_begin, s_acc, s_end, 0);
    const int pc = part * A.n_chunks + chunk;

    double setA[CV_LD], setB[CV_LD];
    auto ld = [&](double (&dst)[CV_LD], int s) {               // this wave's channels of row s: HBM -> registers
        const double* p = base + (int64_t)s * C * WAVE;
#pragma unroll
        for (int i = 0; i < CV_LD; i++) dst[i] = p[lch[i] * WAVE];
    };
    auto st_raw = [&](const double (&src)[CV_LD], int slot) {  // registers -> the ring of rows (a channel past the last: written, never read)
#pragma unroll
        for (int i = 0; i < CV_LD; i++) raw[slot][(part + WG_WAVES * i) * WAVE + lane] = src[i];
    };
    auto st_eta = [&](const double (&src)[CV_LD], int slot) {  // this wave's terms of the row's linear predictors
        double pa = 0.0, pb = 0.0;
#pragma unroll
        for (int i = 0; i < CV_LD; i++) {
            const double xs = ((col_bits >> i) & 1u) ? src[i] : 0.0;      // (an observation may be NaN: 0 * NaN is not 0)
            pa = fma(cA[i], xs, pa);
            if (MODEL != M_BM_SSM) pb = fma(cB[i], xs, pb);
        }
        eta[slot][(2 * part) * WAVE + lane] = pa;
        eta[slot][(2 * part + 1) * WAVE + lane] = pb;
        if (part == 0) eta[slot][(2 * WG_WAVES) * WAVE + lane] = src[0];      // channel 0: the interval after the row (if the tiles hold it)
    };
    auto produce = [&](int slot) {                             // the last wave: the transition of the row whose sums sit in eta[slot]
        if (part != CV_PRODUCER) return;
        const double* e_ = &eta[slot][lane];
        double p1 = A.cv_eta0[0], p2 = A.cv_eta0[1];
#pragma unroll
        for (int w = 0; w < WG_WAVES; w++) { p1 += e_[(2 * w) * WAVE]; p2 += e_[(2 * w + 1) * WAVE]; }
        const double dtc = e_[(2 * WG_WAVES) * WAVE];
        const double dt = c_obs ? dtc : tv.dt_all;
        Trans tr;
        Lane::trans(dt, p1, p2, tr);
        TIO::put(&trs[slot][lane], tr);
    };
    ld(setA, s_begin);
    ld(setB, s_begin + 1);
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
            double y[D];
#pragma unroll
            for (int a = 0; a < D; a++) y[a] = r[(c_obs + a) * WAVE];
            Trans tr;
            TIO::get(&trs[slot][lane], tr);
            double X1[KC], X2[KC];
#pragma unroll
            for (int k = 0; k < KC; k++) {
                const double xl = r[chan[k] * WAVE];
                const double xk = ((ones_bits >> k) & 1u) ? 1.0 : xl;
                X1[k] = ((t1_bits >> k) & 1u) ? xk : 0.0; X2[k] = ((t2_bits >> k) & 1u) ? xk : 0.0;
            }
            S.step(tr, h, mu, y, is_na(y[0], A.any_nan), X1, X2, with_sig, with_mu);
        }
    };
#ifdef SSDE_CV_CLOCK
    // (tuning build: where a wave's cycles go -- staging, the transition, the row, the barrier)
    long long ck[4] = {0, 0, 0, 0};
    long long t_ = __builtin_amdgcn_s_memtime();
#define SSDE_CK(i) { const long long n_ = __builtin_amdgcn_s_memtime(); ck[i] += n_ - t_; t_ = n_; }
#else
#define SSDE_CK(i)
#endif
    // (window bounds are multiples of WIN_ALIGN: row s lives in slot s & 1 of every ring)
    st_eta(setA, 0);                                           // row s_begin
    __syncthreads();
    st_raw(setA, 0);
    st_eta(setB, 1);                                           // row s_begin + 1
    ld(setA, s_begin + 2);
    produce(0);
    __syncthreads();
    for (int s = s_begin; s < s_end; s += 2) {
        // setB holds row s + 1, setA row s + 2
        st_raw(setB, 1);                                       // (slot 1 was last read for row s - 1, before the barrier)
        st_eta(setA, 0);                                       // row s + 2 (slot 0 was last read by the producer for row s, before the barrier)
        ld(setB, s + 3);                                       // (issued AFTER the stores: they wait for loads one and two rows old, not for these)
        SSDE_CK(0)
        produce(1);                                            // row s + 1
        SSDE_CK(1)
        row(0, s);
        SSDE_CK(2)
        __syncthreads();
        SSDE_CK(3)
        st_raw(setA, 0);                                       // row s + 2
        st_eta(setB, 1);                                       // row s + 3
        ld(setA, s + 4);
        SSDE_CK(0)
        produce(0);                                            // row s + 2
        SSDE_CK(1)
        row(1, s + 1);
        SSDE_CK(2)
        __syncthreads();
        SSDE_CK(3)
    }
#ifdef SSDE_CV_CLOCK
    if (A.wave_clock && lane == 0) {
        double* o = A.wave_clock + 4 * ((int64_t)blockIdx.x * WG_WAVES + part);
        for (int i = 0; i < 4; i++) o[i] = (double)ck[i] / (double)(s_end - s_begin);
    }
#endif
    if (active && A.n_chunks > 1 && chunk + 1 < A.n_chunks)
        S.dump_to(A.bnd + (((int64_t)pc * G + g) * 2 + 1) * A.bnd_stride * WAVE + lane);
    const bool empty = s_acc >= s_end || !active;
    {
        const double t = wave_sum(empty ? 0.0 : S.value());
        if (lane == 0) A.partials[((int64_t)pc * nacc + 0) * G + g] = t;
    }
#pragma unroll
    for (int k = 0; k < CV_KC; k++) {
        const double t = wave_sum((empty || k >= KC) ? 0.0 : S.g[k < KC ? k : 0]);
        if (lane == 0) A.partials[((int64_t)pc * nacc + 1 + k) * G + g] = t;
    }
#pragma unroll
    for (int a = 0; a < D; a++) {
        const double t = wave_sum(empty ? 0.0 : S.gmu[a]);
        if (lane == 0) A.partials[((int64_t)pc * nacc + 1 + CV_KC + a) * G + g] = t;
    }
    {
        const double t = wave_sum(empty ? 0.0 : S.g[KC]);
        if (lane == 0) A.partials[((int64_t)pc * nacc + 1 + CV_KC + D) * G + g] = t;
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

// a.n_parts == WG_WAVES parts (one per wave of a workgroup), a.drift_k streamed columns (1 .. DRIFT_KMAX), kc: the widest
// part's column count
template <int MODEL, int D>
static hipError_t launch_cv(const IsoArgs& a, const CvPart* parts, int kc, dim3 grid, dim3 block, hipStream_t s) {
    if (kc <= 2) hipLaunchKernelGGL((iso_colvar_kernel<MODEL, D, 2>), grid, block, 0, s, a, parts);
    else if (kc <= 4) hipLaunchKernelGGL((iso_colvar_kernel<MODEL, D, 4>), grid, block, 0, s, a, parts);
    else if (kc <= 6) hipLaunchKernelGGL((iso_colvar_kernel<MODEL, D, 6>), grid, block, 0, s, a, parts);
    else hipLaunchKernelGGL((iso_colvar_kernel<MODEL, D, 8>), grid, block, 0, s, a, parts);
    return hipGetLastError();
}
hipError_t launch_iso_colvar(int model, int d, const IsoArgs& a, const CvPart* parts, int kc, hipStream_t s) {
    if (a.n_parts != WG_WAVES || a.drift_k < 1 || a.drift_k > DRIFT_KMAX || a.tv.C > CV_CMAX || kc < 0 || kc > CV_KC) return hipErrorInvalidValue;
    dim3 grid(a.tv.n_groups * a.n_chunks), block(WG_WAVES * WAVE);
    if (grid.x == 0) return hipSuccess;
#define SSDE_CASE(M_, D_) if (model == M_ && d == D_) return launch_cv<M_, D_>(a, parts, kc, grid, block, s);
    SSDE_CASE(M_CTCRW, 1) SSDE_CASE(M_CTCRW, 2) SSDE_CASE(M_OU_SSM, 1) SSDE_CASE(M_OU_SSM, 2) SSDE_CASE(M_BM_SSM, 1) SSDE_CASE(M_BM_SSM, 2)
#undef SSDE_CASE
    return hipErrorInvalidValue;
}

}  // namespace ssde
