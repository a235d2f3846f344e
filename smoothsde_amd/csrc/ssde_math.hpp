// ssde_math.hpp -- per-lane arithmetic of the MI355X nllk engine.
//
// One wavefront lane owns one track.  This header holds the arithmetic a lane performs per
// row, written as __host__ __device__ inline functions so that the same source is
//   * inlined into the HIP kernels (ssde_kernels.hip, gfx950), and
//   * compiled by g++ into the test-only harness tests/hostsim/ (the CPU suite checks the
//     kernel arithmetic against the oracle without a GPU; the harness is not shipped and the
//     product library has no CPU evaluation path).
//
// What is restated (reference = /root/reference/src/nllk):
//   isotropic CTCRW Kalman step   nllk_ctcrw.hpp:195-247 with makeT/Q/B (:45-91)
//   isotropic OU / BM Kalman step nllk_ou_ssm.hpp:163-213 (makeT/B/Q :30-69),
//                                 nllk_bm_ssm.hpp:127-175 (makeQ :28-36)
//   direct BM / OU densities      nllk_sde.hpp:77-84 + tr_dens.hpp:32-37, 45-52
// "Isotropic" = H = sigma_obs^2 I (makeH_*), P0 block-identical across dimensions (the
// default diag(1,10,...) / diag(10,...) of R/sde.R:554,584): then the covariance stays
// block-identical and 3 (CTCRW) or 1 (OU/BM) scalars describe it.
//
// The gradient is NOT in the reference (it is a CppAD tape sweep, R/sde.R:656-658); here it
// is hand-derived forward sensitivities carried next to the state in registers.
#ifndef SSDE_MATH_HPP
#define SSDE_MATH_HPP

#include <math.h>
#include <stdint.h>
#include <string.h>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define SSDE_HD __host__ __device__ __forceinline__
#else
#define SSDE_HD inline
#endif

namespace ssde {

// gradient-direction mask bits of the constant-coefficient register kernels
enum { DIR_SIG = 1, DIR_MU = 2, DIR_P1 = 4, DIR_P2 = 8 };

// model codes (== SSDE_MODEL_* of include/ssde.h)
enum { M_BM = 0, M_OU = 1, M_BM_SSM = 2, M_OU_SSM = 3, M_CTCRW = 4, M_BM_T = 5, M_ESEAL = 6, M_CIR = 7 };

// R_IsNA / any-NaN test on the bit pattern (Q5)
SSDE_HD bool is_na(double x, int any_nan) {
    if (!(x != x)) return false;
    if (any_nan) return true;
    uint64_t b;
#if defined(__HIP_DEVICE_COMPILE__)
    b = (uint64_t)__double_as_longlong(x);
#else
    memcpy(&b, &x, 8);
#endif
    return (uint32_t)(b & 0xffffffffu) == 1954u;
}

// 1/x for the innovation variance: hardware reciprocal seed + two Newton steps (full fp64
// accuracy for finite normal x; the compiler's IEEE division sequence is ~2x longer).
SSDE_HD double rcp(double x) {
#if defined(__HIP_DEVICE_COMPILE__)
    double r = __builtin_amdgcn_rcp(x);
    r = fma(fma(-x, r, 1.0), r, r);
    r = fma(fma(-x, r, 1.0), r, r);
    return r;
#else
    return 1.0 / x;
#endif
}

// Running sum of log|F| kept as mantissa * 2^exponent: one multiply and an exponent
// extraction per row instead of a software fp64 log per row; a single log at the end.
struct LogAcc {
    double m;
    int e;
    SSDE_HD void init() { m = 1.0; e = 0; }
    SSDE_HD void mul(double f) {
        int ex;
        m = frexp(m * fabs(f), &ex);
        e += ex;
    }
    SSDE_HD double value() const { return (double)e * 0.6931471805599453094 + log(m); }
};

// ---------------------------------------------------------------------------------------
// CTCRW, isotropic.  State per dimension (x, v); covariance (p11, p12, p22) shared.
// ---------------------------------------------------------------------------------------
struct CtcrwTrans {
    double e, t12, b1, b2, q11, q12, q22;  // makeT/B/Q_ctcrw entries for one interval
    double de, dt12, dq11, dq12, dq22;     // d/d(log tau); db1 = -dt12, db2 = -de
                                           // d/d(log nu): dq = 2 q, everything else 0
    // products the fused step (ctcrw_step) uses every row: on a regular grid they arrive with the kernel arguments
    // and stay in scalar registers
    double e2, dt12x2, dt12e, edex2;       // e^2, 2 dt12, dt12 e, 2 e de
};

// tau = exp(par_tau), beta = 1/tau, sigma = 2 nu / sqrt(pi tau)   (nllk_ctcrw.hpp:152-156)
SSDE_HD void ctcrw_trans(double dt, double tau, double beta, double sigma, CtcrwTrans& o) {
    const double e = exp(-beta * dt);
    const double e2 = e * e;  // exp(-2 beta dt)
    const double ome = 1.0 - e;
    const double s2 = sigma * sigma;
    const double A = s2 * tau;  // sigma^2 / beta = 4 nu^2 / pi: does not depend on tau
    o.e = e;
    o.t12 = ome * tau;          // (1 - e)/beta, makeT line 51
    o.b1 = dt - o.t12;          // makeB line 87
    o.b2 = ome;                 // makeB line 88
    const double G = dt - 2.0 * o.t12 + 0.5 * tau * (1.0 - e2);
    o.q11 = A * tau * G;                     // makeQ line 68-69
    o.q12 = 0.5 * A * tau * ome * ome;       // line 70: 1 - 2e + e^2 = (1-e)^2
    o.q22 = 0.5 * A * (1.0 - e2);            // line 72
    // derivatives w.r.t. log tau (A is constant; d tau = tau; d(beta dt) = -beta dt)
    const double edt = e * dt;
    o.de = e * beta * dt;
    o.dt12 = o.t12 - edt;
    const double dG = -2.0 * o.dt12 + 0.5 * tau * (1.0 - e2) - e2 * dt;
    o.dq11 = o.q11 + A * tau * dG;
    o.dq12 = o.q12 - A * ome * edt;
    o.dq22 = -A * e2 * beta * dt;
    o.e2 = e2; o.dt12x2 = 2.0 * o.dt12; o.dt12e = o.dt12 * e; o.edex2 = 2.0 * e * o.de;
}

// The step is split in two halves:
//   * the COVARIANCE half (P, F, K and their sensitivities) never looks at the observations --
//     only at the interval, the parameters and whether the row is missing;
//   * the MEAN half (state, residual, quadratic form and their sensitivities) consumes the
//     gains the covariance half produced.
// The general kernel runs both halves per lane.  On a regular time grid without missing rows
// the covariance half is identical for every track, so the engine evaluates it ONCE per
// evaluation (ssde_engine.hip: gain table) and the lanes run the mean half only.
constexpr int NDIRP = 3;  // covariance-affecting directions: 0 = log sigma_obs, 1 = par[n_dim], 2 = par[n_dim+1]

struct CtcrwGain {  // what one row's covariance half hands to the mean half
    double iF, k1, k2, bm;                       // bm = 0 only in CTCRW's detF <= 0 branch (Q3)
    double diF[NDIRP], dk1[NDIRP], dk2[NDIRP];
};

template <int MASK>
struct CtcrwCov {
    double p11, p12, p22;
    double d11[NDIRP], d12[NDIRP], d22[NDIRP];
    LogAcc ld;          // sum log|F|
    double gld[NDIRP];  // sum dF/F per direction
    double nupd;        // rows that entered the likelihood
    SSDE_HD void init(double p0_11, double p0_12, double p0_22) {
        p11 = p0_11; p12 = p0_12; p22 = p0_22;
        for (int j = 0; j < NDIRP; j++) d11[j] = d12[j] = d22[j] = gld[j] = 0.0;
        ld.init(); nupd = 0.0;
    }
    SSDE_HD void reset_acc() { ld.init(); nupd = 0.0; for (int j = 0; j < NDIRP; j++) gld[j] = 0.0; }
};

SSDE_HD constexpr int dir_bit(int j) { return j == 0 ? DIR_SIG : (j == 1 ? DIR_P1 : DIR_P2); }

// covariance half of one row (nllk_ctcrw.hpp:223-229, 236, 240-241 and their derivatives)
template <int D, int MASK>
SSDE_HD void ctcrw_cov_step(CtcrwCov<MASK>& C, const CtcrwTrans& tr, double h, bool na, CtcrwGain& G) {
    const double F = C.p11 + h;                                // F = Z P Z' + H (line 223), scalar per dimension
    const double detF = (D == 1) ? F : F * F;                  // det(): lines 16-19
    const bool upd = !na && !(detF <= 0.0);                    // lines 214, 226 (a NaN takes the update branch)
    const double iF = upd ? rcp(F) : 0.0;
    C.ld.mul(upd ? F : 1.0);                                   // log(detF) = D log|F| (line 234)
    C.nupd += upd ? 1.0 : 0.0;
    const double tp11 = C.p11 + tr.t12 * C.p12, tp12 = C.p12 + tr.t12 * C.p22;  // T P
    const double tp21 = tr.e * C.p12, tp22 = tr.e * C.p22;
    const double k1 = tp11 * iF, k2 = tp21 * iF;               // K = T P Z' F^-1 (line 236)
    G.iF = iF; G.k1 = k1; G.k2 = k2;
    G.bm = (na || upd) ? 1.0 : 0.0;                            // Q3: detF <= 0 predicts WITHOUT B mu (lines 226-228)
    for (int j = 0; j < NDIRP; j++) {
        if (!(MASK & dir_bit(j))) { G.diF[j] = G.dk1[j] = G.dk2[j] = 0.0; continue; }
        // seeds: derivatives of (h, e, t12, q11, q12, q22) in direction j
        const double dh = (j == 0) ? 2.0 * h : 0.0;
        const double de = (j == 1) ? tr.de : 0.0, dt12 = (j == 1) ? tr.dt12 : 0.0;
        const double dq11 = (j == 1) ? tr.dq11 : (j == 2 ? 2.0 * tr.q11 : 0.0);
        const double dq12 = (j == 1) ? tr.dq12 : (j == 2 ? 2.0 * tr.q12 : 0.0);
        const double dq22 = (j == 1) ? tr.dq22 : (j == 2 ? 2.0 * tr.q22 : 0.0);
        // (terms whose seed is structurally zero in this direction are left out at compile time: IEEE
        //  arithmetic does not let the compiler drop a multiplication by a literal 0.0 by itself)
        const double dF = (j == 0) ? C.d11[j] + dh : C.d11[j];
        const double diF = -iF * iF * dF;
        C.gld[j] += dF * iF;
        double dtp11 = C.d11[j] + tr.t12 * C.d12[j];
        double dtp12 = C.d12[j] + tr.t12 * C.d22[j];
        double dtp21 = tr.e * C.d12[j];
        double dtp22 = tr.e * C.d22[j];
        if (j == 1) {
            dtp11 += dt12 * C.p12; dtp12 += dt12 * C.p22;
            dtp21 += de * C.p12;   dtp22 += de * C.p22;
        }
        const double dk1 = dtp11 * iF + tp11 * diF;
        const double dk2 = dtp21 * iF + tp21 * diF;
        G.diF[j] = diF; G.dk1[j] = dk1; G.dk2[j] = dk2;
        double n11j = dtp11 * (1.0 - k1) - tp11 * dk1 + dtp12 * tr.t12;
        double n12j = -dtp11 * k2 - tp11 * dk2 + dtp12 * tr.e;
        double n22j = -dtp21 * k2 - tp21 * dk2 + dtp22 * tr.e;
        if (j == 1) { n11j += tp12 * dt12; n12j += tp12 * de; n22j += tp22 * de; }
        if (j != 0) { n11j += dq11; n12j += dq12; n22j += dq22; }
        C.d11[j] = n11j; C.d12[j] = n12j; C.d22[j] = n22j;
    }
    const double n11 = tp11 * (1.0 - k1) + tp12 * tr.t12 + tr.q11;  // P = T P (T - K Z)' + Q (lines 240-241)
    const double n12 = -tp11 * k2 + tp12 * tr.e + tr.q12;
    const double n22 = -tp21 * k2 + tp22 * tr.e + tr.q22;
    C.p11 = n11; C.p12 = n12; C.p22 = n22;
}

template <int D, int MASK>
struct CtcrwMean {
    double x[D], v[D];
    double tx[NDIRP][D], tv[NDIRP][D];   // sensitivities of (x, v) in the covariance-affecting directions
    double mx[D], mv[D];                 // d/d mu_a touches dimension a only
    double accq, gq[NDIRP], gmu[D];
    // a0 = (x_1, v_1, x_2, v_2, ...): one row of the reference's a0 matrix (R/sde.R:576-580)
    SSDE_HD void init(const double* a0) {
        for (int a = 0; a < D; a++) {
            x[a] = a0[2 * a]; v[a] = a0[2 * a + 1]; mx[a] = mv[a] = gmu[a] = 0.0;
            for (int j = 0; j < NDIRP; j++) tx[j][a] = tv[j][a] = 0.0;
        }
        accq = 0.0;
        for (int j = 0; j < NDIRP; j++) gq[j] = 0.0;
    }
    SSDE_HD void reset_acc() {
        accq = 0.0;
        for (int j = 0; j < NDIRP; j++) gq[j] = 0.0;
        for (int a = 0; a < D; a++) gmu[a] = 0.0;
    }
};

// mean half of one row (nllk_ctcrw.hpp:221, 231-234, 238 and their derivatives);
// scored = the row's observation enters the likelihood (not NA and detF > 0)
template <int D, int MASK>
SSDE_HD void ctcrw_mean_step(CtcrwMean<D, MASK>& M, const CtcrwTrans& tr, const CtcrwGain& G, const double* mu,
                             const double* y, bool scored) {
    double u[D];
    double su2 = 0.0;
    for (int a = 0; a < D; a++) { u[a] = scored ? y[a] - M.x[a] : 0.0; su2 += u[a] * u[a]; }  // line 221
    M.accq += G.iF * su2;                                                                    // u' F^-1 u
    for (int j = 0; j < NDIRP; j++) {
        if (!(MASK & dir_bit(j))) continue;
        const double de = (j == 1) ? tr.de : 0.0, dt12 = (j == 1) ? tr.dt12 : 0.0;
        double sud = 0.0;
        double du[D];
        for (int a = 0; a < D; a++) { du[a] = scored ? -M.tx[j][a] : 0.0; sud += u[a] * du[a]; }
        M.gq[j] += 0.5 * G.diF[j] * su2 + G.iF * sud;
        for (int a = 0; a < D; a++) {
            double nx = M.tx[j][a] + tr.t12 * M.tv[j][a] + G.dk1[j] * u[a] + G.k1 * du[a];
            double nv = tr.e * M.tv[j][a] + G.dk2[j] * u[a] + G.k2 * du[a];
            if (j == 1) {                      // only log tau moves T and B: d(B mu) = -(dt12, de) mu
                const double w = M.v[a] - G.bm * mu[a];
                nx += dt12 * w; nv += de * w;
            }
            M.tx[j][a] = nx; M.tv[j][a] = nv;
        }
    }
    if (MASK & DIR_MU) {
        for (int a = 0; a < D; a++) {
            const double du = scored ? -M.mx[a] : 0.0;
            M.gmu[a] += G.iF * u[a] * du;
            const double nx = M.mx[a] + tr.t12 * M.mv[a] + G.k1 * du + G.bm * tr.b1;
            const double nv = tr.e * M.mv[a] + G.k2 * du + G.bm * tr.b2;
            M.mx[a] = nx; M.mv[a] = nv;
        }
    }
    for (int a = 0; a < D; a++) {                              // a = T a + K u + B mu (line 238)
        const double nx = M.x[a] + tr.t12 * M.v[a] + G.k1 * u[a] + G.bm * tr.b1 * mu[a];
        const double nv = tr.e * M.v[a] + G.k2 * u[a] + G.bm * tr.b2 * mu[a];
        M.x[a] = nx; M.v[a] = nv;
    }
}

template <int D, int MASK>
struct CtcrwLane {
    CtcrwCov<MASK> C;
    CtcrwMean<D, MASK> M;
    SSDE_HD void init(const double* a0, double p0_11, double p0_12, double p0_22) {
        C.init(p0_11, p0_12, p0_22);
        M.init(a0);
    }
    // accumulators only (used when a time window starts scoring after its warm-up rows)
    SSDE_HD void reset_acc() { C.reset_acc(); M.reset_acc(); }
    static constexpr int NSTATE = 4 * (2 * D + 3) + 2 * D;
    // filter state + sensitivities, for the window hand-over check
    SSDE_HD void dump(double* o) const {
        int k = 0;
        for (int a = 0; a < D; a++) { o[k++] = M.x[a]; o[k++] = M.v[a]; }
        o[k++] = C.p11; o[k++] = C.p12; o[k++] = C.p22;
        for (int j = 0; j < NDIRP; j++) {
            const bool on = (MASK & dir_bit(j)) != 0;
            o[k++] = on ? C.d11[j] : 0.0; o[k++] = on ? C.d12[j] : 0.0; o[k++] = on ? C.d22[j] : 0.0;
            for (int a = 0; a < D; a++) { o[k++] = on ? M.tx[j][a] : 0.0; o[k++] = on ? M.tv[j][a] : 0.0; }
        }
        for (int a = 0; a < D; a++) { o[k++] = (MASK & DIR_MU) ? M.mx[a] : 0.0; o[k++] = (MASK & DIR_MU) ? M.mv[a] : 0.0; }
    }
};

// One row of a track: score y (unless NA), then propagate over the interval described by tr.
//   h = sigma_obs^2; mu[a] = mean velocity; na = obs(i,0) is NA (nllk_ctcrw.hpp:214)
// The general register kernel's step (k_iso.hip): both halves fused and arranged for fp64 issue, which is what bounds
// that kernel.  Same numbers as ctcrw_cov_step + ctcrw_mean_step up to rounding, but
//   * the covariance update is taken in the filtered form  P~ = P - P z z' P / F,  P' = T P~ T' + Q  with
//     P~11 = p11 h/F, P~12 = p12 h/F, P~22 = p22 - p12^2/F  (nllk_ctcrw.hpp:236-241 multiplied out): fewer products
//     than T P (T - K Z)', and no 1 - k1 cancellation once P >> H;
//   * its sensitivities in the Joseph form  dP~ = A dP A' + kk' dh,  A = I - k z' = [[a, 0], [-k2~, 1]], a = h/F,
//     then dP' = T dP~ T' + (dT P~ T' + T P~ dT') + dQ;
//   * a missing row needs no per-quantity select: iF = 0 makes k~, every gain and every gain sensitivity vanish and
//     a = 1, so the update formulas ARE the prediction formulas; only F, the 0/1 flag and y are selected;
//   * d/d mu_a of the state is the same (data-independent) number for every dimension a: one chain.
template <int D, int MASK>
SSDE_HD void ctcrw_step(CtcrwLane<D, MASK>& L, const CtcrwTrans& tr, double h, const double* mu, const double* y,
                        bool na) {
    CtcrwCov<MASK>& C = L.C;
    CtcrwMean<D, MASK>& M = L.M;
    const double F = C.p11 + h;                                // F = Z P Z' + H (line 223), scalar per dimension
    const double detF = (D == 1) ? F : F * F;                  // det(): lines 16-19
    const bool upd = !na && !(detF <= 0.0);                    // lines 214, 226 (a NaN takes the update branch)
    const double updf = upd ? 1.0 : 0.0;
    const double Fe = upd ? F : 1.0;
    const double iF = rcp(Fe) * updf;
    C.ld.mul(Fe);                                              // log(detF) = D log|F| (line 234)
    C.nupd += updf;
    const double bm = (na || upd) ? 1.0 : 0.0;                 // Q3: detF <= 0 predicts WITHOUT B mu (lines 226-228)
    const double e = tr.e, t12 = tr.t12, e2 = tr.e2;
    // ---- covariance, primal ------------------------------------------------------------------------------------
    const double a = fma(h, iF, 1.0 - updf);                   // h / F (1 on a row that is not scored)
    const double kf1 = C.p11 * iF, kf2 = C.p12 * iF;           // filter gain P z / F
    const double f11 = C.p11 * a, f12 = C.p12 * a, f22 = fma(-C.p12, kf2, C.p22);   // P~
    const double m = fma(t12, f22, f12);
    const double k1 = fma(t12, kf2, kf1), k2 = e * kf2;        // K = T P z / F (line 236)
    const double c1 = 1.0 - k1;
    // ---- residuals ------------------------------------------------------------------------------------------------
    double u[D], mue[D];
    double su2 = 0.0;
    for (int a_ = 0; a_ < D; a_++) {
        const double ys = upd ? y[a_] : M.x[a_];               // (a missing y is NaN: keep it out of the arithmetic)
        u[a_] = ys - M.x[a_];                                  // line 221
        su2 = fma(u[a_], u[a_], su2);
        mue[a_] = bm * mu[a_];
    }
    M.accq = fma(iF, su2, M.accq);                             // u' F^-1 u
    const double w2 = -0.5 * iF * iF, aiF = a * iF;
    // ---- covariance-affecting directions ----------------------------------------------------------------------
    for (int j = 0; j < NDIRP; j++) {
        if (!(MASK & dir_bit(j))) continue;
        const double d11 = C.d11[j], d12 = C.d12[j], d22 = C.d22[j];
        const double h2 = 2.0 * h;
        const double dF = (j == 0) ? d11 + h2 : d11;
        C.gld[j] = fma(dF, iF, C.gld[j]);
        const double w = fma(-kf2, d11, d12);
        double g11 = a * a * d11, g12 = a * w, g22 = fma(-kf2, d12 + w, d22);        // dP~ = A dP A'
        double dkf1 = d11 * aiF, dkf2 = w * iF;                                      // d(P z / F)
        if (j == 0) {                                                               // ... + k k' dh, - k dh / F
            const double q1 = kf1 * h2, q2 = kf2 * h2;
            g11 = fma(kf1, q1, g11); g12 = fma(kf2, q1, g12); g22 = fma(kf2, q2, g22);
            dkf1 = fma(-q1, iF, dkf1); dkf2 = fma(-q2, iF, dkf2);
        }
        const double dm = fma(t12, g22, g12);
        double n11 = fma(t12, g12 + dm, g11), n12 = e * dm, n22 = e2 * g22;          // T dP~ T'
        double dk1 = fma(t12, dkf2, dkf1), dk2 = e * dkf2;
        if (j == 1) {                                                               // only log tau moves T
            const double dt12 = tr.dt12, de = tr.de;
            n11 = fma(tr.dt12x2, m, n11) + tr.dq11;
            n12 = fma(tr.dt12e, f22, fma(de, m, n12)) + tr.dq12;
            n22 = fma(tr.edex2, f22, n22) + tr.dq22;
            dk1 = fma(dt12, kf2, dk1); dk2 = fma(de, kf2, dk2);
        }
        if (j == 2) { n11 = fma(2.0, tr.q11, n11); n12 = fma(2.0, tr.q12, n12); n22 = fma(2.0, tr.q22, n22); }
        C.d11[j] = n11; C.d12[j] = n12; C.d22[j] = n22;
        // mean half of the direction (lines 221, 231-234, 238 differentiated)
        double sud = 0.0;
        for (int a_ = 0; a_ < D; a_++) {
            const double tx = M.tx[j][a_], tv = M.tv[j][a_];
            sud = fma(u[a_], tx, sud);
            double nx = fma(dk1, u[a_], fma(t12, tv, c1 * tx));
            double nv = fma(dk2, u[a_], fma(e, tv, -k2 * tx));
            if (j == 1) {                                      // d(B mu) = -(dt12, de) mu
                const double wv = M.v[a_] - mue[a_];
                nx = fma(tr.dt12, wv, nx); nv = fma(tr.de, wv, nv);
            }
            M.tx[j][a_] = nx; M.tv[j][a_] = nv;
        }
        M.gq[j] = fma(w2 * dF, su2, fma(-iF, sud, M.gq[j]));     // 1/2 dF^-1 |u|^2 + F^-1 u' du,  du = -tx
    }
    if (MASK & DIR_MU) {
        const double mx = M.mx[0], mv = M.mv[0], imx = iF * mx;
        const double nx = fma(bm, tr.b1, fma(t12, mv, c1 * mx)), nv = fma(bm, tr.b2, fma(e, mv, -k2 * mx));
        for (int a_ = 0; a_ < D; a_++) {
            M.gmu[a_] = fma(-imx, u[a_], M.gmu[a_]);
            M.mx[a_] = nx; M.mv[a_] = nv;
        }
    }
    // ---- state and covariance, primal update ------------------------------------------------------------------
    for (int a_ = 0; a_ < D; a_++) {                           // a = T a + K u + B mu (line 238)
        const double nx = fma(tr.b1, mue[a_], fma(k1, u[a_], fma(t12, M.v[a_], M.x[a_])));
        const double nv = fma(tr.b2, mue[a_], fma(k2, u[a_], e * M.v[a_]));
        M.x[a_] = nx; M.v[a_] = nv;
    }
    C.p11 = fma(t12, f12 + m, f11) + tr.q11;                   // P = T P~ T' + Q (lines 240-241)
    C.p12 = fma(e, m, tr.q12);
    C.p22 = fma(e2, f22, tr.q22);
}

// totals: nllk contribution and gradient slots [sig, mu_0..mu_{D-1}, p1, p2]; the covariance
// half contributes (D/2) log-determinant terms, the mean half the quadratic-form terms
template <int D, int MASK>
SSDE_HD void ctcrw_finish_parts(double ld, const double* gld, const CtcrwMean<D, MASK>& M, double* out /* 1 + 3 + D */) {
    out[0] = 0.5 * ((double)D * ld + M.accq);
    out[1] = (MASK & DIR_SIG) ? 0.5 * D * gld[0] + M.gq[0] : 0.0;
    for (int a = 0; a < D; a++) out[2 + a] = (MASK & DIR_MU) ? M.gmu[a] : 0.0;
    out[2 + D] = (MASK & DIR_P1) ? 0.5 * D * gld[1] + M.gq[1] : 0.0;
    out[3 + D] = (MASK & DIR_P2) ? 0.5 * D * gld[2] + M.gq[2] : 0.0;
}
template <int D, int MASK>
SSDE_HD void ctcrw_finish(const CtcrwLane<D, MASK>& L, double* out) {
    ctcrw_finish_parts<D, MASK>(L.C.ld.value(), L.C.gld, L.M, out);
}

// ---------------------------------------------------------------------------------------
// OU_SSM / BM_SSM, isotropic: state x[a], covariance p (scalar), transition a' = t a + b mu_a.
// Same split into a covariance half and a mean half.
// ---------------------------------------------------------------------------------------
struct ScalTrans {
    double t, b, q;     // OU: t = e^{-dt/tau}, b = 1 - t, q = kappa (1 - e^{-2dt/tau});  BM: t = 1, b = dt, q = sigma^2 dt
    double dt_, db, dq; // d/d(par n_dim): log tau (OU) or log sigma (BM)
                        // OU d/d(log kappa): dq2 = q, rest 0
};

SSDE_HD void ou_trans(double dt, double tau, double kappa, ScalTrans& o) {
    const double z = dt / tau;
    const double e = exp(-z);               // makeT_ou_ssm line 35
    const double e2 = e * e;                // exp(-2 dt / tau)
    o.t = e;
    o.b = 1.0 - e;                          // makeB line 50
    o.q = kappa * (1.0 - e2);               // makeQ line 66
    o.dt_ = e * z;
    o.db = -e * z;
    o.dq = -2.0 * kappa * e2 * z;
}
SSDE_HD void bm_trans(double dt, double sigma, ScalTrans& o) {
    o.t = 1.0;
    o.b = dt;                               // drift = mu * dt (nllk_bm_ssm.hpp:139)
    o.q = sigma * sigma * dt;               // makeQ_bm_ssm line 33
    o.dt_ = 0.0;
    o.db = 0.0;
    o.dq = 2.0 * o.q;
}

struct ScalGain {
    double iF, k, c;                 // c = t - k, the closed-loop factor (in its cancellation-free form t h / F)
    double diF[NDIRP], dk[NDIRP];
};

template <int MASK>
struct ScalCov {
    double p, dp[NDIRP];
    LogAcc ld;
    double gld[NDIRP];
    double nupd;        // rows that entered the likelihood
    SSDE_HD void init(double p0) { p = p0; for (int j = 0; j < NDIRP; j++) dp[j] = gld[j] = 0.0; ld.init(); nupd = 0.0; }
    SSDE_HD void reset_acc() { ld.init(); nupd = 0.0; for (int j = 0; j < NDIRP; j++) gld[j] = 0.0; }
};

// Both families take detF = exp(logdet F) > 0 unless F == 0 (nllk_ou_ssm.hpp:190-195,
// nllk_bm_ssm.hpp:152-157) and keep the drift in every branch (Q3).
template <int D, int MASK, bool HAS_P2>
SSDE_HD void scal_cov_step(ScalCov<MASK>& C, const ScalTrans& tr, double h, bool na, ScalGain& G) {
    const double F = C.p + h;
    const bool upd = !na && !(fabs(F) <= 0.0);                 // (a NaN takes the update branch, as in the reference)
    const double updf = upd ? 1.0 : 0.0;
    const double Fe = upd ? F : 1.0;
    const double iF = rcp(Fe) * updf;
    C.ld.mul(Fe);
    C.nupd += updf;
    // The literal update P = T P (T - K Z)' + Q (nllk_ou_ssm.hpp:205-206) is t p (t - k) + q with t - k = t (1 - p/F):
    // once p >> h the difference cancels to nothing and the next p is rounding noise times p.  The same numbers
    // without the cancellation: with a = h/F, b = p/F (a + b = 1), c = t a the closed-loop factor
    //     p'  = t c p + q                                    k  = t b
    //     dp' = t c a dp + t^2 b^2 dh + 2 dt c p + dq         dk = (t/F) (a dp - b dh) + dt b
    // (d[p h / F] = (h^2 dp + p^2 dh) / F^2).  A row that is not scored has iF = 0: a = 1, b = 0, and the update
    // formulas are the prediction formulas -- no select beyond F and the 0/1 flag.
    // (BM_SSM -- the model without a second scale parameter -- has t = 1 and dt = db = 0: folded at compile time)
    const double t = HAS_P2 ? tr.t : 1.0, dt_ = HAS_P2 ? tr.dt_ : 0.0;
    const double a = fma(h, iF, 1.0 - updf), b = C.p * iF;
    const double c = t * a, k = t * b, tc = t * c;
    const double tiF = t * iF, ca = tiF * a, tca = tc * a, cp = c * C.p;
    G.iF = iF; G.k = k; G.c = c;
    const double w2 = -iF * iF;
    for (int j = 0; j < NDIRP; j++) {
        if (!(MASK & dir_bit(j)) || (j == 2 && !HAS_P2)) { G.diF[j] = G.dk[j] = 0.0; continue; }
        const double dp = C.dp[j];
        const double dF = (j == 0) ? dp + 2.0 * h : dp;
        G.diF[j] = w2 * dF;
        C.gld[j] = fma(dF, iF, C.gld[j]);
        double dk = ca * dp, np_ = tca * dp;
        if (j == 0) { const double bh = b * (2.0 * h); dk = fma(-tiF, bh, dk); np_ = fma(k * t, bh, np_); }
        if (j == 1) {
            if (HAS_P2) { dk = fma(dt_, b, dk); np_ = fma(2.0 * dt_, cp, np_); }
            np_ += tr.dq;
        }
        if (j == 2) np_ += tr.q;
        G.dk[j] = dk;
        C.dp[j] = np_;
    }
    C.p = fma(tc, C.p, tr.q);
}

template <int D, int MASK>
struct ScalMean {
    double x[D], tx[NDIRP][D], mx[D];
    double accq, gq[NDIRP], gmu[D];
    SSDE_HD void init(const double* a0) {
        for (int a = 0; a < D; a++) {
            x[a] = a0[a]; mx[a] = gmu[a] = 0.0;
            for (int j = 0; j < NDIRP; j++) tx[j][a] = 0.0;
        }
        accq = 0.0;
        for (int j = 0; j < NDIRP; j++) gq[j] = 0.0;
    }
    SSDE_HD void reset_acc() {
        accq = 0.0;
        for (int j = 0; j < NDIRP; j++) gq[j] = 0.0;
        for (int a = 0; a < D; a++) gmu[a] = 0.0;
    }
};

template <int D, int MASK, bool HAS_P2>
SSDE_HD void scal_mean_step(ScalMean<D, MASK>& M, const ScalTrans& tr, const ScalGain& G, const double* mu,
                            const double* y, bool scored) {
    double u[D];
    double su2 = 0.0;
    for (int a = 0; a < D; a++) {
        const double ys = scored ? y[a] : M.x[a];              // (a missing y is NaN: keep it out of the arithmetic)
        u[a] = ys - M.x[a];
        su2 = fma(u[a], u[a], su2);
    }
    M.accq = fma(G.iF, su2, M.accq);
    // x' = t x + k u + b mu;  tx' = (t - k) tx + dk u [+ dt_ x + db mu]: a row that is not scored has k = dk = 0
    for (int j = 0; j < NDIRP; j++) {
        if (!(MASK & dir_bit(j)) || (j == 2 && !HAS_P2)) continue;
        double sud = 0.0;
        for (int a = 0; a < D; a++) {
            const double tx = M.tx[j][a];
            sud = fma(u[a], tx, sud);
            double nx = fma(G.dk[j], u[a], G.c * tx);
            if (j == 1 && HAS_P2) nx = fma(tr.dt_, M.x[a], fma(tr.db, mu[a], nx));   // (BM_SSM: T = I, B = dt: no parameter in them)
            M.tx[j][a] = nx;
        }
        M.gq[j] = fma(0.5 * G.diF[j], su2, fma(-G.iF, sud, M.gq[j]));
    }
    if (MASK & DIR_MU) {                                       // d x_a / d mu_a: the same number for every dimension
        const double mx = M.mx[0], imx = G.iF * mx, nx = fma(G.c, mx, tr.b);
        for (int a = 0; a < D; a++) {
            M.gmu[a] = fma(-imx, u[a], M.gmu[a]);
            M.mx[a] = nx;
        }
    }
    for (int a = 0; a < D; a++) M.x[a] = fma(tr.b, mu[a], fma(G.k, u[a], HAS_P2 ? tr.t * M.x[a] : M.x[a]));
}

template <int D, int MASK>
struct ScalLane {
    ScalCov<MASK> C;
    ScalMean<D, MASK> M;
    SSDE_HD void init(const double* a0, double p0) { C.init(p0); M.init(a0); }
    SSDE_HD void reset_acc() { C.reset_acc(); M.reset_acc(); }
    static constexpr int NSTATE = 4 * (D + 1) + D;
    SSDE_HD void dump(double* o) const {
        int k = 0;
        for (int a = 0; a < D; a++) o[k++] = M.x[a];
        o[k++] = C.p;
        for (int j = 0; j < NDIRP; j++) {
            const bool on = (MASK & dir_bit(j)) != 0;
            o[k++] = on ? C.dp[j] : 0.0;
            for (int a = 0; a < D; a++) o[k++] = on ? M.tx[j][a] : 0.0;
        }
        for (int a = 0; a < D; a++) o[k++] = (MASK & DIR_MU) ? M.mx[a] : 0.0;
    }
};

template <int D, int MASK, bool HAS_P2>
SSDE_HD void scal_step(ScalLane<D, MASK>& L, const ScalTrans& tr, double h, const double* mu, const double* y,
                       bool na) {
    ScalGain G;
    scal_cov_step<D, MASK, HAS_P2>(L.C, tr, h, na, G);
    scal_mean_step<D, MASK, HAS_P2>(L.M, tr, G, mu, y, G.iF != 0.0);
}

template <int D, int MASK>
SSDE_HD void scal_finish_parts(double ld, const double* gld, const ScalMean<D, MASK>& M, double* out /* 1 + 3 + D */) {
    out[0] = 0.5 * ((double)D * ld + M.accq);
    out[1] = (MASK & DIR_SIG) ? 0.5 * D * gld[0] + M.gq[0] : 0.0;
    for (int a = 0; a < D; a++) out[2 + a] = (MASK & DIR_MU) ? M.gmu[a] : 0.0;
    out[2 + D] = (MASK & DIR_P1) ? 0.5 * D * gld[1] + M.gq[1] : 0.0;
    out[3 + D] = (MASK & DIR_P2) ? 0.5 * D * gld[2] + M.gq[2] : 0.0;
}
template <int D, int MASK>
SSDE_HD void scal_finish(const ScalLane<D, MASK>& L, double* out) {
    scal_finish_parts<D, MASK>(L.C.ld.value(), L.C.gld, L.M, out);
}

// ---------------------------------------------------------------------------------------
// Direct families: one transition z0 -> z1 of one dimension, parameters of the EARLIER row
// (Q6; nllk_sde.hpp:80-81).  Returns -log density and adds d(-log dens)/d par into g[].
// ---------------------------------------------------------------------------------------
#define SSDE_LOG_SQRT_2PI 0.91893853320467274178

// BM: par = (mu_a, log sigma): tr_dens.hpp:35-37
SSDE_HD double bm_direct(double z0, double z1, double dt, double mu, double lsig, double& g_mu, double& g_ls) {
    const double sd = exp(lsig) * sqrt(dt);
    const double r = (z1 - (z0 + mu * dt)) / sd;
    g_mu += -r * dt / sd;
    g_ls += 1.0 - r * r;
    return SSDE_LOG_SQRT_2PI + log(sd) + 0.5 * r * r;
}
// BM_t: Brownian motion with Student-t increments, par = (mu, log sigma), df degrees of freedom (tr_dens.hpp:38-44):
//   scale = sd / sqrt(df / (df - 2)),  log dens = dt(x, df, log) - log(scale),  x = (z1 - z0 - mu dt) / scale
// tconst = lgamma((df+1)/2) - lgamma(df/2) - log(df pi)/2 (R's dt() normalising constant; host side)
SSDE_HD double bmt_direct(double z0, double z1, double dt, double mu, double lsig, double df, double tconst,
                          double& g_mu, double& g_ls) {
    const double sd = exp(lsig) * sqrt(dt);
    const double scale = sd / sqrt(df / (df - 2.0));
    const double x = (z1 - z0 - mu * dt) / scale;
    const double q = x * x / df;
    const double w = (df + 1.0) * x / (df + x * x);        // d/dx of (df+1)/2 log(1 + x^2/df)
    g_mu += -w * dt / scale;
    g_ls += 1.0 - w * x;
    return -(tconst - 0.5 * (df + 1.0) * log1p(q) - log(scale));
}
// OU: par = (mu_a, log tau, log kappa): tr_dens.hpp:49-52
SSDE_HD double ou_direct(double z0, double z1, double dt, double mu, double ltau, double lkap, double& g_mu,
                         double& g_lt, double& g_lk) {
    const double tau = exp(ltau);
    const double z = dt / tau;
    const double e = exp(-z);
    const double e2 = e * e;
    const double var = exp(lkap) * (1.0 - e2);
    const double sd = sqrt(var);
    const double mean = mu + e * (z0 - mu);
    const double r = (z1 - mean) / sd;
    // d mean / d mu = 1 - e ; d mean / d ltau = e z (z0 - mu) ; d log sd / d ltau = -e2 z / (1 - e2) ; d log sd / d lkap = 1/2
    const double dls_lt = -e2 * z / (1.0 - e2);
    g_mu += -r * (1.0 - e) / sd;
    g_lt += -r * (e * z * (z0 - mu)) / sd + (1.0 - r * r) * dls_lt;
    g_lk += 0.5 * (1.0 - r * r);
    return SSDE_LOG_SQRT_2PI + log(sd) + 0.5 * r * r;
}

// ---------------------------------------------------------------------------------------
// CIR (tr_dens.hpp:53-67): the transition density needs log I_q(x) for a real order q > -1 and its derivatives
// with respect to x AND q (q = 2 beta mu / sigma^2 - 1 moves with every parameter).  The reference takes
// log(besselI(x, q)) (TMB's bessel_i, AD through R's algorithm), which overflows to Inf for x > ~700; here the
// logarithm is formed directly from the ascending series
//     I_q(x) = (x/2)^q * sum_k t_k,     t_k = (x^2/4)^k / (k! Gamma(k + q + 1)),
// whose terms are all positive (no cancellation).  The terms rise to k* = (sqrt(q^2 + x^2) - q) / 2 and fall off
// on both sides of it like a Gaussian of width ~ sqrt(x/2), so the sum is walked outwards from k* in units of t_k*
// (log t_k* from lgamma): ~ 12 sqrt(x) terms instead of x/2 + 40, nothing to rescale, any x.
//     d log I / dx = q/x + 2 sum_k k t_k / (x S),      d log I / dq = log(x/2) - sum_k t_k psi(k + q + 1) / S.
// ---------------------------------------------------------------------------------------
SSDE_HD double digamma_pos(double x) {   // psi(x), x > 0: recurrence up to x >= 8, then the asymptotic series
    double r = 0.0;
    while (x < 8.0) { r -= 1.0 / x; x += 1.0; }
    const double i2 = 1.0 / (x * x);
    // B_2n/(2n): 1/12, -1/120, 1/252, -1/240, 1/132, -691/32760, 1/12
    const double s = i2 * (1.0 / 12.0 - i2 * (1.0 / 120.0 - i2 * (1.0 / 252.0 - i2 * (1.0 / 240.0 - i2 * (1.0 / 132.0 -
                     i2 * (691.0 / 32760.0 - i2 * (1.0 / 12.0)))))));
    return r + log(x) - 0.5 / x - s;
}

SSDE_HD double log_bessel_i(double x, double q, double& dlog_dx, double& dlog_dq) {
    const double y = 0.25 * x * x;
    double ks = floor(0.5 * (sqrt(q * q + x * x) - q));          // index of the largest term
    if (!(ks >= 1.0)) ks = 0.0;
    // the terms fall off around k* like a Gaussian of variance k*(k*+q)/(2k*+q): 1e-17 is reached 8.9 standard
    // deviations out.  Beyond k* ~ 1.8e7 (x ~ 3e7) the walk would take > 36000 steps per side: NaN (a rejected step)
    const double sd = sqrt(ks * (ks + q) / (2.0 * ks + q + 1e-300));
    if (!(sd < 3.0e3)) { dlog_dx = dlog_dq = NAN; return NAN; }
    const int cap = 100 + (int)(12.0 * sd);
    const double psi0 = digamma_pos(ks + q + 1.0);
    double S = 1.0, A1 = ks, A2 = psi0;                          // sums in units of t_ks
    double t = 1.0, psi = psi0, k = ks;
    for (int it = 0; it < cap; it++) {                           // upwards: t_k = t_{k-1} y / (k (k + q))
        k += 1.0;
        const double ik = 1.0 / (k + q);
        t *= y * ik / k;
        psi += ik;
        S += t; A1 += k * t; A2 += psi * t;
        if (!(t >= 1e-17 * S)) break;                            // (also leaves on NaN)
    }
    t = 1.0; psi = psi0; k = ks;
    for (int it = 0; it < cap && k >= 1.0; it++) {               // downwards: t_{k-1} = t_k k (k + q) / y
        const double tk = t * k / y;
        t = tk * (k + q);
        A2 += t * psi - tk;      // t_{k-1} psi_{k-1} with psi_{k-1} = psi_k - 1/(k+q), free of the pole at k + q = 0 (q -> -1)
        psi -= 1.0 / (k + q);
        k -= 1.0;
        S += t; A1 += k * t;
        if (!(t >= 1e-17 * S)) break;
    }
    const double lx2 = log(0.5 * x);
    const double lt = (ks > 0.0 ? ks * log(y) - lgamma(ks + 1.0) : 0.0) - lgamma(ks + q + 1.0);
    dlog_dx = q / x + 2.0 * A1 / (x * S);
    dlog_dq = lx2 - A2 / S;
    return q * lx2 + lt + log(S);
}

// psi'(x), x > 0
SSDE_HD double trigamma_pos(double x) {
    double r = 0.0;
    while (x < 8.0) { r += 1.0 / (x * x); x += 1.0; }
    const double ix = 1.0 / x, i2 = ix * ix;
    // 1/x + 1/(2x^2) + sum B_2n / x^(2n+1): 1/6, -1/30, 1/42, -1/30, 5/66, -691/2730, 7/6
    return r + ix + 0.5 * i2 + ix * i2 * (1.0 / 6.0 - i2 * (1.0 / 30.0 - i2 * (1.0 / 42.0 - i2 * (1.0 / 30.0 - i2 * (5.0 / 66.0 -
               i2 * (691.0 / 2730.0 - i2 * (7.0 / 6.0)))))));
}

// log I_q(x) with its first AND second derivatives (the exact Hessian of the CIR density, k_direct_hess.hip).  The same outward walk;
// with the weights w_k = t_k / S of the series the derivatives are moments of k and psi_k = psi(k + q + 1) under w:
//     l_x = (q + 2 E k) / x,   l_q = log(x/2) - E psi,
//     l_xx = (4 Var k - q - 2 E k) / x^2,   l_xq = (1 - 2 Cov(k, psi)) / x,   l_qq = Var psi - E psi'
// (moments taken about the largest term's k* and psi_k*: no cancellation in the variances).  out = {l_x, l_q, l_xx, l_xq, l_qq}.
SSDE_HD double log_bessel_i2(double x, double q, double (&out)[5]) {
    const double y = 0.25 * x * x;
    double ks = floor(0.5 * (sqrt(q * q + x * x) - q));
    if (!(ks >= 1.0)) ks = 0.0;
    const double sd = sqrt(ks * (ks + q) / (2.0 * ks + q + 1e-300));
    if (!(sd < 3.0e3)) { for (int i = 0; i < 5; i++) out[i] = NAN; return NAN; }
    const int cap = 100 + (int)(12.0 * sd);
    const double psi0 = digamma_pos(ks + q + 1.0), tri0 = trigamma_pos(ks + q + 1.0);
    double S = 1.0, Sk = 0.0, Skk = 0.0, Sp = 0.0, Spp = 0.0, Skp = 0.0, St = tri0;
    double t = 1.0, dp = 0.0, tri = tri0, k = ks;
    for (int it = 0; it < cap; it++) {                           // upwards
        k += 1.0;
        const double ik = 1.0 / (k + q);
        t *= y * ik / k;
        dp += ik; tri -= ik * ik;
        const double dk = k - ks;
        S += t; Sk += t * dk; Skk += t * dk * dk; Sp += t * dp; Spp += t * dp * dp; Skp += t * dk * dp; St += t * tri;
        if (!(t >= 1e-17 * S)) break;
    }
    t = 1.0; dp = 0.0; tri = tri0; k = ks;
    for (int it = 0; it < cap && k >= 1.0; it++) {               // downwards: t_{k-1} = t_k k (k + q) / y
        const double ik = 1.0 / (k + q);
        t *= k * (k + q) / y;
        dp -= ik; tri += ik * ik;
        k -= 1.0;
        const double dk = k - ks;
        S += t; Sk += t * dk; Skk += t * dk * dk; Sp += t * dp; Spp += t * dp * dp; Skp += t * dk * dp; St += t * tri;
        if (!(t >= 1e-17 * S)) break;
    }
    const double iS = 1.0 / S, Ek = Sk * iS, Ep = Sp * iS;
    const double Vk = Skk * iS - Ek * Ek, Vp = Spp * iS - Ep * Ep, Ckp = Skp * iS - Ek * Ep;
    const double lx2 = log(0.5 * x), ix = 1.0 / x, km = ks + Ek;
    const double lt = (ks > 0.0 ? ks * log(y) - lgamma(ks + 1.0) : 0.0) - lgamma(ks + q + 1.0);
    out[0] = (q + 2.0 * km) * ix;
    out[1] = lx2 - (psi0 + Ep);
    out[2] = (4.0 * Vk - q - 2.0 * km) * ix * ix;
    out[3] = (1.0 - 2.0 * Ckp) * ix;
    out[4] = Vp - St * iS;
    return q * lx2 + lt + log(S);
}

// CIR: par = (log mu_a, log beta, log sigma).  Returns -log density, adds d/d(log mu_a, log beta, log sigma).
SSDE_HD double cir_direct(double z0, double z1, double dt, double lmu, double lbeta, double lsig, double& g_lm,
                          double& g_lb, double& g_ls) {
    const double mu = exp(lmu), beta = exp(lbeta), s2 = exp(2.0 * lsig);
    const double bd = beta * dt, em = exp(-bd);
    const double c = 2.0 * beta / ((1.0 - em) * s2);               // tr_dens.hpp:60
    const double q1 = 2.0 * beta * mu / s2, q = q1 - 1.0;           // :61
    const double u = c * z0 * em, v = c * z1;                       // :62-63
    const double x = 2.0 * sqrt(u * v);                             // :64
    double dIx, dIq;
    const double lI = log_bessel_i(x, q, dIx, dIq);
    const double lvu = log(v) - log(u);
    const double ld = log(c) - u - v + 0.5 * q * lvu + lI;          // :66
    // log-derivatives w.r.t. (log mu, log beta, log sigma)
    const double dlc_b = 1.0 - bd * em / (1.0 - em), dlc_s = -2.0;
    const double dlu_b = dlc_b - bd, dlu_s = -2.0, dlv_b = dlc_b, dlv_s = -2.0;
    const double dq_m = q1, dq_b = q1, dq_s = -2.0 * q1;
    const double xI = x * dIx;
    const double d_m = 0.5 * dq_m * lvu + dIq * dq_m;
    const double d_b = dlc_b - u * dlu_b - v * dlv_b + 0.5 * dq_b * lvu + 0.5 * q * (dlv_b - dlu_b) + xI * 0.5 * (dlu_b + dlv_b) + dIq * dq_b;
    const double d_s = dlc_s - u * dlu_s - v * dlv_s + 0.5 * dq_s * lvu + 0.5 * q * (dlv_s - dlu_s) + xI * 0.5 * (dlu_s + dlv_s) + dIq * dq_s;
    g_lm -= d_m; g_lb -= d_b; g_ls -= d_s;
    return -ld;
}

}  // namespace ssde
#endif
