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
enum { M_BM = 0, M_OU = 1, M_BM_SSM = 2, M_OU_SSM = 3, M_CTCRW = 4 };

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
}

template <int D>
struct CtcrwTan {  // one covariance-affecting direction (sigma_obs, tau or nu)
    double p11, p12, p22, x[D], v[D], g;
    SSDE_HD void init() {
        p11 = p12 = p22 = g = 0.0;
        for (int a = 0; a < D; a++) x[a] = v[a] = 0.0;
    }
};

template <int D, int MASK>
struct CtcrwLane {
    double x[D], v[D], p11, p12, p22;
    LogAcc ld;
    double accq;
    CtcrwTan<D> ts, tt, tn;      // DIR_SIG, DIR_P1 (log tau), DIR_P2 (log nu)
    double mx[D], mv[D], gmu[D]; // DIR_MU: d/d mu_a touches dimension a only

    // a0 = (x_1, v_1, x_2, v_2, ...): one row of the reference's a0 matrix (R/sde.R:576-580)
    SSDE_HD void init(const double* a0, double p0_11, double p0_12, double p0_22) {
        for (int a = 0; a < D; a++) { x[a] = a0[2 * a]; v[a] = a0[2 * a + 1]; mx[a] = mv[a] = gmu[a] = 0.0; }
        p11 = p0_11; p12 = p0_12; p22 = p0_22;
        ld.init();
        accq = 0.0;
        ts.init(); tt.init(); tn.init();
    }
    // accumulators only (used when a time window starts scoring after its warm-up rows)
    SSDE_HD void reset_acc() {
        ld.init();
        accq = 0.0;
        ts.g = tt.g = tn.g = 0.0;
        for (int a = 0; a < D; a++) gmu[a] = 0.0;
    }
    static constexpr int NSTATE = 4 * (2 * D + 3) + 2 * D;
    SSDE_HD static void dump_tan(const CtcrwTan<D>& t, bool on, double* o, int& k) {
        o[k++] = on ? t.p11 : 0.0; o[k++] = on ? t.p12 : 0.0; o[k++] = on ? t.p22 : 0.0;
        for (int a = 0; a < D; a++) { o[k++] = on ? t.x[a] : 0.0; o[k++] = on ? t.v[a] : 0.0; }
    }
    // filter state + sensitivities, for the window hand-over check
    SSDE_HD void dump(double* o) const {
        int k = 0;
        for (int a = 0; a < D; a++) { o[k++] = x[a]; o[k++] = v[a]; }
        o[k++] = p11; o[k++] = p12; o[k++] = p22;
        dump_tan(ts, (MASK & DIR_SIG) != 0, o, k);
        dump_tan(tt, (MASK & DIR_P1) != 0, o, k);
        dump_tan(tn, (MASK & DIR_P2) != 0, o, k);
        for (int a = 0; a < D; a++) { o[k++] = (MASK & DIR_MU) ? mx[a] : 0.0; o[k++] = (MASK & DIR_MU) ? mv[a] : 0.0; }
    }
};

// tangent of one covariance-affecting direction; seeds are the derivatives of
// (h, e, t12, b1, b2, q11, q12, q22) in that direction.
template <int D>
SSDE_HD void ctcrw_tan_step(CtcrwTan<D>& t, const double* v, double p12, double p22,
                            const CtcrwTrans& tr, double iF, double su2, const double* u, double k1, double k2,
                            double tp11, double tp12, double tp21, double bm, const double* mu, double dh,
                            double de, double dt12, double dq11, double dq12, double dq22, bool upd) {
    const double dF = t.p11 + dh;
    const double diF = -iF * iF * dF;
    double sud = 0.0;
    double du[D];
    for (int a = 0; a < D; a++) { du[a] = upd ? -t.x[a] : 0.0; sud += u[a] * du[a]; }
    t.g += 0.5 * ((double)D * dF * iF + diF * su2) + iF * sud;
    const double dtp11 = t.p11 + dt12 * p12 + tr.t12 * t.p12;
    const double dtp12 = t.p12 + dt12 * p22 + tr.t12 * t.p22;
    const double dtp21 = de * p12 + tr.e * t.p12;
    const double dtp22 = de * p22 + tr.e * t.p22;
    const double dk1 = dtp11 * iF + tp11 * diF;
    const double dk2 = dtp21 * iF + tp21 * diF;
    for (int a = 0; a < D; a++) {
        const double bmu = bm * mu[a];
        const double nx = t.x[a] + dt12 * v[a] + tr.t12 * t.v[a] + dk1 * u[a] + k1 * du[a] - dt12 * bmu;
        const double nv = de * v[a] + tr.e * t.v[a] + dk2 * u[a] + k2 * du[a] - de * bmu;
        t.x[a] = nx; t.v[a] = nv;
    }
    const double tp22 = tr.e * p22;
    t.p11 = dtp11 * (1.0 - k1) - tp11 * dk1 + dtp12 * tr.t12 + tp12 * dt12 + dq11;
    t.p12 = -dtp11 * k2 - tp11 * dk2 + dtp12 * tr.e + tp12 * de + dq12;
    t.p22 = -dtp21 * k2 - tp21 * dk2 + dtp22 * tr.e + tp22 * de + dq22;
}

// One row of a track: score y (unless NA), then propagate over the interval described by tr.
//   h = sigma_obs^2; mu[a] = mean velocity; na = obs(i,0) is NA (nllk_ctcrw.hpp:214)
template <int D, int MASK>
SSDE_HD void ctcrw_step(CtcrwLane<D, MASK>& L, const CtcrwTrans& tr, double h, const double* mu, const double* y,
                        bool na) {
    const double F = L.p11 + h;                                // F = Z P Z' + H (line 223), scalar per dimension
    const double detF = (D == 1) ? F : F * F;                  // det(): lines 16-19
    const bool upd = !na && (detF > 0.0);                      // lines 214, 226
    const double iF = upd ? rcp(F) : 0.0;
    double u[D];
    double su2 = 0.0;
    for (int a = 0; a < D; a++) { u[a] = upd ? y[a] - L.x[a] : 0.0; su2 += u[a] * u[a]; }  // line 221
    L.ld.mul(upd ? F : 1.0);                                   // log(detF) = D log|F| (line 234)
    L.accq += iF * su2;                                        // u' F^-1 u   (lines 231-233)
    const double tp11 = L.p11 + tr.t12 * L.p12, tp12 = L.p12 + tr.t12 * L.p22;  // T P
    const double tp21 = tr.e * L.p12, tp22 = tr.e * L.p22;
    const double k1 = tp11 * iF, k2 = tp21 * iF;               // K = T P Z' F^-1 (line 236)
    // Q3: the detF <= 0 branch of CTCRW predicts WITHOUT B mu (lines 226-228)
    const double bm = (na || upd) ? 1.0 : 0.0;

    if (MASK & DIR_SIG)
        ctcrw_tan_step<D>(L.ts, L.v, L.p12, L.p22, tr, iF, su2, u, k1, k2, tp11, tp12, tp21, bm, mu,
                          2.0 * h, 0.0, 0.0, 0.0, 0.0, 0.0, upd);
    if (MASK & DIR_P1)
        ctcrw_tan_step<D>(L.tt, L.v, L.p12, L.p22, tr, iF, su2, u, k1, k2, tp11, tp12, tp21, bm, mu,
                          0.0, tr.de, tr.dt12, tr.dq11, tr.dq12, tr.dq22, upd);
    if (MASK & DIR_P2)
        ctcrw_tan_step<D>(L.tn, L.v, L.p12, L.p22, tr, iF, su2, u, k1, k2, tp11, tp12, tp21, bm, mu,
                          0.0, 0.0, 0.0, 2.0 * tr.q11, 2.0 * tr.q12, 2.0 * tr.q22, upd);
    if (MASK & DIR_MU) {
        for (int a = 0; a < D; a++) {
            const double du = upd ? -L.mx[a] : 0.0;
            L.gmu[a] += iF * u[a] * du;
            const double nx = L.mx[a] + tr.t12 * L.mv[a] + k1 * du + bm * tr.b1;
            const double nv = tr.e * L.mv[a] + k2 * du + bm * tr.b2;
            L.mx[a] = nx; L.mv[a] = nv;
        }
    }
    for (int a = 0; a < D; a++) {                              // a = T a + K u + B mu (line 238)
        const double nx = L.x[a] + tr.t12 * L.v[a] + k1 * u[a] + bm * tr.b1 * mu[a];
        const double nv = tr.e * L.v[a] + k2 * u[a] + bm * tr.b2 * mu[a];
        L.x[a] = nx; L.v[a] = nv;
    }
    const double n11 = tp11 * (1.0 - k1) + tp12 * tr.t12 + tr.q11;  // P = T P (T - K Z)' + Q (lines 240-241)
    const double n12 = -tp11 * k2 + tp12 * tr.e + tr.q12;
    const double n22 = -tp21 * k2 + tp22 * tr.e + tr.q22;
    L.p11 = n11; L.p12 = n12; L.p22 = n22;
}

// lane totals: nllk contribution and gradient slots [sig, mu_0..mu_{D-1}, p1, p2]
template <int D, int MASK>
SSDE_HD void ctcrw_finish(const CtcrwLane<D, MASK>& L, double* out /* 1 + 3 + D */) {
    out[0] = 0.5 * ((double)D * L.ld.value() + L.accq);
    out[1] = (MASK & DIR_SIG) ? L.ts.g : 0.0;
    for (int a = 0; a < D; a++) out[2 + a] = (MASK & DIR_MU) ? L.gmu[a] : 0.0;
    out[2 + D] = (MASK & DIR_P1) ? L.tt.g : 0.0;
    out[3 + D] = (MASK & DIR_P2) ? L.tn.g : 0.0;
}

// ---------------------------------------------------------------------------------------
// OU_SSM / BM_SSM, isotropic: state x[a], covariance p (scalar), transition a' = t a + c_a.
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

template <int D>
struct ScalTan {
    double p, x[D], g;
    SSDE_HD void init() { p = g = 0.0; for (int a = 0; a < D; a++) x[a] = 0.0; }
};

template <int D, int MASK>
struct ScalLane {
    double x[D], p;
    LogAcc ld;
    double accq;
    ScalTan<D> ts, t1, t2;
    double mx[D], gmu[D];
    SSDE_HD void init(const double* a0x, double p0) {
        for (int a = 0; a < D; a++) { x[a] = a0x[a]; mx[a] = gmu[a] = 0.0; }
        p = p0;
        ld.init();
        accq = 0.0;
        ts.init(); t1.init(); t2.init();
    }
    SSDE_HD void reset_acc() {
        ld.init();
        accq = 0.0;
        ts.g = t1.g = t2.g = 0.0;
        for (int a = 0; a < D; a++) gmu[a] = 0.0;
    }
    static constexpr int NSTATE = 4 * (D + 1) + D;
    SSDE_HD static void dump_tan(const ScalTan<D>& t, bool on, double* o, int& k) {
        o[k++] = on ? t.p : 0.0;
        for (int a = 0; a < D; a++) o[k++] = on ? t.x[a] : 0.0;
    }
    SSDE_HD void dump(double* o) const {
        int k = 0;
        for (int a = 0; a < D; a++) o[k++] = x[a];
        o[k++] = p;
        dump_tan(ts, (MASK & DIR_SIG) != 0, o, k);
        dump_tan(t1, (MASK & DIR_P1) != 0, o, k);
        dump_tan(t2, (MASK & DIR_P2) != 0, o, k);
        for (int a = 0; a < D; a++) o[k++] = (MASK & DIR_MU) ? mx[a] : 0.0;
    }
};

template <int D>
SSDE_HD void scal_tan_step(ScalTan<D>& t, const double* x, double p, const ScalTrans& tr, double iF, double su2,
                           const double* u, double k, double tp, const double* mu, double dh, double dt_, double db,
                           double dq, bool upd) {
    const double dF = t.p + dh;
    const double diF = -iF * iF * dF;
    double sud = 0.0;
    double du[D];
    for (int a = 0; a < D; a++) { du[a] = upd ? -t.x[a] : 0.0; sud += u[a] * du[a]; }
    t.g += 0.5 * ((double)D * dF * iF + diF * su2) + iF * sud;
    const double dtp = dt_ * p + tr.t * t.p;
    const double dk = dtp * iF + tp * diF;
    for (int a = 0; a < D; a++)
        t.x[a] = dt_ * x[a] + tr.t * t.x[a] + dk * u[a] + k * du[a] + db * mu[a];
    t.p = dtp * (tr.t - k) + tp * (dt_ - dk) + dq;
}

// MODEL is M_OU_SSM or M_BM_SSM: both take detF = exp(logdet F) > 0 unless F == 0
// (nllk_ou_ssm.hpp:190-195, nllk_bm_ssm.hpp:152-157) and keep the drift in every branch (Q3).
template <int D, int MASK, bool HAS_P2>
SSDE_HD void scal_step(ScalLane<D, MASK>& L, const ScalTrans& tr, double h, const double* mu, const double* y,
                       bool na) {
    const double F = L.p + h;
    const bool upd = !na && (fabs(F) > 0.0);
    const double iF = upd ? rcp(F) : 0.0;
    double u[D];
    double su2 = 0.0;
    for (int a = 0; a < D; a++) { u[a] = upd ? y[a] - L.x[a] : 0.0; su2 += u[a] * u[a]; }
    L.ld.mul(upd ? F : 1.0);
    L.accq += iF * su2;
    const double tp = tr.t * L.p;
    const double k = tp * iF;
    if (MASK & DIR_SIG) scal_tan_step<D>(L.ts, L.x, L.p, tr, iF, su2, u, k, tp, mu, 2.0 * h, 0.0, 0.0, 0.0, upd);
    if (MASK & DIR_P1) scal_tan_step<D>(L.t1, L.x, L.p, tr, iF, su2, u, k, tp, mu, 0.0, tr.dt_, tr.db, tr.dq, upd);
    if (HAS_P2 && (MASK & DIR_P2)) scal_tan_step<D>(L.t2, L.x, L.p, tr, iF, su2, u, k, tp, mu, 0.0, 0.0, 0.0, tr.q, upd);
    if (MASK & DIR_MU) {
        for (int a = 0; a < D; a++) {
            const double du = upd ? -L.mx[a] : 0.0;
            L.gmu[a] += iF * u[a] * du;
            L.mx[a] = tr.t * L.mx[a] + k * du + tr.b;
        }
    }
    for (int a = 0; a < D; a++) L.x[a] = tr.t * L.x[a] + k * u[a] + tr.b * mu[a];
    L.p = tp * (tr.t - k) + tr.q;
}

template <int D, int MASK>
SSDE_HD void scal_finish(const ScalLane<D, MASK>& L, double* out /* 1 + 3 + D */) {
    out[0] = 0.5 * ((double)D * L.ld.value() + L.accq);
    out[1] = (MASK & DIR_SIG) ? L.ts.g : 0.0;
    for (int a = 0; a < D; a++) out[2 + a] = (MASK & DIR_MU) ? L.gmu[a] : 0.0;
    out[2 + D] = (MASK & DIR_P1) ? L.t1.g : 0.0;
    out[3 + D] = (MASK & DIR_P2) ? L.t2.g : 0.0;
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

}  // namespace ssde
#endif
