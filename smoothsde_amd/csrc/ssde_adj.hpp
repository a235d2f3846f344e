// ssde_adj.hpp -- the REVERSE (adjoint) sweep of the isotropic Kalman lanes with row-varying coefficients.
//
// The model the reference exists for: the SDE parameters are smooth in covariates, par_mat.row(i) = X_fe coeff_fe + X_re coeff_re
// (nllk_ctcrw.hpp:143-156, nllk_ou_ssm.hpp:113-124, nllk_bm_ssm.hpp:98-108), and the filter loop (nllk_ctcrw.hpp:203-241,
// nllk_ou_ssm.hpp:171-207, nllk_bm_ssm.hpp:135-169) reads row i's parameters.  The gradient with respect to the coefficient of design
// column k is sum_i X_k(i) g_j(i) with g_j(i) = d nllk / d par_mat(i, j) -- ONE scalar per row and SDE parameter whatever the number
// of columns.  The forward-tangent lanes (k_iso_colvar_lanes.hpp) carry a 3 + 2 d-double tangent per COLUMN (~40 fp64 instructions
// per column and row); here the row's g_j come from one backward recursion over the adjoint of the filter state
//     lambda_i = Lin_i' lambda_{i+1} + d l_i / d(state_i)            g_j(i) = <lambda_{i+1}, d step_i / d par_j>
// (~75 instructions per row for CTCRW with two response columns), and a column costs one FMA per row and parameter it feeds.
// This is what TMB's reverse sweep over the tape does for the reference (R/sde.R:656-658, src/init.c:8); nothing is taped here:
// the forward pass leaves a RECORD per row (the filter state entering the row, the gain's reciprocal and the transition with its
// log tau derivatives) and the backward pass is the hand-derived transpose of the step, checked against the forward tangents, the
// oracle's dual numbers and finite differences (tests/test_kernel_math_host.py, tests/test_gpu_adjoint.py).
//
// Written as __host__ __device__ inline code: k_iso_adj.hip runs it lane = track; tests/hostsim compiles it with g++.
#ifndef SSDE_ADJ_HPP
#define SSDE_ADJ_HPP
#include <type_traits>

#include "ssde_math.hpp"

namespace ssde {

// what a row hands back: d nllk_row / d (par[d], par[d + 1], mu_a, h) at THIS row's parameters
template <int D>
struct AdjRowGrad {
    double g1, g2, gmu[D], gh;
};

// exp(x) for the row transitions.  The library's exp is one dependent Horner chain of ~12 fp64 FMAs, and three of them per row
// (tau, nu, e^{-beta dt}) were most of what a lone wave per SIMD waited for (8-9 cycles per dependent instruction against the pipe's
// 4: the row took 2200 cycles for 250 instructions).  Same range reduction (x = k ln 2 + r, |r| <= ln 2 / 2), the degree-13 Taylor
// polynomial of e^r (truncation 4e-18 relative) by Estrin's scheme: four levels of independent FMAs instead of thirteen dependent
// ones.  2 ulp at worst (tests/test_kernel_math_host.py); overflow / underflow through ldexp.
SSDE_HD double adj_exp(double x) {
    x = fmin(fmax(x, -1000.0), 1000.0);                         // (e^1000 = inf, e^-1000 = 0 through ldexp; keeps k an int)
    const double k = rint(x * 1.4426950408889634074);
    double r = fma(-k, 6.93147180369123816490e-01, x);          // ln 2, high and low parts (the high part has 32 trailing zero bits)
    r = fma(-k, 1.90821492927058770002e-10, r);
    const double r2 = r * r, r4 = r2 * r2, r8 = r4 * r4;
    const double a0 = fma(r, 1.0, 1.0);
    const double a1 = fma(r, 1.0 / 6.0, 0.5);
    const double a2 = fma(r, 1.0 / 120.0, 1.0 / 24.0);
    const double a3 = fma(r, 1.0 / 5040.0, 1.0 / 720.0);
    const double a4 = fma(r, 1.0 / 362880.0, 1.0 / 40320.0);
    const double a5 = fma(r, 1.0 / 39916800.0, 1.0 / 3628800.0);
    const double a6 = fma(r, 1.0 / 6227020800.0, 1.0 / 479001600.0);
    const double b0 = fma(r2, a1, a0), b1 = fma(r2, a3, a2), b2 = fma(r2, a5, a4);
    const double c0 = fma(r4, b1, b0), c1 = fma(r4, a6, b2);
    const double p = fma(r8, c1, c0);
    return ldexp(p, (int)k);
}

// makeT/Q/B_ctcrw (nllk_ctcrw.hpp:45-91) from the row's linear predictors p1 = log tau, p2 = log nu (:152-156).  sigma^2 / beta
// = 4 nu^2 / pi does not depend on tau: formed directly (no square root; ctcrw_trans squares sigma again).
SSDE_HD void adj_ctcrw_trans(double dt, double p1, double p2, CtcrwTrans& tr) {
    const double tau = adj_exp(p1), nu = adj_exp(p2);
    const double beta = rcp(tau);
    const double A = (4.0 / M_PI) * nu * nu;
    const double e = adj_exp(-beta * dt);
    const double e2 = e * e, ome = 1.0 - e, At = A * tau, hte = 0.5 * tau * (1.0 - e2);
    tr.e = e;
    tr.t12 = ome * tau;
    tr.b1 = dt - tr.t12;
    tr.b2 = ome;
    const double G = dt - 2.0 * tr.t12 + hte;
    tr.q11 = At * G;
    tr.q12 = 0.5 * At * ome * ome;
    tr.q22 = 0.5 * A * (1.0 - e2);
    const double edt = e * dt, bdt = beta * dt;
    tr.de = e * bdt;
    tr.dt12 = tr.t12 - edt;
    const double dG = -2.0 * tr.dt12 + hte - e2 * dt;
    tr.dq11 = fma(At, dG, tr.q11);
    tr.dq12 = fma(-A * ome, edt, tr.q12);
    tr.dq22 = -A * e2 * bdt;
    tr.e2 = e2; tr.dt12x2 = 2.0 * tr.dt12; tr.dt12e = tr.dt12 * e; tr.edex2 = 2.0 * e * tr.de;
}

// ---- CTCRW, H = h I, block-identical P0: state (x_a, v_a), covariance (p11, p12, p22) shared by the dimensions -----------------
template <int D>
struct AdjCtcrw {
    static constexpr int SD = 2 * D;
    static constexpr int NST = 2 * D + 3;                      // doubles of the state (= of its adjoint)
    static constexpr int NF = 4 + 2 * D + 10;                  // doubles of a row's record
    typedef CtcrwTrans Trans;
    double x[D], v[D], p11, p12, p22;

    SSDE_HD void init(const double* a0, const double* p0) {
        for (int a = 0; a < D; a++) { x[a] = a0[2 * a]; v[a] = a0[2 * a + 1]; }
        p11 = p0[0]; p12 = p0[1]; p22 = p0[2];
    }
    template <int RS>
    SSDE_HD void put(double* o) const {
        int n = 0;
        for (int a = 0; a < D; a++) { o[(n++) * RS] = x[a]; o[(n++) * RS] = v[a]; }
        o[(n++) * RS] = p11; o[(n++) * RS] = p12; o[(n++) * RS] = p22;
    }
    template <int RS>
    SSDE_HD void get(const double* o) {
        int n = 0;
        for (int a = 0; a < D; a++) { x[a] = o[(n++) * RS]; v[a] = o[(n++) * RS]; }
        p11 = o[(n++) * RS]; p12 = o[(n++) * RS]; p22 = o[(n++) * RS];
    }
    static SSDE_HD void trans(double dt, double p1, double p2, Trans& tr) { adj_ctcrw_trans(dt, p1, p2, tr); }

    // One row (nllk_ctcrw.hpp:206-241; the arrangement of CvPrimalCtcrw::step): score y unless NA, predict over the row's interval.
    // REC: leave the row's record -- what bwd() needs of the forward pass -- in rec[f * RS].
    template <bool REC, int RS>
    SSDE_HD void fwd(const Trans& tr, double h, const double* mu, const double* y, bool na, LogAcc& ld, double& accq, double* rec) {
        const double F = p11 + h;
        const double detF = (D == 1) ? F : F * F;                  // :16-19, 223
        const double e = tr.e, t12 = tr.t12;
        const bool upd = !na && !(detF <= 0.0);                    // :214, 226
        const double updf = upd ? 1.0 : 0.0;
        const double Fe = upd ? F : 1.0;
        const double iF = rcp(Fe) * updf;
        ld.mul(Fe);
        const bool bm = na || upd;                                 // Q3 (:226-228): detF <= 0 predicts without B mu
        const double a_ = fma(h, iF, 1.0 - updf);
        const double kf1 = p11 * iF, kf2 = p12 * iF;
        const double f11 = p11 * a_, f12 = p12 * a_, f22 = fma(-p12, kf2, p22);
        const double m = fma(t12, f22, f12);
        const double k1 = fma(t12, kf2, kf1), k2 = e * kf2;
        if (REC) {
            int n = 0;
            // (iF == -0.0 flags the detF <= 0 branch: no update AND no drift)
            rec[(n++) * RS] = bm ? iF : -0.0; rec[(n++) * RS] = p11; rec[(n++) * RS] = p12; rec[(n++) * RS] = p22;
        }
        double su2 = 0.0;
        for (int a = 0; a < D; a++) {
            const double ys = upd ? y[a] : x[a];
            const double u = ys - x[a];
            su2 = fma(u, u, su2);
            const double mue = bm ? mu[a] : 0.0;
            if (REC) { rec[(4 + a) * RS] = u; rec[(4 + D + a) * RS] = fma(kf2, u, v[a]); }
            const double nx = fma(tr.b1, mue, fma(k1, u, fma(t12, v[a], x[a])));      // a = T a + K u + B mu (:238)
            const double nv = fma(tr.b2, mue, fma(k2, u, e * v[a]));
            x[a] = nx; v[a] = nv;
        }
        accq = fma(iF, su2, accq);
        p11 = fma(t12, f12 + m, f11) + tr.q11;                     // P = T P~ T' + Q (:240-241)
        p12 = fma(e, m, tr.q12);
        p22 = fma(tr.e2, f22, tr.q22);
    }
    // the transition's share of a row's record (written apart from the state's: the transitions of several rows do not depend on
    // one another and are formed together, ahead of the filter steps that do), and what the forward step needs of it back
    template <int RS>
    static SSDE_HD void put_trans(double* rec, const Trans& tr) {
        int n = 4 + 2 * D;
        rec[(n++) * RS] = tr.e; rec[(n++) * RS] = tr.t12; rec[(n++) * RS] = tr.de; rec[(n++) * RS] = tr.dt12;
        rec[(n++) * RS] = tr.dq11; rec[(n++) * RS] = tr.dq12; rec[(n++) * RS] = tr.dq22;
        rec[(n++) * RS] = tr.q11; rec[(n++) * RS] = tr.q12; rec[(n++) * RS] = tr.q22;
    }
    template <int RS>
    static SSDE_HD void get_trans(const double* rec, double dt, Trans& tr) {
        const int n = 4 + 2 * D;
        tr.e = rec[n * RS]; tr.t12 = rec[(n + 1) * RS]; tr.q11 = rec[(n + 7) * RS]; tr.q12 = rec[(n + 8) * RS]; tr.q22 = rec[(n + 9) * RS];
        tr.b1 = dt - tr.t12; tr.b2 = 1.0 - tr.e; tr.e2 = tr.e * tr.e;
    }

    // the adjoint of the state: lambda = d (nllk of the rows from here on) / d (x_a, v_a, p11, p12, p22)
    struct Adj {
        double bx[D], bv[D], b11, b12, b22;
        SSDE_HD void zero() { for (int a = 0; a < D; a++) bx[a] = bv[a] = 0.0; b11 = b12 = b22 = 0.0; }
        template <int RS>
        SSDE_HD void put(double* o) const {
            int n = 0;
            for (int a = 0; a < D; a++) { o[(n++) * RS] = bx[a]; o[(n++) * RS] = bv[a]; }
            o[(n++) * RS] = b11; o[(n++) * RS] = b12; o[(n++) * RS] = b22;
        }
    };
    // One row backwards: L enters as the adjoint of the state AFTER the row's prediction and leaves as the adjoint of the state
    // ENTERING the row; g = the row's parameter derivatives.  b1 / b2: makeB of the row (dt - t12, 1 - e); mu: the row's drift.
    template <int RS>
    static SSDE_HD void bwd(Adj& L, const double* rec, double h, const double* mu, double dt, AdjRowGrad<D>& g) {
        int n = 0;
        const double iFr = rec[(n++) * RS], p11 = rec[(n++) * RS], p12 = rec[(n++) * RS], p22 = rec[(n++) * RS];
        double u[D], vf[D];
        for (int a = 0; a < D; a++) { u[a] = rec[(4 + a) * RS]; vf[a] = rec[(4 + D + a) * RS]; }
        n = 4 + 2 * D;
        const double e = rec[(n++) * RS], t12 = rec[(n++) * RS], de = rec[(n++) * RS], dt12 = rec[(n++) * RS];
        const double dq11 = rec[(n++) * RS], dq12 = rec[(n++) * RS], dq22 = rec[(n++) * RS];
        const double q11 = rec[(n++) * RS], q12 = rec[(n++) * RS], q22 = rec[(n++) * RS];
#if defined(__HIP_DEVICE_COMPILE__)
        const bool nodrift = (unsigned long long)__double_as_longlong(iFr) == 0x8000000000000000ull;
#else
        uint64_t bits; memcpy(&bits, &iFr, 8);
        const bool nodrift = bits == 0x8000000000000000ull;
#endif
        const double iF = nodrift ? 0.0 : iFr;
        const double a_ = fma(h, iF, iF != 0.0 ? 0.0 : 1.0);       // (1 on a row that was not scored)
        const double kf2 = p12 * iF;
        const double f22 = fma(-p12, kf2, p22), f12 = p12 * a_;
        const double m = fma(t12, f22, f12);
        const double b1 = dt - t12, b2 = 1.0 - e;
        // the prediction: x' = xf + t12 vf + b1 mu, v' = e vf + b2 mu, P' = T P~ T' + Q
        double sxv = 0.0, svv = 0.0, sxm = 0.0, svm = 0.0, kf1b = 0.0, kf2b = 0.0, su2 = 0.0;
        double bvf[D];
        for (int a = 0; a < D; a++) {
            const double mue = nodrift ? 0.0 : mu[a];
            sxv = fma(L.bx[a], vf[a], sxv); svv = fma(L.bv[a], vf[a], svv);
            sxm = fma(L.bx[a], mue, sxm); svm = fma(L.bv[a], mue, svm);
            g.gmu[a] = nodrift ? 0.0 : fma(b1, L.bx[a], b2 * L.bv[a]);
            bvf[a] = fma(t12, L.bx[a], e * L.bv[a]);
            kf1b = fma(L.bx[a], u[a], kf1b); kf2b = fma(bvf[a], u[a], kf2b);
            su2 = fma(u[a], u[a], su2);
        }
        const double ef22 = e * f22;
        const double t12b = fma(2.0 * L.b11, m, fma(L.b12, ef22, sxv));
        const double eb = fma(L.b12, m, fma(2.0 * L.b22, ef22, svv));
        const double w = fma(t12, L.b11, e * L.b12);
        const double f11b = L.b11, f12b = fma(t12, L.b11, w), f22b = fma(t12, w, e * e * L.b22);
        g.g1 = fma(de, eb - svm, fma(dt12, t12b - sxm, fma(L.b11, dq11, fma(L.b12, dq12, L.b22 * dq22))));
        g.g2 = 2.0 * fma(L.b11, q11, fma(L.b12, q12, L.b22 * q22));
        // the update: xf = x + kf1 u, vf = v + kf2 u, P~ = (p11 a, p12 a, p22 - p12 kf2), l = (D log F + iF sum u^2) / 2
        const double gF = iF * fma(-0.5 * iF, su2, 0.5 * (double)D);
        const double c = fma(f11b, p11, f12b * p12);
        const double iFb = fma(h, c, fma(-f22b * p12, p12, fma(kf1b, p11, kf2b * p12)));
        const double Fb = fma(-iF * iF, iFb, gF);
        g.gh = fma(c, iF, Fb);
        L.b11 = fma(f11b, a_, fma(kf1b, iF, Fb));
        L.b12 = fma(f12b, a_, fma(-2.0 * f22b, kf2, kf2b * iF));
        L.b22 = f22b;
        for (int a = 0; a < D; a++) {
            L.bx[a] = fma(a_, L.bx[a], fma(-kf2, bvf[a], -iF * u[a]));
            L.bv[a] = bvf[a];
        }
    }
};

// ---- OU_SSM / BM_SSM, H = h I, P0 = p0 I: state x_a, covariance p ----------------------------------------------------------
// a' = t a + b mu_a (nllk_ou_ssm.hpp:174-207: t = e^{-dt/tau}, b = 1 - t, q = kappa (1 - e^{-2 dt / tau});
// nllk_bm_ssm.hpp:138-169: t = 1, b = dt, q = sigma^2 dt); the drift stays in every branch (Q3)
template <int D, bool HAS_P2>
struct AdjScal {
    static constexpr int SD = D;
    static constexpr int NST = D + 1;
    static constexpr int NF = 2 + 2 * D + 5;
    typedef ScalTrans Trans;
    double x[D], p;

    SSDE_HD void init(const double* a0, const double* p0) {
        for (int a = 0; a < D; a++) x[a] = a0[a];
        p = p0[0];
    }
    template <int RS>
    SSDE_HD void put(double* o) const {
        for (int a = 0; a < D; a++) o[a * RS] = x[a];
        o[D * RS] = p;
    }
    template <int RS>
    SSDE_HD void get(const double* o) {
        for (int a = 0; a < D; a++) x[a] = o[a * RS];
        p = o[D * RS];
    }
    static SSDE_HD void trans(double dt, double p1, double p2, Trans& tr) {
        if (HAS_P2) {                                              // nllk_ou_ssm.hpp:121-124; makeT/B/Q :30-69 (ou_trans with adj_exp)
            const double tau = adj_exp(p1), kappa = adj_exp(p2);
            const double z = dt * rcp(tau);
            const double e = adj_exp(-z), e2 = e * e;
            tr.t = e; tr.b = 1.0 - e; tr.q = kappa * (1.0 - e2);
            tr.dt_ = e * z; tr.db = -tr.dt_; tr.dq = -2.0 * kappa * e2 * z;
        } else {                                                   // nllk_bm_ssm.hpp:106-108; makeQ :28-36
            const double sg = adj_exp(p1);
            tr.t = 1.0; tr.b = dt; tr.q = sg * sg * dt; tr.dt_ = 0.0; tr.db = 0.0; tr.dq = 2.0 * tr.q;
        }
    }
    template <bool REC, int RS>
    SSDE_HD void fwd(const Trans& tr, double h, const double* mu, const double* y, bool na, LogAcc& ld, double& accq, double* rec) {
        const double F = p + h;
        const double t = HAS_P2 ? tr.t : 1.0;
        const bool upd = !na && !(fabs(F) <= 0.0);                 // nllk_ou_ssm.hpp:190-195, nllk_bm_ssm.hpp:152-157
        const double updf = upd ? 1.0 : 0.0;
        const double Fe = upd ? F : 1.0;
        const double iF = rcp(Fe) * updf;
        ld.mul(Fe);
        const double a_ = fma(h, iF, 1.0 - updf), b = p * iF;
        const double k = t * b, tc = t * t * a_;
        if (REC) { rec[0] = iF; rec[RS] = p; }
        double su2 = 0.0;
        for (int a = 0; a < D; a++) {
            const double ys = upd ? y[a] : x[a];
            const double u = ys - x[a];
            su2 = fma(u, u, su2);
            if (REC) { rec[(2 + a) * RS] = u; rec[(2 + D + a) * RS] = fma(b, u, x[a]); }
            x[a] = fma(tr.b, mu[a], fma(k, u, HAS_P2 ? t * x[a] : x[a]));
        }
        accq = fma(iF, su2, accq);
        p = fma(tc, p, tr.q);
    }
    template <int RS>
    static SSDE_HD void put_trans(double* rec, const Trans& tr) {
        int n = 2 + 2 * D;
        rec[(n++) * RS] = HAS_P2 ? tr.t : 1.0; rec[(n++) * RS] = tr.b; rec[(n++) * RS] = tr.q; rec[(n++) * RS] = tr.dt_; rec[(n++) * RS] = tr.dq;
    }
    template <int RS>
    static SSDE_HD void get_trans(const double* rec, double /*dt*/, Trans& tr) {
        const int n = 2 + 2 * D;
        tr.t = rec[n * RS]; tr.b = rec[(n + 1) * RS]; tr.q = rec[(n + 2) * RS];
    }
    struct Adj {
        double bx[D], bp;
        SSDE_HD void zero() { for (int a = 0; a < D; a++) bx[a] = 0.0; bp = 0.0; }
        template <int RS>
        SSDE_HD void put(double* o) const {
            for (int a = 0; a < D; a++) o[a * RS] = bx[a];
            o[D * RS] = bp;
        }
    };
    template <int RS>
    static SSDE_HD void bwd(Adj& L, const double* rec, double h, const double* mu, double /*dt*/, AdjRowGrad<D>& g) {
        const double iF = rec[0], p = rec[RS];
        double u[D], xf[D];
        for (int a = 0; a < D; a++) { u[a] = rec[(2 + a) * RS]; xf[a] = rec[(2 + D + a) * RS]; }
        int n = 2 + 2 * D;
        const double t = rec[(n++) * RS], btr = rec[(n++) * RS], q = rec[(n++) * RS], dt_ = rec[(n++) * RS], dq = rec[(n++) * RS];
        const double a_ = fma(h, iF, iF != 0.0 ? 0.0 : 1.0);       // (1 on a row that was not scored)
        const double pf = p * a_;
        // the prediction: x' = t xf + b mu, p' = t^2 pf + q
        double sxx = 0.0, sxm = 0.0, bb = 0.0, su2 = 0.0, bxf[D];
        for (int a = 0; a < D; a++) {
            sxx = fma(L.bx[a], xf[a], sxx); sxm = fma(L.bx[a], mu[a], sxm);
            g.gmu[a] = btr * L.bx[a];
            bxf[a] = t * L.bx[a];
            bb = fma(bxf[a], u[a], bb);
            su2 = fma(u[a], u[a], su2);
        }
        const double tb = fma(2.0 * t * pf, L.bp, sxx);
        const double pfb = t * t * L.bp;
        // log tau (OU): dt = dt_, db = -dt_, dq;  log sigma (BM): dq = 2 q only
        g.g1 = HAS_P2 ? fma(dt_, tb - sxm, L.bp * dq) : L.bp * dq;
        g.g2 = HAS_P2 ? L.bp * q : 0.0;                            // log kappa: dq = q
        // the update: xf = x + b u, pf = p h iF, b = p iF, l = (D log F + iF sum u^2) / 2
        const double gF = iF * fma(-0.5 * iF, su2, 0.5 * (double)D);
        const double c = pfb * p;
        const double iFb = fma(h, c, bb * p);
        const double Fb = fma(-iF * iF, iFb, gF);
        g.gh = fma(c, iF, Fb);
        L.bp = fma(pfb, a_, fma(bb, iF, Fb));
        for (int a = 0; a < D; a++) L.bx[a] = fma(a_, bxf[a], -iF * u[a]);
    }
};

// ---- two response columns with a FULL covariance: a per-row measurement covariance H_i (H_array, nllk_ctcrw.hpp:203-205) couples the
// dimensions, and / or P0 is not block-identical.  State a (sd), P symmetric (sd (sd + 1) / 2 numbers, row-major upper triangle);
// Z picks the positions (components 0 and 2 of CTCRW's (x0, v0, x1, v1); the identity for OU_SSM / BM_SSM).  The step in its filtered form
//     w = F^-1 u,  af = a + P Z' w,  Pf = P - P Z' F^-1 Z P,  a' = T af + B mu,  P' = T Pf T' + Q        (nllk_ctcrw.hpp:219-242)
// and its transpose; the adjoint of P is kept as the symmetric matrix G with d l = sum_rc G_rc dP_rc over symmetric dP.
template <int MODEL>
struct AdjFull {
    static constexpr int D = 2;
    static constexpr bool CT = MODEL == M_CTCRW, HAS_P2 = MODEL != M_BM_SSM;
    static constexpr int SD = CT ? 4 : 2, NP = SD * (SD + 1) / 2, NST = SD + NP;
    static constexpr int NTR = CT ? 10 : 5, NF = NST + 1 + NTR;
    typedef typename std::conditional<CT, CtcrwTrans, ScalTrans>::type Trans;
    double a[SD], p[NP];
    static SSDE_HD constexpr int z(int i) { return CT ? 2 * i : i; }
    static SSDE_HD constexpr int sidx(int i, int j) { return i <= j ? i * SD - i * (i - 1) / 2 + (j - i) : j * SD - j * (j - 1) / 2 + (i - j); }

    SSDE_HD void init(const double* a0, const double* p0f /* sd x sd, column-major */) {
        for (int i = 0; i < SD; i++) a[i] = a0[i];
        for (int i = 0; i < SD; i++)
            for (int j = i; j < SD; j++) p[sidx(i, j)] = p0f[i + SD * j];
    }
    template <int RS>
    SSDE_HD void put(double* o) const {
        for (int i = 0; i < SD; i++) o[i * RS] = a[i];
        for (int i = 0; i < NP; i++) o[(SD + i) * RS] = p[i];
    }
    template <int RS>
    SSDE_HD void get(const double* o) {
        for (int i = 0; i < SD; i++) a[i] = o[i * RS];
        for (int i = 0; i < NP; i++) p[i] = o[(SD + i) * RS];
    }
    static SSDE_HD void trans(double dt, double p1, double p2, Trans& tr) {
        if constexpr (CT) adj_ctcrw_trans(dt, p1, p2, tr);
        else AdjScal<2, HAS_P2>::trans(dt, p1, p2, tr);
    }
    template <int RS>
    static SSDE_HD void put_trans(double* rec, const Trans& tr) {
        int n = NST + 1;
        if constexpr (CT) {
            rec[(n++) * RS] = tr.e; rec[(n++) * RS] = tr.t12; rec[(n++) * RS] = tr.de; rec[(n++) * RS] = tr.dt12;
            rec[(n++) * RS] = tr.dq11; rec[(n++) * RS] = tr.dq12; rec[(n++) * RS] = tr.dq22;
            rec[(n++) * RS] = tr.q11; rec[(n++) * RS] = tr.q12; rec[(n++) * RS] = tr.q22;
        } else {
            rec[(n++) * RS] = HAS_P2 ? tr.t : 1.0; rec[(n++) * RS] = tr.b; rec[(n++) * RS] = tr.q; rec[(n++) * RS] = tr.dt_; rec[(n++) * RS] = tr.dq;
        }
    }
    template <int RS>
    static SSDE_HD void get_trans(const double* rec, double dt, Trans& tr) {
        const int n = NST + 1;
        if constexpr (CT) {
            tr.e = rec[n * RS]; tr.t12 = rec[(n + 1) * RS]; tr.q11 = rec[(n + 7) * RS]; tr.q12 = rec[(n + 8) * RS]; tr.q22 = rec[(n + 9) * RS];
            tr.b1 = dt - tr.t12; tr.b2 = 1.0 - tr.e; tr.e2 = tr.e * tr.e;
        } else {
            tr.t = rec[n * RS]; tr.b = rec[(n + 1) * RS]; tr.q = rec[(n + 2) * RS];
        }
    }
    // v <- T v, v <- T' v for one vector of the state's layout
    static SSDE_HD void mulT(const Trans& tr, const double* v, double* o) {
        if constexpr (CT) { for (int d = 0; d < D; d++) { o[2 * d] = fma(tr.t12, v[2 * d + 1], v[2 * d]); o[2 * d + 1] = tr.e * v[2 * d + 1]; } }
        else { const double t = HAS_P2 ? tr.t : 1.0; for (int i = 0; i < SD; i++) o[i] = t * v[i]; }
    }
    static SSDE_HD void mulTt(const Trans& tr, const double* v, double* o) {
        if constexpr (CT) { for (int d = 0; d < D; d++) { o[2 * d] = v[2 * d]; o[2 * d + 1] = fma(tr.t12, v[2 * d], tr.e * v[2 * d + 1]); } }
        else { const double t = HAS_P2 ? tr.t : 1.0; for (int i = 0; i < SD; i++) o[i] = t * v[i]; }
    }
    // what both directions form from (a, P, H, y): the update's pieces.  idet: 1 / det F, 0 on a row that is not scored
    struct Upd {
        double P[SD][SD], u[D], w[D], Fi[3], M[SD][D], MFi[SD][D], af[SD], Pf[SD][SD];
    };
    SSDE_HD void update_pieces(const double* H, const double* y, double idet, Upd& U) const {
        for (int i = 0; i < SD; i++)
            for (int j = 0; j < SD; j++) U.P[i][j] = p[sidx(i, j)];
        const double F00 = U.P[z(0)][z(0)] + H[0], F01 = U.P[z(0)][z(1)] + H[1], F11 = U.P[z(1)][z(1)] + H[2];
        U.Fi[0] = F11 * idet; U.Fi[1] = -F01 * idet; U.Fi[2] = F00 * idet;
        const bool upd = idet != 0.0;
        for (int i = 0; i < D; i++) U.u[i] = upd ? y[i] - a[z(i)] : 0.0;
        U.w[0] = fma(U.Fi[0], U.u[0], U.Fi[1] * U.u[1]); U.w[1] = fma(U.Fi[1], U.u[0], U.Fi[2] * U.u[1]);
        for (int r = 0; r < SD; r++) {
            U.M[r][0] = U.P[r][z(0)]; U.M[r][1] = U.P[r][z(1)];
            U.MFi[r][0] = fma(U.M[r][0], U.Fi[0], U.M[r][1] * U.Fi[1]); U.MFi[r][1] = fma(U.M[r][0], U.Fi[1], U.M[r][1] * U.Fi[2]);
            U.af[r] = fma(U.M[r][0], U.w[0], fma(U.M[r][1], U.w[1], a[r]));
        }
        for (int r = 0; r < SD; r++)
            for (int c = r; c < SD; c++) U.Pf[r][c] = U.Pf[c][r] = U.P[r][c] - fma(U.MFi[r][0], U.M[c][0], U.MFi[r][1] * U.M[c][1]);
    }
    // H = (H00, H01, H11) of the row (sigma_obs^2 I without H_array)
    template <bool REC, int RS>
    SSDE_HD void fwd(const Trans& tr, const double* H, const double* mu, const double* y, bool na, LogAcc& ld, double& accq, double* rec) {
        const double F00 = p[sidx(z(0), z(0))] + H[0], F01 = p[sidx(z(0), z(1))] + H[1], F11 = p[sidx(z(1), z(1))] + H[2];
        const double detF = fma(F00, F11, -F01 * F01);             // det(): nllk_ctcrw.hpp:16-19
        const bool upd = !na && (CT ? !(detF <= 0.0) : !(fabs(detF) <= 0.0));     // :214, 226; nllk_ou_ssm.hpp:190-195
        const double dete = upd ? (CT ? detF : fabs(detF)) : 1.0;
        const double idet = upd ? rcp(detF) : 0.0;
        ld.mul(dete);
        const bool bm = na || upd || !CT;                          // Q3: CTCRW's detF <= 0 branch predicts without B mu
        if (REC) {
            put<RS>(rec);
            rec[NST * RS] = bm ? idet : -0.0;
        }
        Upd U;
        update_pieces(H, y, idet, U);
        accq = fma(U.u[0], U.w[0], fma(U.u[1], U.w[1], accq));
        double Taf[SD];
        mulT(tr, U.af, Taf);
        for (int d = 0; d < D; d++) {
            const double mue = bm ? mu[d] : 0.0;
            if constexpr (CT) { a[2 * d] = fma(tr.b1, mue, Taf[2 * d]); a[2 * d + 1] = fma(tr.b2, mue, Taf[2 * d + 1]); }
            else a[d] = fma(tr.b, mue, Taf[d]);
        }
        // P' = T Pf T' + Q
        double G1[SD][SD];
        for (int c = 0; c < SD; c++) {
            double col[SD], o[SD];
            for (int r = 0; r < SD; r++) col[r] = U.Pf[r][c];
            mulT(tr, col, o);
            for (int r = 0; r < SD; r++) G1[r][c] = o[r];
        }
        for (int r = 0; r < SD; r++) {
            double o[SD];
            mulT(tr, G1[r], o);                                    // (row r of G1 T' = T (row r)')
            for (int c = r; c < SD; c++) p[sidx(r, c)] = o[c];
        }
        if constexpr (CT) {
            for (int d = 0; d < D; d++) { p[sidx(2 * d, 2 * d)] += tr.q11; p[sidx(2 * d, 2 * d + 1)] += tr.q12; p[sidx(2 * d + 1, 2 * d + 1)] += tr.q22; }
        } else {
            for (int r = 0; r < SD; r++) p[sidx(r, r)] += tr.q;
        }
    }
    struct Adj {
        double ba[SD], G[NP];                                      // G: the symmetric adjoint of P, stored like P
        SSDE_HD void zero() { for (int i = 0; i < SD; i++) ba[i] = 0.0; for (int i = 0; i < NP; i++) G[i] = 0.0; }
        template <int RS>
        SSDE_HD void put(double* o) const {
            for (int i = 0; i < SD; i++) o[i * RS] = ba[i];
            for (int i = 0; i < NP; i++) o[(SD + i) * RS] = G[i];
        }
    };
    template <int RS>
    static SSDE_HD void bwd(Adj& L, const double* rec, const double* H, const double* mu, const double* y, double dt, AdjRowGrad<2>& g) {
        AdjFull S;
        S.template get<RS>(rec);
        const double idr = rec[NST * RS];
#if defined(__HIP_DEVICE_COMPILE__)
        const bool nodrift = (unsigned long long)__double_as_longlong(idr) == 0x8000000000000000ull;
#else
        uint64_t bits; memcpy(&bits, &idr, 8);
        const bool nodrift = bits == 0x8000000000000000ull;
#endif
        const double idet = nodrift ? 0.0 : idr;
        Trans tr;
        get_trans<RS>(rec, dt, tr);
        Upd U;
        S.update_pieces(H, y, idet, U);
        double Gp[SD][SD];                                         // P-bar' as a full matrix
        for (int i = 0; i < SD; i++)
            for (int j = 0; j < SD; j++) Gp[i][j] = L.G[sidx(i, j)];
        // ---- the prediction: a' = T af + B mu, P' = T Pf T' + Q ----
        double baf[SD];
        mulTt(tr, L.ba, baf);
        const int n = NST + 1;
        if constexpr (CT) {
            const double de = rec[(n + 2) * RS], dt12 = rec[(n + 3) * RS], dq11 = rec[(n + 4) * RS], dq12 = rec[(n + 5) * RS], dq22 = rec[(n + 6) * RS];
            // T-bar = 2 P-bar' (T Pf): only its (x_d, v_d) and (v_d, v_d) entries move t12 and e
            double t12b = 0.0, eb = 0.0, q11b = 0.0, q12b = 0.0, q22b = 0.0, b1b = 0.0, b2b = 0.0;
            for (int d = 0; d < D; d++) {
                double colv[SD], R[SD];                            // column v_d of T Pf = T (column v_d of Pf)
                for (int r = 0; r < SD; r++) colv[r] = U.Pf[r][2 * d + 1];
                mulT(tr, colv, R);
                double tx = 0.0, tv = 0.0;
                for (int r = 0; r < SD; r++) { tx = fma(Gp[2 * d][r], R[r], tx); tv = fma(Gp[2 * d + 1][r], R[r], tv); }
                t12b += fma(2.0, tx, L.ba[2 * d] * U.af[2 * d + 1]);
                eb += fma(2.0, tv, L.ba[2 * d + 1] * U.af[2 * d + 1]);
                q11b += Gp[2 * d][2 * d]; q12b += 2.0 * Gp[2 * d][2 * d + 1]; q22b += Gp[2 * d + 1][2 * d + 1];
                const double mue = nodrift ? 0.0 : mu[d];
                b1b = fma(L.ba[2 * d], mue, b1b); b2b = fma(L.ba[2 * d + 1], mue, b2b);
                g.gmu[d] = nodrift ? 0.0 : fma(tr.b1, L.ba[2 * d], tr.b2 * L.ba[2 * d + 1]);
            }
            g.g1 = fma(de, eb - b2b, fma(dt12, t12b - b1b, fma(q11b, dq11, fma(q12b, dq12, q22b * dq22))));
            g.g2 = 2.0 * fma(q11b, tr.q11, fma(q12b, tr.q12, q22b * tr.q22));
        } else {
            const double t = HAS_P2 ? tr.t : 1.0, dt_ = rec[(n + 3) * RS], dq = rec[(n + 4) * RS];
            double tb = 0.0, qb = 0.0, bb = 0.0, gp = 0.0;
            for (int r = 0; r < SD; r++) {
                tb = fma(L.ba[r], U.af[r], tb); qb += Gp[r][r]; bb = fma(L.ba[r], mu[r], bb);
                for (int c = 0; c < SD; c++) gp = fma(Gp[r][c], U.Pf[r][c], gp);
                g.gmu[r] = tr.b * L.ba[r];
            }
            tb = fma(2.0 * t, gp, tb);
            g.g1 = HAS_P2 ? fma(dt_, tb - bb, qb * dq) : qb * dq;
            g.g2 = HAS_P2 ? qb * tr.q : 0.0;
        }
        // G-bar = T' P-bar' T
        double Gb[SD][SD];
        {
            double X[SD][SD];
            for (int c = 0; c < SD; c++) {                         // X = T' Gp (column by column)
                double col[SD], o[SD];
                for (int r = 0; r < SD; r++) col[r] = Gp[r][c];
                mulTt(tr, col, o);
                for (int r = 0; r < SD; r++) X[r][c] = o[r];
            }
            for (int r = 0; r < SD; r++) {                         // Gb = X T: row r of X T = (T' (row r)')'
                double o[SD];
                mulTt(tr, X[r], o);
                for (int c = 0; c < SD; c++) Gb[r][c] = o[c];
            }
        }
        // ---- the update: af = a + M w, Pf = P - M F^-1 M', l = (log det F + u' w) / 2, w = F^-1 u, M = P Z', F = Z P Z' + H ----
        double Mb[SD][D], wb[D] = {0.0, 0.0};
        for (int r = 0; r < SD; r++) {
            double s0 = 0.0, s1 = 0.0;
            for (int c = 0; c < SD; c++) { s0 = fma(Gb[r][c], U.MFi[c][0], s0); s1 = fma(Gb[r][c], U.MFi[c][1], s1); }
            Mb[r][0] = fma(baf[r], U.w[0], -2.0 * s0); Mb[r][1] = fma(baf[r], U.w[1], -2.0 * s1);
            wb[0] = fma(U.M[r][0], baf[r], wb[0]); wb[1] = fma(U.M[r][1], baf[r], wb[1]);
        }
        // F^-1-bar (symmetric) = - M' Gb M + u u' / 2 + sym(wb u')
        double GM[SD][D];
        for (int r = 0; r < SD; r++) {
            double s0 = 0.0, s1 = 0.0;
            for (int c = 0; c < SD; c++) { s0 = fma(Gb[r][c], U.M[c][0], s0); s1 = fma(Gb[r][c], U.M[c][1], s1); }
            GM[r][0] = s0; GM[r][1] = s1;
        }
        double Fib[3] = {0.0, 0.0, 0.0};
        for (int r = 0; r < SD; r++) { Fib[0] = fma(-U.M[r][0], GM[r][0], Fib[0]); Fib[1] = fma(-U.M[r][0], GM[r][1], Fib[1]); Fib[2] = fma(-U.M[r][1], GM[r][1], Fib[2]); }
        Fib[0] += fma(0.5 * U.u[0], U.u[0], wb[0] * U.u[0]);
        Fib[1] += fma(0.5 * U.u[0], U.u[1], 0.5 * fma(wb[0], U.u[1], wb[1] * U.u[0]));
        Fib[2] += fma(0.5 * U.u[1], U.u[1], wb[1] * U.u[1]);
        const double ub[D] = {U.w[0] + fma(U.Fi[0], wb[0], U.Fi[1] * wb[1]), U.w[1] + fma(U.Fi[1], wb[0], U.Fi[2] * wb[1])};
        // F-bar = F^-1 / 2 - F^-1 F^-1-bar F^-1   (on a row that is not scored F^-1 = 0: nothing)
        const double A00 = fma(U.Fi[0], Fib[0], U.Fi[1] * Fib[1]), A01 = fma(U.Fi[0], Fib[1], U.Fi[1] * Fib[2]);
        const double A10 = fma(U.Fi[1], Fib[0], U.Fi[2] * Fib[1]), A11 = fma(U.Fi[1], Fib[1], U.Fi[2] * Fib[2]);
        const double Fb00 = 0.5 * U.Fi[0] - fma(A00, U.Fi[0], A01 * U.Fi[1]);
        const double Fb01 = 0.5 * U.Fi[1] - fma(A00, U.Fi[1], A01 * U.Fi[2]);
        const double Fb11 = 0.5 * U.Fi[2] - fma(A10, U.Fi[1], A11 * U.Fi[2]);
        g.gh = Fb00 + Fb11;
        const bool upd = idet != 0.0;
        for (int r = 0; r < SD; r++) L.ba[r] = baf[r];
        if (upd) for (int i = 0; i < D; i++) L.ba[z(i)] -= ub[i];
        for (int r = 0; r < SD; r++)
            for (int c = r; c < SD; c++) {
                double v = Gb[r][c];
                if (upd) {
                    for (int j = 0; j < D; j++) {
                        if (c == z(j)) v = fma(0.5, Mb[r][j], v);
                        if (r == z(j)) v = fma(0.5, Mb[c][j], v);
                    }
                    if (r == z(0) && c == z(0)) v += Fb00;
                    if (r == z(0) && c == z(1)) v += Fb01;
                    if (r == z(1) && c == z(1)) v += Fb11;
                }
                L.G[sidx(r, c)] = v;
            }
    }
};

template <int MODEL, int D>
struct AdjModel;
template <int D>
struct AdjModel<M_CTCRW, D> { typedef AdjCtcrw<D> Lane; };
template <int D>
struct AdjModel<M_OU_SSM, D> { typedef AdjScal<D, true> Lane; };
template <int D>
struct AdjModel<M_BM_SSM, D> { typedef AdjScal<D, false> Lane; };

}  // namespace ssde
#endif
