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
