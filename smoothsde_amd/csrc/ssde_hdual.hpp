// ssde_hdual.hpp -- second-order forward mode for the isotropic Kalman recursions: the exact second derivatives TMB gets from
// CppAD (tmb_obj_joint$he, R/sde.R:1363; the H_uu block of the Laplace approximation, R/sde.R:510-525, 656-658) for models whose
// SDE parameters vary from row to row (tau ~ s(temp), nu ~ s(temp): smoothSDE.rmd:476-497).
//
// A hyper-dual number carries a value, its derivatives in TWO directions a and b, and the mixed second derivative:
//     x = v + a eps_a + b eps_b + ab eps_a eps_b,      eps_a^2 = eps_b^2 = 0.
// Running the PRIMAL recursion (nllk_ctcrw.hpp:195-247, nllk_ou_ssm.hpp:163-213, nllk_bm_ssm.hpp:127-175) in this arithmetic,
// with the row's linear predictors seeded by the design-matrix entries of coefficient a and coefficient b, leaves
// d^2 nllk / d coef_a d coef_b in the `ab` part of the accumulated likelihood -- nothing is derived by hand and nothing is
// differenced.  One wavefront lane per coefficient PAIR (k_tv_hess.hip).
//
// __host__ __device__: tests/hostsim checks the arithmetic on the CPU against differences of the first-order lanes.
#ifndef SSDE_HDUAL_HPP
#define SSDE_HDUAL_HPP

#include "ssde_math.hpp"

namespace ssde {

struct HD {
    double v, a, b, ab;
    SSDE_HD HD() : v(0.0), a(0.0), b(0.0), ab(0.0) {}
    SSDE_HD HD(double v_) : v(v_), a(0.0), b(0.0), ab(0.0) {}
    SSDE_HD HD(double v_, double a_, double b_, double ab_) : v(v_), a(a_), b(b_), ab(ab_) {}
};

SSDE_HD HD operator+(const HD& x, const HD& y) { return HD(x.v + y.v, x.a + y.a, x.b + y.b, x.ab + y.ab); }
SSDE_HD HD operator-(const HD& x, const HD& y) { return HD(x.v - y.v, x.a - y.a, x.b - y.b, x.ab - y.ab); }
SSDE_HD HD operator-(const HD& x) { return HD(-x.v, -x.a, -x.b, -x.ab); }
SSDE_HD HD operator+(const HD& x, double c) { return HD(x.v + c, x.a, x.b, x.ab); }
SSDE_HD HD operator+(double c, const HD& x) { return HD(x.v + c, x.a, x.b, x.ab); }
SSDE_HD HD operator-(const HD& x, double c) { return HD(x.v - c, x.a, x.b, x.ab); }
SSDE_HD HD operator-(double c, const HD& x) { return HD(c - x.v, -x.a, -x.b, -x.ab); }
SSDE_HD HD operator*(const HD& x, double c) { return HD(x.v * c, x.a * c, x.b * c, x.ab * c); }
SSDE_HD HD operator*(double c, const HD& x) { return HD(x.v * c, x.a * c, x.b * c, x.ab * c); }
SSDE_HD HD operator*(const HD& x, const HD& y) {
    return HD(x.v * y.v, fma(x.a, y.v, x.v * y.a), fma(x.b, y.v, x.v * y.b),
              fma(x.ab, y.v, fma(x.a, y.b, fma(x.b, y.a, x.v * y.ab))));
}
// f(x) with f', f'' known at x.v
SSDE_HD HD hd_chain(const HD& x, double f, double f1, double f2) { return HD(f, f1 * x.a, f1 * x.b, fma(f2 * x.a, x.b, f1 * x.ab)); }
SSDE_HD HD hd_rcp(const HD& x) {
    const double r = rcp(x.v);
    return hd_chain(x, r, -r * r, 2.0 * r * r * r);
}
SSDE_HD HD operator/(const HD& x, const HD& y) { return x * hd_rcp(y); }
SSDE_HD HD hd_exp(const HD& x) {
    const double e = exp(x.v);
    return hd_chain(x, e, e, e);
}
SSDE_HD HD hd_sqrt(const HD& x) {
    const double s = sqrt(x.v), i = 0.5 / s;
    return hd_chain(x, s, i, -0.5 * i / x.v);
}
// the only part of log(x) the Hessian needs is its derivative structure; the value rides along for the tests
SSDE_HD HD hd_log(const HD& x) {
    const double r = rcp(x.v);
    return hd_chain(x, log(x.v), r, -r * r);
}

// f(x, q) with its first and second partial derivatives known at (x.v, q.v)
SSDE_HD HD hd_chain2(const HD& x, const HD& q, double f, double fx, double fq, double fxx, double fxq, double fqq) {
    return HD(f, fx * x.a + fq * q.a, fx * x.b + fq * q.b,
              fx * x.ab + fq * q.ab + fxx * x.a * x.b + fxq * (x.a * q.b + x.b * q.a) + fqq * q.a * q.b);
}

// the names the general (full-covariance) step of ssde_dense.hpp is written in, so that dense_step_g runs in HD as it runs in DualN
SSDE_HD HD operator/(double c, const HD& x) { return hd_rcp(x) * c; }
SSDE_HD HD operator/(const HD& x, double c) { return x * (1.0 / c); }
SSDE_HD HD dexp(const HD& x) { return hd_exp(x); }
SSDE_HD HD dlog(const HD& x) { return hd_log(x); }
SSDE_HD HD dsqrt(const HD& x) { return hd_sqrt(x); }
SSDE_HD HD dfabs(const HD& x) { return x.v < 0.0 ? -x : x; }

// ---- scalar-type generic helpers: the same template text runs in double (host checks) and in HD ------------------------
SSDE_HD double g_exp(double x) { return exp(x); }
SSDE_HD HD g_exp(const HD& x) { return hd_exp(x); }
SSDE_HD double g_sqrt(double x) { return sqrt(x); }
SSDE_HD HD g_sqrt(const HD& x) { return hd_sqrt(x); }
SSDE_HD double g_rcp(double x) { return rcp(x); }
SSDE_HD HD g_rcp(const HD& x) { return hd_rcp(x); }
SSDE_HD double g_log(double x) { return log(x); }
SSDE_HD HD g_log(const HD& x) { return hd_log(x); }
SSDE_HD double g_val(double x) { return x; }
SSDE_HD double g_val(const HD& x) { return x.v; }

// ---- transitions as functions of the working-scale predictors (A3, A4) ----------------------------------------------------
// CTCRW: tau = exp(p1), nu = exp(p2), beta = 1 / tau, sigma = 2 nu / sqrt(pi tau)   (nllk_ctcrw.hpp:152-156); makeT / makeQ / makeB
// entries :45-91 in the cancellation-aware forms of ctcrw_trans (ssde_math.hpp)
template <class T>
struct CtcrwTr { T e, t12, b1, b2, q11, q12, q22; };
template <class T>
SSDE_HD void ctcrw_trans_g(double dt, const T& p1, const T& p2, CtcrwTr<T>& o) {
    const T tau = g_exp(p1), nu = g_exp(p2);
    const T beta = g_rcp(tau);
    const T e = g_exp(-(beta * dt));
    const T e2 = e * e;
    const T ome = 1.0 - e;
    const T A = (4.0 / M_PI) * (nu * nu);                   // sigma^2 / beta = 4 nu^2 / pi
    o.e = e;
    o.t12 = ome * tau;
    o.b1 = dt - o.t12;
    o.b2 = ome;
    const T G = (dt - 2.0 * o.t12) + 0.5 * (tau * (1.0 - e2));
    o.q11 = A * (tau * G);
    o.q12 = 0.5 * (A * (tau * (ome * ome)));
    o.q22 = 0.5 * (A * (1.0 - e2));
}
// OU_SSM: T = e^{-dt/tau}, B = 1 - T, Q = kappa (1 - T^2) (nllk_ou_ssm.hpp:35, 50, 66); BM_SSM: T = 1, drift mu dt, Q = sigma^2 dt
// (nllk_bm_ssm.hpp:33, 99-100, 138-139)
template <class T>
struct ScalTr { T t, b, q; };
template <class T>
SSDE_HD void ou_trans_g(double dt, const T& p1, const T& p2, ScalTr<T>& o) {
    const T tau = g_exp(p1), kappa = g_exp(p2);
    o.t = g_exp(-(g_rcp(tau) * dt));
    o.b = 1.0 - o.t;
    o.q = kappa * (1.0 - o.t * o.t);
}
template <class T>
SSDE_HD void bm_trans_g(double dt, const T& p1, ScalTr<T>& o) {
    const T s = g_exp(p1);
    o.t = T(1.0);
    o.b = T(dt);
    o.q = (s * s) * dt;
}

// ---- one row of the isotropic filters in generic arithmetic: score y, then propagate over the interval after the row -------
// (the primal halves of tv_ctcrw_step / tv_scal_step, ssde_tv.hpp).  nll accumulates 1/2 (D log F + u'u / F).
template <class T, int D>
struct IsoCtcrwState {
    T x[D], v[D], p11, p12, p22, nll;
};
template <class T, int D>
SSDE_HD void iso_ctcrw_row(IsoCtcrwState<T, D>& S, const CtcrwTr<T>& tr, const T& h, const T* mu, const double* y, int any_nan) {
    const bool na = is_na(y[0], any_nan);                          // obs(i,0) only: nllk_ctcrw.hpp:214
    const T F = S.p11 + h;                                         // :223
    const double Fv = g_val(F);
    const double detF = (D == 1) ? Fv : Fv * Fv;
    const bool upd = !na && !(detF <= 0.0);                        // :214, 226 (NaN: update branch)
    const double bm = (na || upd) ? 1.0 : 0.0;                     // Q3: detF <= 0 predicts without B mu
    const T tp11 = S.p11 + tr.t12 * S.p12, tp12 = S.p12 + tr.t12 * S.p22;
    const T tp21 = tr.e * S.p12, tp22 = tr.e * S.p22;
    if (upd) {
        const T iF = g_rcp(F);
        const T k1 = tp11 * iF, k2 = tp21 * iF;                    // :236
        T su2 = T(0.0);
        T u[D];
        for (int a = 0; a < D; a++) { u[a] = y[a] - S.x[a]; su2 = su2 + u[a] * u[a]; }      // :221
        S.nll = S.nll + 0.5 * ((double)D * g_log(F) + iF * su2);                            // :231-234
        for (int a = 0; a < D; a++) {                                                       // :238
            const T nx = S.x[a] + tr.t12 * S.v[a] + k1 * u[a] + tr.b1 * mu[a];
            const T nv = tr.e * S.v[a] + k2 * u[a] + tr.b2 * mu[a];
            S.x[a] = nx; S.v[a] = nv;
        }
        const T n11 = tp11 * (1.0 - k1) + tp12 * tr.t12 + tr.q11;                           // :240-241
        const T n12 = tp12 * tr.e - tp11 * k2 + tr.q12;
        const T n22 = tp22 * tr.e - tp21 * k2 + tr.q22;
        S.p11 = n11; S.p12 = n12; S.p22 = n22;
    } else {
        for (int a = 0; a < D; a++) {                                                       // :214-217, 226-228
            const T nx = S.x[a] + tr.t12 * S.v[a] + bm * (tr.b1 * mu[a]);
            const T nv = tr.e * S.v[a] + bm * (tr.b2 * mu[a]);
            S.x[a] = nx; S.v[a] = nv;
        }
        const T n11 = tp11 + tp12 * tr.t12 + tr.q11;
        const T n12 = tp12 * tr.e + tr.q12;
        const T n22 = tp22 * tr.e + tr.q22;
        S.p11 = n11; S.p12 = n12; S.p22 = n22;
    }
}

template <class T, int D>
struct IsoScalState {
    T x[D], p, nll;
};
template <class T, int D>
SSDE_HD void iso_scal_row(IsoScalState<T, D>& S, const ScalTr<T>& tr, const T& h, const T* mu, const double* y, int any_nan) {
    const bool na = is_na(y[0], any_nan);
    const T F = S.p + h;
    const bool upd = !na && !(fabs(g_val(F)) <= 0.0);              // detF = exp(logdet F): nllk_ou_ssm.hpp:190-195
    if (upd) {
        const T iF = g_rcp(F);
        const T k = tr.t * (S.p * iF);
        T su2 = T(0.0);
        T u[D];
        for (int a = 0; a < D; a++) { u[a] = y[a] - S.x[a]; su2 = su2 + u[a] * u[a]; }
        S.nll = S.nll + 0.5 * ((double)D * g_log(F) + iF * su2);
        for (int a = 0; a < D; a++) S.x[a] = tr.t * S.x[a] + k * u[a] + tr.b * mu[a];       // nllk_ou_ssm.hpp:204, nllk_bm_ssm.hpp:166
        S.p = (tr.t * tr.t) * (S.p * (h * iF)) + tr.q;             // T P (T - K Z)' + Q with 1 - p / F = h / F
    } else {
        for (int a = 0; a < D; a++) S.x[a] = tr.t * S.x[a] + tr.b * mu[a];
        S.p = (tr.t * tr.t) * S.p + tr.q;
    }
}

}  // namespace ssde
#endif
