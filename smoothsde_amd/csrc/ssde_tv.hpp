// ssde_tv.hpp -- isotropic Kalman step with ROW-VARYING SDE parameters (spline / covariate
// dependent tau, nu, mu, ... : the models smoothSDE exists for), one tangent direction per lane.
//
// Reference: the same recursions as ssde_math.hpp (nllk_ctcrw.hpp:195-247, nllk_ou_ssm.hpp:163-213,
// nllk_bm_ssm.hpp:127-175), with par_mat.row(i) differing from row to row (the linear predictor
// par_vec = X_fe coeff_fe + X_re coeff_re, nllk_ctcrw.hpp:143-156).  H = sigma_obs^2 I and a
// block-identical P0 are still required (else the dense kernel runs), so the covariance keeps its
// 3-scalar (CTCRW) / 1-scalar (OU, BM) form.
//
// Work split (k_tv.hip):
//   * a row-parallel pre-pass evaluates everything that does not depend on the filter state --
//     linear predictor (A2), link transforms (A3), transition entries and their derivatives (A4) --
//     into one 128-byte RECORD per row (tv_make_record below).  All exp() calls live there;
//   * the serial recursion then runs one wavefront per (track, time window), lane = gradient
//     direction: every lane repeats the cheap primal step and carries ONE tangent, for any kind of
//     direction (log sigma_obs, a coefficient of mu_a, of par[D], of par[D+1]) -- the kinds differ
//     only in their seeds, which are lane-wise selects, so the wave never diverges.
// The functions are __host__ __device__: tests/hostsim runs them on the CPU against the oracle.
#ifndef SSDE_TV_HPP
#define SSDE_TV_HPP

#include "ssde_dense.hpp"
#include "ssde_math.hpp"

namespace ssde {

constexpr int TV_RS = 16;   // doubles per row record (128 B: four 16-byte loads per lane)
enum { TVK_NONE = 0, TVK_SIG = 1, TVK_MU = 2, TVK_P1 = 3, TVK_P2 = 4, TVK_A1 = 5, TVK_A2 = 6 };

// Record layout
//   CTCRW : 0 e  1 t12  2 b1  3 q11  4 q12  5 q22 | d/d par[D]: 6 de  7 dt12  8 dq11  9 dq12  10 dq22
//   OU/BM : 0 t  1 b    2 q                        | d/d par[D]: 3 dt_ 4 db    5 dq ; d/d par[D+1]: 6 dq2
//   both  : 11 + a = mu_a (row's linear predictor), 13 + a = y_a (the row's observation)
// (CTCRW d/d par[D+1] = log nu: dq = 2 q, nothing else moves.)
constexpr int TVR_MU = 11, TVR_Y = 13;

// par[] = the row's linear predictors on the working scale (length D + 1 or D + 2)
template <int MODEL, int D>
SSDE_HD void tv_make_record(double dt, const double* par, const double* y, double* r) {
    for (int k = 0; k < TV_RS; k++) r[k] = 0.0;
    if (MODEL == M_CTCRW) {
        const double tau = exp(par[D]), nu = exp(par[D + 1]);      // nllk_ctcrw.hpp:153-154
        const double beta = 1.0 / tau;                             // :155
        const double sigma = 2.0 * nu / sqrt(M_PI * tau);          // :156
        CtcrwTrans tr;
        ctcrw_trans(dt, tau, beta, sigma, tr);
        r[0] = tr.e; r[1] = tr.t12; r[2] = tr.b1; r[3] = tr.q11; r[4] = tr.q12; r[5] = tr.q22;
        r[6] = tr.de; r[7] = tr.dt12; r[8] = tr.dq11; r[9] = tr.dq12; r[10] = tr.dq22;
    } else if (MODEL == M_OU_SSM) {
        ScalTrans tr;
        ou_trans(dt, exp(par[D]), exp(par[D + 1]), tr);            // nllk_ou_ssm.hpp:123-124
        r[0] = tr.t; r[1] = tr.b; r[2] = tr.q; r[3] = tr.dt_; r[4] = tr.db; r[5] = tr.dq; r[6] = tr.q;
    } else {
        ScalTrans tr;
        bm_trans(dt, exp(par[D]), tr);                             // nllk_bm_ssm.hpp:90
        r[0] = tr.t; r[1] = tr.b; r[2] = tr.q; r[3] = tr.dt_; r[4] = tr.db; r[5] = tr.dq; r[6] = 0.0;
    }
    for (int a = 0; a < D; a++) { r[TVR_MU + a] = par[a]; r[TVR_Y + a] = y[a]; }
}

// ---------------------------------------------------------------------------------------
// CTCRW lane: primal (x, v per dimension; p11, p12, p22) + one tangent
// ---------------------------------------------------------------------------------------
template <int D>
struct TvCtcrwLane {
    static constexpr int SD = 2 * D;
    static constexpr int NSTATE = 2 * D + 3 + 3 + 2 * D;
    double x[D], v[D], p11, p12, p22;
    double tx[D], tv[D], d11, d12, d22;
    LogAcc ld;
    double accq, gld, gq;
    SSDE_HD void init(const double* a0 /* x1, v1, x2, v2 */, const double* p0 /* p11, p12, p22 */) {
        for (int a = 0; a < D; a++) { x[a] = a0[2 * a]; v[a] = a0[2 * a + 1]; tx[a] = tv[a] = 0.0; }
        p11 = p0[0]; p12 = p0[1]; p22 = p0[2];
        d11 = d12 = d22 = 0.0;
        reset_acc();
    }
    // a time window warms up from the row's observation, velocities 0 (any state would do)
    SSDE_HD void warm_init(const double* y, const double* p0) {
        double a0[SD];
        for (int a = 0; a < D; a++) { a0[2 * a] = (y[a] == y[a]) ? y[a] : 0.0; a0[2 * a + 1] = 0.0; }
        init(a0, p0);
    }
    SSDE_HD void reset_acc() { ld.init(); accq = gld = gq = 0.0; }
    SSDE_HD void dump(double* o) const {
        int k = 0;
        for (int a = 0; a < D; a++) { o[k++] = x[a]; o[k++] = v[a]; }
        o[k++] = p11; o[k++] = p12; o[k++] = p22;
        o[k++] = d11; o[k++] = d12; o[k++] = d22;
        for (int a = 0; a < D; a++) { o[k++] = tx[a]; o[k++] = tv[a]; }
    }
    SSDE_HD void state(double* o) const { for (int a = 0; a < D; a++) { o[2 * a] = x[a]; o[2 * a + 1] = v[a]; } }
    SSDE_HD double value() const { return 0.5 * ((double)D * ld.value() + accq); }
    SSDE_HD double grad() const { return 0.5 * (double)D * gld + gq; }
};

// One row: score y (record slots 13..), then propagate over the interval after the row.
//   kind / dim / w : the lane's direction (TVK_*), its dimension for TVK_MU, and the row's weight
//                    d par_row / d coefficient (design-matrix entry; 1 for an intercept or sigma_obs)
template <int D, bool GRAD>
SSDE_HD void tv_ctcrw_step(TvCtcrwLane<D>& L, const double* r, double h, int kind, int dim, double w, int any_nan) {
    const double e = r[0], t12 = r[1], b1 = r[2], q11 = r[3], q12 = r[4], q22 = r[5];
    const bool na = is_na(r[TVR_Y], any_nan);                   // obs(i,0) only: nllk_ctcrw.hpp:214
    const double F = L.p11 + h;                                 // line 223
    const double detF = (D == 1) ? F : F * F;
    const bool upd = !na && !(detF <= 0.0);                     // lines 214, 226 (NaN: update branch)
    const double iF = upd ? rcp(F) : 0.0;
    const double bm = (na || upd) ? 1.0 : 0.0;                  // Q3
    L.ld.mul(upd ? F : 1.0);
    const double tp11 = L.p11 + t12 * L.p12, tp12 = L.p12 + t12 * L.p22;
    const double tp21 = e * L.p12, tp22 = e * L.p22;
    const double k1 = tp11 * iF, k2 = tp21 * iF;                // line 236
    double u[D], su2 = 0.0;
    for (int a = 0; a < D; a++) { u[a] = upd ? r[TVR_Y + a] - L.x[a] : 0.0; su2 += u[a] * u[a]; }   // line 221
    L.accq += iF * su2;
    if (GRAD) {
        const double s1 = (kind == TVK_P1) ? w : 0.0, s2 = (kind == TVK_P2) ? 2.0 * w : 0.0;
        const double dh = (kind == TVK_SIG) ? 2.0 * h : 0.0;
        const double de = s1 * r[6], dt12 = s1 * r[7];
        const double dq11 = s1 * r[8] + s2 * q11, dq12 = s1 * r[9] + s2 * q12, dq22 = s1 * r[10] + s2 * q22;
        const double dF = L.d11 + dh;
        const double diF = -iF * iF * dF;
        L.gld += dF * iF;
        const double dtp11 = L.d11 + t12 * L.d12 + dt12 * L.p12;
        const double dtp12 = L.d12 + t12 * L.d22 + dt12 * L.p22;
        const double dtp21 = e * L.d12 + de * L.p12;
        const double dtp22 = e * L.d22 + de * L.p22;
        const double dk1 = dtp11 * iF + tp11 * diF;
        const double dk2 = dtp21 * iF + tp21 * diF;
        L.d11 = dtp11 * (1.0 - k1) - tp11 * dk1 + dtp12 * t12 + tp12 * dt12 + dq11;
        L.d12 = -dtp11 * k2 - tp11 * dk2 + dtp12 * e + tp12 * de + dq12;
        L.d22 = -dtp21 * k2 - tp21 * dk2 + dtp22 * e + tp22 * de + dq22;
        double sud = 0.0;
        for (int a = 0; a < D; a++) {
            const double du = upd ? -L.tx[a] : 0.0;
            sud += u[a] * du;
            const double dmu = (kind == TVK_MU && dim == a) ? w : 0.0;
            const double vm = L.v[a] - bm * r[TVR_MU + a];      // d(B mu) = -(dt12, de) mu + B dmu
            const double nx = L.tx[a] + t12 * L.tv[a] + dk1 * u[a] + k1 * du + dt12 * vm + bm * b1 * dmu;
            const double nv = e * L.tv[a] + dk2 * u[a] + k2 * du + de * vm + bm * (1.0 - e) * dmu;
            L.tx[a] = nx; L.tv[a] = nv;
        }
        L.gq += 0.5 * diF * su2 + iF * sud;
    }
    for (int a = 0; a < D; a++) {                               // a = T a + K u + B mu (line 238)
        const double mu = r[TVR_MU + a];
        const double nx = L.x[a] + t12 * L.v[a] + k1 * u[a] + bm * b1 * mu;
        const double nv = e * L.v[a] + k2 * u[a] + bm * (1.0 - e) * mu;
        L.x[a] = nx; L.v[a] = nv;
    }
    const double n11 = tp11 * (1.0 - k1) + tp12 * t12 + q11;    // lines 240-241
    const double n12 = -tp11 * k2 + tp12 * e + q12;
    const double n22 = -tp21 * k2 + tp22 * e + q22;
    L.p11 = n11; L.p12 = n12; L.p22 = n22;
}

// ---------------------------------------------------------------------------------------
// OU_SSM / BM_SSM lane: x per dimension, scalar p, one tangent
// ---------------------------------------------------------------------------------------
template <int D>
struct TvScalLane {
    static constexpr int SD = D;
    static constexpr int NSTATE = D + 1 + 1 + D;
    double x[D], p, tx[D], dp;
    LogAcc ld;
    double accq, gld, gq;
    SSDE_HD void init(const double* a0, const double* p0) {
        for (int a = 0; a < D; a++) { x[a] = a0[a]; tx[a] = 0.0; }
        p = p0[0]; dp = 0.0;
        reset_acc();
    }
    SSDE_HD void warm_init(const double* y, const double* p0) {
        double a0[SD];
        for (int a = 0; a < D; a++) a0[a] = (y[a] == y[a]) ? y[a] : 0.0;
        init(a0, p0);
    }
    SSDE_HD void reset_acc() { ld.init(); accq = gld = gq = 0.0; }
    SSDE_HD void dump(double* o) const {
        int k = 0;
        for (int a = 0; a < D; a++) o[k++] = x[a];
        o[k++] = p; o[k++] = dp;
        for (int a = 0; a < D; a++) o[k++] = tx[a];
    }
    SSDE_HD void state(double* o) const { for (int a = 0; a < D; a++) o[a] = x[a]; }
    SSDE_HD double value() const { return 0.5 * ((double)D * ld.value() + accq); }
    SSDE_HD double grad() const { return 0.5 * (double)D * gld + gq; }
};

template <int D, bool GRAD>
SSDE_HD void tv_scal_step(TvScalLane<D>& L, const double* r, double h, int kind, int dim, double w, int any_nan) {
    const double t = r[0], b = r[1], q = r[2];
    const bool na = is_na(r[TVR_Y], any_nan);
    const double F = L.p + h;
    const bool upd = !na && !(fabs(F) <= 0.0);                  // detF = exp(logdet F): nllk_ou_ssm.hpp:190-195
    const double iF = upd ? rcp(F) : 0.0;
    L.ld.mul(upd ? F : 1.0);
    // cancellation-free form of P = T P (T - K Z)' + Q and of its derivative: see scal_cov_step (ssde_math.hpp)
    const double ha = upd ? h * iF : 1.0, pb = L.p * iF;         // h/F, p/F
    const double k = t * pb;
    const double t2 = t * t;
    double u[D], su2 = 0.0;
    for (int a = 0; a < D; a++) { u[a] = upd ? r[TVR_Y + a] - L.x[a] : 0.0; su2 += u[a] * u[a]; }
    L.accq += iF * su2;
    if (GRAD) {
        const double s1 = (kind == TVK_P1) ? w : 0.0, s2 = (kind == TVK_P2) ? w : 0.0;
        const double dh = (kind == TVK_SIG) ? 2.0 * h : 0.0;
        const double dt_ = s1 * r[3], db = s1 * r[4], dq = s1 * r[5] + s2 * r[6];
        const double dF = L.dp + dh;
        const double diF = -iF * iF * dF;
        L.gld += dF * iF;
        const double dk = dt_ * pb + t * iF * (ha * L.dp - pb * dh);
        const double ndp = t2 * (ha * ha * L.dp + pb * pb * dh) + 2.0 * t * dt_ * L.p * ha + dq;
        double sud = 0.0;
        for (int a = 0; a < D; a++) {
            const double du = upd ? -L.tx[a] : 0.0;
            sud += u[a] * du;
            const double dmu = (kind == TVK_MU && dim == a) ? w : 0.0;
            L.tx[a] = t * L.tx[a] + dk * u[a] + k * du + dt_ * L.x[a] + db * r[TVR_MU + a] + b * dmu;
        }
        L.gq += 0.5 * diF * su2 + iF * sud;
        L.dp = ndp;
    }
    for (int a = 0; a < D; a++) L.x[a] = t * L.x[a] + k * u[a] + b * r[TVR_MU + a];
    L.p = t2 * L.p * ha + q;
}

// model -> lane type / step
template <int MODEL, int D>
struct TvOps {
    typedef TvScalLane<D> Lane;
    static constexpr int U = 4;            // rows per prefetch block
    static constexpr int Y_OFF = TVR_Y;    // where the record keeps the observation
    static constexpr bool DENSE = false;
    template <bool GRAD>
    SSDE_HD static void step(Lane& L, const double* r, double h, int kind, int dim, double w, int any_nan) {
        tv_scal_step<D, GRAD>(L, r, h, kind, dim, w, any_nan);
    }
};
template <int D>
struct TvOps<M_CTCRW, D> {
    typedef TvCtcrwLane<D> Lane;
    static constexpr int U = 4;
    static constexpr int Y_OFF = TVR_Y;
    static constexpr bool DENSE = false;
    template <bool GRAD>
    SSDE_HD static void step(Lane& L, const double* r, double h, int kind, int dim, double w, int any_nan) {
        tv_ctcrw_step<D, GRAD>(L, r, h, kind, dim, w, any_nan);
    }
};

// ---------------------------------------------------------------------------------------
// General ("dense") variant for the same lane = direction kernels: per-row H_array
// (nllk_ctcrw.hpp:203-205) and / or a P0 that is not block-identical (R/sde.R:552-557, 582-587).
// Full sdim x sdim covariance, the step of ssde_dense.hpp with ONE dual direction per lane.
// Record: 0..Q-1 the row's linear predictors | 4 dt | 5.. y | 7.. H_array[,,i] (column-major d x d)
// ---------------------------------------------------------------------------------------
constexpr int TVD_DT = 4, TVD_Y = 5, TVD_H = 7;

template <int MODEL, int D>
SSDE_HD void tv_make_record_dense(double dt, const double* par, const double* y, const double* hrow /* or NULL */, double* r) {
    constexpr int Q = (MODEL == M_BM_SSM) ? D + 1 : D + 2;
    for (int k = 0; k < TV_RS; k++) r[k] = 0.0;
    for (int j = 0; j < Q; j++) r[j] = par[j];
    r[TVD_DT] = dt;
    for (int a = 0; a < D; a++) r[TVD_Y + a] = y[a];
    if (hrow) for (int k = 0; k < D * D; k++) r[TVD_H + k] = hrow[k];
}

template <int MODEL, int D>
struct TvDenseLane {
    static constexpr int SD = DenseDims<MODEL, D>::SD;
    static constexpr int NSTATE = 2 * (SD + SD * SD);
    DenseLane<MODEL, D, 1> L;
    bool has_h;
    SSDE_HD void init(const double* a0, const double* p0 /* SD x SD column-major */) { L.init(a0, p0); }
    SSDE_HD void warm_init(const double* y, const double* p0) {
        double a0[SD];
        for (int c = 0; c < SD; c++) a0[c] = 0.0;
        for (int a = 0; a < D; a++) a0[DenseDims<MODEL, D>::z(a)] = (y[a] == y[a]) ? y[a] : 0.0;
        L.init(a0, p0);
    }
    SSDE_HD void reset_acc() { L.nll = DualN<1>(0.0); }
    SSDE_HD void dump(double* o) const {
        int k = 0;
        for (int i = 0; i < SD; i++) { o[k++] = L.a[i].v; o[k++] = L.a[i].d[0]; }
        for (int i = 0; i < SD; i++)
            for (int j = 0; j < SD; j++) { o[k++] = L.P[i][j].v; o[k++] = L.P[i][j].d[0]; }
    }
    SSDE_HD void state(double* o) const { for (int i = 0; i < SD; i++) o[i] = L.a[i].v; }
    SSDE_HD double value() const { return L.nll.v; }
    SSDE_HD double grad() const { return L.nll.d[0]; }
};

template <int MODEL, int D>
struct TvDenseOps {
    typedef TvDenseLane<MODEL, D> Lane;
    static constexpr int U = 2;            // the 4 x 4 dual covariance needs the registers
    static constexpr int Y_OFF = TVD_Y;
    static constexpr bool DENSE = true;
    template <bool GRAD>
    SSDE_HD static void step(Lane& S, const double* r, double h, int kind, int dim, double w, int any_nan) {
        constexpr int Q = DenseDims<MODEL, D>::Q;
        DualN<1> par[Q];
        for (int j = 0; j < Q; j++) {
            par[j].v = r[j];
            const bool mine = (j < D) ? (kind == TVK_MU && dim == j) : (j == D ? kind == TVK_P1 : kind == TVK_P2);
            par[j].d[0] = (GRAD && mine) ? w : 0.0;
        }
        DualN<1> H[D][D];
        for (int i = 0; i < D; i++)
            for (int j = 0; j < D; j++) {
                if (S.has_h) H[i][j] = DualN<1>(r[TVD_H + i + j * D]);           // H_array[,,i], no parameter in it
                else { H[i][j].v = (i == j) ? h : 0.0; H[i][j].d[0] = (GRAD && i == j && kind == TVK_SIG) ? 2.0 * h : 0.0; }
            }
        double y[D];
        for (int a = 0; a < D; a++) y[a] = r[TVD_Y + a];
        dense_step<MODEL, D, 1>(S.L, par, H, r[TVD_DT], y, is_na(y[0], any_nan));
    }
};

// ---------------------------------------------------------------------------------------
// ESEAL_SSM (nllk_e_seal_ssm.hpp:139-207).  The first state component is the constant 1 (a0 = (1, L0),
// P0 = diag(0, p0), R/sde.R:602-603; T and Q never touch it), so the 2 x 2 filter is a SCALAR filter on the lipid
// mass L with a row-varying observation loading:
//     y_i = a1 + z_i L + N(0, H_i),  z_i = a2 / R_i,  H_i = tau^2 / h_i;    L' = L + mu_i dt_i + N(0, sigma_i^2 dt_i)
// Record: 0 z  1 H  2 mu dt  3 sigma^2 dt  4 dt  5 y  6 a1.   Directions: TVK_SIG = log tau (dH = 2 H),
// TVK_A1 = a1, TVK_A2 = log a2 (dz = z), TVK_MU = a coefficient of mu (d drift = w dt), TVK_P1 = of log sigma (dq = 2 q w).
// ---------------------------------------------------------------------------------------
constexpr int TVE_Z = 0, TVE_H = 1, TVE_DRIFT = 2, TVE_Q = 3, TVE_DT = 4, TVE_Y = 5, TVE_A1 = 6;

struct TvEsealLane {
    static constexpr int SD = 2;
    static constexpr int NSTATE = 4;
    double x, p, tx, dp;
    LogAcc ld;
    double accq, gld, gq;
    bool has_h;                                    // (unused; the kernel sets it on full-covariance lanes)
    SSDE_HD void init(const double* a0 /* (1, L0) */, const double* p0 /* 2 x 2 column-major */) {
        x = a0[1]; p = p0[3]; tx = dp = 0.0;
        reset_acc();
    }
    SSDE_HD void warm_init(const double*, const double* p0) { x = 0.0; p = p0[3]; tx = dp = 0.0; reset_acc(); }
    SSDE_HD void reset_acc() { ld.init(); accq = gld = gq = 0.0; }
    SSDE_HD void dump(double* o) const { o[0] = x; o[1] = p; o[2] = tx; o[3] = dp; }
    SSDE_HD void state(double* o) const { o[0] = 1.0; o[1] = x; }
    SSDE_HD double value() const { return 0.5 * (ld.value() + accq); }
    SSDE_HD double grad() const { return 0.5 * gld + gq; }
};

struct TvEsealOps {
    typedef TvEsealLane Lane;
    static constexpr int U = 4;
    static constexpr int Y_OFF = TVE_Y;
    static constexpr bool DENSE = true;            // initial covariance from the full P0
    template <bool GRAD>
    SSDE_HD static void step(Lane& L, const double* r, double, int kind, int, double w, int any_nan) {
        const double z = r[TVE_Z], H = r[TVE_H], drift = r[TVE_DRIFT], q = r[TVE_Q];
        const bool na = is_na(r[TVE_Y], any_nan);                      // line 175
        const double F = z * z * L.p + H;                              // line 184
        const bool upd = !na && !(F <= 0.0);                           // line 188 (NaN: update branch)
        const double iF = upd ? rcp(F) : 0.0;
        L.ld.mul(upd ? F : 1.0);
        const double u = upd ? r[TVE_Y] - r[TVE_A1] - z * L.x : 0.0;   // line 182
        const double k = L.p * z * iF;                                 // K = T P Z' F^-1, second row (line 197)
        // 1 - k z = H / F and the derivatives of p H / F and p z / F without the cancellations of the literal forms
        // (a = H/F, b = p/F):  d[p H/F] = a^2 dp + z^2 b^2 dH - 2 z p a b dz,  dk = (z a dp + p (a - z^2 b) dz - z b dH) / F
        const double c = upd ? H * iF : 1.0, pb = L.p * iF;
        L.accq += iF * u * u;
        if (GRAD) {
            const double dz = (kind == TVK_A2) ? z : 0.0, dH = (kind == TVK_SIG) ? 2.0 * H : 0.0;
            const double da1 = (kind == TVK_A1) ? 1.0 : 0.0;
            const double ddrift = (kind == TVK_MU) ? w * r[TVE_DT] : 0.0, dq = (kind == TVK_P1) ? 2.0 * q * w : 0.0;
            const double dF = 2.0 * z * dz * L.p + z * z * L.dp + dH;
            const double du = upd ? -da1 - dz * L.x - z * L.tx : 0.0;
            const double diF = -iF * iF * dF;
            L.gld += dF * iF;
            L.gq += 0.5 * diF * u * u + iF * u * du;
            const double dk = iF * (z * c * L.dp + L.p * (c - z * z * pb) * dz - z * pb * dH);
            L.tx = L.tx + ddrift + dk * u + k * du;
            L.dp = L.dp * c * c + dH * z * z * pb * pb - 2.0 * z * dz * L.p * pb * c + dq;
        }
        L.x = L.x + drift + k * u;                                     // lines 176, 188, 199
        L.p = L.p * c + q;                                             // lines 177, 189, 201-202
    }
};

}  // namespace ssde
#endif
