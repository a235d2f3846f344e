// ssde_dense.hpp -- general (non-isotropic / time-varying) Kalman step for one lane.
//
// Covers everything the register-optimised isotropic kernels do not: per-row H_array
// (nllk_ctcrw.hpp:203-205), a user P0 that is not block-identical (R/sde.R:552-557, 582-587),
// and SDE parameters that vary from row to row through streamed design columns
// (nllk_ctcrw.hpp:143-156).  The covariance is a full sdim x sdim matrix here.
//
// Derivatives: the reference differentiates its template with CppAD; this path carries N
// tangent directions per lane with a small dual-number type written for the device
// (__host__ __device__, so tests/hostsim can run the same code on the CPU).  The transition
// matrices have known sparsity (makeT/Q/B_ctcrw, nllk_ctcrw.hpp:45-91), which the products
// below exploit instead of multiplying dense 4x4 matrices.
#ifndef SSDE_DENSE_HPP
#define SSDE_DENSE_HPP

#include "ssde_math.hpp"

namespace ssde {

// k_dense_wide.hip (five to eight response columns) compiles the step with its loops kept as loops: fully unrolled, the 16 x 16
// covariance of dual numbers needs ~800 spilled scalar registers and the generated code gave wrong tangents at sdim = 16
#ifdef SSDE_DENSE_NOUNROLL
#define SSDE_DLOOP _Pragma("nounroll")
#else
#define SSDE_DLOOP
#endif

template <int N>
struct DualN {
    double v;
    double d[N > 0 ? N : 1];
    SSDE_HD DualN() {}
    SSDE_HD DualN(double x) : v(x) { for (int k = 0; k < N; k++) d[k] = 0.0; }
};

template <int N> SSDE_HD DualN<N> operator+(const DualN<N>& a, const DualN<N>& b) {
    DualN<N> r; r.v = a.v + b.v; for (int k = 0; k < N; k++) r.d[k] = a.d[k] + b.d[k]; return r; }
template <int N> SSDE_HD DualN<N> operator-(const DualN<N>& a, const DualN<N>& b) {
    DualN<N> r; r.v = a.v - b.v; for (int k = 0; k < N; k++) r.d[k] = a.d[k] - b.d[k]; return r; }
template <int N> SSDE_HD DualN<N> operator*(const DualN<N>& a, const DualN<N>& b) {
    DualN<N> r; r.v = a.v * b.v; for (int k = 0; k < N; k++) r.d[k] = a.d[k] * b.v + a.v * b.d[k]; return r; }
template <int N> SSDE_HD DualN<N> operator/(const DualN<N>& a, const DualN<N>& b) {
    DualN<N> r; const double ib = 1.0 / b.v; r.v = a.v * ib;
    for (int k = 0; k < N; k++) r.d[k] = (a.d[k] - r.v * b.d[k]) * ib; return r; }
template <int N> SSDE_HD DualN<N> operator-(const DualN<N>& a) {
    DualN<N> r; r.v = -a.v; for (int k = 0; k < N; k++) r.d[k] = -a.d[k]; return r; }
template <int N> SSDE_HD DualN<N> operator+(const DualN<N>& a, double b) { DualN<N> r = a; r.v += b; return r; }
template <int N> SSDE_HD DualN<N> operator+(double a, const DualN<N>& b) { DualN<N> r = b; r.v += a; return r; }
template <int N> SSDE_HD DualN<N> operator-(const DualN<N>& a, double b) { DualN<N> r = a; r.v -= b; return r; }
template <int N> SSDE_HD DualN<N> operator-(double a, const DualN<N>& b) {
    DualN<N> r; r.v = a - b.v; for (int k = 0; k < N; k++) r.d[k] = -b.d[k]; return r; }
template <int N> SSDE_HD DualN<N> operator*(const DualN<N>& a, double b) {
    DualN<N> r; r.v = a.v * b; for (int k = 0; k < N; k++) r.d[k] = a.d[k] * b; return r; }
template <int N> SSDE_HD DualN<N> operator*(double a, const DualN<N>& b) { return b * a; }
template <int N> SSDE_HD DualN<N> operator/(double a, const DualN<N>& b) { return DualN<N>(a) / b; }
template <int N> SSDE_HD DualN<N> dexp(const DualN<N>& a) {
    DualN<N> r; r.v = exp(a.v); for (int k = 0; k < N; k++) r.d[k] = r.v * a.d[k]; return r; }
template <int N> SSDE_HD DualN<N> dlog(const DualN<N>& a) {
    DualN<N> r; r.v = log(a.v); const double ia = 1.0 / a.v; for (int k = 0; k < N; k++) r.d[k] = a.d[k] * ia; return r; }
template <int N> SSDE_HD DualN<N> dsqrt(const DualN<N>& a) {
    DualN<N> r; r.v = sqrt(a.v); const double h = 0.5 / r.v; for (int k = 0; k < N; k++) r.d[k] = a.d[k] * h; return r; }
template <int N> SSDE_HD DualN<N> dfabs(const DualN<N>& a) { return a.v < 0.0 ? -a : a; }

// ---- d x d helpers for responses wider than two columns (n_dim = 3 ... 8 with a measurement covariance or a P0 that couples the
// columns): what Eigen's F.inverse() and TMB's atomic::logdet do in the reference (nllk_ctcrw.hpp:12-24, 231, 236) -- Gaussian
// elimination with partial pivoting on the VALUES, carried out in dual arithmetic so the tangents follow
template <int D, class T_>
SSDE_HD void dense_lu(const T_ (&F)[D][D], T_ (&LU)[D][D], int (&piv)[D], int& sign) {
    SSDE_DLOOP for (int i = 0; i < D; i++) for (int j = 0; j < D; j++) LU[i][j] = F[i][j];
    sign = 1;
    SSDE_DLOOP for (int k = 0; k < D; k++) {
        int p = k;
        double best = fabs(LU[k][k].v);
        SSDE_DLOOP for (int i = k + 1; i < D; i++) if (fabs(LU[i][k].v) > best) { best = fabs(LU[i][k].v); p = i; }
        piv[k] = p;
        if (p != k) {
            sign = -sign;
            SSDE_DLOOP for (int j = 0; j < D; j++) {
                // (static indices only: a select per candidate row keeps the matrix in registers on the device)
                T_ a = LU[k][j], b = LU[k][j];
                SSDE_DLOOP for (int i = k + 1; i < D; i++) if (i == p) b = LU[i][j];
                LU[k][j] = b;
                SSDE_DLOOP for (int i = k + 1; i < D; i++) if (i == p) LU[i][j] = a;
            }
        }
        const T_ ip = 1.0 / LU[k][k];
        SSDE_DLOOP for (int i = k + 1; i < D; i++) {
            LU[i][k] = LU[i][k] * ip;
            SSDE_DLOOP for (int j = k + 1; j < D; j++) LU[i][j] = LU[i][j] - LU[i][k] * LU[k][j];
        }
    }
}
template <int D, class T_>
SSDE_HD T_ dense_det_lu(const T_ (&F)[D][D]) {
    T_ LU[D][D];
    int piv[D], sign;
    dense_lu<D, T_>(F, LU, piv, sign);
    T_ det((double)sign);
    SSDE_DLOOP for (int k = 0; k < D; k++) det = det * LU[k][k];
    return det;
}
template <int D, class T_>
SSDE_HD void dense_inverse_lu(const T_ (&F)[D][D], T_ (&Fi)[D][D]) {
    T_ LU[D][D];
    int piv[D], sign;
    dense_lu<D, T_>(F, LU, piv, sign);
    SSDE_DLOOP for (int c = 0; c < D; c++) {
        T_ x[D];
        SSDE_DLOOP for (int i = 0; i < D; i++) x[i] = T_(i == c ? 1.0 : 0.0);
        SSDE_DLOOP for (int k = 0; k < D; k++) {                                // the row interchanges, in order
            T_ a = x[k], b = x[k];
            SSDE_DLOOP for (int i = k + 1; i < D; i++) if (i == piv[k]) b = x[i];
            x[k] = b;
            SSDE_DLOOP for (int i = k + 1; i < D; i++) if (i == piv[k]) x[i] = a;
        }
        SSDE_DLOOP for (int i = 1; i < D; i++) for (int j = 0; j < i; j++) x[i] = x[i] - LU[i][j] * x[j];            // L y = P e_c
        SSDE_DLOOP for (int i = D - 1; i >= 0; i--) {
            SSDE_DLOOP for (int j = i + 1; j < D; j++) x[i] = x[i] - LU[i][j] * x[j];
            x[i] = x[i] / LU[i][i];
        }
        SSDE_DLOOP for (int i = 0; i < D; i++) Fi[i][c] = x[i];
    }
}

template <int MODEL, int D>
struct DenseDims {
    static constexpr int SD = (MODEL == M_CTCRW) ? 2 * D : D;
    static constexpr int Q = (MODEL == M_BM_SSM) ? D + 1 : D + 2;
    SSDE_HD static constexpr int z(int i) { return (MODEL == M_CTCRW) ? 2 * i : i; }  // Z picks positions
};

template <int MODEL, int D, int N>
struct DenseLane {
    static constexpr int SD = DenseDims<MODEL, D>::SD;
    DualN<N> a[SD];
    DualN<N> P[SD][SD];
    DualN<N> nll;
    SSDE_HD void init(const double* a0, const double* p0 /* SD x SD column-major */) {
        SSDE_DLOOP for (int i = 0; i < SD; i++) {
            a[i] = DualN<N>(a0[i]);
            SSDE_DLOOP for (int j = 0; j < SD; j++) P[i][j] = DualN<N>(p0[i + j * SD]);
        }
        nll = DualN<N>(0.0);
    }
};

// One row: par[] = the row's linear predictors (working scale) as duals, H = observation
// covariance (d x d, row-major here), dt = interval after the row, y = observation, na = obs(i,0) NA.
// Generic in the number type T_ (DualN<N>: first-order tangents; HD, ssde_hdual.hpp: the hyper-dual numbers of the exact Hessian):
// T_ needs + - * / with itself and with double, dexp / dlog / dsqrt / dfabs, a constructor from double and a member v (the value).
// L: the lane's state -- members a[SD], P[SD][SD], nll of type T_.
template <int MODEL, int D, class T_, class ST>
SSDE_HD void dense_step_g(ST& L, const T_* par, const T_ (&H)[D][D], double dt, const double* y, bool na) {
    typedef DenseDims<MODEL, D> DM;
    constexpr int SD = DM::SD;

    // ---- transition pieces ---------------------------------------------------------------
    T_ t12(0.0), e(1.0), q11(0.0), q12(0.0), q22(0.0);  // CTCRW: 2x2 block; OU/BM: e and q11 only
    T_ drift[SD];
    if (MODEL == M_CTCRW) {
        const T_ tau = dexp(par[D]), nu = dexp(par[D + 1]);       // nllk_ctcrw.hpp:153-154
        const T_ beta = 1.0 / tau;                                 // :155
        const T_ sigma = 2.0 * nu / dsqrt(M_PI * tau);             // :156
        e = dexp(-(beta * dt));
        const T_ e2 = dexp(-2.0 * (beta * dt));
        t12 = (1.0 - e) / beta;                                    // makeT :51
        const T_ sb = sigma / beta;
        q11 = sb * sb * (dt - 2.0 / beta * (1.0 - e) + 1.0 / (2.0 * beta) * (1.0 - e2));  // makeQ :68-69
        q12 = sigma * sigma / (2.0 * beta * beta) * (1.0 - 2.0 * e + e2);                  // :70
        q22 = sigma * sigma / (2.0 * beta) * (1.0 - e2);                                   // :72
        SSDE_DLOOP for (int a = 0; a < D; a++) {
            drift[2 * a] = (dt - t12) * par[a];                    // makeB :87
            drift[2 * a + 1] = (1.0 - e) * par[a];                 // makeB :88
        }
    } else if (MODEL == M_OU_SSM) {
        const T_ tau = dexp(par[D]), kappa = dexp(par[D + 1]);     // nllk_ou_ssm.hpp:123-124
        e = dexp(-dt / tau);                                       // makeT :35   (-dt/tau)
        q11 = kappa * (1.0 - dexp(-2.0 * dt / tau));               // makeQ :66
        SSDE_DLOOP for (int a = 0; a < D; a++) drift[a] = (1.0 - e) * par[a]; // makeB :50
    } else {
        const T_ sigma = dexp(par[D]);                             // nllk_bm_ssm.hpp:90
        q11 = sigma * sigma * dt;                                  // makeQ :33
        SSDE_DLOOP for (int a = 0; a < D; a++) drift[a] = par[a] * dt;        // :139
    }

    // TP = T P using the sparsity of T
    T_ TP[SD][SD];
    SSDE_DLOOP for (int c = 0; c < SD; c++) {
        if (MODEL == M_CTCRW) {
            SSDE_DLOOP for (int a = 0; a < D; a++) {
                TP[2 * a][c] = L.P[2 * a][c] + t12 * L.P[2 * a + 1][c];
                TP[2 * a + 1][c] = e * L.P[2 * a + 1][c];
            }
        } else if (MODEL == M_OU_SSM) {
            SSDE_DLOOP for (int r = 0; r < SD; r++) TP[r][c] = e * L.P[r][c];
        } else {
            SSDE_DLOOP for (int r = 0; r < SD; r++) TP[r][c] = L.P[r][c];
        }
    }
    // Ta = T a
    T_ Ta[SD];
    if (MODEL == M_CTCRW) {
        SSDE_DLOOP for (int a = 0; a < D; a++) { Ta[2 * a] = L.a[2 * a] + t12 * L.a[2 * a + 1]; Ta[2 * a + 1] = e * L.a[2 * a + 1]; }
    } else if (MODEL == M_OU_SSM) {
        SSDE_DLOOP for (int r = 0; r < SD; r++) Ta[r] = e * L.a[r];
    } else {
        SSDE_DLOOP for (int r = 0; r < SD; r++) Ta[r] = L.a[r];
    }
    // TPT = TP T' + Q
    T_ TPT[SD][SD];
    SSDE_DLOOP for (int r = 0; r < SD; r++) {
        if (MODEL == M_CTCRW) {
            SSDE_DLOOP for (int a = 0; a < D; a++) {
                TPT[r][2 * a] = TP[r][2 * a] + t12 * TP[r][2 * a + 1];
                TPT[r][2 * a + 1] = e * TP[r][2 * a + 1];
            }
        } else if (MODEL == M_OU_SSM) {
            SSDE_DLOOP for (int c = 0; c < SD; c++) TPT[r][c] = e * TP[r][c];
        } else {
            SSDE_DLOOP for (int c = 0; c < SD; c++) TPT[r][c] = TP[r][c];
        }
    }
    if (MODEL == M_CTCRW) {
        SSDE_DLOOP for (int a = 0; a < D; a++) {
            TPT[2 * a][2 * a] = TPT[2 * a][2 * a] + q11;
            TPT[2 * a][2 * a + 1] = TPT[2 * a][2 * a + 1] + q12;
            TPT[2 * a + 1][2 * a] = TPT[2 * a + 1][2 * a] + q12;
            TPT[2 * a + 1][2 * a + 1] = TPT[2 * a + 1][2 * a + 1] + q22;
        }
    } else {
        SSDE_DLOOP for (int r = 0; r < SD; r++) TPT[r][r] = TPT[r][r] + q11;
    }

    // ---- measurement --------------------------------------------------------------------
    bool upd = !na;
    T_ F[D][D], det(1.0);
    if (upd) {
        SSDE_DLOOP for (int i = 0; i < D; i++)
            SSDE_DLOOP for (int j = 0; j < D; j++) F[i][j] = L.P[DM::z(i)][DM::z(j)] + H[i][j];   // F = Z P Z' + H
        if (D == 1) det = F[0][0];
        else if (D == 2) det = F[0][0] * F[1][1] - F[1][0] * F[0][1];                    // det(): nllk_ctcrw.hpp:16-19
        else det = dfabs(dense_det_lu<D, T_>(F));                                         // det = exp(atomic::logdet(F)) = |det F|: nllk_ctcrw.hpp:20-22
        // CTCRW tests det <= 0; OU/BM take exp(logdet) = |det|, which fails the test only at 0
        upd = (MODEL == M_CTCRW) ? !(det.v <= 0.0) : !(fabs(det.v) <= 0.0);   // (a NaN takes the update branch)
    }
    if (!upd) {
        // missing observation, or detF <= 0; Q3: CTCRW drops the drift in the latter case only
        const bool keep_drift = na || (MODEL != M_CTCRW);
        SSDE_DLOOP for (int r = 0; r < SD; r++) L.a[r] = keep_drift ? Ta[r] + drift[r] : Ta[r];
        SSDE_DLOOP for (int r = 0; r < SD; r++)
            SSDE_DLOOP for (int c = 0; c < SD; c++) L.P[r][c] = TPT[r][c];
        return;
    }
    T_ Fi[D][D];
    if (D == 1) {
        Fi[0][0] = 1.0 / F[0][0];
    } else if (D > 2) {
        dense_inverse_lu<D, T_>(F, Fi);                                                   // F.inverse(): partial-pivot LU (nllk_ctcrw.hpp:231, 236)
    } else {
        const T_ id = 1.0 / det;
        Fi[0][0] = F[1][1] * id; Fi[0][1] = -(F[0][1] * id);
        Fi[1][0] = -(F[1][0] * id); Fi[1][1] = F[0][0] * id;
    }
    T_ u[D];
    SSDE_DLOOP for (int i = 0; i < D; i++) u[i] = y[i] - L.a[DM::z(i)];                             // line 221
    T_ uFu(0.0);
    SSDE_DLOOP for (int i = 0; i < D; i++) {
        T_ s(0.0);
        SSDE_DLOOP for (int j = 0; j < D; j++) s = s + Fi[j][i] * u[j];                             // F^-T u (lines 231-232)
        uFu = uFu + u[i] * s;
    }
    L.nll = L.nll + (dlog(MODEL == M_CTCRW ? det : dfabs(det)) + uFu) * 0.5;             // line 234
    // K = (T P Z') F^-1   (line 236)
    T_ K[SD][D];
    SSDE_DLOOP for (int r = 0; r < SD; r++)
        SSDE_DLOOP for (int j = 0; j < D; j++) {
            T_ s(0.0);
            SSDE_DLOOP for (int i = 0; i < D; i++) s = s + TP[r][DM::z(i)] * Fi[i][j];
            K[r][j] = s;
        }
    // a = T a + K u + drift   (line 238)
    SSDE_DLOOP for (int r = 0; r < SD; r++) {
        T_ s = Ta[r] + drift[r];
        SSDE_DLOOP for (int j = 0; j < D; j++) s = s + K[r][j] * u[j];
        L.a[r] = s;
    }
    // P = T P (T - K Z)' + Q = (T P T' + Q) - (T P Z') K'   (lines 240-241)
    SSDE_DLOOP for (int r = 0; r < SD; r++)
        SSDE_DLOOP for (int c = 0; c < SD; c++) {
            T_ s = TPT[r][c];
            SSDE_DLOOP for (int j = 0; j < D; j++) s = s - TP[r][DM::z(j)] * K[c][j];
            L.P[r][c] = s;
        }
    // The reference propagates P as a FULL matrix (Q8).  With a measurement covariance that couples the response columns (F has
    // off-diagonal entries) the antisymmetric part rounding leaves in P is AMPLIFIED by this recursion -- ~1.2 per row for
    // H = [[.005, .002], [.002, .004]], tau = 2, nu = 1: the literal recursion in double is 4e-11 from its binary128 evaluation after
    // 100 rows, 3e-3 after 200, 6e-2 after 400, while the joint Gaussian of the track agrees with binary128 to 1e-12
    // (tests/test_oracle_golden.py::test_reference_form_loses_the_likelihood_when_H_couples_the_columns).  P is symmetric in exact
    // arithmetic: keeping it so costs nothing the reference promises and gives the value the model defines.
    if (D > 1) {
        SSDE_DLOOP for (int r = 0; r < SD; r++)
            SSDE_DLOOP for (int c = r + 1; c < SD; c++) {
                const T_ m = (L.P[r][c] + L.P[c][r]) * 0.5;
                L.P[r][c] = m; L.P[c][r] = m;
            }
    }
}

template <int MODEL, int D, int N>
SSDE_HD void dense_step(DenseLane<MODEL, D, N>& L, const DualN<N>* par, const DualN<N> (&H)[D][D], double dt,
                        const double* y, bool na) {
    dense_step_g<MODEL, D, DualN<N>>(L, par, H, dt, y, na);
}

}  // namespace ssde
#endif
