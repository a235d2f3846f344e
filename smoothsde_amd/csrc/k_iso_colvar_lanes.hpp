// k_iso_colvar_lanes.hpp -- the lanes of the row-varying tau / nu kernels (see k_iso_colvar.hip for the design): for every model a
// Primal (the lane = track filter of one row + the LINEARISATION of its update and prediction that it hands on) and a Cols (the
// tangents of up to KC design columns that read that linearisation).  Shared by the eight-wave pipeline (k_iso_colvar.hip) and the
// one-wave kernels (k_iso_onewave.hip: iso_full_kernel, iso_few_kernel); split out of one 1,581-line file in round 4.
#ifndef SSDE_K_ISO_COLVAR_LANES_HPP
#define SSDE_K_ISO_COLVAR_LANES_HPP
#include <type_traits>

#include "ssde_device.hpp"

namespace ssde {

// ---- CTCRW: the primal filter and the linearisation it hands to the column waves ----------------------------------------------
template <int D>
struct CvPrimalCtcrw {
    static constexpr int SD = 2 * D;
    static constexpr int NLIN = 17 + 3 * D;
    static constexpr int NCOL = 3 + 2 * D;                     // doubles of a tangent
    static constexpr int NDUMP = SD + 3 + 2 + NCOL;
    typedef CtcrwTrans Trans;
    double x[D], v[D], p11, p12, p22;
    LogAcc ld;
    double accq;
    double mx, mv, gmu[D];
    double s11, s12, s22, stx[D], stv[D], sg;                  // the log sigma_obs tangent

    __device__ __forceinline__ void init(const double* a0, const double* p0) {
#pragma unroll
        for (int a = 0; a < D; a++) { x[a] = a0[2 * a]; v[a] = a0[2 * a + 1]; gmu[a] = 0.0; stx[a] = stv[a] = 0.0; }
        p11 = p0[0]; p12 = p0[1]; p22 = p0[2];
        ld.init(); accq = 0.0; mx = mv = 0.0; s11 = s12 = s22 = sg = 0.0;
    }
    __device__ __forceinline__ void reset_acc() {
        ld.init(); accq = 0.0; sg = 0.0;
#pragma unroll
        for (int a = 0; a < D; a++) gmu[a] = 0.0;
    }
    // One row: score y (unless NA), then the prediction over the row's interval (ctcrw_step's arrangement: filtered-form
    // covariance update, Joseph-form sensitivities; ssde_math.hpp).  lin[j * WAVE]: the row's linearisation (LDS).
    __device__ __forceinline__ void step(const CtcrwTrans& tr, double h, const double* mu, const double* y, bool na, bool with_sig,
                                         bool with_mu, double* lin) {
        const double F = p11 + h;
        const double detF = (D == 1) ? F : F * F;                  // nllk_ctcrw.hpp:16-19, 223
        const bool upd = !na && !(detF <= 0.0);                    // :214, 226
        const double updf = upd ? 1.0 : 0.0;
        const double Fe = upd ? F : 1.0;
        const double iF = rcp(Fe) * updf;
        ld.mul(Fe);
        const double bm = (na || upd) ? 1.0 : 0.0;                 // Q3 (:226-228)
        const double e = tr.e, t12 = tr.t12, e2 = tr.e2;
        const double a = fma(h, iF, 1.0 - updf), aiF = a * iF;
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
        int n = 0;
        lin[(n++) * WAVE] = iF; lin[(n++) * WAVE] = a; lin[(n++) * WAVE] = aiF; lin[(n++) * WAVE] = kf2; lin[(n++) * WAVE] = t12;
        lin[(n++) * WAVE] = e; lin[(n++) * WAVE] = c1; lin[(n++) * WAVE] = k2; lin[(n++) * WAVE] = gF;
#pragma unroll
        for (int a_ = 0; a_ < D; a_++) lin[(n++) * WAVE] = u[a_];
        // seeds: log tau moves T, B and Q ...
        lin[(n++) * WAVE] = fma(tr.dt12x2, m, tr.dq11);
        lin[(n++) * WAVE] = fma(tr.dt12e, f22, fma(tr.de, m, tr.dq12));
        lin[(n++) * WAVE] = fma(tr.edex2, f22, tr.dq22);
#pragma unroll
        for (int a_ = 0; a_ < D; a_++) {
            const double w = fma(kf2, u[a_], v[a_] - mue[a_]);     // d k u + d(T a + B mu): (dt12, de) (kf2 u + v - mu)
            lin[(n++) * WAVE] = tr.dt12 * w; lin[(n++) * WAVE] = tr.de * w;
        }
        // ... log nu Q only (dQ = 2 Q), a drift column B e_a only
        lin[(n++) * WAVE] = 2.0 * tr.q11; lin[(n++) * WAVE] = 2.0 * tr.q12; lin[(n++) * WAVE] = 2.0 * tr.q22;
        lin[(n++) * WAVE] = bm * tr.b1; lin[(n++) * WAVE] = bm * tr.b2;
        if (with_sig) {
            // log sigma_obs: dh = 2 h enters F, the filtered covariance (k k' dh) and the gain (-k dh / F)
            const double h2 = 2.0 * h, dF = s11 + h2;
            double sud = 0.0;
#pragma unroll
            for (int a_ = 0; a_ < D; a_++) sud = fma(u[a_], stx[a_], sud);
            sg = fma(gF, dF, fma(-iF, sud, sg));
            const double w = fma(-kf2, s11, s12);
            const double q1 = kf1 * h2, q2 = kf2 * h2;
            const double g11 = fma(kf1, q1, a * a * s11), g12 = fma(kf2, q1, a * w), g22 = fma(kf2, q2, fma(-kf2, s12 + w, s22));
            const double dkf1 = fma(-q1, iF, s11 * aiF), dkf2 = fma(-q2, iF, w * iF);
            const double dm = fma(t12, g22, g12);
            s11 = fma(t12, g12 + dm, g11); s12 = e * dm; s22 = e2 * g22;
            const double dk1 = fma(t12, dkf2, dkf1), dk2 = e * dkf2;
#pragma unroll
            for (int a_ = 0; a_ < D; a_++) {
                const double txk = stx[a_], tvk = stv[a_];
                stx[a_] = fma(dk1, u[a_], fma(t12, tvk, c1 * txk));
                stv[a_] = fma(dk2, u[a_], fma(e, tvk, -k2 * txk));
            }
        }
        if (with_mu) {                                             // d / d mu_a: one data-independent chain for every dimension
            const double imx = iF * mx;
            const double nx = fma(bm, tr.b1, fma(t12, mv, c1 * mx)), nv = fma(bm, tr.b2, fma(e, mv, -k2 * mx));
#pragma unroll
            for (int a_ = 0; a_ < D; a_++) gmu[a_] = fma(-imx, u[a_], gmu[a_]);
            mx = nx; mv = nv;
        }
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
        o[(n++) * WAVE] = s11; o[(n++) * WAVE] = s12; o[(n++) * WAVE] = s22;
#pragma unroll
        for (int a = 0; a < D; a++) { o[(n++) * WAVE] = stx[a]; o[(n++) * WAVE] = stv[a]; }
    }
    static constexpr int NSAVE = SD + 3 + 2 + 1 + 2 + D + NCOL + 1;
    __device__ __forceinline__ void save(double* o) const {          // o[k * WAVE]: everything, accumulators included
        int n = 0;
#pragma unroll
        for (int a = 0; a < D; a++) { o[(n++) * WAVE] = x[a]; o[(n++) * WAVE] = v[a]; o[(n++) * WAVE] = gmu[a]; o[(n++) * WAVE] = stx[a]; o[(n++) * WAVE] = stv[a]; }
        o[(n++) * WAVE] = p11; o[(n++) * WAVE] = p12; o[(n++) * WAVE] = p22; o[(n++) * WAVE] = mx; o[(n++) * WAVE] = mv;
        o[(n++) * WAVE] = accq; o[(n++) * WAVE] = ld.m; o[(n++) * WAVE] = (double)ld.e;
        o[(n++) * WAVE] = s11; o[(n++) * WAVE] = s12; o[(n++) * WAVE] = s22; o[(n++) * WAVE] = sg;
    }
    __device__ __forceinline__ void restore(const double* o) {
        int n = 0;
#pragma unroll
        for (int a = 0; a < D; a++) { x[a] = o[(n++) * WAVE]; v[a] = o[(n++) * WAVE]; gmu[a] = o[(n++) * WAVE]; stx[a] = o[(n++) * WAVE]; stv[a] = o[(n++) * WAVE]; }
        p11 = o[(n++) * WAVE]; p12 = o[(n++) * WAVE]; p22 = o[(n++) * WAVE]; mx = o[(n++) * WAVE]; mv = o[(n++) * WAVE];
        accq = o[(n++) * WAVE]; ld.m = o[(n++) * WAVE]; ld.e = (int)o[(n++) * WAVE];
        s11 = o[(n++) * WAVE]; s12 = o[(n++) * WAVE]; s22 = o[(n++) * WAVE]; sg = o[(n++) * WAVE];
    }
    __device__ __forceinline__ double value() const { return 0.5 * ((double)D * ld.value() + accq); }
    // this row's transition from the linear predictors p1 = log tau, p2 = log nu (nllk_ctcrw.hpp:152-156)
    static __device__ __forceinline__ void trans(double dt, double p1, double p2, CtcrwTrans& tr) {
        const double tau = exp(p1), nu = exp(p2);
        const double beta = rcp(tau);
        ctcrw_trans(dt, tau, beta, 2.0 * nu / sqrt(M_PI * tau), tr);
    }
    static constexpr int NTR = 12;
    static __device__ __forceinline__ void put_trans(double* o, const CtcrwTrans& t) {      // o[j * WAVE]
        o[0 * WAVE] = t.e; o[1 * WAVE] = t.t12; o[2 * WAVE] = t.b1; o[3 * WAVE] = t.b2; o[4 * WAVE] = t.q11; o[5 * WAVE] = t.q12;
        o[6 * WAVE] = t.q22; o[7 * WAVE] = t.de; o[8 * WAVE] = t.dt12; o[9 * WAVE] = t.dq11; o[10 * WAVE] = t.dq12; o[11 * WAVE] = t.dq22;
    }
    static __device__ __forceinline__ void get_trans(const double* o, CtcrwTrans& t) {
        t.e = o[0 * WAVE]; t.t12 = o[1 * WAVE]; t.b1 = o[2 * WAVE]; t.b2 = o[3 * WAVE]; t.q11 = o[4 * WAVE]; t.q12 = o[5 * WAVE];
        t.q22 = o[6 * WAVE]; t.de = o[7 * WAVE]; t.dt12 = o[8 * WAVE]; t.dq11 = o[9 * WAVE]; t.dq12 = o[10 * WAVE]; t.dq22 = o[11 * WAVE];
        t.e2 = t.e * t.e; t.dt12x2 = 2.0 * t.dt12; t.dt12e = t.dt12 * t.e; t.edex2 = 2.0 * t.e * t.de;      // as ctcrw_trans forms them
    }
};

// the column tangents of a wave, CTCRW
template <int D, int KC>
struct CvColsCtcrw {
    static constexpr int NCOL = 3 + 2 * D;
    double d11[KC], d12[KC], d22[KC], tx[KC][D], tv[KC][D], g[KC];
    __device__ __forceinline__ void init() {
#pragma unroll
        for (int k = 0; k < KC; k++) {
            d11[k] = d12[k] = d22[k] = g[k] = 0.0;
#pragma unroll
            for (int a = 0; a < D; a++) tx[k][a] = tv[k][a] = 0.0;
        }
    }
    __device__ __forceinline__ void reset_acc() {
#pragma unroll
        for (int k = 0; k < KC; k++) g[k] = 0.0;
    }
    struct Lin {                                               // a row's linearisation, read from LDS once per row
        double iF, a, aiF, kf2, t12, e, c1, k2, gF, u[D], s1_11, s1_12, s1_22, s1_x[D], s1_v[D], s2_11, s2_12, s2_22, sb1, sb2;
        template <bool MU>
        __device__ __forceinline__ void read(const double* lin) {
            int n = 0;
            iF = lin[(n++) * WAVE]; a = lin[(n++) * WAVE]; aiF = lin[(n++) * WAVE]; kf2 = lin[(n++) * WAVE]; t12 = lin[(n++) * WAVE];
            e = lin[(n++) * WAVE]; c1 = lin[(n++) * WAVE]; k2 = lin[(n++) * WAVE]; gF = lin[(n++) * WAVE];
#pragma unroll
            for (int a_ = 0; a_ < D; a_++) u[a_] = lin[(n++) * WAVE];
            s1_11 = lin[(n++) * WAVE]; s1_12 = lin[(n++) * WAVE]; s1_22 = lin[(n++) * WAVE];
#pragma unroll
            for (int a_ = 0; a_ < D; a_++) { s1_x[a_] = lin[(n++) * WAVE]; s1_v[a_] = lin[(n++) * WAVE]; }
            s2_11 = lin[(n++) * WAVE]; s2_12 = lin[(n++) * WAVE]; s2_22 = lin[(n++) * WAVE];
            sb1 = sb2 = 0.0;
            if constexpr (MU) { sb1 = lin[(n++) * WAVE]; sb2 = lin[(n++) * WAVE]; }
        }
    };
    // slots [K0, K1): X[k][j] = the column's value if it is of kind j (0: feeds log tau, 1: log nu, 2: mu_1, 3: mu_2), else 0;
    // MU: drift columns may be among them
    template <int K0, int K1, bool MU>
    __device__ __forceinline__ void step(const Lin& L, const double (*X)[4]) {
        const double iF = L.iF, a = L.a, aiF = L.aiF, kf2 = L.kf2, t12 = L.t12, e = L.e, c1 = L.c1, k2 = L.k2, gF = L.gF;
        const double* u = L.u; const double* s1_x = L.s1_x; const double* s1_v = L.s1_v;
        const double s1_11 = L.s1_11, s1_12 = L.s1_12, s1_22 = L.s1_22, s2_11 = L.s2_11, s2_12 = L.s2_12, s2_22 = L.s2_22;
        const double a2 = a * a, e2 = e * e;
#pragma unroll
        for (int k = K0; k < K1; k++) {
            const double c11 = d11[k], c12 = d12[k], c22 = d22[k];
            double sud = 0.0;
#pragma unroll
            for (int a_ = 0; a_ < D; a_++) sud = fma(u[a_], tx[k][a_], sud);
            g[k] = fma(gF, c11, fma(-iF, sud, g[k]));
            const double w = fma(-kf2, c11, c12);
            const double g11 = a2 * c11, g12 = a * w, g22 = fma(-kf2, c12 + w, c22);
            const double dkf1 = c11 * aiF, dkf2 = w * iF;
            const double dm = fma(t12, g22, g12);
            const double dk1 = fma(t12, dkf2, dkf1), dk2 = e * dkf2;
            const double x1 = X[k][0], x2 = X[k][1];
            d11[k] = fma(x2, s2_11, fma(x1, s1_11, fma(t12, g12 + dm, g11)));
            d12[k] = fma(x2, s2_12, fma(x1, s1_12, e * dm));
            d22[k] = fma(x2, s2_22, fma(x1, s1_22, e2 * g22));
#pragma unroll
            for (int a_ = 0; a_ < D; a_++) {
                const double txk = tx[k][a_], tvk = tv[k][a_];
                double nx = fma(x1, s1_x[a_], fma(dk1, u[a_], fma(t12, tvk, c1 * txk)));
                double nv = fma(x1, s1_v[a_], fma(dk2, u[a_], fma(e, tvk, -k2 * txk)));
                if constexpr (MU) { nx = fma(X[k][2 + a_], L.sb1, nx); nv = fma(X[k][2 + a_], L.sb2, nv); }      // (a drift column of dimension a_: B e_a)
                tx[k][a_] = nx; tv[k][a_] = nv;
            }
        }
    }
    __device__ __forceinline__ void dump_to(double* o) const {
        int n = 0;
#pragma unroll
        for (int k = 0; k < KC; k++) {
            o[(n++) * WAVE] = d11[k]; o[(n++) * WAVE] = d12[k]; o[(n++) * WAVE] = d22[k];
#pragma unroll
            for (int a = 0; a < D; a++) { o[(n++) * WAVE] = tx[k][a]; o[(n++) * WAVE] = tv[k][a]; }
        }
    }
};

// ---- CTCRW, d = 2, FULL 4 x 4 covariance: a per-row measurement covariance H_i (H_array, nllk_ctcrw.hpp:203-205) couples the two
// dimensions (and P0 may be anything).  State s = (x0, v0, x1, v1), Z picks components 0 and 2, P symmetric (10 numbers).  With
// M = T P Z', K = M F^-1, L = T - K Z (nllk_ctcrw.hpp:236-241: P' = T P L' + Q) the tangent of a row is
//     da' = L (da + dP Z' w) + X seed_a          dP' = L dP L' + X seed_P          d nllk = <C, Z dP Z'> - w' Z da
// with w = F^-1 u, C = (F^-1 - w w') / 2, seed_P = dT P L' + L P dT' + dQ, seed_a = dT (a + P Z' w) + dB mu.  No log sigma_obs
// direction (H_i holds no parameter); the drift intercepts are columns of ones of their own kinds (seed_a = B e_a, seed_P = 0).
struct CvPrimalCtcrwFull {
    static constexpr int D = 2, SD = 4, NLIN = 34, NCOL = 14, NDUMP = 14, NTR = 12, NSAVE = 14 + 3;
    typedef CtcrwTrans Trans;
    double a[4], p[10];                                        // p: 00 01 02 03 11 12 13 22 23 33
    LogAcc ld;
    double accq;
    double gmu[2], sg;                                         // (unused here: the kernel's epilogue reads them)
    __device__ __forceinline__ void init(const double* a0, const double* p0f) {
#pragma unroll
        for (int i = 0; i < 4; i++) a[i] = a0[i];
        int n = 0;
#pragma unroll
        for (int i = 0; i < 4; i++)
#pragma unroll
            for (int j = i; j < 4; j++) p[n++] = p0f[i + 4 * j];
        ld.init(); accq = 0.0; gmu[0] = gmu[1] = sg = 0.0;
    }
    __device__ __forceinline__ void reset_acc() { ld.init(); accq = 0.0; }
    // y[2], H = (H00, H01, H11) of this row; lin[j * WAVE]: the row's linearisation
    __device__ __forceinline__ void step(const CtcrwTrans& tr, const double* H, const double* mu, const double* y, bool na, double* lin) {
        const double p00 = p[0], p01 = p[1], p02 = p[2], p03 = p[3], p11 = p[4], p12 = p[5], p13 = p[6], p22 = p[7], p23 = p[8], p33 = p[9];
        const double F11 = p00 + H[0], F12 = p02 + H[1], F22 = p22 + H[2];
        const double detF = fma(F11, F22, -F12 * F12);             // det(): nllk_ctcrw.hpp:16-19
        const bool upd = !na && !(detF <= 0.0);                    // :214, 226
        const double updf = upd ? 1.0 : 0.0;
        const double dete = upd ? detF : 1.0;
        const double idet = rcp(dete) * updf;
        ld.mul(dete);                                              // (log detF itself: value() counts it once)
        const double bm = (na || upd) ? 1.0 : 0.0;                 // Q3 (:226-228)
        const double i11 = F22 * idet, i12 = -F12 * idet, i22 = F11 * idet;
        const double e = tr.e, t = tr.t12;
        const double u0 = upd ? y[0] - a[0] : 0.0, u1 = upd ? y[1] - a[2] : 0.0;
        const double w0 = fma(i11, u0, i12 * u1), w1 = fma(i12, u0, i22 * u1);
        accq = fma(u0, w0, fma(u1, w1, accq));
        // M = T P Z' (4 x 2), K = M F^-1
        const double m00 = fma(t, p01, p00), m10 = e * p01, m20 = fma(t, p03, p02), m30 = e * p03;
        const double m01 = fma(t, p12, p02), m11 = e * p12, m21 = fma(t, p23, p22), m31 = e * p23;
        const double k00 = fma(m00, i11, m01 * i12), k01 = fma(m00, i12, m01 * i22);
        const double k10 = fma(m10, i11, m11 * i12), k11 = fma(m10, i12, m11 * i22);
        const double k20 = fma(m20, i11, m21 * i12), k21 = fma(m20, i12, m21 * i22);
        const double k30 = fma(m30, i11, m31 * i12), k31 = fma(m30, i12, m31 * i22);
        // L = T - K Z: rows (l00, t, l02, 0), (l10, e, l12, 0), (l20, 0, l22, t), (l30, 0, l32, e)
        const double l00 = 1.0 - k00, l02 = -k01, l10 = -k10, l12 = -k11, l20 = -k20, l22 = 1.0 - k21, l30 = -k30, l32 = -k31;
        const double mue0 = bm * mu[0], mue1 = bm * mu[1];
        int n = 0;
        lin[(n++) * WAVE] = l00; lin[(n++) * WAVE] = l02; lin[(n++) * WAVE] = l10; lin[(n++) * WAVE] = l12;
        lin[(n++) * WAVE] = l20; lin[(n++) * WAVE] = l22; lin[(n++) * WAVE] = l30; lin[(n++) * WAVE] = l32;
        lin[(n++) * WAVE] = t; lin[(n++) * WAVE] = e; lin[(n++) * WAVE] = w0; lin[(n++) * WAVE] = w1;
        lin[(n++) * WAVE] = 0.5 * fma(-w0, w0, i11); lin[(n++) * WAVE] = fma(-w0, w1, i12); lin[(n++) * WAVE] = 0.5 * fma(-w1, w1, i22);
        // seed_P of log tau: N = dT P (rows dt12 P1, de P1, dt12 P3, de P3), S[i][j] = N_i . L_j + N_j . L_i + dQ[i][j]
        {
            const double P1[4] = {p01, p11, p12, p13}, P3[4] = {p03, p13, p23, p33};
            const double a1[4] = {fma(l00, P1[0], fma(t, P1[1], l02 * P1[2])), fma(l10, P1[0], fma(e, P1[1], l12 * P1[2])),
                                  fma(l20, P1[0], fma(l22, P1[2], t * P1[3])), fma(l30, P1[0], fma(l32, P1[2], e * P1[3]))};      // P1 . L_j
            const double a3[4] = {fma(l00, P3[0], fma(t, P3[1], l02 * P3[2])), fma(l10, P3[0], fma(e, P3[1], l12 * P3[2])),
                                  fma(l20, P3[0], fma(l22, P3[2], t * P3[3])), fma(l30, P3[0], fma(l32, P3[2], e * P3[3]))};      // P3 . L_j
            const double dt12 = tr.dt12, de = tr.de;
            // N_i . L_j: i = 0: dt12 a1[j]; 1: de a1[j]; 2: dt12 a3[j]; 3: de a3[j]
            lin[(n++) * WAVE] = fma(2.0 * dt12, a1[0], tr.dq11);                        // (0,0)
            lin[(n++) * WAVE] = fma(dt12, a1[1], fma(de, a1[0], tr.dq12));              // (0,1)
            lin[(n++) * WAVE] = fma(dt12, a1[2], dt12 * a3[0]);                         // (0,2)
            lin[(n++) * WAVE] = fma(dt12, a1[3], de * a3[0]);                           // (0,3)
            lin[(n++) * WAVE] = fma(2.0 * de, a1[1], tr.dq22);                          // (1,1)
            lin[(n++) * WAVE] = fma(de, a1[2], dt12 * a3[1]);                           // (1,2)
            lin[(n++) * WAVE] = fma(de, a1[3], de * a3[1]);                             // (1,3)
            lin[(n++) * WAVE] = fma(2.0 * dt12, a3[2], tr.dq11);                        // (2,2)
            lin[(n++) * WAVE] = fma(dt12, a3[3], fma(de, a3[2], tr.dq12));              // (2,3)
            lin[(n++) * WAVE] = fma(2.0 * de, a3[3], tr.dq22);                          // (3,3)
            // seed_a of log tau: dT (a + P Z' w) + dB mu
            const double s0 = a[1] - mue0 + fma(p01, w0, p12 * w1), s1 = a[3] - mue1 + fma(p03, w0, p23 * w1);
            lin[(n++) * WAVE] = dt12 * s0; lin[(n++) * WAVE] = de * s0; lin[(n++) * WAVE] = dt12 * s1; lin[(n++) * WAVE] = de * s1;
        }
        lin[(n++) * WAVE] = 2.0 * tr.q11; lin[(n++) * WAVE] = 2.0 * tr.q12; lin[(n++) * WAVE] = 2.0 * tr.q22;       // seed_P of log nu
        lin[(n++) * WAVE] = bm * tr.b1; lin[(n++) * WAVE] = bm * tr.b2;                                            // seed_a of mu_a
        // a' = T a + K u + B mu (:238)
        const double n0 = fma(tr.b1, mue0, fma(k00, u0, fma(k01, u1, fma(t, a[1], a[0]))));
        const double n1 = fma(tr.b2, mue0, fma(k10, u0, fma(k11, u1, e * a[1])));
        const double n2 = fma(tr.b1, mue1, fma(k20, u0, fma(k21, u1, fma(t, a[3], a[2]))));
        const double n3 = fma(tr.b2, mue1, fma(k30, u0, fma(k31, u1, e * a[3])));
        a[0] = n0; a[1] = n1; a[2] = n2; a[3] = n3;
        // P' = T P T' - M K' + Q (:240-241, symmetric F): A = T P, then A T'
        const double A00 = fma(t, p01, p00), A01 = fma(t, p11, p01), A02 = fma(t, p12, p02), A03 = fma(t, p13, p03);
        const double A11 = e * p11, A12 = e * p12, A13 = e * p13;
        const double A22 = fma(t, p23, p22), A23 = fma(t, p33, p23);
        const double A33 = e * p33;
        p[0] = fma(t, A01, A00) - fma(m00, k00, m01 * k01) + tr.q11;
        p[1] = e * A01 - fma(m00, k10, m01 * k11) + tr.q12;
        p[2] = fma(t, A03, A02) - fma(m00, k20, m01 * k21);
        p[3] = e * A03 - fma(m00, k30, m01 * k31);
        p[4] = e * A11 - fma(m10, k10, m11 * k11) + tr.q22;
        p[5] = fma(t, A13, A12) - fma(m10, k20, m11 * k21);
        p[6] = e * A13 - fma(m10, k30, m11 * k31);
        p[7] = fma(t, A23, A22) - fma(m20, k20, m21 * k21) + tr.q11;
        p[8] = e * A23 - fma(m20, k30, m21 * k31) + tr.q12;
        p[9] = e * A33 - fma(m30, k30, m31 * k31) + tr.q22;
    }
    __device__ __forceinline__ void dump_to(double* o) const {
#pragma unroll
        for (int i = 0; i < 4; i++) o[i * WAVE] = a[i];
#pragma unroll
        for (int i = 0; i < 10; i++) o[(4 + i) * WAVE] = p[i];
    }
    __device__ __forceinline__ void save(double* o) const {
        dump_to(o);
        o[14 * WAVE] = accq; o[15 * WAVE] = ld.m; o[16 * WAVE] = (double)ld.e;
    }
    __device__ __forceinline__ void restore(const double* o) {
#pragma unroll
        for (int i = 0; i < 4; i++) a[i] = o[i * WAVE];
#pragma unroll
        for (int i = 0; i < 10; i++) p[i] = o[(4 + i) * WAVE];
        accq = o[14 * WAVE]; ld.m = o[15 * WAVE]; ld.e = (int)o[16 * WAVE];
        gmu[0] = gmu[1] = sg = 0.0;
    }
    __device__ __forceinline__ double value() const { return 0.5 * (ld.value() + accq); }      // (ld holds log det F of both dimensions)
    static __device__ __forceinline__ void trans(double dt, double p1, double p2, CtcrwTrans& tr) { CvPrimalCtcrw<2>::trans(dt, p1, p2, tr); }
    static __device__ __forceinline__ void put_trans(double* o, const CtcrwTrans& t) { CvPrimalCtcrw<2>::put_trans(o, t); }
    static __device__ __forceinline__ void get_trans(const double* o, CtcrwTrans& t) { CvPrimalCtcrw<2>::get_trans(o, t); }
};

// the column tangents of a wave, CTCRW d = 2, full covariance: dP (10) and da (4) per column.  X[k][j]: the column's value if it is
// of kind j (0: feeds log tau, 1: log nu, 2: mu_1, 3: mu_2), else 0
template <int KC>
struct CvColsCtcrwFull {
    static constexpr int NCOL = 14;
    double dp[KC][10], da[KC][4], g[KC];
    __device__ __forceinline__ void init() {
#pragma unroll
        for (int k = 0; k < KC; k++) {
            g[k] = 0.0;
#pragma unroll
            for (int i = 0; i < 10; i++) dp[k][i] = 0.0;
#pragma unroll
            for (int i = 0; i < 4; i++) da[k][i] = 0.0;
        }
    }
    __device__ __forceinline__ void reset_acc() {
#pragma unroll
        for (int k = 0; k < KC; k++) g[k] = 0.0;
    }
    // (the nine numbers of L, w and C stay in registers for the row; the seed vectors -- 19 doubles -- are read from LDS where a
    //  column needs them: with them resident a wave of four columns spilled 800 bytes per lane to scratch)
    struct Lin {
        double l00, l02, l10, l12, l20, l22, l30, l32, t, e, w0, w1, c00, c02, c22;
        const double* seeds;                                   // s1[10] | sa[4] | s2[3] | sb[2], each [j * WAVE]
        template <bool MU>
        __device__ __forceinline__ void read(const double* lin) {
            int n = 0;
            l00 = lin[(n++) * WAVE]; l02 = lin[(n++) * WAVE]; l10 = lin[(n++) * WAVE]; l12 = lin[(n++) * WAVE];
            l20 = lin[(n++) * WAVE]; l22 = lin[(n++) * WAVE]; l30 = lin[(n++) * WAVE]; l32 = lin[(n++) * WAVE];
            t = lin[(n++) * WAVE]; e = lin[(n++) * WAVE]; w0 = lin[(n++) * WAVE]; w1 = lin[(n++) * WAVE];
            c00 = lin[(n++) * WAVE]; c02 = lin[(n++) * WAVE]; c22 = lin[(n++) * WAVE];
            seeds = lin + n * WAVE;
        }
        __device__ __forceinline__ double s1(int i) const { return seeds[i * WAVE]; }
        __device__ __forceinline__ double sa(int i) const { return seeds[(10 + i) * WAVE]; }
        __device__ __forceinline__ double s2(int i) const { return seeds[(14 + i) * WAVE]; }
        __device__ __forceinline__ double sb(int i) const { return seeds[(17 + i) * WAVE]; }
    };
    template <int K0, int K1, bool MU>
    __device__ __forceinline__ void step(const Lin& L, const double (*X)[4]) {
#pragma unroll
        for (int k = K0; k < K1; k++) {
            const double* d = dp[k];                               // 00 01 02 03 11 12 13 22 23 33
            const double x1 = X[k][0], x2 = X[k][1], x3 = X[k][2], x4 = X[k][3];
            g[k] = fma(L.c00, d[0], fma(L.c02, d[2], fma(L.c22, d[7], fma(-L.w0, da[k][0], fma(-L.w1, da[k][2], g[k])))));
            // z = da + dP Z' w
            const double z0 = fma(d[0], L.w0, fma(d[2], L.w1, da[k][0])), z1 = fma(d[1], L.w0, fma(d[5], L.w1, da[k][1]));
            const double z2 = fma(d[2], L.w0, fma(d[7], L.w1, da[k][2])), z3 = fma(d[3], L.w0, fma(d[8], L.w1, da[k][3]));
            da[k][0] = fma(x3, L.sb(0), fma(x1, L.sa(0), fma(L.l00, z0, fma(L.t, z1, L.l02 * z2))));
            da[k][1] = fma(x3, L.sb(1), fma(x1, L.sa(1), fma(L.l10, z0, fma(L.e, z1, L.l12 * z2))));
            da[k][2] = fma(x4, L.sb(0), fma(x1, L.sa(2), fma(L.l20, z0, fma(L.l22, z2, L.t * z3))));
            da[k][3] = fma(x4, L.sb(1), fma(x1, L.sa(3), fma(L.l30, z0, fma(L.l32, z2, L.e * z3))));
            // R = L dP L' (symmetric), row by row: G_i = L_i dP (a 4-vector), R[i][j] = G_i . L_j for j >= i
            const double q0 = d[0], q1 = d[1], q2 = d[2], q3 = d[3], q4 = d[4], q5 = d[5], q6 = d[6], q7 = d[7], q8 = d[8], q9 = d[9];
            {
                const double G0 = fma(L.l00, q0, fma(L.t, q1, L.l02 * q2)), G1 = fma(L.l00, q1, fma(L.t, q4, L.l02 * q5));
                const double G2 = fma(L.l00, q2, fma(L.t, q5, L.l02 * q7)), G3 = fma(L.l00, q3, fma(L.t, q6, L.l02 * q8));
                dp[k][0] = fma(x2, L.s2(0), fma(x1, L.s1(0), fma(L.l00, G0, fma(L.t, G1, L.l02 * G2))));
                dp[k][1] = fma(x2, L.s2(1), fma(x1, L.s1(1), fma(L.l10, G0, fma(L.e, G1, L.l12 * G2))));
                dp[k][2] = fma(x1, L.s1(2), fma(L.l20, G0, fma(L.l22, G2, L.t * G3)));
                dp[k][3] = fma(x1, L.s1(3), fma(L.l30, G0, fma(L.l32, G2, L.e * G3)));
            }
            {
                const double G0 = fma(L.l10, q0, fma(L.e, q1, L.l12 * q2)), G1 = fma(L.l10, q1, fma(L.e, q4, L.l12 * q5));
                const double G2 = fma(L.l10, q2, fma(L.e, q5, L.l12 * q7)), G3 = fma(L.l10, q3, fma(L.e, q6, L.l12 * q8));
                dp[k][4] = fma(x2, L.s2(2), fma(x1, L.s1(4), fma(L.l10, G0, fma(L.e, G1, L.l12 * G2))));
                dp[k][5] = fma(x1, L.s1(5), fma(L.l20, G0, fma(L.l22, G2, L.t * G3)));
                dp[k][6] = fma(x1, L.s1(6), fma(L.l30, G0, fma(L.l32, G2, L.e * G3)));
            }
            {
                const double G0 = fma(L.l20, q0, fma(L.l22, q2, L.t * q3)), G2 = fma(L.l20, q2, fma(L.l22, q7, L.t * q8));
                const double G3 = fma(L.l20, q3, fma(L.l22, q8, L.t * q9));
                dp[k][7] = fma(x2, L.s2(0), fma(x1, L.s1(7), fma(L.l20, G0, fma(L.l22, G2, L.t * G3))));
                dp[k][8] = fma(x2, L.s2(1), fma(x1, L.s1(8), fma(L.l30, G0, fma(L.l32, G2, L.e * G3))));
            }
            {
                const double G0 = fma(L.l30, q0, fma(L.l32, q2, L.e * q3)), G2 = fma(L.l30, q2, fma(L.l32, q7, L.e * q8));
                const double G3 = fma(L.l30, q3, fma(L.l32, q8, L.e * q9));
                dp[k][9] = fma(x2, L.s2(2), fma(x1, L.s1(9), fma(L.l30, G0, fma(L.l32, G2, L.e * G3))));
            }
        }
    }
    // a drift-intercept tangent (dimension `dim`): dP stays zero (B mu does not enter the covariance), da' = L da + B e_dim
    static constexpr int NMEAN = 4;
    static __device__ __forceinline__ void mean_step(const Lin& L, double* m, double& mg, int dim, bool on) {
        const double z0 = m[0], z1 = m[1], z2 = m[2], z3 = m[3];
        mg = fma(-L.w0, z0, fma(-L.w1, z2, mg));
        const double b1 = on ? L.sb(0) : 0.0, b2 = on ? L.sb(1) : 0.0;
        m[0] = fma(L.l00, z0, fma(L.t, z1, L.l02 * z2)) + (dim == 0 ? b1 : 0.0);
        m[1] = fma(L.l10, z0, fma(L.e, z1, L.l12 * z2)) + (dim == 0 ? b2 : 0.0);
        m[2] = fma(L.l20, z0, fma(L.l22, z2, L.t * z3)) + (dim == 1 ? b1 : 0.0);
        m[3] = fma(L.l30, z0, fma(L.l32, z2, L.e * z3)) + (dim == 1 ? b2 : 0.0);
    }
    __device__ __forceinline__ void dump_to(double* o) const {
        int n = 0;
#pragma unroll
        for (int k = 0; k < KC; k++) {
#pragma unroll
            for (int i = 0; i < 10; i++) o[(n++) * WAVE] = dp[k][i];
#pragma unroll
            for (int i = 0; i < 4; i++) o[(n++) * WAVE] = da[k][i];
        }
    }
};

// ---- OU_SSM / BM_SSM: scalar covariance --------------------------------------------------------------------------------
template <int D, bool HAS_P2>
struct CvPrimalScal {
    static constexpr int SD = D;
    static constexpr int NLIN = 8 + 2 * D;
    static constexpr int NCOL = 1 + D;
    static constexpr int NDUMP = SD + 1 + 1 + NCOL;
    typedef ScalTrans Trans;
    double x[D], p;
    LogAcc ld;
    double accq;
    double mx, gmu[D];
    double sp, stx[D], sg;                                     // the log sigma_obs tangent

    __device__ __forceinline__ void init(const double* a0, const double* p0) {
#pragma unroll
        for (int a = 0; a < D; a++) { x[a] = a0[a]; gmu[a] = 0.0; stx[a] = 0.0; }
        p = p0[0];
        ld.init(); accq = 0.0; mx = 0.0; sp = sg = 0.0;
    }
    __device__ __forceinline__ void reset_acc() {
        ld.init(); accq = 0.0; sg = 0.0;
#pragma unroll
        for (int a = 0; a < D; a++) gmu[a] = 0.0;
    }
    // scal_cov_step + scal_mean_step (ssde_math.hpp)
    __device__ __forceinline__ void step(const ScalTrans& tr, double h, const double* mu, const double* y, bool na, bool with_sig,
                                         bool with_mu, double* lin) {
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
        int n = 0;
        lin[(n++) * WAVE] = iF; lin[(n++) * WAVE] = ca; lin[(n++) * WAVE] = tca; lin[(n++) * WAVE] = c; lin[(n++) * WAVE] = gF;
#pragma unroll
        for (int a_ = 0; a_ < D; a_++) lin[(n++) * WAVE] = u[a_];
        lin[(n++) * WAVE] = HAS_P2 ? fma(2.0 * dt_, cp, tr.dq) : tr.dq;
#pragma unroll
        for (int a_ = 0; a_ < D; a_++) lin[(n++) * WAVE] = HAS_P2 ? fma(s1_k, u[a_], fma(tr.dt_, x[a_], tr.db * mu[a_])) : 0.0;
        lin[(n++) * WAVE] = tr.q;                                   // log kappa (OU)
        lin[(n++) * WAVE] = tr.b;                                   // a drift column: b e_a
        if (with_sig) {
            const double h2 = 2.0 * h, bh = b * h2, dF = sp + h2;
            double sud = 0.0;
#pragma unroll
            for (int a_ = 0; a_ < D; a_++) sud = fma(u[a_], stx[a_], sud);
            sg = fma(gF, dF, fma(-iF, sud, sg));
            const double dk = fma(-tiF, bh, ca * sp);
            sp = fma(k * t, bh, tca * sp);
#pragma unroll
            for (int a_ = 0; a_ < D; a_++) stx[a_] = fma(dk, u[a_], c * stx[a_]);
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
        o[(n++) * WAVE] = sp;
#pragma unroll
        for (int a = 0; a < D; a++) o[(n++) * WAVE] = stx[a];
    }
    static constexpr int NSAVE = 3 * D + 1 + 1 + 1 + 2 + 2;
    __device__ __forceinline__ void save(double* o) const {          // o[k * WAVE]: everything, accumulators included
        int n = 0;
#pragma unroll
        for (int a = 0; a < D; a++) { o[(n++) * WAVE] = x[a]; o[(n++) * WAVE] = gmu[a]; o[(n++) * WAVE] = stx[a]; }
        o[(n++) * WAVE] = p; o[(n++) * WAVE] = mx; o[(n++) * WAVE] = accq; o[(n++) * WAVE] = ld.m; o[(n++) * WAVE] = (double)ld.e;
        o[(n++) * WAVE] = sp; o[(n++) * WAVE] = sg;
    }
    __device__ __forceinline__ void restore(const double* o) {
        int n = 0;
#pragma unroll
        for (int a = 0; a < D; a++) { x[a] = o[(n++) * WAVE]; gmu[a] = o[(n++) * WAVE]; stx[a] = o[(n++) * WAVE]; }
        p = o[(n++) * WAVE]; mx = o[(n++) * WAVE]; accq = o[(n++) * WAVE]; ld.m = o[(n++) * WAVE]; ld.e = (int)o[(n++) * WAVE];
        sp = o[(n++) * WAVE]; sg = o[(n++) * WAVE];
    }
    __device__ __forceinline__ double value() const { return 0.5 * ((double)D * ld.value() + accq); }
    static __device__ __forceinline__ void trans(double dt, double p1, double p2, ScalTrans& tr) {
        if constexpr (HAS_P2) ou_trans(dt, exp(p1), exp(p2), tr);     // nllk_ou_ssm.hpp:121-124
        else bm_trans(dt, exp(p1), tr);                               // nllk_bm_ssm.hpp:106-108
    }
    static constexpr int NTR = 6;
    static __device__ __forceinline__ void put_trans(double* o, const ScalTrans& t) {
        o[0 * WAVE] = t.t; o[1 * WAVE] = t.b; o[2 * WAVE] = t.q; o[3 * WAVE] = t.dt_; o[4 * WAVE] = t.db; o[5 * WAVE] = t.dq;
    }
    static __device__ __forceinline__ void get_trans(const double* o, ScalTrans& t) {
        t.t = o[0 * WAVE]; t.b = o[1 * WAVE]; t.q = o[2 * WAVE]; t.dt_ = o[3 * WAVE]; t.db = o[4 * WAVE]; t.dq = o[5 * WAVE];
    }
};

template <int D, int KC, bool HAS_P2>
struct CvColsScal {
    static constexpr int NCOL = 1 + D;
    double dp[KC], tx[KC][D], g[KC];
    __device__ __forceinline__ void init() {
#pragma unroll
        for (int k = 0; k < KC; k++) {
            dp[k] = g[k] = 0.0;
#pragma unroll
            for (int a = 0; a < D; a++) tx[k][a] = 0.0;
        }
    }
    __device__ __forceinline__ void reset_acc() {
#pragma unroll
        for (int k = 0; k < KC; k++) g[k] = 0.0;
    }
    struct Lin {
        double iF, ca, tca, c, gF, u[D], s1_p, s1_x[D], s2_p, sb;
        template <bool MU>
        __device__ __forceinline__ void read(const double* lin) {
            int n = 0;
            iF = lin[(n++) * WAVE]; ca = lin[(n++) * WAVE]; tca = lin[(n++) * WAVE]; c = lin[(n++) * WAVE]; gF = lin[(n++) * WAVE];
#pragma unroll
            for (int a_ = 0; a_ < D; a_++) u[a_] = lin[(n++) * WAVE];
            s1_p = lin[(n++) * WAVE];
#pragma unroll
            for (int a_ = 0; a_ < D; a_++) s1_x[a_] = lin[(n++) * WAVE];
            s2_p = lin[(n++) * WAVE]; sb = 0.0;
            if constexpr (MU) sb = lin[(n++) * WAVE];
        }
    };
    template <int K0, int K1, bool MU>
    __device__ __forceinline__ void step(const Lin& L, const double (*X)[4]) {
        const double iF = L.iF, ca = L.ca, tca = L.tca, c = L.c, gF = L.gF, s1_p = L.s1_p, s2_p = L.s2_p;
        const double* u = L.u; const double* s1_x = L.s1_x;
#pragma unroll
        for (int k = K0; k < K1; k++) {
            const double cdp = dp[k];
            double sud = 0.0;
#pragma unroll
            for (int a_ = 0; a_ < D; a_++) sud = fma(u[a_], tx[k][a_], sud);
            g[k] = fma(gF, cdp, fma(-iF, sud, g[k]));
            const double dk = ca * cdp;
            const double x1 = X[k][0], x2 = X[k][1];
            dp[k] = HAS_P2 ? fma(x2, s2_p, fma(x1, s1_p, tca * cdp)) : fma(x1, s1_p, tca * cdp);
#pragma unroll
            for (int a_ = 0; a_ < D; a_++) {
                double nx = fma(dk, u[a_], c * tx[k][a_]);
                if constexpr (MU) nx = fma(X[k][2 + a_], L.sb, nx);
                tx[k][a_] = HAS_P2 ? fma(x1, s1_x[a_], nx) : nx;
            }
        }
    }
    __device__ __forceinline__ void dump_to(double* o) const {
        int n = 0;
#pragma unroll
        for (int k = 0; k < KC; k++) {
            o[(n++) * WAVE] = dp[k];
#pragma unroll
            for (int a = 0; a < D; a++) o[(n++) * WAVE] = tx[k][a];
        }
    }
};

// ---- OU_SSM / BM_SSM, d = 2, FULL 2 x 2 covariance (per-row H_array, nllk_ou_ssm.hpp:171-172, nllk_bm_ssm.hpp:135-136): T = t I,
// B = b I, Q = q I, Z = I, so M = t P, K = t P F^-1, L = t I - K; the tangent formulas are those of the CTCRW lanes above.
template <bool HAS_P2>
struct CvPrimalScalFull {
    static constexpr int D = 2, SD = 2, NLIN = 16, NCOL = 5, NDUMP = 5, NTR = 6, NSAVE = 5 + 3;
    typedef ScalTrans Trans;
    double a[2], p[3];                                         // p: 00 01 11
    LogAcc ld;
    double accq;
    double gmu[2], sg;                                         // (unused here: the pipeline kernel's epilogue reads them)
    __device__ __forceinline__ void init(const double* a0, const double* p0f) {
        a[0] = a0[0]; a[1] = a0[1];
        p[0] = p0f[0]; p[1] = p0f[2]; p[2] = p0f[3];               // (column-major 2 x 2)
        ld.init(); accq = 0.0; gmu[0] = gmu[1] = sg = 0.0;
    }
    __device__ __forceinline__ void reset_acc() { ld.init(); accq = 0.0; }
    __device__ __forceinline__ void step(const ScalTrans& tr, const double* H, const double* mu, const double* y, bool na, double* lin) {
        const double p00 = p[0], p01 = p[1], p11 = p[2];
        const double F11 = p00 + H[0], F12 = p01 + H[1], F22 = p11 + H[2];
        const double detF = fma(F11, F22, -F12 * F12);
        const bool upd = !na && !(fabs(detF) <= 0.0);              // nllk_ou_ssm.hpp:190-195, nllk_bm_ssm.hpp:152-157 (the drift stays in every branch)
        const double updf = upd ? 1.0 : 0.0;
        const double dete = upd ? detF : 1.0;
        const double idet = rcp(dete) * updf;
        ld.mul(dete);
        const double i11 = F22 * idet, i12 = -F12 * idet, i22 = F11 * idet;
        const double t = HAS_P2 ? tr.t : 1.0;
        const double u0 = upd ? y[0] - a[0] : 0.0, u1 = upd ? y[1] - a[1] : 0.0;
        const double w0 = fma(i11, u0, i12 * u1), w1 = fma(i12, u0, i22 * u1);
        accq = fma(u0, w0, fma(u1, w1, accq));
        // K = t P F^-1, L = t I - K
        const double k00 = t * fma(p00, i11, p01 * i12), k01 = t * fma(p00, i12, p01 * i22);
        const double k10 = t * fma(p01, i11, p11 * i12), k11 = t * fma(p01, i12, p11 * i22);
        const double l00 = t - k00, l01 = -k01, l10 = -k10, l11 = t - k11;
        int n = 0;
        lin[(n++) * WAVE] = l00; lin[(n++) * WAVE] = l01; lin[(n++) * WAVE] = l10; lin[(n++) * WAVE] = l11;
        lin[(n++) * WAVE] = w0; lin[(n++) * WAVE] = w1;
        lin[(n++) * WAVE] = 0.5 * fma(-w0, w0, i11); lin[(n++) * WAVE] = fma(-w0, w1, i12); lin[(n++) * WAVE] = 0.5 * fma(-w1, w1, i22);
        // seed_P of par[d] (log tau: dT = dt_ I, dQ = dq I; BM_SSM log sigma: dQ only): dt_ (P L' + L P) + dq I
        const double dt_ = HAS_P2 ? tr.dt_ : 0.0;
        lin[(n++) * WAVE] = fma(2.0 * dt_, fma(l00, p00, l01 * p01), tr.dq);
        lin[(n++) * WAVE] = dt_ * (fma(l10, p00, l11 * p01) + fma(l00, p01, l01 * p11));
        lin[(n++) * WAVE] = fma(2.0 * dt_, fma(l10, p01, l11 * p11), tr.dq);
        // seed_a of par[d]: dT (a + P w) + dB mu
        lin[(n++) * WAVE] = HAS_P2 ? fma(dt_, a[0] + fma(p00, w0, p01 * w1), tr.db * mu[0]) : 0.0;
        lin[(n++) * WAVE] = HAS_P2 ? fma(dt_, a[1] + fma(p01, w0, p11 * w1), tr.db * mu[1]) : 0.0;
        lin[(n++) * WAVE] = tr.q;                                  // seed_P of log kappa: q I
        lin[(n++) * WAVE] = tr.b;                                  // seed_a of mu_a: b e_a
        // a' = T a + K u + B mu; P' = T P T' - M K' + Q with M = t P
        const double n0 = fma(tr.b, mu[0], fma(k00, u0, fma(k01, u1, t * a[0]))), n1 = fma(tr.b, mu[1], fma(k10, u0, fma(k11, u1, t * a[1])));
        a[0] = n0; a[1] = n1;
        const double tt = t * t;
        p[0] = fma(tt, p00, -t * fma(p00, k00, p01 * k01)) + tr.q;
        p[1] = fma(tt, p01, -t * fma(p00, k10, p01 * k11));
        p[2] = fma(tt, p11, -t * fma(p01, k10, p11 * k11)) + tr.q;
    }
    __device__ __forceinline__ void dump_to(double* o) const { o[0] = a[0]; o[WAVE] = a[1]; o[2 * WAVE] = p[0]; o[3 * WAVE] = p[1]; o[4 * WAVE] = p[2]; }
    __device__ __forceinline__ void save(double* o) const { dump_to(o); o[5 * WAVE] = accq; o[6 * WAVE] = ld.m; o[7 * WAVE] = (double)ld.e; }
    __device__ __forceinline__ void restore(const double* o) {
        a[0] = o[0]; a[1] = o[WAVE]; p[0] = o[2 * WAVE]; p[1] = o[3 * WAVE]; p[2] = o[4 * WAVE];
        accq = o[5 * WAVE]; ld.m = o[6 * WAVE]; ld.e = (int)o[7 * WAVE];
        gmu[0] = gmu[1] = sg = 0.0;
    }
    __device__ __forceinline__ double value() const { return 0.5 * (ld.value() + accq); }
    static __device__ __forceinline__ void trans(double dt, double p1, double p2, ScalTrans& tr) { CvPrimalScal<2, HAS_P2>::trans(dt, p1, p2, tr); }
    static __device__ __forceinline__ void put_trans(double* o, const ScalTrans& t) { CvPrimalScal<2, HAS_P2>::put_trans(o, t); }
    static __device__ __forceinline__ void get_trans(const double* o, ScalTrans& t) { CvPrimalScal<2, HAS_P2>::get_trans(o, t); }
};

template <int KC, bool HAS_P2>
struct CvColsScalFull {
    static constexpr int NCOL = 5;
    double dp[KC][3], da[KC][2], g[KC];
    __device__ __forceinline__ void init() {
#pragma unroll
        for (int k = 0; k < KC; k++) { g[k] = 0.0; dp[k][0] = dp[k][1] = dp[k][2] = 0.0; da[k][0] = da[k][1] = 0.0; }
    }
    __device__ __forceinline__ void reset_acc() {
#pragma unroll
        for (int k = 0; k < KC; k++) g[k] = 0.0;
    }
    struct Lin {
        double l00, l01, l10, l11, w0, w1, c00, c01, c11, s1[3], sa[2], s2, b;
        template <bool MU>
        __device__ __forceinline__ void read(const double* lin) {
            int n = 0;
            l00 = lin[(n++) * WAVE]; l01 = lin[(n++) * WAVE]; l10 = lin[(n++) * WAVE]; l11 = lin[(n++) * WAVE];
            w0 = lin[(n++) * WAVE]; w1 = lin[(n++) * WAVE]; c00 = lin[(n++) * WAVE]; c01 = lin[(n++) * WAVE]; c11 = lin[(n++) * WAVE];
            s1[0] = lin[(n++) * WAVE]; s1[1] = lin[(n++) * WAVE]; s1[2] = lin[(n++) * WAVE];
            sa[0] = lin[(n++) * WAVE]; sa[1] = lin[(n++) * WAVE]; s2 = lin[(n++) * WAVE]; b = lin[(n++) * WAVE];
        }
    };
    template <int K0, int K1, bool MU>
    __device__ __forceinline__ void step(const Lin& L, const double (*X)[4]) {
#pragma unroll
        for (int k = K0; k < K1; k++) {
            const double q0 = dp[k][0], q1 = dp[k][1], q2 = dp[k][2];
            const double x1 = X[k][0], x2 = X[k][1], x3 = X[k][2], x4 = X[k][3];
            g[k] = fma(L.c00, q0, fma(L.c01, q1, fma(L.c11, q2, fma(-L.w0, da[k][0], fma(-L.w1, da[k][1], g[k])))));
            const double z0 = fma(q0, L.w0, fma(q1, L.w1, da[k][0])), z1 = fma(q1, L.w0, fma(q2, L.w1, da[k][1]));
            da[k][0] = fma(x3, L.b, fma(x1, L.sa[0], fma(L.l00, z0, L.l01 * z1)));
            da[k][1] = fma(x4, L.b, fma(x1, L.sa[1], fma(L.l10, z0, L.l11 * z1)));
            // L dP L'
            const double G00 = fma(L.l00, q0, L.l01 * q1), G01 = fma(L.l00, q1, L.l01 * q2);
            const double G10 = fma(L.l10, q0, L.l11 * q1), G11 = fma(L.l10, q1, L.l11 * q2);
            dp[k][0] = fma(x2, L.s2, fma(x1, L.s1[0], fma(G00, L.l00, G01 * L.l01)));
            dp[k][1] = fma(x1, L.s1[1], fma(G00, L.l10, G01 * L.l11));
            dp[k][2] = fma(x2, L.s2, fma(x1, L.s1[2], fma(G10, L.l10, G11 * L.l11)));
        }
    }
    static constexpr int NMEAN = 2;
    static __device__ __forceinline__ void mean_step(const Lin& L, double* m, double& mg, int dim, bool on) {
        const double z0 = m[0], z1 = m[1];
        mg = fma(-L.w0, z0, fma(-L.w1, z1, mg));
        const double b = on ? L.b : 0.0;
        m[0] = fma(L.l00, z0, L.l01 * z1) + (dim == 0 ? b : 0.0);
        m[1] = fma(L.l10, z0, L.l11 * z1) + (dim == 1 ? b : 0.0);
    }
    __device__ __forceinline__ void dump_to(double* o) const {
        int n = 0;
#pragma unroll
        for (int k = 0; k < KC; k++) {
            o[(n++) * WAVE] = dp[k][0]; o[(n++) * WAVE] = dp[k][1]; o[(n++) * WAVE] = dp[k][2];
            o[(n++) * WAVE] = da[k][0]; o[(n++) * WAVE] = da[k][1];
        }
    }
};

template <int MODEL, int D, int KC, bool FULL>
struct CvModel;
template <int D, int KC>
struct CvModel<M_CTCRW, D, KC, false> { typedef CvPrimalCtcrw<D> Primal; typedef CvColsCtcrw<D, KC> Cols; };
template <int D, int KC>
struct CvModel<M_OU_SSM, D, KC, false> { typedef CvPrimalScal<D, true> Primal; typedef CvColsScal<D, KC, true> Cols; };
template <int D, int KC>
struct CvModel<M_BM_SSM, D, KC, false> { typedef CvPrimalScal<D, false> Primal; typedef CvColsScal<D, KC, false> Cols; };
template <int KC>
struct CvModel<M_CTCRW, 2, KC, true> { typedef CvPrimalCtcrwFull Primal; typedef CvColsCtcrwFull<KC> Cols; };
template <int KC>
struct CvModel<M_OU_SSM, 2, KC, true> { typedef CvPrimalScalFull<true> Primal; typedef CvColsScalFull<KC, true> Cols; };
template <int KC>
struct CvModel<M_BM_SSM, 2, KC, true> { typedef CvPrimalScalFull<false> Primal; typedef CvColsScalFull<KC, false> Cols; };

// components of a part's hand-over dump with kc column slots: the filter's block (written by part 0), then the columns
}  // namespace ssde
#endif
