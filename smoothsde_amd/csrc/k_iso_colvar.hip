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
// One WORKGROUP of EIGHT waves (two per SIMD) per (64-track group, time window).  The waves form a pipeline over the rows, one
// barrier per row, everything between the stages in LDS rings:
//   stage 0, waves 1-6:   each loads a sixth of the channels of the row four ahead (HBM is read ONCE per row: 8 (1 + d + K)
//                         bytes), stores the row two ahead to the ring of rows, and adds ITS channels' terms of the linear
//                         predictors p1 = log tau_i, p2 = log nu_i of the row three ahead (two partial sums per wave and row);
//   stage 1, the last wave:  for the row two ahead, sums the partial predictors, takes the exp's and builds T, Q, B and their
//                         log tau derivatives (makeT/Q/B_ctcrw: nllk_ctcrw.hpp:45-91 through ctcrw_trans);
//   stage 2, wave 0:      the primal filter of the NEXT row (gains, residuals, state, covariance, likelihood terms; the
//                         log sigma_obs and drift-intercept directions), and the row's LINEARISATION: the nine numbers of
//                         Lin_i, the residuals and the seed vectors -- 15 + 3 d doubles per lane.  Its state lives in LDS
//                         between rows: held in registers it would take ~45 of EVERY wave's 256 (a kernel's allocation is the
//                         union of its waves' roles);
//   stage 3, every wave:  the column tangents of the current row from the linearisation -- straight-line code, what a column
//                         feeds is a pair of 0/1 factors on its value, not a branch -- for the up to CV_KC columns dealt to it.
// The two stage waves are long dependent chains -- the row's critical path: they stage nothing, go first on the SIMD they share
// (s_setprio) and get columns only when the six waves between them are full (the engine deals round robin).  A wave's
// registers hold its columns' state and little else, so two waves fit a SIMD and cover each other's LDS / barrier waits.
// Measured on the way here (1e4 tracks x 1e3 rows, 18 columns; lane = direction path 3.70 ms):
//   four waves, each running the primal filter, a uniform branch per column and type             2.06 ms  (the branches: 2100 of 4600 cycles per row)
//   ... straight-line columns, a transition wave                                                 1.54 ms  (> half of the VALU instructions v_accvgpr / v_readlane moves)
//   ... + a filter wave handing the linearisation on, columns in four blocks                      1.36 ms
//   eight waves (two per SIMD, 256 registers each), filter state parked in LDS                    1.26 ms
//   ... a design column both parameters use streamed once (this bench: 9 instead of 18)           1.13-1.16 ms
//   (stage waves freed of staging and columns, 0/1 factors as bit selects instead of an LDS table: 1.13 ms, no change --
//    SQ counters: 22 % of the wave cycles issue VALU, 39 % wait on s_waitcnt, 25 % issue-stalled; ~400 LDS instructions per row)
// Windows, warm-up and the verified hand-over as in k_iso.hip.  Layout: the tiles of ssde_device.hpp with the design
// columns as further channels (as k_iso_drift.hip).
//
// What else is in this file, with the same pipeline and the same interfaces (Primal: the filter + the linearisation it writes; Cols:
// the tangents that read it):
//   * drift design columns next to those of tau / nu (kinds 3, 4; the MU variants of the kernels): mixed designs;
//   * per-row measurement covariances, H_array: full 4 x 4 covariance lanes for CTCRW with two response columns
//     (CvPrimalCtcrwFull / CvColsCtcrwFull), full 2 x 2 lanes for OU_SSM / BM_SSM (Cv...ScalFull), h = H_i on the isotropic lanes
//     for one response column;
//   * iso_few_kernel: few design columns (tau ~ 1 + x) -- one wave per (group, window) runs the whole row;
//   * iso_full_kernel: H_array with CONSTANT coefficients (the Argos model) -- one wave per (group, window) runs filter and tangents;
//   * create-time helpers: column ranges, H statistics, equal-column detection, the reduction of the predictors' ranges.
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
int colvar_nstate(int model, int d, int kc, bool full) {
    if (full) return model == M_CTCRW ? 14 + kc * 14 : 5 + kc * 5;     // d = 2: 4 x 4 covariance (CTCRW), 2 x 2 (OU_SSM, BM_SSM)
    return model == M_CTCRW ? 2 * d + 5 + (3 + 2 * d) + kc * (3 + 2 * d) : d + 2 + (1 + d) + kc * (1 + d);
}

// ---- the kernel ------------------------------------------------------------------------------------------------------------
// accumulators of a part: [value | column 0 .. CV_KC-1 | mu_1 .. mu_d | log sigma_obs]   (value, mu, sigma_obs: part 0)
constexpr int CV_PRODUCER = CV_WAVES - 1;                               // the wave that builds the transitions
constexpr int CV_LOADERS = CV_WAVES - 2;                                // the waves between the two stage waves stage the rows
constexpr int CV_LD = (CV_CMAX + CV_LOADERS - 1) / CV_LOADERS;          // channels a loading wave handles per row (dt, y, H_array entries, the streamed columns)
static_assert(CV_LD * CV_LOADERS >= CV_CMAX && CV_CMAX >= 1 + 2 + 4 + DRIFT_KMAX, "ring of rows too narrow");
constexpr int CV_FILTER = 0;                                            // the wave that runs the primal filter

// KC: column slots per wave (even; the engine picks the instantiation from the widest part)
// MU: drift design columns may be among the columns (always on the full-covariance lanes, whose drift intercepts are such columns)
template <int MODEL, int D, int KC, bool FULL, bool MU>
__global__ __launch_bounds__(CV_WAVES * WAVE) void iso_colvar_kernel(const IsoArgs A, const CvPart* parts) {
    typedef typename CvModel<MODEL, D, KC, FULL>::Primal Primal;
    typedef typename CvModel<MODEL, D, KC, FULL>::Cols Cols;
    typedef typename Primal::Trans Trans;
    constexpr int SD = Primal::SD, NLIN = Primal::NLIN, NTR = Primal::NTR, NPD = Primal::NDUMP;
    __shared__ double raw[3][CV_LD * CV_LOADERS * WAVE];       // the staged rows
    constexpr int NE = MU ? 4 : 2;                             // partial sums per loading wave and row: p1, p2 (and mu_1, mu_2)
    __shared__ double eta[2][(NE * CV_LOADERS + 1) * WAVE];    // ... of every loading wave, and the interval
    __shared__ double trs[2][(NTR + (MU ? 2 : 0)) * WAVE];     // per row: the transition (and the row's drift, when it has design columns)
    __shared__ double lin[2][NLIN * WAVE];                     // per row: the linearisation
    __shared__ double fst[Primal::NSAVE * WAVE];               // the filter's state between rows (wave 0; see below)
    __shared__ double coef[DRIFT_KMAX][4];                     // per streamed column: its coefficient in p1, p2, mu_1, mu_2 (0: not in that predictor)
    __shared__ double wcoef[CV_LOADERS][CV_LD][4];
    if (blockIdx.x == 0 && threadIdx.x == 0 && A.chk_out) *A.chk_out = 0.0;
    const int lane = threadIdx.x & 63, part = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // (a scalar: the part tables are read with scalar loads)
    // the two stage waves are the row's critical path (each a long dependent chain): they neither stage rows nor, unless the
    // other six are full, carry columns
    const bool loader = part != CV_FILTER && part != CV_PRODUCER;
    if (!loader) __builtin_amdgcn_s_setprio(3);                 // (... and they go first on the SIMD they share with a column wave)
    const int ldr = loader ? part - 1 : 0;                      // (waves 1 .. 6: loaders 0 .. 5)
    const TileView& tv = A.tv;
    const int G = tv.n_groups;
    const int g = blockIdx.x % G, chunk = blockIdx.x / G;      // (groups are sorted longest first: the long ones start first)
    const int C = tv.C, c_obs = tv.c_obs, K = A.drift_k, c_col = A.c_col;
    constexpr int nacc = 2 + CV_KC + D;
    if (threadIdx.x == 0) {
#pragma unroll
        for (int k = 0; k < DRIFT_KMAX; k++) { coef[k][0] = A.coefA[k]; coef[k][1] = A.coefB[k]; coef[k][2] = A.coefC[k]; coef[k][3] = A.coefD[k]; }
    }
    __syncthreads();
    // the channels a loading wave stages are c = loader + 6 i; those that are design columns enter the linear predictors with their
    // coefficients, re-packed per wave so that the staging code reads them with one broadcast LDS load per channel
    unsigned col_bits = 0;                                     // bit i: the wave's i-th channel is a design column
#pragma unroll
    for (int i = 0; i < CV_LD; i++) {
        const int k = ldr + CV_LOADERS * i - c_col;
        const bool on = k >= 0 && k < K;
        if (on) col_bits |= 1u << i;
        if (lane < 4 && loader) wcoef[ldr][i][lane] = on ? coef[on ? k : 0][lane] : 0.0;
    }
    const bool grad = A.part_mask[0] != 0;                     // (0: the value only -- no tangents)
    const bool mu_cols = MU && A.cv_mu_cols != 0;              // the drift has design columns too
    const int n_col = grad ? parts[part].n_col : 0;
    const bool with_mu = grad && parts[CV_FILTER].with_mu, with_sig = grad && parts[CV_FILTER].with_sig;
    // per slot: the channel to read, and what the value read is -- a column of ones / a column that feeds par[d] / par[d + 1]
    // (bit masks: the selects are VALU work, of which this kernel has plenty to spare; an LDS table of 0/1 factors cost
    // 16 more LDS reads per wave and row on the LDS pipe, which it has not)
    int chan[KC];
    unsigned ones_bits = 0, t1_bits = 0, t2_bits = 0, t3_bits = 0, t4_bits = 0;      // (kinds 3, 4: the drift of dimension 1, 2 -- full-covariance lanes)
#pragma unroll
    for (int k = 0; k < KC; k++) {
        const bool on = k < n_col;
        const int ch = on ? parts[part].chan[k] : -2, ty = on ? parts[part].type[k] : 0;
        chan[k] = ch >= 0 ? ch : c_col;                        // (an unused slot reads a design column and discards it)
        if (ch == -1) ones_bits |= 1u << k;
        if (ty == 1) t1_bits |= 1u << k;
        if (ty == 2) t2_bits |= 1u << k;
        if (ty == 3) t3_bits |= 1u << k;
        if (ty == 4) t4_bits |= 1u << k;
    }
    const double* base = tv.tiles + tv.group_off[g] + lane;
    const int L = tv.group_len[g];
    const int ns = tv.lane_nsteps[g * WAVE + lane];
    int s_begin, s_acc, s_end;
    window_bounds(L, A.n_chunks, A.window, 0, chunk, s_begin, s_acc, s_end, 0);
    const int pc = part * A.n_chunks + chunk;
    double* const dump0 = A.bnd + (((int64_t)pc * G + g) * 2 + 0) * A.bnd_stride * WAVE + lane;
    double* const dump1 = A.bnd + (((int64_t)pc * G + g) * 2 + 1) * A.bnd_stride * WAVE + lane;
    const bool last_chunk = !(A.n_chunks > 1 && chunk + 1 < A.n_chunks);

    double setA[CV_LD], setB[CV_LD];
    // (a uniform row pointer + a 32-bit lane offset per channel: scalar-base addressing, the row advance is scalar arithmetic)
    const double* const gbase = tv.tiles + tv.group_off[g];
    unsigned voff[CV_LD];
#pragma unroll
    for (int i = 0; i < CV_LD; i++) voff[i] = (unsigned)((ldr + CV_LOADERS * i) * WAVE + lane);
    auto ld = [&](double (&dst)[CV_LD], int s) {               // this wave's channels of row s: HBM -> registers
        const double* rowp = gbase + (int64_t)s * C * WAVE;
#pragma unroll
        for (int i = 0; i < CV_LD; i++) dst[i] = rowp[voff[i]];       // (past the last channel: the next row's first ones -- staged, never used)
    };
    auto st_raw = [&](const double (&src)[CV_LD], int slot) {  // registers -> the ring of rows
#pragma unroll
        for (int i = 0; i < CV_LD; i++) raw[slot][(ldr + CV_LOADERS * i) * WAVE + lane] = src[i];
    };
    auto st_eta = [&](const double (&src)[CV_LD], int slot) {  // this wave's terms of the row's linear predictors
        double pa = 0.0, pb = 0.0, pm0 = 0.0, pm1 = 0.0;
#pragma unroll
        for (int i = 0; i < CV_LD; i++) {
            const double xs = ((col_bits >> i) & 1u) ? src[i] : 0.0;      // (an observation may be NaN: 0 * NaN is not 0)
            pa = fma(wcoef[ldr][i][0], xs, pa);
            if (MODEL != M_BM_SSM) pb = fma(wcoef[ldr][i][1], xs, pb);
            if (mu_cols) { pm0 = fma(wcoef[ldr][i][2], xs, pm0); if (D > 1) pm1 = fma(wcoef[ldr][i][3], xs, pm1); }
        }
        eta[slot][(NE * ldr) * WAVE + lane] = pa;
        eta[slot][(NE * ldr + 1) * WAVE + lane] = pb;
        if constexpr (MU) { if (mu_cols) { eta[slot][(NE * ldr + 2) * WAVE + lane] = pm0; eta[slot][(NE * ldr + 3) * WAVE + lane] = pm1; } }
        if (ldr == 0) eta[slot][(NE * CV_LOADERS) * WAVE + lane] = src[0];      // channel 0: the interval after the row (if the tiles hold it)
    };
    double p1_lo = INFINITY, p1_hi = -INFINITY, p2_lo = INFINITY, p2_hi = -INFINITY;      // (the transition wave: what the predictors reached)
    auto produce = [&](int slot, int s) {                      // stage 1: the transition of row s, whose sums sit in eta[slot]
        const double* e_ = &eta[slot][lane];
        double p1 = A.cv_eta0[0], p2 = A.cv_eta0[1];
#pragma unroll
        for (int w = 0; w < CV_LOADERS; w++) { p1 += e_[(NE * w) * WAVE]; p2 += e_[(NE * w + 1) * WAVE]; }
        if (s < ns) { p1_lo = fmin(p1_lo, p1); p1_hi = fmax(p1_hi, p1); p2_lo = fmin(p2_lo, p2); p2_hi = fmax(p2_hi, p2); }
        const double dtc = e_[(NE * CV_LOADERS) * WAVE];
        const double dt = c_obs ? dtc : tv.dt_all;
        Trans tr;
        Primal::trans(dt, p1, p2, tr);
        Primal::put_trans(&trs[slot][lane], tr);
        if constexpr (MU) if (mu_cols) {                       // a row-varying drift: mu_a(i) = intercept + its columns' terms, handed to the filter with the transition
            double m0 = A.mu[0], m1 = A.mu[D - 1];
#pragma unroll
            for (int w = 0; w < CV_LOADERS; w++) { m0 += e_[(NE * w + 2) * WAVE]; m1 += e_[(NE * w + 3) * WAVE]; }
            trs[slot][NTR * WAVE + lane] = m0; trs[slot][(NTR + 1) * WAVE + lane] = m1;
        }
    };
    // The filter's state lives in LDS between rows: only wave 0 ever touches it, and held in registers across the row loop it
    // would take ~45 of every wave's 256 (a kernel's allocation is the union of its waves' roles)
    Cols S;
    S.init();
    if (part == CV_FILTER) {
        Primal F;
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
        if constexpr (FULL) F.init(a0, A.cv_p0); else F.init(a0, A.p0);
        F.save(&fst[lane]);
    }
    double mu_c[D];
#pragma unroll
    for (int a = 0; a < D; a++) mu_c[a] = A.mu[a];
    const double h = A.h;
    auto filter = [&](int s, int slot3, int slot2) {           // stage 2 (wave 0): row s -- its y in raw[slot3], its transition in trs[slot2]
        Primal F;
        F.restore(&fst[lane]);
        if (s == s_acc && s_acc > s_begin) { F.dump_to(dump0); F.reset_acc(); }
        double* lo = &lin[slot2][lane];
        if (s < ns) {
            const double* r = &raw[slot3][lane];
            double y[D];
#pragma unroll
            for (int a = 0; a < D; a++) y[a] = r[(c_obs + a) * WAVE];
            Trans tr;
            Primal::get_trans(&trs[slot2][lane], tr);
            double mu[D];
#pragma unroll
            for (int a = 0; a < D; a++) { mu[a] = mu_c[a]; if constexpr (MU) { if (mu_cols) mu[a] = trs[slot2][(NTR + a) * WAVE + lane]; } }
            if constexpr (FULL) {
                double H[3] = {h, 0.0, h};                          // H_array[,,i] (symmetric, checked at create)
                if (A.cv_has_h) { H[0] = r[(c_obs + D) * WAVE]; H[1] = r[(c_obs + D + 2) * WAVE]; H[2] = r[(c_obs + D + 3) * WAVE]; }
                F.step(tr, H, mu, y, is_na(y[0], A.any_nan), lo);
            } else {
                // (d = 1 with H_array: the measurement variance of THIS row; no log sigma_obs direction then)
                const double hr = (D == 1 && A.cv_has_h) ? r[(c_obs + D) * WAVE] : h;
                F.step(tr, hr, mu, y, is_na(y[0], A.any_nan), with_sig, with_mu, lo);
            }
        }
        if (s == s_end - 1 && !last_chunk) F.dump_to(dump1);
        F.save(&fst[lane]);
    };
    auto columns = [&](int s, int slot3, int slot2) {          // stage 3: the tangents of row s (waves that carry columns)
        if (s == s_acc && s_acc > s_begin) { S.dump_to(dump0 + NPD * WAVE); S.reset_acc(); }      // (a wave without columns too: the check reads the whole record)
        if (s < ns && n_col > 0) {
            const double* r = &raw[slot3][lane];
            typename Cols::Lin li;
            li.template read<MU>(&lin[slot2][lane]);
            auto quarter = [&](auto k0) {                          // (a wave that also runs a stage is dealt fewer slots: whole quarters are skipped)
                constexpr int K0 = decltype(k0)::value, K1 = K0 + (KC + 3) / 4 < KC ? K0 + (KC + 3) / 4 : KC;
                double X[KC][4];
#pragma unroll
                for (int k = K0; k < K1; k++) {
                    const double xl = r[chan[k] * WAVE];
                    const double xk = ((ones_bits >> k) & 1u) ? 1.0 : xl;
                    X[k][0] = ((t1_bits >> k) & 1u) ? xk : 0.0; X[k][1] = ((t2_bits >> k) & 1u) ? xk : 0.0;
                    X[k][2] = X[k][3] = 0.0;
                    if constexpr (MU) { X[k][2] = ((t3_bits >> k) & 1u) ? xk : 0.0; X[k][3] = ((t4_bits >> k) & 1u) ? xk : 0.0; }
                }
                S.template step<K0, K1, MU>(li, X);
            };
            constexpr int Q = (KC + 3) / 4;
            if (n_col > 0) quarter(std::integral_constant<int, 0>());
            if (Q < KC && n_col > Q) quarter(std::integral_constant<int, (Q < KC ? Q : 0)>());
            if (2 * Q < KC && n_col > 2 * Q) quarter(std::integral_constant<int, (2 * Q < KC ? 2 * Q : 0)>());
            if (3 * Q < KC && n_col > 3 * Q) quarter(std::integral_constant<int, (3 * Q < KC ? 3 * Q : 0)>());
        }
    };
#ifdef SSDE_CV_CLOCK
    // (tuning build: where a wave's cycles go -- staging, the transition / the filter, the columns, the barrier)
    long long ck[4] = {0, 0, 0, 0};
    long long t_ = __builtin_amdgcn_s_memtime();
#define SSDE_CK(i) { const long long n_ = __builtin_amdgcn_s_memtime(); ck[i] += n_ - t_; t_ = n_; }
#else
#define SSDE_CK(i)
#endif
    // Iteration t: rows t + 2 (-> ring of rows) and t + 3 (-> partial predictors) leave the registers, row t + 4 is requested;
    // the transition of row t + 2, the filter on row t + 1, the columns of row t.  X holds row t + 2, Y row t + 3.
    int r3 = 0;                                                // (t + 3 - s_begin) mod 3 == the ring slot of row t
    auto iter = [&](int t, double (&X)[CV_LD], double (&Y)[CV_LD]) {
        const int sl_t = r3, sl_t1 = r3 == 2 ? 0 : r3 + 1, sl_t2 = r3 == 0 ? 2 : r3 - 1;   // slots of rows t, t + 1, t + 2 (t + 2 == t - 1 mod 3)
        if (loader) {
            if (t + 2 >= s_begin) st_raw(X, sl_t2);
            st_eta(Y, (t + 3) & 1);
            ld(X, t + 4);
        }
        SSDE_CK(0)
        if constexpr (!FULL) {                                 // (full-covariance lanes: the stage waves run loops of their own, below)
            if (part == CV_PRODUCER && t + 2 >= s_begin) produce((t + 2) & 1, t + 2);
            if (part == CV_FILTER && t + 1 >= s_begin && t + 1 < s_end) filter(t + 1, sl_t1, (t + 1) & 1);
        }
        SSDE_CK(1)
        if (t >= s_begin) columns(t, sl_t, t & 1);
        SSDE_CK(2)
        __syncthreads();
        SSDE_CK(3)
        r3 = r3 == 2 ? 0 : r3 + 1;
    };
    // Full-covariance lanes: the filter needs ~120 registers of its own and the column state another 120; in ONE loop the
    // allocator keeps both live and spills around the filter in every wave (measured: 800 bytes per lane of scratch, column
    // waves at 12 000 cycles per row).  The two stage waves -- which carry no columns there (the engine sees to it) -- run loops
    // of their own with the same barriers, so that neither allocation contains the other's state.
    if (FULL && part == CV_FILTER) {
        for (int t = s_begin - 3; t < s_end; t++) {
            const int sl_t1 = r3 == 2 ? 0 : r3 + 1;
            SSDE_CK(0)
            if (t + 1 >= s_begin && t + 1 < s_end) filter(t + 1, sl_t1, (t + 1) & 1);
            SSDE_CK(1)
            __syncthreads();
            SSDE_CK(3)
            r3 = r3 == 2 ? 0 : r3 + 1;
        }
    } else if (FULL && part == CV_PRODUCER) {
        for (int t = s_begin - 3; t < s_end; t++) {
            SSDE_CK(0)
            if (t + 2 >= s_begin) produce((t + 2) & 1, t + 2);
            SSDE_CK(1)
            __syncthreads();
            SSDE_CK(3)
        }
    } else {
        if (loader) ld(setB, s_begin);                         // Y of the first iteration (t = s_begin - 3): row s_begin
        iter(s_begin - 3, setA, setB);
        for (int t = s_begin - 2; t < s_end; t += 2) {         // (s_end - s_begin is a multiple of WIN_ALIGN: an even count)
            iter(t, setB, setA);
            iter(t + 1, setA, setB);
        }
    }
#ifdef SSDE_CV_CLOCK
    if (A.wave_clock && lane == 0) {
        double* o = A.wave_clock + 4 * ((int64_t)blockIdx.x * CV_WAVES + part);
        for (int i = 0; i < 4; i++) o[i] = (double)ck[i] / (double)(s_end - s_begin);
    }
#endif
    if (!last_chunk) S.dump_to(dump1 + NPD * WAVE);
    if (part == CV_PRODUCER && A.cv_ranges) {                  // per workgroup: the range of p1 and p2 over its rows (the next evaluation's window plan)
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            p1_lo = fmin(p1_lo, __shfl_xor(p1_lo, o, 64)); p1_hi = fmax(p1_hi, __shfl_xor(p1_hi, o, 64));
            p2_lo = fmin(p2_lo, __shfl_xor(p2_lo, o, 64)); p2_hi = fmax(p2_hi, __shfl_xor(p2_hi, o, 64));
        }
        if (lane == 0) { double* o_ = A.cv_ranges + 4 * (int64_t)blockIdx.x; o_[0] = p1_lo; o_[1] = p1_hi; o_[2] = p2_lo; o_[3] = p2_hi; }
    }
    const bool empty = s_acc >= s_end;
    const bool filt = part == CV_FILTER;
    Primal F;
    F.restore(&fst[lane]);                                     // (wave 0's; the other waves read it for nothing and discard it)
    {
        const double t = wave_sum((empty || !filt) ? 0.0 : F.value());
        if (lane == 0) A.partials[((int64_t)pc * nacc + 0) * G + g] = t;
    }
#pragma unroll
    for (int k = 0; k < CV_KC; k++) {
        const double t = wave_sum((empty || k >= KC) ? 0.0 : S.g[k < KC ? k : 0]);
        if (lane == 0) A.partials[((int64_t)pc * nacc + 1 + k) * G + g] = t;
    }
#pragma unroll
    for (int a = 0; a < D; a++) {
        const double t = wave_sum((empty || !filt) ? 0.0 : F.gmu[a]);
        if (lane == 0) A.partials[((int64_t)pc * nacc + 1 + CV_KC + a) * G + g] = t;
    }
    {
        const double t = wave_sum((empty || !filt) ? 0.0 : F.sg);
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

// the ranges of the linear predictors over the whole launch -> four doubles in host-visible memory (read by the next window plan)
__global__ __launch_bounds__(256) void colvar_range_reduce_kernel(const double* wg, int n_wg, double* out) {
    __shared__ double sh[4][4];
    double v[4] = {INFINITY, -INFINITY, INFINITY, -INFINITY};
    for (int i = threadIdx.x; i < n_wg; i += 256)
        for (int k = 0; k < 4; k++) v[k] = (k & 1) ? fmax(v[k], wg[4 * (int64_t)i + k]) : fmin(v[k], wg[4 * (int64_t)i + k]);
    for (int k = 0; k < 4; k++) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) { const double t = __shfl_xor(v[k], o, 64); v[k] = (k & 1) ? fmax(v[k], t) : fmin(v[k], t); }
        if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6][k] = v[k];
    }
    __syncthreads();
    if (threadIdx.x < 4) {
        const int k = threadIdx.x;
        double t = sh[0][k];
        for (int w = 1; w < 4; w++) t = (k & 1) ? fmax(t, sh[w][k]) : fmin(t, sh[w][k]);
        out[k] = t;
    }
}
hipError_t launch_colvar_range_reduce(const double* wg, int n_wg, double* out_pinned, hipStream_t s) {
    hipLaunchKernelGGL(colvar_range_reduce_kernel, dim3(1), dim3(256), 0, s, wg, n_wg, out_pinned);
    return hipGetLastError();
}

// ---- constant tau / nu with per-row H_array (CTCRW, d = 2): one wave per (64-track group, time window) ------------------------------
// The Argos model: error ellipses on every fix, one tau, one nu.  No design column to stage and at most four tangents (log tau, log nu
// and the two drift intercepts: columns of ones), so the eight-wave pipeline above is overkill -- its row takes the filter wave's
// whole dependent chain whatever the other waves do.  Here a wave runs the filter and its four tangents itself, four independent
// waves per workgroup like k_iso.hip, with the same structs: the filter writes the row's linearisation to the wave's own LDS slab
// and the tangents read it back (no barrier: one wave).  Rows are prefetched two ahead in ping-pong registers.
template <int MODEL, bool UNI>
__global__ __launch_bounds__(WG_WAVES * WAVE, 1) void iso_full_kernel(const IsoArgs A, const CvPart* parts) {
    typedef typename CvModel<MODEL, 2, 2, true>::Primal Primal;
    typedef typename CvModel<MODEL, 2, 2, true>::Cols Cols;    // slots 0, 1 of parts[0]: par[d], par[d + 1] (dP and da)
    typedef typename Primal::Trans Trans;
    constexpr int D = 2, SD = Primal::SD, NM = Cols::NMEAN, U = 2, W = 1 + D + 4;      // register block row: [dt | y | H00 H10 H01 H11]
    __shared__ double lin[WG_WAVES][Primal::NLIN * WAVE];
    if (blockIdx.x == 0 && threadIdx.x == 0 && A.chk_out) *A.chk_out = 0.0;
    int g, part, chunk;
    if (!decode_block(A, A.n_chunks, g, part, chunk)) return;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const TileView& tv = A.tv;
    const int C = tv.C, c_obs = tv.c_obs, G = tv.n_groups;
    constexpr int nacc = 2 + CV_KC + D;
    const bool grad = A.part_mask[0] != 0;
    const int n_col = grad ? parts[0].n_col : 0;
    // (the engine puts log tau / log nu into slots 0, 1 and the drift intercepts into slots 2, 3; type 0: not wanted)
    const int ty0 = n_col > 0 ? parts[0].type[0] : 0, ty1 = n_col > 1 ? parts[0].type[1] : 0;
    const bool mu0 = n_col > 2 && parts[0].type[2] == 3, mu1 = n_col > 3 && parts[0].type[3] == 4;
    const double* base = tv.tiles + tv.group_off[g] + lane;
    const int L = tv.group_len[g];
    const int ns = tv.lane_nsteps[g * WAVE + lane];
    int s_begin, s_acc, s_end;
    window_bounds(L, A.n_chunks, A.window, 0, chunk, s_begin, s_acc, s_end, 0);
    double* const dump0 = A.bnd + (((int64_t)chunk * G + g) * 2 + 0) * A.bnd_stride * WAVE + lane;
    double* const dump1 = A.bnd + (((int64_t)chunk * G + g) * 2 + 1) * A.bnd_stride * WAVE + lane;
    double bufA[U][W], bufB[U][W];
    auto load = [&](double (&dst)[U][W], int s0) {
        const double* p = base + (int64_t)s0 * C * WAVE;
#pragma unroll
        for (int u = 0; u < U; u++) {
            dst[u][0] = 0.0;
            if (!UNI) dst[u][0] = p[(u * C) * WAVE];
#pragma unroll
            for (int a = 0; a < D + 4; a++) dst[u][1 + a] = p[(u * C + c_obs + a) * WAVE];
        }
    };
    load(bufA, s_begin);
    Primal F;
    Cols S;
    S.init();
    // the drift-intercept tangents: dP stays zero (B mu does not enter the covariance), so only da' = L da + B e_a is carried
    double ma[2][NM], mg[2] = {0, 0};
#pragma unroll
    for (int i = 0; i < NM; i++) ma[0][i] = ma[1][i] = 0.0;
    {
        double a0[SD];
        if (s_begin == 0) {
#pragma unroll
            for (int c = 0; c < SD; c++) a0[c] = tv.a0[((int64_t)g * SD + c) * WAVE + lane];
        } else {
#pragma unroll
            for (int a = 0; a < D; a++) {
                const double y0 = bufA[0][1 + a];
                if constexpr (MODEL == M_CTCRW) { a0[2 * a] = (y0 == y0) ? y0 : 0.0; a0[2 * a + 1] = 0.0; }
                else a0[a] = (y0 == y0) ? y0 : 0.0;
            }
        }
        F.init(a0, A.cv_p0);
    }
    double mu[D] = {A.mu[0], A.mu[1]};
    double* lo = &lin[wv][lane];
    auto dump = [&](double* o) {
        F.dump_to(o); S.dump_to(o + Primal::NDUMP * WAVE);
#pragma unroll
        for (int j = 0; j < 2; j++)
#pragma unroll
            for (int i = 0; i < NM; i++) o[(Primal::NDUMP + 2 * Cols::NCOL + NM * j + i) * WAVE] = ma[j][i];
    };
    auto block = [&](const double (&blk)[U][W], int s0) {
        if (s0 == s_acc && s_acc > s_begin) { dump(dump0); F.reset_acc(); S.reset_acc(); mg[0] = mg[1] = 0.0; }
#pragma unroll
        for (int u = 0; u < U; u++)
            if (s0 + u < ns) {
                Trans tr;
                if constexpr (MODEL == M_CTCRW) { if constexpr (UNI) tr = A.ctr; else ctcrw_trans(blk[u][0], A.tau, A.beta, A.sigma, tr); }
                else if constexpr (MODEL == M_OU_SSM) { if constexpr (UNI) tr = A.str; else ou_trans(blk[u][0], A.tau, A.sigma, tr); }
                else { if constexpr (UNI) tr = A.str; else bm_trans(blk[u][0], A.sigma, tr); }
                const double H[3] = {blk[u][1 + D], blk[u][1 + D + 2], blk[u][1 + D + 3]};
                F.step(tr, H, mu, &blk[u][1], is_na(blk[u][1], A.any_nan), lo);
                if (n_col > 0) {
                    typename Cols::Lin li;
                    li.template read<true>(lo);
                    const double X[2][4] = {{ty0 == 1 ? 1.0 : 0.0, ty0 == 2 ? 1.0 : 0.0, 0.0, 0.0}, {ty1 == 1 ? 1.0 : 0.0, ty1 == 2 ? 1.0 : 0.0, 0.0, 0.0}};
                    S.template step<0, 2, true>(li, X);
                    Cols::mean_step(li, ma[0], mg[0], 0, mu0);
                    Cols::mean_step(li, ma[1], mg[1], 1, mu1);
                }
            }
    };
    for (int s0 = s_begin; s0 < s_end; s0 += 2 * U) {
        load(bufB, s0 + U);
        block(bufA, s0);
        load(bufA, s0 + 2 * U);
        if (s0 + U < s_end) block(bufB, s0 + U);
    }
    if (A.n_chunks > 1 && chunk + 1 < A.n_chunks) dump(dump1);
    const bool empty = s_acc >= s_end;
    const double out[nacc] = {F.value(), S.g[0], S.g[1], mg[0], mg[1], 0.0, 0.0, 0.0};
#pragma unroll
    for (int k = 0; k < nacc; k++) {
        const double t = wave_sum(empty ? 0.0 : out[k]);
        if (lane == 0) A.partials[((int64_t)chunk * nacc + k) * G + g] = t;
    }
}
// ---- row-varying tau / nu with FEW columns (H = sigma_obs^2 I): one wave per (64-track group, time window) ------------------------
// A linear covariate effect or two -- tau ~ 1 + x -- is the common case next to splines: at most CV_FEW_K streamed columns and
// CV_KC tangents besides the log sigma_obs and drift-intercept directions the filter carries itself.  The eight-wave pipeline
// spends its ~3500 cycles per row whatever the number of columns; here a wave computes its rows' predictors, exp's and transition,
// runs the filter and the tangents itself (same structs, the linearisation through the wave's own LDS slab), and four such waves
// share a CU.
// KC: tangent slots (4 or 8: slot k is slot k % CV_KC of parts[k / CV_KC], its accumulators are those of that part); KS: streamed columns
template <int MODEL, int D, int KC, int KS>
__global__ __launch_bounds__(WG_WAVES * WAVE, 1) void iso_few_kernel(const IsoArgs A, const CvPart* parts) {
    typedef typename CvModel<MODEL, D, KC, false>::Primal Primal;
    typedef typename CvModel<MODEL, D, KC, false>::Cols Cols;
    typedef typename Primal::Trans Trans;
    constexpr int SD = Primal::SD, U = KC > CV_KC ? 1 : 2, W = 1 + D + KS + KC;     // register block row: [dt | y | the streamed columns | the slots' columns]
    __shared__ double lin[WG_WAVES][Primal::NLIN * WAVE];
    if (blockIdx.x == 0 && threadIdx.x == 0 && A.chk_out) *A.chk_out = 0.0;
    int g, part, chunk;
    if (!decode_block(A, A.n_chunks, g, part, chunk)) return;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const TileView& tv = A.tv;
    const int C = tv.C, c_obs = tv.c_obs, G = tv.n_groups, K = A.drift_k, c_col = A.c_col;
    constexpr int nacc = 2 + CV_KC + D, NP = KC / CV_KC;
    const bool grad = A.part_mask[0] != 0;
    const bool with_mu = grad && parts[0].with_mu, with_sig = grad && parts[0].with_sig;
    int chan[KC], n_col = 0;                                   // per slot: the channel it reads (an unused slot or a column of ones: any column)
    unsigned ones_bits = 0, t1_bits = 0, t2_bits = 0;
#pragma unroll
    for (int k = 0; k < KC; k++) {
        const int p = k / CV_KC, kk = k % CV_KC;
        const bool on = grad && kk < parts[p].n_col;
        const int ch = on ? parts[p].chan[kk] : -2, ty = on ? parts[p].type[kk] : 0;
        chan[k] = ch >= 0 ? ch : c_col;
        if (on) n_col = k + 1;
        if (ch == -1) ones_bits |= 1u << k;
        if (ty == 1) t1_bits |= 1u << k;
        if (ty == 2) t2_bits |= 1u << k;
    }
    const double* base = tv.tiles + tv.group_off[g] + lane;
    const int L = tv.group_len[g];
    const int ns = tv.lane_nsteps[g * WAVE + lane];
    int s_begin, s_acc, s_end;
    window_bounds(L, A.n_chunks, A.window, 0, chunk, s_begin, s_acc, s_end, 0);
    double* const dump0 = A.bnd + (((int64_t)chunk * G + g) * 2 + 0) * A.bnd_stride * WAVE + lane;
    double* const dump1 = A.bnd + (((int64_t)chunk * G + g) * 2 + 1) * A.bnd_stride * WAVE + lane;
    double bufA[U][W], bufB[U][W];
    auto load = [&](double (&dst)[U][W], int s0) {
        const double* p = base + (int64_t)s0 * C * WAVE;
#pragma unroll
        for (int u = 0; u < U; u++) {
            dst[u][0] = tv.dt_all;
            if (c_obs) dst[u][0] = p[(u * C) * WAVE];
#pragma unroll
            for (int a = 0; a < D; a++) dst[u][1 + a] = p[(u * C + c_obs + a) * WAVE];
#pragma unroll
            for (int k = 0; k < KS; k++) dst[u][1 + D + k] = p[(u * C + c_col + (k < K ? k : 0)) * WAVE];
#pragma unroll
            for (int k = 0; k < KC; k++) dst[u][1 + D + KS + k] = p[(u * C + chan[k]) * WAVE];       // (the same lines again: cache hits, no selects)
        }
    };
    load(bufA, s_begin);
    Primal F;
    Cols S;
    S.init();
    {
        double a0[SD];
        if (s_begin == 0) {
#pragma unroll
            for (int c = 0; c < SD; c++) a0[c] = tv.a0[((int64_t)g * SD + c) * WAVE + lane];
        } else {
#pragma unroll
            for (int a = 0; a < D; a++) {
                const double y0 = bufA[0][1 + a];
                if constexpr (MODEL == M_CTCRW) { a0[2 * a] = (y0 == y0) ? y0 : 0.0; a0[2 * a + 1] = 0.0; }
                else a0[a] = (y0 == y0) ? y0 : 0.0;
            }
        }
        F.init(a0, A.p0);
    }
    double mu[D];
#pragma unroll
    for (int a = 0; a < D; a++) mu[a] = A.mu[a];
    const double h = A.h;
    double* lo = &lin[wv][lane];
    auto dump = [&](double* o) { F.dump_to(o); S.dump_to(o + Primal::NDUMP * WAVE); };
    auto block = [&](const double (&blk)[U][W], int s0) {
        if (s0 == s_acc && s_acc > s_begin) { dump(dump0); F.reset_acc(); S.reset_acc(); }
#pragma unroll
        for (int u = 0; u < U; u++)
            if (s0 + u < ns) {
                double p1 = A.cv_eta0[0], p2 = A.cv_eta0[1];
#pragma unroll
                for (int k = 0; k < KS; k++) {                     // (coefficients past the last column are zero)
                    p1 = fma(A.coefA[k], blk[u][1 + D + k], p1);
                    if (MODEL != M_BM_SSM) p2 = fma(A.coefB[k], blk[u][1 + D + k], p2);
                }
                Trans tr;
                Primal::trans(blk[u][0], p1, p2, tr);
                F.step(tr, h, mu, &blk[u][1], is_na(blk[u][1], A.any_nan), with_sig, with_mu, lo);
                if (n_col > 0) {
                    typename Cols::Lin li;
                    li.template read<false>(lo);
                    double X[KC][4];
#pragma unroll
                    for (int k = 0; k < KC; k++) {
                        const double xk = ((ones_bits >> k) & 1u) ? 1.0 : blk[u][1 + D + KS + k];
                        X[k][0] = ((t1_bits >> k) & 1u) ? xk : 0.0; X[k][1] = ((t2_bits >> k) & 1u) ? xk : 0.0;
                        X[k][2] = X[k][3] = 0.0;
                    }
                    S.template step<0, (KC < 4 ? KC : 4), false>(li, X);
                    if constexpr (KC > 4) { if (n_col > 4) S.template step<4, KC, false>(li, X); }
                }
            }
    };
    for (int s0 = s_begin; s0 < s_end; s0 += 2 * U) {
        load(bufB, s0 + U);
        block(bufA, s0);
        load(bufA, s0 + 2 * U);
        if (s0 + U < s_end) block(bufB, s0 + U);
    }
    if (A.n_chunks > 1 && chunk + 1 < A.n_chunks) dump(dump1);
    const bool empty = s_acc >= s_end;
#pragma unroll
    for (int p = 0; p < NP; p++) {
        const int64_t pc = (int64_t)p * A.n_chunks + chunk;
        {
            const double t = wave_sum((empty || p > 0) ? 0.0 : F.value());
            if (lane == 0) A.partials[(pc * nacc + 0) * G + g] = t;
        }
#pragma unroll
        for (int k = 0; k < CV_KC; k++) {
            const double t = wave_sum(empty ? 0.0 : S.g[p * CV_KC + k]);
            if (lane == 0) A.partials[(pc * nacc + 1 + k) * G + g] = t;
        }
#pragma unroll
        for (int a = 0; a < D; a++) {
            const double t = wave_sum((empty || p > 0) ? 0.0 : F.gmu[a]);
            if (lane == 0) A.partials[(pc * nacc + 1 + CV_KC + a) * G + g] = t;
        }
        {
            const double t = wave_sum((empty || p > 0) ? 0.0 : F.sg);
            if (lane == 0) A.partials[(pc * nacc + 1 + CV_KC + D) * G + g] = t;
        }
    }
}
// one wave per (group, window); kc = 4 or 8 tangent slots (parts[0], parts[1]), a.drift_k <= 8 streamed columns; a.n_parts is the number of
// parts the partials / the hand-over records are laid out for (1 or 2), the grid enumerates ONE work item per (group, window)
hipError_t launch_iso_few(int model, int d, const IsoArgs& a0, const CvPart* parts, int kc, hipStream_t s) {
    if (a0.cv_full || a0.cv_has_h || a0.cv_mu_cols || a0.drift_k < 1 || a0.drift_k > 2 * CV_FEW_K || (kc != CV_KC && kc != 2 * CV_KC)) return hipErrorInvalidValue;
    IsoArgs a = a0;
    a.n_parts = 1;
    const int g8 = (a.tv.n_groups + 7) / 8;
    dim3 grid((g8 * 8 * a.n_chunks + WG_WAVES - 1) / WG_WAVES), block(WG_WAVES * WAVE);
    if (grid.x == 0) return hipSuccess;
    const bool wide = kc > CV_KC || a.drift_k > CV_FEW_K;
#define SSDE_CASE(M_, D_) if (model == M_ && d == D_) { \
        if (wide) hipLaunchKernelGGL((iso_few_kernel<M_, D_, 2 * CV_KC, 2 * CV_FEW_K>), grid, block, 0, s, a, parts); \
        else hipLaunchKernelGGL((iso_few_kernel<M_, D_, CV_KC, CV_FEW_K>), grid, block, 0, s, a, parts); \
        return hipGetLastError(); }
    SSDE_CASE(M_CTCRW, 1) SSDE_CASE(M_CTCRW, 2) SSDE_CASE(M_OU_SSM, 1) SSDE_CASE(M_OU_SSM, 2) SSDE_CASE(M_BM_SSM, 1) SSDE_CASE(M_BM_SSM, 2)
#undef SSDE_CASE
    return hipErrorInvalidValue;
}

// a.n_parts == 1; parts[0]: slots 0, 1 = log tau, log nu, slots 2, 3 = the drift intercepts (type 0: not wanted); hand-over record:
// filter 14 | two tangents 2 x 14 | two drift tangents 2 x 4
hipError_t launch_iso_full(int model, const IsoArgs& a, const CvPart* parts, hipStream_t s) {
    if (a.n_parts != 1 || !a.cv_has_h) return hipErrorInvalidValue;
    const int g8 = (a.tv.n_groups + 7) / 8;
    dim3 grid((g8 * 8 * a.n_chunks + WG_WAVES - 1) / WG_WAVES), block(WG_WAVES * WAVE);
    if (grid.x == 0) return hipSuccess;
#define SSDE_CASE(M_) if (model == M_) { if (a.uniform_dt) hipLaunchKernelGGL((iso_full_kernel<M_, true>), grid, block, 0, s, a, parts); \
                                         else hipLaunchKernelGGL((iso_full_kernel<M_, false>), grid, block, 0, s, a, parts); return hipGetLastError(); }
    SSDE_CASE(M_CTCRW) SSDE_CASE(M_OU_SSM) SSDE_CASE(M_BM_SSM)
#undef SSDE_CASE
    return hipErrorInvalidValue;
}

// per group: the largest diagonal entry of H_array[,,i] over its rows, and the largest |H01 - H10| (create time: the window
// planner's observation variance; the full-covariance lanes take a symmetric H)
__global__ __launch_bounds__(WG_WAVES * WAVE) void colvar_h_stats_kernel(TileView tv, int c_h, int d, double* out /* [n_groups][2] */) {
    __shared__ double sh[WG_WAVES][2];
    const int g = blockIdx.x, lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const double* base = tv.tiles + tv.group_off[g] + lane;
    const int ns = tv.lane_nsteps[g * WAVE + lane];
    double hmax = 0.0, asym = 0.0;
    for (int s = wv; s < ns; s += WG_WAVES) {
        const double* p = base + ((int64_t)s * tv.C + c_h) * WAVE;
        const double h00 = p[0], h10 = d == 2 ? p[WAVE] : 0.0, h01 = d == 2 ? p[2 * WAVE] : 0.0, h11 = d == 2 ? p[3 * WAVE] : p[0];
        hmax = fmax(hmax, fmax(h00, h11));
        asym = fmax(asym, fabs(h01 - h10));
        if (!(h00 == h00) || !(h11 == h11) || !(h01 == h01) || !(h10 == h10)) asym = INFINITY;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { hmax = fmax(hmax, __shfl_xor(hmax, o, 64)); asym = fmax(asym, __shfl_xor(asym, o, 64)); }
    if (lane == 0) { sh[wv][0] = hmax; sh[wv][1] = asym; }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < WG_WAVES; w++) { hmax = fmax(hmax, sh[w][0]); asym = fmax(asym, sh[w][1]); }
        out[2 * g] = hmax; out[2 * g + 1] = asym;
    }
}
hipError_t launch_colvar_h_stats(const TileView& tv, int c_h, int d, double* out, hipStream_t s) {
    if (tv.n_groups == 0) return hipSuccess;
    hipLaunchKernelGGL(colvar_h_stats_kernel, dim3(tv.n_groups), dim3(WG_WAVES * WAVE), 0, s, tv, c_h, d, out);
    return hipGetLastError();
}

// are two design columns (device arrays of n doubles) the same numbers?  *differ is raised if not (create time)
__global__ __launch_bounds__(256) void cols_differ_kernel(const double* a, const double* b, int64_t n, int* differ) {
    bool d = false;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const long long x = __double_as_longlong(a[i]), y = __double_as_longlong(b[i]);
        d = d || x != y;
    }
    if (__any(d) && (threadIdx.x & 63) == 0) atomicOr(differ, 1);
}
hipError_t launch_cols_differ(const double* a, const double* b, int64_t n, int* differ, hipStream_t s) {
    hipLaunchKernelGGL(cols_differ_kernel, dim3(1024), dim3(256), 0, s, a, b, n, differ);
    return hipGetLastError();
}

// a.n_parts == CV_WAVES parts (one per wave of a workgroup), a.drift_k streamed columns (1 .. DRIFT_KMAX), kc: the widest
// part's column count
template <int MODEL, int D>
static hipError_t launch_cv(const IsoArgs& a, const CvPart* parts, int kc, dim3 grid, dim3 block, hipStream_t s) {
    if (a.cv_mu_cols) {
        if (kc <= 2) hipLaunchKernelGGL((iso_colvar_kernel<MODEL, D, 2, false, true>), grid, block, 0, s, a, parts);
        else hipLaunchKernelGGL((iso_colvar_kernel<MODEL, D, 4, false, true>), grid, block, 0, s, a, parts);
    } else {
        if (kc <= 2) hipLaunchKernelGGL((iso_colvar_kernel<MODEL, D, 2, false, false>), grid, block, 0, s, a, parts);
        else hipLaunchKernelGGL((iso_colvar_kernel<MODEL, D, 4, false, false>), grid, block, 0, s, a, parts);
    }
    return hipGetLastError();
}
hipError_t launch_iso_colvar(int model, int d, const IsoArgs& a, const CvPart* parts, int kc, hipStream_t s) {
    if (a.n_parts != CV_WAVES || a.drift_k < 0 || a.drift_k > DRIFT_KMAX || a.tv.C > CV_CMAX || kc < 0 || kc > CV_KC) return hipErrorInvalidValue;
    dim3 grid(a.tv.n_groups * a.n_chunks), block(CV_WAVES * WAVE);
    if (grid.x == 0) return hipSuccess;
    if (a.cv_full) {                                           // full-covariance lanes, d = 2: 4 x 4 (CTCRW), 2 x 2 (OU_SSM, BM_SSM)
        if (d != 2) return hipErrorInvalidValue;
        if (model == M_CTCRW) {
            if (kc <= 2) hipLaunchKernelGGL((iso_colvar_kernel<M_CTCRW, 2, 2, true, true>), grid, block, 0, s, a, parts);
            else if (kc <= 3) hipLaunchKernelGGL((iso_colvar_kernel<M_CTCRW, 2, 3, true, true>), grid, block, 0, s, a, parts);
            else hipLaunchKernelGGL((iso_colvar_kernel<M_CTCRW, 2, 4, true, true>), grid, block, 0, s, a, parts);
        } else if (model == M_OU_SSM) hipLaunchKernelGGL((iso_colvar_kernel<M_OU_SSM, 2, 4, true, true>), grid, block, 0, s, a, parts);
        else if (model == M_BM_SSM) hipLaunchKernelGGL((iso_colvar_kernel<M_BM_SSM, 2, 4, true, true>), grid, block, 0, s, a, parts);
        else return hipErrorInvalidValue;
        return hipGetLastError();
    }
#define SSDE_CASE(M_, D_) if (model == M_ && d == D_) return launch_cv<M_, D_>(a, parts, kc, grid, block, s);
    SSDE_CASE(M_CTCRW, 1) SSDE_CASE(M_CTCRW, 2) SSDE_CASE(M_OU_SSM, 1) SSDE_CASE(M_OU_SSM, 2) SSDE_CASE(M_BM_SSM, 1) SSDE_CASE(M_BM_SSM, 2)
#undef SSDE_CASE
    return hipErrorInvalidValue;
}

}  // namespace ssde
