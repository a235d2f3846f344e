// k_iso_shared.hip -- shared-covariance kernels for gfx950: constant coefficients, regular time
// grid, track groups without missing rows (CTCRW, OU_SSM, BM_SSM).
//
// In that regime the covariance half of the Kalman filter (P, F, K and all their sensitivities;
// ssde_math.hpp ctcrw_cov_step / scal_cov_step) does not depend on the observations and is the
// same for every track, so ssde_engine.hip evaluates it ONCE per evaluation into a small gain
// table that becomes stationary after the filter's transient (tens of rows).  The lanes run the
// mean half only:
//   * rows inside the transient read their gains from the table (wave-uniform addresses),
//   * rows past it use the stationary gains held in scalar registers, with the update written as
//     the minimal FMA chain (~62 fp64 instructions per row for 2-D CTCRW with three
//     covariance directions, against ~245 for the general per-lane filter) and the quadratic
//     forms accumulated as plain sums that are scaled once at the end.
// Observations stream from the tiled HBM layout with coalesced 512-B wave loads, prefetched
// SHARED_U rows ahead into registers; the dt channel is never read (16 B/row actual traffic
// for 24 B/row of algorithmic input).  Time windows and their hand-over check work exactly as
// in k_iso.hip, except that only the mean state needs a warm-up (the gains are exact).
#include <hip/hip_ext.h>

#include "ssde_device.hpp"

namespace ssde {

// C = channels per step of the tile, c_obs = channel of the first obs column (a dt channel, if present, is not read)
template <int D>
__device__ __forceinline__ void load_obs_block(double (&dst)[SHARED_U][D], const double* p, int C, int c_obs) {
#pragma unroll
    for (int u = 0; u < SHARED_U; u++)
#pragma unroll
#ifdef SSDE_DIAG_NOSTREAM   // timing-only diagnostic build (results are wrong): every block re-reads one 8-KB window
        for (int a = 0; a < D; a++) dst[u][a] = p[((u * C + c_obs + a) * WAVE) & 1023];
#else
        for (int a = 0; a < D; a++) dst[u][a] = __builtin_nontemporal_load(&p[(u * C + c_obs + a) * WAVE]);   // read once
#endif
}

// ---- CTCRW -------------------------------------------------------------------------------------
template <int D, int MASK>
struct SharedCtcrw {
    static constexpr int SD = 2 * D;
    static constexpr int NSTATE = shared_nstate(2 * D, MASK, true);
    CtcrwMean<D, MASK> M;          // state + table-phase accumulators
    double acc2, sacc[NDIRP], macc[D];  // stationary-phase sums: u'u, u' tx_j, u_a mx_a
    // stationary constants (wave-uniform)
    double k1, k2, c1, t12, e, iF, hd[NDIRP], dk1[NDIRP], dk2[NDIRP], dt12, de, cb1, cb2;
    double cx[D], cv[D], bmu[D];
    CtcrwTrans tr;

    __device__ __forceinline__ void setup(const IsoArgs& A) {
        tr = A.ctr;
        const double* c = A.statc;   // static indices into the kernel argument block: scalar loads
        iF = c[0]; k1 = c[1]; k2 = c[2]; c1 = c[3]; t12 = c[4]; e = c[5]; dt12 = c[6]; de = c[7]; cb1 = c[8]; cb2 = c[9];
#pragma unroll
        for (int j = 0; j < NDIRP; j++) { hd[j] = c[10 + j]; dk1[j] = c[13 + j]; dk2[j] = c[16 + j]; }
#pragma unroll
        for (int a = 0; a < D; a++) {
            cx[a] = c[19 + a]; bmu[a] = c[23 + a];
            // fma(e, v, cv) has two scalar sources (one allowed per VALU op): keep cv in a VGPR for good
            // instead of re-materialising it every row
            double t = c[21 + a];
            asm volatile("v_mov_b64 %0, %1" : "=v"(cv[a]) : "s"(t));
        }
    }
    __device__ __forceinline__ void init(const double* a0) { M.init(a0); reset_acc(); }
    __device__ __forceinline__ void reset_acc() {
        M.reset_acc();
        acc2 = 0.0;
#pragma unroll
        for (int j = 0; j < NDIRP; j++) sacc[j] = 0.0;
#pragma unroll
        for (int a = 0; a < D; a++) macc[a] = 0.0;
    }
    // table phase: generic mean half with the row's gains
    __device__ __forceinline__ void step_table(const double* __restrict__ row, const double* mu, const double* y) {
        CtcrwGain G;
        G.iF = row[0]; G.k1 = row[1]; G.k2 = row[2]; G.bm = row[3];
#pragma unroll
        for (int j = 0; j < NDIRP; j++) { G.diF[j] = row[4 + j]; G.dk1[j] = row[7 + j]; G.dk2[j] = row[10 + j]; }
        ctcrw_mean_step<D, MASK>(M, tr, G, mu, y, G.iF != 0.0);
    }
    // stationary phase: same recursion (nllk_ctcrw.hpp:221, 238 and its derivatives) with
    // du = -tx folded in: tx' = (1-k1) tx + t12 tv + dk1 u [+ dt12 (v - B mu)], etc.
    __device__ __forceinline__ void step_stat(const double* y) {
#pragma unroll
        for (int a = 0; a < D; a++) {
            const double u = y[a] - M.x[a];
            acc2 = fma(u, u, acc2);
#pragma unroll
            for (int j = 0; j < NDIRP; j++) {
                if (!(MASK & dir_bit(j))) continue;
                const double tx = M.tx[j][a], tv = M.tv[j][a];
                sacc[j] = fma(u, tx, sacc[j]);
                double nx = dk1[j] * u, nv = dk2[j] * u;
                if (j == 1) {
                    const double w = M.v[a] - bmu[a];
                    nx = fma(dt12, w, nx);
                    nv = fma(de, w, nv);
                }
                M.tx[j][a] = fma(c1, tx, fma(t12, tv, nx));
                M.tv[j][a] = fma(e, tv, fma(-k2, tx, nv));
            }
            if (MASK & DIR_MU) {
                const double mx = M.mx[a], mv = M.mv[a];
                macc[a] = fma(u, mx, macc[a]);
                M.mx[a] = fma(c1, mx, fma(t12, mv, cb1));
                M.mv[a] = fma(e, mv, fma(-k2, mx, cb2));
            }
            const double x = M.x[a], v = M.v[a];
            M.x[a] = fma(k1, u, fma(t12, v, x)) + cx[a];
            M.v[a] = fma(k2, u, fma(e, v, cv[a]));
        }
    }
    __device__ __forceinline__ void finish(double* out) const {
        const double z[NDIRP] = {0.0, 0.0, 0.0};
        ctcrw_finish_parts<D, MASK>(0.0, z, M, out);
        out[0] += 0.5 * iF * acc2;
        if (MASK & DIR_SIG) out[1] += hd[0] * acc2 - iF * sacc[0];
        if (MASK & DIR_MU) {
#pragma unroll
            for (int a = 0; a < D; a++) out[2 + a] -= iF * macc[a];
        }
        if (MASK & DIR_P1) out[2 + D] += hd[1] * acc2 - iF * sacc[1];
        if (MASK & DIR_P2) out[3 + D] += hd[2] * acc2 - iF * sacc[2];
    }
    // compact hand-over layout of the shared-covariance kernels: state, wanted directions, mu
    __device__ __forceinline__ void dump(double* o) const {
        int k = 0;
#pragma unroll
        for (int a = 0; a < D; a++) { o[k++] = M.x[a]; o[k++] = M.v[a]; }
#pragma unroll
        for (int j = 0; j < NDIRP; j++) {
            if (!(MASK & dir_bit(j))) continue;
#pragma unroll
            for (int a = 0; a < D; a++) { o[k++] = M.tx[j][a]; o[k++] = M.tv[j][a]; }
        }
        if (MASK & DIR_MU) {
#pragma unroll
            for (int a = 0; a < D; a++) { o[k++] = M.mx[a]; o[k++] = M.mv[a]; }
        }
    }
    __device__ static __forceinline__ void warm_a0(const double* y, double* a0) {
#pragma unroll
        for (int a = 0; a < D; a++) { a0[2 * a] = (y[a] == y[a]) ? y[a] : 0.0; a0[2 * a + 1] = 0.0; }
    }
};

// ---- OU_SSM / BM_SSM ---------------------------------------------------------------------------
template <int MODEL, int D, int MASK>
struct SharedScal {
    static constexpr int SD = D;
    static constexpr bool HAS_P2 = (MODEL == M_OU_SSM);
    static constexpr int NSTATE = shared_nstate(D, MASK, HAS_P2);
    ScalMean<D, MASK> M;
    double acc2, sacc[NDIRP], macc[D];
    double k, c, t, b, iF, hd[NDIRP], dk[NDIRP], dt_, cmu[D], dbmu[D];
    ScalTrans tr;

    __device__ __forceinline__ void setup(const IsoArgs& A) {
        tr = A.str;
        const double* cc = A.statc;
        iF = cc[0]; k = cc[1]; c = cc[2]; t = cc[3]; b = cc[4]; dt_ = cc[5];
#pragma unroll
        for (int j = 0; j < NDIRP; j++) { hd[j] = cc[10 + j]; dk[j] = cc[13 + j]; }
#pragma unroll
        for (int a = 0; a < D; a++) { cmu[a] = cc[19 + a]; dbmu[a] = cc[21 + a]; }
    }
    __device__ __forceinline__ void init(const double* a0) { M.init(a0); reset_acc(); }
    __device__ __forceinline__ void reset_acc() {
        M.reset_acc();
        acc2 = 0.0;
#pragma unroll
        for (int j = 0; j < NDIRP; j++) sacc[j] = 0.0;
#pragma unroll
        for (int a = 0; a < D; a++) macc[a] = 0.0;
    }
    __device__ __forceinline__ void step_table(const double* __restrict__ row, const double* mu, const double* y) {
        ScalGain G;
        G.iF = row[0]; G.k = row[1]; G.c = row[2];
#pragma unroll
        for (int j = 0; j < NDIRP; j++) { G.diF[j] = row[4 + j]; G.dk[j] = row[7 + j]; }
        scal_mean_step<D, MASK, HAS_P2>(M, tr, G, mu, y, G.iF != 0.0);
    }
    // x' = t x + k u + b mu;  tx' = (t - k) tx + dk u [+ dt_ x + db mu]   (nllk_ou_ssm.hpp:204, nllk_bm_ssm.hpp:166)
    __device__ __forceinline__ void step_stat(const double* y) {
#pragma unroll
        for (int a = 0; a < D; a++) {
            const double x = M.x[a];
            const double u = y[a] - x;
            acc2 = fma(u, u, acc2);
#pragma unroll
            for (int j = 0; j < NDIRP; j++) {
                if (!(MASK & dir_bit(j)) || (j == 2 && !HAS_P2)) continue;
                const double tx = M.tx[j][a];
                sacc[j] = fma(u, tx, sacc[j]);
                double nx = dk[j] * u;
                if (j == 1) nx = fma(dt_, x, nx) + dbmu[a];
                M.tx[j][a] = fma(c, tx, nx);
            }
            if (MASK & DIR_MU) {
                const double mx = M.mx[a];
                macc[a] = fma(u, mx, macc[a]);
                M.mx[a] = fma(c, mx, b);
            }
            M.x[a] = fma(k, u, fma(t, x, cmu[a]));
        }
    }
    __device__ __forceinline__ void finish(double* out) const {
        const double z[NDIRP] = {0.0, 0.0, 0.0};
        scal_finish_parts<D, MASK>(0.0, z, M, out);
        out[0] += 0.5 * iF * acc2;
        if (MASK & DIR_SIG) out[1] += hd[0] * acc2 - iF * sacc[0];
        if (MASK & DIR_MU) {
#pragma unroll
            for (int a = 0; a < D; a++) out[2 + a] -= iF * macc[a];
        }
        if (MASK & DIR_P1) out[2 + D] += hd[1] * acc2 - iF * sacc[1];
        if (HAS_P2 && (MASK & DIR_P2)) out[3 + D] += hd[2] * acc2 - iF * sacc[2];
    }
    __device__ __forceinline__ void dump(double* o) const {
        int kk = 0;
#pragma unroll
        for (int a = 0; a < D; a++) o[kk++] = M.x[a];
#pragma unroll
        for (int j = 0; j < NDIRP; j++) {
            if (!(MASK & dir_bit(j)) || (j == 2 && !HAS_P2)) continue;
#pragma unroll
            for (int a = 0; a < D; a++) o[kk++] = M.tx[j][a];
        }
        if (MASK & DIR_MU) {
#pragma unroll
            for (int a = 0; a < D; a++) o[kk++] = M.mx[a];
        }
    }
    __device__ static __forceinline__ void warm_a0(const double* y, double* a0) {
#pragma unroll
        for (int a = 0; a < D; a++) a0[a] = (y[a] == y[a]) ? y[a] : 0.0;
    }
};


// ---- stationary-only CTCRW lanes in TRANSFER-FUNCTION form -------------------------------------------
// Past the covariance transient the filter is linear and time-invariant: with the closed-loop matrix
// L = T - K Z = [[c1, t12], [-k2, e]] (c1 = 1 - k1) the innovation is u = [A(q)/D(q)] y, where
//     A(q) = (1 - q^-1)(1 - e q^-1)           (open-loop poles: the integrator and e)
//     D(q) = 1 + d1 q^-1 + d2 q^-2,  d1 = -(c1 + e),  d2 = c1 e + k2 t12    (closed-loop poles),
// and d u / d theta_j = q^-1 (1 - q^-1) (pi0_j + pi1_j q^-1 + pi2_j q^-2) / D(q)^2 y for EVERY covariance
// direction j (the gains' sensitivities are constants there).  So per row and dimension
//     dy = y_t - y_{t-1} - mu dt      (the factor 1 - q^-1 taken on the data: increments are O(1), so are w, r;
//                                      a constant drift mu only shifts the increments)
//     w  = dy - d1 w_{t-1} - d2 w_{t-2}        u = w - e w_{t-1}        r = w - d1 r_{t-1} - d2 r_{t-2}
//     S += u^2,   C_k += u r_{t-k}  (k = 1, 2, 3)
// and at the end  d nll / d theta_j = hd_j S + iF (pi0_j C_1 + pi1_j C_2 + pi2_j C_3):
// 11 fp64 instructions per row and dimension whatever the number of directions (25 in the basis form
// this replaces, ~120 in the general filter), which moves the headline kernel from fp64 issue to HBM.
// The hand-over dump converts back to the direction form the other kernels use (x, v and their
// sensitivities are short linear combinations of y_{t-1}, w and r), so the window check is unchanged.
template <int D, int MASK>
struct TfCtcrw {
    static constexpr int SD = 2 * D;
    static constexpr int NSTATE = shared_nstate(2 * D, MASK, true);
    static constexpr bool ANYP = (MASK & (DIR_SIG | DIR_P1 | DIR_P2)) != 0;
    double yp[D], w1[D], w2[D], r1[D], r2[D], r3[D], su[D];
    double acc2, C1, C2, C3;
    double e, nd1, nd2, cm[D];
    const double* c;   // the argument block's constants (scalar loads, used outside the row loop only)

    __device__ __forceinline__ void setup(const IsoArgs& A) {
        c = A.statc;
        e = c[5]; nd1 = c[26]; nd2 = c[27];
#pragma unroll
        for (int a = 0; a < D; a++) cm[a] = c[29 + a];
    }
    // yprev = the observation of the row BEFORE the window's first row
    __device__ __forceinline__ void init(const double* yprev) {
#pragma unroll
        for (int a = 0; a < D; a++) { yp[a] = yprev[a]; w1[a] = w2[a] = r1[a] = r2[a] = r3[a] = 0.0; }
        reset_acc();
    }
    __device__ __forceinline__ void reset_acc() {
        acc2 = C1 = C2 = C3 = 0.0;
#pragma unroll
        for (int a = 0; a < D; a++) su[a] = 0.0;
    }
    __device__ __forceinline__ void step_table(const double*, const double*, const double*) {}  // never used
    __device__ __forceinline__ void step_stat(const double* y) {
#pragma unroll
        for (int a = 0; a < D; a++) {
            const double dy = (y[a] - yp[a]) - cm[a];
            yp[a] = y[a];
            const double w0 = fma(nd1, w1[a], fma(nd2, w2[a], dy));
            const double u = fma(-e, w1[a], w0);
            acc2 = fma(u, u, acc2);
            if (ANYP) {
                C1 = fma(u, r1[a], C1);
                C2 = fma(u, r2[a], C2);
                C3 = fma(u, r3[a], C3);
                const double r0 = fma(nd1, r1[a], fma(nd2, r2[a], w0));
                r3[a] = r2[a]; r2[a] = r1[a]; r1[a] = r0;
            }
            if (MASK & DIR_MU) su[a] += u;
            w2[a] = w1[a]; w1[a] = w0;
        }
    }
    __device__ __forceinline__ void finish(double* out) const {
        const double iF = c[0];
        out[0] = 0.5 * iF * acc2;
        const int slot[NDIRP] = {1, 2 + D, 3 + D};
#pragma unroll
        for (int k = 1; k < 4 + D; k++) out[k] = 0.0;
#pragma unroll
        for (int j = 0; j < NDIRP; j++)
            if (MASK & dir_bit(j)) out[slot[j]] = c[10 + j] * acc2 + iF * (c[31 + j] * C1 + c[34 + j] * C2 + c[37 + j] * C3);
        if (MASK & DIR_MU) {
#pragma unroll
            for (int a = 0; a < D; a++) out[2 + a] = -iF * c[46] * su[a];
        }
    }
    // hand-over states in DIRECTION form (what the transient kernel and k_iso.hip dump): the state on
    // arrival at the next row t, from y_{t-1}, w_{t-1}, w_{t-2}, r_{t-1..t-3}
    //   x = y_{t-1} + mu dt - c1 w_{t-1} + d2 w_{t-2}          v = k2 w_{t-1} + mu
    //   dx/dtheta_j = -(pi0_j r_{t-1} + pi1_j r_{t-2} + pi2_j r_{t-3})
    //   dv/dtheta_j = dk2_j w_{t-1} - k2 (alpha_j r_{t-2} + beta_j r_{t-3}),  alpha = d d1, beta = d d2
    __device__ __forceinline__ void dump(double* o) const {
        const double k2 = c[2], c1 = c[3], d2 = c[28];
        int k = 0;
#pragma unroll
        for (int a = 0; a < D; a++) {
            o[k++] = yp[a] + cm[a] - c1 * w1[a] + d2 * w2[a];
            o[k++] = k2 * w1[a] + c[23 + a];
        }
#pragma unroll
        for (int j = 0; j < NDIRP; j++) {
            if (!(MASK & dir_bit(j))) continue;
#pragma unroll
            for (int a = 0; a < D; a++) {
                o[k++] = -(c[31 + j] * r1[a] + c[34 + j] * r2[a] + c[37 + j] * r3[a]);
                o[k++] = c[16 + j] * w1[a] - k2 * (c[40 + j] * r2[a] + c[43 + j] * r3[a]);
            }
        }
        if (MASK & DIR_MU) {
#pragma unroll
            for (int a = 0; a < D; a++) { o[k++] = c[46]; o[k++] = c[47]; }
        }
    }
    __device__ static __forceinline__ void warm_a0(const double*, double*) {}
};

// OU_SSM / BM_SSM: tx_j = dk_j A1 + [j = par n_dim] A3 with A1 <- forcing u, A3 <- forcing dt_ x + db mu
template <int MODEL, int D, int MASK>
struct BasisScal {
    static constexpr int SD = D;
    static constexpr bool HAS_P2 = (MODEL == M_OU_SSM);
    static constexpr int NSTATE = shared_nstate(D, MASK, HAS_P2);
    static constexpr bool ANYP = (MASK & (DIR_SIG | DIR_P1 | (HAS_P2 ? DIR_P2 : 0))) != 0;
    static constexpr bool P1 = (MASK & DIR_P1) != 0;
    double x[D], A1[D], A3[D], mx[D];
    double acc2, S1, S3, macc[D];
    double k, c, t, b, iF, hd[NDIRP], dk[NDIRP], dt_, cmu[D], dbmu[D];

    __device__ __forceinline__ void setup(const IsoArgs& A) {
        const double* cc = A.statc;
        iF = cc[0]; k = cc[1]; c = cc[2]; t = cc[3]; b = cc[4]; dt_ = cc[5];
#pragma unroll
        for (int j = 0; j < NDIRP; j++) { hd[j] = cc[10 + j]; dk[j] = cc[13 + j]; }
#pragma unroll
        for (int a = 0; a < D; a++) { cmu[a] = cc[19 + a]; dbmu[a] = cc[21 + a]; }
    }
    __device__ __forceinline__ void init(const double* a0) {
#pragma unroll
        for (int a = 0; a < D; a++) { x[a] = a0[a]; A1[a] = A3[a] = mx[a] = 0.0; }
        reset_acc();
    }
    __device__ __forceinline__ void reset_acc() {
        acc2 = S1 = S3 = 0.0;
#pragma unroll
        for (int a = 0; a < D; a++) macc[a] = 0.0;
    }
    __device__ __forceinline__ void step_table(const double*, const double*, const double*) {}
    __device__ __forceinline__ void step_stat(const double* y) {
#pragma unroll
        for (int a = 0; a < D; a++) {
            const double xx = x[a];
            const double u = y[a] - xx;
            acc2 = fma(u, u, acc2);
            if (ANYP) {
                const double a1 = A1[a];
                S1 = fma(u, a1, S1);
                A1[a] = fma(c, a1, u);
                if (P1) {
                    const double a3 = A3[a];
                    S3 = fma(u, a3, S3);
                    A3[a] = fma(c, a3, fma(dt_, xx, dbmu[a]));
                }
            }
            if (MASK & DIR_MU) {
                const double m1 = mx[a];
                macc[a] = fma(u, m1, macc[a]);
                mx[a] = fma(c, m1, b);
            }
            x[a] = fma(k, u, fma(t, xx, cmu[a]));
        }
    }
    __device__ __forceinline__ void finish(double* out) const {
        out[0] = 0.5 * iF * acc2;
        const double s3[NDIRP] = {0.0, S3, 0.0};
        const int slot[NDIRP] = {1, 2 + D, 3 + D};
#pragma unroll
        for (int kk = 1; kk < 4 + D; kk++) out[kk] = 0.0;
#pragma unroll
        for (int j = 0; j < NDIRP; j++)
            if ((MASK & dir_bit(j)) && (j < 2 || HAS_P2)) out[slot[j]] = hd[j] * acc2 - iF * (dk[j] * S1 + s3[j]);
        if (MASK & DIR_MU) {
#pragma unroll
            for (int a = 0; a < D; a++) out[2 + a] = -iF * macc[a];
        }
    }
    __device__ __forceinline__ void dump(double* o) const {
        int kk = 0;
#pragma unroll
        for (int a = 0; a < D; a++) o[kk++] = x[a];
#pragma unroll
        for (int j = 0; j < NDIRP; j++) {
            if (!(MASK & dir_bit(j)) || (j == 2 && !HAS_P2)) continue;
#pragma unroll
            for (int a = 0; a < D; a++) o[kk++] = dk[j] * A1[a] + (j == 1 ? A3[a] : 0.0);
        }
        if (MASK & DIR_MU) {
#pragma unroll
            for (int a = 0; a < D; a++) o[kk++] = mx[a];
        }
    }
    __device__ static __forceinline__ void warm_a0(const double* y, double* a0) {
#pragma unroll
        for (int a = 0; a < D; a++) a0[a] = (y[a] == y[a]) ? y[a] : 0.0;
    }
};

// direction form for windows that touch the transient, basis form for the stationary-only kernel
template <int MODEL, int D, int MASK, bool STATONLY>
struct SharedSel { typedef SharedScal<MODEL, D, MASK> type; };
template <int D, int MASK>
struct SharedSel<M_CTCRW, D, MASK, false> { typedef SharedCtcrw<D, MASK> type; };
template <int MODEL, int D, int MASK>
struct SharedSel<MODEL, D, MASK, true> { typedef BasisScal<MODEL, D, MASK> type; };
template <int D, int MASK>
struct SharedSel<M_CTCRW, D, MASK, true> { typedef TfCtcrw<D, MASK> type; };

// The gain table in the table phase.  Its rows used to be fetched row by row with wave-uniform vector loads (13 per row,
// an L2 round trip each, competing with the observation prefetch for the 63 outstanding-load slots): 600 cycles per row
// for a ~70-instruction step, and the transient window (~100 such rows, one wave per track group) was a fixed ~30 us in
// EVERY evaluation of this path -- most of C2's 50 us kernel.  Now every wave stages GAIN_SLAB_ROWS rows at a time into
// its own LDS slab with coalesced loads (one round trip per slab) and the steps read them as LDS broadcasts.
constexpr int GAIN_SLAB_ROWS = 64;      // multiple of 2 * SHARED_U
static_assert(GAIN_SLAB_ROWS % (2 * SHARED_U) == 0, "gain slab");

__device__ __forceinline__ void stage_gain(const IsoArgs& A, double* slab, int row0) {
    const int lane = threadIdx.x & 63, glast = A.gain_last;
    const double* __restrict__ gain = A.gain;
#pragma unroll 4
    for (int r = 0; r < GAIN_SLAB_ROWS; r += WAVE / GAIN_ROW) {          // 4 rows of 16 doubles per wave load
        const int rr = r + lane / GAIN_ROW;
        slab[rr * GAIN_ROW + (lane % GAIN_ROW)] = gain[(int64_t)min(row0 + rr, glast) * GAIN_ROW + (lane % GAIN_ROW)];
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// rows [s0, s0 + SHARED_U) from a register block; `grow` = the slab row of s0 (table phase)
template <bool STAT, int D, class Lane>
__device__ __forceinline__ void run_block(Lane& S, const IsoArgs& A, const double (&blk)[SHARED_U][D], int s0, int ns,
                                          int ns_min, const double* mu, const double* grow) {
    if (s0 + SHARED_U <= ns_min) {
        // every lane's track covers the whole block: no per-row predication
#pragma unroll
        for (int u = 0; u < SHARED_U; u++) {
            if (STAT) S.step_stat(blk[u]);
            else S.step_table(grow + u * GAIN_ROW, mu, blk[u]);
        }
    } else {
#pragma unroll
        for (int u = 0; u < SHARED_U; u++) {
            if (s0 + u < ns) {
                if (STAT) S.step_stat(blk[u]);
                else S.step_table(grow + u * GAIN_ROW, mu, blk[u]);
            }
        }
    }
}

// rows [sa, sb) of the lane's window; STAT = stationary gains.  sa is a multiple of SHARED_U.
// Two register blocks in ping-pong: while one is consumed the other is in flight (no copies).
template <bool STAT, int D, class Lane>
__device__ __forceinline__ void run_segment(Lane& S, const IsoArgs& A, const double* base, int sa, int sb, int ns,
                                            int ns_min, const double* mu) {
    const int C = A.tv.C, c_obs = A.tv.c_obs;
    if (sa >= sb) return;
    double bufA[SHARED_U][D], bufB[SHARED_U][D];
#ifdef SSDE_DIAG_NOSTREAM
#define SSDE_ROWPTR(s) (base)
#else
#define SSDE_ROWPTR(s) (base + (int64_t)(s) * C * WAVE)
#endif
    __shared__ double gain_slab[STAT ? 1 : WG_WAVES][STAT ? 1 : GAIN_SLAB_ROWS * GAIN_ROW];
    double* slab = gain_slab[STAT ? 0 : (threadIdx.x >> 6)];
#ifndef SSDE_D1_DEEP
#define SSDE_D1_DEEP 1
#endif
#ifndef SSDE_DEEP_MAXD
#define SSDE_DEEP_MAXD 1        // two columns: measured, no gain (0.280-0.285 against 0.256-0.274 ms in one session)
#endif
    if constexpr (D <= SSDE_DEEP_MAXD && STAT && SSDE_D1_DEEP) {
        // One response column is 8 B per lane and row: a block in flight is half the bytes of the two-column case, and the
        // stream sat at 4.7 TB/s against 5.7-6.2.  Three register blocks in rotation keep TWO blocks (16 rows) in flight.
        static_assert(5 * SHARED_U <= TILE_SPARE, "look-ahead of the three-block rotation");
        double bufC[SHARED_U][D];
        load_obs_block<D>(bufA, SSDE_ROWPTR(sa), C, c_obs);
        load_obs_block<D>(bufB, SSDE_ROWPTR(sa + SHARED_U), C, c_obs);
        for (int s0 = sa; s0 < sb; s0 += 3 * SHARED_U) {       // TILE_SPARE (64 rows) covers the 5 blocks of look-ahead
            load_obs_block<D>(bufC, SSDE_ROWPTR(s0 + 2 * SHARED_U), C, c_obs);
            run_block<STAT, D>(S, A, bufA, s0, ns, ns_min, mu, slab);
            load_obs_block<D>(bufA, SSDE_ROWPTR(s0 + 3 * SHARED_U), C, c_obs);
            if (s0 + SHARED_U < sb) run_block<STAT, D>(S, A, bufB, s0 + SHARED_U, ns, ns_min, mu, slab);
            load_obs_block<D>(bufB, SSDE_ROWPTR(s0 + 4 * SHARED_U), C, c_obs);
            if (s0 + 2 * SHARED_U < sb) run_block<STAT, D>(S, A, bufC, s0 + 2 * SHARED_U, ns, ns_min, mu, slab);
        }
        return;
    }
    load_obs_block<D>(bufA, SSDE_ROWPTR(sa), C, c_obs);
    for (int s0 = sa; s0 < sb; s0 += 2 * SHARED_U) {
        // TILE_SPARE (>= 3 blocks) keeps the look-ahead loads inside the allocation
        load_obs_block<D>(bufB, SSDE_ROWPTR(s0 + SHARED_U), C, c_obs);
        const int srow = (s0 - sa) % GAIN_SLAB_ROWS;
        if (!STAT && srow == 0) stage_gain(A, slab, s0);
        run_block<STAT, D>(S, A, bufA, s0, ns, ns_min, mu, slab + srow * GAIN_ROW);
        load_obs_block<D>(bufA, SSDE_ROWPTR(s0 + 2 * SHARED_U), C, c_obs);
        if (s0 + SHARED_U < sb) run_block<STAT, D>(S, A, bufB, s0 + SHARED_U, ns, ns_min, mu, slab + (srow + SHARED_U) * GAIN_ROW);
    }
}

// STATONLY: the whole window (warm-up included) lies past the covariance transient -- the lean
// kernel; otherwise the window touches the transient and also carries the table-phase code.
template <int MODEL, int D, int MASK, bool STATONLY>
__device__ __forceinline__ void run_lane_shared(const IsoArgs& A, int g, int part, int chunk) {
    typedef typename SharedSel<MODEL, D, MASK, STATONLY>::type Lane;
    constexpr int NACC = 4 + D;
    constexpr int SD = Lane::SD;
    const int lane = threadIdx.x & 63;
    const TileView& tv = A.tv;
    const int C = tv.C, c_obs = tv.c_obs;
    const double* base = tv.tiles + tv.group_off[g] + lane;
    const int L = tv.group_len[g];
    const int ns = tv.lane_nsteps[g * WAVE + lane];
    int ns_min = ns;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) ns_min = min(ns_min, __shfl_xor(ns_min, o, 64));
    ns_min = __builtin_amdgcn_readfirstlane(ns_min);
    const int pc = part * A.n_chunks + chunk;

    int s_begin, s_acc, s_end;
    window_bounds(L, A.n_chunks, A.window, A.t0, chunk, s_begin, s_acc, s_end, A.t0_delta);
    // first row from which the stationary gains apply (gain_stat[0] == 0 would mean "never scored":
    // that degenerate case stays on the table path)
    int s_stat = (A.gain_last + SHARED_U - 1) / SHARED_U * SHARED_U;
    if (A.gain_stat[0] == 0.0) s_stat = INT32_MAX;

    Lane S;
    S.setup(A);
    double mu[D];
#pragma unroll
    for (int a = 0; a < D; a++) mu[a] = A.mu[a];
    {
        double a0[SD];
        if (STATONLY && MODEL == M_CTCRW) {
            // transfer-function lanes start from the observation of the row before the window (s_begin > 0:
            // stationary-only windows lie past the covariance transient)
#pragma unroll
            for (int a = 0; a < D; a++) a0[a] = base[((int64_t)(s_begin - 1) * C + c_obs + a) * WAVE];
        } else if (s_begin == 0) {
#pragma unroll
            for (int c = 0; c < SD; c++) a0[c] = tv.a0[((int64_t)g * SD + c) * WAVE + lane];
        } else {
            double y0[D];
#pragma unroll
            for (int a = 0; a < D; a++) y0[a] = base[((int64_t)s_begin * C + c_obs + a) * WAVE];
            Lane::warm_a0(y0, a0);
        }
        S.init(a0);
    }
    // warm-up rows [s_begin, s_acc), then scored rows [s_acc, s_end); each split at s_stat
    {
        const int m = STATONLY ? s_begin : min(max(s_stat, s_begin), s_acc);
        if (!STATONLY) run_segment<false, D>(S, A, base, s_begin, m, ns, ns_min, mu);
        run_segment<true, D>(S, A, base, m, s_acc, ns, ns_min, mu);
    }
    if (s_acc > s_begin) {
        double st[Lane::NSTATE];
        S.dump(st);
        double* o = A.bnd + (((int64_t)pc * tv.n_groups + g) * 2 + 0) * NSTATE_MAX * WAVE + lane;
#pragma unroll
        for (int k = 0; k < Lane::NSTATE; k++) o[k * WAVE] = st[k];
        S.reset_acc();
    }
    {
        const int m = STATONLY ? s_acc : min(max(s_stat, s_acc), s_end);
        if (!STATONLY) run_segment<false, D>(S, A, base, s_acc, m, ns, ns_min, mu);
        run_segment<true, D>(S, A, base, m, s_end, ns, ns_min, mu);
    }
    if (A.n_chunks > 1 && chunk + 1 < A.n_chunks) {
        double st[Lane::NSTATE];
        S.dump(st);
        double* o = A.bnd + (((int64_t)pc * tv.n_groups + g) * 2 + 1) * NSTATE_MAX * WAVE + lane;
#pragma unroll
        for (int k = 0; k < Lane::NSTATE; k++) o[k * WAVE] = st[k];
    }
    double out[NACC];
    S.finish(out);
    if (s_acc >= s_end) {
#pragma unroll
        for (int k = 0; k < NACC; k++) out[k] = 0.0;
    }
#pragma unroll
    for (int k = 0; k < NACC; k++) {
        const double t = wave_sum(out[k]);
        if (lane == 0) A.partials[((int64_t)pc * NACC + k) * tv.n_groups + g] = t;
    }
}

// One kernel per (model, dimension, direction mask): the register allocation of a kernel is the worst
// case over everything it contains, so the masks are NOT folded into one kernel with a switch here.
// With a dedicated transient window (t0 > 0) the grid enumerates windows 1..n_chunks-1; the wave that
// owns window 1 first runs window 0 (direction form, gain table), every window >= 1 runs the lean
// basis-form code.  Without it (short tracks, a single window) every wave runs the general shared code.
template <int MODEL, int D, int MASK>
__global__ __launch_bounds__(WG_WAVES * WAVE, 1) void iso_shared_kernel(const IsoArgs A) {
    if (blockIdx.x == 0 && threadIdx.x == 0 && A.chk_out) *A.chk_out = 0.0;   // raised by the finalize launch
    int g, part, chunk;
    if (A.t0 > 0) {
        if (!decode_block(A, A.n_chunks - 1, g, part, chunk)) return;
        if (!group_selected(A, g)) return;
        chunk += 1;
        if (chunk == 1) run_lane_shared<MODEL, D, MASK, false>(A, g, part, 0);
        run_lane_shared<MODEL, D, MASK, true>(A, g, part, chunk);
    } else {
        if (!decode_block(A, A.n_chunks, g, part, chunk)) return;
        if (!group_selected(A, g)) return;
        run_lane_shared<MODEL, D, MASK, false>(A, g, part, chunk);
    }
}

// host side: the stationary constants (layout in ssde_device.hpp)
void fill_stat_consts(int model, int d, IsoArgs& a) {
    double* c = a.statc;
    for (int i = 0; i < 48; i++) c[i] = 0.0;
    const double* r = a.gain_stat;
    if (model == M_CTCRW) {
        const CtcrwTrans& tr = a.ctr;
        const double bm = r[3];
        c[0] = r[0]; c[1] = r[1]; c[2] = r[2]; c[3] = 1.0 - r[1]; c[4] = tr.t12; c[5] = tr.e; c[6] = tr.dt12; c[7] = tr.de;
        c[8] = bm * tr.b1; c[9] = bm * tr.b2;
        for (int j = 0; j < NDIRP; j++) { c[10 + j] = 0.5 * r[4 + j]; c[13 + j] = r[7 + j]; c[16 + j] = r[10 + j]; }
        for (int k = 0; k < d; k++) { c[19 + k] = c[8] * a.mu[k]; c[21 + k] = c[9] * a.mu[k]; c[23 + k] = bm * a.mu[k]; }
        // transfer-function form (TfCtcrw)
        const double k1 = r[1], k2 = r[2], c1 = 1.0 - k1, e = tr.e, t12 = tr.t12;
        const double d1 = -(c1 + e), d2 = c1 * e + k2 * t12;
        c[26] = -d1; c[27] = -d2; c[28] = d2;
        const double dt = tr.b1 + tr.t12;                     // b1 = dt - t12
        for (int k = 0; k < d; k++) c[29 + k] = bm * a.mu[k] * dt;
        for (int j = 0; j < NDIRP; j++) {
            const double de = (j == 1) ? tr.de : 0.0, dt12 = (j == 1) ? tr.dt12 : 0.0;
            const double dk1 = r[7 + j], dk2 = r[10 + j];
            const double alpha = dk1 - de;                                        // d d1 / d theta_j
            const double beta = -dk1 * e + c1 * de + dk2 * t12 + k2 * dt12;       // d d2 / d theta_j
            c[31 + j] = -dk1;
            c[34 + j] = -de * d1 - beta + e * alpha;
            c[37 + j] = -de * d2 + e * beta;
            c[40 + j] = alpha; c[43 + j] = beta;
        }
        const double D1 = 1.0 + d1 + d2;                      // D(1): steady-state response to a constant input
        c[46] = dt * (1.0 - e) / D1;                          // d x / d mu_a
        c[47] = 1.0 - k2 * dt / D1;                           // d v / d mu_a
    } else {
        const ScalTrans& tr = a.str;
        c[0] = r[0]; c[1] = r[1]; c[2] = r[2]; c[3] = tr.t; c[4] = tr.b; c[5] = tr.dt_;
        for (int j = 0; j < NDIRP; j++) { c[10 + j] = 0.5 * r[4 + j]; c[13 + j] = r[7 + j]; }
        for (int k = 0; k < d; k++) { c[19 + k] = tr.b * a.mu[k]; c[21 + k] = tr.db * a.mu[k]; }
    }
}

template <int MODEL, int D>
static hipError_t launch_masks(const IsoArgs& a, dim3 grid, hipStream_t s, hipEvent_t ev0, hipEvent_t ev1) {
    dim3 block(WG_WAVES * WAVE);
    switch (a.part_mask[0]) {
    // hipExtLaunchKernelGGL stamps ev0 / ev1 with the kernel's own begin / end (what rocprof reports), not with the
    // stream position of separately recorded events
    // (ev0 == NULL: a plain launch -- SSDE_OPT_KERNEL_STAMPS off -- which costs the host and the queue a few microseconds less)
#define SSDE_CASE(M) case M: if (ev0) hipExtLaunchKernelGGL((iso_shared_kernel<MODEL, D, M>), grid, block, 0, s, ev0, ev1, 0, a); \
                             else hipLaunchKernelGGL((iso_shared_kernel<MODEL, D, M>), grid, block, 0, s, a); break;
        SSDE_CASE(0) SSDE_CASE(1) SSDE_CASE(2) SSDE_CASE(3) SSDE_CASE(4) SSDE_CASE(5) SSDE_CASE(6) SSDE_CASE(7)
        SSDE_CASE(8) SSDE_CASE(9) SSDE_CASE(10) SSDE_CASE(11) SSDE_CASE(12) SSDE_CASE(13) SSDE_CASE(14) SSDE_CASE(15)
#undef SSDE_CASE
        default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

// the shared path runs all directions in one part (n_parts == 1, mask = part_mask[0])
hipError_t launch_iso_shared(int model, int d, const IsoArgs& a, hipStream_t s, hipEvent_t ev0, hipEvent_t ev1) {
    if (a.n_parts != 1) return hipErrorInvalidValue;
    const int g8 = (a.tv.n_groups + 7) / 8;
    const int n_grid_chunks = a.t0 > 0 ? a.n_chunks - 1 : a.n_chunks;
    dim3 grid((g8 * 8 * n_grid_chunks + WG_WAVES - 1) / WG_WAVES);
    if (grid.x == 0) return hipSuccess;
    if (model == M_CTCRW && d == 1) return launch_masks<M_CTCRW, 1>(a, grid, s, ev0, ev1);
    if (model == M_CTCRW && d == 2) return launch_masks<M_CTCRW, 2>(a, grid, s, ev0, ev1);
    if (model == M_OU_SSM && d == 1) return launch_masks<M_OU_SSM, 1>(a, grid, s, ev0, ev1);
    if (model == M_OU_SSM && d == 2) return launch_masks<M_OU_SSM, 2>(a, grid, s, ev0, ev1);
    if (model == M_BM_SSM && d == 1) return launch_masks<M_BM_SSM, 1>(a, grid, s, ev0, ev1);
    if (model == M_BM_SSM && d == 2) return launch_masks<M_BM_SSM, 2>(a, grid, s, ev0, ev1);
    return hipErrorInvalidValue;
}

}  // namespace ssde
