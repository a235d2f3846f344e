// ssde_tf.hpp -- the STATIONARY-ONLY lanes of the register Kalman kernels: the filter past its covariance transient, with the
// gains and their sensitivities as constants (IsoArgs.statc), in the forms with the fewest fp64 instructions per row.  Used by
// the shared-covariance kernels for the windows past the transient (k_iso_shared.inc) and by the general kernel for its quiet
// rows (k_iso.hip).
#pragma once
#include "ssde_device.hpp"

namespace ssde {

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

}  // namespace ssde
