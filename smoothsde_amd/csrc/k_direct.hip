// k_direct.hip -- direct transition-density families "BM" and "OU" (nllk_sde.hpp:77-84 with
// tr_dens.hpp:32-37, 45-52) for gfx950.
//
// No recursion: row i depends only on rows i-1, i and on the parameters of row i-1 (Q6), so
// the long format is streamed as it is -- lane = row, every column read is coalesced
// (the design blocks are column-major n x K exactly as R hands them over).  The linear
// predictor, the density, its derivative w.r.t. the row's SDE parameters and the
// X' g accumulation are fused in one pass: the design row is read once (88 B/row in the C3
// configuration: obs 8 + time 8 + 9 columns 72).  HBM-bound by design; fp64; no MFMA, no LDS
// staging (each byte is used once by one lane).
//
// Everything that does not depend on the row is hoisted out of the loop: column pointers,
// coefficients and slot->parameter routing live in scalar registers, intercept-only parameters
// are folded into per-parameter constants, and exp() of a log-scale parameter is evaluated once
// when that parameter has no streamed column.
#include "ssde_device.hpp"

namespace ssde {

template <int MODEL, int D, int KMAX>
__global__ __launch_bounds__(256) void direct_kernel(const DirectArgs A) {
    const SlotTable* __restrict__ T = A.slots;
    const int ns = A.n_slots;
    constexpr int Q = (MODEL == M_BM) ? D + 1 : D + 2;

    // ---- loop invariants (wave-uniform) ---------------------------------------------------------
    const double* cptr[KMAX];
    double coef[KMAX];
    int pj[KMAX];
    double base[MAX_Q] = {0.0, 0.0, 0.0, 0.0};   // intercept part of every parameter
    bool varies[MAX_Q] = {false, false, false, false};
#pragma unroll
    for (int k = 0; k < KMAX; k++) {
        cptr[k] = nullptr; coef[k] = 0.0; pj[k] = -1;
        if (k < ns) {
            const int c = T->col[k];
            pj[k] = T->par_j[k];
            coef[k] = A.par[T->pidx[k]];
            if (c >= 0) {
                cptr[k] = A.cols[c];
#pragma unroll
                for (int j = 0; j < Q; j++) varies[j] = varies[j] || (pj[k] == j);
            } else {
#pragma unroll
                for (int j = 0; j < Q; j++) base[j] += (pj[k] == j) ? coef[k] : 0.0;
            }
        }
    }
    // natural-scale value of a log-scale parameter that has no streamed column
    const double nat_p1 = exp(base[D]);
    const double nat_p2 = (Q > D + 1) ? exp(base[D + 1]) : 0.0;

    double acc[KMAX];
#pragma unroll
    for (int k = 0; k < KMAX; k++) acc[k] = 0.0;
    double nll = 0.0;

    // every workgroup streams ONE contiguous range of rows (consecutive 2-KB pieces of each column)
    const int64_t per_block = ((A.n + gridDim.x - 1) / gridDim.x + 255) / 256 * 256;
    const int64_t row_lo = (int64_t)blockIdx.x * per_block;
    const int64_t row_hi = row_lo + per_block < A.n ? row_lo + per_block : A.n;
    for (int64_t i = row_lo + threadIdx.x; i < row_hi; i += 256) {
        if (!((A.scored[i >> 5] >> (i & 31)) & 1u)) continue;
        const double dt = A.times[i] - A.times[i - 1];  // dtimes(i-1), nllk_sde.hpp:37,80
        double w[KMAX];
        double par[MAX_Q] = {base[0], base[1], base[2], base[3]};
#pragma unroll
        for (int k = 0; k < KMAX; k++) {
            w[k] = 1.0;
            if (cptr[k] != nullptr) {                    // uniform: a scalar branch, not a per-lane select
                w[k] = cptr[k][i - 1];
                const double t = w[k] * coef[k];
                if (pj[k] == 0) par[0] += t;
                else if (pj[k] == 1) par[1] += t;
                else if (pj[k] == 2) par[2] += t;
                else par[3] += t;
            }
        }
        double g[MAX_Q] = {0.0, 0.0, 0.0, 0.0};
        if (MODEL == M_BM) {
            // tr_dens.hpp:35-37: mean = z0 + mu dt, sd = exp(par[D]) sqrt(dt)
            const double sig = varies[D] ? exp(par[D]) : nat_p1;
            const double sd = sig * sqrt(dt);
            const double lsd = log(sd), isd = 1.0 / sd;
#pragma unroll
            for (int a = 0; a < D; a++) {
                const double z0 = A.obs[(i - 1) + (int64_t)a * A.n], z1 = A.obs[i + (int64_t)a * A.n];
                if (is_na(z0, A.any_nan) || is_na(z1, A.any_nan)) continue;  // tr_dens.hpp:31
                const double r = (z1 - (z0 + par[a] * dt)) * isd;
                g[a] += -r * dt * isd;
                g[D] += 1.0 - r * r;
                nll += SSDE_LOG_SQRT_2PI + lsd + 0.5 * r * r;
            }
        } else {
            // tr_dens.hpp:49-52: mean = mu + e^{-dt/tau} (z0 - mu), sd = sqrt(kappa (1 - e^{-2 dt/tau}))
            const double tau = varies[D] ? exp(par[D]) : nat_p1;
            const double kap = varies[D + 1] ? exp(par[D + 1]) : nat_p2;
            const double z = dt / tau;
            const double e = exp(-z);
            const double e2 = e * e;
            const double ome2 = 1.0 - e2;
            const double sd = sqrt(kap * ome2);
            const double lsd = log(sd), isd = 1.0 / sd;
            const double dls_lt = -e2 * z / ome2;
#pragma unroll
            for (int a = 0; a < D; a++) {
                const double z0 = A.obs[(i - 1) + (int64_t)a * A.n], z1 = A.obs[i + (int64_t)a * A.n];
                if (is_na(z0, A.any_nan) || is_na(z1, A.any_nan)) continue;
                const double mu = par[a];
                const double r = (z1 - (mu + e * (z0 - mu))) * isd;
                g[a] += -r * (1.0 - e) * isd;
                g[D] += -r * (e * z * (z0 - mu)) * isd + (1.0 - r * r) * dls_lt;
                g[D + 1] += 0.5 * (1.0 - r * r);
                nll += SSDE_LOG_SQRT_2PI + lsd + 0.5 * r * r;
            }
        }
#pragma unroll
        for (int k = 0; k < KMAX; k++) {
            if (pj[k] >= 0) {                            // uniform
                const double gj = (pj[k] == 0) ? g[0] : (pj[k] == 1) ? g[1] : (pj[k] == 2) ? g[2] : g[3];
                acc[k] += w[k] * gj;
            }
        }
    }

    // workgroup reduction: wave shuffles, then 4 wave totals through LDS
    __shared__ double sh[4][KMAX + 1];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    double t = wave_sum(nll);
    if (lane == 0) sh[wv][0] = t;
#pragma unroll
    for (int k = 0; k < KMAX; k++) {
        if (k < ns) {
            t = wave_sum(acc[k]);
            if (lane == 0) sh[wv][1 + k] = t;
        }
    }
    __syncthreads();
    if (threadIdx.x <= ns) {
        const int k = threadIdx.x;
        A.partials[(int64_t)k * A.n_blocks + blockIdx.x] = (sh[0][k] + sh[1][k]) + (sh[2][k] + sh[3][k]);
    }
}

hipError_t launch_direct(const DirectArgs& a, hipStream_t s) {
    dim3 grid(a.n_blocks), block(256);
#define SSDE_L(MODEL, D, K)                                                            \
    if (a.model == MODEL && a.d == D && a.n_slots <= K) {                              \
        hipLaunchKernelGGL((direct_kernel<MODEL, D, K>), grid, block, 0, s, a);        \
        return hipGetLastError();                                                      \
    }
#define SSDE_LK(MODEL, D) SSDE_L(MODEL, D, 4) SSDE_L(MODEL, D, 16) SSDE_L(MODEL, D, 32) SSDE_L(MODEL, D, 64)
    SSDE_LK(M_BM, 1) SSDE_LK(M_BM, 2) SSDE_LK(M_OU, 1) SSDE_LK(M_OU, 2)
#undef SSDE_LK
#undef SSDE_L
    return hipErrorInvalidValue;
}

}  // namespace ssde
