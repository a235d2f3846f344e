// k_direct.hip -- direct transition-density families "BM" and "OU" (nllk_sde.hpp:77-84 with
// tr_dens.hpp:32-37, 45-52) for gfx950.
//
// No recursion: row i depends only on rows i-1, i and on the parameters of row i-1 (Q6), so
// the long format is streamed as it is -- lane = row, every column read is coalesced
// (the design blocks are column-major n x K exactly as R hands them over).  The linear
// predictor, the density, its derivative w.r.t. the row's SDE parameters and the
// X' g accumulation are fused in one pass: the design row is read once (88 B/row in the C3
// configuration: obs 8 + time 8 + 9 columns 72).  HBM-bound; fp64; no MFMA, no LDS staging
// (each byte is used once by one lane).
#include "ssde_device.hpp"

namespace ssde {

template <int MODEL, int D, int KMAX>
__global__ __launch_bounds__(256) void direct_kernel(const DirectArgs A) {
    const SlotTable* __restrict__ T = A.slots;
    const int ns = A.n_slots;
    const int q = (MODEL == M_BM) ? D + 1 : D + 2;
    double acc[KMAX];
#pragma unroll
    for (int k = 0; k < KMAX; k++) acc[k] = 0.0;
    double nll = 0.0;

    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < A.n; i += stride) {
        if (!((A.scored[i >> 5] >> (i & 31)) & 1u)) continue;
        const double dt = A.times[i] - A.times[i - 1];  // dtimes(i-1), nllk_sde.hpp:37,80
        double w[KMAX];
        double par[MAX_Q] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int k = 0; k < KMAX; k++) {
            if (k < ns) {
                const int c = T->col[k];
                w[k] = (c >= 0) ? A.cols[c][i - 1] : 1.0;
                const double t = w[k] * A.par[T->pidx[k]];
                const int j = T->par_j[k];
                par[0] += (j == 0) ? t : 0.0;
                par[1] += (j == 1) ? t : 0.0;
                par[2] += (j == 2) ? t : 0.0;
                par[3] += (j == 3) ? t : 0.0;
            }
        }
        double g[MAX_Q] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int a = 0; a < D; a++) {
            const double z0 = A.obs[(i - 1) + (int64_t)a * A.n], z1 = A.obs[i + (int64_t)a * A.n];
            if (is_na(z0, A.any_nan) || is_na(z1, A.any_nan)) continue;  // tr_dens.hpp:31
            if (MODEL == M_BM) nll += bm_direct(z0, z1, dt, par[a], par[D], g[a], g[D]);
            else nll += ou_direct(z0, z1, dt, par[a], par[D], par[D + 1], g[a], g[D], g[D + 1]);
        }
        (void)q;
#pragma unroll
        for (int k = 0; k < KMAX; k++) {
            if (k < ns) {
                const int j = T->par_j[k];
                const double gj = (j == 0) ? g[0] : (j == 1) ? g[1] : (j == 2) ? g[2] : g[3];
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
