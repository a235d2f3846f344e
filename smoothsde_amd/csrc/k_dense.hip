// k_dense.hip -- general Kalman kernel for gfx950: per-row H_array, arbitrary P0, and SDE
// parameters that vary row by row through streamed design columns (ssde_dense.hpp).
//
// Lane = track, workgroup = one wave, as in k_iso.hip; blockIdx.y selects a block of
// DENSE_NT gradient directions, every direction block recomputes the primal filter and
// carries its tangents as dual numbers in registers.  The design columns of a row are read
// from the same time-major tile as the observations (coalesced 512-B wave loads).  This is
// the coverage path (it is what makes every model configuration of the reference run on
// the GPU); the throughput path for the headline configurations is k_iso.hip.
#include "ssde_dense.hpp"
#include "ssde_device.hpp"

namespace ssde {

template <int MODEL, int D, int N, bool REPORT>
__global__ __launch_bounds__(WAVE) void dense_kernel(const DenseArgs A) {
    typedef DenseDims<MODEL, D> DM;
    constexpr int SD = DM::SD, Q = DM::Q;
    const int g = blockIdx.x, b = blockIdx.y, lane = threadIdx.x;
    const TileView& tv = A.tv;
    const SlotTable* __restrict__ T = A.slots;
    const int C = tv.C;
    const int cy = tv.c_obs;                                   // first obs channel (0: no dt channel, regular grid)
    const int c0 = cy + D + (A.has_h ? D * D : 0);
    const double* base = tv.tiles + tv.group_off[g] + lane;
    const int L = tv.group_len[g];
    const int ns = tv.lane_nsteps[g * WAVE + lane];
    const int nslots = A.n_slots;

    int dkind[N > 0 ? N : 1], dslot[N > 0 ? N : 1];
#pragma unroll
    for (int s = 0; s < N; s++) {
        const DenseDir dd = A.dirs[b * N + s];
        dkind[s] = dd.kind;
        dslot[s] = dd.slot;
    }

    DenseLane<MODEL, D, N> S;
    double a0[SD];
#pragma unroll
    for (int c = 0; c < SD; c++) a0[c] = tv.a0[((int64_t)g * SD + c) * WAVE + lane];
    S.init(a0, A.p0);
    const int64_t row0 = REPORT ? A.lane_row0[g * WAVE + lane] : 0;
    if (REPORT && ns > 0) {
#pragma unroll
        for (int c = 0; c < SD; c++) A.report[row0 + (int64_t)c * A.n] = a0[c];
    }

    // H = sigma_obs^2 I (makeH_*), d/d log_sigma_obs = 2 sigma_obs^2
    const double sig = exp(A.par[0]);
    DualN<N> hiso(sig * sig);
#pragma unroll
    for (int s = 0; s < N; s++) hiso.d[s] = (dkind[s] == 1) ? 2.0 * sig * sig : 0.0;

    for (int s0 = 0; s0 < L; s0++) {
        if (s0 >= ns) continue;
        const double* o = base + (int64_t)s0 * C * WAVE;
        // dtimes(i): the tile's dt channel, or the one interval of a globally regular grid (dtimes(n-1) = 1,
        // nllk_ctcrw.hpp:126-129: only REPORT(aest_all) ever shows the state propagated with it)
        double dt = cy ? o[0] : tv.dt_all;
        if (REPORT && !cy && row0 + 1 + s0 == A.n - 1) dt = A.last_dt;
        double y[D];
#pragma unroll
        for (int a = 0; a < D; a++) y[a] = o[(cy + a) * WAVE];
        DualN<N> H[D][D];
#pragma unroll
        for (int i = 0; i < D; i++)
#pragma unroll
            for (int j = 0; j < D; j++) {
                if (A.has_h) H[i][j] = DualN<N>(o[(cy + D + i + j * D) * WAVE]);   // H_array[,,i] column-major
                else H[i][j] = (i == j) ? hiso : DualN<N>(0.0);
            }
        // linear predictors of the row and their tangents
        DualN<N> par[Q];
#pragma unroll
        for (int j = 0; j < Q; j++) par[j] = DualN<N>(0.0);
        for (int k = 0; k < nslots; k++) {
            const int col = T->col[k];
            const double w = (col >= 0) ? (A.pp.nb ? ppd_value_hbm(A.pp, col, o + c0 * WAVE) : o[(c0 + col) * WAVE]) : 1.0;
            const double t = w * A.par[T->pidx[k]];
            const int j = T->par_j[k];
#pragma unroll
            for (int jj = 0; jj < Q; jj++) {
                par[jj].v += (j == jj) ? t : 0.0;
#pragma unroll
                for (int s = 0; s < N; s++) par[jj].d[s] += (j == jj && dkind[s] == 2 && dslot[s] == k) ? w : 0.0;
            }
        }
        dense_step<MODEL, D, N>(S, par, H, dt, y, is_na(y[0], A.any_nan));
        if (REPORT) {
#pragma unroll
            for (int c = 0; c < SD; c++) A.report[row0 + 1 + s0 + (int64_t)c * A.n] = S.a[c].v;
        }
    }
    if (REPORT) return;
    double t = wave_sum(S.nll.v);
    if (lane == 0) A.partials[((int64_t)b * (1 + N) + 0) * tv.n_groups + g] = t;
#pragma unroll
    for (int s = 0; s < N; s++) {
        t = wave_sum(S.nll.d[s]);
        if (lane == 0) A.partials[((int64_t)b * (1 + N) + 1 + s) * tv.n_groups + g] = t;
    }
}

hipError_t launch_dense_wide(const DenseArgs& a, bool want_grad, hipStream_t s);     // k_dense_wide.hip: five to eight columns

#define SSDE_L(MODEL, D)                                                                          \
    if (a.model == MODEL && a.d == D) {                                                           \
        if (a.report)                                                                             \
            hipLaunchKernelGGL((dense_kernel<MODEL, D, 0, true>), dim3(a.tv.n_groups, 1), block, 0, s, a); \
        else if (!want_grad)                                                                      \
            hipLaunchKernelGGL((dense_kernel<MODEL, D, 0, false>), dim3(a.tv.n_groups, 1), block, 0, s, a); \
        else                                                                                      \
            hipLaunchKernelGGL((dense_kernel<MODEL, D, DENSE_NT, false>), dim3(a.tv.n_groups, a.n_dirblocks), block, 0, s, a); \
        return hipGetLastError();                                                                 \
    }
#ifndef SSDE_DENSE_WIDE_TU
hipError_t launch_dense(const DenseArgs& a, bool want_grad, hipStream_t s) {
    if (a.tv.n_groups == 0) return hipSuccess;
    dim3 block(WAVE);
    SSDE_L(M_CTCRW, 1) SSDE_L(M_CTCRW, 2) SSDE_L(M_OU_SSM, 1) SSDE_L(M_OU_SSM, 2) SSDE_L(M_BM_SSM, 1) SSDE_L(M_BM_SSM, 2)
    // responses of three or four columns whose measurement covariance or P0 couples the columns (ssde_engine_dist.hip sends every
    // other wide response to this engine pair by pair): one filter over all columns, F by LU as the reference does it
    SSDE_L(M_CTCRW, 3) SSDE_L(M_CTCRW, 4) SSDE_L(M_OU_SSM, 3) SSDE_L(M_OU_SSM, 4) SSDE_L(M_BM_SSM, 3) SSDE_L(M_BM_SSM, 4)
    return launch_dense_wide(a, want_grad, s);
}
#else
// five to eight columns: the same step; the lane's covariance (up to 16 x 16 duals) lives in scratch memory there -- the coverage
// path of a rare configuration (the reference's atomic::logdet branch, nllk_ctcrw.hpp:20-22), not a throughput path
hipError_t launch_dense_wide(const DenseArgs& a, bool want_grad, hipStream_t s) {
    dim3 block(WAVE);
    SSDE_L(M_CTCRW, 5) SSDE_L(M_CTCRW, 6) SSDE_L(M_CTCRW, 7) SSDE_L(M_CTCRW, 8)
    SSDE_L(M_OU_SSM, 5) SSDE_L(M_OU_SSM, 6) SSDE_L(M_OU_SSM, 7) SSDE_L(M_OU_SSM, 8)
    SSDE_L(M_BM_SSM, 5) SSDE_L(M_BM_SSM, 6) SSDE_L(M_BM_SSM, 7) SSDE_L(M_BM_SSM, 8)
    return hipErrorInvalidValue;
}
#endif
#undef SSDE_L

}  // namespace ssde
