// k_tv_filter.hpp -- the recursion kernel of the lane = gradient direction path (see k_tv.hip), shared by the
// two translation units that instantiate it: k_tv.hip (isotropic lanes) and k_tv_dense.hip (full-covariance
// lanes).  Split only to keep the compile time of either file down.
#ifndef SSDE_K_TV_FILTER_HPP
#define SSDE_K_TV_FILTER_HPP

#include "ssde_device.hpp"
#include "ssde_tv.hpp"

namespace ssde {
namespace {

// ---- the recursion --------------------------------------------------------------------------------
struct TvRow { double r[TV_RS]; double w; };

template <int U>
__device__ __forceinline__ void tv_load_block(TvRow (&dst)[U], const double* rec, const double* wp, int ndp,
                                              int64_t i0, int64_t imax) {
#pragma unroll
    for (int u = 0; u < U; u++) {
        const int64_t i = (i0 + u < imax) ? i0 + u : imax;          // look-ahead rows stay inside the buffers
        const double2* p = (const double2*)(rec + i * TV_RS);
#pragma unroll
        for (int k = 0; k < TV_RS / 2; k++) { const double2 t = p[k]; dst[u].r[2 * k] = t.x; dst[u].r[2 * k + 1] = t.y; }
        dst[u].w = wp[i * ndp];
    }
}

template <class Ops, bool GRAD, bool REPORT>
__global__ __launch_bounds__(WG_WAVES * WAVE, 1) void tv_filter_kernel(const TvArgs A) {
    typedef typename Ops::Lane Lane;
    constexpr int SD = Lane::SD;
    constexpr int TV_U = Ops::U;
    const int item = blockIdx.x * WG_WAVES + (threadIdx.x >> 6);      // one work item per WAVE, no barriers
    if (item >= A.n_items) return;
    const int lane = threadIdx.x & 63;
    const TvItem it = A.items[item];
    const int lpt = 1 << A.lpt_shift, tpw = WAVE >> A.lpt_shift;
    const int tslot = lane >> A.lpt_shift, dslot = lane & (lpt - 1);
    const int64_t trk = (int64_t)it.pack * tpw + tslot;
    const bool has = trk < A.n_tracks;
    const int64_t row0 = has ? A.trk_row0[trk] : 0;
    const int ns = has ? A.trk_ns[trk] : 0;
    const int L = A.trk_ns[(int64_t)it.pack * tpw];                   // longest track of the pack (sorted)
    int s_begin, s_acc, s_end;
    window_bounds(L, it.nc, A.window, 0, it.c, s_begin, s_acc, s_end);

    const int k = it.b * lpt + dslot;
    const TvDir dd = A.dirs[k];
    const int kind = dd.kind, dim = dd.dim;
    const double* wp = A.wdir + k;
    const int64_t imax = A.n - 1;
    // sigma_obs^2: by value, or from the parameter vector when the launch is replayed from a hipGraph
    const double h = A.h_from_par ? exp(2.0 * (A.par0_w ? A.par0_w : A.par)[0]) : A.h;
    double p0[Ops::DENSE ? 16 : 3];
#pragma unroll
    for (int q = 0; q < (Ops::DENSE ? SD * SD : 3); q++) p0[q] = Ops::DENSE ? A.p0f[q] : A.p0[q];

    int ns_min = ns;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) ns_min = min(ns_min, __shfl_xor(ns_min, o, 64));
    ns_min = __builtin_amdgcn_readfirstlane(ns_min);

    TvRow bufA[TV_U], bufB[TV_U];
    tv_load_block<TV_U>(bufA, A.rec, wp, A.ndp, row0 + 1 + s_begin, imax);
    Lane S;
    if constexpr (Ops::DENSE) S.has_h = A.has_h != 0;
    if (s_begin == 0) {
        double a0[SD];
#pragma unroll
        for (int c = 0; c < SD; c++) a0[c] = has ? A.a0[trk * SD + c] : 0.0;
        S.init(a0, p0);
        if (REPORT && has && dslot == 0 && ns > 0) {
#pragma unroll
            for (int c = 0; c < SD; c++) A.report[row0 + (int64_t)c * A.n] = a0[c];
        }
    } else {
        S.warm_init(&bufA[0].r[Ops::Y_OFF], p0);
    }

    auto one = [&](const TvRow& row, int s) {
        Ops::template step<GRAD>(S, row.r, h, kind, dim, row.w, A.any_nan);
        if (REPORT && dslot == 0) {
            double st[SD];
            S.state(st);
#pragma unroll
            for (int c = 0; c < SD; c++) A.report[row0 + 1 + s + (int64_t)c * A.n] = st[c];
        }
    };
    auto run_block = [&](const TvRow (&blk)[TV_U], int s0) {
        if (s0 + TV_U <= ns_min) {
#pragma unroll
            for (int u = 0; u < TV_U; u++) one(blk[u], s0 + u);
        } else {
#pragma unroll
            for (int u = 0; u < TV_U; u++)
                if (s0 + u < ns) one(blk[u], s0 + u);
        }
    };
    auto handover = [&](int s0) {
        if (s0 == s_acc && s_acc > s_begin) {
            double st[Lane::NSTATE];
            S.dump(st);
            double* o = A.bnd + ((int64_t)item * 2 + 0) * TV_NSTATE * WAVE + lane;
#pragma unroll
            for (int q = 0; q < Lane::NSTATE; q++) o[q * WAVE] = st[q];
            S.reset_acc();
        }
    };
    for (int s0 = s_begin; s0 < s_end; s0 += 2 * TV_U) {
        tv_load_block<TV_U>(bufB, A.rec, wp, A.ndp, row0 + 1 + s0 + TV_U, imax);
        handover(s0);
        run_block(bufA, s0);
        tv_load_block<TV_U>(bufA, A.rec, wp, A.ndp, row0 + 1 + s0 + 2 * TV_U, imax);
        if (s0 + TV_U < s_end) {
            handover(s0 + TV_U);
            run_block(bufB, s0 + TV_U);
        }
    }
    if (REPORT) return;
    if (it.c + 1 < it.nc) {
        double st[Lane::NSTATE];
        S.dump(st);
        double* o = A.bnd + ((int64_t)item * 2 + 1) * TV_NSTATE * WAVE + lane;
#pragma unroll
        for (int q = 0; q < Lane::NSTATE; q++) o[q * WAVE] = st[q];
    }
    const bool empty = s_acc >= s_end;
    A.gval[(int64_t)item * WAVE + lane] = (empty || !has) ? 0.0 : S.value();
    A.gdir[(int64_t)item * WAVE + lane] = (empty || !has || !GRAD) ? 0.0 : S.grad();
}

#define SSDE_TV_MODELS(X) X(M_CTCRW, 1) X(M_CTCRW, 2) X(M_OU_SSM, 1) X(M_OU_SSM, 2) X(M_BM_SSM, 1) X(M_BM_SSM, 2)

// launch the instantiation of OPS<MODEL, D> that matches (a.model, a.d, a.report, want_grad)
#define SSDE_TV_LAUNCH_FILTER(OPS)                                                                              \
    if (a.n_items == 0) return hipSuccess;                                                                      \
    dim3 grid((a.n_items + WG_WAVES - 1) / WG_WAVES), block(WG_WAVES * WAVE);                                   \
    SSDE_TV_MODELS(SSDE_TV_LAUNCH_ONE_##OPS)                                                                    \
    return hipErrorInvalidValue;
#define SSDE_TV_LAUNCH_BODY(O)                                                                                  \
        if (a.report) hipLaunchKernelGGL((tv_filter_kernel<O, false, true>), grid, block, 0, s, a);             \
        else if (want_grad) hipLaunchKernelGGL((tv_filter_kernel<O, true, false>), grid, block, 0, s, a);       \
        else hipLaunchKernelGGL((tv_filter_kernel<O, false, false>), grid, block, 0, s, a);                     \
        return hipGetLastError();
#define SSDE_TV_LAUNCH_ONE_TvOps(MODEL, D) if (a.model == MODEL && a.d == D) { typedef TvOps<MODEL, D> O; SSDE_TV_LAUNCH_BODY(O) }
#define SSDE_TV_LAUNCH_ONE_TvDenseOps(MODEL, D) if (a.model == MODEL && a.d == D) { typedef TvDenseOps<MODEL, D> O; SSDE_TV_LAUNCH_BODY(O) }

}  // namespace
}  // namespace ssde
#endif
