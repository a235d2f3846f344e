// k_tv.hip -- isotropic Kalman kernels for ROW-VARYING SDE parameters on gfx950 (CTCRW, OU_SSM,
// BM_SSM with H = sigma_obs^2 I): the models with smooth / covariate-dependent parameters that
// the reference is built for (nllk_ctcrw.hpp:143-156 feeding :195-247).  Arithmetic: ssde_tv.hpp.
//
// These problems are usually a handful of tracks (one animal, a few thousand fixes, tens of
// spline coefficients): latency-bound, not bandwidth-bound.  So, instead of lane = track:
//
//   tv_prepare_kernel  row-parallel.  Linear predictor (A2), link functions (A3), transition
//                      matrices and their derivatives (A4) for EVERY row at once -> one 128-byte
//                      record per row.  All exp() calls of an evaluation happen here, off the
//                      serial chain.  Also reduces the parameter ranges the window planner uses.
//   tv_filter_kernel   one WAVE per (pack of tracks, time window, direction block);
//                      lane = (track of the pack, gradient direction).  Every lane repeats the
//                      primal recursion (~45 fp64 ops/row) and carries ONE tangent (~70): a whole
//                      gradient with up to 64 coefficients costs one lane's latency.  Records and
//                      weights are prefetched a 4-row block ahead in ping-pong registers.
//                      Time windows with a verified hand-over (as in k_iso.hip) cut the serial
//                      chain of a long track into concurrent pieces.
//   tv_finalize_kernel largest relative disagreement at every window hand-over + fixed-order sums
//                      -> [nllk, gradient..., hand-over check], one launch.
#include <algorithm>

#include "ssde_device.hpp"
#include "ssde_tv.hpp"

namespace ssde {

namespace {

// ---- one-off: per-row weight of every direction, row-major [n][ndp] ----------------------------
__global__ __launch_bounds__(256) void tv_weights_kernel(const TvArgs A, double* wdir) {
    const int64_t total = A.n * A.ndp;
    for (int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x; t < total; t += (int64_t)gridDim.x * 256) {
        const int64_t i = t / A.ndp;
        const int k = (int)(t - i * A.ndp);
        const TvDir dd = A.dirs[k];
        double w = 1.0;
        if (dd.kind >= TVK_MU && dd.slot >= 0) {
            const int col = A.slots->col[dd.slot];
            if (col >= 0) w = A.colbuf[(int64_t)col * A.col_stride + i];
        }
        wdir[t] = (dd.kind == TVK_NONE) ? 0.0 : w;
    }
}

// ---- one-off: initial states, a0[t][c] (R/sde.R:549, 576-580 when the caller gives none) -------
__global__ __launch_bounds__(256) void tv_a0_kernel(const TvArgs A, const double* a0_src, const int64_t* trk_seg,
                                                    int64_t n_seg, int sdim, double* a0_dst) {
    const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (t >= A.n_tracks) return;
    for (int c = 0; c < sdim; c++) {
        double v;
        if (a0_src) v = a0_src[trk_seg[t] + (int64_t)c * n_seg];                 // a0 is n_seg x sdim column-major
        else if (A.model == M_CTCRW) v = (c & 1) ? 0.0 : A.obs[A.trk_row0[t] + (int64_t)(c >> 1) * A.n];
        else v = A.obs[A.trk_row0[t] + (int64_t)c * A.n];
        a0_dst[t * sdim + c] = v;
    }
}

// ---- per evaluation: row records ----------------------------------------------------------------
template <int MODEL, int D, bool DENSE>
__global__ __launch_bounds__(256) void tv_prepare_kernel(const TvArgs A) {
    constexpr int Q = (MODEL == M_BM_SSM) ? D + 1 : D + 2;
    constexpr int NS = TV_STATS / 2;
    __shared__ double sh[TV_STATS][4];
    const SlotTable* __restrict__ T = A.slots;
    if (blockIdx.x == 0 && threadIdx.x == 0 && A.out) A.out[A.n_out] = 0.0;   // raised by the finalize launch's checks
    double smin[NS], smax[NS];
#pragma unroll
    for (int k = 0; k < NS; k++) { smin[k] = INFINITY; smax[k] = -INFINITY; }
    // slot table and coefficients once per workgroup into LDS (the row loop then reads them as broadcasts instead
    // of chasing two dependent scalar loads per slot and row); intercept slots are folded into per-parameter constants
    __shared__ double s_coef[MAX_COLS];
    __shared__ int s_col[MAX_COLS], s_pj[MAX_COLS];
    __shared__ double s_base[MAX_Q];
    __shared__ int s_ns;
    __shared__ double t_coef[MAX_COLS];
    __shared__ int t_col[MAX_COLS], t_pj[MAX_COLS];
    if ((int)threadIdx.x < A.n_slots) {                               // all slots' loads in flight together
        const int k = threadIdx.x;
        t_col[k] = T->col[k]; t_pj[k] = T->par_j[k]; t_coef[k] = A.par[T->pidx[k]];
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        int m = 0;
        double base[MAX_Q] = {0.0, 0.0, 0.0, 0.0};
        for (int k = 0; k < A.n_slots; k++) {
            if (t_col[k] < 0) {
#pragma unroll
                for (int j = 0; j < MAX_Q; j++) base[j] += (t_pj[k] == j) ? t_coef[k] : 0.0;
                continue;
            }
            s_col[m] = t_col[k]; s_pj[m] = t_pj[k]; s_coef[m] = t_coef[k]; m++;
        }
        s_ns = m;
        for (int j = 0; j < MAX_Q; j++) s_base[j] = base[j];
    }
    __syncthreads();
    const int nsl = s_ns;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < A.n; i += (int64_t)gridDim.x * 256) {
        double par[Q];
#pragma unroll
        for (int j = 0; j < Q; j++) par[j] = s_base[j];
        for (int k = 0; k < nsl; k++) {                               // par_vec = X_fe coeff_fe + X_re coeff_re
            const int j = s_pj[k];
            const double t = A.colbuf[(int64_t)s_col[k] * A.col_stride + i] * s_coef[k];
#pragma unroll
            for (int jj = 0; jj < Q; jj++) par[jj] += (j == jj) ? t : 0.0;
        }
        // the interval after the row, dtimes(n-1) = 1 (nllk_ctcrw.hpp:126-129).  At a track's last row it spans
        // to the next track (Q4): the filter state it produces is discarded, but REPORT(aest_all) shows it, so
        // the record follows the reference there too; only the planner's statistics skip those rows
        const bool used = (i + 1 < A.n) && ((A.scored[(i + 1) >> 5] >> ((i + 1) & 31)) & 1u);
        const double dt = (i + 1 < A.n) ? A.times[i + 1] - A.times[i] : 1.0;
        double y[D];
#pragma unroll
        for (int a = 0; a < D; a++) y[a] = A.obs[i + (int64_t)a * A.n];
        double r[TV_RS];
        double hmax = A.h;
        if (DENSE) {
            double hrow[D * D];
            if (A.has_h) {
                hmax = -INFINITY;
#pragma unroll
                for (int k = 0; k < D * D; k++) hrow[k] = A.h_array[i * (D * D) + k];   // H_array[,,i]
#pragma unroll
                for (int a = 0; a < D; a++) hmax = fmax(hmax, hrow[a + a * D]);
            }
            tv_make_record_dense<MODEL, D>(dt, par, y, A.has_h ? hrow : nullptr, r);
        } else {
            tv_make_record<MODEL, D>(dt, par, y, r);
        }
        double2* o = (double2*)(A.rec + i * TV_RS);
#pragma unroll
        for (int k = 0; k < TV_RS / 2; k++) o[k] = make_double2(r[2 * k], r[2 * k + 1]);
        if (used) {
            const double v[NS] = {dt, par[D], Q > D + 1 ? par[Q - 1] : 0.0, hmax};
#pragma unroll
            for (int k = 0; k < NS; k++) { smin[k] = fmin(smin[k], v[k]); smax[k] = fmax(smax[k], v[k]); }
        }
    }
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
#pragma unroll
    for (int k = 0; k < NS; k++) {
        double a = smin[k], b = smax[k];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) { a = fmin(a, __shfl_xor(a, o, 64)); b = fmax(b, __shfl_xor(b, o, 64)); }
        if (lane == 0) { sh[2 * k][wv] = a; sh[2 * k + 1][wv] = b; }
    }
    __syncthreads();
    if (threadIdx.x < TV_STATS) {
        const int k = threadIdx.x;
        const bool mn = (k & 1) == 0;
        double a = sh[k][0];
        for (int w = 1; w < 4; w++) a = mn ? fmin(a, sh[k][w]) : fmax(a, sh[k][w]);
        A.stats[blockIdx.x * TV_STATS + k] = a;
    }
}

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
    const double h = A.h_from_par ? exp(2.0 * A.par[0]) : A.h;
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

// ---- finalize: hand-over checks + fixed-order sums in one launch -------------------------------------------------
// check: item (pack, c, b) arrived vs item + 1 = (pack, c + 1, b) warmed up; one wave per item
__device__ __forceinline__ double tv_check_item(const TvArgs& A, int item, int lane, int nstate) {
    const TvItem it = A.items[item];
    double worst = 0.0;
    if (it.c + 1 < it.nc) {
        const int tpw = WAVE >> A.lpt_shift;
        const int64_t trk = (int64_t)it.pack * tpw + (lane >> A.lpt_shift);
        const int ns = trk < A.n_tracks ? A.trk_ns[trk] : 0;
        const int L = A.trk_ns[(int64_t)it.pack * tpw];
        int sb_, s_next, se_;
        window_bounds(L, it.nc, A.window, 0, it.c + 1, sb_, s_next, se_);
        const bool valid = (ns > s_next) && (s_next < L);
        const double* arrived = A.bnd + ((int64_t)item * 2 + 1) * TV_NSTATE * WAVE + lane;
        const double* warmed = A.bnd + ((int64_t)(item + 1) * 2 + 0) * TV_NSTATE * WAVE + lane;
        for (int q = 0; q < nstate; q++) {
            const double a = valid ? arrived[q * WAVE] : 0.0, b = valid ? warmed[q * WAVE] : 0.0;
            double err = fabs(a - b), sc = fmax(fabs(a), fabs(b));
            if (valid && !(err == err)) err = INFINITY;
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) {
                err = fmax(err, __shfl_xor(err, o, 64));
                sc = fmax(sc, __shfl_xor(sc, o, 64));
            }
            if (err > 0.0) worst = fmax(worst, err / sc);
        }
    }
    return worst;
}

// workgroups [0, n_check): four items' checks each (one per wave), raising out[n_out] (zeroed by the pre-pass; a
// non-negative double orders like its bit pattern); the others: one output slot each (0 = nllk, 1.. = gradient)
__global__ __launch_bounds__(256) void tv_finalize_kernel(const TvArgs A, int nstate, int n_check) {
    __shared__ double sh[256];
    const int tid = threadIdx.x;
    if ((int)blockIdx.x < n_check) {
        const int item = blockIdx.x * 4 + (tid >> 6);
        if (item >= A.n_items) return;
        const double w = tv_check_item(A, item, tid & 63, nstate);
        if ((tid & 63) == 0 && w > 0.0)
            atomicMax((unsigned long long*)(A.out + A.n_out), (unsigned long long)__double_as_longlong(w == w ? w : INFINITY));
        return;
    }
    const int slot = blockIdx.x - n_check;
    const int lpt = 1 << A.lpt_shift, tpw = WAVE >> A.lpt_shift;
    double acc = 0.0;
    int b = 0, ds = 0;
    const double* src = A.gval;
    bool any = true;
    if (slot > 0) {
        const int k = A.dir_of_par[slot - 1];
        any = k >= 0;
        b = k >> A.lpt_shift; ds = k & (lpt - 1);
        src = A.gdir;
    }
    if (any) {
        for (int i = tid; i < A.n_items; i += 256) {
            if (A.items[i].b != b) continue;
            const double* p = src + (int64_t)i * WAVE + ds;
            for (int t = 0; t < tpw; t++) acc += p[t * lpt];
        }
    }
    sh[tid] = acc;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (tid < o) sh[tid] += sh[tid + o];
        __syncthreads();
    }
    if (tid == 0) A.out[slot] = sh[0];
}

}  // namespace

hipError_t launch_tv_weights(const TvArgs& a, hipStream_t s) {
    const int64_t total = a.n * a.ndp;
    const unsigned blocks = (unsigned)std::min<int64_t>((total + 255) / 256, 65535);
    hipLaunchKernelGGL(tv_weights_kernel, dim3(blocks), dim3(256), 0, s, a, const_cast<double*>(a.wdir));
    return hipGetLastError();
}

hipError_t launch_tv_a0(const TvArgs& a, const double* a0_src, const int64_t* trk_seg, int64_t n_seg, int sdim,
                        double* a0_dst, hipStream_t s) {
    hipLaunchKernelGGL(tv_a0_kernel, dim3((unsigned)((a.n_tracks + 255) / 256)), dim3(256), 0, s, a, a0_src, trk_seg,
                       n_seg, sdim, a0_dst);
    return hipGetLastError();
}

#define SSDE_TV_MODELS(X) X(M_CTCRW, 1) X(M_CTCRW, 2) X(M_OU_SSM, 1) X(M_OU_SSM, 2) X(M_BM_SSM, 1) X(M_BM_SSM, 2)

hipError_t launch_tv_prepare(const TvArgs& a, hipStream_t s) {
#define X(MODEL, D)                                                                                         \
    if (a.model == MODEL && a.d == D) {                                                                     \
        if (a.dense) hipLaunchKernelGGL((tv_prepare_kernel<MODEL, D, true>), dim3(a.stats_blocks), dim3(256), 0, s, a);  \
        else hipLaunchKernelGGL((tv_prepare_kernel<MODEL, D, false>), dim3(a.stats_blocks), dim3(256), 0, s, a);        \
        return hipGetLastError();                                                                           \
    }
    SSDE_TV_MODELS(X)
#undef X
    return hipErrorInvalidValue;
}

hipError_t launch_tv_filter(const TvArgs& a, bool want_grad, hipStream_t s) {
    if (a.n_items == 0) return hipSuccess;
    dim3 grid((a.n_items + WG_WAVES - 1) / WG_WAVES), block(WG_WAVES * WAVE);
#define XO(OPS)                                                                                             \
        if (a.report) hipLaunchKernelGGL((tv_filter_kernel<OPS, false, true>), grid, block, 0, s, a);       \
        else if (want_grad) hipLaunchKernelGGL((tv_filter_kernel<OPS, true, false>), grid, block, 0, s, a); \
        else hipLaunchKernelGGL((tv_filter_kernel<OPS, false, false>), grid, block, 0, s, a);
#define X(MODEL, D)                                                                                         \
    if (a.model == MODEL && a.d == D) {                                                                     \
        typedef TvOps<MODEL, D> OpsI;                                                                       \
        typedef TvDenseOps<MODEL, D> OpsD;                                                                  \
        if (a.dense) { XO(OpsD) } else { XO(OpsI) }                                                         \
        return hipGetLastError();                                                                           \
    }
    SSDE_TV_MODELS(X)
#undef X
#undef XO
    return hipErrorInvalidValue;
}

hipError_t launch_tv_finalize(const TvArgs& a, hipStream_t s) {
    const int sd = a.model == M_CTCRW ? 2 * a.d : a.d;
    const int nstate = a.dense ? 2 * (sd + sd * sd) : (a.model == M_CTCRW ? 4 * a.d + 6 : 2 * a.d + 2);
    const int n_check = (a.n_items + 3) / 4;
    hipLaunchKernelGGL(tv_finalize_kernel, dim3(n_check + a.n_out), dim3(256), 0, s, a, nstate, n_check);
    return hipGetLastError();
}

}  // namespace ssde
