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

#include "k_tv_filter.hpp"

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
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        if (A.out && !A.chk_items) A.out[A.n_out] = 0.0;             // raised by the finalize launch's checks
        if (A.par0_w) A.par0_w[0] = A.par[0];
    }
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
    // (the compaction is done by all threads at once: one thread walking 26 slots through LDS was 1.5 us of the C1 pre-pass's 10)
    {
        const int k = threadIdx.x;
        const bool in = k < A.n_slots;
        int col = -1, pj = 0;
        double coef = 0.0;
        if (in) { col = T->col[k]; pj = T->par_j[k]; coef = A.par[T->pidx[k]]; }       // all slots' loads in flight together
        const bool streamed = in && col >= 0;
        const unsigned long long bal = __ballot(streamed);
        const int wv_ = k >> 6, ln_ = k & 63;
        __shared__ int s_cnt[4];
        if (ln_ == 0) s_cnt[wv_] = __popcll(bal);
        if (in) { t_col[k] = col; t_pj[k] = pj; t_coef[k] = coef; }
        __syncthreads();
        int m0 = 0;
        for (int w = 0; w < wv_; w++) m0 += s_cnt[w];
        if (streamed) {
            const int m = m0 + __popcll(bal & ((1ull << ln_) - 1ull));
            s_col[m] = col; s_pj[m] = pj; s_coef[m] = coef;
        }
        if (k < MAX_Q) {                                              // the intercepts of parameter k, in slot order
            double base = 0.0;
            for (int q = 0; q < A.n_slots; q++) base += (t_col[q] < 0 && t_pj[q] == k) ? t_coef[q] : 0.0;
            s_base[k] = base;
        }
        if (k == 0) s_ns = s_cnt[0] + s_cnt[1] + s_cnt[2] + s_cnt[3];
    }
    __syncthreads();
    const int nsl = s_ns;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < A.n; i += (int64_t)gridDim.x * 256) {
        double par[Q];
#pragma unroll
        for (int j = 0; j < Q; j++) par[j] = s_base[j];
        // (measured, same session, 10^7 rows x 18 columns: keeping 8 or 16 columns' loads in flight per thread instead of this
        //  plain loop is SLOWER -- 1.89 against 1.37 ms, 10.8 against 9.0 us on a single track: the registers cost occupancy,
        //  and occupancy is what hides the round trips here)
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
        const double dt = (i + 1 < A.n) ? A.times[i + 1] - A.times[i] : A.last_dt;
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

// ---- ESEAL_SSM pre-pass (nllk_e_seal_ssm.hpp:136-137, 170-173): z_i, H_i, drift, q per row ---------------------------
__global__ __launch_bounds__(256) void tv_prepare_eseal_kernel(const TvArgs A) {
    const SlotTable* __restrict__ T = A.slots;
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        if (A.out && !A.chk_items) A.out[A.n_out] = 0.0;
        if (A.par0_w) A.par0_w[0] = A.par[0];
    }
    const double tau = exp(A.par[0]), a1 = A.par[1], a2 = exp(A.par[2]);          // :114-118
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < A.n; i += (int64_t)gridDim.x * 256) {
        double par[2] = {0.0, 0.0};
        for (int k = 0; k < A.n_slots; k++) {
            const int col = T->col[k], j = T->par_j[k];
            const double t = ((col >= 0) ? A.colbuf[(int64_t)col * A.col_stride + i] : 1.0) * A.par[T->pidx[k]];
            par[0] += (j == 0) ? t : 0.0; par[1] += (j == 1) ? t : 0.0;
        }
        const double dt = (i + 1 < A.n) ? A.times[i + 1] - A.times[i] : A.last_dt;       // :104-107
        const double sigma = exp(par[1]);
        double r[TV_RS];
#pragma unroll
        for (int k = 0; k < TV_RS; k++) r[k] = 0.0;
        r[TVE_Z] = a2 / A.eseal_R[i];                                              // makeZ :43-48
        r[TVE_H] = tau * tau / A.eseal_h[i];                                       // makeH :55-59
        r[TVE_DRIFT] = par[0] * dt;                                                // makeT :16-23
        r[TVE_Q] = sigma * sigma * dt;                                             // makeQ :30-35
        r[TVE_DT] = dt; r[TVE_Y] = A.obs[i]; r[TVE_A1] = a1;
        double2* o = (double2*)(A.rec + i * TV_RS);
#pragma unroll
        for (int k = 0; k < TV_RS / 2; k++) o[k] = make_double2(r[2 * k], r[2 * k + 1]);
    }
    // no planner statistics: ESEAL tracks run as one sequential window
    if (threadIdx.x < TV_STATS) A.stats[blockIdx.x * TV_STATS + threadIdx.x] = (threadIdx.x & 1) ? -INFINITY : INFINITY;
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
        // (eight state entries at a time: their loads are in flight together and their reductions interleave -- the launch is a chain
        //  of latencies, C1 waits for it)
        for (int q0 = 0; q0 < nstate; q0 += 8) {
            double err[8], sc[8];
#pragma unroll
            for (int u = 0; u < 8; u++) {
                const bool on = valid && q0 + u < nstate;
                const int q = q0 + u < nstate ? q0 + u : 0;
                const double a = on ? arrived[q * WAVE] : 0.0, b = on ? warmed[q * WAVE] : 0.0;
                err[u] = fabs(a - b); sc[u] = fmax(fabs(a), fabs(b));
                if (on && !(err[u] == err[u])) err[u] = INFINITY;
            }
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) {
#pragma unroll
                for (int u = 0; u < 8; u++) {
                    err[u] = fmax(err[u], __shfl_xor(err[u], o, 64));
                    sc[u] = fmax(sc[u], __shfl_xor(sc[u], o, 64));
                }
            }
#pragma unroll
            for (int u = 0; u < 8; u++)
                if (err[u] > 0.0) worst = fmax(worst, err[u] / sc[u]);
        }
    }
    return worst;
}

// workgroups [0, n_check): four items' checks each (one per wave), raising out[n_out] (zeroed by the pre-pass; a
// non-negative double orders like its bit pattern); the others: one output slot each (0 = nllk, 1.. = gradient)
__global__ __launch_bounds__(256) void tv_finalize_kernel(const TvArgs A, int nstate, int n_check) {
    __shared__ double sh[4];
    const int tid = threadIdx.x;
    if ((int)blockIdx.x < n_check) {
        const int item = blockIdx.x * 4 + (tid >> 6);
        if (item >= A.n_items) return;
        const double w = tv_check_item(A, item, tid & 63, nstate);
        if (A.chk_items) { if ((tid & 63) == 0) A.chk_items[item] = w == w ? w : INFINITY; return; }
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
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o, 64);     // (a fixed order: the same sums on every launch)
    if ((tid & 63) == 0) sh[tid >> 6] = acc;
    __syncthreads();
    if (tid == 0) A.out[slot] = (sh[0] + sh[1]) + (sh[2] + sh[3]);
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

hipError_t launch_tv_prepare(const TvArgs& a, hipStream_t s) {
    if (a.model == M_ESEAL) {
        hipLaunchKernelGGL(tv_prepare_eseal_kernel, dim3(a.stats_blocks), dim3(256), 0, s, a);
        return hipGetLastError();
    }
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

hipError_t launch_tv_filter_dense(const TvArgs& a, bool want_grad, hipStream_t s);   // k_tv_dense.hip

hipError_t launch_tv_filter(const TvArgs& a, bool want_grad, hipStream_t s) {
    if (a.dense) return launch_tv_filter_dense(a, want_grad, s);
    SSDE_TV_LAUNCH_FILTER(TvOps)
}

hipError_t launch_tv_finalize(const TvArgs& a, hipStream_t s) {
    const int sd = a.model == M_CTCRW ? 2 * a.d : a.d;
    const int nstate = a.model == M_ESEAL ? TvEsealLane::NSTATE
                     : a.dense ? 2 * (sd + sd * sd) : (a.model == M_CTCRW ? 4 * a.d + 6 : 2 * a.d + 2);
    const int n_check = (a.n_items + 3) / 4;
    hipLaunchKernelGGL(tv_finalize_kernel, dim3(n_check + a.n_out), dim3(256), 0, s, a, nstate, n_check);
    return hipGetLastError();
}

}  // namespace ssde
