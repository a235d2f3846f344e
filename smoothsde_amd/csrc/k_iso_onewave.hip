// k_iso_onewave.hip -- the lanes of k_iso_colvar_lanes.hpp with ONE wave per (64-track group, time window) running the whole row
// (filter, linearisation, tangents): iso_full_kernel -- per-row measurement covariances (H_array, nllk_ctcrw.hpp:203-205) with CONSTANT
// tau / nu, the Argos model -- and iso_few_kernel -- row-varying tau / nu with few design columns (tau ~ 1 + x).  The eight-wave
// pipeline of k_iso_colvar.hip pays the filter wave's whole dependent chain per row whatever its other waves do; with at most a
// handful of tangents that chain IS the row, and four independent waves per workgroup (as in k_iso.hip) fill the chip better.
#include "k_iso_colvar_lanes.hpp"

namespace ssde {

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

}  // namespace ssde
