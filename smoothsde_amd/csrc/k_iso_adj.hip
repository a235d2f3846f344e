// k_iso_adj.hip -- lane = track Kalman lanes with ROW-VARYING tau / nu (kappa, sigma) and drift, gradient by a REVERSE sweep (gfx950).
//
// The model class of k_iso_colvar.hip (par_mat.row(i) = X coeff, nllk_ctcrw.hpp:143-156; the loop :203-241) with the gradient the
// way SURVEY 7.2(ii) prescribes it: not one forward tangent per design column (3 + 2 d doubles of state and ~40 fp64 instructions per
// column and row: 1828 VALU wave-instructions per 64-track row with 18 columns) but ONE backward recursion over the adjoint of the
// filter state (ssde_adj.hpp), which yields g_j(i) = d nllk / d par_mat(i, j) per row; a column then costs an FMA per row and
// parameter it feeds (grad_coeff += X_k(i) g_j(i), fused into the backward rows).
//
// One WAVE per (64-track group, time window), four independent waves per workgroup (k_iso.hip's arrangement), two passes:
//   pass 1, rows s_begin .. s_hi:  the primal filter -- predictors, exp's, transition, update, prediction; the value over
//            [s_acc, s_end); from s_acc on the state entering every CB-th row goes to a CHECKPOINT buffer in HBM (7 doubles per
//            CB rows and track: 14 B/row for CTCRW with CB = 4 -- the only non-algorithmic traffic);
//   pass 2, blocks of CB rows from s_hi down to s_acc:  the block's rows forwards again from its checkpoint, now leaving each row's
//            RECORD (state entering the row, 1 / F, the transition and its log tau derivatives: 9-18 doubles) in the wave's own LDS
//            slab, then the CB rows BACKWARDS: the transposed step, the row's g_j, and the FMAs of X' g.  The next block's rows
//            and checkpoint are requested before the backward half, which reads its columns from the registers the forward
//            half filled.
// Windows (ssde_engine_iso.hip: plan_windows) as everywhere, in BOTH directions: the adjoint forgets through the same closed-loop
// matrix transposed, so a window that is not the last runs `adj_tail` = `window` rows past its end (s_hi = s_end + adj_tail: rows it walks but
// whose gradient terms belong to the next window) and starts the backward recursion there from zero.  The hand-over record of a
// boundary holds the forward state AND the adjoint at that row from both sides; iso_finalize_kernel compares them like any other
// record, and a failed check widens the windows (ssde_engine.hip: run_checked).
//
// Every direction comes out of the one sweep: log sigma_obs (2 h sum gh), the drift intercepts (sum gmu), the intercepts of
// par[d] / par[d + 1] (sum g1, sum g2) and, per streamed column k, X_k' g1, X_k' g2 (and X_k' gmu_a when the drift has design
// columns too: MU).  Fixed parameters are simply not mapped by the engine.
#include <type_traits>

#include "ssde_adj.hpp"
#include "ssde_device.hpp"

namespace ssde {

// rows of a backward block: its records fill the wave's LDS slab (at most 36 KB: four waves per CU)
constexpr int adj_block_rows(int nf) { return 76 / nf < 2 ? 2 : (76 / nf > 4 ? 4 : 76 / nf); }
// streamed columns the instantiations are built for
int adj_ks(int k) { return k <= 6 ? 6 : k <= 9 ? 9 : k <= 12 ? 12 : k <= 18 ? 18 : -1; }
int adj_nk(int model, int d, bool mu) { return (model != M_BM_SSM ? 2 : 1) + (mu ? d : 0); }
// accumulators: [value | log sigma_obs | mu_1 .. mu_d | par[d] | par[d + 1] | per streamed column: its kinds]
int adj_nacc(int model, int d, int k, bool mu) { return 4 + d + adj_ks(k) * adj_nk(model, d, mu); }
int adj_nstate(int model, int d, bool full) { return full ? 2 * (model == M_CTCRW ? 14 : 5) : 2 * (model == M_CTCRW ? 2 * d + 3 : d + 1); }
int adj_ckpt_rows(int model, int d, bool full) {
    if (full) return model == M_CTCRW ? adj_block_rows(AdjFull<M_CTCRW>::NF) : adj_block_rows(AdjFull<M_OU_SSM>::NF);
    if (model == M_CTCRW) return d == 1 ? adj_block_rows(AdjCtcrw<1>::NF) : adj_block_rows(AdjCtcrw<2>::NF);
    return d == 1 ? adj_block_rows(AdjScal<1, true>::NF) : adj_block_rows(AdjScal<2, true>::NF);
}

// FULL: two response columns with a full covariance (per-row H_array: ssde_adj.hpp, AdjFull) instead of the isotropic lanes
template <int MODEL, int D, bool FULL>
struct AdjLaneSel { typedef typename AdjModel<MODEL, D>::Lane type; };
template <int MODEL>
struct AdjLaneSel<MODEL, 2, true> { typedef AdjFull<MODEL> type; };

template <int MODEL, int D, int KS, bool MU, bool FULL>
__global__ __launch_bounds__(WG_WAVES * WAVE, 1) void iso_adj_kernel(const IsoArgs A) {
    static_assert(!FULL || D == 2, "full-covariance lanes: two response columns");
    typedef typename AdjLaneSel<MODEL, D, FULL>::type Lane;
    typedef typename Lane::Trans Trans;
    typedef typename Lane::Adj Adj;
    constexpr int NST = Lane::NST, NF = Lane::NF, SD = Lane::SD, CB = adj_block_rows(NF);
    constexpr bool P2 = MODEL != M_BM_SSM;
    constexpr int NKP = P2 ? 2 : 1, NK = NKP + (MU ? D : 0);
    constexpr int NHS = FULL ? 3 : 1, CO = 1 + D + NHS;        // register row: [dt | y | h (d = 1 with H_array: H_i; FULL: H00 H01 H11) | the streamed columns]
    constexpr int W = CO + KS;
    constexpr double LDS_ = FULL ? 1.0 : (double)D;            // the value: (LDS_ sum log F + sum u' F^-1 u) / 2 (FULL: the lanes accumulate log det F)
    constexpr int nacc = 4 + D + KS * NK;
    __shared__ double recs[WG_WAVES][CB * NF * WAVE];
    if (blockIdx.x == 0 && threadIdx.x == 0 && A.chk_out) *A.chk_out = 0.0;
    // the wave's index as a SCALAR (the compiler takes threadIdx.x >> 6 for a per-lane value: every window bound, row address and
    // event test below would be vector arithmetic and exec-mask branches): the work item decoded as decode_block does
    const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int item = blockIdx.x * WG_WAVES + wv;
    const int chunk = (item >> 3) % A.n_chunks, g = ((item >> 3) / A.n_chunks) * 8 + (item & 7);
    if (g >= A.tv.n_groups) {
        if (lane == 0 && A.cv_ranges) { double* o = A.cv_ranges + 4 * (int64_t)item; o[0] = o[2] = INFINITY; o[1] = o[3] = -INFINITY; }
        return;
    }
    const TileView& tv = A.tv;
    const int C = tv.C, c_obs = tv.c_obs, G = tv.n_groups, K = A.drift_k, c_col = A.c_col;
    const bool grad = A.part_mask[0] != 0;
    const bool has_h = (D == 1 || FULL) && A.cv_has_h != 0;
    const double* const gbase = tv.tiles + tv.group_off[g];      // (uniform: a row's address is scalar arithmetic + the lane's 32-bit offset)
    const int L = tv.group_len[g];
    const int ns = tv.lane_nsteps[g * WAVE + lane];
    int s_begin, s_acc, s_end;
    window_bounds(L, A.n_chunks, A.window, 0, chunk, s_begin, s_acc, s_end, 0);
    const bool last_chunk = !(A.n_chunks > 1 && chunk + 1 < A.n_chunks);
    // the backward recursion of a window that is not the last starts `window` rows past its end, from zero
    int s_hi = s_end;
    if (grad && !last_chunk) { s_hi = s_end + A.adj_tail; if (s_hi > L) s_hi = L; }
    const int lim = ns < s_hi ? ns : s_hi;                     // this lane's rows
    // (timing experiments, SSDE_ADJ_DIAG: bit 0 rows from the cache, bit 1 one checkpoint slot, bit 2 no backward pass -- the hand-over
    //  records then go to a scratch slot, so that the check sees the zeros the buffer was created with)
    const int64_t dslot = A.adj_diag ? (int64_t)A.n_chunks * G : (int64_t)chunk * G + g;
    double* const dump0 = A.bnd + (dslot * 2 + 0) * A.bnd_stride * WAVE + lane;
    double* const dump1 = A.bnd + (dslot * 2 + 1) * A.bnd_stride * WAVE + lane;
    double* const ck = A.adj_ckpt + (int64_t)item * A.adj_ckpt_stride + lane;
    const int ck_mul = (A.adj_diag & 2) ? 0 : 1;
    double* const rec = &recs[wv][lane];
    const double h = A.h;

    // a row's channels by BUFFER loads: one resource over the window's rows (scalar registers), the row as a scalar byte offset, the
    // channel + lane as a 32-bit vector offset formed once -- no vector address arithmetic per load
#ifndef ADJ_LOADS
#define ADJ_LOADS 1
#endif
#if ADJ_LOADS == 1
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)(gbase + (int64_t)s_begin * C * WAVE), 0, 0x7fffffff, 0x00020000);
    unsigned voff[W];
    voff[0] = (unsigned)lane * 8u;
#pragma unroll
    for (int a = 0; a < D; a++) voff[1 + a] = (unsigned)((c_obs + a) * WAVE + lane) * 8u;
#pragma unroll
    for (int i = 0; i < NHS; i++) voff[1 + D + i] = (unsigned)((c_obs + D + (i == 0 ? 0 : i + 1)) * WAVE + lane) * 8u;      // H_array[,,i] column-major: 00 | (10) 01 11
#pragma unroll
    for (int k = 0; k < KS; k++) voff[CO + k] = (unsigned)((c_col + (k < K ? k : 0)) * WAVE + lane) * 8u;      // (past the last column: column 0 again, coefficient 0)
    auto load_row = [&](double (&dst)[W], int s) {
        if (A.adj_diag & 1) s = s_begin + (s & 3);                 // (timing experiment: every row from the cache -- the numbers mean nothing)
        const int so = (s - s_begin) * C * (WAVE * 8);
#ifndef ADJ_AUX
#define ADJ_AUX 0
#endif
        auto at = [&](int i) { return __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(rsrc, voff[i], so, ADJ_AUX)); };
        dst[0] = tv.dt_all;
        if (c_obs) dst[0] = at(0);
#pragma unroll
        for (int a = 0; a < D; a++) dst[1 + a] = at(1 + a);
#pragma unroll
        for (int i = 0; i < NHS; i++) dst[1 + D + i] = i == 1 ? 0.0 : h;      // (sigma_obs^2 I without H_array)
        if (has_h) {
#pragma unroll
            for (int i = 0; i < NHS; i++) dst[1 + D + i] = at(1 + D + i);
        }
#pragma unroll
        for (int k = 0; k < KS; k++) dst[CO + k] = at(CO + k);
    };
#else
    auto load_row = [&](double (&dst)[W], int s) {
        if (A.adj_diag & 1) s = s_begin + (s & 3);
        const double* p = gbase + (int64_t)s * C * WAVE;
        const unsigned ul = (unsigned)lane;
        dst[0] = tv.dt_all;
        if (c_obs) dst[0] = p[ul];
#pragma unroll
        for (int a = 0; a < D; a++) dst[1 + a] = p[(unsigned)((c_obs + a) * WAVE) + ul];
#pragma unroll
        for (int i = 0; i < NHS; i++) dst[1 + D + i] = i == 1 ? 0.0 : h;
        if (has_h) {
#pragma unroll
            for (int i = 0; i < NHS; i++) dst[1 + D + i] = p[(unsigned)((c_obs + D + (i == 0 ? 0 : i + 1)) * WAVE) + ul];
        }
#pragma unroll
        for (int k = 0; k < KS; k++) dst[CO + k] = p[(unsigned)((c_col + (k < K ? k : 0)) * WAVE) + ul];
    };
#endif
    // the row's linear predictors (nllk_ctcrw.hpp:143-149) and drift
    auto predictors = [&](const double (&r)[W], double& p1, double& p2, double (&mu)[D]) {
        p1 = A.cv_eta0[0]; p2 = A.cv_eta0[1];
#pragma unroll
        for (int a = 0; a < D; a++) mu[a] = A.mu[a];
#pragma unroll
        for (int k = 0; k < KS; k++) {
            const double x = r[CO + k];
            p1 = fma(A.coefA[k], x, p1);
            if (P2) p2 = fma(A.coefB[k], x, p2);
            if (MU) { mu[0] = fma(A.coefC[k], x, mu[0]); if (D > 1) mu[D - 1] = fma(A.coefD[k], x, mu[D - 1]); }
        }
    };
    // (the prediction after a track's last row is never used -- Q4 -- and its interval may be anything: a benign one)
    auto row_dt = [&](const double (&r)[W], int s) { return s == ns - 1 ? 1.0 : r[0]; };

    // ---- pass 1: the primal filter over [s_begin, s_hi) ------------------------------------------------------------------------
    Lane F;
    {
        double a0[SD];
        if (s_begin == 0) {
#pragma unroll
            for (int c = 0; c < SD; c++) a0[c] = tv.a0[((int64_t)g * SD + c) * WAVE + lane];
        } else {
#pragma unroll
            for (int a = 0; a < D; a++) {                      // a window past the first starts from its first observation
                const double y0 = gbase[((int64_t)s_begin * C + c_obs + a) * WAVE + lane];
                if constexpr (MODEL == M_CTCRW) { a0[2 * a] = (y0 == y0) ? y0 : 0.0; a0[2 * a + 1] = 0.0; }
                else a0[a] = (y0 == y0) ? y0 : 0.0;
            }
        }
        if constexpr (FULL) F.init(a0, A.cv_p0); else F.init(a0, A.p0);
    }
    LogAcc ld;
    ld.init();
    double accq = 0.0, value = 0.0;
    double p1_lo = INFINITY, p1_hi = -INFINITY, p2_lo = INFINITY, p2_hi = -INFINITY;
    {
        // blocks of U rows: their predictors, exp's and transitions first -- U independent chains the scheduler can interleave
        // (a lone wave per SIMD has nothing else to cover the latency of a dependent fp64 chain) --, then the U filter steps
#ifndef ADJ_U1
#define ADJ_U1 4
#endif
        constexpr int U = ADJ_U1;
        double bufA[U][W], bufB[U][W];
        auto block = [&](const double (&blk)[U][W], int s0) {
            Trans tr[U];
            double mu[U][D];
#pragma unroll
            for (int u = 0; u < U; u++) {
                double p1, p2;
                predictors(blk[u], p1, p2, mu[u]);
#ifndef ADJ_RANGE_EVERY
#define ADJ_RANGE_EVERY 0
#endif
                if (ADJ_RANGE_EVERY || u == 0) {                   // (the range the predictors reach, for the next plan: every U-th row is plenty for
                    const bool mine = s0 + u < lim;                //  smooth functions of a covariate; rows past the lane's track hold padding)
                    p1_lo = fmin(p1_lo, mine ? p1 : INFINITY); p1_hi = fmax(p1_hi, mine ? p1 : -INFINITY);
                    p2_lo = fmin(p2_lo, mine ? p2 : INFINITY); p2_hi = fmax(p2_hi, mine ? p2 : -INFINITY);
                }
                Lane::trans(row_dt(blk[u], s0 + u), p1, p2, tr[u]);
            }
#pragma unroll
            for (int u = 0; u < U; u++) {
                const int s = s0 + u;
                if (s >= s_hi) break;                              // (uniform)
                if (s == s_acc) { if (s_acc > s_begin) F.template put<WAVE>(dump0); ld.init(); accq = 0.0; }
                if (s == s_end) { value = 0.5 * (LDS_ * ld.value() + accq); if (!last_chunk) F.template put<WAVE>(dump1); }
                if (grad && s >= s_acc && (s - s_acc) % CB == 0) F.template put<WAVE>(ck + (int64_t)((s - s_acc) / CB * ck_mul) * NST * WAVE);
                if (s < lim) {
                    if constexpr (FULL) F.template fwd<false, WAVE>(tr[u], &blk[u][1 + D], mu[u], &blk[u][1], is_na(blk[u][1], A.any_nan), ld, accq, nullptr);
                    else F.template fwd<false, WAVE>(tr[u], blk[u][1 + D], mu[u], &blk[u][1], is_na(blk[u][1], A.any_nan), ld, accq, nullptr);
                }
            }
        };
#pragma unroll
        for (int u = 0; u < U; u++) load_row(bufA[u], s_begin + u);
        for (int s0 = s_begin; s0 < s_hi; s0 += 2 * U) {
#pragma unroll
            for (int u = 0; u < U; u++) load_row(bufB[u], s0 + U + u);
            block(bufA, s0);
#pragma unroll
            for (int u = 0; u < U; u++) load_row(bufA[u], s0 + 2 * U + u);
            if (s0 + U < s_hi) block(bufB, s0 + U);
        }
        if (s_end >= s_hi) {                                       // (no rows past the window's end: the events of row s_end)
            if (s_acc >= s_hi) { ld.init(); accq = 0.0; }
            value = 0.5 * (LDS_ * ld.value() + accq);
            if (!last_chunk) F.template put<WAVE>(dump1);
        }
    }
    if (A.cv_ranges) {                                             // the range of p1 and p2 over the wave's rows (the next evaluation's window plan)
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            p1_lo = fmin(p1_lo, __shfl_xor(p1_lo, o, 64)); p1_hi = fmax(p1_hi, __shfl_xor(p1_hi, o, 64));
            p2_lo = fmin(p2_lo, __shfl_xor(p2_lo, o, 64)); p2_hi = fmax(p2_hi, __shfl_xor(p2_hi, o, 64));
        }
        if (lane == 0) { double* o_ = A.cv_ranges + 4 * (int64_t)item; o_[0] = p1_lo; o_[1] = p1_hi; o_[2] = p2_lo; o_[3] = p2_hi; }
    }

    // ---- pass 2: blocks of CB rows, forwards from the checkpoint (records), then backwards ------------------------------------------
    double gs1 = 0.0, gs2 = 0.0, gsm[D], gsh = 0.0, acc[KS][NK];
#pragma unroll
    for (int a = 0; a < D; a++) gsm[a] = 0.0;
#pragma unroll
    for (int k = 0; k < KS; k++)
#pragma unroll
        for (int j = 0; j < NK; j++) acc[k][j] = 0.0;
    const bool empty = s_acc >= s_end;
    if (grad && !empty && !(A.adj_diag & 4)) {
        const int nblk = (s_hi - s_acc + CB - 1) / CB;
        Adj Lm;
        Lm.zero();
        // ONE block of rows in registers.  The forward half reads a block's rows in order, the backward half in reverse; the slot a
        // backward row has just released takes the NEXT block's forward rows in THEIR order -- so consecutive blocks use the slots in
        // opposite orders (REV), every row is loaded once per block, and a load has three backward rows to arrive.
        double rows[CB][W], ckc[NST];
#pragma unroll
        for (int i = 0; i < NST; i++) ckc[i] = ck[((int64_t)(nblk - 1) * ck_mul * NST + i) * WAVE];
#pragma unroll
        for (int r = 0; r < CB; r++) load_row(rows[r], s_acc + (nblk - 1) * CB + r);      // (rows past s_hi: inside the allocation -- the spare rows -- and never used)
        auto do_blk = [&](auto rev_tag, int j) {
            constexpr bool REV = decltype(rev_tag)::value;
            const int b0 = s_acc + j * CB;
            Lane Fb;
            Fb.template get<1>(ckc);
            LogAcc ld2;
            ld2.init();
            double aq2 = 0.0, mub[CB][D];
            // the block's transitions first (CB independent chains), straight into the rows' records ...
#pragma unroll
            for (int r = 0; r < CB; r++) {
                const double (&row)[W] = rows[REV ? CB - 1 - r : r];
                double p1, p2;
                predictors(row, p1, p2, mub[r]);
                Trans tr;
                Lane::trans(row_dt(row, b0 + r), p1, p2, tr);
                Lane::template put_trans<WAVE>(rec + r * NF * WAVE, tr);
            }
            // ... then the filter steps, which read theirs back
#pragma unroll
            for (int r = 0; r < CB; r++) {
                const int s = b0 + r;
                const double (&row)[W] = rows[REV ? CB - 1 - r : r];
                if (s < lim) {
                    Trans tr;
                    Lane::template get_trans<WAVE>(rec + r * NF * WAVE, row_dt(row, s), tr);
                    if constexpr (FULL) Fb.template fwd<true, WAVE>(tr, &row[1 + D], mub[r], &row[1], is_na(row[1], A.any_nan), ld2, aq2, rec + r * NF * WAVE);
                    else Fb.template fwd<true, WAVE>(tr, row[1 + D], mub[r], &row[1], is_na(row[1], A.any_nan), ld2, aq2, rec + r * NF * WAVE);
                }
            }
            if (j > 0) {
#pragma unroll
                for (int i = 0; i < NST; i++) ckc[i] = ck[((int64_t)(j - 1) * ck_mul * NST + i) * WAVE];
            }
#pragma unroll
            for (int r = CB - 1; r >= 0; r--) {
                const int s = b0 + r;
                double (&row)[W] = rows[REV ? CB - 1 - r : r];
                if (s < lim) {
                    AdjRowGrad<D> gr;
                    if constexpr (FULL) Lane::template bwd<WAVE>(Lm, rec + r * NF * WAVE, &row[1 + D], mub[r], &row[1], row_dt(row, s), gr);
                    else Lane::template bwd<WAVE>(Lm, rec + r * NF * WAVE, row[1 + D], mub[r], row_dt(row, s), gr);
                    if (s < s_end) {                               // (uniform: the rows past the window's end belong to the next window)
                        gs1 += gr.g1; gs2 += gr.g2; gsh += gr.gh;
#pragma unroll
                        for (int a = 0; a < D; a++) gsm[a] += gr.gmu[a];
#pragma unroll
                        for (int k = 0; k < KS; k++) {
                            const double x = row[CO + k];
                            acc[k][0] = fma(x, gr.g1, acc[k][0]);
                            if (P2) acc[k][NKP - 1] = fma(x, gr.g2, acc[k][NKP - 1]);
                            if (MU) {
#pragma unroll
                                for (int a = 0; a < D; a++) acc[k][NKP + a] = fma(x, gr.gmu[a], acc[k][NKP + a]);
                            }
                        }
                    }
                }
                if (s == s_end && !last_chunk) Lm.template put<WAVE>(dump1 + NST * WAVE);      // the adjoint entering the next window's first row
                if (j > 0) load_row(row, b0 - CB + (CB - 1 - r));  // the slot is free: the next block's forward row CB - 1 - r
            }
        };
        for (int j = nblk - 1; j >= 0; j -= 2) {
            do_blk(std::false_type(), j);
            if (j >= 1) do_blk(std::true_type(), j - 1);
        }
        if (s_acc > s_begin) Lm.template put<WAVE>(dump0 + NST * WAVE);
    } else if (grad) {
        // an empty window still owns its boundary records (the check reads both sides): zeros
        Adj Lm;
        Lm.zero();
        if (s_acc > s_begin) Lm.template put<WAVE>(dump0 + NST * WAVE);
        if (!last_chunk) Lm.template put<WAVE>(dump1 + NST * WAVE);
    }
    auto out = [&](int k, double v) {
        const double t = wave_sum(empty ? 0.0 : v);
        if (lane == 0) A.partials[((int64_t)chunk * nacc + k) * G + g] = t;
    };
    out(0, value);
    out(1, 2.0 * h * gsh);                                         // d / d log sigma_obs: h = sigma_obs^2 (not mapped with H_array)
#pragma unroll
    for (int a = 0; a < D; a++) out(2 + a, gsm[a]);
    out(2 + D, gs1);
    out(3 + D, gs2);
#pragma unroll
    for (int k = 0; k < KS; k++)
#pragma unroll
        for (int j = 0; j < NK; j++) out(4 + D + k * NK + j, acc[k][j]);
}

// one wave per (group, window); a.drift_k <= 18 streamed columns (adj_ks), <= 9 when the drift has design columns too; a.adj_ckpt: [work item][a.adj_ckpt_stride] doubles
hipError_t launch_iso_adj(int model, int d, const IsoArgs& a0, hipStream_t s) {
    const int ks = adj_ks(a0.drift_k);
    if ((a0.cv_full && d != 2) || (a0.cv_has_h && d != 1 && !a0.cv_full) || ks < 0 || !a0.adj_ckpt) return hipErrorInvalidValue;
    IsoArgs a = a0;
    a.n_parts = 1;
    const int g8 = (a.tv.n_groups + 7) / 8;
    dim3 grid((g8 * 8 * a.n_chunks + WG_WAVES - 1) / WG_WAVES), block(WG_WAVES * WAVE);
    if (grid.x == 0) return hipSuccess;
    const bool mu = a.cv_mu_cols != 0;
#define SSDE_KS(M_, D_, F_) { \
        if (mu) { \
            if (ks == 6) hipLaunchKernelGGL((iso_adj_kernel<M_, D_, 6, true, F_>), grid, block, 0, s, a); \
            else if (ks == 9) hipLaunchKernelGGL((iso_adj_kernel<M_, D_, 9, true, F_>), grid, block, 0, s, a); \
            else return hipErrorInvalidValue; \
        } \
        else if (ks == 6) hipLaunchKernelGGL((iso_adj_kernel<M_, D_, 6, false, F_>), grid, block, 0, s, a); \
        else if (ks == 9) hipLaunchKernelGGL((iso_adj_kernel<M_, D_, 9, false, F_>), grid, block, 0, s, a); \
        else if (ks == 12) hipLaunchKernelGGL((iso_adj_kernel<M_, D_, 12, false, F_>), grid, block, 0, s, a); \
        else hipLaunchKernelGGL((iso_adj_kernel<M_, D_, 18, false, F_>), grid, block, 0, s, a); \
        return hipGetLastError(); }
#define SSDE_CASE(M_, D_) if (model == M_ && d == D_ && !a.cv_full) SSDE_KS(M_, D_, false)
    SSDE_CASE(M_CTCRW, 1) SSDE_CASE(M_CTCRW, 2) SSDE_CASE(M_OU_SSM, 1) SSDE_CASE(M_OU_SSM, 2) SSDE_CASE(M_BM_SSM, 1) SSDE_CASE(M_BM_SSM, 2)
#undef SSDE_CASE
#define SSDE_CASE(M_) if (model == M_ && a.cv_full) SSDE_KS(M_, 2, true)
    SSDE_CASE(M_CTCRW) SSDE_CASE(M_OU_SSM) SSDE_CASE(M_BM_SSM)
#undef SSDE_CASE
#undef SSDE_KS
    return hipErrorInvalidValue;
}
// work items of a launch (the grid's waves) and checkpoints a window of `rows` scored + trailing rows needs
int adj_items(int n_groups, int n_chunks) { return ((n_groups + 7) / 8 * 8 * n_chunks + WG_WAVES - 1) / WG_WAVES * WG_WAVES; }

}  // namespace ssde
