// k_iso_colvar.hip -- lane = track Kalman lanes with ROW-VARYING tau / nu (kappa, sigma) for gfx950 (CTCRW, OU_SSM, BM_SSM).
//
// The batch-scale form of the model the reference exists for: the SDE parameters smooth in covariates,
//     par_mat.row(i) = X_fe coeff_fe + X_re coeff_re ;  tau_i = exp(par_mat(i, d)), nu_i = exp(par_mat(i, d + 1))
// (nllk_ctcrw.hpp:143-156, nllk_ou_ssm.hpp:113-124, nllk_bm_ssm.hpp:98-108), filtered by the same loop
// (nllk_ctcrw.hpp:206-241).  The lane = direction kernels (k_tv.hip) spend a wave-row per track-row whatever the batch:
// right for one animal, an order of magnitude off the chip's fp64 rate for 10^4 of them.  Here a lane is a TRACK, as in the
// constant-coefficient kernels, and the gradient with respect to the coefficient of design column k comes from the
// TANGENT of the filter in that direction: the linearised step is the same for every column, only the seed differs --
//     (dP, da)_k  <-  Lin_i (dP, da)_k  +  X_k(i) * seed_type(k)(i)
// where Lin_i is the Jacobian of row i's update + prediction with respect to (P, a) and seed_t the derivative of
// (T, Q, B, H) with respect to log tau (t = 1), log nu / kappa (t = 2) or log sigma_obs (t = 0) at THIS row's parameters.
// 3 + 2 d doubles of state and ~45 fp64 instructions per column and row (CTCRW, d = 2).
//
// One WORKGROUP of EIGHT waves (two per SIMD) per (64-track group, time window).  The waves form a pipeline over the rows, one
// barrier per row, everything between the stages in LDS rings:
//   stage 0, waves 1-6:   each loads a sixth of the channels of the row four ahead (HBM is read ONCE per row: 8 (1 + d + K)
//                         bytes), stores the row two ahead to the ring of rows, and adds ITS channels' terms of the linear
//                         predictors p1 = log tau_i, p2 = log nu_i of the row three ahead (two partial sums per wave and row);
//   stage 1, the last wave:  for the row two ahead, sums the partial predictors, takes the exp's and builds T, Q, B and their
//                         log tau derivatives (makeT/Q/B_ctcrw: nllk_ctcrw.hpp:45-91 through ctcrw_trans);
//   stage 2, wave 0:      the primal filter of the NEXT row (gains, residuals, state, covariance, likelihood terms; the
//                         log sigma_obs and drift-intercept directions), and the row's LINEARISATION: the nine numbers of
//                         Lin_i, the residuals and the seed vectors -- 15 + 3 d doubles per lane.  Its state lives in LDS
//                         between rows: held in registers it would take ~45 of EVERY wave's 256 (a kernel's allocation is the
//                         union of its waves' roles);
//   stage 3, every wave:  the column tangents of the current row from the linearisation -- straight-line code, what a column
//                         feeds is a pair of 0/1 factors on its value, not a branch -- for the up to CV_KC columns dealt to it.
// The two stage waves are long dependent chains -- the row's critical path: they stage nothing, go first on the SIMD they share
// (s_setprio) and get columns only when the six waves between them are full (the engine deals round robin).  A wave's
// registers hold its columns' state and little else, so two waves fit a SIMD and cover each other's LDS / barrier waits.
// Measured on the way here (1e4 tracks x 1e3 rows, 18 columns; lane = direction path 3.70 ms):
//   four waves, each running the primal filter, a uniform branch per column and type             2.06 ms  (the branches: 2100 of 4600 cycles per row)
//   ... straight-line columns, a transition wave                                                 1.54 ms  (> half of the VALU instructions v_accvgpr / v_readlane moves)
//   ... + a filter wave handing the linearisation on, columns in four blocks                      1.36 ms
//   eight waves (two per SIMD, 256 registers each), filter state parked in LDS                    1.26 ms
//   ... a design column both parameters use streamed once (this bench: 9 instead of 18)           1.13-1.16 ms
//   (stage waves freed of staging and columns, 0/1 factors as bit selects instead of an LDS table: 1.13 ms, no change --
//    SQ counters: 22 % of the wave cycles issue VALU, 39 % wait on s_waitcnt, 25 % issue-stalled; ~400 LDS instructions per row)
// Windows, warm-up and the verified hand-over as in k_iso.hip.  Layout: the tiles of ssde_device.hpp with the design
// columns as further channels (as k_iso_drift.hip).
//
// What else uses the same lanes (k_iso_colvar_lanes.hpp) and the same interfaces (Primal: the filter + the linearisation it writes; Cols:
// the tangents that read it):
//   * drift design columns next to those of tau / nu (kinds 3, 4; the MU variants of the kernels): mixed designs;
//   * per-row measurement covariances, H_array: full 4 x 4 covariance lanes for CTCRW with two response columns
//     (CvPrimalCtcrwFull / CvColsCtcrwFull), full 2 x 2 lanes for OU_SSM / BM_SSM (Cv...ScalFull), h = H_i on the isotropic lanes
//     for one response column;
//   * iso_few_kernel: few design columns (tau ~ 1 + x) -- one wave per (group, window) runs the whole row;
//   * iso_full_kernel: H_array with CONSTANT coefficients (the Argos model) -- one wave per (group, window) runs filter and tangents;
//   * k_iso_onewave.hip: iso_few_kernel and iso_full_kernel (below); k_iso_colvar_support.hip: create-time helpers -- column ranges,
//     H statistics, equal-column detection, the reduction of the predictors' ranges.
#include <type_traits>

#include "k_iso_colvar_lanes.hpp"

namespace ssde {

int colvar_nstate(int model, int d, int kc, bool full) {
    if (full) return model == M_CTCRW ? 14 + kc * 14 : 5 + kc * 5;     // d = 2: 4 x 4 covariance (CTCRW), 2 x 2 (OU_SSM, BM_SSM)
    return model == M_CTCRW ? 2 * d + 5 + (3 + 2 * d) + kc * (3 + 2 * d) : d + 2 + (1 + d) + kc * (1 + d);
}

// ---- the kernel ------------------------------------------------------------------------------------------------------------
// accumulators of a part: [value | column 0 .. CV_KC-1 | mu_1 .. mu_d | log sigma_obs]   (value, mu, sigma_obs: part 0)
constexpr int CV_PRODUCER = CV_WAVES - 1;                               // the wave that builds the transitions
constexpr int CV_LOADERS = CV_WAVES - 2;                                // the waves between the two stage waves stage the rows
constexpr int CV_LD = (CV_CMAX + CV_LOADERS - 1) / CV_LOADERS;          // channels a loading wave handles per row (dt, y, H_array entries, the streamed columns)
static_assert(CV_LD * CV_LOADERS >= CV_CMAX && CV_CMAX >= 1 + 2 + 4 + DRIFT_KMAX, "ring of rows too narrow");
constexpr int CV_FILTER = 0;                                            // the wave that runs the primal filter

// KC: column slots per wave (even; the engine picks the instantiation from the widest part)
// MU: drift design columns may be among the columns (always on the full-covariance lanes, whose drift intercepts are such columns)
template <int MODEL, int D, int KC, bool FULL, bool MU>
__global__ __launch_bounds__(CV_WAVES * WAVE) void iso_colvar_kernel(const IsoArgs A, const CvPart* parts) {
    typedef typename CvModel<MODEL, D, KC, FULL>::Primal Primal;
    typedef typename CvModel<MODEL, D, KC, FULL>::Cols Cols;
    typedef typename Primal::Trans Trans;
    constexpr int SD = Primal::SD, NLIN = Primal::NLIN, NTR = Primal::NTR, NPD = Primal::NDUMP;
    __shared__ double raw[3][CV_LD * CV_LOADERS * WAVE];       // the staged rows
    constexpr int NE = MU ? 4 : 2;                             // partial sums per loading wave and row: p1, p2 (and mu_1, mu_2)
    __shared__ double eta[2][(NE * CV_LOADERS + 1) * WAVE];    // ... of every loading wave, and the interval
    __shared__ double trs[2][(NTR + (MU ? 2 : 0)) * WAVE];     // per row: the transition (and the row's drift, when it has design columns)
    __shared__ double lin[2][NLIN * WAVE];                     // per row: the linearisation
    __shared__ double fst[Primal::NSAVE * WAVE];               // the filter's state between rows (wave 0; see below)
    __shared__ double coef[DRIFT_KMAX][4];                     // per streamed column: its coefficient in p1, p2, mu_1, mu_2 (0: not in that predictor)
    __shared__ double wcoef[CV_LOADERS][CV_LD][4];
    if (blockIdx.x == 0 && threadIdx.x == 0 && A.chk_out) *A.chk_out = 0.0;
    const int lane = threadIdx.x & 63, part = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // (a scalar: the part tables are read with scalar loads)
    // the two stage waves are the row's critical path (each a long dependent chain): they neither stage rows nor, unless the
    // other six are full, carry columns
    const bool loader = part != CV_FILTER && part != CV_PRODUCER;
    if (!loader) __builtin_amdgcn_s_setprio(3);                 // (... and they go first on the SIMD they share with a column wave)
    const int ldr = loader ? part - 1 : 0;                      // (waves 1 .. 6: loaders 0 .. 5)
    const TileView& tv = A.tv;
    const int G = tv.n_groups;
    const int g = blockIdx.x % G, chunk = blockIdx.x / G;      // (groups are sorted longest first: the long ones start first)
    const int C = tv.C, c_obs = tv.c_obs, K = A.drift_k, c_col = A.c_col;
    constexpr int nacc = 2 + CV_KC + D;
    if (threadIdx.x == 0) {
#pragma unroll
        for (int k = 0; k < DRIFT_KMAX; k++) { coef[k][0] = A.coefA[k]; coef[k][1] = A.coefB[k]; coef[k][2] = A.coefC[k]; coef[k][3] = A.coefD[k]; }
    }
    __syncthreads();
    // the channels a loading wave stages are c = loader + 6 i; those that are design columns enter the linear predictors with their
    // coefficients, re-packed per wave so that the staging code reads them with one broadcast LDS load per channel
    unsigned col_bits = 0;                                     // bit i: the wave's i-th channel is a design column
#pragma unroll
    for (int i = 0; i < CV_LD; i++) {
        const int k = ldr + CV_LOADERS * i - c_col;
        const bool on = k >= 0 && k < K;
        if (on) col_bits |= 1u << i;
        if (lane < 4 && loader) wcoef[ldr][i][lane] = on ? coef[on ? k : 0][lane] : 0.0;
    }
    const bool grad = A.part_mask[0] != 0;                     // (0: the value only -- no tangents)
    const bool mu_cols = MU && A.cv_mu_cols != 0;              // the drift has design columns too
    const int n_col = grad ? parts[part].n_col : 0;
    const bool with_mu = grad && parts[CV_FILTER].with_mu, with_sig = grad && parts[CV_FILTER].with_sig;
    // per slot: the channel to read, and what the value read is -- a column of ones / a column that feeds par[d] / par[d + 1]
    // (bit masks: the selects are VALU work, of which this kernel has plenty to spare; an LDS table of 0/1 factors cost
    // 16 more LDS reads per wave and row on the LDS pipe, which it has not)
    int chan[KC];
    unsigned ones_bits = 0, t1_bits = 0, t2_bits = 0, t3_bits = 0, t4_bits = 0;      // (kinds 3, 4: the drift of dimension 1, 2 -- full-covariance lanes)
#pragma unroll
    for (int k = 0; k < KC; k++) {
        const bool on = k < n_col;
        const int ch = on ? parts[part].chan[k] : -2, ty = on ? parts[part].type[k] : 0;
        chan[k] = ch >= 0 ? ch : c_col;                        // (an unused slot reads a design column and discards it)
        if (ch == -1) ones_bits |= 1u << k;
        if (ty == 1) t1_bits |= 1u << k;
        if (ty == 2) t2_bits |= 1u << k;
        if (ty == 3) t3_bits |= 1u << k;
        if (ty == 4) t4_bits |= 1u << k;
    }
    const double* base = tv.tiles + tv.group_off[g] + lane;
    const int L = tv.group_len[g];
    const int ns = tv.lane_nsteps[g * WAVE + lane];
    int s_begin, s_acc, s_end;
    window_bounds(L, A.n_chunks, A.window, 0, chunk, s_begin, s_acc, s_end, 0);
    const int pc = part * A.n_chunks + chunk;
    double* const dump0 = A.bnd + (((int64_t)pc * G + g) * 2 + 0) * A.bnd_stride * WAVE + lane;
    double* const dump1 = A.bnd + (((int64_t)pc * G + g) * 2 + 1) * A.bnd_stride * WAVE + lane;
    const bool last_chunk = !(A.n_chunks > 1 && chunk + 1 < A.n_chunks);

    double setA[CV_LD], setB[CV_LD];
    // (a uniform row pointer + a 32-bit lane offset per channel: scalar-base addressing, the row advance is scalar arithmetic)
    const double* const gbase = tv.tiles + tv.group_off[g];
    unsigned voff[CV_LD];
#pragma unroll
    for (int i = 0; i < CV_LD; i++) voff[i] = (unsigned)((ldr + CV_LOADERS * i) * WAVE + lane);
    auto ld = [&](double (&dst)[CV_LD], int s) {               // this wave's channels of row s: HBM -> registers
        const double* rowp = gbase + (int64_t)s * C * WAVE;
#pragma unroll
        for (int i = 0; i < CV_LD; i++) dst[i] = rowp[voff[i]];       // (past the last channel: the next row's first ones -- staged, never used)
    };
    auto st_raw = [&](const double (&src)[CV_LD], int slot) {  // registers -> the ring of rows
#pragma unroll
        for (int i = 0; i < CV_LD; i++) raw[slot][(ldr + CV_LOADERS * i) * WAVE + lane] = src[i];
    };
    auto st_eta = [&](const double (&src)[CV_LD], int slot) {  // this wave's terms of the row's linear predictors
        double pa = 0.0, pb = 0.0, pm0 = 0.0, pm1 = 0.0;
#pragma unroll
        for (int i = 0; i < CV_LD; i++) {
            const double xs = ((col_bits >> i) & 1u) ? src[i] : 0.0;      // (an observation may be NaN: 0 * NaN is not 0)
            pa = fma(wcoef[ldr][i][0], xs, pa);
            if (MODEL != M_BM_SSM) pb = fma(wcoef[ldr][i][1], xs, pb);
            if (mu_cols) { pm0 = fma(wcoef[ldr][i][2], xs, pm0); if (D > 1) pm1 = fma(wcoef[ldr][i][3], xs, pm1); }
        }
        eta[slot][(NE * ldr) * WAVE + lane] = pa;
        eta[slot][(NE * ldr + 1) * WAVE + lane] = pb;
        if constexpr (MU) { if (mu_cols) { eta[slot][(NE * ldr + 2) * WAVE + lane] = pm0; eta[slot][(NE * ldr + 3) * WAVE + lane] = pm1; } }
        if (ldr == 0) eta[slot][(NE * CV_LOADERS) * WAVE + lane] = src[0];      // channel 0: the interval after the row (if the tiles hold it)
    };
    double p1_lo = INFINITY, p1_hi = -INFINITY, p2_lo = INFINITY, p2_hi = -INFINITY;      // (the transition wave: what the predictors reached)
    auto produce = [&](int slot, int s) {                      // stage 1: the transition of row s, whose sums sit in eta[slot]
        const double* e_ = &eta[slot][lane];
        double p1 = A.cv_eta0[0], p2 = A.cv_eta0[1];
#pragma unroll
        for (int w = 0; w < CV_LOADERS; w++) { p1 += e_[(NE * w) * WAVE]; p2 += e_[(NE * w + 1) * WAVE]; }
        if (s < ns) { p1_lo = fmin(p1_lo, p1); p1_hi = fmax(p1_hi, p1); p2_lo = fmin(p2_lo, p2); p2_hi = fmax(p2_hi, p2); }
        const double dtc = e_[(NE * CV_LOADERS) * WAVE];
        const double dt = c_obs ? dtc : tv.dt_all;
        Trans tr;
        Primal::trans(dt, p1, p2, tr);
        Primal::put_trans(&trs[slot][lane], tr);
        if constexpr (MU) if (mu_cols) {                       // a row-varying drift: mu_a(i) = intercept + its columns' terms, handed to the filter with the transition
            double m0 = A.mu[0], m1 = A.mu[D - 1];
#pragma unroll
            for (int w = 0; w < CV_LOADERS; w++) { m0 += e_[(NE * w + 2) * WAVE]; m1 += e_[(NE * w + 3) * WAVE]; }
            trs[slot][NTR * WAVE + lane] = m0; trs[slot][(NTR + 1) * WAVE + lane] = m1;
        }
    };
    // The filter's state lives in LDS between rows: only wave 0 ever touches it, and held in registers across the row loop it
    // would take ~45 of every wave's 256 (a kernel's allocation is the union of its waves' roles)
    Cols S;
    S.init();
    if (part == CV_FILTER) {
        Primal F;
        double a0[SD];
        if (s_begin == 0) {
#pragma unroll
            for (int c = 0; c < SD; c++) a0[c] = tv.a0[((int64_t)g * SD + c) * WAVE + lane];
        } else {
#pragma unroll
            for (int a = 0; a < D; a++) {                      // a window past the first starts from its first observation
                const double y0 = base[((int64_t)s_begin * C + c_obs + a) * WAVE];
                if constexpr (MODEL == M_CTCRW) { a0[2 * a] = (y0 == y0) ? y0 : 0.0; a0[2 * a + 1] = 0.0; }
                else a0[a] = (y0 == y0) ? y0 : 0.0;
            }
        }
        if constexpr (FULL) F.init(a0, A.cv_p0); else F.init(a0, A.p0);
        F.save(&fst[lane]);
    }
    double mu_c[D];
#pragma unroll
    for (int a = 0; a < D; a++) mu_c[a] = A.mu[a];
    const double h = A.h;
    auto filter = [&](int s, int slot3, int slot2) {           // stage 2 (wave 0): row s -- its y in raw[slot3], its transition in trs[slot2]
        Primal F;
        F.restore(&fst[lane]);
        if (s == s_acc && s_acc > s_begin) { F.dump_to(dump0); F.reset_acc(); }
        double* lo = &lin[slot2][lane];
        if (s < ns) {
            const double* r = &raw[slot3][lane];
            double y[D];
#pragma unroll
            for (int a = 0; a < D; a++) y[a] = r[(c_obs + a) * WAVE];
            Trans tr;
            Primal::get_trans(&trs[slot2][lane], tr);
            double mu[D];
#pragma unroll
            for (int a = 0; a < D; a++) { mu[a] = mu_c[a]; if constexpr (MU) { if (mu_cols) mu[a] = trs[slot2][(NTR + a) * WAVE + lane]; } }
            if constexpr (FULL) {
                double H[3] = {h, 0.0, h};                          // H_array[,,i] (symmetric, checked at create)
                if (A.cv_has_h) { H[0] = r[(c_obs + D) * WAVE]; H[1] = r[(c_obs + D + 2) * WAVE]; H[2] = r[(c_obs + D + 3) * WAVE]; }
                F.step(tr, H, mu, y, is_na(y[0], A.any_nan), lo);
            } else {
                // (d = 1 with H_array: the measurement variance of THIS row; no log sigma_obs direction then)
                const double hr = (D == 1 && A.cv_has_h) ? r[(c_obs + D) * WAVE] : h;
                F.step(tr, hr, mu, y, is_na(y[0], A.any_nan), with_sig, with_mu, lo);
            }
        }
        if (s == s_end - 1 && !last_chunk) F.dump_to(dump1);
        F.save(&fst[lane]);
    };
    auto columns = [&](int s, int slot3, int slot2) {          // stage 3: the tangents of row s (waves that carry columns)
        if (s == s_acc && s_acc > s_begin) { S.dump_to(dump0 + NPD * WAVE); S.reset_acc(); }      // (a wave without columns too: the check reads the whole record)
        if (s < ns && n_col > 0) {
            const double* r = &raw[slot3][lane];
            typename Cols::Lin li;
            li.template read<MU>(&lin[slot2][lane]);
            auto quarter = [&](auto k0) {                          // (a wave that also runs a stage is dealt fewer slots: whole quarters are skipped)
                constexpr int K0 = decltype(k0)::value, K1 = K0 + (KC + 3) / 4 < KC ? K0 + (KC + 3) / 4 : KC;
                double X[KC][4];
#pragma unroll
                for (int k = K0; k < K1; k++) {
                    const double xl = r[chan[k] * WAVE];
                    const double xk = ((ones_bits >> k) & 1u) ? 1.0 : xl;
                    X[k][0] = ((t1_bits >> k) & 1u) ? xk : 0.0; X[k][1] = ((t2_bits >> k) & 1u) ? xk : 0.0;
                    X[k][2] = X[k][3] = 0.0;
                    if constexpr (MU) { X[k][2] = ((t3_bits >> k) & 1u) ? xk : 0.0; X[k][3] = ((t4_bits >> k) & 1u) ? xk : 0.0; }
                }
                S.template step<K0, K1, MU>(li, X);
            };
            constexpr int Q = (KC + 3) / 4;
            if (n_col > 0) quarter(std::integral_constant<int, 0>());
            if (Q < KC && n_col > Q) quarter(std::integral_constant<int, (Q < KC ? Q : 0)>());
            if (2 * Q < KC && n_col > 2 * Q) quarter(std::integral_constant<int, (2 * Q < KC ? 2 * Q : 0)>());
            if (3 * Q < KC && n_col > 3 * Q) quarter(std::integral_constant<int, (3 * Q < KC ? 3 * Q : 0)>());
        }
    };
#ifdef SSDE_CV_CLOCK
    // (tuning build: where a wave's cycles go -- staging, the transition / the filter, the columns, the barrier)
    long long ck[4] = {0, 0, 0, 0};
    long long t_ = __builtin_amdgcn_s_memtime();
#define SSDE_CK(i) { const long long n_ = __builtin_amdgcn_s_memtime(); ck[i] += n_ - t_; t_ = n_; }
#else
#define SSDE_CK(i)
#endif
    // Iteration t: rows t + 2 (-> ring of rows) and t + 3 (-> partial predictors) leave the registers, row t + 4 is requested;
    // the transition of row t + 2, the filter on row t + 1, the columns of row t.  X holds row t + 2, Y row t + 3.
    int r3 = 0;                                                // (t + 3 - s_begin) mod 3 == the ring slot of row t
    auto iter = [&](int t, double (&X)[CV_LD], double (&Y)[CV_LD]) {
        const int sl_t = r3, sl_t1 = r3 == 2 ? 0 : r3 + 1, sl_t2 = r3 == 0 ? 2 : r3 - 1;   // slots of rows t, t + 1, t + 2 (t + 2 == t - 1 mod 3)
        if (loader) {
            if (t + 2 >= s_begin) st_raw(X, sl_t2);
            st_eta(Y, (t + 3) & 1);
            ld(X, t + 4);
        }
        SSDE_CK(0)
        if constexpr (!FULL) {                                 // (full-covariance lanes: the stage waves run loops of their own, below)
            if (part == CV_PRODUCER && t + 2 >= s_begin) produce((t + 2) & 1, t + 2);
            if (part == CV_FILTER && t + 1 >= s_begin && t + 1 < s_end) filter(t + 1, sl_t1, (t + 1) & 1);
        }
        SSDE_CK(1)
        if (t >= s_begin) columns(t, sl_t, t & 1);
        SSDE_CK(2)
        __syncthreads();
        SSDE_CK(3)
        r3 = r3 == 2 ? 0 : r3 + 1;
    };
    // Full-covariance lanes: the filter needs ~120 registers of its own and the column state another 120; in ONE loop the
    // allocator keeps both live and spills around the filter in every wave (measured: 800 bytes per lane of scratch, column
    // waves at 12 000 cycles per row).  The two stage waves -- which carry no columns there (the engine sees to it) -- run loops
    // of their own with the same barriers, so that neither allocation contains the other's state.
    if (FULL && part == CV_FILTER) {
        for (int t = s_begin - 3; t < s_end; t++) {
            const int sl_t1 = r3 == 2 ? 0 : r3 + 1;
            SSDE_CK(0)
            if (t + 1 >= s_begin && t + 1 < s_end) filter(t + 1, sl_t1, (t + 1) & 1);
            SSDE_CK(1)
            __syncthreads();
            SSDE_CK(3)
            r3 = r3 == 2 ? 0 : r3 + 1;
        }
    } else if (FULL && part == CV_PRODUCER) {
        for (int t = s_begin - 3; t < s_end; t++) {
            SSDE_CK(0)
            if (t + 2 >= s_begin) produce((t + 2) & 1, t + 2);
            SSDE_CK(1)
            __syncthreads();
            SSDE_CK(3)
        }
    } else {
        if (loader) ld(setB, s_begin);                         // Y of the first iteration (t = s_begin - 3): row s_begin
        iter(s_begin - 3, setA, setB);
        for (int t = s_begin - 2; t < s_end; t += 2) {         // (s_end - s_begin is a multiple of WIN_ALIGN: an even count)
            iter(t, setB, setA);
            iter(t + 1, setA, setB);
        }
    }
#ifdef SSDE_CV_CLOCK
    if (A.wave_clock && lane == 0) {
        double* o = A.wave_clock + 4 * ((int64_t)blockIdx.x * CV_WAVES + part);
        for (int i = 0; i < 4; i++) o[i] = (double)ck[i] / (double)(s_end - s_begin);
    }
#endif
    if (!last_chunk) S.dump_to(dump1 + NPD * WAVE);
    if (part == CV_PRODUCER && A.cv_ranges) {                  // per workgroup: the range of p1 and p2 over its rows (the next evaluation's window plan)
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            p1_lo = fmin(p1_lo, __shfl_xor(p1_lo, o, 64)); p1_hi = fmax(p1_hi, __shfl_xor(p1_hi, o, 64));
            p2_lo = fmin(p2_lo, __shfl_xor(p2_lo, o, 64)); p2_hi = fmax(p2_hi, __shfl_xor(p2_hi, o, 64));
        }
        if (lane == 0) { double* o_ = A.cv_ranges + 4 * (int64_t)blockIdx.x; o_[0] = p1_lo; o_[1] = p1_hi; o_[2] = p2_lo; o_[3] = p2_hi; }
    }
    const bool empty = s_acc >= s_end;
    const bool filt = part == CV_FILTER;
    Primal F;
    F.restore(&fst[lane]);                                     // (wave 0's; the other waves read it for nothing and discard it)
    {
        const double t = wave_sum((empty || !filt) ? 0.0 : F.value());
        if (lane == 0) A.partials[((int64_t)pc * nacc + 0) * G + g] = t;
    }
#pragma unroll
    for (int k = 0; k < CV_KC; k++) {
        const double t = wave_sum((empty || k >= KC) ? 0.0 : S.g[k < KC ? k : 0]);
        if (lane == 0) A.partials[((int64_t)pc * nacc + 1 + k) * G + g] = t;
    }
#pragma unroll
    for (int a = 0; a < D; a++) {
        const double t = wave_sum((empty || !filt) ? 0.0 : F.gmu[a]);
        if (lane == 0) A.partials[((int64_t)pc * nacc + 1 + CV_KC + a) * G + g] = t;
    }
    {
        const double t = wave_sum((empty || !filt) ? 0.0 : F.sg);
        if (lane == 0) A.partials[((int64_t)pc * nacc + 1 + CV_KC + D) * G + g] = t;
    }
}


// a.n_parts == CV_WAVES parts (one per wave of a workgroup), a.drift_k streamed columns (1 .. DRIFT_KMAX), kc: the widest
// part's column count
template <int MODEL, int D>
static hipError_t launch_cv(const IsoArgs& a, const CvPart* parts, int kc, dim3 grid, dim3 block, hipStream_t s) {
    if (a.cv_mu_cols) {
        if (kc <= 2) hipLaunchKernelGGL((iso_colvar_kernel<MODEL, D, 2, false, true>), grid, block, 0, s, a, parts);
        else hipLaunchKernelGGL((iso_colvar_kernel<MODEL, D, 4, false, true>), grid, block, 0, s, a, parts);
    } else {
        if (kc <= 2) hipLaunchKernelGGL((iso_colvar_kernel<MODEL, D, 2, false, false>), grid, block, 0, s, a, parts);
        else hipLaunchKernelGGL((iso_colvar_kernel<MODEL, D, 4, false, false>), grid, block, 0, s, a, parts);
    }
    return hipGetLastError();
}
hipError_t launch_iso_colvar(int model, int d, const IsoArgs& a, const CvPart* parts, int kc, hipStream_t s) {
    if (a.n_parts != CV_WAVES || a.drift_k < 0 || a.drift_k > DRIFT_KMAX || a.tv.C > CV_CMAX || kc < 0 || kc > CV_KC) return hipErrorInvalidValue;
    dim3 grid(a.tv.n_groups * a.n_chunks), block(CV_WAVES * WAVE);
    if (grid.x == 0) return hipSuccess;
    if (a.cv_full) {                                           // full-covariance lanes, d = 2: 4 x 4 (CTCRW), 2 x 2 (OU_SSM, BM_SSM)
        if (d != 2) return hipErrorInvalidValue;
        if (model == M_CTCRW) {
            if (kc <= 2) hipLaunchKernelGGL((iso_colvar_kernel<M_CTCRW, 2, 2, true, true>), grid, block, 0, s, a, parts);
            else if (kc <= 3) hipLaunchKernelGGL((iso_colvar_kernel<M_CTCRW, 2, 3, true, true>), grid, block, 0, s, a, parts);
            else hipLaunchKernelGGL((iso_colvar_kernel<M_CTCRW, 2, 4, true, true>), grid, block, 0, s, a, parts);
        } else if (model == M_OU_SSM) hipLaunchKernelGGL((iso_colvar_kernel<M_OU_SSM, 2, 4, true, true>), grid, block, 0, s, a, parts);
        else if (model == M_BM_SSM) hipLaunchKernelGGL((iso_colvar_kernel<M_BM_SSM, 2, 4, true, true>), grid, block, 0, s, a, parts);
        else return hipErrorInvalidValue;
        return hipGetLastError();
    }
#define SSDE_CASE(M_, D_) if (model == M_ && d == D_) return launch_cv<M_, D_>(a, parts, kc, grid, block, s);
    SSDE_CASE(M_CTCRW, 1) SSDE_CASE(M_CTCRW, 2) SSDE_CASE(M_OU_SSM, 1) SSDE_CASE(M_OU_SSM, 2) SSDE_CASE(M_BM_SSM, 1) SSDE_CASE(M_BM_SSM, 2)
#undef SSDE_CASE
    return hipErrorInvalidValue;
}

}  // namespace ssde
