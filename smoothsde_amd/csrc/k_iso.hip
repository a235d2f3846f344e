// k_iso.hip -- constant-coefficient isotropic Kalman kernels for gfx950 (CTCRW, OU_SSM, BM_SSM).
//
// One wavefront lane = one track; a workgroup is ONE wave (64 lanes).  State, covariance and
// all forward sensitivities stay in VGPRs; observations stream from the tiled HBM layout
// (ssde_device.hpp) with coalesced 512-B wave loads, prefetched one ISO_U-step block ahead
// into registers.  fp64 throughout, no MFMA: per lane the recursion is a serial chain of
// scalar fp64 FMAs, so the kernel is bound by fp64 VALU issue, and a lone wave on a SIMD can
// only issue every other fp64 slot.  A 10^4-track batch is just 157 waves for 1024 SIMDs, so
// two more axes of parallelism are layered on the track axis:
//
//  * TIME WINDOWS.  A track is cut into n_chunks windows; window c > 0 starts `window` rows
//    early from an arbitrary state (first observation, P0, zero sensitivities), runs the same
//    filter WITHOUT scoring, and starts scoring at its own first row.  The Kalman filter and
//    its sensitivity recursion forget their initial condition geometrically (the closed-loop
//    matrix T - K Z is a contraction whenever observations arrive), so after the warm-up the
//    window carries the sequential filter's state to rounding error.  This is VERIFIED, not
//    assumed: every window dumps its state at its first scored row, the previous window dumps
//    its state at the same row, window_check_kernel reduces the largest relative disagreement
//    over every component (state, covariance, all sensitivities), and the evaluation reports
//    it next to the result; the host re-evaluates with a longer warm-up (up to one single
//    window = the plain sequential filter) when it exceeds its threshold.
//  * DIRECTION PARTS.  The gradient directions may be split over "parts" that each recompute
//    the primal filter; all parts of one (group, window) get workgroup ids that are equal
//    mod 8, i.e. one XCD, so the tile is fetched from HBM once and re-served from that L2.
//
// Reference arithmetic: ssde_math.hpp (nllk_ctcrw.hpp:195-247, nllk_ou_ssm.hpp:163-213,
// nllk_bm_ssm.hpp:127-175).
#include "ssde_device.hpp"
#include "ssde_tf.hpp"

namespace ssde {

// Steps per prefetch block of THIS kernel (the tiles are padded to TILE_U = 4 rows and end with 64 spare rows; windows
// are multiples of 16 rows): the look-ahead is one block, so the block length is the distance the loads lead by.
#ifndef SSDE_ISO_U
#define SSDE_ISO_U 4
#endif
constexpr int ISO_U = SSDE_ISO_U;
#ifndef SSDE_ISO_WAVES
#define SSDE_ISO_WAVES 2        // waves per SIMD the general kernels are built for
#endif
#ifndef SSDE_ISO_WAVES_IRR
#define SSDE_ISO_WAVES_IRR 1    // ... CTCRW on an irregular grid (a per-row transition: 16 more doubles per lane)
#endif
#ifndef SSDE_ISO_WAVES_SCAL
#define SSDE_ISO_WAVES_SCAL 2   // ... the scalar-covariance models (OU_SSM, BM_SSM)
#endif
static_assert(ISO_U % TILE_U == 0 && WIN_ALIGN % (2 * ISO_U) == 0 && 3 * ISO_U <= TILE_SPARE, "prefetch block");
// ... of the kernels with quiet rows (below): those rows cost a third of a general row, so the loads lead by twice as many
// Kernels with quiet rows (below): block flags and general rows in blocks of quiet_u rows -- 4 for CTCRW, whose general lane fills
// the registers of two waves per SIMD with 4-row blocks as it is --, quiet rows in blocks of QUIET_UQ (they cost a fifth of a
// general row, so the loads lead by more rows)
template <int MODEL>
__host__ __device__ constexpr int quiet_u() { return MODEL == M_CTCRW ? 4 : 8; }
constexpr int QUIET_UQ = 8;
static_assert(16 % TILE_U == 0 && WIN_ALIGN % 16 == 0 && 3 * 16 <= TILE_SPARE, "prefetch block");

// register block [dt | y_1..y_D]; the tile may carry no dt channel (c_obs == 0): slot 0 is then left alone
template <int C, int U>
__device__ __forceinline__ void load_block(double (&dst)[U][C], const double* p, int Cr, int c_obs) {
#pragma unroll
    for (int u = 0; u < U; u++) {
        dst[u][0] = 0.0;
        if (c_obs) dst[u][0] = p[(u * Cr) * WAVE];
#pragma unroll
        for (int c = 1; c < C; c++) dst[u][c] = p[(u * Cr + c_obs + c - 1) * WAVE];
    }
}

// ---- per-model lane policies --------------------------------------------------------------
template <int MODEL, int D, int MASK>
struct LaneOps;

template <int D, int MASK>
struct LaneOps<M_CTCRW, D, MASK> {
    typedef CtcrwLane<D, MASK> State;
    typedef CtcrwTrans Trans;
    static constexpr int SD = 2 * D;
    __device__ static __forceinline__ Trans hoisted(const IsoArgs& A) { return A.ctr; }
    __device__ static __forceinline__ void init(State& S, const double* a0, const IsoArgs& A) {
        S.init(a0, A.p0[0], A.p0[1], A.p0[2]);
    }
    __device__ static __forceinline__ void warm_init(State& S, const double* y, const IsoArgs& A) {
        double a0[SD];
#pragma unroll
        for (int a = 0; a < D; a++) { a0[2 * a] = (y[a] == y[a]) ? y[a] : 0.0; a0[2 * a + 1] = 0.0; }
        S.init(a0, A.p0[0], A.p0[1], A.p0[2]);
    }
    // UNI: regular grid -- the transition is the one of the kernel arguments (scalar registers), the dt slot is not read
    template <bool UNI>
    __device__ static __forceinline__ void step(State& S, const IsoArgs& A, const double* mu, const double* row) {
        if (UNI) {
            ctcrw_step<D, MASK>(S, A.ctr, A.h, mu, row + 1, is_na(row[1], A.any_nan));
        } else {
            Trans tr;
            ctcrw_trans(row[0], A.tau, A.beta, A.sigma, tr);
            ctcrw_step<D, MASK>(S, tr, A.h, mu, row + 1, is_na(row[1], A.any_nan));
        }
    }
    __device__ static __forceinline__ void finish(const State& S, double* out) { ctcrw_finish<D, MASK>(S, out); }
};

template <int MODEL, int D, int MASK>
struct LaneOps {  // OU_SSM / BM_SSM
    typedef ScalLane<D, MASK> State;
    typedef ScalTrans Trans;
    static constexpr int SD = D;
    __device__ static __forceinline__ Trans hoisted(const IsoArgs& A) { return A.str; }
    __device__ static __forceinline__ void init(State& S, const double* a0, const IsoArgs& A) { S.init(a0, A.p0[0]); }
    __device__ static __forceinline__ void warm_init(State& S, const double* y, const IsoArgs& A) {
        double a0[SD];
#pragma unroll
        for (int a = 0; a < D; a++) a0[a] = (y[a] == y[a]) ? y[a] : 0.0;
        S.init(a0, A.p0[0]);
    }
    template <bool UNI>
    __device__ static __forceinline__ void step(State& S, const IsoArgs& A, const double* mu, const double* row) {
        if (UNI) {
            scal_step<D, MASK, MODEL == M_OU_SSM>(S, A.str, A.h, mu, row + 1, is_na(row[1], A.any_nan));
        } else {
            Trans tr;
            if (MODEL == M_OU_SSM) ou_trans(row[0], A.tau, A.sigma, tr);
            else bm_trans(row[0], A.sigma, tr);
            scal_step<D, MASK, MODEL == M_OU_SSM>(S, tr, A.h, mu, row + 1, is_na(row[1], A.any_nan));
        }
    }
    __device__ static __forceinline__ void finish(const State& S, double* out) { scal_finish<D, MASK>(S, out); }
};


// ---- quiet rows (regular grid) ---------------------------------------------------------------------------------------
// A lane whose covariance is at its stationary value needs the mean half of the step only, with the gains and their
// sensitivities as constants (IsoArgs.statc): the stationary-only lanes of the shared-covariance kernels (ssde_tf.hpp: the
// transfer-function form for CTCRW, the basis form for OU_SSM / BM_SSM -- 11 and 10 fp64 instructions per row and dimension
// against ~80 of the general step's mean half).  The engine marks the blocks of 8 rows in which some lane of the group
// misses an observation (IsoArgs.nan_bits).  After such a block the lanes run the general step for quiet_w + 2 blocks (their
// covariance forgets the prediction step) with the stationary lanes WARMING UP beside it -- a fixed linear filter of the
// observations that forgets its start at the same rate --, then the stationary lanes alone score the rows, until the next marked
// block: there their state goes back to the direction form (their hand-over dump is that conversion), the covariance and its
// sensitivities restart from the stationary values, and the general step takes over.  A window that starts past the transient
// of P0 warms up on the stationary lanes alone, like a window of the shared-covariance kernels.  The sums of a quiet stretch are
// folded into the general accumulators; the data-independent terms (log F, dF / F) are counted per lane (nq).
// (error over the wave) / (scale over the wave) of one component, as in window_check_block
__device__ __forceinline__ double pair_ratio(double a, double b, bool valid) {
    double err = valid ? fabs(a - b) : 0.0, sc = valid ? fmax(fabs(a), fabs(b)) : 0.0;
    if (valid && !(err == err)) err = INFINITY;
#pragma unroll
    for (int of = 32; of > 0; of >>= 1) {
        err = fmax(err, __shfl_xor(err, of, 64));
        sc = fmax(sc, __shfl_xor(sc, of, 64));
    }
    return err > 0.0 ? err / sc : 0.0;
}

template <int MODEL, int D, int MASK>
struct QuietOps {   // OU_SSM / BM_SSM
    typedef ScalLane<D, MASK> State;
    typedef BasisScal<MODEL, D, MASK> Stat;
    static constexpr bool HAS_P2 = (MODEL == M_OU_SSM);
    // start at a row: its observation as the state (what the general lanes' warm_init takes); obs_row points at the row's first
    // response channel for this lane, channels `stride` doubles apart
    __device__ static __forceinline__ void start_at(Stat& F, const double* obs_row, const double*, int64_t stride) {
        double y[D], a0[D];
#pragma unroll
        for (int a = 0; a < D; a++) y[a] = obs_row[a * stride];
        Stat::warm_a0(y, a0);
        F.init(a0);
    }
    // back to the direction form, covariance at its stationary value
    __device__ static __forceinline__ void leave(const Stat& F, State& S, const IsoArgs& A) {
        double o[Stat::NSTATE];
        F.dump(o);
        int k = 0;
#pragma unroll
        for (int a = 0; a < D; a++) S.M.x[a] = o[k++];
#pragma unroll
        for (int j = 0; j < NDIRP; j++) {
            if (!(MASK & dir_bit(j)) || (j == 2 && !HAS_P2)) continue;
#pragma unroll
            for (int a = 0; a < D; a++) S.M.tx[j][a] = o[k++];
            S.C.dp[j] = A.quiet_p[3 + 3 * j];
        }
        if (MASK & DIR_MU) {
#pragma unroll
            for (int a = 0; a < D; a++) S.M.mx[a] = o[k++];
        }
        S.C.p = A.quiet_p[0];
    }
    // What the switch to quiet rows assumes, checked where it happens: the lane's covariance and its sensitivities are back at
    // the stationary values, and the state the stationary lanes warmed up to is the general lane's.  (The sensitivities of the
    // state forget at the same rate as the state and are what every window that starts on the stationary lanes hands over.)
    __device__ static __forceinline__ double switch_check(const Stat& F, const State& S, const IsoArgs& A, bool valid) {
        double o[Stat::NSTATE];
        F.dump(o);
        double w = pair_ratio(A.quiet_p[0], S.C.p, valid);
#pragma unroll
        for (int a = 0; a < D; a++) w = fmax(w, pair_ratio(o[a], S.M.x[a], valid));
#pragma unroll
        for (int j = 0; j < NDIRP; j++) {
            if (!(MASK & dir_bit(j)) || (j == 2 && !HAS_P2)) continue;
            w = fmax(w, pair_ratio(A.quiet_p[3 + 3 * j], S.C.dp[j], valid));
        }
        return w;
    }
};

template <int D, int MASK>
struct QuietOps<M_CTCRW, D, MASK> {
    typedef CtcrwLane<D, MASK> State;
    typedef TfCtcrw<D, MASK> Stat;
    // the transfer-function lanes start from the observation of the row before (a block without a missing row)
    __device__ static __forceinline__ void start_at(Stat& F, const double*, const double* obs_prev, int64_t stride) {
        double yp[D];
#pragma unroll
        for (int a = 0; a < D; a++) yp[a] = obs_prev[a * stride];
        F.init(yp);
    }
    __device__ static __forceinline__ void leave(const Stat& F, State& S, const IsoArgs& A) {
        double o[Stat::NSTATE];
        F.dump(o);
        int k = 0;
#pragma unroll
        for (int a = 0; a < D; a++) { S.M.x[a] = o[k++]; S.M.v[a] = o[k++]; }
#pragma unroll
        for (int j = 0; j < NDIRP; j++) {
            if (!(MASK & dir_bit(j))) continue;
#pragma unroll
            for (int a = 0; a < D; a++) { S.M.tx[j][a] = o[k++]; S.M.tv[j][a] = o[k++]; }
            S.C.d11[j] = A.quiet_p[3 + 3 * j]; S.C.d12[j] = A.quiet_p[4 + 3 * j]; S.C.d22[j] = A.quiet_p[5 + 3 * j];
        }
        if (MASK & DIR_MU) {
#pragma unroll
            for (int a = 0; a < D; a++) { S.M.mx[a] = o[k++]; S.M.mv[a] = o[k++]; }
        }
        S.C.p11 = A.quiet_p[0]; S.C.p12 = A.quiet_p[1]; S.C.p22 = A.quiet_p[2];
    }
    __device__ static __forceinline__ double switch_check(const Stat& F, const State& S, const IsoArgs& A, bool valid) {
        double o[Stat::NSTATE];
        F.dump(o);
        double w = fmax(pair_ratio(A.quiet_p[0], S.C.p11, valid), fmax(pair_ratio(A.quiet_p[1], S.C.p12, valid), pair_ratio(A.quiet_p[2], S.C.p22, valid)));
#pragma unroll
        for (int a = 0; a < D; a++) w = fmax(w, fmax(pair_ratio(o[2 * a], S.M.x[a], valid), pair_ratio(o[2 * a + 1], S.M.v[a], valid)));
#pragma unroll
        for (int j = 0; j < NDIRP; j++) {
            if (!(MASK & dir_bit(j))) continue;
            w = fmax(w, fmax(pair_ratio(A.quiet_p[3 + 3 * j], S.C.d11[j], valid),
                             fmax(pair_ratio(A.quiet_p[4 + 3 * j], S.C.d12[j], valid), pair_ratio(A.quiet_p[5 + 3 * j], S.C.d22[j], valid))));
        }
        return w;
    }
};

// DER >= 0: the covariance direction DER (1 = par[d] for BM_SSM, 2 = par[d+1] for CTCRW / OU_SSM) is NOT carried
// through the recursion but derived from the log sigma_obs direction.  The filter is homogeneous in the variances:
// scaling (sigma_obs^2, process variance, P0) by c scales P and F by c and leaves the mean and the gains alone, so
//   CTCRW / BM_SSM (variance ~ nu^2, sigma^2):  dP_der = 2 P - dP_sig,  dx_der = -dx_sig,  g_der = D N - Q - g_sig
//   OU_SSM (variance ~ kappa):                  dP_der = P - dP_sig/2,  dx_der = -dx_sig/2, g_der = (D N - Q - g_sig)/2
// (N = scored rows, Q = sum u'F^-1 u) up to the sensitivity to P0, which the filter forgets like everything else
// about its start: windows c >= 1 use the identity (a third less sensitivity arithmetic), window 0 carries every
// direction, and the hand-over check compares the derived components with window c - 1's at every boundary.
template <int MODEL, int DER>
struct DeriveOps {
    static constexpr double CP = (MODEL == M_OU_SSM) ? 1.0 : 2.0, CB = (MODEL == M_OU_SSM) ? 0.5 : 1.0;
    // fill block DER of a direction-form dump (layout of CtcrwLane / ScalLane ::dump) from block 0 (sigma_obs)
    template <int D>
    __device__ static __forceinline__ void fill(double* st) {
        if (MODEL == M_CTCRW) {
            constexpr int B0 = 2 * D + 3, BS = 3 + 2 * D;
#pragma unroll
            for (int q = 0; q < 3; q++) st[B0 + DER * BS + q] = CP * st[2 * D + q] - CB * st[B0 + q];
#pragma unroll
            for (int q = 3; q < BS; q++) st[B0 + DER * BS + q] = -CB * st[B0 + q];
        } else {
            constexpr int B0 = D + 1, BS = 1 + D;
            st[B0 + DER * BS] = CP * st[D] - CB * st[B0];
#pragma unroll
            for (int q = 1; q < BS; q++) st[B0 + DER * BS + q] = -CB * st[B0 + q];
        }
    }
};

template <int MODEL, int D, int MASK, bool UNI, int DER = -1>
__device__ __forceinline__ void run_lane(const IsoArgs& A, int g, int part, int chunk) {
    typedef LaneOps<MODEL, D, MASK> Ops;
    constexpr int C = 1 + D;
    constexpr int NACC = 4 + D;
    constexpr int SD = Ops::SD;
    const int lane = threadIdx.x & 63;
    const TileView& tv = A.tv;
    const int Cr = tv.C, c_obs = tv.c_obs;     // channels per step in the tile, channel of the first obs column
    const double* base = tv.tiles + tv.group_off[g] + lane;
    const int L = tv.group_len[g];
    const int ns = tv.lane_nsteps[g * WAVE + lane];
    const int pc = part * A.n_chunks + chunk;

    // this workgroup's window: rows [s_acc, s_end) are scored, rows [s_begin, s_acc) warm up
    int s_begin, s_acc, s_end;
    window_bounds(L, A.n_chunks, A.window, A.t0, chunk, s_begin, s_acc, s_end, A.t0_delta);

    typename Ops::State S;
    double mu[D];
#pragma unroll
    for (int a = 0; a < D; a++) mu[a] = A.mu[a];

    int ns_min = ns;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) ns_min = min(ns_min, __shfl_xor(ns_min, o, 64));
    ns_min = __builtin_amdgcn_readfirstlane(ns_min);

    // two register blocks in ping-pong: while one is consumed the other is in flight (no copies)
    double bufA[ISO_U][C], bufB[ISO_U][C];
    load_block<C, ISO_U>(bufA, base + (int64_t)s_begin * Cr * WAVE, Cr, c_obs);
    if (s_begin == 0) {
        double a0[SD];
#pragma unroll
        for (int c = 0; c < SD; c++) a0[c] = tv.a0[((int64_t)g * SD + c) * WAVE + lane];
        Ops::init(S, a0, A);
    } else {
        Ops::warm_init(S, &bufA[0][1], A);
    }

    auto run_block = [&](const double (&blk)[ISO_U][C], int s0) {
        if (s0 + ISO_U <= ns_min) {          // every lane's track covers the block: no per-row predication
#pragma unroll
            for (int u = 0; u < ISO_U; u++) Ops::template step<UNI>(S, A, mu, blk[u]);
        } else {
#pragma unroll
            for (int u = 0; u < ISO_U; u++)
                if (s0 + u < ns) Ops::template step<UNI>(S, A, mu, blk[u]);
        }
    };
    auto handover = [&](int s0) {
        if (s0 == s_acc && s_acc > s_begin) {
            // end of warm-up: publish the state for the hand-over check, start scoring from zero
            double st[Ops::State::NSTATE];
            S.dump(st);
            if (DER >= 0) DeriveOps<MODEL, DER>::template fill<D>(st);
            double* o = A.bnd + (((int64_t)pc * tv.n_groups + g) * 2 + 0) * A.bnd_stride * WAVE + lane;
#pragma unroll
            for (int k = 0; k < Ops::State::NSTATE; k++) o[k * WAVE] = st[k];
            S.reset_acc();
        }
    };
    for (int s0 = s_begin; s0 < s_end; s0 += 2 * ISO_U) {
        // TILE_SPARE keeps the look-ahead loads inside the allocation
        load_block<C, ISO_U>(bufB, base + (int64_t)(s0 + ISO_U) * Cr * WAVE, Cr, c_obs);
        handover(s0);
        run_block(bufA, s0);
        load_block<C, ISO_U>(bufA, base + (int64_t)(s0 + 2 * ISO_U) * Cr * WAVE, Cr, c_obs);
        if (s0 + ISO_U < s_end) {
            handover(s0 + ISO_U);
            run_block(bufB, s0 + ISO_U);
        }
    }
    if (A.n_chunks > 1 && chunk + 1 < A.n_chunks) {
        // state on arrival at the next window's first scored row
        double st[Ops::State::NSTATE];
        S.dump(st);
        if (DER >= 0) DeriveOps<MODEL, DER>::template fill<D>(st);
        double* o = A.bnd + (((int64_t)pc * tv.n_groups + g) * 2 + 1) * A.bnd_stride * WAVE + lane;
#pragma unroll
        for (int k = 0; k < Ops::State::NSTATE; k++) o[k * WAVE] = st[k];
    }
    double out[NACC];
    Ops::finish(S, out);
    if (DER >= 0) {   // gradient of the derived direction: slots [value, sig, mu.., par[d], par[d+1]]
        constexpr double CB = DeriveOps<MODEL, DER>::CB;
        out[1 + D + DER] = CB * ((double)D * S.C.nupd - S.M.accq - out[1]);
    }
    if (s_acc >= s_end) {
#pragma unroll
        for (int k = 0; k < NACC; k++) out[k] = 0.0;  // empty window
    }
#pragma unroll
    for (int k = 0; k < NACC; k++) {
        const double t = wave_sum(out[k]);
        if (lane == 0) A.partials[((int64_t)pc * NACC + k) * tv.n_groups + g] = t;
    }
}

// The general lane with QUIET ROWS (see QuietOps above): stretches of general rows and stretches of quiet rows as loops of their
// own, so that neither the general lane's state is live in the quiet loop nor the stationary lanes' in the general one (as one
// loop with a mode flag the kernel needed both at once: 516 bytes of scratch at two waves per SIMD, or one wave per SIMD).  The
// stationary lanes catch up at a switch by re-reading the quiet_w blocks before it (once per stretch, from L2 mostly).
template <int MODEL, int D, int MASK, int DER = -1>
__device__ __forceinline__ void run_lane_quiet(const IsoArgs& A, int g, int part, int chunk) {
    typedef LaneOps<MODEL, D, MASK> Ops;
    typedef QuietOps<MODEL, D, MASK> QOps;
    constexpr int C = 1 + D;
    constexpr int U = quiet_u<MODEL>();      // rows per prefetch block = rows per block flag
    constexpr int NACC = 4 + D;
    constexpr int SD = Ops::SD;
    const int lane = threadIdx.x & 63;
    const TileView& tv = A.tv;
    const int Cr = tv.C, c_obs = tv.c_obs;
    const double* base = tv.tiles + tv.group_off[g] + lane;
    const int L = tv.group_len[g];
    const int ns = tv.lane_nsteps[g * WAVE + lane];
    const int pc = part * A.n_chunks + chunk;

    int s_begin, s_acc, s_end;
    window_bounds(L, A.n_chunks, A.window, A.t0, chunk, s_begin, s_acc, s_end, A.t0_delta);

    typename Ops::State S;
    double mu[D];
#pragma unroll
    for (int a = 0; a < D; a++) mu[a] = A.mu[a];

    int ns_min = ns;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) ns_min = min(ns_min, __shfl_xor(ns_min, o, 64));
    ns_min = __builtin_amdgcn_readfirstlane(ns_min);

    constexpr int UQ = QUIET_UQ, R = UQ / U;  // rows of a quiet block, block flags per quiet block
    // the block being consumed and the one in flight behind it (general rows); the quiet loop has its own pair
    double cur[U][C], nxt[U][C];
    load_block<C, U>(cur, base + (int64_t)s_begin * Cr * WAVE, Cr, c_obs);
    if (s_begin == 0) {
        double a0[SD];
#pragma unroll
        for (int c = 0; c < SD; c++) a0[c] = tv.a0[((int64_t)g * SD + c) * WAVE + lane];
        Ops::init(S, a0, A);
    } else {
        Ops::warm_init(S, &cur[0][1], A);
    }

    // q_last = the latest block with a missing observation (a window that does not start quiet counts its own start as one: the
    // lane's covariance is a guess there), q_word = the 64 block flags around the current block
    typename QOps::Stat F;
    F.setup(A);
    const unsigned long long* qbits = A.nan_bits + (int64_t)g * A.nan_words;
    int q_last = -A.quiet_w - 2, q_wi = -1;
    unsigned long long q_word = 0ull;
    bool q_mode = false;
    double nq = 0.0;                         // rows this lane scored in quiet mode
    double q_worst = 0.0;                    // largest disagreement found at a switch to quiet rows (wave-uniform)
    auto nanbit = [&](int b) -> bool {
        if ((b >> 6) != q_wi) {
            q_wi = b >> 6;
            const unsigned long long w = qbits[q_wi];
            q_word = ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(w >> 32)) << 32) |
                     (unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(w & 0xffffffffull));
        }
        return ((q_word >> (b & 63)) & 1ull) != 0ull;
    };
    {
        const int b0 = s_begin / U;
        if (b0 > 0) {
            // the row before the window is in a block without a missing row, and the transient of P0 is over: the stationary
            // lanes alone from the first row on (the window's warm-up rows are theirs, as in k_iso_shared.inc)
            if (!nanbit(b0 - 1) && b0 >= A.quiet_b0) {
                QOps::start_at(F, base + ((int64_t)s_begin * Cr + c_obs) * WAVE, base + ((int64_t)(s_begin - 1) * Cr + c_obs) * WAVE, WAVE);
                q_mode = true;
            } else {
                q_last = b0 - 1;
            }
        }
    }
    // fold the sums of a quiet stretch into the general accumulators (ctcrw_finish_parts / scal_finish_parts add them up)
    auto fold = [&]() {
        double fo[NACC];
        F.finish(fo);
        S.M.accq += 2.0 * fo[0];
        if (MASK & DIR_SIG) S.M.gq[0] += fo[1];
        if (MASK & DIR_MU) {
#pragma unroll
            for (int a = 0; a < D; a++) S.M.gmu[a] += fo[2 + a];
        }
        if (MASK & DIR_P1) S.M.gq[1] += fo[2 + D];
        if (MASK & DIR_P2) S.M.gq[2] += fo[3 + D];
        F.reset_acc();
    };
    auto dump_to = [&](int slot) {
        double st[Ops::State::NSTATE];
        S.dump(st);
        if (DER >= 0) DeriveOps<MODEL, DER>::template fill<D>(st);
        double* o = A.bnd + (((int64_t)pc * tv.n_groups + g) * 2 + slot) * A.bnd_stride * WAVE + lane;
#pragma unroll
        for (int k = 0; k < Ops::State::NSTATE; k++) o[k * WAVE] = st[k];
    };
    auto advance = [&]() {
#pragma unroll
        for (int u = 0; u < U; u++)
#pragma unroll
            for (int c = 0; c < C; c++) cur[u][c] = nxt[u][c];
    };
    const bool has_warmup = s_acc > s_begin;
    int s0 = s_begin;
    while (s0 < s_end) {
        if (q_mode) {
            // ---- a quiet stretch: the stationary lanes alone, until a block with a missing observation comes up ----
            double qcur[UQ][C], qnxt[UQ][C];
            load_block<C, UQ>(qcur, base + (int64_t)s0 * Cr * WAVE, Cr, c_obs);              // (once per stretch: not prefetched)
            while (s0 < s_end) {
                if (nanbit(s0 / U) || (R > 1 && nanbit(s0 / U + 1))) break;
                load_block<C, UQ>(qnxt, base + (int64_t)(s0 + UQ) * Cr * WAVE, Cr, c_obs);   // (TILE_SPARE keeps it inside the allocation)
                if (has_warmup && s0 == s_acc) {             // end of warm-up: the state for the hand-over check, scoring from zero
                    QOps::leave(F, S, A);
                    F.reset_acc();
                    dump_to(0);
                    S.reset_acc();
                    nq = 0.0;
                }
                if (s0 + UQ <= ns_min) {
#pragma unroll
                    for (int u = 0; u < UQ; u++) F.step_stat(&qcur[u][1]);
                    nq += (double)UQ;
                } else {
#pragma unroll
                    for (int u = 0; u < UQ; u++)
                        if (s0 + u < ns) { F.step_stat(&qcur[u][1]); nq += 1.0; }
                }
#pragma unroll
                for (int u = 0; u < UQ; u++)
#pragma unroll
                    for (int c = 0; c < C; c++) qcur[u][c] = qnxt[u][c];
                s0 += UQ;
            }
            if (s0 < s_end) {                                // a missing row ahead: the lane's own covariance again
                fold();
                QOps::leave(F, S, A);
                q_mode = false;
#pragma unroll
                for (int u = 0; u < U; u++)
#pragma unroll
                    for (int c = 0; c < C; c++) cur[u][c] = qcur[u][c];
            }
        } else {
            // ---- a general stretch: until every lane has forgotten its last missing row (and the transient of P0) ----
            bool sw = false;
            while (s0 < s_end) {
                const int b = s0 / U;
                if (nanbit(b)) q_last = b;
                else if (b - q_last >= A.quiet_w + 2 && b >= A.quiet_b0 + A.quiet_w && s0 % UQ == 0 && !(R > 1 && nanbit(b + 1))) {
                    sw = true;                // (the whole quiet block ahead is free of missing rows: the quiet loop will take it)
                    break;
                }
                load_block<C, U>(nxt, base + (int64_t)(s0 + U) * Cr * WAVE, Cr, c_obs);
                if (has_warmup && s0 == s_acc) {
                    dump_to(0);
                    S.reset_acc();
                    nq = 0.0;
                }
                if (s0 + U <= ns_min) {
#pragma unroll
                    for (int u = 0; u < U; u++) Ops::template step<true>(S, A, mu, cur[u]);
                } else {
#pragma unroll
                    for (int u = 0; u < U; u++)
                        if (s0 + u < ns) Ops::template step<true>(S, A, mu, cur[u]);
                }
                advance();
                s0 += U;
            }
            if (sw) {
                // the stationary lanes catch up over the quiet_w blocks before this one (no missing row in them or in the block
                // before them): a fixed stable filter of the observations, it forgets its start at the lanes' own rate
                const int r0 = s0 - A.quiet_w * U;
                QOps::start_at(F, base + ((int64_t)r0 * Cr + c_obs) * WAVE, base + ((int64_t)(r0 - 1) * Cr + c_obs) * WAVE, WAVE);
                for (int r = r0; r < s0; r += U) {
                    load_block<C, U>(nxt, base + (int64_t)r * Cr * WAVE, Cr, c_obs);
#pragma unroll
                    for (int u = 0; u < U; u++)
                        if (r + u < ns) F.step_stat(&nxt[u][1]);
                }
                // VERIFIED like a window hand-over: the lane's covariance and its sensitivities against the stationary ones, the
                // state the stationary lanes arrived with against the general lane's (QuietOps::switch_check); the evaluation
                // reports the largest ratio with the hand-over checks' and is repeated with a longer memory when it is too large
                q_worst = fmax(q_worst, QOps::switch_check(F, S, A, s0 < ns));
                F.reset_acc();
                q_mode = true;
            }
        }
    }
    if (A.n_chunks > 1 && chunk + 1 < A.n_chunks) {
        // state on arrival at the next window's first scored row
        if (q_mode) QOps::leave(F, S, A);
        dump_to(1);
    }
    if (q_mode) fold();
    S.C.nupd += nq;
    // (iso_finalize_kernel folds the word into the evaluation's check value and clears it for the next launch)
    if (lane == 0 && q_worst > 0.0 && A.quiet_flag)
        atomicMax((unsigned long long*)A.quiet_flag, (unsigned long long)__double_as_longlong(q_worst == q_worst ? q_worst : INFINITY));
    double out[NACC];
    Ops::finish(S, out);
    {                                        // the data-independent terms of the rows scored in quiet mode
        const double hn = 0.5 * (double)D * nq;
        out[0] = fma(hn, A.quiet_ld, out[0]);
        if (MASK & DIR_SIG) out[1] = fma(hn, A.quiet_gld[0], out[1]);
        if (MASK & DIR_P1) out[2 + D] = fma(hn, A.quiet_gld[1], out[2 + D]);
        if (MASK & DIR_P2) out[3 + D] = fma(hn, A.quiet_gld[2], out[3 + D]);
    }
    if (DER >= 0) {   // gradient of the derived direction: slots [value, sig, mu.., par[d], par[d+1]]
        constexpr double CB = DeriveOps<MODEL, DER>::CB;
        out[1 + D + DER] = CB * ((double)D * S.C.nupd - S.M.accq - out[1]);
    }
    if (s_acc >= s_end) {
#pragma unroll
        for (int k = 0; k < NACC; k++) out[k] = 0.0;  // empty window
    }
#pragma unroll
    for (int k = 0; k < NACC; k++) {
        const double t = wave_sum(out[k]);
        if (lane == 0) A.partials[((int64_t)pc * NACC + k) * tv.n_groups + g] = t;
    }
}

// One kernel per (direction mask, regular / irregular grid) when the whole launch uses a single mask (n_parts == 1, the
// default): a kernel's register allocation is the worst case over everything it contains.
template <int MODEL, int D, int MASK, bool UNI>
__device__ __forceinline__ void iso_mask_body(const IsoArgs& A) {
    if (blockIdx.x == 0 && threadIdx.x == 0 && A.chk_out) *A.chk_out = 0.0;   // raised by the finalize launch
    int g, part, chunk;
    if (!decode_block(A, A.n_chunks, g, part, chunk)) return;
    if (!group_selected(A, g)) return;
    constexpr int DERJ = (MODEL == M_BM_SSM) ? 1 : 2;
    constexpr bool CAN = (MASK & DIR_SIG) != 0 && (MASK & dir_bit(DERJ)) != 0;
    if (CAN && chunk > 0 && A.derive) run_lane<MODEL, D, (MASK & ~dir_bit(DERJ)), UNI, DERJ>(A, g, part, chunk);
    else run_lane<MODEL, D, MASK, UNI>(A, g, part, chunk);
}
template <int MODEL, int D, int MASK>
__device__ __forceinline__ void iso_quiet_body(const IsoArgs& A) {
    if (blockIdx.x == 0 && threadIdx.x == 0 && A.chk_out) *A.chk_out = 0.0;
    int g, part, chunk;
    if (!decode_block(A, A.n_chunks, g, part, chunk)) return;
    if (!group_selected(A, g)) return;
    constexpr int DERJ = (MODEL == M_BM_SSM) ? 1 : 2;
    constexpr bool CAN = (MASK & DIR_SIG) != 0 && (MASK & dir_bit(DERJ)) != 0;
    if (CAN && chunk > 0 && A.derive) run_lane_quiet<MODEL, D, (MASK & ~dir_bit(DERJ)), DERJ>(A, g, part, chunk);
    else run_lane_quiet<MODEL, D, MASK>(A, g, part, chunk);
}

// Two waves per SIMD (256 registers each): a lone wave leaves the fp64 pipe idle behind its dependent instructions
// and its loads (measured 4.5-5.4 cycles per wave-instruction against the pipe's 4); the engine plans twice as many
// time windows for the general kernel (ssde_engine.hip).  With the transition of a regular grid in scalar registers
// and the fused step of ssde_math.hpp the CTCRW lanes fit as well (they did not in round 1: one wave, 416 + 160
// registers, a fifth of the issue slots spent on AGPR copies).
// (CTCRW on an irregular grid carries a per-row transition -- 16 more doubles per lane and an exp -- and spills under
// that cap: one wave per SIMD with the whole register file, as before.)
template <int MODEL, int D, int MASK, bool UNI>
__global__ __launch_bounds__(WG_WAVES * WAVE, (MODEL == M_CTCRW) ? (UNI ? SSDE_ISO_WAVES : SSDE_ISO_WAVES_IRR) : SSDE_ISO_WAVES_SCAL) void iso_mask_kernel(const IsoArgs A) {
    iso_mask_body<MODEL, D, MASK, UNI>(A);
}
// ... the same lanes with quiet rows (regular grid, a.quiet_w > 0): a kernel of its own, so that batches without quiet rows run
// the code (and the register allocation) they always ran
template <int MODEL, int D, int MASK>
__global__ __launch_bounds__(WG_WAVES * WAVE, (MODEL == M_CTCRW) ? SSDE_ISO_WAVES : SSDE_ISO_WAVES_SCAL) void iso_quiet_kernel(const IsoArgs A) {
    iso_quiet_body<MODEL, D, MASK>(A);
}

// Direction-split launches (several parts with different masks; a testing path) keep the masks in one kernel.
template <int MODEL, int D>
__global__ __launch_bounds__(WG_WAVES * WAVE) void iso_kernel(const IsoArgs A) {
    if (blockIdx.x == 0 && threadIdx.x == 0 && A.chk_out) *A.chk_out = 0.0;
    int g, part, chunk;
    if (!decode_block(A, A.n_chunks, g, part, chunk)) return;
    if (!group_selected(A, g)) return;
    // (no dynamic indexing into the by-value argument block: that would force a scratch copy)
    const int mask = part == 0 ? A.part_mask[0] : part == 1 ? A.part_mask[1] : part == 2 ? A.part_mask[2] : A.part_mask[3];
    const bool uni = A.uniform_dt != 0;
    switch (mask) {
#define SSDE_CASE(M) case M: if (uni) run_lane<MODEL, D, M, true>(A, g, part, chunk); else run_lane<MODEL, D, M, false>(A, g, part, chunk); break;
        SSDE_CASE(0) SSDE_CASE(1) SSDE_CASE(2) SSDE_CASE(3) SSDE_CASE(4) SSDE_CASE(5) SSDE_CASE(6) SSDE_CASE(7)
        SSDE_CASE(8) SSDE_CASE(9) SSDE_CASE(10) SSDE_CASE(11) SSDE_CASE(12) SSDE_CASE(13) SSDE_CASE(14) SSDE_CASE(15)
#undef SSDE_CASE
        default: break;
    }
}

template <int MODEL, int D>
static void launch_one_mask(const IsoArgs& a, dim3 grid, hipStream_t s) {
    dim3 block(WG_WAVES * WAVE);
    const bool uni = a.uniform_dt != 0;
    switch (a.part_mask[0]) {
#define SSDE_CASE(M)                                                                                          \
    case M:                                                                                                   \
        if (uni && a.quiet_w > 0 && a.nan_bits) hipLaunchKernelGGL((iso_quiet_kernel<MODEL, D, M>), grid, block, 0, s, a); \
        else if (uni) hipLaunchKernelGGL((iso_mask_kernel<MODEL, D, M, true>), grid, block, 0, s, a);         \
        else hipLaunchKernelGGL((iso_mask_kernel<MODEL, D, M, false>), grid, block, 0, s, a);                 \
        break;
        SSDE_CASE(0) SSDE_CASE(1) SSDE_CASE(2) SSDE_CASE(3) SSDE_CASE(4) SSDE_CASE(5) SSDE_CASE(6) SSDE_CASE(7)
        SSDE_CASE(8) SSDE_CASE(9) SSDE_CASE(10) SSDE_CASE(11) SSDE_CASE(12) SSDE_CASE(13) SSDE_CASE(14) SSDE_CASE(15)
#undef SSDE_CASE
        default: break;
    }
}

// Hand-over check: for every (part, group, window boundary) compare the state the previous
// window arrived with against the state the next window warmed up to, component by component.
// Per component the error is max over the wave's lanes of |a - b| and the scale is the max over
// the lanes of max(|a|, |b|); chk[] gets the largest error/scale ratio of the workgroup.
// (g, c, part): the boundary between windows c and c + 1.  Returns the workgroup's largest error / scale
// ratio in thread 0.  Groups handled by the shared-covariance kernels dump the compact layout.
__device__ __forceinline__ double window_check_block(const IsoArgs& A, int nstate_full, int g, int c, int part, double* sh) {
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const TileView& tv = A.tv;
    const int L = tv.group_len[g];
    const int ns = tv.lane_nsteps[g * WAVE + lane];
    // (all_clean: the compact layout everywhere -- known from the arguments, so that the dump loads below do not wait for a flag)
    const int nstate = A.all_clean ? A.nstate_clean : (A.nstate_clean > 0 && (A.group_flags[g] & 1)) ? A.nstate_clean : nstate_full;
    // the group's window plan: the launch's, or -- mixed batch, group on the general kernel -- that launch's own
    int nc = A.n_chunks, win = A.window, t0 = A.t0, t0d = A.t0_delta;
    if (A.dual && !(A.group_flags[g] & 1)) { nc = A.n_chunks_d; win = A.window_d; t0 = A.t0_d; t0d = A.t0_delta_d; }
    if (c + 1 >= nc) return 0.0;                       // (workgroup-uniform) no such boundary in this group's plan
    int sb_, s_next, se_;
    window_bounds(L, nc, win, t0, c + 1, sb_, s_next, se_, t0d);   // s_next = first scored row of window c+1
    const bool valid = (ns > s_next) && (s_next < L);      // (applied AFTER the loads: their addresses do not depend on it, and
    const int pc0 = part * nc + c, pc1 = pc0 + 1;          //  one memory round trip then covers the lengths and the dumps)
    const double* out_c = A.bnd + (((int64_t)pc0 * tv.n_groups + g) * 2 + 1) * A.bnd_stride * WAVE + lane;
    const double* in_n = A.bnd + (((int64_t)pc1 * tv.n_groups + g) * 2 + 0) * A.bnd_stride * WAVE + lane;
    double worst = 0.0;
    // the block's 4 waves take every 4th component; all of a wave's loads (at most NSTATE_MAX / 4 pairs per pass) are issued
    // before the first reduction, so that one HBM round trip covers them instead of one per component; dumps wider than
    // NSTATE_MAX components (drift columns, k_iso_drift.hip) take further passes
    constexpr int KMAX = NSTATE_MAX / 4;
    for (int k0 = 0; k0 < nstate; k0 += NSTATE_MAX) {
        double av[KMAX], bv[KMAX];
#pragma unroll
        for (int i = 0; i < KMAX; i++) {
            const int k = k0 + wv + 4 * i;
            const bool on = k < nstate;
            av[i] = on ? out_c[k * WAVE] : 0.0;
            bv[i] = on ? in_n[k * WAVE] : 0.0;
        }
#pragma unroll
        for (int i = 0; i < KMAX; i++) {          // a lane without that row: nothing to compare (what it loaded may be stale)
            av[i] = valid ? av[i] : 0.0;
            bv[i] = valid ? bv[i] : 0.0;
        }
#pragma unroll
        for (int i = 0; i < KMAX; i++) {
            if (k0 + wv + 4 * i >= nstate) break;             // (wave-uniform)
            double err = fabs(av[i] - bv[i]), sc = fmax(fabs(av[i]), fabs(bv[i]));
            if (valid && !(err == err)) err = INFINITY;  // NaN on either side must not pass
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) {
                err = fmax(err, __shfl_xor(err, o, 64));
                sc = fmax(sc, __shfl_xor(sc, o, 64));
            }
            if (err > 0.0) worst = fmax(worst, err / sc);
        }
    }
    if (lane == 0) sh[wv] = worst;
    __syncthreads();
    return fmax(fmax(sh[0], sh[1]), fmax(sh[2], sh[3]));
}

__global__ __launch_bounds__(4 * WAVE) void window_check_kernel(const IsoArgs A, int nstate, double* chk) {
    __shared__ double sh[4];
    const int g = blockIdx.x, c = blockIdx.y, part = blockIdx.z;
    const double w = window_check_block(A, nstate, g, c, part, sh);
    if (threadIdx.x == 0) chk[((int64_t)part * (A.n_chunks - 1) + c) * A.tv.n_groups + g] = w;
}

int iso_nstate(int model, int d) { return model == M_CTCRW ? 4 * (2 * d + 3) + 2 * d : 4 * (d + 1) + d; }

// a.group_mode: 0 = general kernel on every group; 3 = the shared-covariance kernels
// (launch_iso_shared, issued by the engine on its side streams) own the NaN-free groups and this
// launch runs the general kernel on the others (only when `any_dirty`)
hipError_t launch_iso(int model, int d, const IsoArgs& a0, bool any_dirty, hipStream_t s) {
    const int g8 = ((a0.use_group_list ? a0.n_dirty_groups : a0.tv.n_groups) + 7) / 8;
    dim3 grid((g8 * 8 * a0.n_parts * a0.n_chunks + WG_WAVES - 1) / WG_WAVES), block(WG_WAVES * WAVE);
    if (grid.x == 0) return hipSuccess;
    bool done = false;
    IsoArgs a = a0;
    const bool shared = a0.group_mode == 3;
#define SSDE_LAUNCH(MODEL, D)                                                          \
    if (model == MODEL && d == D) {                                                    \
        if (shared) a.group_mode = 1;                                                  \
        if (!shared || any_dirty) {                                                    \
            if (a.n_parts == 1) launch_one_mask<MODEL, D>(a, grid, s);                 \
            else hipLaunchKernelGGL((iso_kernel<MODEL, D>), grid, block, 0, s, a);     \
        }                                                                              \
        done = true;                                                                   \
    }
    SSDE_LAUNCH(M_CTCRW, 1) SSDE_LAUNCH(M_CTCRW, 2)
    SSDE_LAUNCH(M_OU_SSM, 1) SSDE_LAUNCH(M_OU_SSM, 2)
    SSDE_LAUNCH(M_BM_SSM, 1) SSDE_LAUNCH(M_BM_SSM, 2)
#undef SSDE_LAUNCH
    if (!done) return hipErrorInvalidValue;
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    return e;
}

// One launch for everything after the main kernel(s): workgroups [0, n_check) run one hand-over check each and
// raise out[n_out] (a non-negative double orders like its bit pattern; the main kernel zeroed it), the others
// produce one output slot each (reduce_slot, k_reduce.hip).
__global__ __launch_bounds__(256) void iso_finalize_kernel(const IsoArgs A, const ReduceArgs R, int nstate, int n_check) {
    __shared__ double sh[256];
    if ((int)blockIdx.x < n_check) {
        const int G = A.tv.n_groups;
        int g, c, part = 0;
        if (A.dual) {
            // two plans (n_parts == 1): every group's boundaries under the shared launch's plan first -- the general launch's groups
            // have none there and leave --, then the general launch's groups under theirs, through the list of those groups
            const int nA = G * (A.n_chunks - 1);
            if ((int)blockIdx.x < nA) {
                g = blockIdx.x % G; c = blockIdx.x / G;
                if (!(A.group_flags[g] & 1)) { publish_if_last(R, 0ull); return; }
            } else {
                const int k = blockIdx.x - nA;
                g = A.dirty_groups[k % A.n_dirty_groups]; c = k / A.n_dirty_groups;
            }
        } else {
            const int nb = A.n_chunks - 1;
            g = blockIdx.x % G; c = (blockIdx.x / G) % nb; part = blockIdx.x / (G * nb);
        }
        const double w = window_check_block(A, nstate, g, c, part, sh);
        unsigned long long dep = 0ull;
        if (threadIdx.x == 0 && w > 0.0)
            dep = atomicMax((unsigned long long*)A.chk_out, (unsigned long long)__double_as_longlong(w == w ? w : INFINITY));
        publish_if_last(R, dep);
    } else {
        unsigned long long dep = 0ull;
        if ((int)blockIdx.x == n_check && threadIdx.x == 0 && A.quiet_flag) {
            // what the switches to quiet rows found (k_iso.hip: run_lane), into the evaluation's check value
            const unsigned long long v = atomicExch((unsigned long long*)A.quiet_flag, 0ull);
            if (v) dep = atomicMax((unsigned long long*)A.chk_out, v);
        }
        publish_if_last(R, reduce_slot(R, blockIdx.x - n_check, sh) ^ dep);      // (the count depends on BOTH atomics' return values)
    }
}

hipError_t launch_iso_finalize(int model, int d, const IsoArgs& a, const ReduceArgs& r, hipStream_t s) {
    const int n_check = a.dual ? a.tv.n_groups * (a.n_chunks - 1) + a.n_dirty_groups * (a.n_chunks_d - 1)
                               : (a.n_chunks > 1 ? a.tv.n_groups * (a.n_chunks - 1) * a.n_parts : 0);
    ReduceArgs rr = r;
    rr.pub_blocks = n_check + r.n_out;
    hipLaunchKernelGGL(iso_finalize_kernel, dim3(n_check + r.n_out), dim3(256), 0, s, a, rr, iso_nstate(model, d), n_check);
    return hipGetLastError();
}

// ---- which blocks of 8 rows hold a missing observation (ssde_create, once; IsoArgs.nan_bits) ------------------------
// One wave per (group, 64 blocks): lane = track, bit b of the word = some lane's row in block b is not a number in some
// response column (rows past a lane's last step are padding and do not count).
__global__ __launch_bounds__(WAVE) void nan_blocks_kernel(const TileView tv, int d, int U, unsigned long long* bits, int nwords) {
    const int g = blockIdx.x, w = blockIdx.y, lane = threadIdx.x;
    const int ns = tv.lane_nsteps[g * WAVE + lane];
    const int L = tv.group_len[g];
    const double* base = tv.tiles + tv.group_off[g] + lane;
    unsigned long long word = 0ull;
    for (int bb = 0; bb < 64; bb++) {
        const int s0 = (w * 64 + bb) * U;
        if (s0 >= L) break;                                   // (workgroup-uniform)
        bool bad = false;
        for (int u = 0; u < U; u++) {
            const int s = s0 + u;
            if (s < ns && s < L)
                for (int a = 0; a < d; a++) { const double v = base[((int64_t)s * tv.C + tv.c_obs + a) * WAVE]; bad = bad || (v != v); }
        }
        if (__ballot(bad) != 0ull) word |= 1ull << bb;
    }
    if (lane == 0) bits[(int64_t)g * nwords + w] = word;
}
int iso_block_rows(int model) { return model == M_CTCRW ? quiet_u<M_CTCRW>() : quiet_u<M_OU_SSM>(); }
hipError_t launch_nan_blocks(const TileView& tv, int d, int block_rows, unsigned long long* bits, int nwords, hipStream_t s) {
    if (tv.n_groups == 0 || nwords == 0) return hipSuccess;
    hipLaunchKernelGGL(nan_blocks_kernel, dim3(tv.n_groups, nwords), dim3(WAVE), 0, s, tv, d, block_rows, bits, nwords);
    return hipGetLastError();
}

hipError_t launch_window_check(int model, int d, const IsoArgs& a, hipStream_t s) {
    if (a.n_chunks <= 1 || a.tv.n_groups == 0) return hipSuccess;
    hipLaunchKernelGGL(window_check_kernel, dim3(a.tv.n_groups, a.n_chunks - 1, a.n_parts), dim3(4 * WAVE), 0, s, a,
                       iso_nstate(model, d), a.chk);
    return hipGetLastError();
}

}  // namespace ssde
