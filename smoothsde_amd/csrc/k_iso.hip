// k_iso.hip -- constant-coefficient isotropic Kalman kernels for gfx950 (CTCRW, OU_SSM, BM_SSM).
//
// One wavefront lane = one track; a workgroup is ONE wave (64 lanes), so that the ~10^2..10^3
// waves of a 10^4-track batch spread over all 256 CUs.  State, covariance and all forward
// sensitivities stay in VGPRs for the whole track; observations stream from the tiled HBM
// layout (ssde_device.hpp) with coalesced 512-B wave loads, prefetched one TILE_U-step block
// ahead into registers (a stream consumed by a single wave gains nothing from an LDS round
// trip; the LDS-shared variant for direction-split workgroups is k_iso_lds.hip).
// fp64 throughout, no MFMA: the recursion is a serial chain of scalar fp64 FMAs per lane.
//
// Reference arithmetic: ssde_math.hpp (nllk_ctcrw.hpp:195-247, nllk_ou_ssm.hpp:163-213,
// nllk_bm_ssm.hpp:127-175).  Gradient: forward sensitivities selected by a DIR_* mask; the
// grid may be split into "parts" that each carry a subset of the directions (more waves for
// the same batch) -- parts of one track group are placed on one XCD so the group's tile is
// fetched from HBM once and re-served from that XCD's L2.
#include "ssde_device.hpp"

namespace ssde {

template <int C>
__device__ __forceinline__ void load_block(double (&dst)[TILE_U][C], const double* p) {
#pragma unroll
    for (int u = 0; u < TILE_U; u++)
#pragma unroll
        for (int c = 0; c < C; c++) dst[u][c] = p[(u * C + c) * WAVE];
}

template <int D, int MASK>
__device__ __forceinline__ void run_ctcrw(const IsoArgs& A, int g, int part) {
    constexpr int C = 1 + D;
    constexpr int NACC = 4 + D;
    const int lane = threadIdx.x;
    const TileView& tv = A.tv;
    const double* base = tv.tiles + tv.group_off[g] + lane;
    const int L = tv.group_len[g];
    const int ns = tv.lane_nsteps[g * WAVE + lane];

    CtcrwLane<D, MASK> S;
    double a0[2 * D];
#pragma unroll
    for (int c = 0; c < 2 * D; c++) a0[c] = tv.a0[((int64_t)g * 2 * D + c) * WAVE + lane];
    S.init(a0, A.p0[0], A.p0[1], A.p0[2]);
    double mu[D];
#pragma unroll
    for (int a = 0; a < D; a++) mu[a] = A.mu[a];
    CtcrwTrans tr = A.ctr;
    const bool uni = A.uniform_dt != 0;

    double cur[TILE_U][C], nxt[TILE_U][C];
    load_block<C>(cur, base);
    for (int s0 = 0; s0 < L; s0 += TILE_U) {
        load_block<C>(nxt, base + (int64_t)(s0 + TILE_U) * C * WAVE);  // spare block keeps this in bounds
#pragma unroll
        for (int u = 0; u < TILE_U; u++) {
            if (s0 + u < ns) {
                if (!uni) ctcrw_trans(cur[u][0], A.tau, A.beta, A.sigma, tr);
                ctcrw_step<D, MASK>(S, tr, A.h, mu, &cur[u][1], is_na(cur[u][1], A.any_nan));
            }
        }
#pragma unroll
        for (int u = 0; u < TILE_U; u++)
#pragma unroll
            for (int c = 0; c < C; c++) cur[u][c] = nxt[u][c];
    }
    double out[NACC];
    ctcrw_finish<D, MASK>(S, out);
#pragma unroll
    for (int k = 0; k < NACC; k++) {
        const double t = wave_sum(out[k]);
        if (lane == 0) A.partials[((int64_t)part * NACC + k) * tv.n_groups + g] = t;
    }
}

template <int MODEL, int D, int MASK>
__device__ __forceinline__ void run_scal(const IsoArgs& A, int g, int part) {
    constexpr int C = 1 + D;
    constexpr int NACC = 4 + D;
    const int lane = threadIdx.x;
    const TileView& tv = A.tv;
    const double* base = tv.tiles + tv.group_off[g] + lane;
    const int L = tv.group_len[g];
    const int ns = tv.lane_nsteps[g * WAVE + lane];

    ScalLane<D, MASK> S;
    double a0[D];
#pragma unroll
    for (int c = 0; c < D; c++) a0[c] = tv.a0[((int64_t)g * D + c) * WAVE + lane];
    S.init(a0, A.p0[0]);
    double mu[D];
#pragma unroll
    for (int a = 0; a < D; a++) mu[a] = A.mu[a];
    ScalTrans tr = A.str;
    const bool uni = A.uniform_dt != 0;

    double cur[TILE_U][C], nxt[TILE_U][C];
    load_block<C>(cur, base);
    for (int s0 = 0; s0 < L; s0 += TILE_U) {
        load_block<C>(nxt, base + (int64_t)(s0 + TILE_U) * C * WAVE);
#pragma unroll
        for (int u = 0; u < TILE_U; u++) {
            if (s0 + u < ns) {
                if (!uni) {
                    if (MODEL == M_OU_SSM) ou_trans(cur[u][0], A.tau, A.sigma, tr);
                    else bm_trans(cur[u][0], A.sigma, tr);
                }
                scal_step<D, MASK, MODEL == M_OU_SSM>(S, tr, A.h, mu, &cur[u][1], is_na(cur[u][1], A.any_nan));
            }
        }
#pragma unroll
        for (int u = 0; u < TILE_U; u++)
#pragma unroll
            for (int c = 0; c < C; c++) cur[u][c] = nxt[u][c];
    }
    double out[NACC];
    scal_finish<D, MASK>(S, out);
#pragma unroll
    for (int k = 0; k < NACC; k++) {
        const double t = wave_sum(out[k]);
        if (lane == 0) A.partials[((int64_t)part * NACC + k) * tv.n_groups + g] = t;
    }
}

template <int MODEL, int D, int MASK>
__device__ __forceinline__ void run_any(const IsoArgs& A, int g, int part) {
    if (MODEL == M_CTCRW) run_ctcrw<D, MASK>(A, g, part);
    else run_scal<MODEL, D, MASK>(A, g, part);
}

// Workgroup id -> (track group, part).  Workgroups are dealt round-robin over the 8 XCDs, so
// ids that are equal mod 8 share an XCD (and its L2): all parts of one group get such ids.
__device__ __forceinline__ bool decode_block(const IsoArgs& A, int& g, int& part) {
    const int id = blockIdx.x;
    const int np = A.n_parts;
    g = (id / (8 * np)) * 8 + (id & 7);
    part = (id >> 3) % np;
    return g < A.tv.n_groups;
}

template <int MODEL, int D>
__global__ __launch_bounds__(WAVE) void iso_kernel(const IsoArgs A) {
    int g, part;
    if (!decode_block(A, g, part)) return;
    // (no dynamic indexing into the by-value argument block: that would force a scratch copy)
    const int mask = part == 0 ? A.part_mask[0] : part == 1 ? A.part_mask[1] : part == 2 ? A.part_mask[2] : A.part_mask[3];
    switch (mask) {
#define SSDE_CASE(M) case M: run_any<MODEL, D, M>(A, g, part); break;
        SSDE_CASE(0) SSDE_CASE(1) SSDE_CASE(2) SSDE_CASE(3) SSDE_CASE(4) SSDE_CASE(5) SSDE_CASE(6) SSDE_CASE(7)
        SSDE_CASE(8) SSDE_CASE(9) SSDE_CASE(10) SSDE_CASE(11) SSDE_CASE(12) SSDE_CASE(13) SSDE_CASE(14) SSDE_CASE(15)
#undef SSDE_CASE
        default: break;
    }
}

hipError_t launch_iso(int model, int d, const IsoArgs& a, hipStream_t s) {
    const int g8 = (a.tv.n_groups + 7) / 8;
    dim3 grid(g8 * 8 * a.n_parts), block(WAVE);
    if (grid.x == 0) return hipSuccess;
#define SSDE_LAUNCH(MODEL, D)                                                          \
    if (model == MODEL && d == D) {                                                    \
        hipLaunchKernelGGL((iso_kernel<MODEL, D>), grid, block, 0, s, a);              \
        return hipGetLastError();                                                      \
    }
    SSDE_LAUNCH(M_CTCRW, 1) SSDE_LAUNCH(M_CTCRW, 2)
    SSDE_LAUNCH(M_OU_SSM, 1) SSDE_LAUNCH(M_OU_SSM, 2)
    SSDE_LAUNCH(M_BM_SSM, 1) SSDE_LAUNCH(M_BM_SSM, 2)
#undef SSDE_LAUNCH
    return hipErrorInvalidValue;
}

}  // namespace ssde
