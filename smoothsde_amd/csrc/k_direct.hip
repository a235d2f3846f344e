// k_direct.hip -- direct transition-density families "BM", "BM_t", "OU" and "CIR" (nllk_sde.hpp:77-84 with
// tr_dens.hpp:32-52) for gfx950.
//
// No recursion: row i depends only on rows i-1, i and on the parameters of row i-1 (Q6), so
// the long format is streamed as it is -- lane = row, every column read is coalesced
// (the design blocks are column-major n x K exactly as R hands them over).  The linear
// predictor, the density, its derivative w.r.t. the row's SDE parameters and the
// X' g accumulation are fused in one pass: the design row is read once (88 B/row in the C3
// configuration: obs 8 + time 8 + 9 columns 72).  HBM-bound by design; fp64; no MFMA, no LDS
// staging (each byte is used once by one lane).
//
// Everything that does not depend on the row is hoisted out of the loop: column pointers,
// coefficients and slot->parameter routing live in scalar registers, intercept-only parameters
// are folded into per-parameter constants, and exp() of a log-scale parameter is evaluated once
// when that parameter has no streamed column.
#include <algorithm>

#include "ssde_device.hpp"

namespace ssde {

template <int MODEL, int D, int KMAX>
__global__ __launch_bounds__(256) void direct_kernel(const DirectArgs A) {
    const SlotTable* __restrict__ T = A.slots;
    const int ns = A.n_slots;
    constexpr int Q = (MODEL == M_BM || MODEL == M_BM_T) ? D + 1 : D + 2;

    // ---- loop invariants (wave-uniform) ---------------------------------------------------------
    const double* cptr[KMAX];
    double coef[KMAX];
    int pj[KMAX];
    int dk[KMAX];                                // decay-rate index of a decaying column, else -1
    double rho[MAX_DECAY] = {0.0, 0.0, 0.0, 0.0};
    const bool any_decay = A.n_decay > 0;
    if (any_decay) {
#pragma unroll
        for (int k = 0; k < MAX_DECAY; k++) rho[k] = (k < A.n_decay) ? exp(A.par[A.off_decay + k]) : 0.0;   // nllk_sde.hpp:47
    }
    double base[MAX_Q] = {0.0, 0.0, 0.0, 0.0};   // intercept part of every parameter
    bool varies[MAX_Q] = {false, false, false, false};
#pragma unroll
    for (int k = 0; k < KMAX; k++) {
        cptr[k] = nullptr; coef[k] = 0.0; pj[k] = -1; dk[k] = -1;
        if (k < ns) {
            const int c = T->col[k];
            pj[k] = T->par_j[k];
            dk[k] = any_decay ? T->decay[k] : -1;
            coef[k] = A.par[T->pidx[k]];
            if (c >= 0) {
                cptr[k] = A.cols[c];
#pragma unroll
                for (int j = 0; j < Q; j++) varies[j] = varies[j] || (pj[k] == j);
            } else {
#pragma unroll
                for (int j = 0; j < Q; j++) base[j] += (pj[k] == j) ? coef[k] : 0.0;
            }
        }
    }
    // natural-scale value of a log-scale parameter that has no streamed column
    const double nat_p1 = exp(base[D]);
    const double nat_p2 = (Q > D + 1) ? exp(base[D + 1]) : 0.0;

    double acc[KMAX];
#pragma unroll
    for (int k = 0; k < KMAX; k++) acc[k] = 0.0;
    double accd[MAX_DECAY] = {0.0, 0.0, 0.0, 0.0};   // d nll / d log_decay
    double nll = 0.0;

    // every workgroup streams ONE contiguous range of rows (consecutive 2-KB pieces of each column)
    const int64_t per_block = ((A.n + gridDim.x - 1) / gridDim.x + 255) / 256 * 256;
    const int64_t row_lo = (int64_t)blockIdx.x * per_block;
    const int64_t row_hi = row_lo + per_block < A.n ? row_lo + per_block : A.n;
    for (int64_t i = row_lo + threadIdx.x; i < row_hi; i += 256) {
        if (!((A.scored[i >> 5] >> (i & 31)) & 1u)) continue;
        const double dt = A.times[i] - A.times[i - 1];  // dtimes(i-1), nllk_sde.hpp:37,80
        double w[KMAX], dfac[KMAX];
        double par[MAX_Q] = {base[0], base[1], base[2], base[3]};
#pragma unroll
        for (int k = 0; k < KMAX; k++) {
            w[k] = 1.0; dfac[k] = 0.0;
            if (cptr[k] != nullptr) {                    // uniform: a scalar branch, not a per-lane select
                w[k] = cptr[k][i - 1];
                if (any_decay && dk[k] >= 0) {           // X_re_copy.col = X_re.col * exp(-rate * t_decay), row i-1 (Q6)
                    const double r = (dk[k] == 0) ? rho[0] : (dk[k] == 1) ? rho[1] : (dk[k] == 2) ? rho[2] : rho[3];
                    const double rt = r * A.t_decay[(int64_t)pj[k] * A.n + (i - 1)];
                    w[k] *= exp(-rt);
                    dfac[k] = -rt;                       // d w / d log_decay = w * (-rate * t)
                }
                const double t = w[k] * coef[k];
                if (pj[k] == 0) par[0] += t;
                else if (pj[k] == 1) par[1] += t;
                else if (pj[k] == 2) par[2] += t;
                else par[3] += t;
            }
        }
        double g[MAX_Q] = {0.0, 0.0, 0.0, 0.0};
        if (MODEL == M_BM_T) {
            // tr_dens.hpp:38-44: Student-t increment, scale = sd / sqrt(df / (df - 2))
            const double z0 = A.obs[i - 1], z1 = A.obs[i];
            if (!(is_na(z0, A.any_nan) || is_na(z1, A.any_nan)))
                nll += bmt_direct(z0, z1, dt, par[0], varies[1] ? par[1] : base[1], A.tdf, A.tconst, g[0], g[1]);
        } else if (MODEL == M_CIR) {
            // tr_dens.hpp:53-67: par = (log mu_a, log beta, log sigma), non-central chi-square transition
#pragma unroll
            for (int a = 0; a < D; a++) {
                const double z0 = A.obs[(i - 1) + (int64_t)a * A.n], z1 = A.obs[i + (int64_t)a * A.n];
                if (is_na(z0, A.any_nan) || is_na(z1, A.any_nan)) continue;
                nll += cir_direct(z0, z1, dt, par[a], par[D], par[D + 1], g[a], g[D], g[D + 1]);
            }
        } else if (MODEL == M_BM) {
            // tr_dens.hpp:35-37: mean = z0 + mu dt, sd = exp(par[D]) sqrt(dt)
            const double sig = varies[D] ? exp(par[D]) : nat_p1;
            const double sd = sig * sqrt(dt);
            const double lsd = log(sd), isd = 1.0 / sd;
#pragma unroll
            for (int a = 0; a < D; a++) {
                const double z0 = A.obs[(i - 1) + (int64_t)a * A.n], z1 = A.obs[i + (int64_t)a * A.n];
                if (is_na(z0, A.any_nan) || is_na(z1, A.any_nan)) continue;  // tr_dens.hpp:31
                const double r = (z1 - (z0 + par[a] * dt)) * isd;
                g[a] += -r * dt * isd;
                g[D] += 1.0 - r * r;
                nll += SSDE_LOG_SQRT_2PI + lsd + 0.5 * r * r;
            }
        } else {
            // tr_dens.hpp:49-52: mean = mu + e^{-dt/tau} (z0 - mu), sd = sqrt(kappa (1 - e^{-2 dt/tau}))
            const double tau = varies[D] ? exp(par[D]) : nat_p1;
            const double kap = varies[D + 1] ? exp(par[D + 1]) : nat_p2;
            const double z = dt / tau;
            const double e = exp(-z);
            const double e2 = e * e;
            const double ome2 = 1.0 - e2;
            const double sd = sqrt(kap * ome2);
            const double lsd = log(sd), isd = 1.0 / sd;
            const double dls_lt = -e2 * z / ome2;
#pragma unroll
            for (int a = 0; a < D; a++) {
                const double z0 = A.obs[(i - 1) + (int64_t)a * A.n], z1 = A.obs[i + (int64_t)a * A.n];
                if (is_na(z0, A.any_nan) || is_na(z1, A.any_nan)) continue;
                const double mu = par[a];
                const double r = (z1 - (mu + e * (z0 - mu))) * isd;
                g[a] += -r * (1.0 - e) * isd;
                g[D] += -r * (e * z * (z0 - mu)) * isd + (1.0 - r * r) * dls_lt;
                g[D + 1] += 0.5 * (1.0 - r * r);
                nll += SSDE_LOG_SQRT_2PI + lsd + 0.5 * r * r;
            }
        }
#pragma unroll
        for (int k = 0; k < KMAX; k++) {
            if (pj[k] >= 0) {                            // uniform
                const double gj = (pj[k] == 0) ? g[0] : (pj[k] == 1) ? g[1] : (pj[k] == 2) ? g[2] : g[3];
                acc[k] += w[k] * gj;
                if (any_decay && dk[k] >= 0) {
                    const double t = gj * coef[k] * w[k] * dfac[k];
#pragma unroll
                    for (int q = 0; q < MAX_DECAY; q++) accd[q] += (dk[k] == q) ? t : 0.0;
                }
            }
        }
    }

    // workgroup reduction: wave shuffles, then 4 wave totals through LDS
    // accumulator order: [nll | n_slots coefficients | n_decay decay rates]
    __shared__ double sh[4][KMAX + 1 + MAX_DECAY];
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
    if (any_decay) {
#pragma unroll
        for (int q = 0; q < MAX_DECAY; q++) {
            t = wave_sum(accd[q]);
            if (lane == 0 && q < A.n_decay) sh[wv][1 + ns + q] = t;
        }
    }
    __syncthreads();
    if ((int)threadIdx.x <= ns + A.n_decay) {
        const int k = threadIdx.x;
        A.partials[(int64_t)k * A.n_blocks + blockIdx.x] = (sh[0][k] + sh[1][k]) + (sh[2][k] + sh[3][k]);
    }
}

// ---- design blocks given as piecewise-cubic functions of a covariate (ssde_ppbasis) ------------------------------
__device__ __forceinline__ int pp_interval(const PPRef& P, const double* knots, double x) {
    int iv;
    if (P.uniform) {
        iv = (int)floor((x - P.k0) * P.inv_h);
        iv = iv < 0 ? 0 : (iv > P.nk - 2 ? P.nk - 2 : iv);
        // the multiplication may land one interval off at a breakpoint: settle with the knots themselves
        if (iv > 0 && x < knots[iv]) iv--;
        else if (iv < P.nk - 2 && x >= knots[iv + 1]) iv++;
    } else {
        iv = 0;
        for (int k = 1; k < P.nk - 1; k++) iv += (x >= knots[k]) ? 1 : 0;
    }
    return iv;
}

// one-off: the dense n x K block a table stands for (generic kernels, Kalman families)
__global__ __launch_bounds__(256) void pp_materialise_kernel(const PPRef P, int K, int64_t n, double* dst, int64_t stride) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const double x = P.x[i];
        const int iv = pp_interval(P, P.knots, x);
        const double t = x - P.knots[iv];
        for (int c = 0; c < K; c++) {
            const double* q = P.tab + ((int64_t)iv * K + c) * 4;
            dst[(int64_t)c * stride + i] = fma(fma(fma(q[3], t, q[2]), t, q[1]), t, q[0]);
        }
    }
}
hipError_t launch_pp_materialise(const PPRef& pp, int K, int64_t n, double* dst, int64_t stride, hipStream_t s) {
    hipLaunchKernelGGL(pp_materialise_kernel, dim3((unsigned)std::min<int64_t>((n + 255) / 256, 4096)), dim3(256), 0, s, pp, K, n, dst, stride);
    return hipGetLastError();
}

// ---- fast kernel -------------------------------------------------------------------------------------
// KA / KB: register capacity for the streamed columns of the (at most two) parameters that have any.
// PPA: group A is KNOWN to be table-backed (and there is no group B): a kernel of its own, without the registers of the
// streamed-column look-ahead, which it spends on a deeper ring of its three-doubles-per-row inputs instead -- the
// table variant is bound by occupancy x round-trip latency (167 VGPRs = three waves per SIMD, one row ahead: ~4.7 MB in
// flight chip-wide where 6 TB/s x 1.5 us want ~9 MB), not by its arithmetic.
#ifndef SSDE_PP_PF
#define SSDE_PP_PF 2          // rows of look-ahead per thread in the table-backed kernel
#endif
template <int MODEL, int D, int KA, int KB, bool PPA>
__global__ __launch_bounds__(256) void direct_fast_kernel(const DirectFastArgs A) {
    constexpr int Q = (MODEL == M_BM || MODEL == M_BM_T) ? D + 1 : D + 2;
    const int ncA = A.ncA, ncB = A.ncB, ja = A.ja, jb = A.jb;
    const int64_t n = A.n;
    double accI[MAX_Q] = {0.0, 0.0, 0.0, 0.0};
    double accA[KA > 0 ? KA : 1], accB[KB > 0 ? KB : 1];
#pragma unroll
    for (int c = 0; c < KA; c++) accA[c] = 0.0;
#pragma unroll
    for (int c = 0; c < KB; c++) accB[c] = 0.0;
    double nll = 0.0;
    const bool varies1 = (ja == D) || (jb == D), varies2 = (ja == D + 1) || (jb == D + 1);
    // groups given as basis tables: coefficients and knots into LDS once per workgroup
    __shared__ double ldsA[(KA > 0) ? PP_LDS : 1], ldsB[(KB > 0) ? PP_LDS : 1];
    const bool ppa = PPA || (KA > 0 && A.ppA.x != nullptr), ppb = KB > 0 && A.ppB.x != nullptr;
    if (ppa) {
        const int nt = (A.ppA.nk - 1) * ncA * 4;
        for (int k = threadIdx.x; k < nt + A.ppA.nk; k += 256) ldsA[k] = k < nt ? A.ppA.tab[k] : A.ppA.knots[k - nt];
    }
    if (ppb) {
        const int nt = (A.ppB.nk - 1) * ncB * 4;
        for (int k = threadIdx.x; k < nt + A.ppB.nk; k += 256) ldsB[k] = k < nt ? A.ppB.tab[k] : A.ppB.knots[k - nt];
    }
    if (ppa || ppb) __syncthreads();
    // natural-scale values / derived constants of parameters that have no streamed column
    const double nat_p1 = exp(A.base[D]);
    const double nat_p2 = (Q > D + 1) ? exp(A.base[D + 1]) : 0.0;
    const double inv_p1 = 1.0 / nat_p1;
    // with a regular grid and constant scale parameters the whole transition kernel is row-independent
    const bool all_const = A.uniform_dt && !varies1 && !varies2;
    double c_e = 0.0, c_isd = 0.0, c_lsd = 0.0, c_dls = 0.0, c_z = 0.0;
    if (all_const) {
        const double dt = A.dt_uniform;
        if (MODEL == M_BM_T || MODEL == M_CIR) {
            // (nothing hoisted: the per-row log1p / Bessel series dominates anyway)
        } else if (MODEL == M_BM) {
            const double sd = nat_p1 * sqrt(dt);
            c_isd = 1.0 / sd; c_lsd = log(sd);
        } else {
            c_z = dt * inv_p1; c_e = exp(-c_z);
            const double ome2 = 1.0 - c_e * c_e, var = nat_p2 * ome2;
            c_isd = 1.0 / sqrt(var); c_lsd = 0.5 * log(var); c_dls = -c_e * c_e * c_z / ome2;
        }
    }

    const int64_t per_block = ((n + gridDim.x - 1) / gridDim.x + 255) / 256 * 256;
    const int64_t row_lo = (int64_t)blockIdx.x * per_block;
    const int64_t row_hi = row_lo + per_block < n ? row_lo + per_block : n;
    // The loop runs over the EARLIER row r of each transition (r -> r+1): the design columns, which are
    // read at the earlier row (Q6), are then fetched at 512-B-aligned wave addresses.  Streams are read
    // once: non-temporal loads.
    // Inputs of the NEXT row of this thread (256 rows ahead) are requested before the current row is worked on: the
    // covariate of a basis-evaluated block heads a dependent chain (x -> interval -> LDS table -> predictor) and the
    // observations are only needed after it, so without the look-ahead every row exposed two HBM round trips.
    const int64_t nlast = n - 1;
    constexpr int PF = PPA ? SSDE_PP_PF : 1;               // rows of look-ahead per thread
    double zq0[PF][D], zq1[PF][D], xqA[PF], xqB[PF];
    // ... and so are the word of the scored mask that holds row r + 1 and, where dt is not hoisted, the two time stamps:
    // they used to be fetched inside the row (a dependent L2 / HBM round trip per row: 0.90 -> 0.71 ms on C3's table variant)
    uint32_t sq[PF];
    double tq0[PF], tq1[PF];
    double wqA[(KA > 0 && !PPA) ? KA : 1], wqB[KB > 0 ? KB : 1];   // streamed columns of the next row
#pragma unroll
    for (int k = 0; k < PF; k++) {
        const int64_t rr = row_lo + threadIdx.x + (int64_t)256 * k;
        const int64_t r0 = rr < nlast ? rr : nlast, i0 = r0 + 1 < nlast ? r0 + 1 : nlast;
#pragma unroll
        for (int a = 0; a < D; a++) { zq0[k][a] = __builtin_nontemporal_load(&A.obs[r0 + (int64_t)a * n]); zq1[k][a] = A.obs[i0 + (int64_t)a * n]; }
        xqA[k] = ppa ? __builtin_nontemporal_load(&A.ppA.x[r0]) : 0.0;
        xqB[k] = ppb ? __builtin_nontemporal_load(&A.ppB.x[r0]) : 0.0;
        sq[k] = A.scored[i0 >> 5];
        tq0[k] = all_const ? 0.0 : __builtin_nontemporal_load(&A.times[r0]);
        tq1[k] = all_const ? 0.0 : A.times[i0];
        if constexpr (!PPA) {
            if (k == 0 && !ppa) {
#pragma unroll
                for (int c = 0; c < KA; c++) wqA[c] = __builtin_nontemporal_load(&A.colA[(int64_t)(c < ncA ? c : ncA - 1) * A.col_stride + r0]);
            }
        }
        if (k == 0 && !ppb) {
#pragma unroll
            for (int c = 0; c < KB; c++) wqB[c] = __builtin_nontemporal_load(&A.colB[(int64_t)(c < ncB ? c : ncB - 1) * A.col_stride + r0]);
        }
    }
    for (int64_t rbase = row_lo + threadIdx.x; rbase < row_hi; rbase += (int64_t)256 * PF) {
#pragma unroll
    for (int kpf = 0; kpf < PF; kpf++) {
        const int64_t r = rbase + (int64_t)256 * kpf;
        if (r >= row_hi) continue;
        const int64_t i = r + 1;
        double zc0[D], zc1[D];
#pragma unroll
        for (int a = 0; a < D; a++) { zc0[a] = zq0[kpf][a]; zc1[a] = zq1[kpf][a]; }
        const double xcA = xqA[kpf], xcB = xqB[kpf];
        const uint32_t sc = sq[kpf];
        const double tc0 = tq0[kpf], tc1 = tq1[kpf];
        {
            const int64_t rq = r + (int64_t)256 * PF;
            const int64_t rn = rq < nlast ? rq : nlast, in = rn + 1 < nlast ? rn + 1 : nlast;
#pragma unroll
            for (int a = 0; a < D; a++) { zq0[kpf][a] = __builtin_nontemporal_load(&A.obs[rn + (int64_t)a * n]); zq1[kpf][a] = A.obs[in + (int64_t)a * n]; }
            if (ppa) xqA[kpf] = __builtin_nontemporal_load(&A.ppA.x[rn]);
            if (ppb) xqB[kpf] = __builtin_nontemporal_load(&A.ppB.x[rn]);
            sq[kpf] = A.scored[in >> 5];
            if (!all_const) { tq0[kpf] = __builtin_nontemporal_load(&A.times[rn]); tq1[kpf] = A.times[in]; }
        }
        // streamed columns of row i-1 (Q6): this row's values were requested one iteration ago, the next row's are
        // requested now.  ALL KA / KB register slots are loaded unconditionally (slots past the column count re-read
        // the last real column and carry a zero coefficient): a guard per slot would make hipcc branch around every
        // load and wait for it (one dependent HBM round trip per column; cdna_hip_programming.md section 5, trap (c)).
        double wA[KA > 0 ? KA : 1], wB[KB > 0 ? KB : 1];
        if constexpr (!PPA) if (!ppa) {
            const int64_t rn = r + 256 < nlast ? r + 256 : nlast;
#pragma unroll
            for (int c = 0; c < KA; c++) { wA[c] = wqA[c]; wqA[c] = __builtin_nontemporal_load(&A.colA[(int64_t)(c < ncA ? c : ncA - 1) * A.col_stride + rn]); }
        }
        if (!ppb) {
            const int64_t rn = r + 256 < nlast ? r + 256 : nlast;
#pragma unroll
            for (int c = 0; c < KB; c++) { wB[c] = wqB[c]; wqB[c] = __builtin_nontemporal_load(&A.colB[(int64_t)(c < ncB ? c : ncB - 1) * A.col_stride + rn]); }
        }
        if (i >= n || !((sc >> (i & 31)) & 1u)) continue;
        const double dt = all_const ? A.dt_uniform : tc1 - tc0;                                  // dtimes(i-1)
        // the two linear predictors the column groups feed
        double sumA = 0.0, sumB = 0.0;
        if (ppa) {          // uniform branch: the block is a function of one covariate, 8 B/row instead of 8 K
            const double xr = xcA;
            const double* kn = ldsA + (A.ppA.nk - 1) * ncA * 4;
            const int iv = pp_interval(A.ppA, kn, xr);
            const double t = xr - kn[iv];
#pragma unroll
            for (int c = 0; c < KA; c++) {
                const double* q = ldsA + (iv * ncA + (c < ncA ? c : ncA - 1)) * 4;
                wA[c] = fma(fma(fma(q[3], t, q[2]), t, q[1]), t, q[0]);
            }
        }
        if (ppb) {
            const double xr = xcB;
            const double* kn = ldsB + (A.ppB.nk - 1) * ncB * 4;
            const int iv = pp_interval(A.ppB, kn, xr);
            const double t = xr - kn[iv];
#pragma unroll
            for (int c = 0; c < KB; c++) {
                const double* q = ldsB + (iv * ncB + (c < ncB ? c : ncB - 1)) * 4;
                wB[c] = fma(fma(fma(q[3], t, q[2]), t, q[1]), t, q[0]);
            }
        }
#pragma unroll
        for (int c = 0; c < KA; c++) sumA = fma(wA[c], A.coefA[c], sumA);
#pragma unroll
        for (int c = 0; c < KB; c++) sumB = fma(wB[c], A.coefB[c], sumB);
        double par[MAX_Q];
#pragma unroll
        for (int j = 0; j < MAX_Q; j++) par[j] = A.base[j] + ((j == ja) ? sumA : 0.0) + ((j == jb) ? sumB : 0.0);

        double g[MAX_Q] = {0.0, 0.0, 0.0, 0.0};
        if (MODEL == M_BM_T) {
            const double z0 = zc0[0], z1 = zc1[0];
            if (!(is_na(z0, A.any_nan) || is_na(z1, A.any_nan)))                       // tr_dens.hpp:31
                nll += bmt_direct(z0, z1, dt, par[0], par[1], A.tdf, A.tconst, g[0], g[1]);   // :38-44
        } else if (MODEL == M_CIR) {
#pragma unroll
            for (int a = 0; a < D; a++) {
                const double z0 = zc0[a], z1 = zc1[a];
                if (is_na(z0, A.any_nan) || is_na(z1, A.any_nan)) continue;              // tr_dens.hpp:31
                nll += cir_direct(z0, z1, dt, par[a], par[D], par[D + 1], g[a], g[D], g[D + 1]);   // :53-67
            }
        } else if (MODEL == M_BM) {
            double isd = c_isd, lsd = c_lsd;
            if (!all_const) {
                const double sd = (varies1 ? exp(par[D]) : nat_p1) * sqrt(dt);   // tr_dens.hpp:36
                isd = rcp(sd); lsd = log(sd);
            }
#pragma unroll
            for (int a = 0; a < D; a++) {
                const double z0 = zc0[a], z1 = zc1[a];
                if (is_na(z0, A.any_nan) || is_na(z1, A.any_nan)) continue;          // tr_dens.hpp:31
                const double r = (z1 - (z0 + par[a] * dt)) * isd;                    // :35, :37
                g[a] += -r * dt * isd;
                g[D] += 1.0 - r * r;
                nll += SSDE_LOG_SQRT_2PI + lsd + 0.5 * r * r;
            }
        } else {
            double e = c_e, isd = c_isd, lsd = c_lsd, dls_lt = c_dls, z = c_z;
            if (!all_const) {
                z = dt * (varies1 ? rcp(exp(par[D])) : inv_p1);                       // dt / tau
                e = exp(-z);
                const double ome2 = 1.0 - e * e;
                const double var = (varies2 ? exp(par[D + 1]) : nat_p2) * ome2;      // tr_dens.hpp:50-51
                isd = rcp(sqrt(var)); lsd = 0.5 * log(var);
                dls_lt = -e * e * z * rcp(ome2);
            }
#pragma unroll
            for (int a = 0; a < D; a++) {
                const double z0 = zc0[a], z1 = zc1[a];
                if (is_na(z0, A.any_nan) || is_na(z1, A.any_nan)) continue;
                const double mu = par[a];
                const double r = (z1 - (mu + e * (z0 - mu))) * isd;                  // :49, :52
                g[a] += -r * (1.0 - e) * isd;
                g[D] += -r * (e * z * (z0 - mu)) * isd + (1.0 - r * r) * dls_lt;
                g[D + 1] += 0.5 * (1.0 - r * r);
                nll += SSDE_LOG_SQRT_2PI + lsd + 0.5 * r * r;
            }
        }
#pragma unroll
        for (int j = 0; j < MAX_Q; j++) accI[j] += g[j];
        const double gA = (ja == 0) ? g[0] : (ja == 1) ? g[1] : (ja == 2) ? g[2] : g[3];
        const double gB = (jb == 0) ? g[0] : (jb == 1) ? g[1] : (jb == 2) ? g[2] : g[3];
#pragma unroll
        for (int c = 0; c < KA; c++) accA[c] = fma(wA[c], gA, accA[c]);   // slots >= ncA are never read back
#pragma unroll
        for (int c = 0; c < KB; c++) accB[c] = fma(wB[c], gB, accB[c]);
    }
    }

    // workgroup reduction: accumulator order = [nll | Q intercept slots | ncA | ncB]
    constexpr int NMAX = 1 + MAX_Q + KA + KB;
    __shared__ double sh[4][NMAX];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    double t = wave_sum(nll);
    if (lane == 0) sh[wv][0] = t;
#pragma unroll
    for (int j = 0; j < MAX_Q; j++) { t = wave_sum(accI[j]); if (lane == 0) sh[wv][1 + j] = t; }
#pragma unroll
    for (int c = 0; c < KA; c++) { t = wave_sum(accA[c]); if (lane == 0) sh[wv][1 + MAX_Q + c] = t; }
#pragma unroll
    for (int c = 0; c < KB; c++) { t = wave_sum(accB[c]); if (lane == 0) sh[wv][1 + MAX_Q + KA + c] = t; }
    __syncthreads();
    const int nacc = 1 + MAX_Q + ncA + ncB;
    if ((int)threadIdx.x < nacc) {
        int k = threadIdx.x;   // output slot
        int src = k < 1 + MAX_Q + ncA ? k : 1 + MAX_Q + KA + (k - (1 + MAX_Q + ncA));
        A.partials[(int64_t)k * A.n_blocks + blockIdx.x] = (sh[0][src] + sh[1][src]) + (sh[2][src] + sh[3][src]);
    }
}

hipError_t launch_direct_fast(const DirectFastArgs& a, hipStream_t s) {
    dim3 grid(a.n_blocks), block(256);
#define SSDE_F(MODEL, D, KA, KB)                                                            \
    if (a.model == MODEL && a.d == D && a.ncA <= KA && a.ncB <= KB && (KA == 0 || a.ncA > 0) && (KB == 0 || a.ncB > 0)) { \
        if constexpr (KA > 0 && KB == 0) {                                                  \
            if (a.ppA.x != nullptr) { hipLaunchKernelGGL((direct_fast_kernel<MODEL, D, KA, KB, true>), grid, block, 0, s, a); return hipGetLastError(); } \
        }                                                                                   \
        hipLaunchKernelGGL((direct_fast_kernel<MODEL, D, KA, KB, false>), grid, block, 0, s, a);   \
        return hipGetLastError();                                                           \
    }
#define SSDE_FK(MODEL, D) SSDE_F(MODEL, D, 0, 0) SSDE_F(MODEL, D, 9, 0) SSDE_F(MODEL, D, 12, 0) SSDE_F(MODEL, D, 24, 0) SSDE_F(MODEL, D, 12, 12) SSDE_F(MODEL, D, 24, 24)
    SSDE_FK(M_BM, 1) SSDE_FK(M_BM, 2) SSDE_FK(M_OU, 1) SSDE_FK(M_OU, 2) SSDE_FK(M_BM_T, 1) SSDE_FK(M_CIR, 1) SSDE_FK(M_CIR, 2)
#undef SSDE_FK
#undef SSDE_F
    return hipErrorInvalidValue;
}

// min / max of the scored intervals (regular-grid detection for the direct families)
__global__ __launch_bounds__(256) void dt_minmax_kernel(const double* times, const uint32_t* scored, int64_t n, double* out) {
    __shared__ double sh[2][4];
    double mn = INFINITY, mx = -INFINITY;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        if (i >= 1 && (!scored || ((scored[i >> 5] >> (i & 31)) & 1u))) {       // scored == NULL: every consecutive pair
            const double dt = times[i] - times[i - 1];
            mn = fmin(mn, dt); mx = fmax(mx, dt);
            if (dt != dt) mx = INFINITY;
        }
    }
    for (int o = 32; o > 0; o >>= 1) { mn = fmin(mn, __shfl_xor(mn, o, 64)); mx = fmax(mx, __shfl_xor(mx, o, 64)); }
    if ((threadIdx.x & 63) == 0) { sh[0][threadIdx.x >> 6] = mn; sh[1][threadIdx.x >> 6] = mx; }
    __syncthreads();
    if (threadIdx.x == 0) {
        out[2 * blockIdx.x] = fmin(fmin(sh[0][0], sh[0][1]), fmin(sh[0][2], sh[0][3]));
        out[2 * blockIdx.x + 1] = fmax(fmax(sh[1][0], sh[1][1]), fmax(sh[1][2], sh[1][3]));
    }
}
hipError_t launch_dt_minmax(const double* times, const uint32_t* scored, int64_t n, double* out, int n_blocks, hipStream_t s) {
    hipLaunchKernelGGL(dt_minmax_kernel, dim3(n_blocks), dim3(256), 0, s, times, scored, n, out);
    return hipGetLastError();
}

hipError_t launch_direct(const DirectArgs& a, hipStream_t s) {
    dim3 grid(a.n_blocks), block(256);
#define SSDE_L(MODEL, D, K)                                                            \
    if (a.model == MODEL && a.d == D && a.n_slots <= K) {                              \
        hipLaunchKernelGGL((direct_kernel<MODEL, D, K>), grid, block, 0, s, a);        \
        return hipGetLastError();                                                      \
    }
#define SSDE_LK(MODEL, D) SSDE_L(MODEL, D, 4) SSDE_L(MODEL, D, 16) SSDE_L(MODEL, D, 32) SSDE_L(MODEL, D, 64)
    SSDE_LK(M_BM, 1) SSDE_LK(M_BM, 2) SSDE_LK(M_OU, 1) SSDE_LK(M_OU, 2) SSDE_LK(M_BM_T, 1) SSDE_LK(M_CIR, 1) SSDE_LK(M_CIR, 2)
#undef SSDE_LK
#undef SSDE_L
    return hipErrorInvalidValue;
}

}  // namespace ssde
