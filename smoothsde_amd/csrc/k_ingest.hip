// k_ingest.hip -- one-off re-tiling of the reference's long-format data into the wavefront-major
// HBM layout of ssde_device.hpp (run once per ssde_create; every evaluation afterwards streams
// the tiles with fully coalesced loads).  Also: per-row flags for the direct families and
// first-row flags for segment discovery when the caller's data already live in HBM.
#include "ssde_device.hpp"

namespace ssde {

// grid: (n_groups, step chunks); block: 256 threads = 4 waves.  The long format is read along the tracks (64
// consecutive rows of one track by one wave: coalesced), transposed through LDS and written lane = track (coalesced):
// every fetched line is used whole.  (The first version read lane = track directly from the long format: one 8-byte
// element per fetched line, 24x the traffic -- profiles/r01_j_pmc_bytes.csv.)
__global__ __launch_bounds__(4 * WAVE) void ingest_kernel(const IngestArgs A) {
    __shared__ double tile[WAVE][WAVE + 1];      // [track][step], padded: conflict-free both ways
    __shared__ int64_t s_row0[WAVE];
    __shared__ int s_ns[WAVE];
    __shared__ double s_red[3][4];
    const int g = blockIdx.x;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int L = A.group_len[g];
    const int chunk = ((L + gridDim.y - 1) / gridDim.y + WAVE - 1) / WAVE * WAVE;
    const int s_lo = blockIdx.y * chunk;
    const int s_hi = min(L, s_lo + chunk);
    if (threadIdx.x < WAVE) { s_row0[threadIdx.x] = A.lane_row0[g * WAVE + threadIdx.x]; s_ns[threadIdx.x] = A.lane_nsteps[g * WAVE + threadIdx.x]; }
    __syncthreads();
    double* out = A.tiles + A.group_off[g];
    const int C = A.C, d = A.d;
    const int n_src = (A.c_obs ? 1 : 0) + d + (A.h_array ? d * d : 0) + A.ncols;
    double dmin = INFINITY, dmax = -INFINITY, seen_nan = 0.0;
    for (int s0 = s_lo; s0 < s_hi; s0 += WAVE) {
        // the interval channel is read even when it is not stored: its range decides whether the grid is regular
        for (int c = A.c_obs ? 0 : -1; c < n_src; c++) {
            const int ch = c;                                    // -1: intervals for the statistics only
            const bool is_dt = (A.c_obs && ch == 0) || ch < 0;
            const int k = ch - A.c_obs;                          // 0.. : obs columns, then H entries, then design columns
            // ---- read phase: wave wv takes tracks wv, wv + 4, ...; lane = step within the block ------------------
            for (int t = wv; t < WAVE; t += 4) {
                const int s = s0 + lane;
                const int ns = s_ns[t];
                double v = (is_dt && A.c_obs) ? 1.0 : 0.0;
                if (s < ns) {
                    const int64_t i = s_row0[t] + 1 + s;
                    if (is_dt) {
                        // dtimes(i): nllk_ctcrw.hpp:126-129 (the cross-track value at a track's last row is kept: the
                        // engine never uses that prediction for the likelihood, only ssde_report shows it)
                        v = (i < A.n - 1) ? A.times[i + 1] - A.times[i] : A.last_dt;
                        if (s < ns - 1) { dmin = fmin(dmin, v); dmax = fmax(dmax, v); if (v != v) dmax = INFINITY; }
                    } else if (k < d) {
                        v = A.obs[i + (int64_t)k * A.n];
                        if (v != v) seen_nan = 1.0;
                    } else if (A.h_array && k < d + d * d) {
                        v = A.h_array[(k - d) + i * (int64_t)(d * d)];
                    } else {
                        v = A.cols[k - d - (A.h_array ? d * d : 0)][i];
                    }
                }
                tile[t][lane] = v;
            }
            __syncthreads();
            // ---- write phase: wave wv takes steps wv, wv + 4, ...; lane = track --------------------------------------
            if (ch >= 0) {
                for (int u = wv; u < WAVE; u += 4) {
                    const int s = s0 + u;
                    if (s < s_hi) out[((int64_t)s * C + ch) * WAVE + lane] = tile[lane][u];
                }
            }
            __syncthreads();
        }
    }
    for (int o = 32; o > 0; o >>= 1) {
        dmin = fmin(dmin, __shfl_xor(dmin, o, 64));
        dmax = fmax(dmax, __shfl_xor(dmax, o, 64));
        seen_nan = fmax(seen_nan, __shfl_xor(seen_nan, o, 64));
    }
    if (lane == 0) { s_red[0][wv] = dmin; s_red[1][wv] = dmax; s_red[2][wv] = seen_nan; }
    __syncthreads();
    if (threadIdx.x == 0) {
        const int64_t o3 = ((int64_t)g * gridDim.y + blockIdx.y) * 3;
        A.dt_minmax[o3 + 0] = fmin(fmin(s_red[0][0], s_red[0][1]), fmin(s_red[0][2], s_red[0][3]));
        A.dt_minmax[o3 + 1] = fmax(fmax(s_red[1][0], s_red[1][1]), fmax(s_red[1][2], s_red[1][3]));
        A.dt_minmax[o3 + 2] = fmax(fmax(s_red[2][0], s_red[2][1]), fmax(s_red[2][2], s_red[2][3]));
    }
    if (blockIdx.y == 0 && threadIdx.x < WAVE) {
        // initial state: a0 = first observation (velocities 0 for CTCRW), R/sde.R:549, 576-580
        const int64_t row0 = s_row0[lane];
        const int ns = s_ns[lane];
        double* a0 = A.a0 + (int64_t)g * A.sdim * WAVE + lane;
        for (int c = 0; c < A.sdim; c++) {
            double v = 0.0;
            if (ns > 0 || row0 >= 0) {
                if (A.a0_src) v = A.a0_src[A.lane_seg[g * WAVE + lane] + (int64_t)c * A.n_seg];
                else if (A.model == M_CTCRW) v = (c & 1) ? 0.0 : A.obs[row0 + (int64_t)(c >> 1) * A.n];
                else v = A.obs[row0 + (int64_t)c * A.n];
            }
            a0[c * WAVE] = v;
        }
    }
}

int ingest_ychunks(int n_groups) {
    int ychunks = 1;
    // enough workgroups to fill the chip even for few, long groups
    while ((int64_t)n_groups * ychunks < 2048 && ychunks < 64) ychunks *= 2;
    return ychunks;
}
hipError_t launch_ingest(const IngestArgs& a, hipStream_t s) {
    if (a.n_groups == 0) return hipSuccess;
    hipLaunchKernelGGL(ingest_kernel, dim3(a.n_groups, a.ychunks), dim3(4 * WAVE), 0, s, a);
    return hipGetLastError();
}

__global__ void scored_mask_kernel(const double* id, int64_t n, uint32_t* mask) {
    const int64_t w = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;  // one 32-row word per thread
    const int64_t i0 = w * 32;
    if (i0 >= n) return;
    uint32_t m = 0;
    for (int b = 0; b < 32; b++) {
        const int64_t i = i0 + b;
        if (i >= 1 && i < n && id[i] == id[i - 1]) m |= (1u << b);  // nllk_sde.hpp:79
    }
    mask[w] = m;
}
hipError_t launch_scored_mask(const double* id, int64_t n, uint32_t* mask, hipStream_t s) {
    const int64_t words = (n + 31) / 32;
    if (words == 0) return hipSuccess;
    hipLaunchKernelGGL(scored_mask_kernel, dim3((unsigned)((words + 255) / 256)), dim3(256), 0, s, id, n, mask);
    return hipGetLastError();
}

__global__ void first_flags_kernel(const double* id, int64_t n, uint8_t* flags) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    flags[i] = (i == 0 || id[i] != id[i - 1]) ? 1 : 0;  // nllk_ctcrw.hpp:196
}
hipError_t launch_first_flags(const double* id, int64_t n, uint8_t* flags, hipStream_t s) {
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(first_flags_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, id, n, flags);
    return hipGetLastError();
}

// ---- which tracks have a row that is not a number (ssde_engine.hip: tracks are dealt to wavefronts clean ones first) ----
__global__ void seg_nan_kernel(const double* obs, int64_t n, int d, const int64_t* starts, int64_t n_seg, int* flags) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    bool bad = false;
    for (int a = 0; a < d; a++) { const double v = obs[i + (int64_t)a * n]; bad = bad || (v != v); }
    if (!bad) return;
    int64_t lo = 0, hi = n_seg;                     // the segment with starts[lo] <= i < starts[lo + 1]
    while (hi - lo > 1) { const int64_t mid = (lo + hi) >> 1; if (starts[mid] <= i) lo = mid; else hi = mid; }
    atomicMax(&flags[lo], (int)min((long long)(i - starts[lo]), (long long)(INT32_MAX - 1)) + 1);   // != 0: the track misses a row; 1 + its last such row
}
hipError_t launch_seg_nan(const double* obs, int64_t n, int d, const int64_t* starts, int64_t n_seg, int* flags, hipStream_t s) {
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(seg_nan_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, obs, n, d, starts, n_seg, flags);
    return hipGetLastError();
}

// ---- lattice padding (ssde_engine.hip: lattice_pad) --------------------------------------------------------------------
// pos[i] = lattice row of the caller's row i.  One thread per caller row: its own time stamp and observations go to pos[i];
// the lattice rows between it and the previous row of the same track (fixes absent from the data) get interpolated time
// stamps -- the observation columns were pre-filled with NA_real_.
__global__ void lattice_scatter_kernel(const int64_t* pos, const double* id, const double* times, const double* obs, int64_t n, int d,
                                       int64_t np, double delta, double* times_p, double* obs_p) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int64_t p = pos[i];
    times_p[p] = times[i];
    for (int a = 0; a < d; a++) obs_p[p + (int64_t)a * np] = obs[i + (int64_t)a * n];
    if (i > 0 && id[i] == id[i - 1]) {
        const int64_t p0 = pos[i - 1];
        for (int64_t j = p0 + 1; j < p; j++) times_p[j] = times[i - 1] + (double)(j - p0) * delta;
    }
}
__global__ void fill_bits_kernel(double* x, int64_t n, unsigned long long bits) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) x[i] = __longlong_as_double((long long)bits);
}
hipError_t launch_lattice_scatter(const int64_t* pos, const double* id, const double* times, const double* obs, int64_t n, int d,
                                  int64_t np, double delta, double* times_p, double* obs_p, hipStream_t s) {
    hipLaunchKernelGGL(fill_bits_kernel, dim3((unsigned)((np * d + 255) / 256)), dim3(256), 0, s, obs_p, np * d, 0x7FF00000000007A2ull);   // NA_real_
    hipLaunchKernelGGL(lattice_scatter_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, pos, id, times, obs, n, d, np, delta, times_p, obs_p);
    return hipGetLastError();
}
// aest_all of the caller's rows out of the lattice rows' (ssde_report)
__global__ void lattice_gather_kernel(const int64_t* pos, const double* src, int64_t n, int64_t np, int ncol, double* dst) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    for (int c = 0; c < ncol; c++) dst[i + (int64_t)c * n] = src[pos[i] + (int64_t)c * np];
}
hipError_t launch_lattice_gather(const int64_t* pos, const double* src, int64_t n, int64_t np, int ncol, double* dst, hipStream_t s) {
    hipLaunchKernelGGL(lattice_gather_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, pos, src, n, np, ncol, dst);
    return hipGetLastError();
}

// Dimension parts of a Kalman problem wider than two columns (ssde_engine_dist.hip): the reference decides "missing" on
// column 0 of the WHOLE response (nllk_ctcrw.hpp:214), a part other than the first one reads it off its own first column.
// Rewrites that column so that the two agree: missing where the lead column is, and an OBSERVED NaN (not R's NA payload)
// where the lead column is observed and this one is not a number -- the reference then scores a NaN innovation.  With
// SSDE_NA_ANY_NAN there is no NaN that is not "missing": such a row raises *poison and the parent returns NaN.
// First rows of a track are only ever read through the default a0 (nllk_ctcrw.hpp:196-200) and stay as they are.
__global__ void na_follow_kernel(const double* id, const double* lead, double* col, int64_t n, int any_nan, int* poison) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n || i == 0 || id[i] != id[i - 1]) return;
    const double v = col[i];
    if (is_na(lead[i], any_nan)) col[i] = __longlong_as_double(0x7FF00000000007A2ll);        // NA_real_
    else if (v != v) {
        if (any_nan) atomicOr(poison, 1);
        else col[i] = __longlong_as_double(0x7FF8000000000000ll);
    }
}
// ---- a response wider than two columns with a device-resident H_array (ssde_engine_dist.hip) ------------------------------------
// H is [D x D x n], column-major per row.  couples != 0 afterwards: some row has an entry between different column pairs.
__global__ void h_couples_kernel(const double* H, int64_t n, int D, int* couples) {
    const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n) return;
    const double* Hr = H + (size_t)r * D * D;
    bool bad = false;
    for (int j = 0; j < D; j++)
        for (int i = 0; i < D; i++)
            if (i / 2 != j / 2) { const double v = Hr[i + (size_t)j * D]; bad = bad || !(v == 0.0); }   // (a NaN counts as coupling)
    if (bad) atomicOr(couples, 1);
}
// the cnt x cnt block of every row that belongs to response columns [dlo, dlo + cnt)
__global__ void h_block_kernel(const double* H, int64_t n, int D, int dlo, int cnt, double* out) {
    const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n) return;
    for (int jj = 0; jj < cnt; jj++)
        for (int ii = 0; ii < cnt; ii++)
            out[(size_t)r * cnt * cnt + ii + (size_t)jj * cnt] = H[(size_t)r * D * D + (dlo + ii) + (size_t)(dlo + jj) * D];
}
hipError_t launch_h_couples(const double* H, int64_t n, int D, int* couples, hipStream_t s) {
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(h_couples_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, H, n, D, couples);
    return hipGetLastError();
}
hipError_t launch_h_block(const double* H, int64_t n, int D, int dlo, int cnt, double* out, hipStream_t s) {
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(h_block_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, H, n, D, dlo, cnt, out);
    return hipGetLastError();
}

hipError_t launch_na_follow(const double* id, const double* lead, double* col, int64_t n, int any_nan, int* poison, hipStream_t s) {
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(na_follow_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, id, lead, col, n, any_nan, poison);
    return hipGetLastError();
}

}  // namespace ssde
