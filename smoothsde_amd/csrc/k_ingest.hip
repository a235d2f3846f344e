// k_ingest.hip -- one-off re-tiling of the reference's long-format data into the wavefront-major
// HBM layout of ssde_device.hpp (run once per ssde_create; every evaluation afterwards streams
// the tiles with fully coalesced loads).  Also: per-row flags for the direct families and
// first-row flags for segment discovery when the caller's data already live in HBM.
#include "ssde_device.hpp"

namespace ssde {

// grid: (n_groups, step chunks); block: 64 lanes.  Reads are strided (one track per lane), writes
// are coalesced; this runs once, the evaluation kernels run hundreds of times.
__global__ __launch_bounds__(WAVE) void ingest_kernel(const IngestArgs A) {
    const int g = blockIdx.x;
    const int lane = threadIdx.x;
    const int L = A.group_len[g];
    const int chunk = (L + gridDim.y - 1) / gridDim.y;
    const int s_lo = blockIdx.y * chunk;
    const int s_hi = min(L, s_lo + chunk);
    const int64_t row0 = A.lane_row0[g * WAVE + lane];
    const int ns = A.lane_nsteps[g * WAVE + lane];
    double* out = A.tiles + A.group_off[g] + lane;
    const int C = A.C, d = A.d;
    double dmin = INFINITY, dmax = -INFINITY, seen_nan = 0.0;
    for (int s = s_lo; s < s_hi; s++) {
        double* o = out + (int64_t)s * C * WAVE;
        if (s < ns) {
            const int64_t i = row0 + 1 + s;
            // dtimes(i): nllk_ctcrw.hpp:126-129 (the cross-track value at a track's last row is kept:
            // the engine never uses that prediction for the likelihood, only ssde_report shows it)
            const double dti = (i < A.n - 1) ? A.times[i + 1] - A.times[i] : 1.0;
            if (A.c_obs) o[0] = dti;
            if (s < ns - 1) { dmin = fmin(dmin, dti); dmax = fmax(dmax, dti); if (dti != dti) dmax = INFINITY; }
            for (int a = 0; a < d; a++) {
                const double yv = A.obs[i + (int64_t)a * A.n];
                o[(A.c_obs + a) * WAVE] = yv;
                if (yv != yv) seen_nan = 1.0;
            }
            int c = A.c_obs + d;
            if (A.h_array)
                for (int k = 0; k < d * d; k++) o[(c++) * WAVE] = A.h_array[k + i * (int64_t)(d * d)];
            for (int k = 0; k < A.ncols; k++) o[(c++) * WAVE] = A.cols[k][i];
        } else {
            for (int c = 0; c < C; c++) o[c * WAVE] = (c == 0 && A.c_obs) ? 1.0 : 0.0;
        }
    }
    for (int o = 32; o > 0; o >>= 1) {
        dmin = fmin(dmin, __shfl_xor(dmin, o, 64));
        dmax = fmax(dmax, __shfl_xor(dmax, o, 64));
        seen_nan = fmax(seen_nan, __shfl_xor(seen_nan, o, 64));
    }
    if (lane == 0) {
        A.dt_minmax[((int64_t)g * gridDim.y + blockIdx.y) * 3 + 0] = dmin;
        A.dt_minmax[((int64_t)g * gridDim.y + blockIdx.y) * 3 + 1] = dmax;
        A.dt_minmax[((int64_t)g * gridDim.y + blockIdx.y) * 3 + 2] = seen_nan;
    }
    if (blockIdx.y == 0) {
        // initial state: a0 = first observation (velocities 0 for CTCRW), R/sde.R:549, 576-580
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
    hipLaunchKernelGGL(ingest_kernel, dim3(a.n_groups, a.ychunks), dim3(WAVE), 0, s, a);
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

}  // namespace ssde
