// k_iso_colvar_support.hip -- create-time and per-evaluation helpers of the row-varying tau / nu kernels (k_iso_colvar.hip): the range
// of every streamed column per group and the statistics of H_array (the window planner's bounds), the reduction of the linear
// predictors' ranges a launch saw, and the detection of design columns that two parameters share (streamed once).
#include "ssde_device.hpp"

namespace ssde {

// range of every streamed column over the rows of a group (create time: the window planner bounds the linear predictors with it)
__global__ __launch_bounds__(WG_WAVES * WAVE) void colvar_ranges_kernel(TileView tv, int c_col, int K, double* out /* [n_groups][K][2] */) {
    __shared__ double sh[WG_WAVES][2];
    const int g = blockIdx.x, lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const double* base = tv.tiles + tv.group_off[g] + lane;
    const int ns = tv.lane_nsteps[g * WAVE + lane];
    for (int k = 0; k < K; k++) {
        double lo = INFINITY, hi = -INFINITY;
        for (int s = wv; s < ns; s += WG_WAVES) {
            const double x = base[((int64_t)s * tv.C + c_col + k) * WAVE];
            lo = fmin(lo, x); hi = fmax(hi, x);
            if (x != x) { lo = -INFINITY; hi = INFINITY; }
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) { lo = fmin(lo, __shfl_xor(lo, o, 64)); hi = fmax(hi, __shfl_xor(hi, o, 64)); }
        if (lane == 0) { sh[wv][0] = lo; sh[wv][1] = hi; }
        __syncthreads();
        if (threadIdx.x == 0) {
            for (int w = 1; w < WG_WAVES; w++) { lo = fmin(lo, sh[w][0]); hi = fmax(hi, sh[w][1]); }
            out[((int64_t)g * K + k) * 2] = lo; out[((int64_t)g * K + k) * 2 + 1] = hi;
        }
        __syncthreads();
    }
}
hipError_t launch_colvar_ranges(const TileView& tv, int c_col, int K, double* out, hipStream_t s) {
    if (tv.n_groups == 0 || K == 0) return hipSuccess;
    hipLaunchKernelGGL(colvar_ranges_kernel, dim3(tv.n_groups), dim3(WG_WAVES * WAVE), 0, s, tv, c_col, K, out);
    return hipGetLastError();
}

// the ranges of the linear predictors over the whole launch -> four doubles in host-visible memory (read by the next window plan)
__global__ __launch_bounds__(256) void colvar_range_reduce_kernel(const double* wg, int n_wg, double* out) {
    __shared__ double sh[4][4];
    double v[4] = {INFINITY, -INFINITY, INFINITY, -INFINITY};
    for (int i = threadIdx.x; i < n_wg; i += 256)
        for (int k = 0; k < 4; k++) v[k] = (k & 1) ? fmax(v[k], wg[4 * (int64_t)i + k]) : fmin(v[k], wg[4 * (int64_t)i + k]);
    for (int k = 0; k < 4; k++) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) { const double t = __shfl_xor(v[k], o, 64); v[k] = (k & 1) ? fmax(v[k], t) : fmin(v[k], t); }
        if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6][k] = v[k];
    }
    __syncthreads();
    if (threadIdx.x < 4) {
        const int k = threadIdx.x;
        double t = sh[0][k];
        for (int w = 1; w < 4; w++) t = (k & 1) ? fmax(t, sh[w][k]) : fmin(t, sh[w][k]);
        out[k] = t;
    }
}
hipError_t launch_colvar_range_reduce(const double* wg, int n_wg, double* out_pinned, hipStream_t s) {
    hipLaunchKernelGGL(colvar_range_reduce_kernel, dim3(1), dim3(256), 0, s, wg, n_wg, out_pinned);
    return hipGetLastError();
}

// per group: the largest diagonal entry of H_array[,,i] over its rows, and the largest |H01 - H10| (create time: the window
// planner's observation variance; the full-covariance lanes take a symmetric H)
__global__ __launch_bounds__(WG_WAVES * WAVE) void colvar_h_stats_kernel(TileView tv, int c_h, int d, double* out /* [n_groups][2] */) {
    __shared__ double sh[WG_WAVES][2];
    const int g = blockIdx.x, lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const double* base = tv.tiles + tv.group_off[g] + lane;
    const int ns = tv.lane_nsteps[g * WAVE + lane];
    double hmax = 0.0, asym = 0.0;
    for (int s = wv; s < ns; s += WG_WAVES) {
        const double* p = base + ((int64_t)s * tv.C + c_h) * WAVE;
        const double h00 = p[0], h10 = d == 2 ? p[WAVE] : 0.0, h01 = d == 2 ? p[2 * WAVE] : 0.0, h11 = d == 2 ? p[3 * WAVE] : p[0];
        hmax = fmax(hmax, fmax(h00, h11));
        asym = fmax(asym, fabs(h01 - h10));
        if (!(h00 == h00) || !(h11 == h11) || !(h01 == h01) || !(h10 == h10)) asym = INFINITY;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { hmax = fmax(hmax, __shfl_xor(hmax, o, 64)); asym = fmax(asym, __shfl_xor(asym, o, 64)); }
    if (lane == 0) { sh[wv][0] = hmax; sh[wv][1] = asym; }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < WG_WAVES; w++) { hmax = fmax(hmax, sh[w][0]); asym = fmax(asym, sh[w][1]); }
        out[2 * g] = hmax; out[2 * g + 1] = asym;
    }
}
hipError_t launch_colvar_h_stats(const TileView& tv, int c_h, int d, double* out, hipStream_t s) {
    if (tv.n_groups == 0) return hipSuccess;
    hipLaunchKernelGGL(colvar_h_stats_kernel, dim3(tv.n_groups), dim3(WG_WAVES * WAVE), 0, s, tv, c_h, d, out);
    return hipGetLastError();
}

// are two design columns (device arrays of n doubles) the same numbers?  *differ is raised if not (create time)
__global__ __launch_bounds__(256) void cols_differ_kernel(const double* a, const double* b, int64_t n, int* differ) {
    bool d = false;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const long long x = __double_as_longlong(a[i]), y = __double_as_longlong(b[i]);
        d = d || x != y;
    }
    if (__any(d) && (threadIdx.x & 63) == 0) atomicOr(differ, 1);
}
hipError_t launch_cols_differ(const double* a, const double* b, int64_t n, int* differ, hipStream_t s) {
    hipLaunchKernelGGL(cols_differ_kernel, dim3(1024), dim3(256), 0, s, a, b, n, differ);
    return hipGetLastError();
}
}  // namespace ssde
