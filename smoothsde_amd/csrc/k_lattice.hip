// k_lattice.hip -- device side of the lattice layout (ssde_engine.hip: lattice_pad): the interval test, the
// lattice position of every caller row (a prefix sum over the rows' step counts) and the report map, without a trip of the
// time stamps through host memory -- the create-time cost of the layout is what a short fit has to amortise.
#include <hipcub/hipcub.hpp>

#include "ssde_device.hpp"

namespace ssde {

namespace {

// an interval the likelihood USES: rows i - 1 and i of one track, and i - 1 is not the track's first row (a track's first
// interval is never used: a0 is the prediction for the second row as it stands, nllk_ctcrw.hpp:195-200)
__device__ __forceinline__ bool used_interval(const double* id, int64_t i) {
    return i >= 2 && id[i] == id[i - 1] && id[i - 1] == id[i - 2];
}

__global__ __launch_bounds__(256) void used_dt_minmax_kernel(const double* id, const double* times, int64_t n, double* out) {
    __shared__ double sh[2][4];
    double mn = INFINITY, mx = -INFINITY;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        if (used_interval(id, i)) {
            const double dt = times[i] - times[i - 1];
            mn = fmin(mn, dt); mx = fmax(mx, dt);
            if (!(dt > 0.0) || dt != dt || dt - dt != 0.0) { mn = -INFINITY; mx = INFINITY; }      // not a grid at all
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

// lattice rows row i adds: the steps of the interval before it (1 for a track's first and second row); *bad is raised
// by an interval that is not a whole multiple of delta to rtol
__global__ void lattice_steps_kernel(const double* id, const double* times, int64_t n, double delta, double rtol, int64_t* inc, int* bad) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    int64_t k = 1;
    if (used_interval(id, i)) {
        const double r = (times[i] - times[i - 1]) / delta;
        const double kr = rint(r);
        if (!(fabs(r - kr) <= rtol * kr) || kr < 1.0) { *bad = 1; }
        else k = (int64_t)kr;
    }
    inc[i] = k;
}

// pos[i] = (inclusive prefix sum)[i] - 1, in place; and the report map: REPORT(aest_all) row i is the state predicted to row
// i + 1's time (nllk_ctcrw.hpp:246) = the lattice row just before row i + 1's (a track's first and last row: their own)
__global__ void lattice_maps_kernel(const double* id, int64_t n, const int64_t* scan, int64_t* pos, int64_t* rep) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int64_t p = scan[i] - 1;
    pos[i] = p;
    const bool first = i == 0 || id[i] != id[i - 1];
    const bool last = i + 1 >= n || id[i + 1] != id[i];
    rep[i] = (first || last) ? p : scan[i + 1] - 2;
}

__global__ void gather_i64_kernel(const int64_t* src, const int64_t* idx, int64_t m, int64_t* dst) {
    const int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k < m) dst[k] = src[idx[k]];
}

}  // namespace

hipError_t launch_used_dt_minmax(const double* id, const double* times, int64_t n, double* out, int n_blocks, hipStream_t s) {
    hipLaunchKernelGGL(used_dt_minmax_kernel, dim3(n_blocks), dim3(256), 0, s, id, times, n, out);
    return hipGetLastError();
}

// inc (n int64, scratch) -> pos, rep (n int64 each); *n_lattice = lattice rows; *bad != 0: not a lattice
hipError_t lattice_positions(const double* id, const double* times, int64_t n, double delta, double rtol, int64_t* inc, int64_t* pos,
                             int64_t* rep, int* bad_dev, int64_t* n_lattice, int* bad_host, hipStream_t s) {
    hipError_t e = hipMemsetAsync(bad_dev, 0, sizeof(int), s);
    if (e != hipSuccess) return e;
    const unsigned nb = (unsigned)((n + 255) / 256);
    hipLaunchKernelGGL(lattice_steps_kernel, dim3(nb), dim3(256), 0, s, id, times, n, delta, rtol, inc, bad_dev);
    size_t tmp_bytes = 0;
    e = hipcub::DeviceScan::InclusiveSum(nullptr, tmp_bytes, inc, inc, n, s);
    if (e != hipSuccess) return e;
    void* tmp = nullptr;
    e = hipMalloc(&tmp, tmp_bytes);
    if (e != hipSuccess) return e;
    e = hipcub::DeviceScan::InclusiveSum(tmp, tmp_bytes, inc, inc, n, s);
    if (e == hipSuccess) {
        hipLaunchKernelGGL(lattice_maps_kernel, dim3(nb), dim3(256), 0, s, id, n, inc, pos, rep);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipMemcpyAsync(n_lattice, inc + (n - 1), sizeof(int64_t), hipMemcpyDeviceToHost, s);
    if (e == hipSuccess) e = hipMemcpyAsync(bad_host, bad_dev, sizeof(int), hipMemcpyDeviceToHost, s);
    if (e == hipSuccess) e = hipStreamSynchronize(s);
    (void)hipFree(tmp);
    return e;
}

hipError_t launch_gather_i64(const int64_t* src, const int64_t* idx, int64_t m, int64_t* dst, hipStream_t s) {
    if (m == 0) return hipSuccess;
    hipLaunchKernelGGL(gather_i64_kernel, dim3((unsigned)((m + 255) / 256)), dim3(256), 0, s, src, idx, m, dst);
    return hipGetLastError();
}

}  // namespace ssde
