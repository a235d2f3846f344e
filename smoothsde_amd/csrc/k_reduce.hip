// k_reduce.hip -- deterministic final reduction of the per-workgroup partial sums.
// One workgroup per output slot; each thread adds a fixed strided subset, then a fixed LDS tree:
// bitwise reproducible from run to run (no atomics).
#include "ssde_device.hpp"

namespace ssde {

__global__ __launch_bounds__(256) void reduce_kernel(const ReduceArgs A) {
    __shared__ double sh[256];
    publish_if_last(A, reduce_slot(A, blockIdx.x, sh));
}

hipError_t launch_reduce(const ReduceArgs& a, hipStream_t s) {
    ReduceArgs r = a;
    r.pub_blocks = a.n_out + 1;
    hipLaunchKernelGGL(reduce_kernel, dim3(a.n_out + 1), dim3(256), 0, s, r);
    return hipGetLastError();
}

// dst[i] += src[i]: sums the result vectors of shards that share one device (the rehearsal mode of a multi-device
// handle on a one-GPU machine, where RCCL refuses two ranks on a device); fixed order, one thread per entry
__global__ __launch_bounds__(256) void sum_into_kernel(double* dst, const double* src, int n) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < n) dst[i] += src[i];
}

}  // namespace ssde

namespace ssde_engine {
hipError_t launch_sum_into(double* dst, const double* src, int n, hipStream_t s) {
    hipLaunchKernelGGL(ssde::sum_into_kernel, dim3((n + 255) / 256), dim3(256), 0, s, dst, src, n);
    return hipGetLastError();
}
}  // namespace ssde_engine
