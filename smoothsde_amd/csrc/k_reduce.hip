// k_reduce.hip -- deterministic final reduction of the per-workgroup partial sums.
// One workgroup per output slot; each thread adds a fixed strided subset, then a fixed LDS tree:
// bitwise reproducible from run to run (no atomics).
#include "ssde_device.hpp"

namespace ssde {

__global__ __launch_bounds__(256) void reduce_kernel(const ReduceArgs A) {
    __shared__ double sh[256];
    reduce_slot(A, blockIdx.x, sh);
}

hipError_t launch_reduce(const ReduceArgs& a, hipStream_t s) {
    hipLaunchKernelGGL(reduce_kernel, dim3(a.n_out + 1), dim3(256), 0, s, a);
    return hipGetLastError();
}

}  // namespace ssde
