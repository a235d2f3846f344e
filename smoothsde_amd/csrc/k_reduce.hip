// k_reduce.hip -- deterministic final reduction of the per-workgroup partial sums.
// One workgroup per output slot; each thread adds a fixed strided subset, then a fixed LDS tree:
// bitwise reproducible from run to run (no atomics).
#include "ssde_device.hpp"

namespace ssde {

__global__ __launch_bounds__(256) void reduce_kernel(const ReduceArgs A) {
    __shared__ double sh[256];
    const int slot = blockIdx.x;  // 0 = nllk, 1.. = gradient entries, n_out = hand-over check
    const int tid = threadIdx.x;
    double acc = 0.0;
    if (slot == A.n_out) {
        for (int b = tid; b < A.n_chk; b += 256) acc = fmax(acc, A.chk[b] == A.chk[b] ? A.chk[b] : INFINITY);
        sh[tid] = acc;
        __syncthreads();
        for (int o = 128; o > 0; o >>= 1) {
            if (tid < o) sh[tid] = fmax(sh[tid], sh[tid + o]);
            __syncthreads();
        }
        if (tid == 0) A.out[slot] = sh[0];
        return;
    }
    if (slot == 0) {
        for (int part = 0; part < A.n_value_parts; part++) {
            const double* p = A.partials + ((int64_t)part * A.nacc) * A.n_blocks;  // accumulator 0
            for (int b = tid; b < A.n_blocks; b += 256) acc += p[b];
        }
    } else {
        const int nk = A.nacc - 1;
        for (int part = 0; part < A.n_parts; part++) {
            for (int k = 1; k < A.nacc; k++) {
                if (A.map[(part / A.chunks_per_part) * nk + (k - 1)] != slot) continue;
                const double* p = A.partials + ((int64_t)part * A.nacc + k) * A.n_blocks;
                for (int b = tid; b < A.n_blocks; b += 256) acc += p[b];
            }
        }
    }
    sh[tid] = acc;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (tid < o) sh[tid] += sh[tid + o];
        __syncthreads();
    }
    if (tid == 0) {
        double r = sh[0];
        for (int i = 0; i < 4; i++)
            if (A.add_slot[i] == slot) r += A.add[i];
        A.out[slot] = r;
    }
}

hipError_t launch_reduce(const ReduceArgs& a, hipStream_t s) {
    hipLaunchKernelGGL(reduce_kernel, dim3(a.n_out + 1), dim3(256), 0, s, a);
    return hipGetLastError();
}

}  // namespace ssde
