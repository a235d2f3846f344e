// tools/microbench_fetch.hip -- calibration of rocprofv3's FETCH_SIZE for THIS engine's access pattern:
// coalesced 8-byte-per-lane wave loads (512 B per wave instruction) of a buffer far larger than the
// 256 MiB Infinity Cache, each byte read once.  MI355X_MICROARCH.md (HBM section) calibrates only the
// 16-B-per-lane case (FETCH_SIZE = 1/2 of the bytes) and asks for a calibration of anything else.
//   rocprofv3 --pmc FETCH_SIZE --kernel-include-regex read8 -- build/microbench_fetch
#include <hip/hip_runtime.h>
#include <cstdio>

__global__ __launch_bounds__(64) void read8_kernel(const double* in, double* out, long n_per_wave) {
    const double* p = in + (long)blockIdx.x * n_per_wave * 64 + threadIdx.x;
    double acc = 0;
    for (long i = 0; i < n_per_wave; i += 4) {
        double a = p[(i + 0) * 64], b = p[(i + 1) * 64], c = p[(i + 2) * 64], d = p[(i + 3) * 64];
        acc += (a + b) + (c + d);
    }
    out[blockIdx.x * 64 + threadIdx.x] = acc;
}

int main() {
    const long blocks = 2048, n_per_wave = 1024;          // 2048 * 1024 * 512 B = 1 GiB
    const long n = blocks * n_per_wave * 64;
    double *in, *out;
    hipMalloc(&in, n * 8); hipMalloc(&out, blocks * 64 * 8);
    hipMemset(in, 0, n * 8);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 3; rep++) {
        hipEventRecord(e0);
        read8_kernel<<<blocks, 64>>>(in, out, n_per_wave);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("read8: %ld bytes in %.3f ms = %.2f TB/s\n", n * 8, ms, n * 8 / (ms * 1e-3) / 1e12);
    }
    return 0;
}
