// tools/microbench_fetch.hip -- (1) calibration of rocprofv3's FETCH_SIZE for THIS engine's access pattern:
// coalesced 8-byte-per-lane wave loads (512 B per wave instruction) of a buffer far larger than the
// 256 MiB Infinity Cache, each byte read once.  MI355X_MICROARCH.md (HBM section) calibrates only the
// 16-B-per-lane case (FETCH_SIZE = 1/2 of the bytes) and asks for a calibration of anything else.
//   rocprofv3 --pmc FETCH_SIZE --kernel-include-regex read8 -- build/microbench_fetch
// (2) what streaming rate one-wave workgroups reach with 8-B vs 16-B per-lane loads and 4..16 loads in flight.
#include <hip/hip_runtime.h>
#include <cstdio>

template <int INFLIGHT>
__global__ __launch_bounds__(64) void read8_kernel(const double* in, double* out, long n_per_wave) {
    const double* p = in + (long)blockIdx.x * n_per_wave * 64 + threadIdx.x;
    double acc = 0;
    for (long i = 0; i < n_per_wave; i += INFLIGHT) {
        double v[INFLIGHT];
#pragma unroll
        for (int k = 0; k < INFLIGHT; k++) v[k] = p[(i + k) * 64];
#pragma unroll
        for (int k = 0; k < INFLIGHT; k++) acc += v[k];
    }
    out[blockIdx.x * 64 + threadIdx.x] = acc;
}

template <int INFLIGHT>
__global__ __launch_bounds__(64) void read16_kernel(const double2* in, double* out, long n_per_wave) {
    const double2* p = in + (long)blockIdx.x * n_per_wave * 64 + threadIdx.x;
    double acc = 0;
    for (long i = 0; i < n_per_wave; i += INFLIGHT) {
        double2 v[INFLIGHT];
#pragma unroll
        for (int k = 0; k < INFLIGHT; k++) v[k] = p[(i + k) * 64];
#pragma unroll
        for (int k = 0; k < INFLIGHT; k++) acc += v[k].x + v[k].y;
    }
    out[blockIdx.x * 64 + threadIdx.x] = acc;
}

// (3) the same per-wave blocks (INFLIGHT x 512 B), but laid out so that the waves sweep memory as ONE front: wave w
// reads block (i * n_waves + w) at its i-th iteration -- neighbouring waves touch neighbouring blocks, the set of
// pages in use at any time is small (the layout question for the shared-covariance kernel's tiles)
template <int INFLIGHT>
__global__ __launch_bounds__(64) void read8_front_kernel(const double* in, double* out, long n_per_wave) {
    const long nw = gridDim.x;
    double acc = 0;
    for (long i = 0; i < n_per_wave; i += INFLIGHT) {
        const double* p = in + ((i / INFLIGHT) * nw + blockIdx.x) * (long)(INFLIGHT * 64) + threadIdx.x;
        double v[INFLIGHT];
#pragma unroll
        for (int k = 0; k < INFLIGHT; k++) v[k] = p[k * 64];
#pragma unroll
        for (int k = 0; k < INFLIGHT; k++) acc += v[k];
    }
    out[blockIdx.x * 64 + threadIdx.x] = acc;
}

template <class F>
void timeit(const char* name, long bytes, F launch) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float best = 1e9;
    for (int rep = 0; rep < 5; rep++) {
        hipEventRecord(e0);
        launch();
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
    }
    printf("%s: %ld bytes, best %.3f ms = %.2f TB/s\n", name, bytes, best, bytes / (best * 1e-3) / 1e12);
}

int main() {
    const long total = 1L << 31;  // 2 GiB
    double *in, *out;
    hipMalloc(&in, total); hipMalloc(&out, 16384 * 64 * 8);
    hipMemset(in, 0, total);
    for (long blocks : {1024L, 2048L, 4096L, 8192L}) {
        long npw8 = total / 8 / 64 / blocks, npw16 = total / 16 / 64 / blocks;
        char nm[128];
        snprintf(nm, 128, "read8  blocks=%ld inflight=4 ", blocks); timeit(nm, total, [&] { read8_kernel<4><<<blocks, 64>>>(in, out, npw8); });
        snprintf(nm, 128, "read8  blocks=%ld inflight=16", blocks); timeit(nm, total, [&] { read8_kernel<16><<<blocks, 64>>>(in, out, npw8); });
        snprintf(nm, 128, "front8 blocks=%ld inflight=4 ", blocks); timeit(nm, total, [&] { read8_front_kernel<4><<<blocks, 64>>>(in, out, npw8); });
        snprintf(nm, 128, "front8 blocks=%ld inflight=16", blocks); timeit(nm, total, [&] { read8_front_kernel<16><<<blocks, 64>>>(in, out, npw8); });
        snprintf(nm, 128, "read16 blocks=%ld inflight=4 ", blocks); timeit(nm, total, [&] { read16_kernel<4><<<blocks, 64>>>((double2*)in, out, npw16); });
        snprintf(nm, 128, "read16 blocks=%ld inflight=16", blocks); timeit(nm, total, [&] { read16_kernel<16><<<blocks, 64>>>((double2*)in, out, npw16); });
    }
    // (4) does the rate depend on the size of the launch (ramp-up and tail of a 0.3 ms kernel)?
    hipFree(in);
    const long big = 1L << 33;  // 8 GiB
    hipMalloc(&in, big);
    hipMemset(in, 0, big);
    for (long tot : {1L << 30, 1L << 31, 1L << 32, 1L << 33}) {
        char nm[128];
        snprintf(nm, 128, "read8  blocks=2048 inflight=16 total=%ld MiB", tot >> 20);
        timeit(nm, tot, [&] { read8_kernel<16><<<2048, 64>>>(in, out, tot / 8 / 64 / 2048); });
        snprintf(nm, 128, "read8  blocks=16384 inflight=8 total=%ld MiB", tot >> 20);
        timeit(nm, tot, [&] { read8_kernel<8><<<16384, 64>>>(in, out, tot / 8 / 64 / 16384); });
    }
    return 0;
}
