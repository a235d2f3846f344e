// tools/microbench_fp64_sgpr.hip -- does an SGPR (scalar) source operand change the fp64 FMA issue rate?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

template <int NCHAIN, int MODE>
__global__ __launch_bounds__(64) void fma_kernel(double* out, int iters, double xs, double ys, unsigned long long* clk) {
    double a[NCHAIN];
    const double xv = 1.0 + 1e-9 * threadIdx.x, yv = 1e-9 * blockIdx.x;
#pragma unroll
    for (int k = 0; k < NCHAIN; k++) a[k] = k * 0.5 + 1e-3 * threadIdx.x;
    unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int r = 0; r < 8; r++)
#pragma unroll
            for (int k = 0; k < NCHAIN; k++) {
                if (MODE == 0) a[k] = fma(a[k], xv, yv);        // all VGPR
                else if (MODE == 1) a[k] = fma(a[k], xs, yv);   // SGPR multiplier
                else if (MODE == 2) a[k] = fma(xs, a[(k + 1) % NCHAIN], a[k]);  // fmac form: acc += s * other chain
                else a[k] = a[k] * xs + ys;                     // two SGPRs (needs a mov or two ops)
            }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    double s = 0;
#pragma unroll
    for (int k = 0; k < NCHAIN; k++) s += a[k];
    out[blockIdx.x * 64 + threadIdx.x] = s;
    if (threadIdx.x == 0) { clk[2 * blockIdx.x] = t1 - t0; clk[2 * blockIdx.x + 1] = r1 - r0; }
}

template <int NCHAIN, int MODE>
void run(int blocks, int iters) {
    double* out; unsigned long long* clk;
    hipMalloc(&out, blocks * 64 * 8); hipMalloc(&clk, blocks * 16);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    fma_kernel<NCHAIN, MODE><<<blocks, 64>>>(out, iters, 0.999999, 1e-7, clk);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    fma_kernel<NCHAIN, MODE><<<blocks, 64>>>(out, iters, 0.999999, 1e-7, clk);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> h(2 * blocks);
    hipMemcpy(h.data(), clk, blocks * 16, hipMemcpyDeviceToHost);
    double ghz = (double)h[0] / (double)h[1] * 0.1;
    double ninstr = (double)iters * 8 * NCHAIN;
    printf("mode=%d chains=%d blocks=%d: %.3f ms, clock %.2f GHz, %.2f cycles per source-level fma (one wave's view)\n", MODE, NCHAIN,
           blocks, ms, ghz, (double)h[0] / ninstr);
    hipFree(out); hipFree(clk);
}

int main() {
    for (int blocks : {1024, 2048}) {
        run<8, 0>(blocks, 10000);
        run<8, 1>(blocks, 10000);
        run<8, 2>(blocks, 10000);
        run<8, 3>(blocks, 10000);
        run<4, 1>(blocks, 10000);
        run<4, 2>(blocks, 10000);
    }
    return 0;
}
