// tools/microbench_fp64.hip -- calibrates the fp64 VALU issue rate that bounds the register
// kernels: dependent / independent v_fma_f64 chains, 1 or 2 waves per SIMD, and the clock the
// chip holds under that load (s_memtime vs s_memrealtime, MI355X_MICROARCH.md "DVFS give-back").
//   hipcc --offload-arch=gfx950 -O3 tools/microbench_fp64.hip -o build/microbench_fp64 && build/microbench_fp64
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

template <int NCHAIN>
__global__ __launch_bounds__(64) void fma_kernel(double* out, int iters, unsigned long long* clk) {
    double a[NCHAIN];
    const double x = 1.0 + 1e-9 * threadIdx.x, y = 1e-9 * blockIdx.x;
#pragma unroll
    for (int k = 0; k < NCHAIN; k++) a[k] = k * 0.5;
    unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int r = 0; r < 8; r++)
#pragma unroll
            for (int k = 0; k < NCHAIN; k++) a[k] = fma(a[k], x, y);
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    double s = 0;
#pragma unroll
    for (int k = 0; k < NCHAIN; k++) s += a[k];
    out[blockIdx.x * 64 + threadIdx.x] = s;
    if (threadIdx.x == 0) { clk[2 * blockIdx.x] = t1 - t0; clk[2 * blockIdx.x + 1] = r1 - r0; }
}

template <int NCHAIN>
void run(int blocks, int iters) {
    double* out; unsigned long long* clk;
    hipMalloc(&out, blocks * 64 * 8); hipMalloc(&clk, blocks * 16);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    fma_kernel<NCHAIN><<<blocks, 64>>>(out, iters, clk);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    fma_kernel<NCHAIN><<<blocks, 64>>>(out, iters, clk);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> h(2 * blocks);
    hipMemcpy(h.data(), clk, blocks * 16, hipMemcpyDeviceToHost);
    double ghz = (double)h[0] / (double)h[1] * 0.1;  // s_memrealtime ticks at 100 MHz
    double ninstr = (double)iters * 8 * NCHAIN;      // wave-instructions per wave
    double per_instr_cycles = (double)h[0] / ninstr;
    double tflops = (double)blocks * 64 * ninstr * 2 / (ms * 1e-3) / 1e12;
    printf("chains=%d blocks=%d: %.3f ms, %.2f TFLOP/s fp64, in-kernel clock %.2f GHz, %.2f cycles per wave-instr (one wave's view)\n",
           NCHAIN, blocks, ms, tflops, ghz, per_instr_cycles);
    hipFree(out); hipFree(clk);
}

int main() {
    for (int blocks : {1024, 2048, 4096}) {
        run<1>(blocks, 20000);
        run<2>(blocks, 20000);
        run<4>(blocks, 10000);
        run<8>(blocks, 10000);
    }
    return 0;
}
