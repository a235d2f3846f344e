// tools/microbench_fp64_ops.hip -- issue rate of the fp64 VALU operations the filter lanes are made of, on one MI355X:
// v_fma_f64 / v_mul_f64 / v_add_f64 (8 independent chains per lane, and ONE dependent chain), at one and at two waves per SIMD.
// Prints cycles per wave-instruction and SIMD (wall time x 2.4 GHz / instructions issued per SIMD).
//   hipcc -O3 --offload-arch=gfx950 tools/microbench_fp64_ops.hip -o build/tools/microbench_fp64_ops && build/tools/microbench_fp64_ops
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

constexpr int ITER = 2048, CH = 8;

template <int OP, int NCH>
__global__ __launch_bounds__(256) void k(double* out, double a, double b) {
    double x[NCH];
#pragma unroll
    for (int c = 0; c < NCH; c++) x[c] = a + 1e-9 * (threadIdx.x + c);
    for (int i = 0; i < ITER; i++) {
#pragma unroll
        for (int r = 0; r < CH / NCH * 4; r++)
#pragma unroll
            for (int c = 0; c < NCH; c++) {
                if (OP == 0) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(x[c]) : "v"(b), "v"(a));
                if (OP == 1) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(x[c]) : "v"(b));
                if (OP == 2) asm volatile("v_add_f64 %0, %0, %1" : "+v"(x[c]) : "v"(a));
                if (OP == 3) asm volatile("v_fmac_f64 %0, %1, %2" : "+v"(x[c]) : "v"(b), "v"(a));
            }
    }
    double s = 0.0;
#pragma unroll
    for (int c = 0; c < NCH; c++) s += x[c];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int OP, int NCH>
static void run(const char* name, int wgs, double* d) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((k<OP, NCH>), dim3(wgs), dim3(256), 0, 0, d, 1.0, 0.999999);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int r = 0; r < 5; r++) hipLaunchKernelGGL((k<OP, NCH>), dim3(wgs), dim3(256), 0, 0, d, 1.0, 0.999999);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    ms /= 5;
    const double inst_per_wave = (double)ITER * CH * 4;
    const double waves_per_simd = wgs * 4.0 / 1024.0;
    const double cyc = ms * 1e-3 * 2.4e9 / (inst_per_wave * waves_per_simd);
    printf("%-22s chains %d, %4d workgroups (%.0f waves/SIMD): %.3f ms, %.2f cycles per wave-instruction and SIMD\n", name, NCH, wgs, waves_per_simd, ms, cyc);
}

int main() {
    double* d;
    hipMalloc(&d, 1024 * 256 * 8);
    for (int wgs : {256, 512, 1024}) {
        run<0, 8>("v_fma_f64", wgs, d);
        run<1, 8>("v_mul_f64", wgs, d);
        run<2, 8>("v_add_f64", wgs, d);
        run<3, 8>("v_fmac_f64", wgs, d);
        run<0, 1>("v_fma_f64 dependent", wgs, d);
        run<1, 1>("v_mul_f64 dependent", wgs, d);
        run<0, 2>("v_fma_f64 2 chains", wgs, d);
    }
    return 0;
}
