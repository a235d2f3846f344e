// tools/microbench_occupancy.hip -- do two 256-thread workgroups of a register-heavy kernel share a CU?  512 workgroups
// (two per CU) each spin for 20 us: if the launch takes ~20 us they were co-resident, ~40 us means one per CU at a time.
// Variants: live VGPRs per lane (held across the spin), static LDS per workgroup, scalar-register pressure.
//     hipcc -O2 --offload-arch=gfx950 -o /tmp/mb_occ tools/microbench_occupancy.hip && /tmp/mb_occ
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>

#define CK(x) do { hipError_t e__ = (x); if (e__ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e__)); exit(1); } } while (0)

template <int NV, int LDS_DOUBLES>
__global__ __launch_bounds__(256, 1) void occ_kernel(const double* in, double* out, long long cycles, unsigned* census) {
    __shared__ double lds[LDS_DOUBLES > 0 ? LDS_DOUBLES : 1];
    double v[NV];
#pragma unroll
    for (int k = 0; k < NV; k++) v[k] = in[(threadIdx.x + 67 * k) & 1023];
    if (LDS_DOUBLES > 0) lds[threadIdx.x] = v[0];
    const long long t0 = wall_clock64();
    while (wall_clock64() - t0 < cycles) {
#pragma unroll
        for (int k = 0; k < NV; k++) v[k] = fma(v[k], 1.0000001, 1e-9);      // every value stays live across the spin
    }
    double s = LDS_DOUBLES > 0 ? lds[(threadIdx.x + 1) & 255] : 0.0;
#pragma unroll
    for (int k = 0; k < NV; k++) s += v[k];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0) {
        unsigned hw;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
        census[blockIdx.x] = hw;
    }
}

static double now_us() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

template <int NV, int LDS_DOUBLES>
static void run(const double* in, double* out, unsigned* census, int wgs) {
    const long long cycles = 2000;       // 20 us at 100 MHz
    for (int it = 0; it < 3; it++) hipLaunchKernelGGL((occ_kernel<NV, LDS_DOUBLES>), dim3(wgs), dim3(256), 0, 0, in, out, cycles, census);
    CK(hipDeviceSynchronize());
    const double t0 = now_us();
    for (int it = 0; it < 20; it++) hipLaunchKernelGGL((occ_kernel<NV, LDS_DOUBLES>), dim3(wgs), dim3(256), 0, 0, in, out, cycles, census);
    CK(hipDeviceSynchronize());
    const double per = (now_us() - t0) / 20;
    hipFuncAttributes fa;
    CK(hipFuncGetAttributes(&fa, (const void*)occ_kernel<NV, LDS_DOUBLES>));
    int occ = 0;
    CK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, (const void*)occ_kernel<NV, LDS_DOUBLES>, 256, 0));
    printf("live doubles %3d  LDS %6d B  regs/lane %3d  wgs %4d : %6.1f us per launch (20-us spin)  occupancy API %d blocks/CU\n", NV,
           LDS_DOUBLES * 8, fa.numRegs, wgs, per, occ);
}

int main() {
    double *in, *out;
    unsigned* census;
    CK(hipMalloc(&in, 1024 * 8));
    CK(hipMalloc(&out, 2048 * 256 * 8));
    CK(hipMalloc(&census, 2048 * 4));
    CK(hipMemset(in, 0, 1024 * 8));
    for (int wgs : {256, 512, 1024}) {
        run<16, 0>(in, out, census, wgs);
        run<100, 0>(in, out, census, wgs);
        run<100, 4096>(in, out, census, wgs);
        run<115, 4096>(in, out, census, wgs);
        run<60, 4096>(in, out, census, wgs);
    }
    return 0;
}
