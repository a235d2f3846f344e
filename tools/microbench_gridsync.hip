// microbench_gridsync.hip -- what ONE cooperative launch with two grid-wide barriers costs against a replayed hipGraph of three
// dependent kernels, per synchronous "evaluation" (the question behind fusing C1's pre-pass / filter / finalising launch into one
// kernel: DESIGN.md 3.4b).  Every phase writes a word per block that the next phase reads from ANOTHER block, so the barriers
// carry the same cross-XCD visibility the real kernels would need.   (experiment, not product code)
//   hipcc -O3 --offload-arch=gfx950 -o build/mb_gridsync tools/microbench_gridsync.hip && build/mb_gridsync
#include <hip/hip_runtime.h>
#include <hip/hip_cooperative_groups.h>
#include <chrono>
#include <cstdio>
namespace cg = cooperative_groups;

__device__ __forceinline__ void spin(int cycles) {
    const long long t0 = wall_clock64();
    while (wall_clock64() - t0 < cycles) {}
}
__global__ __launch_bounds__(256) void phase_kernel(double* buf, int phase, int work) {
    const int nb = gridDim.x, b = blockIdx.x;
    double v = phase == 0 ? 1.0 : buf[(phase - 1) * nb + (b + 1) % nb];
    spin(work);
    if (threadIdx.x == 0) buf[phase * nb + b] = v + 1.0;
}
__global__ __launch_bounds__(256) void fused_kernel(double* buf, int work0, int work1, int work2) {
    cg::grid_group grid = cg::this_grid();
    const int nb = gridDim.x, b = blockIdx.x;
    spin(work0);
    if (threadIdx.x == 0) buf[b] = 2.0;
    __threadfence();
    grid.sync();
    double v = buf[(b + 1) % nb];
    spin(work1);
    if (threadIdx.x == 0) buf[nb + b] = v + 1.0;
    __threadfence();
    grid.sync();
    v = buf[nb + (b + 1) % nb];
    spin(work2);
    if (threadIdx.x == 0) buf[2 * nb + b] = v + 1.0;
}
#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
int main() {
    using clk = std::chrono::steady_clock;
    double* buf; CHK(hipMalloc(&buf, 3 * 1024 * 8));
    double* pin; CHK(hipHostMalloc(&pin, 3 * 1024 * 8));
    hipStream_t s; CHK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    // wall_clock64 ticks at 100 MHz: 10 us of "work" = 1000 ticks
    const int w0 = 500, w1 = 3300, w2 = 500;        // 5 us, 33 us, 5 us: C1's three launches without their launch floors
    for (int nb : {58, 128, 256}) {
        // --- three dependent kernels replayed from a graph + read-back copy
        hipGraph_t g; hipGraphExec_t ge;
        CHK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
        hipLaunchKernelGGL(phase_kernel, dim3(nb), dim3(256), 0, s, buf, 0, w0);
        hipLaunchKernelGGL(phase_kernel, dim3(nb), dim3(256), 0, s, buf, 1, w1);
        hipLaunchKernelGGL(phase_kernel, dim3(nb), dim3(256), 0, s, buf, 2, w2);
        CHK(hipStreamEndCapture(s, &g));
        CHK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
        for (int k = 0; k < 20; k++) { CHK(hipGraphLaunch(ge, s)); CHK(hipStreamSynchronize(s)); }
        auto t0 = clk::now();
        const int reps = 300;
        for (int k = 0; k < reps; k++) { CHK(hipGraphLaunch(ge, s)); CHK(hipStreamSynchronize(s)); }
        const double graph_us = std::chrono::duration<double, std::micro>(clk::now() - t0).count() / reps;
        // --- one cooperative launch with two grid barriers
        int a0 = w0, a1 = w1, a2 = w2;
        void* args[] = {&buf, &a0, &a1, &a2};
        hipError_t e = hipLaunchCooperativeKernel((const void*)fused_kernel, dim3(nb), dim3(256), args, 0, s);
        if (e != hipSuccess) { printf("blocks %d: cooperative launch refused: %s\n", nb, hipGetErrorString(e)); continue; }
        CHK(hipStreamSynchronize(s));
        for (int k = 0; k < 20; k++) { CHK(hipLaunchCooperativeKernel((const void*)fused_kernel, dim3(nb), dim3(256), args, 0, s)); CHK(hipStreamSynchronize(s)); }
        t0 = clk::now();
        for (int k = 0; k < reps; k++) { CHK(hipLaunchCooperativeKernel((const void*)fused_kernel, dim3(nb), dim3(256), args, 0, s)); CHK(hipStreamSynchronize(s)); }
        const double coop_us = std::chrono::duration<double, std::micro>(clk::now() - t0).count() / reps;
        // --- the same work in ONE plain kernel without barriers (lower bound: launch + 43 us of spinning)
        hipLaunchKernelGGL(phase_kernel, dim3(nb), dim3(256), 0, s, buf, 0, w0 + w1 + w2);
        CHK(hipStreamSynchronize(s));
        t0 = clk::now();
        for (int k = 0; k < reps; k++) { hipLaunchKernelGGL(phase_kernel, dim3(nb), dim3(256), 0, s, buf, 0, w0 + w1 + w2); CHK(hipStreamSynchronize(s)); }
        const double one_us = std::chrono::duration<double, std::micro>(clk::now() - t0).count() / reps;
        CHK(hipMemcpy(pin, buf, 3 * nb * 8, hipMemcpyDeviceToHost));
        printf("blocks %3d: graph of three kernels %.1f us, one cooperative kernel with two grid barriers %.1f us, one plain kernel (no barriers) %.1f us per synchronous call (43 us of work in each); check %.0f\n",
               nb, graph_us, coop_us, one_us, pin[2 * nb]);
        (void)hipGraphExecDestroy(ge); (void)hipGraphDestroy(g);
    }
    return 0;
}
