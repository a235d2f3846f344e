// microbench_bar.hip -- can the host write straight into fine-grained device memory (large BAR), and what does it
// cost compared with hipMemcpyAsync of the same 8 KB before a dependent kernel?   (experiment, not product code)
#include <hip/hip_runtime.h>
#include <chrono>
#include <csetjmp>
#include <csignal>
#include <cstdio>
#include <cstring>
static sigjmp_buf jb;
static void on_segv(int) { siglongjmp(jb, 1); }
__global__ void sum_kernel(const double* t, int n, double* out) {
    double a = 0;
    for (int i = threadIdx.x; i < n; i += 64) a += t[i];
    for (int o = 32; o > 0; o >>= 1) a += __shfl_xor(a, o, 64);
    if (threadIdx.x == 0) *out = a;
}
int main() {
    const int n = 1024;
    double *fg = nullptr, *dev = nullptr, *out = nullptr, *pin = nullptr;
    hipError_t e = hipExtMallocWithFlags((void**)&fg, n * 8, hipDeviceMallocFinegrained);
    printf("hipExtMallocWithFlags(finegrained): %s\n", hipGetErrorString(e));
    hipMalloc(&dev, n * 8); hipMalloc(&out, 8); hipHostMalloc(&pin, n * 8);
    for (int i = 0; i < n; i++) pin[i] = i;
    signal(SIGSEGV, on_segv); signal(SIGBUS, on_segv);
    bool host_ok = false;
    if (e == hipSuccess && sigsetjmp(jb, 1) == 0) { for (int i = 0; i < n; i++) fg[i] = pin[i]; host_ok = true; }
    printf("host write to fine-grained device memory: %s\n", host_ok ? "ok" : "FAULT");
    double r = -1;
    using clk = std::chrono::steady_clock;
    for (int variant = 0; variant < 2; variant++) {
        if (variant == 1 && !host_ok) break;
        double best = 1e9, res = 0;
        for (int rep = 0; rep < 200; rep++) {
            for (int i = 0; i < n; i++) pin[i] = i + rep;
            hipDeviceSynchronize();
            auto t0 = clk::now();
            if (variant == 0) { hipMemcpyAsync(dev, pin, n * 8, hipMemcpyHostToDevice, 0); hipLaunchKernelGGL(sum_kernel, 1, 64, 0, 0, dev, n, out); }
            else { memcpy(fg, pin, n * 8); __sync_synchronize(); hipLaunchKernelGGL(sum_kernel, 1, 64, 0, 0, fg, n, out); }
            hipMemcpy(&r, out, 8, hipMemcpyDeviceToHost);
            double us = std::chrono::duration<double, std::micro>(clk::now() - t0).count();
            if (us < best) best = us;
            res = r;
            const double want = (double)n * (n - 1) / 2 + (double)n * rep;
            if (r != want) { printf("variant %d rep %d WRONG %.1f vs %.1f\n", variant, rep, r, want); break; }
        }
        printf("%s: best %.1f us (last sum %.0f)\n", variant == 0 ? "hipMemcpyAsync + kernel + D2H" : "BAR write + kernel + D2H", best, res);
    }
    return 0;
}
