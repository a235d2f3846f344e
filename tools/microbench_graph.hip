// tools/microbench_graph.hip -- what a hipGraph could save on the fixed latency of one synchronous evaluation of the
// shared-covariance path: the four stream operations of ssde_eval -- H2D copy of the gain table, main
// kernel, finalize kernel, D2H read-back of 48 bytes -- issued one by one (as the engine does) against one hipGraphLaunch
// of the captured sequence.  The kernels spin for a given number of microseconds so that the launch overheads are what
// differs.      hipcc -O2 --offload-arch=gfx950 -o /tmp/mb_graph tools/microbench_graph.hip && /tmp/mb_graph
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdio>
#include <cstring>
#include <vector>

#define CK(x) do { hipError_t e__ = (x); if (e__ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e__)); return 1; } } while (0)

__global__ void spin_kernel(const double* tab, double* out, long long cycles) {
    const long long t0 = wall_clock64();
    double acc = tab[threadIdx.x & 63];
    while (wall_clock64() - t0 < cycles) acc += 1e-9;
    if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = acc;
}
__global__ void fin_kernel(const double* in, double* out) {
    if (threadIdx.x < 6) out[threadIdx.x] = in[0] + threadIdx.x;
}

static double now_us() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main() {
    const size_t tab_bytes = 20 * 1024;
    double *tab_pinned, *out_pinned, *tab_dev, *tmp_dev, *out_dev;
    CK(hipHostMalloc(&tab_pinned, tab_bytes));
    CK(hipHostMalloc(&out_pinned, 64));
    CK(hipMalloc(&tab_dev, tab_bytes));
    CK(hipMalloc(&tmp_dev, 64));
    CK(hipMalloc(&out_dev, 64));
    memset(tab_pinned, 0, tab_bytes);
    hipStream_t s;
    CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    const int reps = 300;
    for (double kernel_us : {50.0, 280.0}) {
        const long long cycles = (long long)(kernel_us * 100.0);      // wall_clock64 ticks at 100 MHz
        // (a) one by one on the null stream, blocking read-back: what ssde_eval does
        double host_out[6];
        for (int it = 0; it < 20; it++) {
            CK(hipMemcpyAsync(tab_dev, tab_pinned, tab_bytes, hipMemcpyHostToDevice, 0));
            hipLaunchKernelGGL(spin_kernel, dim3(240), dim3(256), 0, 0, tab_dev, tmp_dev, cycles);
            hipLaunchKernelGGL(fin_kernel, dim3(1), dim3(64), 0, 0, tmp_dev, out_dev);
            CK(hipMemcpy(host_out, out_dev, 48, hipMemcpyDeviceToHost));
        }
        double t0 = now_us();
        for (int it = 0; it < reps; it++) {
            tab_pinned[0] = it;
            CK(hipMemcpyAsync(tab_dev, tab_pinned, tab_bytes, hipMemcpyHostToDevice, 0));
            hipLaunchKernelGGL(spin_kernel, dim3(240), dim3(256), 0, 0, tab_dev, tmp_dev, cycles);
            hipLaunchKernelGGL(fin_kernel, dim3(1), dim3(64), 0, 0, tmp_dev, out_dev);
            CK(hipMemcpy(host_out, out_dev, 48, hipMemcpyDeviceToHost));
        }
        const double plain = (now_us() - t0) / reps;
        // (b) the same four operations captured once, replayed with one launch + one synchronisation
        hipGraph_t g;
        hipGraphExec_t ge;
        CK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
        CK(hipMemcpyAsync(tab_dev, tab_pinned, tab_bytes, hipMemcpyHostToDevice, s));
        hipLaunchKernelGGL(spin_kernel, dim3(240), dim3(256), 0, s, tab_dev, tmp_dev, cycles);
        hipLaunchKernelGGL(fin_kernel, dim3(1), dim3(64), 0, s, tmp_dev, out_dev);
        CK(hipMemcpyAsync(out_pinned, out_dev, 48, hipMemcpyDeviceToHost, s));
        CK(hipStreamEndCapture(s, &g));
        CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
        for (int it = 0; it < 20; it++) { CK(hipGraphLaunch(ge, s)); CK(hipStreamSynchronize(s)); }
        t0 = now_us();
        for (int it = 0; it < reps; it++) {
            tab_pinned[0] = it;
            CK(hipGraphLaunch(ge, s));
            CK(hipStreamSynchronize(s));
            memcpy(host_out, out_pinned, 48);
        }
        const double graph = (now_us() - t0) / reps;
        // (c) one by one on a non-blocking stream with an asynchronous read-back into pinned memory + synchronisation
        t0 = now_us();
        for (int it = 0; it < reps; it++) {
            tab_pinned[0] = it;
            CK(hipMemcpyAsync(tab_dev, tab_pinned, tab_bytes, hipMemcpyHostToDevice, s));
            hipLaunchKernelGGL(spin_kernel, dim3(240), dim3(256), 0, s, tab_dev, tmp_dev, cycles);
            hipLaunchKernelGGL(fin_kernel, dim3(1), dim3(64), 0, s, tmp_dev, out_dev);
            CK(hipMemcpyAsync(out_pinned, out_dev, 48, hipMemcpyDeviceToHost, s));
            CK(hipStreamSynchronize(s));
            memcpy(host_out, out_pinned, 48);
        }
        const double async_ = (now_us() - t0) / reps;
        printf("kernel %.0f us: one by one + blocking copy %.1f us | graph replay %.1f us | one by one + async copy + sync %.1f us   (overheads %.1f / %.1f / %.1f)\n",
               kernel_us, plain, graph, async_, plain - kernel_us, graph - kernel_us, async_ - kernel_us);
        (void)hipGraphExecDestroy(ge);
        (void)hipGraphDestroy(g);
    }
    return 0;
}
