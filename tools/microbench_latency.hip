// tools/microbench_latency.hip -- where the fixed latency of one synchronous evaluation goes, and which ways round it
// the runtime offers.  One "evaluation" = [a 12.8-KB gain table reaches the GPU] -> main kernel (240 workgroups that
// spin for a given time; the first 40 read the table) -> finalize kernel (6 doubles) -> [the 6 doubles reach the host].
// Variants of the two bracketed transfers, of the stream and of the launch call are timed against the kernel's own
// spin time; host time spent inside the enqueue calls is reported next to the total.
//     hipcc -O2 --offload-arch=gfx950 -o /tmp/mb_lat tools/microbench_latency.hip && /tmp/mb_lat
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>

#include <chrono>
#include <cstddef>
#include <cstdio>
#include <cstring>
#include <vector>

#define CK(x) do { hipError_t e__ = (x); if (e__ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e__)); exit(1); } } while (0)

constexpr int TAB = 1600;            // doubles: 100 rows x 16

struct ArgsPtr { const double* tab; double* tmp; long long cycles; int readers; };
struct ArgsVal { double* tmp; long long cycles; int readers; int pad; double tab[TAB]; };

__device__ __forceinline__ double spin(double acc, long long cycles) {
    const long long t0 = wall_clock64();
    while (wall_clock64() - t0 < cycles) acc += 1e-9;
    return acc;
}

// table through a pointer (device memory, or pinned host memory read in place)
__global__ __launch_bounds__(256) void main_ptr(const ArgsPtr A) {
    double acc = 0.0;
    if ((int)blockIdx.x < A.readers) {
        const int lane = threadIdx.x & 63;
        for (int r = 0; r < TAB; r += 64) acc += A.tab[r + lane];
    }
    acc = spin(acc, A.cycles);
    if (threadIdx.x == 0) A.tmp[blockIdx.x] = acc;
}

// table by value in the kernel argument segment, read through the segment pointer (dynamic, wave-uniform index)
__global__ __launch_bounds__(256) void main_val(const ArgsVal A) {
    double acc = 0.0;
    if ((int)blockIdx.x < A.readers) {
        const double __attribute__((address_space(4)))* t =
            (const double __attribute__((address_space(4)))*)((const char __attribute__((address_space(4)))*)__builtin_amdgcn_kernarg_segment_ptr() + offsetof(ArgsVal, tab));
        for (int r = 0; r < TAB / 16; r++) {
            const int row = __builtin_amdgcn_readfirstlane(r);
            acc += t[row * 16] + t[row * 16 + 5] + t[row * 16 + 12];
        }
    }
    acc = spin(acc, A.cycles);
    if (threadIdx.x == 0) A.tmp[blockIdx.x] = acc;
}

// finalize: 6 doubles to `out` (device or pinned host memory); optionally a sequence word the host spins on
__global__ void fin_kernel(const double* tmp, double* out, unsigned long long* flag, unsigned long long seq) {
    const double v = tmp[threadIdx.x] + tmp[64 + threadIdx.x];
    double s = v;
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
    if (threadIdx.x < 6) out[threadIdx.x] = s + threadIdx.x;
    if (flag) {
        __threadfence_system();
        if (threadIdx.x == 0) __hip_atomic_store(flag, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

static double now_us() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

enum Src { SRC_COPY, SRC_KERNARG, SRC_PINNED };
enum Dst { DST_BLOCKING_COPY, DST_ASYNC_COPY_SYNC, DST_PINNED_SYNC, DST_PINNED_EVENT, DST_PINNED_SPIN, DST_PINNED_SPIN_NOFIN };
static const char* src_name[] = {"H2D copy", "kernarg", "pinned in place"};
static const char* dst_name[] = {"blocking hipMemcpy", "async copy + stream sync", "kernel->pinned + stream sync", "kernel->pinned + event sync",
                                 "kernel->pinned + host spin", "ONE kernel ->pinned + host spin"};

int main(int argc, char** argv) {
    const double kernel_us = argc > 1 ? atof(argv[1]) : 40.0;
    const int reps = 400;
    double *tab_pinned, *out_pinned, *tab_dev, *tmp_dev, *out_dev;
    unsigned long long* flag_pinned;
    CK(hipHostMalloc(&tab_pinned, TAB * 8));
    CK(hipHostMalloc(&out_pinned, 64));
    CK(hipHostMalloc(&flag_pinned, 64));
    CK(hipMalloc(&tab_dev, TAB * 8));
    CK(hipMalloc(&tmp_dev, 4096 * 8));
    CK(hipMalloc(&out_dev, 64));
    CK(hipMemset(tmp_dev, 0, 4096 * 8));
    for (int i = 0; i < TAB; i++) tab_pinned[i] = 1e-3 * i;
    *flag_pinned = 0;
    hipStream_t own;
    CK(hipStreamCreateWithFlags(&own, hipStreamNonBlocking));
    hipEvent_t ev0, ev1, evs;
    CK(hipEventCreate(&ev0)); CK(hipEventCreate(&ev1));
    CK(hipEventCreateWithFlags(&evs, hipEventDisableTiming));
    const long long cycles = (long long)(kernel_us * 100.0);
    static ArgsVal av;
    unsigned long long seq = 0;

    printf("main kernel spins %.0f us; 240 workgroups, 40 read the 12.8-KB table; %d evaluations per line\n", kernel_us, reps);
    printf("%-16s | %-34s | %-5s | %-6s | %9s | %9s | %9s\n", "table", "result", "strm", "stamps", "total us", "overhead", "host enq");
    auto run = [&](Src src, Dst dst, bool own_stream, bool stamps, int readers) {
        hipStream_t s = own_stream ? own : 0;
        double host_out[6];
        double enq = 0.0, total = 0.0;
        for (int it = -30; it < reps; it++) {
            if (it == 0) { CK(hipDeviceSynchronize()); enq = 0.0; total = now_us(); }
            const double t_in = now_us();
            tab_pinned[0] = it;
            seq++;
            const bool one_kernel = dst == DST_PINNED_SPIN_NOFIN;
            double* out = (dst == DST_BLOCKING_COPY || dst == DST_ASYNC_COPY_SYNC) ? out_dev : out_pinned;
            if (src == SRC_KERNARG) {
                av.tmp = one_kernel ? out_pinned : tmp_dev; av.cycles = cycles; av.readers = readers;
                memcpy(av.tab, tab_pinned, TAB * 8);
                if (stamps) hipExtLaunchKernelGGL(main_val, dim3(240), dim3(256), 0, s, ev0, ev1, 0, av);
                else hipLaunchKernelGGL(main_val, dim3(240), dim3(256), 0, s, av);
            } else {
                ArgsPtr ap;
                ap.tmp = one_kernel ? out_pinned : tmp_dev; ap.cycles = cycles; ap.readers = readers;
                if (src == SRC_COPY) { CK(hipMemcpyAsync(tab_dev, tab_pinned, TAB * 8, hipMemcpyHostToDevice, s)); ap.tab = tab_dev; }
                else ap.tab = tab_pinned;
                if (stamps) hipExtLaunchKernelGGL(main_ptr, dim3(240), dim3(256), 0, s, ev0, ev1, 0, ap);
                else hipLaunchKernelGGL(main_ptr, dim3(240), dim3(256), 0, s, ap);
            }
            const bool spin_wait = dst == DST_PINNED_SPIN;
            if (!one_kernel) hipLaunchKernelGGL(fin_kernel, dim3(1), dim3(64), 0, s, tmp_dev, out, spin_wait ? flag_pinned : nullptr, seq);
            const double t_enq = now_us();
            switch (dst) {
            case DST_BLOCKING_COPY: CK(hipMemcpy(host_out, out_dev, 48, hipMemcpyDeviceToHost)); break;
            case DST_ASYNC_COPY_SYNC:
                CK(hipMemcpyAsync(out_pinned, out_dev, 48, hipMemcpyDeviceToHost, s));
                CK(hipStreamSynchronize(s)); memcpy(host_out, out_pinned, 48); break;
            case DST_PINNED_SYNC: CK(hipStreamSynchronize(s)); memcpy(host_out, out_pinned, 48); break;
            case DST_PINNED_EVENT: CK(hipEventRecord(evs, s)); CK(hipEventSynchronize(evs)); memcpy(host_out, out_pinned, 48); break;
            case DST_PINNED_SPIN:
                while (__atomic_load_n(flag_pinned, __ATOMIC_ACQUIRE) != seq) { }
                memcpy(host_out, out_pinned, 48); break;
            case DST_PINNED_SPIN_NOFIN:
                // the main kernel's thread 0 of block 0 wrote out_pinned[0]; completion still through a stream sync
                CK(hipStreamSynchronize(s)); memcpy(host_out, out_pinned, 48); break;
            }
            enq += t_enq - t_in;
        }
        if (dst == DST_PINNED_SPIN) CK(hipDeviceSynchronize());
        total = (now_us() - total) / reps;
        printf("%-16s | %-34s | %-5s | %-6s | %9.1f | %9.1f | %9.1f\n", src_name[src], dst_name[dst], own_stream ? "own" : "null",
               stamps ? "yes" : "no", total, total - kernel_us, enq / reps);
        fflush(stdout);
    };
    run(SRC_COPY, DST_BLOCKING_COPY, false, true, 40);          // what ssde_eval does today
    run(SRC_COPY, DST_BLOCKING_COPY, false, false, 40);
    run(SRC_COPY, DST_BLOCKING_COPY, true, false, 40);
    run(SRC_KERNARG, DST_BLOCKING_COPY, false, true, 40);
    run(SRC_KERNARG, DST_BLOCKING_COPY, false, false, 40);
    run(SRC_KERNARG, DST_BLOCKING_COPY, true, false, 40);
    run(SRC_PINNED, DST_BLOCKING_COPY, false, false, 40);
    run(SRC_PINNED, DST_BLOCKING_COPY, false, false, 5);
    run(SRC_KERNARG, DST_ASYNC_COPY_SYNC, true, false, 40);
    run(SRC_KERNARG, DST_PINNED_SYNC, false, false, 40);
    run(SRC_KERNARG, DST_PINNED_SYNC, true, false, 40);
    run(SRC_KERNARG, DST_PINNED_EVENT, true, false, 40);
    run(SRC_KERNARG, DST_PINNED_SPIN, false, false, 40);
    run(SRC_KERNARG, DST_PINNED_SPIN, true, false, 40);
    run(SRC_KERNARG, DST_PINNED_SPIN, true, true, 40);
    run(SRC_KERNARG, DST_PINNED_SPIN_NOFIN, true, false, 40);
    run(SRC_COPY, DST_PINNED_SPIN, true, false, 40);
    return 0;
}
