// tools/microbench_issue.hip -- what does ONE wave per SIMD pay per row of the stationary transfer-function lanes (ssde_tf.hpp: TfCtcrw,
// two dimensions, three covariance directions: 22 fp64 instructions per row) when nothing but the instruction stream limits it?
//   mode 0: the recursion on register data, constants as kernel arguments (scalar operands), no memory
//   mode 1: the same with the constants moved into vector registers
//   mode 2: two independent items per wave, statement-interleaved, no memory
//   mode 3: mode 0 + the row loads of the engine (16 x 512-B wave loads per 8 rows), ping-pong pair (8 KB in flight), from a buffer that fits L2 / from a 200 MB buffer
//   deep NB: NB register blocks in rotation, NB - 1 in flight, 200 MB (re-read every launch: Infinity Cache) and 1.6 GB (HBM)
// Prints cycles per row (s_memtime) and the wall time per launch; 256 workgroups of 4 waves = one wave per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

struct K { double e, nd1, nd2, cm0, cm1; };

template <bool VREG>
struct Item {
    double yp[2], w1[2], w2[2], r1[2], r2[2], r3[2], acc2, C1, C2, C3;
    __device__ __forceinline__ void init(double s) {
        for (int a = 0; a < 2; a++) { yp[a] = s + a; w1[a] = 0.1 * s; w2[a] = 0.05; r1[a] = 0.01; r2[a] = 0.02; r3[a] = 0.03; }
        acc2 = C1 = C2 = C3 = 0.0;
    }
    __device__ __forceinline__ void step(const double* y, double e, double nd1, double nd2, double cm0, double cm1) {
#pragma unroll
        for (int a = 0; a < 2; a++) {
            const double dy = (y[a] - yp[a]) - (a ? cm1 : cm0);
            yp[a] = y[a];
            const double w0 = fma(nd1, w1[a], fma(nd2, w2[a], dy));
            const double u = fma(-e, w1[a], w0);
            acc2 = fma(u, u, acc2);
            C1 = fma(u, r1[a], C1); C2 = fma(u, r2[a], C2); C3 = fma(u, r3[a], C3);
            const double r0 = fma(nd1, r1[a], fma(nd2, r2[a], w0));
            r3[a] = r2[a]; r2[a] = r1[a]; r1[a] = r0;
            w2[a] = w1[a]; w1[a] = w0;
        }
    }
    __device__ __forceinline__ double sum() const { return acc2 + C1 + C2 + C3 + w1[0] + w1[1] + r1[0] + r1[1]; }
};

template <int MODE>
__global__ __launch_bounds__(256, 1) void k(K c, const double* buf, long long stride_rows, int nblocks8, double* out, unsigned long long* clk) {
    const int lane = threadIdx.x & 63, wave = blockIdx.x * 4 + (threadIdx.x >> 6);
    double e = c.e, nd1 = c.nd1, nd2 = c.nd2, cm0 = c.cm0, cm1 = c.cm1;
    if (MODE == 1) {
        asm volatile("v_mov_b64 %0, %1" : "=v"(e) : "s"(c.e)); asm volatile("v_mov_b64 %0, %1" : "=v"(nd1) : "s"(c.nd1));
        asm volatile("v_mov_b64 %0, %1" : "=v"(nd2) : "s"(c.nd2)); asm volatile("v_mov_b64 %0, %1" : "=v"(cm0) : "s"(c.cm0));
        asm volatile("v_mov_b64 %0, %1" : "=v"(cm1) : "s"(c.cm1));
    }
    Item<false> A, B;
    A.init(1.0 + 1e-3 * lane); B.init(2.0 + 1e-3 * lane);
    const double* p = buf + (long long)wave * stride_rows * 128 + lane;
    double yA[8][2], yB[8][2];
#pragma unroll
    for (int u = 0; u < 8; u++) { yA[u][0] = 1.0 + u; yA[u][1] = 2.0 + u; yB[u][0] = 3.0 + u; yB[u][1] = 4.0 + u; }
    if (MODE >= 3) {
#pragma unroll
        for (int u = 0; u < 8; u++) { yA[u][0] = p[(u * 2) * 64]; yA[u][1] = p[(u * 2 + 1) * 64]; }
    }
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int b = 0; b < nblocks8; b += 2) {
        if (MODE >= 3) {
#pragma unroll
            for (int u = 0; u < 8; u++) { yB[u][0] = p[((b + 1) * 16 + u * 2) * 64]; yB[u][1] = p[((b + 1) * 16 + u * 2 + 1) * 64]; }
        }
#pragma unroll
        for (int u = 0; u < 8; u++) {
            if (MODE == 2) {
                // statement-level interleave of two items
                A.step(yA[u], e, nd1, nd2, cm0, cm1);
                B.step(yB[u], e, nd1, nd2, cm0, cm1);
            } else {
                A.step(yA[u], e, nd1, nd2, cm0, cm1);
            }
            if (MODE < 3) { yA[u][0] += 0.5; yA[u][1] -= 0.25; }      // (fresh data without memory: two more adds per row)
        }
        if (MODE >= 3) {
#pragma unroll
            for (int u = 0; u < 8; u++) { yA[u][0] = p[((b + 2) * 16 + u * 2) * 64]; yA[u][1] = p[((b + 2) * 16 + u * 2 + 1) * 64]; }
        }
#pragma unroll
        for (int u = 0; u < 8; u++) {
            if (MODE == 2) { A.step(yB[u], e, nd1, nd2, cm0, cm1); B.step(yA[u], e, nd1, nd2, cm0, cm1); }
            else A.step(MODE >= 3 ? yB[u] : yA[u], e, nd1, nd2, cm0, cm1);
            if (MODE < 3) { yB[u][0] += 0.5; yB[u][1] -= 0.25; }
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    out[wave * 64 + lane] = A.sum() + (MODE == 2 ? B.sum() : 0.0);
    if (lane == 0) clk[wave] = t1 - t0;
}

// the same lanes with NB register blocks of 8 rows in rotation: NB - 1 blocks (8 KB each per wave) in flight
template <int NB>
__global__ __launch_bounds__(256, 1) void kdeep(K c, const double* buf, long long stride_rows, int nblocks8, double* out, unsigned long long* clk) {
    const int lane = threadIdx.x & 63, wave = blockIdx.x * 4 + (threadIdx.x >> 6);
    const double e = c.e, nd1 = c.nd1, nd2 = c.nd2, cm0 = c.cm0, cm1 = c.cm1;
    Item<false> A;
    A.init(1.0 + 1e-3 * lane);
    const double* p = buf + (long long)wave * stride_rows * 128 + lane;
    double y[NB][8][2];
#pragma unroll
    for (int j = 0; j < NB - 1; j++)
#pragma unroll
        for (int u = 0; u < 8; u++) { y[j][u][0] = p[((j * 8 + u) * 2) * 64]; y[j][u][1] = p[((j * 8 + u) * 2 + 1) * 64]; }
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int b = 0; b < nblocks8; b += NB) {
#pragma unroll
        for (int j = 0; j < NB; j++) {
            constexpr int dummy = 0; (void)dummy;
            const int jl = (j + NB - 1) % NB;                 // the block that was consumed last: refill it, NB - 1 blocks ahead
#pragma unroll
            for (int u = 0; u < 8; u++) {
                y[jl][u][0] = p[(((long long)(b + j + NB - 1) * 8 + u) * 2) * 64];
                y[jl][u][1] = p[(((long long)(b + j + NB - 1) * 8 + u) * 2 + 1) * 64];
            }
#pragma unroll
            for (int u = 0; u < 8; u++) A.step(y[j][u], e, nd1, nd2, cm0, cm1);
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    out[wave * 64 + lane] = A.sum();
    if (lane == 0) clk[wave] = t1 - t0;
}

template <int NB>
void run_deep(const double* buf, long long stride_rows, int rows, const char* what) {
    const int waves = 1024, nb8 = rows / 8 / NB * NB;
    double* out; unsigned long long* clk;
    hipMalloc(&out, waves * 64 * 8); hipMalloc(&clk, waves * 8);
    K c{0.6, 1.1, -0.3, 0.01, 0.02};
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float best = 1e9f;
    for (int it = 0; it < 6; it++) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(kdeep<NB>, dim3(waves / 4), dim3(256), 0, 0, c, buf, stride_rows, nb8, out, clk);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (it > 0 && ms < best) best = ms;
    }
    std::vector<unsigned long long> h(waves);
    hipMemcpy(h.data(), clk, waves * 8, hipMemcpyDeviceToHost);
    double avg = 0; for (auto v : h) avg += (double)v; avg /= waves;
    printf("deep NB=%d (%d blocks = %d KB in flight per wave), %s: %d rows per wave: %.1f us per launch = %.2f TB/s, %.0f cycles per row per wave\n", NB, NB - 1,
           8 * (NB - 1), what, nb8 * 8, 1e3 * best, 1024.0 * nb8 * 8 * 1024.0 / (best * 1e-3) / 1e12, avg / (nb8 * 8));
    hipFree(out); hipFree(clk);
}

template <int MODE>
void run(const double* buf, long long stride_rows, int rows) {
    const int waves = 1024, nb8 = rows / 8;
    double* out; unsigned long long* clk;
    hipMalloc(&out, waves * 64 * 8); hipMalloc(&clk, waves * 8);
    K c{0.6, 1.1, -0.3, 0.01, 0.02};
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float best = 1e9f;
    for (int it = 0; it < 6; it++) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(k<MODE>, dim3(waves / 4), dim3(256), 0, 0, c, buf, stride_rows, nb8, out, clk);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (it > 0 && ms < best) best = ms;
    }
    std::vector<unsigned long long> h(waves);
    hipMemcpy(h.data(), clk, waves * 8, hipMemcpyDeviceToHost);
    double avg = 0; for (auto v : h) avg += (double)v; avg /= waves;
    const double item_rows = (double)rows * (MODE == 2 ? 2 : 1);
    printf("mode %d: %d rows per wave (%s): %.1f us per launch, %.0f shader cycles (s_memtime) per wave = %.0f cycles per item-row\n", MODE, rows,
           MODE == 2 ? "x 2 items" : "1 item", 1e3 * best, avg, avg / item_rows);
    hipFree(out); hipFree(clk);
}

int main() {
    const int rows = 288;
    double* small; double* big;
    // L2-resident: every wave reads the same 288-row window of 64 tracks x 2 channels (stride 0); 200 MB: a region of its own per wave
    hipMalloc(&small, (size_t)(rows + 64) * 128 * 8); hipMemset(small, 0, (size_t)(rows + 64) * 128 * 8);
    const long long stride = 190;          // rows between the waves' regions: 1024 waves x 190 rows x 1 KB = 199 MB
    hipMalloc(&big, (size_t)(1024 * stride + rows + 64) * 128 * 8); hipMemset(big, 0, (size_t)(1024 * stride + rows + 64) * 128 * 8);
    run<0>(small, 0, rows); run<1>(small, 0, rows); run<2>(small, 0, rows); run<3>(small, 0, rows); run<3>(big, stride, rows);
    run<0>(small, 0, 4 * rows); run<2>(small, 0, 4 * rows); run<3>(small, 0, 4 * rows);
    // bytes in flight: the 200 MB buffer (re-read by every launch: does the Infinity Cache serve it, and how fast?) and a 1.6 GB one
    run_deep<2>(big, stride, rows, "200 MB"); run_deep<3>(big, stride, rows, "200 MB"); run_deep<4>(big, stride, rows, "200 MB"); run_deep<6>(big, stride, rows, "200 MB");
    double* huge;
    const long long hstride = 1530;        // 1024 x 1530 KB = 1.6 GB
    if (hipMalloc(&huge, (size_t)(1024 * hstride + 4096) * 128 * 8) == hipSuccess) {
        hipMemset(huge, 0, (size_t)(1024 * hstride + 4096) * 128 * 8);
        run_deep<2>(huge, hstride, 1488, "1.6 GB"); run_deep<3>(huge, hstride, 1488, "1.6 GB"); run_deep<4>(huge, hstride, 1488, "1.6 GB"); run_deep<6>(huge, hstride, 1488, "1.6 GB");
    }
    return 0;
}
