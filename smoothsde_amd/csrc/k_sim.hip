// k_sim.hip -- ssde_simulate: synthetic track batches generated in HBM (gfx950).
//
// The exact-transition simulator of the reference (R/sde.R:1434-1478: BM increments :1434-1438, OU transition
// :1439-1448, CTCRW joint (velocity, position) transition :1449-1478 with CTCRW_cov of R/utility.R:188-196),
// vectorised over tracks -- lane = track, the serial recursion in registers -- plus the N(0, sigma_obs^2) observation
// error the state-space families are fitted with.  The reference draws from R's generator;
// here every normal deviate is a pure function of (seed, GLOBAL track index, row, dimension): Philox4x32-10 as a
// counter-based generator + Box-Muller.  A batch is therefore the same set of tracks however it is cut over ranks
// (bench.py --scaling strong: rank r generates tracks [r M / N, (r + 1) M / N) of the one batch) and is reproduced
// bit for bit by the numpy restatement the tests hold (tests/sim_ref.py), up to the last bits of log / sincos.
//
// Output is the reference's long format (all tracks concatenated, obs n x d column-major).  A wave keeps a
// 64-track x 32-row tile in LDS and writes it out along the tracks (256 contiguous bytes per track and store).
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstring>
#include <string>

#include "../../include/ssde.h"

namespace ssde_engine { extern thread_local std::string g_create_error; }   // what ssde_last_error(NULL) returns
#define g_sim_error ssde_engine::g_create_error

namespace {

struct SimArgs {
    int kind;                  // 0 CTCRW, 1 OU, 2 BM
    int d, n_steps;
    int64_t track0, n_tracks, n_rows, row_offset;
    const int64_t* row0;       // NULL = every track has n_steps rows
    double mu[8], z0[8];
    double dt, sigma_obs;
    double e, ib1e, l11, l21, l22;   // CTCRW: exp(-beta dt), (1 - e) / beta, Cholesky factor of CTCRW_cov in (v, z) order
    double sd;                       // OU / BM: standard deviation of one transition
    uint32_t k0, k1;
    double* id;
    double* times;
    double* obs;
};

__device__ __forceinline__ void philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1, uint32_t (&o)[4]) {
#pragma unroll
    for (int r = 0; r < 10; r++) {
        const uint64_t p0 = (uint64_t)0xD2511F53u * c0, p1 = (uint64_t)0xCD9E8D57u * c2;
        const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0, n1 = (uint32_t)p1, n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1, n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    o[0] = c0; o[1] = c1; o[2] = c2; o[3] = c3;
}

// two independent N(0, 1) deviates of (track, row, stream)
__device__ __forceinline__ void normal_pair(const SimArgs& A, uint64_t track, uint32_t row, uint32_t stream, double& n1, double& n2) {
    uint32_t o[4];
    philox4x32_10(row, (uint32_t)track, (uint32_t)(track >> 32), stream, A.k0, A.k1, o);
    const double u1 = ((double)((((uint64_t)o[0] << 32) | o[1]) >> 11) + 0.5) * 0x1.0p-53;     // (0, 1)
    const double u2 = ((double)((((uint64_t)o[2] << 32) | o[3]) >> 11) + 0.5) * 0x1.0p-53;
    const double r = sqrt(-2.0 * log(u1));
    double s, c;
    sincospi(2.0 * u2, &s, &c);
    n1 = r * c; n2 = r * s;
}

constexpr int SIM_TT = 32;     // rows per LDS tile

template <int KIND>
__global__ __launch_bounds__(64) void sim_kernel(const SimArgs A) {
    __shared__ double tile[64][SIM_TT + 1];
    const int lane = threadIdx.x, dim = blockIdx.y;
    const int64_t ml = (int64_t)blockIdx.x * 64 + lane;
    const bool valid = ml < A.n_tracks;
    const uint64_t track = (uint64_t)(A.track0 + (valid ? ml : 0));
    const int64_t r0 = !valid ? 0 : (A.row0 ? A.row0[ml] : ml * (int64_t)A.n_steps);
    const int T = !valid ? 0 : (A.row0 ? (int)(A.row0[ml + 1] - r0) : A.n_steps);
    const double mu = A.mu[dim];
    double z = A.z0[dim], v = 0.0;
    for (int t0 = 0; t0 < A.n_steps; t0 += SIM_TT) {
        for (int s = 0; s < SIM_TT; s++) {
            const int t = t0 + s;
            if (t >= T) break;
            double na = 0.0, nb = 0.0, no = 0.0, dummy;
            if (KIND == 0) {
                if (t > 0) normal_pair(A, track, (uint32_t)t, (uint32_t)(dim * 2), na, nb);
                if (A.sigma_obs > 0.0) normal_pair(A, track, (uint32_t)t, (uint32_t)(dim * 2 + 1), no, dummy);
                if (t > 0) {                                           // R/sde.R:1465-1475
                    z = z + mu * A.dt + (v - mu) * A.ib1e + A.l21 * na + A.l22 * nb;
                    v = A.e * v + (1.0 - A.e) * mu + A.l11 * na;
                }
            } else {
                normal_pair(A, track, (uint32_t)t, (uint32_t)(dim * 2), na, no);
                if (t > 0) {
                    if (KIND == 1) z = A.e * z + (1.0 - A.e) * mu + A.sd * na;   // R/sde.R:1441-1447
                    else z = z + mu * A.dt + A.sd * na;                          // R/sde.R:1435-1438
                }
            }
            tile[lane][s] = z + A.sigma_obs * no;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        // the tile goes out along the tracks: lanes 0-31 / 32-63 write 32 consecutive rows of two tracks per store
        for (int j = 0; j < 32; j++) {
            const int trk = 2 * j + (lane >> 5), s = lane & 31, t = t0 + s;
            const int64_t m2 = (int64_t)blockIdx.x * 64 + trk;
            if (m2 >= A.n_tracks) continue;
            const int64_t rr0 = A.row0 ? A.row0[m2] : m2 * (int64_t)A.n_steps;
            const int T2 = A.row0 ? (int)(A.row0[m2 + 1] - rr0) : A.n_steps;
            if (t >= T2) continue;
            const int64_t i = rr0 + t;
            A.obs[(int64_t)dim * A.n_rows + i] = tile[trk][s];
            if (dim == 0) {
                if (A.id) A.id[i] = (double)(A.track0 + m2);
                if (A.times) A.times[i] = (double)(A.row_offset + i + 1) * A.dt;     // increasing globally (inst/example.R:17)
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    }
}

}  // namespace

extern "C" {

int ssde_simulate(const ssde_sim_desc* d, double* id_dev, double* times_dev, double* obs_dev, void* stream) {
    if (!d || !obs_dev) { g_sim_error = "NULL argument"; return SSDE_ERR_ARG; }
    if (d->abi_version != SSDE_ABI_VERSION) { g_sim_error = "descriptor built for another ABI version"; return SSDE_ERR_ARG; }
    if (d->n_dim < 1 || d->n_dim > 8 || d->n_tracks < 0 || d->n_steps < 1 || d->n_rows < 0 || !(d->dt > 0.0)) {
        g_sim_error = "ssde_simulate: n_dim in 1..8, n_steps >= 1 and dt > 0 are required";
        return SSDE_ERR_ARG;
    }
    if (!d->row0 && d->n_rows != d->n_tracks * (int64_t)d->n_steps) { g_sim_error = "ssde_simulate: n_rows != n_tracks * n_steps"; return SSDE_ERR_ARG; }
    SimArgs a;
    memset(&a, 0, sizeof(a));
    switch (d->model) {
    case SSDE_MODEL_CTCRW: a.kind = 0; break;
    case SSDE_MODEL_OU: case SSDE_MODEL_OU_SSM: a.kind = 1; break;
    case SSDE_MODEL_BM: case SSDE_MODEL_BM_SSM: a.kind = 2; break;
    default: g_sim_error = "Simulation not implemented yet for this model (R/sde.R:1496)"; return SSDE_ERR_MODEL;
    }
    // the parameters the Cholesky factors below are formed from: a NaN / Inf factor would be written into HBM silently
    {
        const bool err_obs = d->model == SSDE_MODEL_CTCRW || d->model == SSDE_MODEL_OU_SSM || d->model == SSDE_MODEL_BM_SSM;
        bool ok = !err_obs || (d->sigma_obs >= 0.0 && std::isfinite(d->sigma_obs));
        if (a.kind == 0) ok = ok && d->tau > 0.0 && std::isfinite(d->tau) && d->nu > 0.0 && std::isfinite(d->nu);
        else if (a.kind == 1) ok = ok && d->tau > 0.0 && std::isfinite(d->tau) && d->kappa >= 0.0 && std::isfinite(d->kappa);
        else ok = ok && d->sigma >= 0.0 && std::isfinite(d->sigma);
        if (!ok) {
            g_sim_error = "ssde_simulate: tau > 0, nu > 0 (CTCRW), kappa >= 0 (OU), sigma >= 0 (BM) and sigma_obs >= 0, all finite, are required";
            return SSDE_ERR_ARG;
        }
    }
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count == 0) { g_sim_error = "no HIP device visible: there is no CPU fallback"; return SSDE_ERR_NODEVICE; }
    if (d->device >= 0 && hipSetDevice(d->device) != hipSuccess) { g_sim_error = "hipSetDevice failed"; return SSDE_ERR_HIP; }
    a.d = d->n_dim; a.n_steps = d->n_steps; a.track0 = d->track0; a.n_tracks = d->n_tracks; a.n_rows = d->n_rows;
    a.row_offset = d->row_offset; a.row0 = d->row0;
    for (int k = 0; k < 8; k++) { a.mu[k] = d->mu[k]; a.z0[k] = d->z0[k]; }
    a.dt = d->dt;
    const bool with_error = d->model == SSDE_MODEL_CTCRW || d->model == SSDE_MODEL_OU_SSM || d->model == SSDE_MODEL_BM_SSM;
    a.sigma_obs = with_error ? d->sigma_obs : 0.0;
    if (a.kind == 0) {
        const double beta = 1.0 / d->tau, sigma = 2.0 * d->nu / sqrt(d->tau * M_PI);       // R/sde.R:1457-1458
        const double e = exp(-beta * d->dt), e2 = exp(-2.0 * beta * d->dt);
        const double qvv = sigma * sigma / (2.0 * beta) * (1.0 - e2);                        // R/utility.R:190-194
        const double qzz = (sigma / beta) * (sigma / beta) * (d->dt + (1.0 - e2) / (2.0 * beta) - 2.0 * (1.0 - e) / beta);
        const double qvz = sigma * sigma / (2.0 * beta * beta) * (1.0 - 2.0 * e + e2);
        a.e = e; a.ib1e = (1.0 - e) / beta;
        a.l11 = sqrt(qvv); a.l21 = qvz / a.l11; a.l22 = sqrt(fmax(qzz - a.l21 * a.l21, 0.0));
    } else if (a.kind == 1) {
        a.e = exp(-d->dt / d->tau);
        a.sd = sqrt(d->kappa * (1.0 - exp(-2.0 * d->dt / d->tau)));
    } else {
        a.sd = d->sigma * sqrt(d->dt);
    }
    a.k0 = (uint32_t)d->seed; a.k1 = (uint32_t)(d->seed >> 32);
    a.id = id_dev; a.times = times_dev; a.obs = obs_dev;
    if (d->n_tracks == 0) return SSDE_OK;
    const dim3 grid((unsigned)((d->n_tracks + 63) / 64), (unsigned)d->n_dim), block(64);
    hipStream_t s = (hipStream_t)stream;
    if (a.kind == 0) hipLaunchKernelGGL(sim_kernel<0>, grid, block, 0, s, a);
    else if (a.kind == 1) hipLaunchKernelGGL(sim_kernel<1>, grid, block, 0, s, a);
    else hipLaunchKernelGGL(sim_kernel<2>, grid, block, 0, s, a);
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) { g_sim_error = std::string("ssde_simulate launch: ") + hipGetErrorString(e); return SSDE_ERR_HIP; }
    return SSDE_OK;
}

}  // extern "C"
