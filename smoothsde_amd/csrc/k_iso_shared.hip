// k_iso_shared.hip -- host side of the shared-covariance kernels (k_iso_shared.inc): the stationary constants and the
// dispatch to the per-model translation units.
#include <hip/hip_ext.h>

#include "ssde_device.hpp"

namespace ssde {

hipError_t launch_iso_shared_ctcrw(int d, const IsoArgs& a, const ReduceArgs& r, dim3 grid, hipStream_t s, hipEvent_t ev0, hipEvent_t ev1, bool deep);
hipError_t launch_iso_shared_ou(int d, const IsoArgs& a, const ReduceArgs& r, dim3 grid, hipStream_t s, hipEvent_t ev0, hipEvent_t ev1, bool deep);
hipError_t launch_iso_shared_bm(int d, const IsoArgs& a, const ReduceArgs& r, dim3 grid, hipStream_t s, hipEvent_t ev0, hipEvent_t ev1, bool deep);

// host side: the stationary constants (layout in ssde_device.hpp)
void fill_stat_consts(int model, int d, IsoArgs& a) {
    double* c = a.statc;
    for (int i = 0; i < 48; i++) c[i] = 0.0;
    const double* r = a.gain_stat;
    if (model == M_CTCRW) {
        const CtcrwTrans& tr = a.ctr;
        const double bm = r[3];
        c[0] = r[0]; c[1] = r[1]; c[2] = r[2]; c[3] = 1.0 - r[1]; c[4] = tr.t12; c[5] = tr.e; c[6] = tr.dt12; c[7] = tr.de;
        c[8] = bm * tr.b1; c[9] = bm * tr.b2;
        for (int j = 0; j < NDIRP; j++) { c[10 + j] = 0.5 * r[4 + j]; c[13 + j] = r[7 + j]; c[16 + j] = r[10 + j]; }
        for (int k = 0; k < d; k++) { c[19 + k] = c[8] * a.mu[k]; c[21 + k] = c[9] * a.mu[k]; c[23 + k] = bm * a.mu[k]; }
        // transfer-function form (TfCtcrw)
        const double k1 = r[1], k2 = r[2], c1 = 1.0 - k1, e = tr.e, t12 = tr.t12;
        const double d1 = -(c1 + e), d2 = c1 * e + k2 * t12;
        c[26] = -d1; c[27] = -d2; c[28] = d2;
        const double dt = tr.b1 + tr.t12;                     // b1 = dt - t12
        for (int k = 0; k < d; k++) c[29 + k] = bm * a.mu[k] * dt;
        for (int j = 0; j < NDIRP; j++) {
            const double de = (j == 1) ? tr.de : 0.0, dt12 = (j == 1) ? tr.dt12 : 0.0;
            const double dk1 = r[7 + j], dk2 = r[10 + j];
            const double alpha = dk1 - de;                                        // d d1 / d theta_j
            const double beta = -dk1 * e + c1 * de + dk2 * t12 + k2 * dt12;       // d d2 / d theta_j
            c[31 + j] = -dk1;
            c[34 + j] = -de * d1 - beta + e * alpha;
            c[37 + j] = -de * d2 + e * beta;
            c[40 + j] = alpha; c[43 + j] = beta;
        }
        const double D1 = 1.0 + d1 + d2;                      // D(1): steady-state response to a constant input
        c[46] = dt * (1.0 - e) / D1;                          // d x / d mu_a
        c[47] = 1.0 - k2 * dt / D1;                           // d v / d mu_a
    } else {
        const ScalTrans& tr = a.str;
        c[0] = r[0]; c[1] = r[1]; c[2] = r[2]; c[3] = tr.t; c[4] = tr.b; c[5] = tr.dt_;
        for (int j = 0; j < NDIRP; j++) { c[10 + j] = 0.5 * r[4 + j]; c[13 + j] = r[7 + j]; }
        for (int k = 0; k < d; k++) { c[19 + k] = tr.b * a.mu[k]; c[21 + k] = tr.db * a.mu[k]; }
    }
}

// the shared path runs all directions in one part (n_parts == 1, mask = part_mask[0]).  r: the reduction's arguments, read by the kernel
// only when a.fused (the finalising work inside this launch); a.fuse_items is set here
hipError_t launch_iso_shared(int model, int d, const IsoArgs& a0, const ReduceArgs& r, hipStream_t s, hipEvent_t ev0, hipEvent_t ev1) {
    if (a0.n_parts != 1) return hipErrorInvalidValue;
    IsoArgs a = a0;
    const int g8 = (a.tv.n_groups + 7) / 8;
    const int n_grid_chunks = a.t0 > 0 ? a.n_chunks - 1 : a.n_chunks;
    dim3 grid((g8 * 8 * n_grid_chunks + WG_WAVES - 1) / WG_WAVES);
    if (grid.x == 0) return hipSuccess;
    a.fuse_items = a.tv.n_groups * n_grid_chunks;              // (fused launches run every group: the engine sees to it)
    const bool deep = a.deep_prefetch != 0;
    if (model == M_CTCRW) return launch_iso_shared_ctcrw(d, a, r, grid, s, ev0, ev1, deep);
    if (model == M_OU_SSM) return launch_iso_shared_ou(d, a, r, grid, s, ev0, ev1, deep);
    if (model == M_BM_SSM) return launch_iso_shared_bm(d, a, r, grid, s, ev0, ev1, deep);
    return hipErrorInvalidValue;
}

}  // namespace ssde
