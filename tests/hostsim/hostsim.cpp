// tests/hostsim/hostsim.cpp -- TEST-ONLY host build of the kernel arithmetic.
//
// Compiles smoothsde_amd/csrc/ssde_math.hpp (the per-lane step functions the HIP kernels
// inline) with g++ and walks tracks sequentially the way one wavefront lane does, so the
// CPU suite (-m "not gpu") can check the hand-derived forward sensitivities against the
// oracle without a GPU.  This is NOT a product path: it is built only by the tests, lives
// outside the package, and libssde_hip.so contains no CPU evaluation code.
#include <cstdint>
#include <cstdio>
#include <cmath>

#include "../../smoothsde_amd/csrc/ssde_math.hpp"
#include "../../smoothsde_amd/csrc/ssde_tv.hpp"
#include "../../smoothsde_amd/csrc/ssde_adj.hpp"
#include <vector>

using namespace ssde;

namespace {

struct IsoArgs {
    int model, d, mask, any_nan;
    int64_t n, n_tracks;
    const int64_t* row0;   // first row of each track
    const int64_t* nrows;  // rows of each track
    const double* times;
    const double* obs;     // n x d column-major
    double log_sigma_obs, mu[2], p1, p2;  // working-scale parameters
    double p0[3];          // CTCRW: p11,p12,p22 ; OU/BM: p
};

template <int D, int MASK>
void run_ctcrw(const IsoArgs& A, double* out) {
    const double h = exp(A.log_sigma_obs) * exp(A.log_sigma_obs);
    const double tau = exp(A.p1), nu = exp(A.p2), beta = 1.0 / tau;
    const double sigma = 2.0 * nu / sqrt(M_PI * tau);
    for (int k = 0; k < 4 + D; k++) out[k] = 0.0;
    for (int64_t m = 0; m < A.n_tracks; m++) {
        CtcrwLane<D, MASK> L;
        double a0[2 * D];
        for (int a = 0; a < D; a++) { a0[2 * a] = A.obs[A.row0[m] + a * A.n]; a0[2 * a + 1] = 0.0; }
        L.init(a0, A.p0[0], A.p0[1], A.p0[2]);
        for (int64_t s = 1; s < A.nrows[m]; s++) {
            int64_t i = A.row0[m] + s;
            double dt = (s < A.nrows[m] - 1) ? A.times[i + 1] - A.times[i] : 1.0;
            CtcrwTrans tr;
            ctcrw_trans(dt, tau, beta, sigma, tr);
            double y[D];
            for (int a = 0; a < D; a++) y[a] = A.obs[i + a * A.n];
            ctcrw_step<D, MASK>(L, tr, h, A.mu, y, is_na(y[0], A.any_nan));
        }
        double o[4 + D];
        ctcrw_finish<D, MASK>(L, o);
        for (int k = 0; k < 4 + D; k++) out[k] += o[k];
    }
}

template <int D, int MASK, int MODEL>
void run_scal(const IsoArgs& A, double* out) {
    const double h = exp(A.log_sigma_obs) * exp(A.log_sigma_obs);
    for (int k = 0; k < 4 + D; k++) out[k] = 0.0;
    for (int64_t m = 0; m < A.n_tracks; m++) {
        ScalLane<D, MASK> L;
        double a0x[D];
        for (int a = 0; a < D; a++) a0x[a] = A.obs[A.row0[m] + a * A.n];
        L.init(a0x, A.p0[0]);
        for (int64_t s = 1; s < A.nrows[m]; s++) {
            int64_t i = A.row0[m] + s;
            double dt = (s < A.nrows[m] - 1) ? A.times[i + 1] - A.times[i] : 1.0;
            ScalTrans tr;
            if (MODEL == M_OU_SSM) ou_trans(dt, exp(A.p1), exp(A.p2), tr);
            else bm_trans(dt, exp(A.p1), tr);
            double y[D];
            for (int a = 0; a < D; a++) y[a] = A.obs[i + a * A.n];
            scal_step<D, MASK, MODEL == M_OU_SSM>(L, tr, h, A.mu, y, is_na(y[0], A.any_nan));
        }
        double o[4 + D];
        scal_finish<D, MASK>(L, o);
        for (int k = 0; k < 4 + D; k++) out[k] += o[k];
    }
}

template <int D, int MASK>
void run_model(const IsoArgs& A, double* out) {
    if (A.model == M_CTCRW) run_ctcrw<D, MASK>(A, out);
    else if (A.model == M_OU_SSM) run_scal<D, MASK, M_OU_SSM>(A, out);
    else run_scal<D, MASK, M_BM_SSM>(A, out);
}

template <int D>
void run_mask(const IsoArgs& A, double* out) {
    switch (A.mask) {
#define C(M) case M: run_model<D, M>(A, out); break;
        C(0) C(1) C(2) C(3) C(4) C(5) C(6) C(7) C(8) C(9) C(10) C(11) C(12) C(13) C(14) C(15)
#undef C
    }
}

// row-varying coefficients (ssde_tv.hpp): records by the pre-pass arithmetic, then one lane per direction
template <int MODEL, int D>
void run_tv(int any_nan, int64_t n, int64_t n_tracks, const int64_t* row0, const int64_t* nrows, const double* times,
            const double* obs, const double* parmat, int q, int nd, const int* kinds, const int* dims, const double* wmat,
            double log_sigma_obs, const double* p0, const double* a0, double* out) {
    typedef TvOps<MODEL, D> Ops;
    constexpr int SD = Ops::Lane::SD;
    const double h = exp(log_sigma_obs) * exp(log_sigma_obs);
    for (int k = 0; k < 1 + nd; k++) out[k] = 0.0;
    for (int kk = -1; kk < nd; kk++) {              // kk = -1: value-only lane
        for (int64_t m = 0; m < n_tracks; m++) {
            typename Ops::Lane L;
            double a[SD];
            for (int c = 0; c < SD; c++) a[c] = 0.0;
            if (a0) for (int c = 0; c < SD; c++) a[c] = a0[m * SD + c];
            else for (int c = 0; c < D; c++) a[MODEL == M_CTCRW ? 2 * c : c] = obs[row0[m] + c * n];
            L.init(a, p0);
            for (int64_t s = 1; s < nrows[m]; s++) {
                const int64_t i = row0[m] + s;
                const double dt = (s < nrows[m] - 1) ? times[i + 1] - times[i] : 1.0;
                double par[4], y[D], rec[TV_RS];
                for (int j = 0; j < q; j++) par[j] = parmat[i + (int64_t)j * n];
                for (int c = 0; c < D; c++) y[c] = obs[i + c * n];
                tv_make_record<MODEL, D>(dt, par, y, rec);
                if (kk < 0) Ops::template step<false>(L, rec, h, 0, 0, 0.0, any_nan);
                else Ops::template step<true>(L, rec, h, kinds[kk], dims[kk], wmat[i * nd + kk], any_nan);
            }
            if (kk < 0) out[0] += L.value();
            else out[1 + kk] += L.grad();
        }
    }
}

// reverse sweep (ssde_adj.hpp): forward pass with a record per row, backward pass; G[i + j n] = d nllk / d parmat(i, j)
template <int MODEL, int D>
void run_adj(int any_nan, int64_t n, int64_t n_tracks, const int64_t* row0, const int64_t* nrows, const double* times,
             const double* obs, const double* parmat, int q, double log_sigma_obs, const double* p0, const double* a0,
             double* out, double* G) {
    typedef typename AdjModel<MODEL, D>::Lane Lane;
    constexpr int SD = Lane::SD, NF = Lane::NF;
    const double h = exp(log_sigma_obs) * exp(log_sigma_obs);
    out[0] = out[1] = 0.0;
    for (int64_t k = 0; k < n * q; k++) G[k] = 0.0;
    for (int64_t m = 0; m < n_tracks; m++) {
        Lane L;
        double a[SD];
        for (int c = 0; c < SD; c++) a[c] = 0.0;
        if (a0) for (int c = 0; c < SD; c++) a[c] = a0[m * SD + c];
        else for (int c = 0; c < D; c++) a[MODEL == M_CTCRW ? 2 * c : c] = obs[row0[m] + c * n];
        L.init(a, p0);
        LogAcc ld; ld.init();
        double accq = 0.0;
        std::vector<double> recs((size_t)nrows[m] * NF);
        auto dt_of = [&](int64_t s) { const int64_t i = row0[m] + s; return (s < nrows[m] - 1) ? times[i + 1] - times[i] : 1.0; };
        for (int64_t s = 1; s < nrows[m]; s++) {
            const int64_t i = row0[m] + s;
            double mu[D], y[D];
            for (int c = 0; c < D; c++) { mu[c] = parmat[i + (int64_t)c * n]; y[c] = obs[i + c * n]; }
            typename Lane::Trans tr;
            Lane::trans(dt_of(s), parmat[i + (int64_t)D * n], q > D + 1 ? parmat[i + (int64_t)(D + 1) * n] : 0.0, tr);
            Lane::template put_trans<1>(&recs[(size_t)s * NF], tr);
            typename Lane::Trans tb;                                // (the forward step reads its transition back from the record, as the kernel does)
            Lane::template get_trans<1>(&recs[(size_t)s * NF], dt_of(s), tb);
            L.template fwd<true, 1>(tb, h, mu, y, is_na(y[0], any_nan), ld, accq, &recs[(size_t)s * NF]);
        }
        out[0] += 0.5 * ((double)D * ld.value() + accq);
        typename Lane::Adj A;
        A.zero();
        for (int64_t s = nrows[m] - 1; s >= 1; s--) {
            const int64_t i = row0[m] + s;
            double mu[D];
            for (int c = 0; c < D; c++) mu[c] = parmat[i + (int64_t)c * n];
            AdjRowGrad<D> g;
            Lane::template bwd<1>(A, &recs[(size_t)s * NF], h, mu, dt_of(s), g);
            for (int c = 0; c < D; c++) G[i + (int64_t)c * n] = g.gmu[c];
            G[i + (int64_t)D * n] = g.g1;
            if (q > D + 1) G[i + (int64_t)(D + 1) * n] = g.g2;
            out[1] += 2.0 * h * g.gh;
        }
    }
}

// ... on the full-covariance lanes (AdjFull: two response columns, per-row H_array or sigma_obs^2 I, any P0)
template <int MODEL>
void run_adj_full(int any_nan, int64_t n, int64_t n_tracks, const int64_t* row0, const int64_t* nrows, const double* times,
                  const double* obs, const double* parmat, int q, double log_sigma_obs, const double* p0f, const double* a0,
                  const double* harr /* d x d x n or NULL */, double* out, double* G) {
    typedef AdjFull<MODEL> Lane;
    constexpr int SD = Lane::SD, NF = Lane::NF, D = 2;
    const double h = exp(log_sigma_obs) * exp(log_sigma_obs);
    out[0] = out[1] = 0.0;
    for (int64_t k = 0; k < n * q; k++) G[k] = 0.0;
    for (int64_t m = 0; m < n_tracks; m++) {
        Lane L;
        double a[SD];
        for (int c = 0; c < SD; c++) a[c] = 0.0;
        if (a0) for (int c = 0; c < SD; c++) a[c] = a0[m * SD + c];
        else for (int c = 0; c < D; c++) a[Lane::z(c)] = obs[row0[m] + c * n];
        L.init(a, p0f);
        LogAcc ld; ld.init();
        double accq = 0.0;
        std::vector<double> recs((size_t)nrows[m] * NF);
        auto dt_of = [&](int64_t s) { const int64_t i = row0[m] + s; return (s < nrows[m] - 1) ? times[i + 1] - times[i] : 1.0; };
        auto H_of = [&](int64_t i, double* H) { if (harr) { H[0] = harr[i * 4]; H[1] = harr[i * 4 + 2]; H[2] = harr[i * 4 + 3]; } else { H[0] = h; H[1] = 0.0; H[2] = h; } };
        for (int64_t s = 1; s < nrows[m]; s++) {
            const int64_t i = row0[m] + s;
            double mu[D], y[D], H[3];
            for (int c = 0; c < D; c++) { mu[c] = parmat[i + (int64_t)c * n]; y[c] = obs[i + c * n]; }
            H_of(i, H);
            typename Lane::Trans tr, tb;
            Lane::trans(dt_of(s), parmat[i + (int64_t)D * n], q > D + 1 ? parmat[i + (int64_t)(D + 1) * n] : 0.0, tr);
            Lane::template put_trans<1>(&recs[(size_t)s * NF], tr);
            Lane::template get_trans<1>(&recs[(size_t)s * NF], dt_of(s), tb);
            L.template fwd<true, 1>(tb, H, mu, y, is_na(y[0], any_nan), ld, accq, &recs[(size_t)s * NF]);
        }
        out[0] += 0.5 * (ld.value() + accq);
        typename Lane::Adj A;
        A.zero();
        for (int64_t s = nrows[m] - 1; s >= 1; s--) {
            const int64_t i = row0[m] + s;
            double mu[D], y[D], H[3];
            for (int c = 0; c < D; c++) { mu[c] = parmat[i + (int64_t)c * n]; y[c] = obs[i + c * n]; }
            H_of(i, H);
            AdjRowGrad<2> g;
            Lane::template bwd<1>(A, &recs[(size_t)s * NF], H, mu, y, dt_of(s), g);
            for (int c = 0; c < D; c++) G[i + (int64_t)c * n] = g.gmu[c];
            G[i + (int64_t)D * n] = g.g1;
            if (q > D + 1) G[i + (int64_t)(D + 1) * n] = g.g2;
            if (!harr) out[1] += 2.0 * h * g.gh;
        }
    }
}

}  // namespace

extern "C" {

// out = [nllk (data term), d/d direction_0 .. d/d direction_{nd-1}]
int hostsim_kalman_tv(int model, int d, int any_nan, int64_t n, int64_t n_tracks, const int64_t* row0, const int64_t* nrows,
                      const double* times, const double* obs, const double* parmat, int q, int nd, const int* kinds,
                      const int* dims, const double* wmat, double log_sigma_obs, const double* p0, const double* a0,
                      double* out) {
#define TV(MODEL, D) if (model == MODEL && d == D) { run_tv<MODEL, D>(any_nan, n, n_tracks, row0, nrows, times, obs, parmat, q, nd, kinds, dims, wmat, log_sigma_obs, p0, a0, out); return 0; }
    TV(M_CTCRW, 1) TV(M_CTCRW, 2) TV(M_OU_SSM, 1) TV(M_OU_SSM, 2) TV(M_BM_SSM, 1) TV(M_BM_SSM, 2)
#undef TV
    return 1;
}

// out = [nllk (data term), d / d log_sigma_obs]; G (n x q, column-major) = d nllk / d parmat
int hostsim_kalman_adj(int model, int d, int any_nan, int64_t n, int64_t n_tracks, const int64_t* row0, const int64_t* nrows,
                       const double* times, const double* obs, const double* parmat, int q, double log_sigma_obs, const double* p0,
                       const double* a0, double* out, double* G) {
#define ADJ(MODEL, D) if (model == MODEL && d == D) { run_adj<MODEL, D>(any_nan, n, n_tracks, row0, nrows, times, obs, parmat, q, log_sigma_obs, p0, a0, out, G); return 0; }
    ADJ(M_CTCRW, 1) ADJ(M_CTCRW, 2) ADJ(M_OU_SSM, 1) ADJ(M_OU_SSM, 2) ADJ(M_BM_SSM, 1) ADJ(M_BM_SSM, 2)
#undef ADJ
    return 1;
}

int hostsim_kalman_adj_full(int model, int any_nan, int64_t n, int64_t n_tracks, const int64_t* row0, const int64_t* nrows,
                            const double* times, const double* obs, const double* parmat, int q, double log_sigma_obs, const double* p0f,
                            const double* a0, const double* harr, double* out, double* G) {
#define ADJF(MODEL) if (model == MODEL) { run_adj_full<MODEL>(any_nan, n, n_tracks, row0, nrows, times, obs, parmat, q, log_sigma_obs, p0f, a0, harr, out, G); return 0; }
    ADJF(M_CTCRW) ADJF(M_OU_SSM) ADJF(M_BM_SSM)
#undef ADJF
    return 1;
}

// out = [nllk, g_sig, g_mu_0.., g_p1, g_p2]  (1 + 3 + d doubles)
int hostsim_kalman_iso(int model, int d, int mask, int any_nan, int64_t n, int64_t n_tracks, const int64_t* row0,
                       const int64_t* nrows, const double* times, const double* obs, const double* theta /* ls, mu[d], p1, p2 */,
                       const double* p0, double* out) {
    IsoArgs A;
    A.model = model; A.d = d; A.mask = mask; A.any_nan = any_nan; A.n = n; A.n_tracks = n_tracks;
    A.row0 = row0; A.nrows = nrows; A.times = times; A.obs = obs;
    A.log_sigma_obs = theta[0];
    A.mu[0] = theta[1]; A.mu[1] = d > 1 ? theta[2] : 0.0;
    A.p1 = theta[1 + d]; A.p2 = theta[2 + d];
    A.p0[0] = p0[0]; A.p0[1] = p0[1]; A.p0[2] = p0[2];
    if (d == 1) run_mask<1>(A, out);
    else if (d == 2) run_mask<2>(A, out);
    else return 1;
    return 0;
}

// log I_q(x) and its derivatives w.r.t. x and q (CIR); out = [value, d/dx, d/dq]
void hostsim_log_bessel_i(double x, double q, double* out) { out[0] = log_bessel_i(x, q, out[1], out[2]); }
// ... with the second derivatives (the CIR Hessian, k_direct_hess.hip); out = [value, l_x, l_q, l_xx, l_xq, l_qq]
void hostsim_log_bessel_i2(double x, double q, double* out) {
    double d5[5];
    out[0] = log_bessel_i2(x, q, d5);
    for (int i = 0; i < 5; i++) out[1 + i] = d5[i];
}
// CIR transition: returns nll and adds gradient wrt (log mu, log beta, log sigma)
double hostsim_cir(double z0, double z1, double dt, double lmu, double lb, double ls, double* g) {
    return cir_direct(z0, z1, dt, lmu, lb, ls, g[0], g[1], g[2]);
}

// direct families, one transition: returns nll and adds gradient wrt (mu, p1, p2)
double hostsim_direct(int model, double z0, double z1, double dt, double mu, double p1, double p2, double* g) {
    if (model == M_BM) return bm_direct(z0, z1, dt, mu, p1, g[0], g[1]);
    return ou_direct(z0, z1, dt, mu, p1, p2, g[0], g[1], g[2]);
}

}  // extern "C"
