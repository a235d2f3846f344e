// oracle/cpu_fast.cpp -- CPU baseline of the headline path: constant-coefficient isotropic Kalman families
// (CTCRW / OU_SSM / BM_SSM, H = sigma_obs^2 I, default block-identical P0), value + analytic gradient, threads over tracks.
//
// TEST / MEASUREMENT INFRASTRUCTURE ONLY, and NOT a checker: this is what SURVEY.md 8(d)(i) asks to be timed beside
// the GPU -- "the build's CPU restatement (-O2, fp64, analytic gradient), over tracks on all host cores" -- i.e. the
// best CPU implementation this build has, not the literal dual-number restatement of ssde_oracle.hpp (which stays the
// checker, and which this file is itself checked against: tests/test_oracle_golden.py::test_cpu_fast_matches_the_oracle).
// The per-row arithmetic is the engine's own hand-derived step (smoothsde_amd/csrc/ssde_math.hpp: the recursion of
// /root/reference/src/nllk/nllk_ctcrw.hpp:195-247, nllk_ou_ssm.hpp:163-213, nllk_bm_ssm.hpp:127-175 with forward
// sensitivities), compiled for the host with hardware FMA; on a regular grid the transition is formed once, as a
// careful CPU programmer would.  Nothing under smoothsde_amd/ includes, links or calls this file; bench.py's
// cpu_baseline leg is its only user outside tests/.
#include <cmath>
#include <cstdint>
#include <thread>
#include <vector>

#include "../smoothsde_amd/csrc/ssde_math.hpp"

using namespace ssde;

namespace {

struct Job {
    int model, d, mask, any_nan;
    int64_t n;
    const int64_t *row0, *nrows;
    const double *times, *obs;
    double lso, mu[2], p1, p2, p0[3];
    int uniform_dt;
    double dt_uniform;
};

template <int D, int MASK>
void tracks_ctcrw(const Job& A, int64_t m0, int64_t m1, double* out) {
    const double sig = std::exp(A.lso), h = sig * sig;
    const double tau = std::exp(A.p1), nu = std::exp(A.p2), beta = 1.0 / tau, sigma = 2.0 * nu / std::sqrt(M_PI * tau);
    CtcrwTrans tru;
    if (A.uniform_dt) ctcrw_trans(A.dt_uniform, tau, beta, sigma, tru);
    for (int64_t m = m0; m < m1; m++) {
        CtcrwLane<D, MASK> L;
        double a0[2 * D];
        for (int a = 0; a < D; a++) { a0[2 * a] = A.obs[A.row0[m] + a * A.n]; a0[2 * a + 1] = 0.0; }
        L.init(a0, A.p0[0], A.p0[1], A.p0[2]);
        for (int64_t s = 1; s < A.nrows[m]; s++) {
            const int64_t i = A.row0[m] + s;
            double y[D];
            for (int a = 0; a < D; a++) y[a] = A.obs[i + a * A.n];
            if (A.uniform_dt) {
                ctcrw_step<D, MASK>(L, tru, h, A.mu, y, is_na(y[0], A.any_nan));
            } else {
                CtcrwTrans tr;
                ctcrw_trans((s < A.nrows[m] - 1) ? A.times[i + 1] - A.times[i] : 1.0, tau, beta, sigma, tr);
                ctcrw_step<D, MASK>(L, tr, h, A.mu, y, is_na(y[0], A.any_nan));
            }
        }
        double o[4 + D];
        ctcrw_finish<D, MASK>(L, o);
        for (int k = 0; k < 4 + D; k++) out[k] += o[k];
    }
}

template <int D, int MASK, int MODEL>
void tracks_scal(const Job& A, int64_t m0, int64_t m1, double* out) {
    const double sig = std::exp(A.lso), h = sig * sig;
    ScalTrans tru;
    if (A.uniform_dt) { if (MODEL == M_OU_SSM) ou_trans(A.dt_uniform, std::exp(A.p1), std::exp(A.p2), tru); else bm_trans(A.dt_uniform, std::exp(A.p1), tru); }
    for (int64_t m = m0; m < m1; m++) {
        ScalLane<D, MASK> L;
        double a0[D];
        for (int a = 0; a < D; a++) a0[a] = A.obs[A.row0[m] + a * A.n];
        L.init(a0, A.p0[0]);
        for (int64_t s = 1; s < A.nrows[m]; s++) {
            const int64_t i = A.row0[m] + s;
            double y[D];
            for (int a = 0; a < D; a++) y[a] = A.obs[i + a * A.n];
            ScalTrans tr = tru;
            if (!A.uniform_dt) {
                const double dt = (s < A.nrows[m] - 1) ? A.times[i + 1] - A.times[i] : 1.0;
                if (MODEL == M_OU_SSM) ou_trans(dt, std::exp(A.p1), std::exp(A.p2), tr); else bm_trans(dt, std::exp(A.p1), tr);
            }
            scal_step<D, MASK, MODEL == M_OU_SSM>(L, tr, h, A.mu, y, is_na(y[0], A.any_nan));
        }
        double o[4 + D];
        scal_finish<D, MASK>(L, o);
        for (int k = 0; k < 4 + D; k++) out[k] += o[k];
    }
}

template <int D, int MASK>
void tracks(const Job& A, int64_t m0, int64_t m1, double* out) {
    if (A.model == M_CTCRW) tracks_ctcrw<D, MASK>(A, m0, m1, out);
    else if (A.model == M_OU_SSM) tracks_scal<D, MASK, M_OU_SSM>(A, m0, m1, out);
    else tracks_scal<D, MASK, M_BM_SSM>(A, m0, m1, out);
}

template <int D>
void tracks_mask(const Job& A, int64_t m0, int64_t m1, double* out) {
    switch (A.mask) {
#define C(M) case M: tracks<D, M>(A, m0, m1, out); break;
        C(0) C(1) C(2) C(3) C(4) C(5) C(6) C(7) C(8) C(9) C(10) C(11) C(12) C(13) C(14) C(15)
#undef C
    }
}

}  // namespace

// out[0] = nllk (data term), out[1..] = d/d(log_sigma_obs, mu_1..mu_d, par[d], par[d+1]) for the directions in `mask`
// (DIR_SIG = 1, DIR_MU = 2, DIR_P1 = 4, DIR_P2 = 8).  theta = (log_sigma_obs, mu_1..mu_d, par[d], par[d+1]).
extern "C" int cpu_fast_kalman(int model, int d, int mask, int any_nan, int64_t n, int64_t n_tracks, const int64_t* row0,
                               const int64_t* nrows, const double* times, const double* obs, const double* theta,
                               const double* p0, int uniform_dt, double dt_uniform, int n_threads, double* out) {
    if (!__builtin_cpu_supports("fma")) return 2;                 // built with -mfma: refuse rather than fault
    if (d < 1 || d > 2 || (model != M_CTCRW && model != M_OU_SSM && model != M_BM_SSM)) return 1;
    Job A;
    A.model = model; A.d = d; A.mask = mask; A.any_nan = any_nan; A.n = n; A.row0 = row0; A.nrows = nrows;
    A.times = times; A.obs = obs; A.lso = theta[0];
    A.mu[0] = theta[1]; A.mu[1] = d > 1 ? theta[2] : 0.0;
    A.p1 = theta[1 + d]; A.p2 = theta[2 + d];
    for (int k = 0; k < 3; k++) A.p0[k] = p0[k];
    A.uniform_dt = uniform_dt; A.dt_uniform = dt_uniform;
    const int T = (int)std::max<int64_t>(1, std::min<int64_t>(n_threads, n_tracks));
    std::vector<std::vector<double>> part(T, std::vector<double>(4 + d, 0.0));
    std::vector<std::thread> th;
    for (int t = 0; t < T; t++)
        th.emplace_back([&, t]() {
            const int64_t m0 = n_tracks * t / T, m1 = n_tracks * (t + 1) / T;
            if (d == 1) tracks_mask<1>(A, m0, m1, part[t].data()); else tracks_mask<2>(A, m0, m1, part[t].data());
        });
    for (auto& x : th) x.join();
    for (int k = 0; k < 4 + d; k++) { out[k] = 0.0; for (int t = 0; t < T; t++) out[k] += part[t][k]; }
    return 0;
}
