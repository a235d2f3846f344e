// ssde_host.hpp -- host-side bookkeeping shared by the engine and the test harness: parameter
// vector layout (include/ssde.h), coefficient slot table, a0/P0 defaults, path selection,
// smoothing penalty.  Plain C++17, no HIP.
#ifndef SSDE_HOST_HPP
#define SSDE_HOST_HPP

#include <cmath>
#include <cstdint>
#include <string>
#include <vector>

#include "../../include/ssde.h"

namespace ssde_host {

constexpr int kMaxCols = 96;   // == ssde::MAX_COLS
constexpr int kMaxPar = 320;   // == ssde::MAX_PAR

inline bool is_kalman(int model) {
    return model == SSDE_MODEL_BM_SSM || model == SSDE_MODEL_OU_SSM || model == SSDE_MODEL_CTCRW;
}
inline bool is_eseal(int model) { return model == SSDE_MODEL_ESEAL_SSM; }
inline int state_dim(int model, int d) {
    if (is_eseal(model)) return 2;                        // (1, lipid mass): nllk_e_seal_ssm.hpp:150-158
    return model == SSDE_MODEL_CTCRW ? 2 * d : (is_kalman(model) ? d : 0);
}
inline int n_sde_par(int model, int d) {
    return (model == SSDE_MODEL_BM || model == SSDE_MODEL_BM_SSM || model == SSDE_MODEL_BM_T || is_eseal(model)) ? d + 1 : d + 2;
}

// full parameter vector layout (include/ssde.h, PARAMETER VECTOR)
struct ParLayout {
    int q = 0, n_fe = 0, n_re = 0, n_lambda = 0, n_decay = 0;
    int off_sig = -1, off_fe = 0, off_lambda = 0, off_decay = 0, off_re = 0, n_full = 0;
    std::vector<int> fe_off, re_off, ncol_fe, ncol_re;
};

inline ParLayout make_layout(const ssde_desc* d) {
    ParLayout L;
    L.q = d->n_par;
    for (int j = 0; j < L.q; j++) {
        L.fe_off.push_back(L.n_fe);
        L.ncol_fe.push_back(d->ncol_fe[j]);
        L.n_fe += d->ncol_fe[j];
        int nr = d->ncol_re ? d->ncol_re[j] : 0;
        L.re_off.push_back(L.n_re);
        L.ncol_re.push_back(nr);
        L.n_re += nr;
    }
    L.n_lambda = d->n_smooth;
    int o = 0;
    if (is_kalman(d->model)) { L.off_sig = 0; o = 1; }   // PARAMETER(log_sigma_obs) first (nllk_ctcrw.hpp:135)
    if (is_eseal(d->model)) o = 3;                       // log_tau, a1, log_a2 (nllk_e_seal_ssm.hpp:114-116)
    L.off_fe = o; o += L.n_fe;
    L.off_lambda = o; o += L.n_lambda;
    L.n_decay = (!is_kalman(d->model) && d->n_decay > 0) ? d->n_decay : 0;     // PARAMETER_VECTOR(log_decay), nllk_sde.hpp:44
    L.off_decay = o; o += L.n_decay;
    L.off_re = o; o += L.n_re;
    L.n_full = o;
    return L;
}

// Layout of a DIMENSION PART (responses wider than two columns are evaluated as pairs of columns, ssde_engine_dist.hip):
// the part sees the SDE parameters jmap[0..) of the whole problem -- its own mu's and the shared ones -- but indexes them
// inside the WHOLE problem's vector, so that its gradient lands where the parent's does and the parts can simply be summed.
inline ParLayout child_layout(const ParLayout& P, const std::vector<int>& jmap) {
    ParLayout L = P;
    L.q = (int)jmap.size();
    L.fe_off.clear(); L.re_off.clear(); L.ncol_fe.clear(); L.ncol_re.clear();
    for (int j : jmap) {
        L.fe_off.push_back(P.fe_off[j]); L.ncol_fe.push_back(P.ncol_fe[j]);
        L.re_off.push_back(P.re_off[j]); L.ncol_re.push_back(P.ncol_re[j]);
    }
    return L;
}

// One coefficient of the linear predictor par_vec = X_fe coeff_fe + X_re coeff_re (nllk_ctcrw.hpp:143)
struct Slot {
    int par_j;            // SDE parameter
    int col;              // index into the list of streamed columns, -1 = intercept (column of ones)
    int pidx;             // index in the full parameter vector
    const double* src;    // caller's column (length n) or NULL
    int decay = -1;       // index into log_decay of a decaying random-effect column (nllk_sde.hpp:47-57), -1 = none
    int basis_c = -1;     // column inside the parameter's ssde_ppbasis when the block is given as a function, else -1
};

inline std::vector<Slot> make_slots(const ssde_desc* d, const ParLayout& L, int* n_stream_cols) {
    std::vector<Slot> s;
    int ncol = 0;
    for (int j = 0; j < L.q; j++) {
        for (int c = 0; c < L.ncol_fe[j]; c++) {
            Slot t;
            t.par_j = j;
            t.pidx = L.off_fe + L.fe_off[j] + c;
            if (d->x_fe && d->x_fe[j]) { t.src = d->x_fe[j] + (int64_t)c * d->n; t.col = ncol++; }
            else { t.src = nullptr; t.col = -1; }
            s.push_back(t);
        }
        for (int c = 0; c < L.ncol_re[j]; c++) {
            Slot t;
            t.par_j = j;
            t.pidx = L.off_re + L.re_off[j] + c;
            const bool pp = d->basis_re && d->basis_re[j];
            t.src = pp ? nullptr : d->x_re[j] + (int64_t)c * d->n;   // basis-backed blocks are materialised or evaluated by the engine
            t.basis_c = pp ? c : -1;
            t.col = ncol++;
            if (L.n_decay > 0)
                for (int k = 0; k < d->n_decay_cols; k++)
                    if (d->col_decay[k] == L.re_off[j] + c) t.decay = d->ind_decay[k];
            s.push_back(t);
        }
    }
    *n_stream_cols = ncol;
    return s;
}

inline double p0_entry(const ssde_desc* d, int i, int j) {
    const int sdim = state_dim(d->model, d->n_dim);
    if (d->p0) return d->p0[i + j * sdim];
    if (i != j) return 0.0;
    if (is_eseal(d->model)) return i == 1 ? 10.0 : 0.0;                    // diag(c(0, 10)), R/sde.R:603
    if (d->model == SSDE_MODEL_CTCRW) return (i % 2 == 0) ? 1.0 : 10.0;   // R/sde.R:584
    return 10.0;                                                           // R/sde.R:554
}

// Is P0 block-identical across dimensions and decoupled (so that the isotropic register path
// applies)?  CTCRW: blockdiag of one symmetric 2x2 block; OU/BM: p * I.
inline bool p0_is_isotropic(const ssde_desc* d, double iso[3]) {
    const int dd = d->n_dim, sdim = state_dim(d->model, dd);
    if (d->model == SSDE_MODEL_CTCRW) {
        iso[0] = p0_entry(d, 0, 0); iso[1] = p0_entry(d, 0, 1); iso[2] = p0_entry(d, 1, 1);
        if (p0_entry(d, 1, 0) != iso[1]) return false;
        for (int i = 0; i < sdim; i++)
            for (int j = 0; j < sdim; j++) {
                double want = 0.0;
                if (i / 2 == j / 2) want = (i % 2 == 0 && j % 2 == 0) ? iso[0] : (i % 2 == 1 && j % 2 == 1) ? iso[2] : iso[1];
                if (p0_entry(d, i, j) != want) return false;
            }
        return true;
    }
    iso[0] = p0_entry(d, 0, 0); iso[1] = iso[2] = 0.0;
    for (int i = 0; i < sdim; i++)
        for (int j = 0; j < sdim; j++)
            if (p0_entry(d, i, j) != (i == j ? iso[0] : 0.0)) return false;
    return true;
}

// log|det| of a small dense matrix by partial-pivot LU (atomic::matinvpd's log-determinant,
// nllk_sde.hpp:110)
inline double logabsdet(std::vector<double> A, int n) {
    double ld = 0.0;
    for (int k = 0; k < n; k++) {
        int piv = k;
        for (int i = k + 1; i < n; i++)
            if (std::fabs(A[i + k * n]) > std::fabs(A[piv + k * n])) piv = i;
        if (piv != k)
            for (int j = 0; j < n; j++) std::swap(A[k + j * n], A[piv + j * n]);
        ld += std::log(std::fabs(A[k + k * n]));
        for (int i = k + 1; i < n; i++) {
            double f = A[i + k * n] / A[k + k * n];
            for (int j = k + 1; j < n; j++) A[i + j * n] -= f * A[k + j * n];
        }
    }
    return ld;
}

// Smoothing penalty and its gradient: nllk_ctcrw.hpp:254-280 (Kalman families: no constants,
// include_penalty ignored) and nllk_sde.hpp:89-124 (direct families: + Sn/2 log(2 pi)
// + log det(S^-1)/2, gated by include_penalty).  Parameter-only, O(sum Sn^2): host arithmetic.
struct Penalty {
    int model = 0, include_penalty = 1;
    std::vector<int> ncol;
    std::vector<std::vector<double>> S;   // column-major blocks
    std::vector<double> logdet;           // log|det S_s|
    // ESEAL_SSM priors (nllk_e_seal_ssm.hpp:212-216): inverse gamma on sigma(0)^2 -- the FIRST row's sigma -- and
    // on tau^2; (full-par index, design weight at row 0) of every coefficient of log sigma
    int64_t eseal_n = 0;
    std::vector<std::pair<int, double>> eseal_sig0;

    void setup(const ssde_desc* d) {
        model = d->model;
        include_penalty = d->include_penalty;
        const double* p = d->s_blocks;
        for (int s = 0; s < d->n_smooth; s++) {
            int n = d->smooth_ncol[s];
            ncol.push_back(n);
            S.emplace_back(p, p + (size_t)n * n);
            logdet.push_back((is_kalman(model) || is_eseal(model)) ? 0.0 : logabsdet(S.back(), n));
            p += (size_t)n * n;
        }
    }
    // -(log priors) and their gradient; dinvgamma(x, shape, scale) = shape log(scale) - lgamma(shape) - (shape+1) log x - scale/x
    double eseal_priors(const double* par, double* grad) const {
        const double n = (double)eseal_n, nh = (double)(eseal_n / 2);   // integer division, as in the reference
        double ls0 = 0.0;
        for (auto& e : eseal_sig0) ls0 += e.second * par[e.first];
        const double x1 = std::exp(2.0 * ls0), sh1 = 10.0 * n, sc1 = 4.0 * (10.0 * n - 1.0);
        const double x2 = std::exp(2.0 * par[0]), sh2 = nh, sc2 = nh - 1.0;
        const double lp = (sh1 * std::log(sc1) - std::lgamma(sh1) - (sh1 + 1.0) * std::log(x1) - sc1 / x1) +
                          (sh2 * std::log(sc2) - std::lgamma(sh2) - (sh2 + 1.0) * std::log(x2) - sc2 / x2);
        if (grad) {
            const double d1 = 2.0 * (sh1 + 1.0) - 2.0 * sc1 / x1;       // d(-lp)/d log sigma0
            for (auto& e : eseal_sig0) grad[e.first] += d1 * e.second;
            grad[0] += 2.0 * (sh2 + 1.0) - 2.0 * sc2 / x2;              // d(-lp)/d log tau
        }
        return -lp;
    }
    double eval(const ParLayout& L, const double* par, double* grad /* may be NULL; added into */) const {
        if (is_eseal(model)) return eval_smooth(L, par, grad) + eseal_priors(par, grad);
        return eval_smooth(L, par, grad);
    }
    double eval_smooth(const ParLayout& L, const double* par, double* grad) const {
        if (ncol.empty()) return 0.0;                                   // ncol_re(0) > 0
        if (!is_kalman(model) && !is_eseal(model) && !include_penalty) return 0.0;   // nllk_sde.hpp:91
        double pen = 0.0;
        int start = 0;
        for (size_t s = 0; s < ncol.size(); s++) {
            const int n = ncol[s];
            const double* b = par + L.off_re + start;
            const double ll = par[L.off_lambda + s];
            const double lam = std::exp(ll);
            double quad = 0.0;
            for (int a = 0; a < n; a++) {
                double Sx = 0.0, Stx = 0.0;
                for (int c = 0; c < n; c++) { Sx += S[s][a + c * n] * b[c]; Stx += S[s][c + a * n] * b[c]; }
                quad += b[a] * Sx;
                if (grad) grad[L.off_re + start + a] += 0.5 * lam * (Sx + Stx);
            }
            pen += -0.5 * n * ll + 0.5 * lam * quad;
            if (!is_kalman(model) && !is_eseal(model)) pen += 0.5 * n * std::log(2.0 * M_PI) - 0.5 * logdet[s];
            if (grad) grad[L.off_lambda + s] += -0.5 * n + 0.5 * lam * quad;
            start += n;
        }
        return pen;
    }
};

}  // namespace ssde_host
#endif
