// oracle/ssde_oracle.hpp -- CPU restatement of smoothSDE's nllk templates.
//
// TEST INFRASTRUCTURE ONLY.  Nothing under smoothsde_amd/ may include, link or call this
// file; it is the checker for the HIP path (tests/, __graft_entry__.smoke(), and the
// cpu_baseline leg of bench.py).  PARITY UNPINNED: the reference's own tests hold no
// numeric result for this path (/root/reference/tests/testthat/test_sde.R:4-72) and the
// reference cannot be built here (it needs TMB.hpp / Eigen / R headers, none installed),
// so this restatement is pinned only by independent cross-checks (oracle/README.md).
//
// Every function names the reference lines it follows.  Like the reference, the code is
// templated on the scalar `Type`; the reference instantiates it with CppAD's AD<double>,
// the oracle with double (value) and ssde_oracle::Dual<N> (gradient).
#ifndef SSDE_ORACLE_HPP
#define SSDE_ORACLE_HPP

#include <cmath>
#include <cstdint>
#include <cstring>
#include <utility>
#include <vector>

#include "../include/ssde.h"
#include "dual.hpp"

namespace ssde_oracle {

// ---------------------------------------------------------------------------------------
// R_IsNA(asDouble(x)) -- nllk_ctcrw.hpp:214, nllk_ou_ssm.hpp:179, nllk_bm_ssm.hpp:141,
// tr_dens.hpp:31.  R's NA_real_ is a NaN whose low 32-bit word is 1954.
// ---------------------------------------------------------------------------------------
inline bool is_na(double x, int na_mode) {
    if (!(x != x)) return false;
    if (na_mode == SSDE_NA_ANY_NAN) return true;
    uint64_t bits;
    std::memcpy(&bits, &x, 8);
    return (uint32_t)(bits & 0xffffffffu) == 1954u;
}

// ---------------------------------------------------------------------------------------
// Small dense matrices (the reference uses heap-allocated Eigen dynamic matrices,
// tmbutils matrix<Type>; state dimension is at most 2*n_dim).
// ---------------------------------------------------------------------------------------
// ARBITER MODE (off by default; ssde_oracle_keep_P_symmetric(1) in either library).  The reference propagates P as a full matrix
// (nllk_ctcrw.hpp:241, Q8); with a measurement covariance that couples the response columns that recursion AMPLIFIES the antisymmetric
// part rounding leaves in P (~1.2 per row), and the literal evaluation in double loses the likelihood after ~150 rows -- the literal
// evaluation in binary128 after ~400.  With this switch the update ends with P <- (P + P') / 2, which is the identity in exact
// arithmetic: NOT the reference's arithmetic, the value the reference's MODEL defines; used only to arbitrate between the engine and
// the literal restatement where the latter is roundoff (tests/test_oracle_golden.py::test_reference_form_loses_...).
inline bool& keep_P_symmetric() { static bool on = false; return on; }

constexpr int MAXD = 16;         // widest state: CTCRW with eight response columns run as one filter
constexpr int MAT_INLINE = 64;   // entries kept in the object (sdim <= 8: every configuration but the widest coupled responses)

template <class Type>
struct Mat {
    int r, c;
    Type small_[MAT_INLINE];
    std::vector<Type> big_;       // used beyond MAT_INLINE entries only
    Mat() : r(0), c(0) {}
    Mat(int r_, int c_) : r(r_), c(c_) { if (r * c > MAT_INLINE) big_.resize((size_t)r * c); setZero(); }
    Type* data() { return big_.empty() ? small_ : big_.data(); }
    const Type* data() const { return big_.empty() ? small_ : big_.data(); }
    void setZero() { Type* a = data(); for (int i = 0; i < r * c; i++) a[i] = Type(0.0); }
    Type& operator()(int i, int j) { return data()[i + j * r]; }
    const Type& operator()(int i, int j) const { return data()[i + j * r]; }
};

template <class Type>
Mat<Type> mul(const Mat<Type>& A, const Mat<Type>& B) {
    Mat<Type> C(A.r, B.c);
    for (int j = 0; j < B.c; j++)
        for (int k = 0; k < A.c; k++)
            for (int i = 0; i < A.r; i++) C(i, j) = C(i, j) + A(i, k) * B(k, j);
    return C;
}
template <class Type>
Mat<Type> transpose(const Mat<Type>& A) {
    Mat<Type> C(A.c, A.r);
    for (int i = 0; i < A.r; i++)
        for (int j = 0; j < A.c; j++) C(j, i) = A(i, j);
    return C;
}
template <class Type>
Mat<Type> add(const Mat<Type>& A, const Mat<Type>& B) {
    Mat<Type> C(A.r, A.c);
    for (int i = 0; i < A.r * A.c; i++) C.data()[i] = A.data()[i] + B.data()[i];
    return C;
}
template <class Type>
Mat<Type> sub(const Mat<Type>& A, const Mat<Type>& B) {
    Mat<Type> C(A.r, A.c);
    for (int i = 0; i < A.r * A.c; i++) C.data()[i] = A.data()[i] - B.data()[i];
    return C;
}

// Partial-pivoting LU, the algorithm behind Eigen's dynamic-size `inverse()`
// (nllk_ctcrw.hpp:231,236) and behind TMB's atomic::logdet (X.lu().matrixLU(),
// log|diag| summed; used at nllk_ctcrw.hpp:21, nllk_ou_ssm.hpp:190, nllk_bm_ssm.hpp:152).
template <class Type>
struct LU {
    Mat<Type> lu;
    int perm[MAXD];
    explicit LU(const Mat<Type>& A) : lu(A) {
        int n = A.r;
        for (int i = 0; i < n; i++) perm[i] = i;
        for (int k = 0; k < n; k++) {
            int piv = k;
            double best = std::fabs(asDouble(lu(k, k)));
            for (int i = k + 1; i < n; i++) {
                double v = std::fabs(asDouble(lu(i, k)));
                if (v > best) { best = v; piv = i; }
            }
            if (piv != k) {
                for (int j = 0; j < n; j++) { Type t = lu(k, j); lu(k, j) = lu(piv, j); lu(piv, j) = t; }
                int t = perm[k]; perm[k] = perm[piv]; perm[piv] = t;
            }
            for (int i = k + 1; i < n; i++) {
                lu(i, k) = lu(i, k) / lu(k, k);
                for (int j = k + 1; j < n; j++) lu(i, j) = lu(i, j) - lu(i, k) * lu(k, j);
            }
        }
    }
    Mat<Type> inverse() const {
        int n = lu.r;
        Mat<Type> X(n, n);
        for (int col = 0; col < n; col++) {
            Type y[MAXD];
            for (int i = 0; i < n; i++) {
                Type s = (perm[i] == col) ? Type(1.0) : Type(0.0);
                for (int j = 0; j < i; j++) s = s - lu(i, j) * y[j];
                y[i] = s;
            }
            for (int i = n - 1; i >= 0; i--) {
                Type s = y[i];
                for (int j = i + 1; j < n; j++) s = s - lu(i, j) * X(j, col);
                X(i, col) = s / lu(i, i);
            }
        }
        return X;
    }
    Type logabsdet() const {
        Type s = Type(0.0);
        for (int i = 0; i < lu.r; i++) s = s + log(fabs(lu(i, i)));
        return s;
    }
};

// log|det| of an n x n column-major matrix of any size (penalty blocks can exceed MAXD):
// atomic::matinvpd's log-determinant, nllk_sde.hpp:110
inline double logabsdet_dyn(const double* S, int n) {
    std::vector<double> A(S, S + (size_t)n * n);
    double ld = 0.0;
    for (int k = 0; k < n; k++) {
        int piv = k;
        for (int i = k + 1; i < n; i++)
            if (std::fabs(A[i + (size_t)k * n]) > std::fabs(A[piv + (size_t)k * n])) piv = i;
        if (piv != k)
            for (int j = 0; j < n; j++) std::swap(A[k + (size_t)j * n], A[piv + (size_t)j * n]);
        ld += std::log(std::fabs(A[k + (size_t)k * n]));
        for (int i = k + 1; i < n; i++) {
            double f = A[i + (size_t)k * n] / A[k + (size_t)k * n];
            for (int j = k + 1; j < n; j++) A[i + (size_t)j * n] -= f * A[k + (size_t)j * n];
        }
    }
    return ld;
}

// det(): nllk_ctcrw.hpp:12-24
template <class Type>
Type det(const Mat<Type>& M) {
    int n_dim = M.c;
    if (n_dim == 1) return M(0, 0);
    if (n_dim == 2) return M(0, 0) * M(1, 1) - M(1, 0) * M(0, 1);
    return exp(LU<Type>(M).logabsdet());
}

// ---------------------------------------------------------------------------------------
// Problem view shared by all families.
// ---------------------------------------------------------------------------------------
struct Problem {
    const ssde_desc* d;
    int64_t row_lo, row_hi;  // evaluate rows [row_lo, row_hi) (a whole number of segments)
    int64_t seg_lo;          // index of the a0 row of the segment starting at row_lo
    int n_fe, n_re, n_lambda, n_decay;  // totals
    int off_sigobs, off_fe, off_lambda, off_decay, off_re, n_par_full;
    std::vector<int> decay_of_col;     // per coeff_re column: index into log_decay, or -1
    std::vector<int> fe_off, re_off;  // per SDE parameter offsets inside coeff_fe / coeff_re
};

inline bool is_kalman(int model) {
    return model == SSDE_MODEL_BM_SSM || model == SSDE_MODEL_OU_SSM || model == SSDE_MODEL_CTCRW;
}
inline int state_dim(const ssde_desc* d) {
    if (d->model == SSDE_MODEL_ESEAL_SSM) return 2;
    if (d->model == SSDE_MODEL_CTCRW) return 2 * d->n_dim;
    if (is_kalman(d->model)) return d->n_dim;
    return 0;
}

inline Problem make_problem(const ssde_desc* d) {
    Problem p;
    p.d = d;
    p.row_lo = 0;
    p.row_hi = d->n;
    p.seg_lo = 0;
    p.n_fe = p.n_re = 0;
    p.fe_off.resize(d->n_par);
    p.re_off.resize(d->n_par);
    for (int j = 0; j < d->n_par; j++) {
        p.fe_off[j] = p.n_fe;
        p.n_fe += d->ncol_fe[j];
        p.re_off[j] = p.n_re;
        p.n_re += d->ncol_re ? d->ncol_re[j] : 0;
    }
    p.n_lambda = d->n_smooth;
    int o = 0;
    p.off_sigobs = -1;
    if (is_kalman(d->model)) { p.off_sigobs = 0; o = 1; }  // PARAMETER(log_sigma_obs) first: nllk_ctcrw.hpp:135
    if (d->model == SSDE_MODEL_ESEAL_SSM) o = 3;           // PARAMETER(log_tau), (a1), (log_a2): nllk_e_seal_ssm.hpp:114-116
    p.off_fe = o; o += p.n_fe;
    p.off_lambda = o; o += p.n_lambda;
    p.n_decay = (!is_kalman(d->model) && d->n_decay > 0) ? d->n_decay : 0;     // PARAMETER_VECTOR(log_decay), nllk_sde.hpp:44
    p.off_decay = o; o += p.n_decay;
    p.off_re = o; o += p.n_re;
    p.n_par_full = o;
    p.decay_of_col.assign(p.n_re, -1);
    if (p.n_decay > 0)
        for (int c = 0; c < d->n_decay_cols; c++) p.decay_of_col[d->col_decay[c]] = d->ind_decay[c];   // :50-51
    return p;
}

// Linear predictor, one row: par_vec = X_fe * coeff_fe + X_re * coeff_re and the reshape
// to par_mat(i, j) (nllk_ctcrw.hpp:143-149; identical in nllk_ou_ssm.hpp:113-119,
// nllk_bm_ssm.hpp:80-86, nllk_sde.hpp:61-67).  The block-diagonal X_fe / X_re of
// R/sde.R:443-447 are passed as their diagonal blocks.
template <class Type>
Type linpred(const Problem& p, const Type* par, int64_t i, int j) {
    const ssde_desc* d = p.d;
    Type fe = Type(0.0);
    for (int c = 0; c < d->ncol_fe[j]; c++) {
        double x = d->x_fe[j] ? d->x_fe[j][i + (int64_t)c * d->n] : 1.0;
        fe = fe + par[p.off_fe + p.fe_off[j] + c] * x;
    }
    Type re = Type(0.0);
    int nre = d->ncol_re ? d->ncol_re[j] : 0;
    for (int c = 0; c < nre; c++) {
        const int k = p.n_decay > 0 ? p.decay_of_col[p.re_off[j] + c] : -1;
        if (k >= 0) {
            // X_re_copy.col = X_re.col * exp(-decay_rate * t_decay), decay_rate = exp(log_decay) (nllk_sde.hpp:47-57);
            // t_decay runs over the rows of the block-diagonal X_re: entry j*n + i
            Type decay = exp(-(exp(par[p.off_decay + k]) * d->t_decay[(int64_t)j * d->n + i]));
            re = re + par[p.off_re + p.re_off[j] + c] * (decay * d->x_re[j][i + (int64_t)c * d->n]);
        } else {
            re = re + par[p.off_re + p.re_off[j] + c] * d->x_re[j][i + (int64_t)c * d->n];
        }
    }
    return fe + re;
}

// dtimes for the Kalman families: nllk_ctcrw.hpp:126-129 (same nllk_ou_ssm.hpp:96-99,
// nllk_bm_ssm.hpp:63-66)
inline double dtimes_kalman(const ssde_desc* d, int64_t i) {
    return (i < d->n - 1) ? d->times[i + 1] - d->times[i] : 1.0;
}

// Smoothing penalty of the Kalman families: nllk_ctcrw.hpp:254-280 (same
// nllk_ou_ssm.hpp:220-246, nllk_bm_ssm.hpp:182-208).  GMRF(S).Quadform(x) = x' S x.
template <class Type>
Type penalty_kalman(const Problem& p, const Type* par) {
    const ssde_desc* d = p.d;
    Type pen = Type(0.0);
    if (d->n_smooth <= 0) return pen;  // ncol_re(0) > 0 test, line 256
    int S_start = 0;
    const double* Sb = d->s_blocks;
    for (int s = 0; s < d->n_smooth; s++) {
        int Sn = d->smooth_ncol[s];
        Type quad = Type(0.0);
        for (int a = 0; a < Sn; a++) {
            Type Sx = Type(0.0);
            for (int b = 0; b < Sn; b++) Sx = Sx + par[p.off_re + S_start + b] * Sb[a + b * Sn];
            quad = quad + par[p.off_re + S_start + a] * Sx;
        }
        Type ll = par[p.off_lambda + s];
        pen = pen - Type(0.5) * Type((double)Sn) * ll + Type(0.5) * exp(ll) * quad;
        S_start += Sn;
        Sb += Sn * Sn;
    }
    return pen;
}

// Smoothing penalty of nllk_sde: nllk_sde.hpp:89-124 (adds the Gaussian normalising
// constants; gated by include_penalty).  atomic::matinvpd returns log det(S).
template <class Type>
Type penalty_sde(const Problem& p, const Type* par) {
    const ssde_desc* d = p.d;
    Type pen = Type(0.0);
    if (!(d->n_smooth > 0 && d->include_penalty)) return pen;  // line 91
    int S_start = 0;
    const double* Sb = d->s_blocks;
    for (int s = 0; s < d->n_smooth; s++) {
        int Sn = d->smooth_ncol[s];
        double log_det = -logabsdet_dyn(Sb, Sn);  // line 110-111: det(S^-1) = 1/det(S)
        Type quad = Type(0.0);
        for (int a = 0; a < Sn; a++) {
            Type Sx = Type(0.0);
            for (int b = 0; b < Sn; b++) Sx = Sx + par[p.off_re + S_start + b] * Sb[a + b * Sn];
            quad = quad + par[p.off_re + S_start + a] * Sx;
        }
        Type ll = par[p.off_lambda + s];
        pen = pen + Type(0.5 * Sn * std::log(2.0 * M_PI)) + Type(0.5 * log_det) -
              Type(0.5) * Type((double)Sn) * ll + Type(0.5) * exp(ll) * quad;
        S_start += Sn;
        Sb += Sn * Sn;
    }
    return pen;
}

// ---------------------------------------------------------------------------------------
// Kalman families.  One function restates the three loops; the per-family pieces are
// the make* helpers, kept separate and cited.
// ---------------------------------------------------------------------------------------

// makeH_*: nllk_ctcrw.hpp:30-38, nllk_ou_ssm.hpp:15-23, nllk_bm_ssm.hpp:14-22
template <class Type>
Mat<Type> makeH(Type sigma_obs, int n_dim) {
    Mat<Type> H(n_dim, n_dim);
    for (int i = 0; i < n_dim; i++) H(i, i) = sigma_obs * sigma_obs;
    return H;
}
// makeT_ctcrw: nllk_ctcrw.hpp:45-55
template <class Type>
Mat<Type> makeT_ctcrw(Type beta, double dt, int n_dim) {
    Mat<Type> T(2 * n_dim, 2 * n_dim);
    for (int i = 0; i < n_dim; i++) {
        T(2 * i, 2 * i) = Type(1.0);
        T(2 * i, 2 * i + 1) = (Type(1.0) - exp(-beta * dt)) / beta;
        T(2 * i + 1, 2 * i + 1) = exp(-beta * dt);
    }
    return T;
}
// makeQ_ctcrw: nllk_ctcrw.hpp:63-75
template <class Type>
Mat<Type> makeQ_ctcrw(Type beta, Type sigma, double dt, int n_dim) {
    Mat<Type> Q(2 * n_dim, 2 * n_dim);
    for (int i = 0; i < n_dim; i++) {
        Q(2 * i, 2 * i) = (sigma / beta) * (sigma / beta) *
                          (Type(dt) - Type(2.0) / beta * (Type(1.0) - exp(-beta * dt)) +
                           Type(1.0) / (Type(2.0) * beta) * (Type(1.0) - exp(Type(-2.0) * beta * dt)));
        Q(2 * i, 2 * i + 1) = sigma * sigma / (Type(2.0) * beta * beta) *
                              (Type(1.0) - Type(2.0) * exp(-beta * dt) + exp(Type(-2.0) * beta * dt));
        Q(2 * i + 1, 2 * i) = Q(2 * i, 2 * i + 1);
        Q(2 * i + 1, 2 * i + 1) = sigma * sigma / (Type(2.0) * beta) * (Type(1.0) - exp(Type(-2.0) * beta * dt));
    }
    return Q;
}
// makeB_ctcrw: nllk_ctcrw.hpp:82-91
template <class Type>
Mat<Type> makeB_ctcrw(Type beta, double dt, int n_dim) {
    Mat<Type> B(2 * n_dim, n_dim);
    for (int i = 0; i < n_dim; i++) {
        B(2 * i, i) = Type(dt) - (Type(1.0) - exp(-beta * dt)) / beta;
        B(2 * i + 1, i) = Type(1.0) - exp(-beta * dt);
    }
    return B;
}
// makeT/B/Q_ou_ssm: nllk_ou_ssm.hpp:30-69
template <class Type>
Mat<Type> makeT_ou(Type tau, double dt, int n_dim) {
    Mat<Type> T(n_dim, n_dim);
    for (int i = 0; i < n_dim; i++) T(i, i) = exp(Type(-dt) / tau);
    return T;
}
template <class Type>
Mat<Type> makeB_ou(Type tau, double dt, int n_dim) {
    Mat<Type> B(n_dim, n_dim);
    for (int i = 0; i < n_dim; i++) B(i, i) = Type(1.0) - exp(Type(-dt) / tau);
    return B;
}
template <class Type>
Mat<Type> makeQ_ou(Type tau, Type kappa, double dt, int n_dim) {
    Mat<Type> Q(n_dim, n_dim);
    for (int i = 0; i < n_dim; i++) Q(i, i) = kappa * (Type(1.0) - exp(Type(-2.0 * dt) / tau));
    return Q;
}
// makeQ_bm_ssm: nllk_bm_ssm.hpp:28-36
template <class Type>
Mat<Type> makeQ_bm(Type sigma, double dt, int n_dim) {
    Mat<Type> Q(n_dim, n_dim);
    for (int i = 0; i < n_dim; i++) Q(i, i) = sigma * sigma * dt;
    return Q;
}

// Default a0 / P0 packing of SDE$setup: R/sde.R:547-557 (BM_SSM / OU_SSM), :574-587 (CTCRW)
inline double a0_entry(const ssde_desc* d, int64_t seg, int64_t first_row, int comp) {
    int sdim = state_dim(d);
    (void)sdim;
    if (d->a0) return d->a0[seg + (int64_t)comp * d->n_seg];
    if (d->model == SSDE_MODEL_CTCRW) {
        if (comp % 2 == 1) return 0.0;                              // velocities 0 (R/sde.R:576)
        return d->obs[first_row + (int64_t)(comp / 2) * d->n];      // positions = first obs (:577-579)
    }
    return d->obs[first_row + (int64_t)comp * d->n];                // R/sde.R:549
}
inline double p0_entry(const ssde_desc* d, int i, int j) {
    int sdim = state_dim(d);
    if (d->p0) return d->p0[i + j * sdim];
    if (i != j) return 0.0;
    if (d->model == SSDE_MODEL_CTCRW) return (i % 2 == 0) ? 1.0 : 10.0;  // diag(rep(c(1,10), n_dim)), R/sde.R:584
    return 10.0;                                                          // diag(rep(10, n_dim)),  R/sde.R:554
}

// nllk_ctcrw (nllk_ctcrw.hpp:102-283), nllk_ou_ssm (nllk_ou_ssm.hpp:73-249),
// nllk_bm_ssm (nllk_bm_ssm.hpp:40-211): returns -llk WITHOUT the penalty (added by the
// caller so that track shards can be summed).  aest_all (n x sdim column-major) may be
// NULL; it is the REPORT(aest_all) output.
template <class Type>
Type nllk_kalman(const Problem& p, const Type* par, double* aest_all) {
    const ssde_desc* d = p.d;
    const int model = d->model;
    const int n_dim = d->n_dim;
    const int sdim = state_dim(d);
    const int64_t n = d->n;

    Type sigma_obs = exp(par[p.off_sigobs]);  // nllk_ctcrw.hpp:135-136

    // Z: nllk_ctcrw.hpp:162-166; identity for OU/BM (nllk_ou_ssm.hpp:130-131)
    Mat<Type> Z(n_dim, sdim);
    for (int i = 0; i < n_dim; i++) Z(i, model == SSDE_MODEL_CTCRW ? 2 * i : i) = Type(1.0);
    Mat<Type> Zt = transpose(Z);
    Mat<Type> H = makeH(sigma_obs, n_dim);  // line 167

    Mat<Type> P0(sdim, sdim);
    for (int i = 0; i < sdim; i++)
        for (int j = 0; j < sdim; j++) P0(i, j) = Type(p0_entry(d, i, j));

    Mat<Type> aest(sdim, 1), Pest(sdim, sdim);
    int64_t k = p.seg_lo;
    Type llk = Type(0.0);

    for (int64_t i = p.row_lo; i < p.row_hi; i++) {
        bool first = (i == p.row_lo) || (d->id[i] != d->id[i - 1]);  // lines 182-188 and 196
        if (first) {
            for (int c = 0; c < sdim; c++) aest(c, 0) = Type(a0_entry(d, k, i, c));  // aest = a0.row(k)
            k = k + 1;
            Pest = P0;
        } else {
            if (d->h_array) {  // lines 203-205: H = H_array.col(i).matrix()
                for (int a = 0; a < n_dim; a++)
                    for (int b = 0; b < n_dim; b++)
                        H(a, b) = Type(d->h_array[a + b * n_dim + (int64_t)i * n_dim * n_dim]);
            }
            double dt = dtimes_kalman(d, i);
            Mat<Type> T(sdim, sdim), Q(sdim, sdim), drift(sdim, 1);
            if (model == SSDE_MODEL_CTCRW) {
                // lines 152-156: tau, nu, beta, sigma; lines 206-212
                Type tau = exp(linpred(p, par, i, n_dim));
                Type nu = exp(linpred(p, par, i, n_dim + 1));
                Type beta = Type(1.0) / tau;
                Type sigma = Type(2.0) * nu / sqrt(Type(M_PI) * tau);
                T = makeT_ctcrw(beta, dt, n_dim);
                Q = makeQ_ctcrw(beta, sigma, dt, n_dim);
                Mat<Type> B = makeB_ctcrw(beta, dt, n_dim);
                Mat<Type> mu_i(n_dim, 1);
                for (int a = 0; a < n_dim; a++) mu_i(a, 0) = linpred(p, par, i, a);
                drift = mul(B, mu_i);
            } else if (model == SSDE_MODEL_OU_SSM) {
                // nllk_ou_ssm.hpp:122-124, 174-177
                Type tau = exp(linpred(p, par, i, n_dim));
                Type kappa = exp(linpred(p, par, i, n_dim + 1));
                T = makeT_ou(tau, dt, n_dim);
                Mat<Type> B = makeB_ou(tau, dt, n_dim);
                Q = makeQ_ou(tau, kappa, dt, n_dim);
                Mat<Type> mu_i(n_dim, 1);
                for (int a = 0; a < n_dim; a++) mu_i(a, 0) = linpred(p, par, i, a);
                drift = mul(B, mu_i);
            } else {
                // nllk_bm_ssm.hpp:89-90, 99-100 (T identity), 138-139
                Type sigma = exp(linpred(p, par, i, n_dim));
                for (int a = 0; a < n_dim; a++) T(a, a) = Type(1.0);
                Q = makeQ_bm(sigma, dt, n_dim);
                for (int a = 0; a < n_dim; a++) drift(a, 0) = linpred(p, par, i, a) * dt;
            }
            Mat<Type> Tt = transpose(T);

            if (is_na(d->obs[i], d->na_mode)) {  // column 0 only: line 214
                aest = add(mul(T, aest), drift);
                Pest = add(mul(mul(T, Pest), Tt), Q);
            } else {
                Mat<Type> u(n_dim, 1);
                Mat<Type> Za = mul(Z, aest);
                for (int a = 0; a < n_dim; a++) u(a, 0) = Type(d->obs[i + (int64_t)a * n]) - Za(a, 0);  // line 221
                Mat<Type> F = add(mul(mul(Z, Pest), Zt), H);                                          // line 223
                Type detF;
                if (model == SSDE_MODEL_CTCRW) detF = det(F);                // nllk_ctcrw.hpp:224
                else detF = exp(LU<Type>(F).logabsdet());                    // nllk_ou_ssm.hpp:190, nllk_bm_ssm.hpp:152
                if (detF <= 0.0) {
                    // Q3: CTCRW drops B*mu here (nllk_ctcrw.hpp:226-228); OU/BM keep it
                    if (model == SSDE_MODEL_CTCRW) aest = mul(T, aest);
                    else aest = add(mul(T, aest), drift);
                    Pest = add(mul(mul(T, Pest), Tt), Q);
                } else {
                    Mat<Type> Finv = LU<Type>(F).inverse();
                    Mat<Type> FinvT = transpose(Finv);            // line 231
                    Mat<Type> FinvTu = mul(FinvT, u);             // line 232
                    Type uFu = Type(0.0);
                    for (int a = 0; a < n_dim; a++) uFu = uFu + u(a, 0) * FinvTu(a, 0);  // line 233
                    llk = llk - (log(detF) + uFu) / Type(2.0);    // line 234
                    Mat<Type> K = mul(mul(mul(T, Pest), Zt), Finv);      // line 236
                    aest = add(add(mul(T, aest), mul(K, u)), drift);    // line 238
                    Mat<Type> L = sub(T, mul(K, Z));                     // line 240
                    Pest = add(mul(mul(T, Pest), transpose(L)), Q);     // line 241
                    if (keep_P_symmetric())                             // (arbiter mode only: see the top of this file)
                        for (int a = 0; a < sdim; a++)
                            for (int b = a + 1; b < sdim; b++) { const Type m = (Pest(a, b) + Pest(b, a)) / Type(2.0); Pest(a, b) = m; Pest(b, a) = m; }
                }
            }
        }
        if (aest_all)
            for (int c = 0; c < sdim; c++) aest_all[i + (int64_t)c * n] = asDouble(aest(c, 0));  // line 246
    }
    return -llk;  // line 254
}

// dnorm(x, mean, sd, true): TMB's definition (not in the repository; restated):
//   resid = (x - mean)/sd;  -log(sqrt(2*pi)) - log(sd) - resid^2/2
template <class Type>
Type dnorm_log(Type x, Type mean, Type sd) {
    Type resid = (x - mean) / sd;
    return Type(-std::log(std::sqrt(2.0 * M_PI))) - log(sd) - Type(0.5) * resid * resid;
}

// log(besselI(x, nu)) of tr_dens.hpp:64-66 for a real order nu > -1.  TMB's besselI is not in the container; the
// published ascending series I_nu(x) = (x/2)^nu sum_k (x^2/4)^k / (k! Gamma(nu+k+1)) is summed here (all terms
// positive) with the logarithm taken term-wise so that large x does not overflow (the reference's unscaled besselI
// returns Inf beyond x ~ 700; where it is finite the two agree to rounding).  Templated: duals differentiate it
// with respect to x AND nu, as TMB's AD does.
template <class Type>
Type log_besselI(Type x, Type nu) {
    // I_nu(x) = (x/2)^nu sum_k t_k, t_k = (x^2/4)^k / (k! Gamma(k+nu+1)) (all terms positive), summed outwards from
    // the largest term k* in units of t_k*: the same sum in any order, ~sqrt(x) terms, never overflows
    Type y = x * x * Type(0.25);
    const double xd = asDouble(x), nd = asDouble(nu);
    double ks = std::floor(0.5 * (std::sqrt(nd * nd + xd * xd) - nd));
    if (!(ks >= 1.0)) ks = 0.0;
    // Gaussian fall-off around k* with variance k*(k*+nu)/(2k*+nu); NaN (a rejected step) beyond x ~ 3e7, where the
    // walk would take > 36000 steps per side
    const double sd = std::sqrt(ks * (ks + nd) / (2.0 * ks + nd + 1e-300));
    if (!(sd < 3.0e3)) return Type(NAN) * x * nu;      // (NaN in the value and in every derivative)
    const int cap = 100 + (int)(12.0 * sd);
    Type S = Type(1.0), t = Type(1.0);
    double k = ks;
    for (int it = 0; it < cap; it++) {
        k += 1.0;
        t = t * y / (Type(k) * (Type(k) + nu));
        S = S + t;
        if (!(asDouble(t) >= 1e-18 * asDouble(S))) break;
    }
    t = Type(1.0);
    k = ks;
    for (int it = 0; it < cap && k >= 1.0; it++) {
        t = t * (Type(k) * (Type(k) + nu)) / y;
        k -= 1.0;
        S = S + t;
        if (!(asDouble(t) >= 1e-18 * asDouble(S))) break;
    }
    Type lt = Type(-1.0) * lgamma(Type(ks + 1.0) + nu);
    if (ks > 0.0) lt = lt + Type(ks) * log(y) - Type(std::lgamma(ks + 1.0));
    return nu * log(x * Type(0.5)) + lt + log(S);
}

// nllk_sde (nllk_sde.hpp:16-127) with tr_dens' BM (tr_dens.hpp:32-37), BM_t (:38-44), OU (:45-52) and CIR (:53-67)
// branches; returns -llk without the penalty.
template <class Type>
Type nllk_direct(const Problem& p, const Type* par) {
    const ssde_desc* d = p.d;
    const int n_dim = d->n_dim;
    const int64_t n = d->n;
    Type llk = Type(0.0);
    int64_t lo = p.row_lo > 0 ? p.row_lo : 1;
    for (int64_t i = lo; i < p.row_hi; i++) {
        if (i == p.row_lo) continue;                 // first row of a shard is a first row of a track
        if (!(d->id[i - 1] == d->id[i])) continue;   // nllk_sde.hpp:79
        double dt = d->times[i] - d->times[i - 1];   // dtimes(i-1), nllk_sde.hpp:37,80
        Type res = Type(0.0);
        for (int a = 0; a < n_dim; a++) {
            double z0 = d->obs[(i - 1) + (int64_t)a * n], z1 = d->obs[i + (int64_t)a * n];
            if (is_na(z0, d->na_mode) || is_na(z1, d->na_mode)) continue;  // tr_dens.hpp:31
            if (d->model == SSDE_MODEL_BM_T) {
                const double df = d->other_data[0];                                    // tr_dens.hpp:40
                Type mean = linpred(p, par, i - 1, 0) * dt;                            // :41 (par(0), par(1) whatever i)
                Type sd = exp(linpred(p, par, i - 1, 1)) * std::sqrt(dt);              // :42
                Type scale = sd / std::sqrt(df / (df - 2.0));                          // :43
                Type x = (Type(z1) - Type(z0) - mean) / scale;
                // R's dt(x, df, log = TRUE) = lgamma((df+1)/2) - lgamma(df/2) - log(df pi)/2 - (df+1)/2 log(1 + x^2/df)
                Type logdt = Type(std::lgamma(0.5 * (df + 1.0)) - std::lgamma(0.5 * df) - 0.5 * std::log(df * M_PI)) -
                             Type(0.5 * (df + 1.0)) * log(Type(1.0) + x * x / df);
                res = res + logdt - log(scale);                                        // :44
            } else if (d->model == SSDE_MODEL_CIR) {
                Type mu = exp(linpred(p, par, i - 1, a));                              // tr_dens.hpp:56
                Type beta = exp(linpred(p, par, i - 1, n_dim));                        // :57
                Type sigma = exp(linpred(p, par, i - 1, n_dim + 1));                   // :58
                Type c = Type(2.0) * beta / ((Type(1.0) - exp(-(beta * dt))) * sigma * sigma);   // :60
                Type q = Type(2.0) * beta * mu / (sigma * sigma) - Type(1.0);          // :61
                Type u = c * z0 * exp(-(beta * dt));                                   // :62
                Type v = c * z1;                                                       // :63
                Type logb = log_besselI(Type(2.0) * sqrt(u * v), q);                   // :64, log(b) of :66
                res = res + log(c) - u - v + q / Type(2.0) * (log(v) - log(u)) + logb; // :66
            } else if (d->model == SSDE_MODEL_BM) {
                Type mean = Type(z0) + linpred(p, par, i - 1, a) * dt;                 // tr_dens.hpp:35
                Type sd = exp(linpred(p, par, i - 1, n_dim)) * std::sqrt(dt);          // :36
                res = res + dnorm_log(Type(z1), mean, sd);                             // :37
            } else {
                Type mu = linpred(p, par, i - 1, a);
                Type ltau = linpred(p, par, i - 1, n_dim);
                Type lkap = linpred(p, par, i - 1, n_dim + 1);
                Type mean = mu + exp(Type(-dt) / exp(ltau)) * (Type(z0) - mu);         // tr_dens.hpp:49
                Type sd = sqrt(exp(lkap) * (Type(1.0) - exp(Type(-2.0 * dt) / exp(ltau))));  // :50-51
                res = res + dnorm_log(Type(z1), mean, sd);                             // :52
            }
        }
        llk = llk + res;  // nllk_sde.hpp:82
    }
    return -llk;
}

// nllk_eseal_ssm (nllk_e_seal_ssm.hpp:83-250), Kalman loop only (lines 139-207): state (1, L), time-varying
// Z_i = (a1, a2 / R_i) (makeZ :43-48), H_i = tau^2 / h_i (:55-59), T_i = [[1, 0], [mu_i dt_i, 1]] (:16-23),
// Q_i = diag(0, sigma_i^2 dt_i) (:30-35).  Returns -llk without priors / penalty.
template <class Type>
Type nllk_eseal(const Problem& p, const Type* par) {
    const ssde_desc* d = p.d;
    Type tau = exp(par[0]), a1 = par[1], a2 = exp(par[2]);                       // :114-118
    Mat<Type> P0(2, 2);
    for (int i = 0; i < 2; i++)
        for (int j = 0; j < 2; j++) P0(i, j) = Type(d->p0 ? d->p0[i + j * 2] : ((i == 1 && j == 1) ? 10.0 : 0.0));   // R/sde.R:603
    Mat<Type> aest(2, 1), Pest(2, 2);
    int64_t k = p.seg_lo;
    Type llk = Type(0.0);
    for (int64_t i = p.row_lo; i < p.row_hi; i++) {
        bool first = (i == p.row_lo) || (d->id[i] != d->id[i - 1]);             // :154-158, 163
        if (first) {
            for (int c = 0; c < 2; c++) aest(c, 0) = Type(d->a0[k + (int64_t)c * d->n_seg]);
            k = k + 1;
            Pest = P0;
            continue;
        }
        Mat<Type> Z(1, 2), H(1, 1), T(2, 2), Q(2, 2);
        Z(0, 0) = a1; Z(0, 1) = a2 / Type(d->eseal_R[i]);                         // :170
        H(0, 0) = tau * tau / Type(d->eseal_h[i]);                                // :171
        double dt = dtimes_kalman(d, i);                                          // :104-107
        Type mu = linpred(p, par, i, 0), sigma = exp(linpred(p, par, i, 1));     // :136-137
        T(0, 0) = Type(1.0); T(1, 0) = mu * dt; T(1, 1) = Type(1.0);              // :172
        Q(1, 1) = sigma * sigma * dt;                                             // :173
        Mat<Type> Tt = transpose(T), Zt = transpose(Z);
        if (is_na(d->obs[i], d->na_mode)) {                                       // :175-178
            aest = mul(T, aest);
            Pest = add(mul(mul(T, Pest), Tt), Q);
            continue;
        }
        Mat<Type> Za = mul(Z, aest);
        Type u = Type(d->obs[i]) - Za(0, 0);                                      // :181-182
        Mat<Type> F = add(mul(mul(Z, Pest), Zt), H);                              // :184
        Type detF = F(0, 0);                                                      // :185
        if (detF <= 0.0) {                                                        // :187-189
            aest = mul(T, aest);
            Pest = add(mul(mul(T, Pest), Tt), Q);
        } else {
            Type Finv = Type(1.0) / F(0, 0);
            llk = llk - (log(detF) + u * Finv * u) / Type(2.0);                   // :192-195
            Mat<Type> K = mul(mul(T, Pest), Zt);                                  // :197 (2 x 1)
            K(0, 0) = K(0, 0) * Finv; K(1, 0) = K(1, 0) * Finv;
            Mat<Type> Ku(2, 1);
            Ku(0, 0) = K(0, 0) * u; Ku(1, 0) = K(1, 0) * u;
            aest = add(mul(T, aest), Ku);                                         // :199
            Mat<Type> L = sub(T, mul(K, Z));                                      // :201
            Pest = add(mul(mul(T, Pest), transpose(L)), Q);                       // :202
        }
    }
    return -llk;
}

// dinvgamma(x, shape, scale, log) of nllk_e_seal_ssm.hpp:68-78
template <class Type>
Type dinvgamma_log(Type x, double shape, double scale) {
    return Type(shape * std::log(scale) - std::lgamma(shape)) - Type(shape + 1.0) * log(x) - Type(scale) / x;
}
// -(priors) of lines 212-216: on sigma(0)^2 (first ROW's sigma) and tau^2; integer division n/2 as in the reference
template <class Type>
Type eseal_priors(const Problem& p, const Type* par) {
    const ssde_desc* d = p.d;
    const long n = (long)d->n;
    Type sigma0 = exp(linpred(p, par, 0, 1)), tau = exp(par[0]);
    Type lp = dinvgamma_log(sigma0 * sigma0, (double)(10 * n), (double)(4 * (10 * n - 1))) +
              dinvgamma_log(tau * tau, (double)(n / 2), (double)(n / 2 - 1));
    return -lp;
}

template <class Type>
Type nllk_data(const Problem& p, const Type* par, double* aest_all) {
    if (p.d->model == SSDE_MODEL_ESEAL_SSM) return nllk_eseal<Type>(p, par);
    if (is_kalman(p.d->model)) return nllk_kalman<Type>(p, par, aest_all);
    return nllk_direct<Type>(p, par);
}
// parameter-only terms, added once: smoothing penalty (+ the ESEAL priors)
template <class Type>
Type penalty(const Problem& p, const Type* par) {
    if (p.d->model == SSDE_MODEL_ESEAL_SSM) return penalty_kalman<Type>(p, par) + eseal_priors<Type>(p, par);   // :218-246
    if (is_kalman(p.d->model)) return penalty_kalman<Type>(p, par);
    return penalty_sde<Type>(p, par);
}

}  // namespace ssde_oracle
#endif
