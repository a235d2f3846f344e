// k_direct_hess.hip -- EXACT second derivatives of the data term of the direct families "BM", "OU" and "BM_t" with respect to
// coefficients of the linear predictor (nllk_sde.hpp:61-84 with tr_dens.hpp:32-37, 45-52), for gfx950.
//
// What the reference gets from TMB's second-order AD (tmb_obj_joint$he, R/sde.R:1363; the Laplace approximation's
// H_uu behind random = "coeff_re", R/sde.R:510-525, 656-658).  The SDE parameters of a row are LINEAR in the
// coefficients, par_j(i) = sum_k X_k(i) coef_k, so
//     d2 nllk / d coef_k d coef_l = sum_i X_k(i) X_l(i) D_i[j(k), j(l)],      D_i = d2 (-log dens_i) / d par d par'
// exactly (the Gauss-Newton form IS the Hessian here), and D_i is closed-form for the Gaussian transitions:
//   BM   r = (z1 - z0 - mu dt) / sd, sd = e^{ls} sqrt(dt):   D[mu,mu] = dt^2 / sd^2,  D[mu,ls] = 2 r dt / sd,  D[ls,ls] = 2 r^2
//   OU   l = log(v)/2 + delta^2 / (2 v),  delta = z1 - mu - e (z0 - mu),  v = kappa (1 - e^2),  e = exp(-dt / tau):  chain rule
//        through (delta, v) with  d e / d log tau = e z,  d2 e / d log tau^2 = e z (z - 1),  z = dt / tau.
// Lane = row on the long format (the layout of k_direct.hip); the wanted coefficient pairs are cut into HT x HT tiles,
// one tile per blockIdx.y, accumulated in registers and reduced in a fixed order (bitwise reproducible).  Not a hot
// kernel: it runs a handful of times per marginal-likelihood evaluation.
#include "ssde_device.hpp"
#include "ssde_hdual.hpp"

namespace ssde {

template <int MODEL, int D>
__device__ __forceinline__ void row_hessian(const DirectHessArgs& A, int64_t i, double dt, const double* par, double (&Dm)[MAX_Q][MAX_Q],
                                            double (&gm)[MAX_Q]) {
    // Dm = d2 (-log dens_i) / d par d par', gm = d (-log dens_i) / d par (used by the decaying-column variant only)
#pragma unroll
    for (int p = 0; p < MAX_Q; p++) {
        gm[p] = 0.0;
#pragma unroll
        for (int q = 0; q < MAX_Q; q++) Dm[p][q] = 0.0;
    }
    if (MODEL == M_CIR) {
        // CIR (tr_dens.hpp:53-67): -log dens = -(log c - u - v + q/2 (log v - log u) + log I_q(2 sqrt(u v))) in (log mu_a, log beta,
        // log sigma).  log I_q and its five derivatives once per (row, dimension) from the series (log_bessel_i2); the rest of the
        // density is run in hyper-dual arithmetic, one pass per parameter pair -- nothing derived by hand, nothing differenced.
#pragma unroll
        for (int a = 0; a < D; a++) {
            const double z0 = A.obs[(i - 1) + (int64_t)a * A.n], z1 = A.obs[i + (int64_t)a * A.n];
            if (is_na(z0, A.any_nan) || is_na(z1, A.any_nan)) continue;              // tr_dens.hpp:31
            double lb5[5], lI = 0.0;
            {
                const double mu = exp(par[a]), beta = exp(par[D]), s2 = exp(2.0 * par[D + 1]);
                const double em = exp(-beta * dt), c = 2.0 * beta / ((1.0 - em) * s2);
                lI = log_bessel_i2(2.0 * sqrt(c * z0 * em * c * z1), 2.0 * beta * mu / s2 - 1.0, lb5);
            }
            const int pos[3] = {a, D, D + 1};
#pragma unroll
            for (int pa = 0; pa < 3; pa++)
#pragma unroll
                for (int pb = pa; pb < 3; pb++) {
                    const HD lm(par[a], pa == 0 ? 1.0 : 0.0, pb == 0 ? 1.0 : 0.0, 0.0);
                    const HD lbt(par[D], pa == 1 ? 1.0 : 0.0, pb == 1 ? 1.0 : 0.0, 0.0);
                    const HD ls(par[D + 1], pa == 2 ? 1.0 : 0.0, pb == 2 ? 1.0 : 0.0, 0.0);
                    const HD mu = hd_exp(lm), beta = hd_exp(lbt), s2 = hd_exp(2.0 * ls);
                    const HD em = hd_exp(-(beta * dt));
                    const HD c = 2.0 * beta / ((1.0 - em) * s2);                          // :60
                    const HD q = 2.0 * beta * mu / s2 - 1.0;                              // :61
                    const HD u = c * (z0 * em), v = c * z1;                                // :62-63
                    const HD x = 2.0 * hd_sqrt(u * v);                                     // :64
                    const HD lbes = hd_chain2(x, q, lI, lb5[0], lb5[1], lb5[2], lb5[3], lb5[4]);
                    const HD ld = hd_log(c) - u - v + 0.5 * (q * (hd_log(v) - hd_log(u))) + lbes;   // :66
                    if (pa == pb) {
                        if (pa == 0) { Dm[a][a] = -ld.ab; gm[a] = -ld.a; }
                        else { Dm[pos[pa]][pos[pa]] -= ld.ab; gm[pos[pa]] -= ld.a; }
                    } else if (pa == 0) {
                        Dm[a][pos[pb]] = Dm[pos[pb]][a] = -ld.ab;
                    } else {
                        Dm[D][D + 1] -= ld.ab;
                    }
                }
        }
        Dm[D + 1][D] = Dm[D][D + 1];
    } else
    if (MODEL == M_BM_T) {
        // BM_t (tr_dens.hpp:38-44; one response column): l = phi(x) + log scale + const, phi(x) = (df + 1) / 2 log(1 + x^2 / df),
        // x = (z1 - z0 - mu dt) / scale, scale = e^{ls} sqrt(dt) / sqrt(df / (df - 2)):  d x / d mu = -dt / scale,  d x / d ls = -x
        const double df = A.tdf;
        const double scale = exp(par[1]) * sqrt(dt) / sqrt(df / (df - 2.0)), isc = 1.0 / scale;
        const double z0 = A.obs[i - 1], z1 = A.obs[i];
        if (!(is_na(z0, A.any_nan) || is_na(z1, A.any_nan))) {
            const double x = (z1 - z0 - par[0] * dt) * isc, den = 1.0 / (df + x * x);
            const double p1 = (df + 1.0) * x * den, p2 = (df + 1.0) * (df - x * x) * den * den;      // phi', phi''
            Dm[0][0] = p2 * dt * dt * isc * isc;
            Dm[0][1] = Dm[1][0] = (p2 * x + p1) * dt * isc;
            Dm[1][1] = (p2 * x + p1) * x;
            gm[0] = -p1 * dt * isc; gm[1] = 1.0 - p1 * x;
        }
    } else
    if (MODEL == M_BM) {
        const double sd = exp(par[D]) * sqrt(dt), isd = 1.0 / sd;
#pragma unroll
        for (int a = 0; a < D; a++) {
            const double z0 = A.obs[(i - 1) + (int64_t)a * A.n], z1 = A.obs[i + (int64_t)a * A.n];
            if (is_na(z0, A.any_nan) || is_na(z1, A.any_nan)) continue;          // tr_dens.hpp:31
            const double r = (z1 - (z0 + par[a] * dt)) * isd;
            Dm[a][a] = dt * dt * isd * isd;
            Dm[a][D] = Dm[D][a] = 2.0 * r * dt * isd;
            Dm[D][D] += 2.0 * r * r;
            gm[a] = -r * dt * isd; gm[D] += 1.0 - r * r;
        }
    } else {
        const double tau = exp(par[D]), kap = exp(par[D + 1]);
        const double z = dt / tau, e = exp(-z);
        const double e1 = e * z, e2 = e * z * (z - 1.0);                           // d e / d log tau, d2 e / d log tau^2
        const double w = 1.0 - e * e, v = kap * w, iv = 1.0 / v;
        const double v_t = -2.0 * kap * e * e1, v_tt = -2.0 * kap * (e1 * e1 + e * e2);   // v_k = v_kk = v, v_tk = v_t
#pragma unroll
        for (int a = 0; a < D; a++) {
            const double z0 = A.obs[(i - 1) + (int64_t)a * A.n], z1 = A.obs[i + (int64_t)a * A.n];
            if (is_na(z0, A.any_nan) || is_na(z1, A.any_nan)) continue;
            const double mu = par[a];
            const double dl = z1 - (mu + e * (z0 - mu));
            const double d_m = -(1.0 - e), d_t = -(z0 - mu) * e1, d_mt = e1, d_tt = -(z0 - mu) * e2;
            const double L_d = dl * iv, L_v = 0.5 * iv - 0.5 * dl * dl * iv * iv;
            const double L_dd = iv, L_dv = -dl * iv * iv, L_vv = -0.5 * iv * iv + dl * dl * iv * iv * iv;
            Dm[a][a] = L_dd * d_m * d_m;
            const double h_mt = L_dd * d_m * d_t + L_dv * d_m * v_t + L_d * d_mt;
            const double h_mk = L_dv * d_m * v;
            Dm[a][D] = Dm[D][a] = h_mt;
            Dm[a][D + 1] = Dm[D + 1][a] = h_mk;
            Dm[D][D] += L_dd * d_t * d_t + 2.0 * L_dv * d_t * v_t + L_vv * v_t * v_t + L_d * d_tt + L_v * v_tt;
            const double h_tk = L_dv * d_t * v + L_vv * v_t * v + L_v * v_t;
            Dm[D][D + 1] += h_tk;
            Dm[D + 1][D + 1] += L_vv * v * v + L_v * v;
            gm[a] = L_d * d_m; gm[D] += L_d * d_t + L_v * v_t; gm[D + 1] += L_v * v;
        }
        Dm[D + 1][D] = Dm[D][D + 1];
    }
}

__device__ __forceinline__ double pick4(const double (&r)[MAX_Q], int j) {      // (uniform j: scalar selects, no scratch)
    return j == 0 ? r[0] : j == 1 ? r[1] : j == 2 ? r[2] : r[3];
}

template <int MODEL, int D>
__global__ __launch_bounds__(256) void direct_hess_kernel(const DirectHessArgs A) {
    const SlotTable* __restrict__ T = A.slots;
    const int ns = A.n_slots;
    const int tile = blockIdx.y, ti = A.tile_i[tile], tj = A.tile_j[tile];
    double acc[HESS_T][HESS_T];
#pragma unroll
    for (int a = 0; a < HESS_T; a++)
#pragma unroll
        for (int b = 0; b < HESS_T; b++) acc[a][b] = 0.0;
    // the tile's coefficients: slot, parameter, column pointer (NULL = intercept: X = 1)
    int ja[HESS_T], jb[HESS_T];
    const double *ca[HESS_T], *cb[HESS_T];
    bool oa[HESS_T], ob[HESS_T];
#pragma unroll
    for (int a = 0; a < HESS_T; a++) {
        const int ka = ti * HESS_T + a, kb = tj * HESS_T + a;
        oa[a] = ka < A.nu; ob[a] = kb < A.nu;
        const int sa = oa[a] ? A.uslot[ka] : 0, sb = ob[a] ? A.uslot[kb] : 0;
        ja[a] = T->par_j[sa]; jb[a] = T->par_j[sb];
        ca[a] = T->col[sa] >= 0 ? A.cols[T->col[sa]] : nullptr;
        cb[a] = T->col[sb] >= 0 ? A.cols[T->col[sb]] : nullptr;
    }
    // the slot table once per workgroup, in LDS: column pointer (NULL = intercept), coefficient, SDE parameter
    __shared__ const double* s_col[MAX_COLS];
    __shared__ double s_coef[MAX_COLS];
    __shared__ int s_pj[MAX_COLS];
    for (int k = threadIdx.x; k < ns; k += 256) {
        const int c = T->col[k];
        s_col[k] = c >= 0 ? A.cols[c] : nullptr;
        s_coef[k] = A.par[T->pidx[k]];
        s_pj[k] = T->par_j[k];
    }
    __syncthreads();
    const int64_t per_block = ((A.n + gridDim.x - 1) / gridDim.x + 255) / 256 * 256;
    const int64_t row_lo = (int64_t)blockIdx.x * per_block;
    const int64_t row_hi = row_lo + per_block < A.n ? row_lo + per_block : A.n;
    for (int64_t i = row_lo + threadIdx.x; i < row_hi; i += 256) {
        if (!((A.scored[i >> 5] >> (i & 31)) & 1u)) continue;
        const double dt = A.times[i] - A.times[i - 1];                             // dtimes(i-1), nllk_sde.hpp:37, 80
        double par[MAX_Q] = {0.0, 0.0, 0.0, 0.0};
        for (int k = 0; k < ns; k++) {                                             // the linear predictor of row i-1 (Q6)
            const double* cp = s_col[k];
            const int j = s_pj[k];
            const double t = (cp ? cp[i - 1] : 1.0) * s_coef[k];
            par[0] += j == 0 ? t : 0.0; par[1] += j == 1 ? t : 0.0; par[2] += j == 2 ? t : 0.0; par[3] += j == 3 ? t : 0.0;
        }
        double Dm[MAX_Q][MAX_Q], gm[MAX_Q];
        row_hessian<MODEL, D>(A, i, dt, par, Dm, gm);
        double xb[HESS_T];
#pragma unroll
        for (int b = 0; b < HESS_T; b++) xb[b] = !ob[b] ? 0.0 : (cb[b] ? cb[b][i - 1] : 1.0);
#pragma unroll
        for (int a = 0; a < HESS_T; a++) {
            if (!oa[a]) continue;                                                  // (uniform)
            const double xa = ca[a] ? ca[a][i - 1] : 1.0;
            double row[MAX_Q];                                                     // X_a D[j_a, .]
#pragma unroll
            for (int q = 0; q < MAX_Q; q++) row[q] = xa * (ja[a] == 0 ? Dm[0][q] : ja[a] == 1 ? Dm[1][q] : ja[a] == 2 ? Dm[2][q] : Dm[3][q]);
#pragma unroll
            for (int b = 0; b < HESS_T; b++) acc[a][b] = fma(pick4(row, jb[b]), xb[b], acc[a][b]);
        }
    }
    // workgroup sums in a fixed order (wave shuffles, then the four waves): [tile][a * HESS_T + b][block]
    __shared__ double sh[HESS_T * HESS_T][4];
#pragma unroll
    for (int a = 0; a < HESS_T; a++)
#pragma unroll
        for (int b = 0; b < HESS_T; b++) {
            const double t = wave_sum(acc[a][b]);
            if ((threadIdx.x & 63) == 0) sh[a * HESS_T + b][threadIdx.x >> 6] = t;
        }
    __syncthreads();
    if (threadIdx.x < HESS_T * HESS_T)
        A.partials[((int64_t)tile * HESS_T * HESS_T + threadIdx.x) * gridDim.x + blockIdx.x] =
            (sh[threadIdx.x][0] + sh[threadIdx.x][1]) + (sh[threadIdx.x][2] + sh[threadIdx.x][3]);
}

// ---- decaying random-effect columns (nllk_sde.hpp:30-32, 47-58): X_k(i) is scaled by exp(-rho_m t_j(i)), rho_m = exp(log_decay_m) --------
// The predictors stay linear in the coefficients but not in log_decay.  With w_k = X_k exp(-rho t) and the tangent of every unknown
//     coefficient k:  v = w_k e_{j(k)}            log_decay_m:  v_j = e1[j][m] = sum_{k in m, j(k) = j} coef_k w_k (-rho_m t_j)
// the Hessian entry is  sum_i [ v_a' D_i v_b  +  g_i . d2 par / d a d b ],  and the second part is non-zero for
//     (coefficient k in m, log_decay_m):  g[j(k)] w_k (-rho_m t)      (log_decay_m, log_decay_m):  sum_j g[j] e2[j][m],
//     e2[j][m] = sum_k coef_k w_k ((rho t)^2 - rho t).
// An unknown is a slot (uslot >= 0) or a decay rate (uslot = -1 - m).  Same tiles, same fixed-order sums as the kernel above.
template <int MODEL, int D>
__global__ __launch_bounds__(256) void direct_hess_decay_kernel(const DirectHessArgs A) {
    const SlotTable* __restrict__ T = A.slots;
    const int ns = A.n_slots;
    const int tile = blockIdx.y, ti = A.tile_i[tile], tj = A.tile_j[tile];
    double acc[HESS_T][HESS_T];
#pragma unroll
    for (int a = 0; a < HESS_T; a++)
#pragma unroll
        for (int b = 0; b < HESS_T; b++) acc[a][b] = 0.0;
    double rho[MAX_DECAY];
#pragma unroll
    for (int m = 0; m < MAX_DECAY; m++) rho[m] = m < A.n_decay ? exp(A.par[A.off_decay + m]) : 0.0;      // nllk_sde.hpp:47
    // the tile's unknowns (side 0: rows of the tile, side 1: columns)
    int uj[2][HESS_T], um[2][HESS_T];              // SDE parameter of a coefficient; decay rate (of the coefficient's column, or the unknown itself)
    bool on[2][HESS_T], isd[2][HESS_T];            // inside the wanted set; the unknown IS a decay rate
    const double* uc[2][HESS_T];
#pragma unroll
    for (int sd = 0; sd < 2; sd++)
#pragma unroll
        for (int a = 0; a < HESS_T; a++) {
            const int k = (sd == 0 ? ti : tj) * HESS_T + a;
            on[sd][a] = k < A.nu;
            const int us = on[sd][a] ? A.uslot[k] : 0;
            isd[sd][a] = us < 0;
            const int s = us < 0 ? 0 : us;
            uj[sd][a] = T->par_j[s];
            um[sd][a] = us < 0 ? -1 - us : T->decay[s];
            uc[sd][a] = (us >= 0 && T->col[s] >= 0) ? A.cols[T->col[s]] : nullptr;
        }
    __shared__ const double* s_col[MAX_COLS];
    __shared__ double s_coef[MAX_COLS];
    __shared__ int s_pj[MAX_COLS], s_dk[MAX_COLS];
    for (int k = threadIdx.x; k < ns; k += 256) {
        const int c = T->col[k];
        s_col[k] = c >= 0 ? A.cols[c] : nullptr;
        s_coef[k] = A.par[T->pidx[k]];
        s_pj[k] = T->par_j[k];
        s_dk[k] = T->decay[k];
    }
    __syncthreads();
    auto rho_of = [&](int m) { return m == 0 ? rho[0] : m == 1 ? rho[1] : m == 2 ? rho[2] : rho[3]; };
    const int64_t per_block = ((A.n + gridDim.x - 1) / gridDim.x + 255) / 256 * 256;
    const int64_t row_lo = (int64_t)blockIdx.x * per_block;
    const int64_t row_hi = row_lo + per_block < A.n ? row_lo + per_block : A.n;
    for (int64_t i = row_lo + threadIdx.x; i < row_hi; i += 256) {
        if (!((A.scored[i >> 5] >> (i & 31)) & 1u)) continue;
        const double dt = A.times[i] - A.times[i - 1];
        double par[MAX_Q] = {0.0, 0.0, 0.0, 0.0};
        double e1[MAX_Q][MAX_DECAY], e2[MAX_Q][MAX_DECAY];
#pragma unroll
        for (int j = 0; j < MAX_Q; j++)
#pragma unroll
            for (int m = 0; m < MAX_DECAY; m++) e1[j][m] = e2[j][m] = 0.0;
        for (int k = 0; k < ns; k++) {                                             // the linear predictor of row i-1 (Q6) and its log_decay tangents
            const double* cp = s_col[k];
            const int j = s_pj[k], m = s_dk[k];
            double w = cp ? cp[i - 1] : 1.0, rt = 0.0;
            if (cp && m >= 0) { rt = rho_of(m) * A.t_decay[(int64_t)j * A.n + (i - 1)]; w *= exp(-rt); }      // nllk_sde.hpp:47-57
            const double t = w * s_coef[k];
#pragma unroll
            for (int jj = 0; jj < MAX_Q; jj++) {
                par[jj] += j == jj ? t : 0.0;
#pragma unroll
                for (int mm = 0; mm < MAX_DECAY; mm++) {
                    const bool hit = j == jj && m == mm;
                    e1[jj][mm] += hit ? -t * rt : 0.0;
                    e2[jj][mm] += hit ? t * (rt * rt - rt) : 0.0;
                }
            }
        }
        double Dm[MAX_Q][MAX_Q], gm[MAX_Q];
        row_hessian<MODEL, D>(A, i, dt, par, Dm, gm);
        // tangents of the tile's unknowns, and for coefficients their own d w / d log_decay
        double v[2][HESS_T][MAX_Q], dw[2][HESS_T];
#pragma unroll
        for (int sd = 0; sd < 2; sd++)
#pragma unroll
            for (int a = 0; a < HESS_T; a++) {
                const int j = uj[sd][a], m = um[sd][a];
                double w = 0.0, rt = 0.0;
                if (on[sd][a] && !isd[sd][a]) {
                    w = uc[sd][a] ? uc[sd][a][i - 1] : 1.0;
                    if (uc[sd][a] && m >= 0) { rt = rho_of(m) * A.t_decay[(int64_t)j * A.n + (i - 1)]; w *= exp(-rt); }
                }
                dw[sd][a] = -w * rt;
#pragma unroll
                for (int q = 0; q < MAX_Q; q++) {
                    double e = 0.0;
#pragma unroll
                    for (int mm = 0; mm < MAX_DECAY; mm++) e += m == mm ? e1[q][mm] : 0.0;
                    v[sd][a][q] = !on[sd][a] ? 0.0 : isd[sd][a] ? e : (j == q ? w : 0.0);
                }
            }
#pragma unroll
        for (int a = 0; a < HESS_T; a++) {
            if (!on[0][a]) continue;                                               // (uniform)
            double row[MAX_Q];                                                     // v_a' D
#pragma unroll
            for (int q = 0; q < MAX_Q; q++) row[q] = v[0][a][0] * Dm[0][q] + v[0][a][1] * Dm[1][q] + v[0][a][2] * Dm[2][q] + v[0][a][3] * Dm[3][q];
#pragma unroll
            for (int b = 0; b < HESS_T; b++) {
                if (!on[1][b]) continue;
                double t = row[0] * v[1][b][0] + row[1] * v[1][b][1] + row[2] * v[1][b][2] + row[3] * v[1][b][3];
                // the part through d2 par / d a d b
                const bool same = um[0][a] >= 0 && um[0][a] == um[1][b];
                if (same && isd[0][a] != isd[1][b]) {                              // (coefficient of a decaying column, its decay rate)
                    const int j = isd[0][a] ? uj[1][b] : uj[0][a];
                    t += pick4(gm, j) * (isd[0][a] ? dw[1][b] : dw[0][a]);
                } else if (same && isd[0][a] && isd[1][b]) {                       // (log_decay_m, log_decay_m)
#pragma unroll
                    for (int q = 0; q < MAX_Q; q++) {
                        double e = 0.0;
#pragma unroll
                        for (int mm = 0; mm < MAX_DECAY; mm++) e += um[0][a] == mm ? e2[q][mm] : 0.0;
                        t += gm[q] * e;
                    }
                }
                acc[a][b] += t;
            }
        }
    }
    __shared__ double sh[HESS_T * HESS_T][4];
#pragma unroll
    for (int a = 0; a < HESS_T; a++)
#pragma unroll
        for (int b = 0; b < HESS_T; b++) {
            const double t = wave_sum(acc[a][b]);
            if ((threadIdx.x & 63) == 0) sh[a * HESS_T + b][threadIdx.x >> 6] = t;
        }
    __syncthreads();
    if (threadIdx.x < HESS_T * HESS_T)
        A.partials[((int64_t)tile * HESS_T * HESS_T + threadIdx.x) * gridDim.x + blockIdx.x] =
            (sh[threadIdx.x][0] + sh[threadIdx.x][1]) + (sh[threadIdx.x][2] + sh[threadIdx.x][3]);
}

// H[k + l nu] (and its mirror) = sum over blocks of the tile's partials, in block order
__global__ __launch_bounds__(64) void direct_hess_reduce_kernel(const DirectHessArgs A, int n_blocks) {
    const int tile = blockIdx.x, e = blockIdx.y, a = e / HESS_T, b = e % HESS_T;
    const int k = A.tile_i[tile] * HESS_T + a, l = A.tile_j[tile] * HESS_T + b;
    if (k >= A.nu || l >= A.nu) return;
    const double* p = A.partials + ((int64_t)tile * HESS_T * HESS_T + e) * n_blocks;
    double s = 0.0;
    for (int i = threadIdx.x; i < n_blocks; i += 64) s += p[i];
    s = wave_sum(s);
    if (threadIdx.x == 0) { A.hess[k + (int64_t)l * A.nu] = s; A.hess[l + (int64_t)k * A.nu] = s; }
}

hipError_t launch_direct_hess(const DirectHessArgs& a, int n_tiles, int n_blocks, hipStream_t s) {
    dim3 grid(n_blocks, n_tiles), block(256);
    if (a.n_decay > 0) {
        if (a.model == M_BM && a.d == 1) hipLaunchKernelGGL((direct_hess_decay_kernel<M_BM, 1>), grid, block, 0, s, a);
        else if (a.model == M_BM && a.d == 2) hipLaunchKernelGGL((direct_hess_decay_kernel<M_BM, 2>), grid, block, 0, s, a);
        else if (a.model == M_OU && a.d == 1) hipLaunchKernelGGL((direct_hess_decay_kernel<M_OU, 1>), grid, block, 0, s, a);
        else if (a.model == M_OU && a.d == 2) hipLaunchKernelGGL((direct_hess_decay_kernel<M_OU, 2>), grid, block, 0, s, a);
        else if (a.model == M_BM_T && a.d == 1) hipLaunchKernelGGL((direct_hess_decay_kernel<M_BM_T, 1>), grid, block, 0, s, a);
        else if (a.model == M_CIR && a.d == 1) hipLaunchKernelGGL((direct_hess_decay_kernel<M_CIR, 1>), grid, block, 0, s, a);
        else if (a.model == M_CIR && a.d == 2) hipLaunchKernelGGL((direct_hess_decay_kernel<M_CIR, 2>), grid, block, 0, s, a);
        else return hipErrorInvalidValue;
    } else
    if (a.model == M_BM && a.d == 1) hipLaunchKernelGGL((direct_hess_kernel<M_BM, 1>), grid, block, 0, s, a);
    else if (a.model == M_BM && a.d == 2) hipLaunchKernelGGL((direct_hess_kernel<M_BM, 2>), grid, block, 0, s, a);
    else if (a.model == M_OU && a.d == 1) hipLaunchKernelGGL((direct_hess_kernel<M_OU, 1>), grid, block, 0, s, a);
    else if (a.model == M_OU && a.d == 2) hipLaunchKernelGGL((direct_hess_kernel<M_OU, 2>), grid, block, 0, s, a);
    else if (a.model == M_BM_T && a.d == 1) hipLaunchKernelGGL((direct_hess_kernel<M_BM_T, 1>), grid, block, 0, s, a);
    else if (a.model == M_CIR && a.d == 1) hipLaunchKernelGGL((direct_hess_kernel<M_CIR, 1>), grid, block, 0, s, a);
    else if (a.model == M_CIR && a.d == 2) hipLaunchKernelGGL((direct_hess_kernel<M_CIR, 2>), grid, block, 0, s, a);
    else return hipErrorInvalidValue;
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(direct_hess_reduce_kernel, dim3(n_tiles, HESS_T * HESS_T), dim3(64), 0, s, a, n_blocks);
    return hipGetLastError();
}

}  // namespace ssde
