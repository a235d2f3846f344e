// ssde_engine_iso.hip -- one evaluation on the register Kalman path (constant coefficients, or a row-varying drift): the
// window plan, the gain table of the shared-covariance kernels, the launches (shared / general / drift / mixed batch) and
// the finalising launch (hand-over checks + fixed-order sums).  Called from eval_device (ssde_engine.hip).
#include "ssde_engine.hpp"

#include <chrono>

using namespace ssde_engine;

namespace {

// Warm-up length of a time window: iterate the (data-independent) covariance recursion on the
// host at the smallest interval of the batch until it is stationary, take the spectral radius
// rho of the closed-loop matrix T - K Z there, and ask for rho^W <= 1e-18 (plus slack for the
// t * rho^t growth of the sensitivity recursions).  The device-side hand-over check decides
// whether the estimate was good enough; it never has to be trusted.
void plan_windows(ssde_handle* h, const IsoArgs& a, int* n_chunks, int* window) {
    *n_chunks = 1;
    *window = 0;
    h->plan_warmup = 0;
    if (h->max_chunks <= 1) return;
    const double dt = h->uniform_dt ? h->dt_uniform : h->dt_min;
    double rho = 1.0;
    if (dt > 0.0 && std::isfinite(dt)) {
        if (h->model == SSDE_MODEL_CTCRW) {
            CtcrwTrans tr;
            ctcrw_trans(dt, a.tau, a.beta, a.sigma, tr);
            double p11 = a.p0[0], p12 = a.p0[1], p22 = a.p0[2], k1 = 0, k2 = 0;
            for (int it = 0; it < 20000; it++) {
                const double F = p11 + a.h, iF = 1.0 / F;
                const double tp11 = p11 + tr.t12 * p12, tp12 = p12 + tr.t12 * p22, tp21 = tr.e * p12, tp22 = tr.e * p22;
                k1 = tp11 * iF; k2 = tp21 * iF;
                const double n11 = tp11 * (1.0 - k1) + tp12 * tr.t12 + tr.q11, n12 = -tp11 * k2 + tp12 * tr.e + tr.q12,
                             n22 = -tp21 * k2 + tp22 * tr.e + tr.q22;
                const double ch = std::fabs(n11 - p11) + std::fabs(n12 - p12) + std::fabs(n22 - p22);
                p11 = n11; p12 = n12; p22 = n22;
                if (ch <= 1e-15 * (std::fabs(p11) + std::fabs(p22))) break;
            }
            // L = [[1 - k1, t12], [-k2, e]]
            const double trc = (1.0 - k1) + tr.e, det = (1.0 - k1) * tr.e + k2 * tr.t12;
            const double disc = trc * trc - 4.0 * det;
            rho = disc >= 0.0 ? std::max(std::fabs(0.5 * (trc + std::sqrt(disc))), std::fabs(0.5 * (trc - std::sqrt(disc))))
                              : std::sqrt(std::fabs(det));
        } else {
            ScalTrans tr;
            if (h->model == SSDE_MODEL_OU_SSM) ou_trans(dt, a.tau, a.sigma, tr);
            else bm_trans(dt, a.sigma, tr);
            double p = a.p0[0], k = 0;
            for (int it = 0; it < 20000; it++) {
                const double F = p + a.h, tp = tr.t * p;
                k = tp / F;
                const double np_ = tp * (tr.t - k) + tr.q;
                const double ch = std::fabs(np_ - p);
                p = np_;
                if (ch <= 1e-15 * std::fabs(p)) break;
            }
            rho = std::fabs(tr.t - k);
        }
    }
    if (h->drift == 3 && dt > 0.0 && std::isfinite(dt)) {
        // row-varying tau / nu: the slowest-forgetting corner of the ranges the linear predictors can reach on this design
        // (the hand-over check decides whether that was enough, as everywhere)
        rho = 0.0;
        const double dts[2] = {dt, h->uniform_dt ? dt : h->dt_max};
        for (int c = 0; c < 8; c++) {
            const double r = closed_loop_rho(h->model, dts[c & 1], (c & 2) ? h->cv_eta_hi[0] : h->cv_eta_lo[0],
                                             (c & 4) ? h->cv_eta_hi[1] : h->cv_eta_lo[1], a.h, a.p0);
            rho = std::max(rho, std::isfinite(r) ? r : 1.0);
        }
    }
    int W = 0;
    h->plan_rho = rho;
    if (!(rho < 0.9995) || !std::isfinite(rho)) return;  // no usable forgetting: sequential filter
    // The stationary CTCRW lanes run the filter as 1/D(q)^2 recursions (k_iso_shared.hip): with closed-loop poles
    // close to 1 their intermediate signals grow like 1/(1-rho)^2 and cancel in the innovation -- below rho = 0.97
    // that costs < 1e-12 relative; above, the evaluation stays on the sequential direction-form filter
    if (h->use_shared && !h->drift && h->model == SSDE_MODEL_CTCRW && rho > 0.97) return;
    // (+ 16 rows of slack for the t rho^t growth of the forward sensitivity recursions; the reverse sweep carries none -- state and
    //  adjoint forget like rho^t -- and every warm-up row costs it a forward AND a backward row: none there.  10^4 x 10^3 rows, 18
    //  columns: W 48 -> 32, kernel 0.617 -> 0.572 ms, hand-over check 9e-15 -> 6e-14 against the 1e-11 it has to meet)
    W = (int)std::ceil(std::log(1e-18) / std::log(std::max(rho, 1e-300))) + (h->cv_adj ? 0 : 16);
    W = std::max(W, 16);
    if (h->env_window > 0) W = h->env_window;                             // testing: deliberately short overlaps
    if ((int64_t)W * h->window_boost > (int64_t)h->glen_max) return;     // longer than a track: sequential filter
    W *= h->window_boost;
    W = (W + WIN_ALIGN - 1) / WIN_ALIGN * WIN_ALIGN;
    // a window must be long enough to amortise its warm-up
    int glmax = 0;
    {
        // group lengths are sorted descending: the first group is the longest
        glmax = h->glen_max;
    }
    int nc = h->want_chunks;
    if (h->drift == 3 && !h->cv_one_wave() && !h->chunks_forced) { while (nc > 1 && (glmax / nc) < 2 * WIN_ALIGN) nc--; }     // (the cost below decides, not a rule)
    else if (h->cv_adj && !h->chunks_forced) {
        // one wave per (group, window) and per SIMD; a window walks its warm-up, its rows and -- for the backward recursion -- W rows
        // past its end forwards (~0.8 us a row), then its rows and that tail backwards (~1.2 us a row): rounds x that is what the
        // launch takes.  With SIMDs idle (few groups) windows shorter than their warm-up pay: the redundant rows run in parallel.
        nc = std::max(1, std::min(h->max_chunks, glmax / (2 * WIN_ALIGN)));
        const int ng = h->n_groups;                           // (the grid pads the groups to eight: those waves leave at once)
        int best = 1;
        double best_cost = (double)((ng + 1023) / 1024) * 2.0 * glmax;
        for (int c = 2; c <= nc; c++) {
            const double len = (double)glmax / c;
            const double cost = (double)(((int64_t)ng * c + 1023) / 1024) * (0.8 * (len + 2.0 * W) + 1.2 * (len + W));
            if (cost < best_cost) { best_cost = cost; best = c; }
        }
        nc = best;
    }
    else
    while (nc > 1 && (glmax / nc) < 2 * W) nc--;
    if (h->drift == 3 && !h->cv_one_wave() && !h->chunks_forced) {
        // one workgroup per (group, window) and per CU: rounds x (rows of a window + its warm-up) is what the launch takes
        int best = 1;
        double best_cost = (double)((h->n_groups + 255) / 256) * glmax;
        for (int c = 2; c <= nc; c++) {
            const double cost = (double)(((int64_t)h->n_groups * c + 255) / 256) * ((double)glmax / c + W);
            if (cost < best_cost) { best_cost = cost; best = c; }
        }
        nc = best;
    }
    *n_chunks = nc;
    *window = nc > 1 ? W : 0;
    h->plan_warmup = W;                                  // usable warm-up length even when one window is planned
}

// Shared-covariance path: run the covariance half of the filter (ssde_math.hpp) ONCE on the host
// for the regular grid -- it does not depend on the observations -- until it is bitwise
// stationary, upload the gains, and return the data-independent likelihood terms
// (D/2 sum log F and its derivatives, weighted by how many tracks reach each row).
template <int D>
int build_gain_table(ssde_handle* h, IsoArgs& a, int mask, hipStream_t s, double add[4]) {
    const int slot = h->par_next;
    h->par_next = (h->par_next + 1) % PAR_RING;
    // the ring protects the pinned slot of an ASYNCHRONOUS caller's earlier evaluation (ssde_eval_device); a
    // synchronous ssde_eval has read its result back before the next call: no event traffic on that path
    if (h->use_shared && (!h->sync_call || h->par_ev_pending[slot])) { HIPCHK(h, hipEventSynchronize(h->par_ev[slot])); h->par_ev_pending[slot] = false; }
    double* host = h->gain_pinned + (size_t)slot * h->gain_rows_cap * GAIN_ROW;
    double* dev = h->gain_ring.p + (size_t)slot * h->gain_rows_cap * GAIN_ROW;
    const int tmax = h->glen_max;                 // rows 0 .. tmax-1 can be asked for
    // running sums of log F and of its derivatives, row by row (member buffers: no allocation per evaluation)
    std::vector<double>& cum_ld = h->gain_cum[0];
    std::vector<double>* cum_g = &h->gain_cum[1];
    for (int j = 0; j < 1 + NDIRP; j++) { if ((int)h->gain_cum[j].capacity() < tmax) h->gain_cum[j].reserve(tmax); h->gain_cum[j].clear(); }
    int last = 0, stable = 0;
    (void)mask;
    // Stationarity test.  In floating point the recursion ends in a last-bit limit cycle rather than a
    // bitwise fixed point, so "stationary" = every component moved by less than 2e-15 relative for 4 rows
    // in a row; the row reached then is used for all later rows (a 1e-15 relative perturbation of gains
    // that themselves carry rounding errors of that size).
    auto close = [](double a, double b) { return std::fabs(a - b) <= 2e-15 * (std::fabs(a) + std::fabs(b)) + 1e-300; };
    if (h->model == SSDE_MODEL_CTCRW) {
        CtcrwCov<15> C;
        C.init(a.p0[0], a.p0[1], a.p0[2]);
        double ld = 0.0;
        for (int t = 0; t < tmax; t++) {
            const CtcrwCov<15> prev = C;
            CtcrwGain G;
            const double F = C.p11 + a.h;
            ctcrw_cov_step<D, 15>(C, a.ctr, a.h, false, G);
            double* r = host + (size_t)t * GAIN_ROW;
            r[0] = G.iF; r[1] = G.k1; r[2] = G.k2; r[3] = G.bm;
            for (int j = 0; j < NDIRP; j++) { r[4 + j] = G.diF[j]; r[7 + j] = G.dk1[j]; r[10 + j] = G.dk2[j]; }
            r[13] = r[14] = r[15] = 0.0;
            ld += (G.iF != 0.0) ? std::log(std::fabs(F)) : 0.0;
            cum_ld.push_back(ld);
            for (int j = 0; j < NDIRP; j++) cum_g[j].push_back(C.gld[j]);
            last = t;
            bool same = close(C.p11, prev.p11) && close(C.p12, prev.p12) && close(C.p22, prev.p22);
            for (int j = 0; j < NDIRP && same; j++)
                same = close(C.d11[j], prev.d11[j]) && close(C.d12[j], prev.d12[j]) && close(C.d22[j], prev.d22[j]);
            stable = same ? stable + 1 : 0;
            if (stable >= 4) break;
        }
        h->stat_p[0] = C.p11; h->stat_p[1] = C.p12; h->stat_p[2] = C.p22;
        for (int j = 0; j < NDIRP; j++) { h->stat_p[3 + 3 * j] = C.d11[j]; h->stat_p[4 + 3 * j] = C.d12[j]; h->stat_p[5 + 3 * j] = C.d22[j]; }
    } else {
        ScalCov<15> C;
        C.init(a.p0[0]);
        double ld = 0.0;
        for (int t = 0; t < tmax; t++) {
            const ScalCov<15> prev = C;
            ScalGain G;
            const double F = C.p + a.h;
            if (h->model == SSDE_MODEL_OU_SSM) scal_cov_step<D, 15, true>(C, a.str, a.h, false, G);
            else scal_cov_step<D, 15, false>(C, a.str, a.h, false, G);
            double* r = host + (size_t)t * GAIN_ROW;
            for (int k = 0; k < GAIN_ROW; k++) r[k] = 0.0;
            r[0] = G.iF; r[1] = G.k; r[2] = G.c;
            for (int j = 0; j < NDIRP; j++) { r[4 + j] = G.diF[j]; r[7 + j] = G.dk[j]; }
            ld += (G.iF != 0.0) ? std::log(std::fabs(F)) : 0.0;
            cum_ld.push_back(ld);
            for (int j = 0; j < NDIRP; j++) cum_g[j].push_back(C.gld[j]);
            last = t;
            bool same = close(C.p, prev.p);
            for (int j = 0; j < NDIRP && same; j++) same = close(C.dp[j], prev.dp[j]);
            stable = same ? stable + 1 : 0;
            if (stable >= 4) break;
        }
        for (int i = 0; i < 12; i++) h->stat_p[i] = 0.0;
        h->stat_p[0] = C.p;
        for (int j = 0; j < NDIRP; j++) h->stat_p[3 + 3 * j] = C.dp[j];
    }
    const int rows = last + 1;
    h->last_gain_rows = rows;
    // (quiet rows of the general kernel: the covariance after the last row, and what a further row adds to sum log F / sum dF / F)
    h->gain_stationary = stable >= 4 && last >= 1;
    if (last >= 1) {
        h->stat_ld = cum_ld[last] - cum_ld[last - 1];
        for (int j = 0; j < NDIRP; j++) h->stat_gld[j] = cum_g[j][last] - cum_g[j][last - 1];
    }
    if (h->use_shared) {                                    // (otherwise only the stationary constants are wanted)
        HIPCHK(h, hipMemcpyAsync(dev, host, (size_t)rows * GAIN_ROW * 8, hipMemcpyHostToDevice, s));
        if (!h->sync_call) { HIPCHK(h, hipEventRecord(h->par_ev[slot], s)); h->par_ev_pending[slot] = true; }
    }
    a.gain = dev;
    a.gain_last = last;
    for (int k = 0; k < GAIN_ROW; k++) a.gain_stat[k] = host[(size_t)last * GAIN_ROW + k];
    fill_stat_consts(h->model, h->d, a);
    // data-independent terms: a track with ns scored rows contributes cum(ns - 1); past the
    // stationary row every further row adds the same increment
    auto cum_at = [&](const std::vector<double>& c, int idx) {
        if (idx <= last) return c[idx];
        const double inc = last > 0 ? c[last] - c[last - 1] : c[last];
        return c[last] + inc * (double)(idx - last);
    };
    double s_ld = 0.0, s_g[NDIRP] = {0, 0, 0};
    for (auto& e : h->clean_ns_hist) {
        s_ld += (double)e.second * cum_at(cum_ld, e.first - 1);
        for (int j = 0; j < NDIRP; j++) s_g[j] += (double)e.second * cum_at(cum_g[j], e.first - 1);
    }
    add[0] = 0.5 * D * s_ld;
    for (int j = 0; j < NDIRP; j++) add[1 + j] = 0.5 * D * s_g[j];
    return SSDE_OK;
}

}  // namespace

namespace ssde_engine {

int eval_iso(ssde_handle* h, const double* par, int order, double* out_dev, hipStream_t s, ReduceArgs& ra) {
    const ParLayout& L = h->L;
    IsoArgs a;
    memset(&a, 0, sizeof(a));
    a.tv.tiles = h->tiles.p; a.tv.group_off = h->group_off.p; a.tv.group_len = h->group_len.p;
    a.tv.lane_nsteps = h->lane_nsteps.p; a.tv.a0 = h->a0.p; a.tv.n_groups = h->n_groups; a.tv.C = h->C; a.tv.c_obs = h->c_obs; a.tv.dt_all = h->dt_all;
    a.partials = h->partials.p;
    if (order >= 1) {
        a.n_parts = h->iso_parts;
        for (int p = 0; p < MAX_PARTS; p++) a.part_mask[p] = h->iso_masks[p];
    } else {
        a.n_parts = 1;
    }
    if (h->drift == 3) { a.n_parts = h->cv_adj ? 1 : h->cv_few ? h->cv_kc / CV_KC : h->cv_single ? 1 : CV_WAVES; a.part_mask[0] = order >= 1 ? 1 : 0; }    // k_iso_colvar.hip: the parts are the waves of a workgroup
    a.any_nan = h->na_any;
    a.uniform_dt = h->uniform_dt ? 1 : 0;
    const double sig = exp(par[0]);                     // nllk_ctcrw.hpp:136
    a.h = sig * sig;                                    // makeH: sigma_obs * sigma_obs
    for (int i = 0; i < h->d; i++) a.mu[i] = par[L.off_fe + L.fe_off[i]];
    for (int i = 0; i < 3; i++) a.p0[i] = h->p0_iso[i];
    if (h->drift == 3) {
        // p_j(i) = intercept_j + sum_k coef_k X_k(i) for the rows of par[d] and par[d + 1] (nllk_ctcrw.hpp:143-156); and the range
        // each can reach on this design (column ranges found at create), for the window plan
        for (int j = 0; j < 2; j++) h->cv_eta_lo[j] = h->cv_eta_hi[j] = 0.0;
        // (a drift with a fixed-effect design of its own -- mu ~ 1 + x -- has NO intercept slot: its column of ones is a streamed
        // column like the others, and the constant part of mu_a is zero)
        for (int i = 0; i < h->d; i++) a.mu[i] = 0.0;
        for (auto& sl : h->slots) {
            const int j = sl.par_j - h->d;
            if (j < 0) {                                        // a design column of the drift: mu_a(i) = intercept + sum_k coef_k X_k(i)
                if (sl.col >= 0) { (sl.par_j == 0 ? a.coefC : a.coefD)[sl.col] = par[sl.pidx]; a.cv_mu_cols = 1; }
                else a.mu[sl.par_j] = par[sl.pidx];
                continue;
            }
            const double b = par[sl.pidx];
            if (sl.col < 0) { a.cv_eta0[j] = b; h->cv_eta_lo[j] += b; h->cv_eta_hi[j] += b; continue; }
            (j == 0 ? a.coefA : a.coefB)[sl.col] = b;
            const double lo = h->cv_col_lo[sl.col], hi = h->cv_col_hi[sl.col];
            h->cv_eta_lo[j] += std::min(b * lo, b * hi);
            h->cv_eta_hi[j] += std::max(b * lo, b * hi);
        }
        a.drift_k = h->n_stream_cols; a.c_col = h->c_obs + h->d + (h->has_h ? h->d * h->d : 0);
        a.cv_full = h->cv_full ? 1 : 0; a.cv_has_h = h->has_h ? 1 : 0;
        for (int i = 0; i < 16; i++) a.cv_p0[i] = h->p0_full[i];
        if (h->has_h) a.h = h->cv_hmax;                         // (the window planner's observation variance; the lanes read H_array[,,i])
        // ... which is loose (a partition-of-unity basis reaches max |coef|, the bound says sum |coef|): once a launch has run,
        // the range it actually saw, widened by a quarter of its width (+ 0.05), bounds the plan; a parameter jump that leaves
        // it fails the hand-over check and the retry plans from that evaluation's own range
        if (h->cv_ranges_pinned && !getenv("SSDE_CV_DESIGN_BOUND"))
            for (int j = 0; j < 2; j++) {
                const double lo = h->cv_ranges_pinned[2 * j], hi = h->cv_ranges_pinned[2 * j + 1];
                if (!(lo <= hi) || !std::isfinite(lo) || !std::isfinite(hi)) continue;
                const double m = 0.25 * (hi - lo) + 0.05;
                h->cv_eta_lo[j] = std::max(h->cv_eta_lo[j], lo - m);
                h->cv_eta_hi[j] = std::min(h->cv_eta_hi[j], hi + m);
                if (h->cv_eta_lo[j] > h->cv_eta_hi[j]) { h->cv_eta_lo[j] = lo - m; h->cv_eta_hi[j] = hi + m; }
            }
    } else
    if (h->drift) {
        // mu_a(i) = intercept + sum_k coef_k X_k(i) (nllk_ctcrw.hpp:143-149): the intercept slot (if any) goes where the
        // constant-drift kernels keep mu, the streamed columns get their coefficients by the dimension they feed
        for (int i = 0; i < h->d; i++) a.mu[i] = 0.0;
        for (auto& sl : h->slots) {
            if (sl.col < 0) { if (sl.par_j < h->d) a.mu[sl.par_j] = par[sl.pidx]; continue; }
            if (sl.par_j == 0) a.coefA[sl.col] = par[sl.pidx];
            else { a.coefB[sl.col] = par[sl.pidx]; a.drift_dim1 |= 1u << sl.col; }
        }
        a.drift_k = h->n_stream_cols; a.c_col = h->c_obs + h->d;
        a.pp = h->pp_drift;
    }
    const double p1 = par[L.off_fe + L.fe_off[h->d]];
    const double p2 = (h->q > h->d + 1) ? par[L.off_fe + L.fe_off[h->d + 1]] : 0.0;
    if (h->model == SSDE_MODEL_CTCRW) {
        a.tau = exp(p1);                                // :153
        const double nu = exp(p2);                      // :154
        a.beta = 1.0 / a.tau;                           // :155
        a.sigma = 2.0 * nu / sqrt(M_PI * a.tau);        // :156
        if (h->uniform_dt) ctcrw_trans(h->dt_uniform, a.tau, a.beta, a.sigma, a.ctr);
    } else if (h->model == SSDE_MODEL_OU_SSM) {
        a.tau = exp(p1);
        a.sigma = exp(p2);                              // kappa
        if (h->uniform_dt) ou_trans(h->dt_uniform, a.tau, a.sigma, a.str);
    } else {
        a.sigma = exp(p1);
        if (h->uniform_dt) bm_trans(h->dt_uniform, a.sigma, a.str);
    }
    auto tick = [&]() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    double tk0 = h->trace ? tick() : 0.0;
    plan_windows(h, a, &a.n_chunks, &a.window);
    if (h->trace) { const double t = tick(); h->trace_us[0] += t - tk0; tk0 = t; }
    a.bnd = h->bnd.p; a.chk = h->chk.p;
    a.bnd_stride = h->drift ? std::max(NSTATE_MAX, h->drift_nstate) : NSTATE_MAX;
    a.chk_out = out_dev + (1 + L.n_full);
    a.derive = (h->env_no_derive || h->drift) ? 0 : 1;
    a.stream_nt = h->stream_nt ? 1 : 0;
    a.all_clean = ((h->use_shared && h->n_clean_groups == h->n_groups) || h->drift) ? 1 : 0;      // (drift: one dump layout for every group)
    a.nstate_clean = h->drift ? h->drift_nstate
                   : h->use_shared ? shared_nstate(h->sdim, order >= 1 ? a.part_mask[0] : 0, h->model != SSDE_MODEL_BM_SSM) : 0;
    h->last_chunks = a.n_chunks; h->last_window = a.window;
    a.group_flags = h->group_flags.p;
    a.group_mode = 0;
    double add[4] = {0, 0, 0, 0};
    if (!h->use_shared && h->quiet_ok) {
        int st = (h->d == 1) ? build_gain_table<1>(h, a, h->iso_free_mask, s, add) : build_gain_table<2>(h, a, h->iso_free_mask, s, add);
        if (st) return st;
        a.gain = nullptr;
        for (double& v : add) v = 0.0;
    }
    if (h->use_shared) {
        int st = (h->d == 1) ? build_gain_table<1>(h, a, h->iso_free_mask, s, add)
                             : build_gain_table<2>(h, a, h->iso_free_mask, s, add);
        if (st) return st;
        if (h->trace) { const double t = tick(); h->trace_us[1] += t - tk0; tk0 = t; }
        a.group_mode = 3;
        // the covariance transient gets its own short window [0, t0): every other window (warm-up
        // included) then lies in the stationary regime and runs the lean kernel
        // A batch with more track groups than SIMDs needs no time windows to fill the chip, but the lean
        // stationary kernel only exists for windows past the covariance transient: split every track into
        // the transient window and ONE stationary window (same wave, so no extra work items)
        if (h->drift) {
            // every row costs the same here (HBM-bound; the table rows and the stationary rows run the same step): plain equal windows
            a.t0 = 0;
        } else
        if (a.n_chunks == 1 && h->plan_warmup > 0 && h->max_chunks >= 2 && !h->chunks_forced) {
            a.n_chunks = 1; a.window = h->plan_warmup;
            const int s_stat0 = (a.gain_last + SHARED_U - 1) / SHARED_U * SHARED_U;
            const int t0c = (s_stat0 + a.window + WIN_ALIGN - 1) / WIN_ALIGN * WIN_ALIGN;
            if (t0c + 2 * a.window < h->glen_max) { a.t0 = t0c; a.n_chunks = 2; h->last_window = a.window; }
            else a.window = 0;
        } else
        if (a.n_chunks > 1) {
            const int s_stat = (a.gain_last + SHARED_U - 1) / SHARED_U * SHARED_U;
            a.t0 = (s_stat + a.window + WIN_ALIGN - 1) / WIN_ALIGN * WIN_ALIGN;
            if (a.t0 + 2 * a.window >= h->glen_max) { a.t0 = 0; }            // tracks too short to bother
            else if (a.n_chunks < h->max_chunks) a.n_chunks += 1;           // window 0 + the planned ones
        }
        h->last_chunks = a.n_chunks;
    }
    // the transient window (gain table, direction form) runs on the wave that owns window 1: that window is
    // shortened by what the transient rows cost, in stationary rows (SSDE_T0_COST x t0)
    {
        const double cost = h->env_t0_cost;
        a.t0_delta = (int)(cost * a.t0 + WIN_ALIGN - 1) / WIN_ALIGN * WIN_ALIGN;
    }
    if (!h->use_shared && !h->drift && a.n_chunks > 1 && h->env_w0_ratio > 0.0) {
        // General kernel, every window on its own wave: window 0 carries EVERY direction (windows >= 1 derive one,
        // k_iso.hip) but has no warm-up rows.  With equal windows its waves are the last to finish and the whole
        // launch waits for them (CTCRW: 212 against 163 instructions per row).  Balance: window 0 = [0, L0) with
        // r L0 = L1 + W, the others split [L0, L) equally -- the geometry window_bounds already has for a transient
        // window (t0 = L0), with nothing to subtract from window 1 (t0_delta = 0: it has a wave of its own).
        const bool can_derive = order >= 1 && a.derive && (a.part_mask[0] & DIR_SIG) &&
                                (a.part_mask[0] & (h->model == SSDE_MODEL_BM_SSM ? DIR_P1 : DIR_P2));
        const double r = can_derive ? h->env_w0_ratio : 1.0;
        const int nc = a.n_chunks;
        const double L0 = ((double)h->glen_max / (nc - 1) + a.window) / (r + 1.0 / (nc - 1));
        const int t0 = (int)(L0 / WIN_ALIGN) * WIN_ALIGN;
        if (t0 >= 2 * WIN_ALIGN && t0 + 2 * a.window < h->glen_max) { a.t0 = t0; a.t0_delta = 0; }
    }
    h->last_t0 = a.t0; h->last_t0_delta = a.t0_delta;
    h->last_quiet_window = 0;
    // (CTCRW: the transfer-function lanes lose digits when the closed-loop poles approach 1 -- the limit plan_windows has for them)
    // (a plan that has given up on windows gives up on quiet rows too: with them the retry would run the identical plan again)
    if (h->quiet_ok && !h->gave_up && h->gain_stationary && h->plan_warmup > 0 && a.gain_stat[0] != 0.0 && a.n_parts == 1 &&
        !(h->model == SSDE_MODEL_CTCRW && h->plan_rho > 0.97)) {
        const int U = iso_block_rows(h->model);
        h->last_quiet_window = h->plan_warmup;
        a.nan_bits = h->nan_bits.p; a.nan_words = h->nan_words; a.quiet_flag = h->quiet_flag.p;
        a.quiet_w = (h->plan_warmup + U - 1) / U;
        // (testing: a memory of its own, deliberately short -- the check at the switch has to notice; a retry doubles it like a warm-up)
        if (h->env_quiet_window > 0) { a.quiet_w = (h->env_quiet_window * h->window_boost + U - 1) / U; h->last_quiet_window = h->env_quiet_window * h->window_boost; }
        a.quiet_b0 = (a.gain_last + SHARED_U - 1) / SHARED_U * SHARED_U / U + 1;
        for (int i = 0; i < 12; i++) a.quiet_p[i] = h->stat_p[i];
        a.quiet_ld = h->stat_ld;
        for (int j = 0; j < NDIRP; j++) a.quiet_gld[j] = h->stat_gld[j];
    }
    // The finalising work inside the main launch (fused_finalize_wave, ssde_device.hpp): the shared-covariance kernel alone on the
    // batch (no group on the general kernel, no drift columns); SSDE_FUSED_FINALIZE=0: the two-launch form (A/B -- bitwise the same)
    const bool fused = h->use_shared && !h->drift && h->n_clean_groups == h->n_groups && !h->hess_req && h->fuse_words.p && !h->env_no_fused;
    // the reduction's arguments: which accumulator of which part feeds which output slot (needed BEFORE the main launch when the
    // finalising work is fused into it)
    auto fill_ra = [&]() {
        for (int i = 0; i < 4; i++) { ra.add[i] = add[i]; ra.add_slot[i] = -1; }
        if (h->use_shared) {
            ra.add_slot[0] = 0;
            if (order >= 1) {
                const int pj[NDIRP] = {0, L.off_fe + L.fe_off[h->d], h->q > h->d + 1 ? L.off_fe + L.fe_off[h->d + 1] : 0};
                for (int j = 0; j < NDIRP; j++)
                    if (pj[j] < L.n_full && !h->fixed[pj[j]] && (j < 2 || h->q > h->d + 1)) ra.add_slot[1 + j] = (int16_t)(1 + pj[j]);
            }
        }
        const int nacc = h->cv_adj ? adj_nacc(h->model, h->d, h->n_stream_cols, a.cv_mu_cols != 0)
                       : h->drift == 3 ? 2 + CV_KC + h->d : 4 + h->d + (h->drift ? h->n_stream_cols : 0);
        const int ncr = (a.dual && a.n_chunks_d > a.n_chunks) ? a.n_chunks_d : a.n_chunks;     // windows the final sums run over
        ra.n_parts = a.n_parts * ncr; ra.nacc = nacc; ra.n_blocks = h->n_groups;
        ra.n_value_parts = ncr; ra.chunks_per_part = ncr;
        ra.chk = h->chk.p; ra.n_chk = a.n_chunks > 1 ? a.n_parts * (a.n_chunks - 1) * h->n_groups : 0;
        if (order >= 1 && h->cv_adj) {
            // accumulators of k_iso_adj.hip: [value | log sigma_obs | mu_a | par[d] | par[d + 1] | per streamed column: par[d], par[d + 1] (, mu_a)]
            const int nkp = h->model == SSDE_MODEL_BM_SSM ? 1 : 2, nk = adj_nk(h->model, h->d, a.cv_mu_cols != 0);
            if (!h->fixed[0] && !h->has_h) ra.map[0] = 1;
            for (auto& sl : h->slots) {
                if (h->fixed[sl.pidx]) continue;
                const int kind = sl.par_j < h->d ? nkp + sl.par_j : sl.par_j - h->d;
                const int k = sl.col >= 0 ? 4 + h->d + sl.col * nk + kind : (sl.par_j < h->d ? 2 + sl.par_j : 2 + h->d + (sl.par_j - h->d));
                ra.map[k - 1] = (int16_t)(1 + sl.pidx);
            }
        } else
        if (order >= 1 && h->drift == 3) {
            // accumulators of k_iso_colvar.hip, per part: [value | the part's columns | mu_1 .. mu_d | log sigma_obs]
            for (int p = 0; p < CV_WAVES; p++) {
                for (int k = 0; k < CV_KC; k++) {
                    const int pidx = h->cv_pidx[(size_t)p * CV_KC + k];
                    if (pidx >= 0) ra.map[p * (nacc - 1) + k] = (int16_t)(1 + pidx);
                }
                if (p == h->cv_mu_part)
                    for (auto& sl : h->slots)
                        if (sl.par_j < h->d && sl.col < 0 && !h->fixed[sl.pidx]) ra.map[p * (nacc - 1) + CV_KC + sl.par_j] = (int16_t)(1 + sl.pidx);
                if (p == h->cv_sig_part) ra.map[p * (nacc - 1) + CV_KC + h->d] = 1;
            }
        } else
        if (order >= 1 && h->drift) {
            // accumulators of k_iso_drift.hip: [value | sigma_obs | mu intercepts | par d | par d+1 | streamed columns]
            if (!h->fixed[0]) ra.map[0] = 1;
            for (auto& sl : h->slots) {
                if (h->fixed[sl.pidx]) continue;
                const int k = sl.col >= 0 ? 4 + h->d + sl.col : (sl.par_j < h->d ? 2 + sl.par_j : sl.par_j == h->d ? 2 + h->d : 3 + h->d);
                ra.map[k - 1] = (int16_t)(1 + sl.pidx);
            }
        } else
        if (order >= 1) {
            for (int p = 0; p < a.n_parts; p++)
                for (int k = 1; k < nacc; k++) {
                    // accumulators are ordered like the constant-coefficient parameter vector: sigma_obs, one per SDE parameter
                    const int j = k - 2;     // SDE parameter of accumulator k (k == 1: log_sigma_obs)
                    if (j >= h->q) continue;
                    const int pidx = j < 0 ? 0 : L.off_fe + L.fe_off[j];
                    if (pidx < L.n_full && !h->fixed[pidx]) ra.map[p * (nacc - 1) + (k - 1)] = (int16_t)(1 + pidx);
                }
        }
    };
    if (h->use_shared) {
        // two independent launches (NaN-free groups on the shared-covariance kernel, NaN-carrying groups on
        // the general kernel): fork onto a side stream so they share the chip, join before the hand-over check
        const bool any_dirty = h->n_clean_groups < h->n_groups;
        IsoArgs ad = a;                      // the general launch: this plan, or -- mixed batch -- one of its own
        if (any_dirty && h->want_chunks_d > 0 && a.n_chunks > 1 && h->max_chunks > 1 && !h->gave_up) {
            int nc = h->want_chunks_d;
            while (nc > 1 && (h->glen_max / nc) < 2 * a.window) nc--;
            if (nc > 1) {
                ad.n_chunks = nc; ad.t0 = 0; ad.t0_delta = 0;
                // window 0 carries every direction and has no warm-up: the balance of the all-general case (below)
                const bool can_derive = order >= 1 && a.derive && (a.part_mask[0] & DIR_SIG) &&
                                        (a.part_mask[0] & (h->model == SSDE_MODEL_BM_SSM ? DIR_P1 : DIR_P2));
                const double r = (can_derive && h->env_w0_ratio > 0.0) ? h->env_w0_ratio : 1.0;
                const double L0 = ((double)h->glen_max / (nc - 1) + a.window) / (r + 1.0 / (nc - 1));
                const int t0 = (int)(L0 / WIN_ALIGN) * WIN_ALIGN;
                if (t0 >= 2 * WIN_ALIGN && t0 + 2 * a.window < h->glen_max) ad.t0 = t0;
                a.dual = 1; a.n_chunks_d = ad.n_chunks; a.window_d = ad.window; a.t0_d = ad.t0; a.t0_delta_d = ad.t0_delta;
                a.dirty_groups = h->dirty_groups.p; a.n_dirty_groups = h->n_dirty_groups;
                ad.dirty_groups = h->dirty_groups.p; ad.n_dirty_groups = h->n_dirty_groups; ad.use_group_list = 1;
                // the final sums run over the longer of the two plans: the slots the shorter one does not write must be zero
                HIPCHK(h, hipMemsetAsync(h->partials.p, 0, (size_t)std::max(a.n_chunks, ad.n_chunks) * (4 + h->d) * h->n_groups * 8, s));
            }
        }
        IsoArgs b = a;
        b.group_mode = 2;
        if (!h->wave_clock_file.empty()) {
            const int items = ((h->n_groups + 7) / 8 * 8) * a.n_chunks + 8;
            if ((int)h->wave_clock.n < 4 * items) { h->wave_clock.release(); HIPCHK(h, h->wave_clock.alloc((size_t)4 * items)); }
            HIPCHK(h, hipMemsetAsync(h->wave_clock.p, 0, (size_t)4 * items * 8, s));
            b.wave_clock = h->wave_clock.p; h->wave_clock_items = items;
        }
        if (any_dirty) {
            HIPCHK(h, hipEventRecord(h->ev_fork, s));
            HIPCHK(h, hipStreamWaitEvent(h->aux[1], h->ev_fork, 0));
            HIPCHK(h, launch_iso(h->model, h->d, ad, true, h->aux[1]));
            HIPCHK(h, hipEventRecord(h->ev_join[1], h->aux[1]));
        }
        if (h->drift && h->hess_req) {
            // ssde_hess on a drift handle: the same plan, the same gains, the Hessian kernels instead of the evaluation
            h->hess_req = false;
            DriftHessArgs hx = h->hess_args;
            HIPCHK(h, launch_iso_drift_hess(h->model, b, hx, h->hess_tiles, s));
            return SSDE_OK;
        }
        if (h->drift) HIPCHK(h, launch_iso_drift(h->model, h->d, b, s, h->stamps ? h->ev_k0 : nullptr, h->stamps ? h->ev_k1 : nullptr));
        else {
            if (fused) {
                fill_ra();
                ra.kfast = 1;
                b.fused = 1;
                b.fuse_arrive = h->fuse_words.p + 4; b.fuse_done = h->fuse_words.p;
                b.chk_out = (double*)(h->fuse_words.p + 2);         // (its own word, zero between launches: the last wave moves it to out[n_out])
            }
            HIPCHK(h, launch_iso_shared(h->model, h->d, b, ra, s, h->stamps ? h->ev_k0 : nullptr, h->stamps ? h->ev_k1 : nullptr));
        }
        h->last_kernel_id = h->drift ? SSDE_KERNEL_ISO_DRIFT : any_dirty ? SSDE_KERNEL_ISO_MIXED : SSDE_KERNEL_ISO_SHARED;
        h->ev_k_valid = h->stamps;
        h->last_s_stat = h->drift ? -1 : (a.gain_last + SHARED_U - 1) / SHARED_U * SHARED_U;
        if (any_dirty) HIPCHK(h, hipStreamWaitEvent(s, h->ev_join[1], 0));
    } else {
        if (h->stamps) HIPCHK(h, hipEventRecord(h->ev_k0, s));
        if (h->drift == 3) {
            a.t0 = 0; a.t0_delta = 0;
            if (!h->wave_clock_file.empty()) {                  // (a build with -DSSDE_CV_CLOCK fills it: cycles per row and phase of every wave)
                const int items = h->n_groups * a.n_chunks * CV_WAVES;
                if ((int)h->wave_clock.n < 4 * items) { h->wave_clock.release(); HIPCHK(h, h->wave_clock.alloc((size_t)4 * items)); }
                HIPCHK(h, hipMemsetAsync(h->wave_clock.p, 0, (size_t)4 * items * 8, s));
                a.wave_clock = h->wave_clock.p; h->wave_clock_items = items;
            }
            h->last_kernel_id = h->cv_adj ? SSDE_KERNEL_ISO_ADJ : h->cv_single ? SSDE_KERNEL_ISO_FULL : h->cv_few ? SSDE_KERNEL_ISO_FEW : SSDE_KERNEL_ISO_COLVAR;
            if (h->cv_adj) {
                // checkpoints: the state entering every CB-th row from a window's first scored row to where its backward recursion starts
                const int items = adj_items(h->n_groups, a.n_chunks), cb = adj_ckpt_rows(h->model, h->d, h->cv_full), nst = adj_nstate(h->model, h->d, h->cv_full) / 2;
                const int units = (h->glen_max + WIN_ALIGN - 1) / WIN_ALIGN;
                a.adj_tail = a.window;
                if (const char* e = getenv("SSDE_ADJ_DIAG")) a.adj_diag = atoi(e);
                if (h->env_adj_tail > 0) a.adj_tail = (int)std::min<int64_t>((int64_t)h->env_adj_tail * h->window_boost, h->glen_max);      // (testing)
                const int len = (units / a.n_chunks + 1) * WIN_ALIGN + (a.n_chunks > 1 ? a.adj_tail : 0);
                a.adj_ckpt_stride = (int64_t)((len + cb - 1) / cb + 1) * nst * WAVE;
                const size_t need = order >= 1 ? (size_t)items * (size_t)a.adj_ckpt_stride : (size_t)WAVE;
                if (h->adj_ckpt.n < need) { h->adj_ckpt.release(); HIPCHK(h, h->adj_ckpt.alloc(need)); }
                a.adj_ckpt = h->adj_ckpt.p;
                if ((int)h->cv_ranges.n < 4 * items) { h->cv_ranges.release(); HIPCHK(h, h->cv_ranges.alloc((size_t)4 * items)); }
                a.cv_ranges = h->cv_ranges.p;
                HIPCHK(h, launch_iso_adj(h->model, h->d, a, s));
                HIPCHK(h, launch_colvar_range_reduce(h->cv_ranges.p, items, h->cv_ranges_pinned, s));
            } else
            if (h->cv_single) HIPCHK(h, launch_iso_full(h->model, a, h->cv_parts.p, s));
            else if (h->cv_few) HIPCHK(h, launch_iso_few(h->model, h->d, a, h->cv_parts.p, h->cv_kc, s));
            else {
            const int n_wg = h->n_groups * a.n_chunks;
            if ((int)h->cv_ranges.n < 4 * n_wg) { h->cv_ranges.release(); HIPCHK(h, h->cv_ranges.alloc((size_t)4 * n_wg)); }
            a.cv_ranges = h->cv_ranges.p;
            HIPCHK(h, launch_iso_colvar(h->model, h->d, a, h->cv_parts.p, h->cv_kc, s));
            HIPCHK(h, launch_colvar_range_reduce(h->cv_ranges.p, n_wg, h->cv_ranges_pinned, s));
            }
        }
        else if (h->drift) { a.t0 = 0; a.t0_delta = 0; HIPCHK(h, launch_iso_drift_general(h->model, h->d, a, s)); h->last_kernel_id = SSDE_KERNEL_ISO_DRIFT_GEN; }
        else {
            HIPCHK(h, launch_iso(h->model, h->d, a, false, s));
            h->last_kernel_id = a.n_parts > 1 ? SSDE_KERNEL_ISO_SPLIT : a.quiet_w > 0 ? SSDE_KERNEL_ISO_QUIET
                              : a.uniform_dt ? SSDE_KERNEL_ISO_MASK_UNI : SSDE_KERNEL_ISO_MASK;
        }
        if (h->stamps) HIPCHK(h, hipEventRecord(h->ev_k1, s));
        h->ev_k_valid = h->stamps;
        h->last_s_stat = -1;
    }
    if (h->trace) { const double t = tick(); h->trace_us[2] += t - tk0; tk0 = t; }
    if (!fused) fill_ra();
    // the hand-over checks and the final sums in one launch (unless the main launch has done them)
    if (!fused) HIPCHK(h, launch_iso_finalize(h->model, h->d, a, ra, s));
    h->last_fused = fused;
    if (h->trace) {
        const double t = tick(); h->trace_us[3] += t - tk0; h->trace_n++;
        if (h->trace_skip < 8) {                        // the first calls load code objects: not what is being measured
            h->trace_skip++;
            for (double& v : h->trace_us) v = 0.0;
            h->trace_n = 0;
        }
    }
    return SSDE_OK;
}

}  // namespace ssde_engine
