// ssde_engine_build.hip -- ssde_create's work: descriptor checks, segment discovery, the choice of path and layout (direct
// long format; Kalman tiles, lattice padding, clean-first dealing, drift columns), one-off upload + re-tiling, and the window /
// buffer plan of the register path.  Nothing here runs per evaluation.
#include "ssde_engine.hpp"

#include <chrono>
#include <limits>

using namespace ssde_engine;

namespace {

int choose_iso_split(ssde_handle* h) {
    // Which gradient directions are wanted at all
    int m = 0;
    const ParLayout& L = h->L;
    if (!h->fixed[0]) m |= DIR_SIG;
    for (int a = 0; a < h->d; a++)
        if (!h->fixed[L.off_fe + L.fe_off[a]]) m |= DIR_MU;         // (fe_off[j] == j here: constant coefficients; a dimension
    if (!h->fixed[L.off_fe + L.fe_off[h->d]]) m |= DIR_P1;          //  part indexes the whole problem's vector, child_layout)
    if (h->q > h->d + 1 && !h->fixed[L.off_fe + L.fe_off[h->d + 1]]) m |= DIR_P2;
    h->iso_free_mask = m;
    // Direction split: a 10^4-track batch is only ~160 waves for 1024 SIMDs; give every
    // covariance-affecting direction its own wave (each recomputes the cheap primal) until
    // the grid holds a few waves per SIMD.  SSDE_ISO_SPLIT=fused|split overrides.
    const char* env = getenv("SSDE_ISO_SPLIT");
    bool split = false;  // time windows (below) fill the chip without recomputing the primal
    if (env && !strcmp(env, "fused")) split = false;
    if (env && !strcmp(env, "split")) split = true;
    int np = 0;
    if (split) {
        // mu rides with the cheapest covariance direction (sigma_obs), else alone
        int first = (m & DIR_SIG) | (m & DIR_MU);
        if (first) h->iso_masks[np++] = first;
        if (m & DIR_P1) h->iso_masks[np++] = DIR_P1;
        if (m & DIR_P2) h->iso_masks[np++] = DIR_P2;
    }
    if (np == 0) { h->iso_masks[0] = m; np = 1; }
    if (env && strchr(env, ',')) {  // explicit masks, e.g. "3,4,8"
        np = 0;
        int covered = 0;
        for (const char* p = env; *p && np < MAX_PARTS;) {
            int v = atoi(p) & m;
            h->iso_masks[np++] = v;
            covered |= v;
            p = strchr(p, ',');
            if (!p) break;
            p++;
        }
        if (covered != m) { h->iso_masks[0] |= (m & ~covered); }
    }
    h->iso_parts = np;
    return 0;
}


}  // namespace

namespace {

// Is the time grid a lattice -- every scored interval a small whole multiple of the smallest one -- and is laying the
// tracks out on it cheap enough?  If so: pad_pos / n_pad in the handle, the lattice's segment starts, time stamps and
// observation columns (NA_real_ where the data have no row) in device buffers.  Otherwise h->n_pad stays 0.
constexpr int LATTICE_MAX_MULT = 16;           // longest run of absent fixes that is still padded
constexpr double LATTICE_MAX_GROWTH = 1.35;    // break-even of the two general kernels is ~1.45 lattice rows per data row
int lattice_pad(const ssde_desc* d, ssde_handle* h, const std::vector<int64_t>& starts, bool on_dev,
                std::vector<int64_t>& starts_pad, DevBuf<double>& times_p, DevBuf<double>& obs_p) {
    const int64_t n = d->n, n_seg = (int64_t)starts.size() - 1;
    // Everything below runs on the device (k_lattice.hip): what a short fit has to amortise is this function's time.
    DevBuf<double> s_times, s_obs, s_id, mm;
    DevBuf<int64_t> inc, pos, rep, sidx, spos;
    DevBuf<int> bad;
    auto cleanup = [&]() { s_times.release(); s_obs.release(); s_id.release(); mm.release(); inc.release(); pos.release(); rep.release();
                           sidx.release(); spos.release(); bad.release(); };
#define LP_CHK(call) do { hipError_t e__ = (call); if (e__ != hipSuccess) { cleanup(); h->err = std::string(#call) + ": " + hipGetErrorString(e__); return SSDE_ERR_HIP; } } while (0)
    const double *p_times = d->times, *p_obs = d->obs, *p_id = d->id;
    if (!on_dev) {
        LP_CHK(stage(d->times, (size_t)n, false, s_times)); p_times = s_times.p;
        LP_CHK(stage(d->id, (size_t)n, false, s_id)); p_id = s_id.p;
    }
    // (a track's FIRST interval is never used -- a0 is the prediction for the second row as it stands, nllk_ctcrw.hpp:195-200,
    //  SURVEY Q1 -- so it is neither part of the lattice test nor padded: a lattice row there would be a prediction step
    //  the reference does not take)
    const int nb = 1024;
    LP_CHK(mm.alloc((size_t)nb * 2));
    LP_CHK(launch_used_dt_minmax(p_id, p_times, n, mm.p, nb, 0));
    std::vector<double> mmh((size_t)nb * 2);
    LP_CHK(hipMemcpy(mmh.data(), mm.p, mmh.size() * 8, hipMemcpyDeviceToHost));
    double delta = INFINITY, dmax = -INFINITY;
    for (int b = 0; b < nb; b++) { delta = std::min(delta, mmh[2 * b]); dmax = std::max(dmax, mmh[2 * b + 1]); }
    // How far an interval may be from a whole multiple of the step and still count as one.  Time stamps with a decimal step
    // (0.1, 1/24 ...) are regular only to the last bits of the stamps; taking such a grid as exactly regular moves every dt
    // by at most this relative amount, and the nllk by no more (each row's term has an O(1) log-derivative in dt): the
    // default leaves two orders of magnitude to the 1e-10 parity bar.  Long series with inexact steps (stamp / step > ~5000)
    // are beyond it and keep the per-row transition; SSDE_GRID_RTOL loosens it at the caller's own risk.
    double rtol = 1e-12;
    if (const char* e = getenv("SSDE_GRID_RTOL")) rtol = std::max(0.0, atof(e));
    if (!std::isfinite(delta) || !std::isfinite(dmax) || !(delta > 0.0)) { cleanup(); return SSDE_OK; }   // no used interval, or not a grid
    if (dmax == delta) { h->snap_dt = delta; cleanup(); return SSDE_OK; }           // regular (whatever the unused first intervals are)
    if (dmax <= delta * (1.0 + rtol)) { h->snap_dt = 0.5 * (delta + dmax); cleanup(); return SSDE_OK; }   // regular to the last bits
    if (dmax > (LATTICE_MAX_MULT + 0.5) * delta) { cleanup(); return SSDE_OK; }     // too wide
    LP_CHK(inc.alloc((size_t)n)); LP_CHK(pos.alloc((size_t)n)); LP_CHK(rep.alloc((size_t)n)); LP_CHK(bad.alloc(1));
    int64_t np = 0;
    int bad_h = 0;
    LP_CHK(lattice_positions(p_id, p_times, n, delta, rtol, inc.p, pos.p, rep.p, bad.p, &np, &bad_h, 0));
    if (bad_h || np <= n || (double)np > LATTICE_MAX_GROWTH * (double)n + 64.0) { cleanup(); return SSDE_OK; }   // not a (cheap) lattice
    // the lattice's segment starts
    LP_CHK(sidx.upload(std::vector<int64_t>(starts.begin(), starts.begin() + n_seg)));
    LP_CHK(spos.alloc((size_t)n_seg));
    LP_CHK(launch_gather_i64(pos.p, sidx.p, n_seg, spos.p, 0));
    starts_pad.assign((size_t)n_seg + 1, np);
    LP_CHK(hipMemcpy(starts_pad.data(), spos.p, (size_t)n_seg * 8, hipMemcpyDeviceToHost));
    // the lattice's time stamps and observation columns
    if (!on_dev) { LP_CHK(stage(d->obs, (size_t)n * d->n_dim, false, s_obs)); p_obs = s_obs.p; }
    LP_CHK(times_p.alloc((size_t)np));
    LP_CHK(obs_p.alloc((size_t)np * d->n_dim));
    LP_CHK(launch_lattice_scatter(pos.p, p_id, p_times, p_obs, n, d->n_dim, np, delta, times_p.p, obs_p.p, 0));
    LP_CHK(hipDeviceSynchronize());
    // REPORT(aest_all): row i of the reference holds the state AFTER row i's step, i.e. predicted to the time of row i + 1
    // (nllk_ctcrw.hpp:246) -- on the lattice that is the row just before row i + 1's (lattice_maps_kernel)
    h->pad_pos.p = rep.p; h->pad_pos.n = rep.n; rep.p = nullptr; rep.n = 0;
    h->n_pad = np; h->pad_step = delta;
    cleanup();
#undef LP_CHK
    return SSDE_OK;
}

}  // namespace

namespace ssde_engine {
constexpr int SSDE_RETRY_WITHOUT_DRIFT = -77;     // internal: the drift layout was tried and the data do not qualify
constexpr int SSDE_RETRY_WITHOUT_PP = -78;        // internal: the drift's blocks were tiled as covariates and the lanes that would read them are the general ones
// ---- step 5 (register path): which lanes run it (shared covariance / own covariance, with or without drift columns), how many
// time windows, and the buffers of the hand-over check.  gflags[g] != 0: group g has no missing row; glen: padded steps per group ----
static int plan_register_path(ssde_handle* h, int G, const std::vector<int32_t>& gflags, const std::vector<int32_t>& lane_ns,
                              const std::vector<int32_t>& glen) {
    if (h->drift == 3) {
        // row-varying tau / nu: deal the design columns whose coefficients are free (an intercept is a column of ones) to the four
        // waves of a workgroup (k_iso_colvar.hip)
        struct Col { int type, chan, pidx; };
        std::vector<Col> cols;
        const int c_col = h->c_obs + h->d + (h->has_h ? h->d * h->d : 0);
        if (h->C > CV_CMAX) return SSDE_RETRY_WITHOUT_DRIFT;
        if (h->cv_full)                                            // the drift intercepts are columns of ones of kinds 3, 4 on those lanes
            for (auto& sl : h->slots)
                if (sl.par_j < h->d && sl.col < 0 && !h->fixed[sl.pidx]) cols.push_back({3 + sl.par_j, -1, sl.pidx});
        for (auto& sl : h->slots)                                  // the drift's design columns (a mixed design, or a smooth drift with H_array)
            if (sl.par_j < h->d && sl.col >= 0 && !h->fixed[sl.pidx]) cols.push_back({3 + sl.par_j, c_col + sl.col, sl.pidx});
        for (int type = 1; type <= 2; type++)
            for (auto& sl : h->slots)
                if (sl.par_j == h->d + type - 1 && !h->fixed[sl.pidx]) cols.push_back({type, sl.col >= 0 ? c_col + sl.col : -1, sl.pidx});
        {
            // few columns and few tangents besides the directions the filter carries: one wave per (group, window) does it all
            bool mu_streamed = false;
            for (auto& c : cols) mu_streamed = mu_streamed || c.type >= 3;
            // (a drift column whose coefficient is held FIXED is no tangent, but the drift still varies from row to row: iso_few_kernel
            //  has no such lanes -- found by fuzz seed 552, round 5: launch_iso_few refused the launch)
            for (auto& sl : h->slots) mu_streamed = mu_streamed || (sl.par_j < h->d && sl.col >= 0);
            h->cv_few = !h->cv_full && !h->has_h && !mu_streamed && h->n_stream_cols >= 1 && h->n_stream_cols <= 2 * CV_FEW_K &&
                        (int)cols.size() <= 2 * CV_KC && !getenv("SSDE_CV_NO_FEW");
            // (the wide instantiation -- more than four columns or tangents -- of CTCRW with two response columns spills: the pipeline)
            const bool wide = h->n_stream_cols > CV_FEW_K || (int)cols.size() > CV_KC;
            if (wide && h->model == SSDE_MODEL_CTCRW && h->d == 2 && !getenv("SSDE_CV_FEW_WIDE")) h->cv_few = false;
        }
        {
            // the gradient by a reverse sweep (k_iso_adj.hip): its cost does not depend on the number of columns, so it takes every batch
            // the eight-wave pipeline would; iso_few_kernel keeps the handful-of-columns shapes it was built for unless told otherwise
            // (SSDE_CV_ADJ=2: those too; SSDE_CV_ADJ=0: forward tangents everywhere, the A/B)
            bool mu_streamed = false;
            for (auto& c : cols) mu_streamed = mu_streamed || (c.type >= 3 && c.chan >= 0);
            for (auto& sl : h->slots) mu_streamed = mu_streamed || (sl.par_j < h->d && sl.col >= 0);       // (free or fixed: the lanes are the MU ones either way)
            const char* e = getenv("SSDE_CV_ADJ");
            const int mode = e ? atoi(e) : 1;
            h->cv_adj = mode > 0 && !h->cv_single && (!h->has_h || h->d == 1 || h->cv_full) && (!h->cv_full || h->d == 2) && h->n_stream_cols >= 1 &&
                        adj_ks(h->n_stream_cols) >= 0 && (!mu_streamed || adj_ks(h->n_stream_cols) <= 9) && (!h->cv_few || mode > 1);
            if (h->cv_adj) h->cv_few = false;
        }
        if (!h->cv_adj && (int)cols.size() > (h->cv_full ? CV_WAVES - 2 : CV_WAVES) * CV_KC) return SSDE_RETRY_WITHOUT_DRIFT;     // (full-covariance lanes: the two stage waves carry no columns)
        if (h->cv_adj) cols.clear();                               // (no tangents to deal: every direction comes out of the one sweep)
        std::vector<CvPart> parts(CV_WAVES);
        memset(parts.data(), 0, sizeof(CvPart) * CV_WAVES);
        h->cv_pidx.assign((size_t)CV_WAVES * CV_KC, -1);
        // The two stage waves (wave 0: filter, wave 7: transition) are the row's critical path -- long dependent chains -- and
        // get columns only when the six waves between them are full; those take the columns round robin.
        const int N = (int)cols.size();
        int n_w[CV_WAVES];
        for (int p = 0; p < CV_WAVES; p++) n_w[p] = 0;
        bool dealt_by_env = false;
        if (const char* e = getenv("SSDE_CV_DEAL")) {              // testing: "n0,n1,...,n7"
            int v[8], tot = 0;
            if (sscanf(e, "%d,%d,%d,%d,%d,%d,%d,%d", &v[0], &v[1], &v[2], &v[3], &v[4], &v[5], &v[6], &v[7]) == 8) {
                for (int p = 0; p < 8; p++) { v[p] = std::min(std::max(v[p], 0), CV_KC); tot += v[p]; }
                if (tot == N) { for (int p = 0; p < 8; p++) n_w[p] = v[p]; dealt_by_env = true; }
            }
        }
        if (h->cv_few) { n_w[0] = std::min(N, CV_KC); n_w[1] = N - n_w[0]; }     // (iso_few_kernel: every column on the one wave; slots 0-3 / 4-7 = parts 0 / 1)
        else
        if (h->cv_single) {
            // one wave does it all (iso_full_kernel): slots 0, 1 = the log tau, log nu intercepts, slots 2, 3 = the drift intercepts
            static_assert(CV_KC >= 4, "four slots");
            parts[0].n_col = 4;
            for (int k = 0; k < 4; k++) { parts[0].chan[k] = -1; parts[0].type[k] = 0; }
            for (auto& c : cols) {
                if (c.chan >= 0 || c.type < 1 || c.type > 4) return SSDE_RETRY_WITHOUT_DRIFT;
                const int k = c.type <= 2 ? c.type - 1 : c.type - 1;       // kinds 1, 2, 3, 4 -> slots 0, 1, 2, 3
                parts[0].type[k] = c.type;
                h->cv_pidx[k] = c.pidx;
            }
            cols.clear();                                          // (dealt)
        } else
        if (!dealt_by_env) {
            int left = N;
            for (int k = 0; k < CV_KC && left > 0; k++)
                for (int p = 1; p < CV_WAVES - 1 && left > 0; p++) { n_w[p]++; left--; }
            for (int k = 0; k < CV_KC && left > 0; k++)
                for (int p : {CV_WAVES - 1, 0})
                    if (left > 0) { n_w[p]++; left--; }
        }
        int best_kc = 2;
        {
            size_t i = 0;
            for (int p = 0; p < CV_WAVES; p++) {
                best_kc = std::max(best_kc, (n_w[p] + 1) / 2 * 2);
                for (int k = 0; k < n_w[p] && i < cols.size(); k++, i++) {
                    parts[p].chan[k] = cols[i].chan; parts[p].type[k] = cols[i].type; parts[p].n_col = k + 1;
                    h->cv_pidx[(size_t)p * CV_KC + k] = cols[i].pidx;
                }
            }
            if (i != cols.size()) return fail(h, SSDE_ERR_ARG, "internal: the design columns were not all dealt");
        }
        // the log sigma_obs and drift-intercept directions ride on the wave that runs the filter
        h->cv_sig_part = h->cv_mu_part = -1;
        if (!h->cv_full) {
            if (!h->fixed[0] && !h->has_h) { parts[0].with_sig = 1; h->cv_sig_part = 0; }     // (H_array: log sigma_obs is not in the model)
            for (auto& sl : h->slots)
                if (sl.par_j < h->d && !h->fixed[sl.pidx]) { parts[0].with_mu = 1; h->cv_mu_part = 0; }
        }
        h->cv_kc = h->cv_few ? ((N > CV_KC || h->n_stream_cols > CV_FEW_K) ? 2 * CV_KC : CV_KC) : h->cv_single ? CV_KC : best_kc;
        HIPCHK(h, h->cv_parts.upload(parts));
        HIPCHK(h, hipHostMalloc((void**)&h->cv_ranges_pinned, 4 * sizeof(double), hipHostMallocDefault));
        h->cv_ranges_pinned[0] = h->cv_ranges_pinned[2] = INFINITY; h->cv_ranges_pinned[1] = h->cv_ranges_pinned[3] = -INFINITY;
        {
            // the range of every streamed column over the batch
            const int K = h->n_stream_cols;
            DevBuf<double> rg;
            HIPCHK(h, rg.alloc((size_t)G * K * 2));
            TileView tv;
            memset(&tv, 0, sizeof(tv));
            tv.tiles = h->tiles.p; tv.group_off = h->group_off.p; tv.group_len = h->group_len.p; tv.lane_nsteps = h->lane_nsteps.p;
            tv.n_groups = G; tv.C = h->C; tv.c_obs = h->c_obs;
            HIPCHK(h, launch_colvar_ranges(tv, c_col, K, rg.p, 0));
            std::vector<double> rh((size_t)G * K * 2);
            HIPCHK(h, hipMemcpy(rh.data(), rg.p, rh.size() * 8, hipMemcpyDeviceToHost));
            h->cv_col_lo.assign(K, INFINITY); h->cv_col_hi.assign(K, -INFINITY);
            for (int g = 0; g < G; g++)
                for (int k = 0; k < K; k++) {
                    h->cv_col_lo[k] = std::min(h->cv_col_lo[k], rh[((size_t)g * K + k) * 2]);
                    h->cv_col_hi[k] = std::max(h->cv_col_hi[k], rh[((size_t)g * K + k) * 2 + 1]);
                }
        }
        h->drift_nstate = h->cv_adj ? adj_nstate(h->model, h->d, h->cv_full)
                        : h->cv_single ? (h->model == SSDE_MODEL_CTCRW ? 14 + 2 * 14 + 2 * 4 : 5 + 2 * 5 + 2 * 2) : colvar_nstate(h->model, h->d, h->cv_kc, h->cv_full);
        if (h->has_h) {
            DevBuf<double> hs;
            HIPCHK(h, hs.alloc((size_t)G * 2));
            TileView tv;
            memset(&tv, 0, sizeof(tv));
            tv.tiles = h->tiles.p; tv.group_off = h->group_off.p; tv.group_len = h->group_len.p; tv.lane_nsteps = h->lane_nsteps.p;
            tv.n_groups = G; tv.C = h->C; tv.c_obs = h->c_obs;
            HIPCHK(h, launch_colvar_h_stats(tv, h->c_obs + h->d, h->d, hs.p, 0));
            std::vector<double> hh((size_t)G * 2);
            HIPCHK(h, hipMemcpy(hh.data(), hs.p, hh.size() * 8, hipMemcpyDeviceToHost));
            double asym = 0.0;
            h->cv_hmax = 0.0;
            for (int g = 0; g < G; g++) { h->cv_hmax = std::max(h->cv_hmax, hh[2 * (size_t)g]); asym = std::max(asym, hh[2 * (size_t)g + 1]); }
            if (!(asym == 0.0) || !(h->cv_hmax > 0.0) || !std::isfinite(h->cv_hmax)) return SSDE_RETRY_WITHOUT_DRIFT;     // (not a covariance: the literal path)
        }
    } else
    if (h->drift) {
        // regular grid and every track complete: the shared-covariance lanes; otherwise the lanes carry their own covariance
        // (SSDE_NO_DRIFT_GENERAL: back to the lane = direction path instead, for A/B)
        bool all_clean = h->uniform_dt;
        for (int g = 0; g < G; g++) all_clean = all_clean && gflags[g] != 0;
        // (the lanes that carry their own covariance are issue-bound already: the table form costs them a third more -- tile the columns)
        if (!all_clean && h->pp_drift.nb > 0 && !getenv("SSDE_DRIFT_PP_ALL")) return SSDE_RETRY_WITHOUT_PP;
        if (!all_clean && getenv("SSDE_NO_DRIFT_GENERAL")) return SSDE_RETRY_WITHOUT_DRIFT;
        if (getenv("SSDE_NO_SHARED")) all_clean = false;                  // (testing: the general lanes on a batch the shared ones would take)
        h->drift = all_clean ? 1 : 2;
        h->drift_nstate = all_clean ? drift_nstate(h->model, h->d, h->n_stream_cols) : drift_general_nstate(h->model, h->d, h->n_stream_cols);
    }
    if (h->drift) { h->iso_parts = (h->drift == 3 && !h->cv_one_wave()) ? CV_WAVES : (h->cv_few ? h->cv_kc / CV_KC : 1); h->iso_masks[0] = DIR_SIG | DIR_MU | DIR_P1 | DIR_P2; h->iso_free_mask = h->iso_masks[0]; }
    else choose_iso_split(h);
    // shared-covariance path: regular grid + groups without missing rows
    HIPCHK(h, h->group_flags.upload(gflags));
    {
        std::vector<int32_t> dl;
        for (int g = 0; g < G; g++)
            if (!gflags[g]) dl.push_back(g);
        h->n_dirty_groups = (int)dl.size();
        if (!dl.empty()) HIPCHK(h, h->dirty_groups.upload(dl));
    }
    std::vector<int64_t> cnt;
    for (int g = 0; g < G; g++) {
        if (!gflags[g]) continue;
        h->n_clean_groups++;
        for (int l = 0; l < WAVE; l++) {
            const int ns = lane_ns[(size_t)g * WAVE + l];
            if (ns <= 0) continue;
            if ((size_t)ns >= cnt.size()) cnt.resize(ns + 1, 0);
            cnt[ns]++;
        }
    }
    for (size_t ns = 1; ns < cnt.size(); ns++)
        if (cnt[ns]) h->clean_ns_hist.push_back({(int)ns, cnt[ns]});
    h->use_shared = h->uniform_dt && h->n_clean_groups > 0 && h->iso_parts == 1 && !getenv("SSDE_NO_SHARED");
    if (h->drift) h->use_shared = h->drift == 1;
    // (two concurrent launches, a fork / join and a longer finalize cost ~40 us: with fewer than a quarter of the groups on
    //  the shared kernel that is more than it wins -- measured 0.82 against 0.78 ms at one tenth -- and everything stays general)
    if (h->use_shared && h->n_clean_groups < G && 4 * h->n_clean_groups < G && !getenv("SSDE_SHARED_ALWAYS")) h->use_shared = false;
    // time windows: enough (group, window, part) workgroups for ~2 waves on each of the 1024 SIMDs
    int glmax = 0;
    for (int g = 0; g < G; g++) glmax = std::max(glmax, glen[g]);
    h->glen_max = glmax;
    // one wave per SIMD (1024 work items INCLUDING the padding of the group count to a multiple of 8):
    // the shared-covariance lanes are bound by memory -- a lone wave per SIMD reaches the read ceiling (profiles/r04_a_issue_microbench.txt;
    // pure fp64 issue is another matter: 6.5 cycles per instruction for a lone wave against 4.7 with four, profiles/r05_fp64_issue_microbench.txt) --
    // and fewer windows mean fewer warm-up rows
    int want = std::max(1, 1024 / (((G + 7) / 8 * 8) * h->iso_parts));
    // ... except for the scalar-covariance models on the general kernel (irregular grid or missing rows in
    // most groups): too few independent chains per row for one wave, so two waves per SIMD (k_iso.hip)
    // (k_iso.hip: every general kernel but CTCRW's irregular-grid one is built for two waves per SIMD)
    // (a mixed batch whose general launch gets a plan of its own, below, keeps the shared kernel's plan here whatever the
    //  share of its groups)
    const bool own_plans = h->use_shared && h->n_clean_groups < G && !getenv("SSDE_CHUNKS") && !getenv("SSDE_ONE_PLAN");
    if (!h->drift && !(h->model == SSDE_MODEL_CTCRW && !h->uniform_dt) && (!h->uniform_dt || (2 * h->n_clean_groups < G && !own_plans)) && !getenv("SSDE_NO_LIGHT2"))
        // measured (tools/bench_na.py, SSDE_CHUNKS sweep 12 .. 32): the scalar-covariance models run best with
        // 1.5 work items per wave slot (shorter items even out the tail; their hand-over dumps are small), CTCRW
        // with one (its 32-component dumps make every further boundary cost what the shorter tail gains)
        want = std::max(1, (h->model == SSDE_MODEL_CTCRW ? 2048 : 3072) / (((G + 7) / 8 * 8) * h->iso_parts));
    // row-varying tau / nu: a WORKGROUP per (group, window), one per CU -- up to eight rounds' worth; plan_windows picks the count
    if (h->drift == 3 && !h->cv_one_wave()) want = std::max(1, (8 * 256 + G - 1) / G);
    // a smooth drift evaluated from tables: the lanes wait for LDS, two waves per SIMD hide each other's round trips (1.88 -> 1.24 ms)
    if (h->drift && h->pp_drift.nb > 0) want = std::max(1, 2048 / (((G + 7) / 8 * 8) * h->iso_parts));
    // The resident tiles against the 256 MB Infinity Cache: a batch that fits is still there at the next evaluation if the loads
    // are ordinary ones; a batch far beyond it is streamed past the caches (measured, profiles/r04_a_nt_vs_plain.txt: 205 MB
    // 0.053 -> 0.044 ms with ordinary loads, 260 / 330 / 410 MB no difference, 1.6 GB 0.259 -> 0.274 ms)
    h->stream_nt = (double)h->tile_doubles * 8.0 > 300e6;
    if (const char* e = getenv("SSDE_NT")) h->stream_nt = atoi(e) != 0;
    if (const char* e = getenv("SSDE_CHUNKS")) { want = atoi(e); h->chunks_forced = true; }   // testing
    h->max_chunks = std::max(1, std::min(want + 1, std::max(1, glmax / (4 * WIN_ALIGN))));
    // (row-varying tau / nu with few groups: windows down to two alignment units, shorter than their warm-up -- with CUs idle
    //  the redundant warm-up rows run in parallel, only a workgroup's own chain of rows matters; ssde_engine_iso.hip picks)
    if (h->drift == 3 && !h->cv_one_wave() && !h->chunks_forced) h->max_chunks = std::max(1, std::min(want + 1, std::max(1, glmax / (2 * WIN_ALIGN))));
    // (the reverse sweep: plan_windows picks the count by cost, up to two alignment units per window)
    if (h->cv_adj && !h->chunks_forced) h->max_chunks = std::max(1, std::min((1024 + G - 1) / G * 2, std::max(1, glmax / (2 * WIN_ALIGN))));
    h->want_chunks = std::max(1, std::min(want, h->max_chunks));
    // Mixed batch: most wavefronts on the shared-covariance kernel, the few that hold the tracks with missing rows on
    // the general kernel.  With ONE plan -- the shared kernel's few long windows -- the general launch is a handful of
    // waves each running a seventh of a track at 0.5 us per row, and its critical path, not its share of the rows,
    // sets the evaluation's time (0.75-0.9 ms with 1-30 % of the tracks affected).  The general launch gets a plan of
    // its own: enough windows to fill its two waves per SIMD (plan_windows keeps them at least two warm-ups long).
    int buf_chunks = h->max_chunks;
    if (h->use_shared && !h->drift && h->n_clean_groups < G && !h->chunks_forced && !getenv("SSDE_ONE_PLAN")) {
        const int gd8 = ((G - h->n_clean_groups) + 7) / 8 * 8;
        const int want_d = std::max(1, (h->model == SSDE_MODEL_CTCRW ? 2048 : 3072) / gd8);
        // (the hand-over dumps are sized for every group x the longer plan: keep them under ~0.5 GB)
        const int by_mem = std::max(2, (int)(512e6 / ((double)G * 2.0 * NSTATE_MAX * WAVE * 8.0)) - 1);
        h->want_chunks_d = std::max(1, std::min(std::min(std::min(want_d, 64), by_mem), std::max(1, glmax / (4 * WIN_ALIGN))));
        buf_chunks = std::max(buf_chunks, h->want_chunks_d + 1);
    }
    // Quiet rows: a regular-grid batch whose groups hold missing rows here and there (no complete group to keep on the shared
    // kernel, or some) -- the general lanes drop their covariance wherever it is stationary on all 64 lanes (k_iso.hip).  Worth it
    // when a fair share of the blocks qualifies (with missing rows in every block of every group the flags cost a little and win nothing).
    if (h->uniform_dt && !h->drift && h->iso_parts == 1 && h->n_clean_groups < G && glmax >= 8 * WIN_ALIGN && !getenv("SSDE_NO_QUIET")) {
        const int U = iso_block_rows(h->model);
        h->nan_words = (glmax / U + 63) / 64 + 1;
        HIPCHK(h, h->nan_bits.alloc((size_t)G * h->nan_words));
        HIPCHK(h, hipMemset(h->nan_bits.p, 0, (size_t)G * h->nan_words * 8));
        TileView tv;
        tv.tiles = h->tiles.p; tv.group_off = h->group_off.p; tv.group_len = h->group_len.p; tv.lane_nsteps = h->lane_nsteps.p;
        tv.a0 = h->a0.p; tv.n_groups = G; tv.C = h->C; tv.c_obs = h->c_obs; tv.dt_all = h->dt_all;
        HIPCHK(h, launch_nan_blocks(tv, h->d, U, h->nan_bits.p, h->nan_words, 0));
        std::vector<unsigned long long> bits((size_t)G * h->nan_words);
        HIPCHK(h, hipMemcpy(bits.data(), h->nan_bits.p, bits.size() * 8, hipMemcpyDeviceToHost));
        // share of the dirty groups' blocks that would be quiet with a nominal 128-row memory
        const int wq = 128 / U;
        int64_t n_blocks = 0, n_quiet = 0;
        for (int g = 0; g < G; g++) {
            if (gflags[g]) continue;
            const int nb = glen[g] / U;
            int last = -wq - 1;
            for (int b = 0; b < nb; b++) {
                if ((bits[(size_t)g * h->nan_words + (b >> 6)] >> (b & 63)) & 1ull) last = b;
                n_quiet += (b - last > wq) ? 1 : 0;
            }
            n_blocks += nb;
        }
        h->quiet_share = n_blocks > 0 ? (double)n_quiet / (double)n_blocks : 0.0;
        // (measured, 10^4 x 10^4 CTCRW with 1 / 2 / 3 / 5 missing rows per track = shares 0.98 / 0.53 / 0.33 / 0.18: 0.40 / 0.70 / 0.78 /
        //  0.86 ms against 0.84-0.87 without quiet rows)
        const double need = getenv("SSDE_QUIET_ALWAYS") ? 0.0 : h->model == SSDE_MODEL_CTCRW ? 0.25 : 0.2;
        h->quiet_ok = n_blocks > 0 && (double)n_quiet >= need * (double)n_blocks;
        if (!h->quiet_ok) { h->nan_bits.release(); h->nan_words = 0; }
        else { HIPCHK(h, h->quiet_flag.alloc(1)); HIPCHK(h, hipMemset(h->quiet_flag.p, 0, 8)); }
        if (const char* e = getenv("SSDE_QUIET_WINDOW")) h->env_quiet_window = std::max(0, atoi(e));
        if (h->quiet_ok && !h->chunks_forced) {
            // nearly every row a quiet row: the rows cost what the shared-covariance kernels' cost, and windows cost warm-up rows and
            // hand-over checks -- one work item per SIMD (10^4 tracks: 6 windows 0.40 ms, 12 windows 0.435); otherwise one per wave
            // slot (two waves per SIMD), which evens out the stretches of general rows (2 missing rows per track: 0.70 against 0.84)
            const int slots = getenv("SSDE_QUIET_SLOTS") ? atoi(getenv("SSDE_QUIET_SLOTS")) : h->quiet_share >= 0.9 ? 1024 : 2048;
            if (h->use_shared && h->want_chunks_d > 0) {
                const int gd8 = ((G - h->n_clean_groups) + 7) / 8 * 8;
                h->want_chunks_d = std::max(1, std::min(h->want_chunks_d, slots / gd8));
            } else if (!h->use_shared) {
                h->want_chunks = std::max(1, std::min(h->want_chunks, slots / ((G + 7) / 8 * 8)));
            }
        }
    }
    if (h->use_shared || h->quiet_ok) {
        h->gain_rows_cap = (size_t)glmax + 1;
        HIPCHK(h, h->gain_ring.alloc((size_t)PAR_RING * h->gain_rows_cap * GAIN_ROW));
        HIPCHK(h, hipHostMalloc((void**)&h->gain_pinned, (size_t)PAR_RING * h->gain_rows_cap * GAIN_ROW * 8,
                                hipHostMallocDefault));
    }
    HIPCHK(h, h->bnd.alloc((size_t)h->iso_parts * (buf_chunks + (h->cv_adj ? 1 : 0)) * G * 2 * (h->drift ? std::max(NSTATE_MAX, h->drift_nstate) : NSTATE_MAX) * WAVE));
    HIPCHK(h, h->chk.alloc((size_t)h->iso_parts * buf_chunks * G));
    if (h->use_shared && !h->drift) {
        HIPCHK(h, h->fuse_words.alloc(4 + (size_t)(buf_chunks + 1) * G));
        HIPCHK(h, hipMemset(h->fuse_words.p, 0, h->fuse_words.n * sizeof(unsigned)));
        // (measured, profiles/r05_fused_finalize_ab.txt: SLOWER than the dependent launch it replaces -- a wave's way from "my rows are
        //  done" to "the result is out" is six device-scope round trips of ~2 us across the XCDs -- so the two-launch form stays the
        //  default and SSDE_FUSED_FINALIZE=1 selects this one, bitwise the same numbers)
        h->env_no_fused = true;
        if (const char* e = getenv("SSDE_FUSED_FINALIZE")) h->env_no_fused = atoi(e) == 0;
    }
    if (h->drift == 3) HIPCHK(h, hipMemset(h->bnd.p, 0, h->bnd.n * 8));      // (a part dumps its own block of a hand-over record; the check reads all of it)
    h->partial_doubles = (size_t)std::max(MAX_PARTS, CV_WAVES) * buf_chunks * std::max(NACC_MAX, 2 + CV_KC + 2) * G;
    if (h->cv_adj) h->partial_doubles = std::max(h->partial_doubles, (size_t)buf_chunks * adj_nacc(h->model, h->d, h->n_stream_cols, true) * G);
    h->hbm_bytes += (int64_t)(h->bnd.n + h->chk.n) * 8;
    return SSDE_OK;
}

static int build_impl(const ssde_desc* d, ssde_handle* h, const ParLayout* part_layout, bool allow_drift);
int build(const ssde_desc* d, ssde_handle* h, const ParLayout* part_layout) {
    // (what the caller set on the handle before build() survives a retry: the reset below wipes everything else -- ADVICE r04)
    const bool force_tv = h->force_tv, wide_ok = h->wide_ok;
    bool no_drift_pp = h->no_drift_pp;
    auto reset = [&]() {
        release_device(h);
        *h = ssde_handle();
        h->force_tv = force_tv; h->wide_ok = wide_ok; h->no_drift_pp = no_drift_pp;
    };
    int st = build_impl(d, h, part_layout, true);
    if (st == SSDE_RETRY_WITHOUT_PP) {
        no_drift_pp = true;
        reset();
        st = build_impl(d, h, part_layout, true);
    }
    if (st == SSDE_RETRY_WITHOUT_DRIFT) {
        // the row-varying-drift layout needs a regular grid and no missing row, which only the tiling pass finds out:
        // start over on the path such a batch takes otherwise
        reset();
        st = build_impl(d, h, part_layout, false);
    }
    return st;
}

// ---- ssde_create, step 1: the descriptor -- model / dimension checks, parameter layout, coefficient slots, the map ----
static int check_descriptor(const ssde_desc* d, ssde_handle* h, const ParLayout* part_layout) {
    // ---- descriptor checks -------------------------------------------------------------------
    if (d->abi_version != SSDE_ABI_VERSION) return fail(h, SSDE_ERR_ARG, "ssde_desc.abi_version mismatch");
    if (d->model < SSDE_MODEL_BM || d->model > SSDE_MODEL_CIR) return fail(h, SSDE_ERR_MODEL, "Unknown SDE type");
    if (is_eseal(d->model)) {
        // nllk_e_seal_ssm.hpp: one response, state (1, lipid mass) with a0 = (1, L0) and P0 = diag(0, p0) (R/sde.R:602-603):
        // the constant first component is what turns the 2 x 2 filter into the scalar filter of ssde_tv.hpp
        if (d->n_dim != 1) return fail(h, SSDE_ERR_MODEL, "ESEAL_SSM takes one response variable");
        if (!d->a0 || !d->eseal_h || !d->eseal_R) return fail(h, SSDE_ERR_ARG, "ESEAL_SSM needs a0, eseal_h and eseal_R");
        if (d->p0 && (d->p0[0] != 0.0 || d->p0[1] != 0.0 || d->p0[2] != 0.0))
            return fail(h, SSDE_ERR_ARG, "ESEAL_SSM: P0 must be diag(0, p0) (R/sde.R:603)");
        if (d->flags & SSDE_FLAG_DEVICE_DATA) return fail(h, SSDE_ERR_ARG, "ESEAL_SSM takes host arrays");
    }
    if (d->model == SSDE_MODEL_BM_T) {
        // tr_dens.hpp:38-44 reads par(0), par(1) whatever the dimension: one response variable
        if (d->n_dim != 1) return fail(h, SSDE_ERR_MODEL, "BM_t takes one response variable");
        if (!d->other_data || d->n_other_data < 1 || !(d->other_data[0] > 2.0))
            return fail(h, SSDE_ERR_ARG, "BM_t needs other_data[0] = degrees of freedom > 2");
        h->tdf = d->other_data[0];
        h->tconst = std::lgamma(0.5 * (h->tdf + 1.0)) - std::lgamma(0.5 * h->tdf) - 0.5 * std::log(h->tdf * M_PI);
    }
    // (three to eight response columns as ONE filter: only the lane = track general kernel, and only where ssde_create found that the
    //  measurement covariance or P0 couples the column pairs -- every other wide response is evaluated pair by pair)
    if (d->n_dim < 1 || (d->n_dim > 2 && !(h->wide_ok && d->n_dim <= DENSE_MAXD && is_kalman(d->model))))
        return fail(h, SSDE_ERR_MODEL, "n_dim must be 1 or 2 (wider responses are outside this engine's kernels)");
    if (d->n_par != n_sde_par(d->model, d->n_dim)) return fail(h, SSDE_ERR_ARG, "n_par does not match model / n_dim");
    if (d->n < 2) return fail(h, SSDE_ERR_ARG, "need at least two rows");
    if (!d->id || !d->times || !d->obs || !d->ncol_fe) return fail(h, SSDE_ERR_ARG, "id/times/obs/ncol_fe must be non-NULL");
    h->model = d->model; h->d = d->n_dim; h->q = d->n_par; h->n = d->n;
    if (const char* e = getenv("SSDE_WINDOW")) h->env_window = std::max(1, atoi(e));     // testing: deliberately short overlaps
    if (const char* e = getenv("SSDE_ADJ_TAIL")) h->env_adj_tail = std::max(1, atoi(e));  // testing: ... of the backward recursion only (k_iso_adj.hip)
    if (const char* e = getenv("SSDE_TV_WAVES")) h->env_tv_waves = std::max(1, atoi(e));
    if (const char* e = getenv("SSDE_TV_MINLEN")) h->env_tv_minlen = std::max(WIN_ALIGN, atoi(e) / WIN_ALIGN * WIN_ALIGN);
    if (const char* e = getenv("SSDE_T0_COST")) h->env_t0_cost = atof(e);
    if (const char* e = getenv("SSDE_W0_RATIO")) h->env_w0_ratio = atof(e);     // 0 = equal windows on the general kernel
    h->env_no_derive = getenv("SSDE_NO_DERIVE") != nullptr;
    h->env_no_graph = getenv("SSDE_NO_GRAPH") != nullptr;
    h->env_no_exact_hess = getenv("SSDE_NO_EXACT_HESS") != nullptr;
    h->env_own_stream = getenv("SSDE_SYNC_OWN_STREAM") != nullptr;
    h->trace = getenv("SSDE_TRACE") != nullptr;
    if (const char* e = getenv("SSDE_WAVE_CLOCK")) h->wave_clock_file = e;
    h->sdim = state_dim(d->model, d->n_dim);
    h->na_any = d->na_mode == SSDE_NA_ANY_NAN;
    h->has_h = is_kalman(d->model) && d->h_array != nullptr;
    for (int j = 0; j < d->n_par; j++) {
        if (d->ncol_fe[j] < 1) return fail(h, SSDE_ERR_ARG, "every SDE parameter needs at least one fixed-effect column");
        if (!(d->x_fe && d->x_fe[j]) && d->ncol_fe[j] != 1)
            return fail(h, SSDE_ERR_ARG, "x_fe[j] == NULL means intercept-only: ncol_fe[j] must be 1");
        const ssde_ppbasis* pb = d->basis_re ? d->basis_re[j] : nullptr;
        if (d->ncol_re && d->ncol_re[j] > 0 && !(d->x_re && d->x_re[j]) && !pb)
            return fail(h, SSDE_ERR_ARG, "x_re[j] missing for a parameter with random-effect columns");
        if (pb && (!d->ncol_re || pb->n_cols != d->ncol_re[j] || pb->n_knots < 2 || !pb->x || !pb->knots || !pb->coef))
            return fail(h, SSDE_ERR_ARG, "basis_re[j]: n_cols must equal ncol_re[j]; x, knots (>= 2) and coef are required");
    }
    if (d->n_decay > 0) {
        if (is_kalman(d->model)) return fail(h, SSDE_ERR_ARG, "decaying terms are a feature of the direct families (nllk_sde.hpp:47-58)");
        if (d->n_decay > MAX_DECAY) return fail(h, SSDE_ERR_ARG, "more than 4 decay rates");
        if (!d->t_decay || !d->col_decay || !d->ind_decay || d->n_decay_cols < 1) return fail(h, SSDE_ERR_ARG, "t_decay / col_decay / ind_decay missing");
        for (int c = 0; c < d->n_decay_cols; c++)
            if (d->ind_decay[c] < 0 || d->ind_decay[c] >= d->n_decay) return fail(h, SSDE_ERR_ARG, "ind_decay out of range");
    }
    // a dimension part of a wider problem (ssde_engine_dist.hip) indexes the WHOLE problem's parameter vector: its layout,
    // the decay / smooth bookkeeping and the penalty are the parent's, validated there
    h->L = part_layout ? *part_layout : make_layout(d);
    if (d->n_decay > 0 && !part_layout) {
        // a column index that matches no random-effect column would silently not decay while log_decay stays a
        // free parameter with a zero gradient (the reference stops on an unknown name, R/sde.R:637-640)
        std::vector<uint8_t> seen((size_t)std::max(h->L.n_re, 1), 0);
        for (int c = 0; c < d->n_decay_cols; c++) {
            const int k = d->col_decay[c];
            if (k < 0 || k >= h->L.n_re) return fail(h, SSDE_ERR_ARG, "col_decay out of range (0-based index into coeff_re)");
            if (seen[k]) return fail(h, SSDE_ERR_ARG, "col_decay names a column twice");
            seen[k] = 1;
        }
    }
    if (h->L.n_full > MAX_PAR) return fail(h, SSDE_ERR_ARG, "too many parameters for the kernel argument block");
    if (!part_layout) {
        int nsm = 0;
        for (int s = 0; s < d->n_smooth; s++) nsm += d->smooth_ncol[s];
        if (nsm != h->L.n_re) return fail(h, SSDE_ERR_ARG, "smooth_ncol does not add up to the random-effect columns");
        h->pen.setup(d);
    }
    h->slots = make_slots(d, h->L, &h->n_stream_cols);
    if ((int)h->slots.size() > MAX_COLS || h->n_stream_cols > MAX_COLS)
        return fail(h, SSDE_ERR_ARG, "too many design columns (limit 96)");
    h->const_coeff = h->n_stream_cols == 0;
    h->fixed.assign(h->L.n_full, 0);
    if (d->par_fixed) h->fixed.assign(d->par_fixed, d->par_fixed + h->L.n_full);
    if (h->has_h) h->fixed[0] = 1;  // log_sigma_obs is mapped when H is supplied (R/sde.R:565, 595)
    // log_lambda never enters the data term
    h->n_free = 0;
    for (int k = 0; k < h->L.n_full; k++) h->n_free += h->fixed[k] ? 0 : 1;

    return SSDE_OK;
}

// ---- step 2: random-effect blocks given as piecewise-cubic tables (ssde_ppbasis): staged, and materialised where no kernel evaluates them ----
static int stage_basis_tables(const ssde_desc* d, ssde_handle* h, bool on_dev, int64_t n) {
    // ---- design blocks given as functions of a covariate (ssde_ppbasis) ---------------------------------------
    // Fast route: a direct family whose fast kernel applies (<= 2 parameters with columns, no decay), the whole
    // random-effect block of the parameter is the table and its fixed-effect part is the intercept: the kernel
    // evaluates the block from x (8 B/row).  Everything else gets the dense block materialised once in HBM.
    if (d->basis_re && d->n_par > MAX_Q) {
        // (a response of three to eight columns run as one filter: q = d + 2 > MAX_Q parameters, and the table slots are MAX_Q wide)
        for (int j = 0; j < d->n_par; j++)
            if (d->basis_re[j]) return fail(h, SSDE_ERR_MODEL, "basis_re is not available for responses wider than two columns that run as one filter (coupling H_array / P0): pass the dense block");
    } else
    if (d->basis_re) {
        std::vector<int> with_cols;
        for (auto& sl : h->slots)
            if (sl.col >= 0 && (with_cols.empty() || with_cols.back() != sl.par_j)) with_cols.push_back(sl.par_j);
        // the on-the-fly route exists in the FAST direct kernel only, so everything that later decides direct_fast is
        // decided here already: at most two parameters with columns, none of them with more than DIRECT_KCAP, no decay
        // (a table-backed block left without a resident column on the generic kernel would be scored as an intercept)
        int cols_of[MAX_Q] = {0, 0, 0, 0};
        for (auto& sl : h->slots)
            if (sl.col >= 0) cols_of[sl.par_j]++;
        bool kcap_ok = true;
        for (int j = 0; j < MAX_Q; j++) kcap_ok = kcap_ok && cols_of[j] <= DIRECT_KCAP;
        const bool fast_family = !is_kalman(d->model) && !is_eseal(d->model) && h->L.n_decay == 0 && with_cols.size() <= 2 &&
                                 kcap_ok && !getenv("SSDE_NO_DIRECT_FAST") && !getenv("SSDE_NO_PP_FAST");
        for (int j = 0; j < d->n_par; j++) {
            const ssde_ppbasis* pb = d->basis_re[j];
            if (!pb) continue;
            const int K = pb->n_cols, nk = pb->n_knots;
            for (int k = 1; k < nk; k++)
                if (!(pb->knots[k] > pb->knots[k - 1])) return fail(h, SSDE_ERR_ARG, "basis_re[j]: knots must increase");
            HIPCHK(h, stage(pb->knots, (size_t)nk, false, h->pp_knots[j]));
            HIPCHK(h, stage(pb->coef, (size_t)(nk - 1) * K * 4, false, h->pp_tab[j]));
            // engine-owned copy in either case: ssde.h promises that nothing of the caller's is aliased after create
            HIPCHK(h, stage(pb->x, (size_t)n, on_dev, h->pp_x[j]));
            const double* xdev = h->pp_x[j].p;
            PPRef& P = h->pp[j];
            P.x = xdev; P.knots = h->pp_knots[j].p; P.tab = h->pp_tab[j].p; P.nk = nk;
            const double hstep = (pb->knots[nk - 1] - pb->knots[0]) / (nk - 1);
            P.uniform = 1;
            for (int k = 0; k < nk; k++)
                if (std::fabs(pb->knots[k] - (pb->knots[0] + k * hstep)) > 1e-12 * std::fabs(hstep) * nk) P.uniform = 0;
            P.k0 = pb->knots[0]; P.inv_h = 1.0 / hstep;
            h->pp_fast[j] = fast_family && !(d->x_fe && d->x_fe[j]) && K <= DIRECT_KCAP && (nk - 1) * K * 4 + nk <= PP_LDS;
            if (!h->pp_fast[j]) {
                HIPCHK(h, h->pp_mat[j].alloc((size_t)n * K));
                HIPCHK(h, launch_pp_materialise(P, K, n, h->pp_mat[j].p, n, 0));
                HIPCHK(h, hipDeviceSynchronize());
                for (auto& sl : h->slots)
                    if (sl.par_j == j && sl.basis_c >= 0) sl.src = h->pp_mat[j].p + (size_t)sl.basis_c * n;
            }
        }
    }

    return SSDE_OK;
}

// ---- step 3: the ID segments (tracks), one-row tracks, the result buffers every path shares ----
static int find_segments(const ssde_desc* d, ssde_handle* h, bool on_dev, int64_t n, std::vector<int64_t>& starts) {
    {
        std::vector<uint8_t> flags;
        if (on_dev) {
            DevBuf<uint8_t> f;
            HIPCHK(h, f.alloc(n));
            HIPCHK(h, launch_first_flags(d->id, n, f.p, 0));
            flags.resize(n);
            HIPCHK(h, hipMemcpy(flags.data(), f.p, n, hipMemcpyDeviceToHost));
            f.release();
            for (int64_t i = 0; i < n; i++)
                if (flags[i]) starts.push_back(i);
        } else {
            for (int64_t i = 0; i < n; i++)
                if (i == 0 || d->id[i] != d->id[i - 1]) starts.push_back(i);
        }
    }
    h->n_seg = (int64_t)starts.size();
    if (d->a0 && d->n_seg != h->n_seg) return fail(h, SSDE_ERR_ARG, "a0 rows do not match the number of ID segments");
    if (is_eseal(d->model))   // (after the row count of a0 is known to be right)
        for (int64_t sgi = 0; sgi < h->n_seg; sgi++)
            if (d->a0[sgi] != 1.0) return fail(h, SSDE_ERR_ARG, "ESEAL_SSM: the first column of a0 must be 1 (R/sde.R:602)");
    h->n_steps = n - h->n_seg;
    starts.push_back(n);
    if (is_kalman(d->model)) {
        // a track of one row is initialised and never stepped: the kernels have no lane for it, but the reference's
        // aest_all carries its initial state (a0 row, or the default of R/sde.R:576-580: first observation, velocity 0)
        for (int64_t seg = 0; seg < h->n_seg; seg++) {
            if (starts[seg + 1] - starts[seg] != 1) continue;
            const int64_t row = starts[seg];
            h->single_rows.push_back(row);
            for (int c = 0; c < h->sdim; c++) {
                double v = 0.0;
                if (d->a0) {
                    v = d->a0[seg + (int64_t)c * h->n_seg];
                } else if (d->model != SSDE_MODEL_CTCRW || c % 2 == 0) {
                    const int a = d->model == SSDE_MODEL_CTCRW ? c / 2 : c;
                    const double* src = d->obs + row + (int64_t)a * n;
                    if (on_dev) HIPCHK(h, hipMemcpy(&v, src, 8, hipMemcpyDeviceToHost));
                    else v = *src;
                }
                h->single_a0.push_back(v);
            }
        }
    }

    HIPCHK(h, h->out.alloc(2 + h->L.n_full));
    HIPCHK(h, hipHostMalloc((void**)&h->out_pinned, (size_t)(2 + h->L.n_full) * 8, hipHostMallocDefault));
    {
        const size_t words = ((size_t)(2 + h->L.n_full) + 15) / 16 * 16;           // the sequence word gets a 128-byte line of its own
        HIPCHK(h, hipHostMalloc((void**)&h->pub_pinned, (words + 16) * 8, hipHostMallocDefault));
        memset(h->pub_pinned, 0, (words + 16) * 8);
        h->pub_flag = (unsigned long long*)(h->pub_pinned + words);
        HIPCHK(h, h->pub_count.alloc(1));
        HIPCHK(h, hipMemset(h->pub_count.p, 0, sizeof(unsigned int)));
        // measured in the engine (tools/bench_strong.py, same session): 27-30 us outside the kernel either way -- the read-back
        // copy it saves is paid back in the counting and the system-scope store; opt-in, for the record
        h->pub_ok = getenv("SSDE_PUBLISH") != nullptr;
    }
    for (auto& pr : h->ev_ring) { HIPCHK(h, hipEventCreate(&pr[0])); HIPCHK(h, hipEventCreate(&pr[1])); }
    h->ev_k0 = h->ev_ring[0][0]; h->ev_k1 = h->ev_ring[0][1];

    return SSDE_OK;
}

// ---- step 4a: the direct families keep the long format (lane = row is already coalesced) ----
static int build_direct(const ssde_desc* d, ssde_handle* h, bool on_dev, int64_t n) {
    h->path = PATH_DIRECT;
    HIPCHK(h, stage(d->times, (size_t)n, on_dev, h->times));
    HIPCHK(h, stage(d->obs, (size_t)n * d->n_dim, on_dev, h->obs));
    // column stride: padded so that the same row of different columns does not fall on addresses that are
    // equal modulo a large power of two (all columns of a row are fetched together)
    h->col_stride = ((n + 63) / 64) * 64 + 160;
    if (const char* e = getenv("SSDE_COL_PAD")) h->col_stride = ((n + 63) / 64) * 64 + atoi(e);
    // columns evaluated on the fly from a basis table need no copy: col = -2
    int ncb = 0;
    for (auto& s : h->slots)
        if (s.col >= 0) s.col = s.src ? ncb++ : -2;
    HIPCHK(h, h->colbuf.alloc((size_t)h->col_stride * ncb));
    std::vector<const double*> cp(ncb, nullptr);
    for (auto& s : h->slots)
        if (s.col >= 0) {
            double* dst = h->colbuf.p + (size_t)s.col * h->col_stride;
            HIPCHK(h, hipMemcpy(dst, s.src, (size_t)n * 8, hipMemcpyDefault));   // caller's array or a materialised basis block
            cp[s.col] = dst;
        }
    h->n_stream_cols = ncb;
    HIPCHK(h, h->colptr.upload(cp));
    if (h->L.n_decay > 0) HIPCHK(h, stage(d->t_decay, (size_t)n * d->n_par, on_dev, h->tdecay));
    {
        DevBuf<double> idb;
        const double* idp = d->id;
        if (!on_dev) { HIPCHK(h, stage(d->id, (size_t)n, false, idb)); idp = idb.p; }
        HIPCHK(h, h->scored.alloc((size_t)((n + 31) / 32)));
        HIPCHK(h, launch_scored_mask(idp, n, h->scored.p, 0));
        HIPCHK(h, hipDeviceSynchronize());
        idb.release();
    }
    h->direct_blocks = (int)std::min<int64_t>((n + 255) / 256, 2048);
    if (const char* e = getenv("SSDE_DIRECT_BLOCKS")) h->direct_blocks = std::max(1, std::min(atoi(e), 65536));
    h->partial_doubles = (size_t)(1 + MAX_Q + h->slots.size() + MAX_DECAY) * h->direct_blocks;
    {
        // regular grid?  (min / max over the scored intervals)
        const int nb = 1024;
        DevBuf<double> mm;
        HIPCHK(h, mm.alloc((size_t)nb * 2));
        HIPCHK(h, launch_dt_minmax(h->times.p, h->scored.p, n, mm.p, nb, 0));
        std::vector<double> mmh((size_t)nb * 2);
        HIPCHK(h, hipMemcpy(mmh.data(), mm.p, mmh.size() * 8, hipMemcpyDeviceToHost));
        double dmin = INFINITY, dmax = -INFINITY;
        for (int b = 0; b < nb; b++) { dmin = std::min(dmin, mmh[2 * b]); dmax = std::max(dmax, mmh[2 * b + 1]); }
        // regular to the last bits counts as regular (the tolerance and its argument: lattice_pad)
        double rtol = 1e-12;
        if (const char* e = getenv("SSDE_GRID_RTOL")) rtol = std::max(0.0, atof(e));
        h->direct_uniform_dt = (dmax <= dmin * (1.0 + rtol)) && dmin > 0.0 && std::isfinite(dmin) && !(d->flags & SSDE_FLAG_NO_UNIFORM_DT);
        h->direct_dt = h->direct_uniform_dt ? 0.5 * (dmin + dmax) : 0.0;
        h->uniform_dt = h->direct_uniform_dt;
        mm.release();
        // which parameters have streamed columns (slots are ordered parameter by parameter)
        std::vector<int> streamed_par;
        for (auto& sl : h->slots) {
            if (sl.col == -1) { h->df_icpt[sl.par_j] = sl.pidx; continue; }
            if (streamed_par.empty() || streamed_par.back() != sl.par_j) streamed_par.push_back(sl.par_j);
        }
        bool ok = streamed_par.size() <= 2 && !getenv("SSDE_NO_DIRECT_FAST") && h->L.n_decay == 0;   // decaying columns: generic kernel
        if (ok) {
            for (auto& sl : h->slots) {
                if (sl.col == -1) continue;
                const bool isA = sl.par_j == streamed_par[0];
                auto& pid = isA ? h->df_pidxA : h->df_pidxB;
                const double*& base = isA ? h->df_colA : h->df_colB;
                if (sl.col == -2) { pid.push_back(sl.pidx); continue; }          // evaluated from the basis table
                if (pid.empty()) base = h->colbuf.p + (size_t)sl.col * h->col_stride;
                else if (h->colbuf.p + (size_t)sl.col * h->col_stride != base + pid.size() * (size_t)h->col_stride) ok = false;  // contiguous
                pid.push_back(sl.pidx);
            }
            if ((int)h->df_pidxA.size() > DIRECT_KCAP || (int)h->df_pidxB.size() > DIRECT_KCAP) ok = false;
        }
        h->direct_fast = ok;
        if (!ok)
            for (int j = 0; j < MAX_Q; j++)
                if (h->pp_fast[j]) return fail(h, SSDE_ERR_ARG, "internal: a basis table was left unmaterialised for the generic direct kernel");
        if (ok) {
            h->df_ja = streamed_par.size() > 0 ? streamed_par[0] : -1;
            h->df_jb = streamed_par.size() > 1 ? streamed_par[1] : -1;
        }
    }
    h->hbm_bytes = (int64_t)(h->times.n + h->obs.n + h->colbuf.n) * 8 + (int64_t)h->scored.n * 4;
    return SSDE_OK;
}

// Row-varying tau / nu: a design column of par[d + 1] that holds the same numbers as one of par[d] (the same smooth of the same
// covariate in both formulas -- the usual case) is streamed ONCE: the two coefficients share the tile channel.  Returns the
// number of distinct columns; `col` of the slots is renumbered.
static int share_equal_columns(const ssde_desc* d, ssde_handle* h, bool on_dev, int64_t n, int* n_distinct) {
    DevBuf<int> flag;
    if (on_dev) HIPCHK(h, flag.alloc(1));
    auto equal = [&](const double* a, const double* b, bool* eq) -> int {
        *eq = false;
        if (a == b) { *eq = true; return SSDE_OK; }
        if (!a || !b) return SSDE_OK;
        if (on_dev) {
            HIPCHK(h, hipMemset(flag.p, 0, sizeof(int)));
            HIPCHK(h, launch_cols_differ(a, b, n, flag.p, 0));
            int f = 1;
            HIPCHK(h, hipMemcpy(&f, flag.p, sizeof(int), hipMemcpyDeviceToHost));
            *eq = f == 0;
            return SSDE_OK;
        }
        for (int k = 0; k < 64; k++) {                             // a sample first: most pairs differ at once
            const int64_t i = (int64_t)((double)k / 63.0 * (double)(n - 1));
            if (memcmp(a + i, b + i, 8) != 0) return SSDE_OK;
        }
        *eq = memcmp(a, b, (size_t)n * 8) == 0;
        return SSDE_OK;
    };
    for (auto& b : h->slots) {
        if (b.col < 0 || b.par_j != h->d + 1 || b.basis_c >= 0) continue;      // (materialised basis blocks live on the device whatever the caller's arrays: left alone)
        for (auto& a : h->slots) {
            if (a.col < 0 || a.par_j != h->d || a.basis_c >= 0) continue;
            bool eq;
            int st = equal(a.src, b.src, &eq);
            if (st) return st;
            if (eq) { b.col = a.col; break; }
        }
    }
    std::vector<int> remap(h->n_stream_cols, -1);
    int nd = 0;
    for (auto& sl : h->slots)
        if (sl.col >= 0) { if (remap[sl.col] < 0) remap[sl.col] = nd++; sl.col = remap[sl.col]; }
    *n_distinct = nd;
    (void)d;
    return SSDE_OK;
}

static int build_impl(const ssde_desc* d, ssde_handle* h, const ParLayout* part_layout, bool allow_drift) {
    { int st = check_descriptor(d, h, part_layout); if (st) return st; }

    // ---- device ------------------------------------------------------------------------------
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0)
        return fail(h, SSDE_ERR_NODEVICE, "no HIP device visible: this engine has no CPU fallback");
    if (d->device >= 0) HIPCHK(h, hipSetDevice(d->device));
    HIPCHK(h, hipGetDevice(&h->device));
    hipDeviceProp_t prop;
    HIPCHK(h, hipGetDeviceProperties(&prop, h->device));
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0)
        return fail(h, SSDE_ERR_NODEVICE, std::string("device is ") + prop.gcnArchName + ", this library holds gfx950 code only");
    const bool on_dev = (d->flags & SSDE_FLAG_DEVICE_DATA) != 0;
    const int64_t n = d->n;
    h->n_stream_cols_algo = h->n_stream_cols;

    { int st = stage_basis_tables(d, h, on_dev, n); if (st) return st; }
    std::vector<int64_t> starts;
    { int st = find_segments(d, h, on_dev, n, starts); if (st) return st; }

    // ---- ESEAL_SSM: the lane = direction kernels with the scalar lipid-mass lanes ----------------------------
    if (is_eseal(d->model)) {
        for (int i = 0; i < 2; i++)
            for (int j = 0; j < 2; j++) h->p0_full[i + j * 2] = p0_entry(d, i, j);
        h->tv_dense = true;
        int st = build_tv(d, h, starts, on_dev);
        if (st) return st;
    } else
    // ---- direct families --------------------------------------------------------------------------
    if (!is_kalman(d->model)) {
        int st = build_direct(d, h, on_dev, n);
        if (st) return st;
    } else {
        // ---- Kalman families: pick the path, then tile ------------------------------------------------
        for (int i = 0; i < h->sdim; i++)
            for (int j = 0; j < h->sdim; j++) h->p0_full[i + j * h->sdim] = p0_entry(d, i, j);
        // (force_tv: the companion of SSDE_FLAG_EXACT_HESS -- the lane = direction path whatever the design, constant coefficients
        //  included: their intercepts are directions like any other there)
        const bool iso_ok = h->d <= 2 && !h->has_h && h->const_coeff && p0_is_isotropic(d, h->p0_iso) &&
                            !(d->flags & SSDE_FLAG_FORCE_DENSE) && !h->force_tv;
        h->path = iso_ok ? PATH_ISO : PATH_DENSE;
        // Row-varying DRIFT only (design columns in the rows of mu_1 .. mu_d, everything else constant), many tracks: the
        // register path with the design columns streamed next to the observations (k_iso_drift.hip) -- on a regular grid with
        // complete tracks (which the tiling pass below finds out) the covariance half is as data-independent as with constant
        // coefficients and the shared-covariance lanes run; otherwise the lanes carry their own covariance.  Few tracks (C1:
        // one animal) stay on the lane = direction path, whose windows cut ONE track into a hundred concurrent pieces.
        // (a per-row H_array couples the dimensions: CTCRW with d = 2 has 4 x 4 covariance lanes in k_iso_colvar.hip, any P0)
        const bool iso_cfg = !h->has_h && p0_is_isotropic(d, h->p0_iso);
        const bool full_cfg = h->has_h && h->d == 2 && !getenv("SSDE_NO_COLVAR_FULL");
        // (one response column: H_array[,,i] is the row's measurement variance -- the isotropic lanes of k_iso_colvar.hip with h = H_i)
        const bool h1_cfg = h->has_h && h->d == 1 && p0_is_isotropic(d, h->p0_iso) && !getenv("SSDE_NO_COLVAR_FULL");
        // (... also with CONSTANT coefficients: tracks with error ellipses and one tau, one nu -- the intercepts are columns of ones)
        if (!iso_ok && h->d <= 2 && allow_drift && (iso_cfg || full_cfg || h1_cfg) && (!h->const_coeff || full_cfg || h1_cfg) && !(d->flags & SSDE_FLAG_FORCE_DENSE) &&
            !getenv("SSDE_NO_DRIFT") && !h->force_tv) {
            bool mu_only = true;
            for (auto& sl : h->slots)
                if (sl.col >= 0 && sl.par_j >= h->d) mu_only = false;
            // Which side wins is a matter of the batch's ROWS, not of its tracks (tools/sweep_dispatch.py, profiles/r04_d_dispatch_sweep*.txt:
            // the lane = direction path costs ~0.07 ms + 0.2 us per 1000 rows, the register lanes ~0.09 ms (0.18 for CTCRW: its transient
            // window) + 0.012 us per 1000 rows -- round 3's rule, >= 32 tracks, was wrong by 17-117 % at 32-64 tracks x 10^3 rows).
            // SSDE_DRIFT_MIN_TRACKS keeps its meaning for the tests: a track count decides.
            int min_tracks = 32;
            const bool by_tracks = getenv("SSDE_DRIFT_MIN_TRACKS") != nullptr;
            if (by_tracks) min_tracks = atoi(getenv("SSDE_DRIFT_MIN_TRACKS"));
            const double n_rows = (double)n;
            const bool drift_pays = by_tracks ? h->n_seg >= min_tracks : n_rows >= (h->model == SSDE_MODEL_CTCRW ? 5e5 : 1.5e5);
            if (iso_cfg && mu_only && drift_pays && h->n_stream_cols <= DRIFT_KMAX) {
                h->drift = 1; h->path = PATH_ISO;
                // Every streamed column belongs to a block given as a FUNCTION of a covariate (basis_re): the tiles carry the covariate, the lanes
                // evaluate the block from its table in LDS (k_iso_drift_pp.hip): 8 B/row per block where the columns cost 8 K.
                // Measured (tools/bench_drift.py ... table, 10^4 x 10^4, K = 9, two waves per SIMD): the lanes' per-row table reads -- 2 K
                // 16-byte LDS reads at per-lane addresses -- bind, not HBM: BM_SSM 1.13 against 1.27 ms streamed, OU_SSM d = 1 1.24 / 1.29,
                // d = 2 1.54 / 1.49; CTCRW (AGPR spills on top) 2.3 / 1.44 and the general lanes 1.99 / 1.48 are slower and keep the
                // streamed form (SSDE_DRIFT_PP_ALL=1: everything that qualifies, for the tests).  What the table form always wins is
                // HBM: tiles of 8 (d + blocks) bytes per row instead of 8 (d + K).
                const bool pp_all = getenv("SSDE_DRIFT_PP_ALL") != nullptr;
                bool pp = !getenv("SSDE_NO_DRIFT_PP") && !h->no_drift_pp && (h->model != SSDE_MODEL_CTCRW || pp_all);
                int nb = 0, blk_j[2] = {-1, -1};
                for (auto& sl : h->slots) {
                    if (sl.col < 0) continue;
                    if (sl.basis_c < 0) { pp = false; break; }
                    if (nb == 0 || blk_j[nb - 1] != sl.par_j) { if (nb == 2) { pp = false; break; } blk_j[nb++] = sl.par_j; }
                }
                const int ng = (h->n_stream_cols + 3) / 4;
                // (the instantiations the compiler spills to scratch: the streamed form keeps those shapes)
                if (h->model == SSDE_MODEL_CTCRW && ((h->d == 2 && ng >= 4) || ng >= 6)) pp = false;
                PpDrift P;
                memset(&P, 0, sizeof(P));
                if (pp && nb > 0) {
                    int k0 = 0;
                    for (int b = 0; b < nb; b++) {
                        const int j = blk_j[b];
                        const PPRef& R = h->pp[j];
                        const int K = h->L.ncol_re[j];
                        if (!R.x || (R.nk - 1) * (K * 4 + 2) + R.nk > PPD_LDS) { pp = false; break; }
                        P.kcols[b] = K; P.k0[b] = k0; P.nk[b] = R.nk; P.uniform[b] = R.uniform; P.x0[b] = R.k0; P.inv_h[b] = R.inv_h;
                        P.tab[b] = R.tab; P.knots[b] = R.knots;
                        k0 += K;
                    }
                    if (pp && k0 != h->n_stream_cols) pp = false;
                    if (pp) { P.nb = nb; h->pp_drift = P; h->pp_drift_j[0] = blk_j[0]; h->pp_drift_j[1] = blk_j[1]; }
                }
            }
            // Row-varying tau / nu (kappa, sigma) with a constant drift, many tracks: lane = track lanes that carry one filter
            // tangent per design column (k_iso_colvar.hip) -- the lane = direction path below costs a wave-row per track-row
            // whatever the batch.
            // (a smooth drift alone, H = sigma_obs^2 I, took the drift kernels above; with H_array, or next to columns of tau / nu, the
            //  drift's design columns are columns of kinds of their own here)
            bool par_only = (!(mu_only && iso_cfg) || h->const_coeff) && !getenv("SSDE_NO_COLVAR");
            if (h->const_coeff && !(full_cfg || h1_cfg)) par_only = false;
            for (auto& sl : h->slots)
                if (sl.col >= 0 && sl.basis_c >= 0 && !sl.src) par_only = false;      // (a basis block materialised at create is a block of columns)
            if (getenv("SSDE_CV_NO_MU_COLS"))
                for (auto& sl : h->slots)
                    if (sl.col >= 0 && sl.par_j < h->d) par_only = false;
            // (measured, tools/bench_colvar.py --tracks M --rows 1000, 18 columns: 0.18 / 0.19 / 0.20 / 0.22 / 0.24 ms at M = 32 / 128 /
            //  256 / 512 / 1024 against 0.11 / 0.18 / 0.22 / 0.32 / 0.50 on the lane = direction path: the crossover is near 128 tracks)
            //  The lane = direction path costs in proportion to tracks x directions, this one does not depend on the directions: the
            //  crossover moves with their number -- 3 directions (tau ~ 1 + x): 0.094 / 0.106 / 0.119 / 0.25 ms at 32 / 128 / 256 / 1024
            //  tracks there, 0.26-0.28 ms on the one-wave kernel for few columns (iso_few_kernel), which wins from ~1200 tracks on.
            int n_dirs = h->fixed[0] ? 0 : 1, n_tan = 0, n_mu_cols = 0;
            for (auto& sl : h->slots)
                if (!h->fixed[sl.pidx]) { n_dirs++; if (sl.par_j >= h->d) n_tan++; else if (sl.col >= 0) n_mu_cols++; }
            const bool few_shape = iso_cfg && n_mu_cols == 0 && n_tan <= CV_KC * ((h->model == SSDE_MODEL_CTCRW && h->d == 2) ? 1 : 2) && h->n_stream_cols >= 1 && h->n_stream_cols <= 2 * CV_FEW_K &&
                                   !getenv("SSDE_CV_NO_FEW");                   // (columns may still be shared below: checked again at the plan)
            //  (With H_array the lane = direction lanes carry a full covariance and cost three times as much: 64 / 640 / 1280 tracks x 10^3 rows,
            //   constant tau / nu: 0.245 / 0.30 / 0.53 ms there against 0.23-0.24 on iso_full_kernel; row-varying, 160 tracks: 0.60 against 0.36.)
            // (round 4, the same sweep: rows decide here too.  The lane = direction path costs ~0.1 ms + 0.35 us per 1000 rows and 32 lanes per
            //  track, the eight-wave pipeline ~0.175 ms + 0.11 us per 1000 rows whatever the directions: it pays from rows x lanes-per-track
            //  ~ 4.5 10^6 on (160 tracks x 10^3 rows and 32 lanes: 0.19 against 0.22 ms; 64 tracks x 10^4 rows: 0.24 against 0.37 ms, which round 3's track rule sent the other way); iso_few_kernel from
            //  1.5 10^6 rows on (CTCRW, d = 2; 5 10^5 for the scalar-covariance models: 1000 tracks x 10^4 rows 0.41 against 1.32 ms);
            //  with H_array the lane = direction lanes cost 55 ms per 10^6 rows: iso_full_kernel / the full-covariance pipeline from 8 tracks on.)
            int lpt_dirs = 1;
            while (lpt_dirs < n_dirs && lpt_dirs < 64) lpt_dirs *= 2;
            const bool cv_pays = by_tracks ? h->n_seg >= min_tracks
                               : h->has_h ? h->n_seg >= 8
                               : few_shape ? n_rows >= ((h->model == SSDE_MODEL_CTCRW && h->d == 2) ? 1.5e6 : 5e5)
                               : n_rows * lpt_dirs >= 4.5e6;
            if (par_only && cv_pays && h->d <= 2 && h->n_stream_cols <= 2 * DRIFT_KMAX) {
                int nd = h->n_stream_cols;
                if (!getenv("SSDE_CV_NO_SHARE")) { int st = share_equal_columns(d, h, on_dev, n, &nd); if (st) return st; }
                if (nd <= DRIFT_KMAX) { h->n_stream_cols = nd; h->drift = 3; h->path = PATH_ISO; h->cv_full = full_cfg; h->cv_single = h->cv_full && nd == 0 && !getenv("SSDE_CV_NO_SINGLE"); }
                else return SSDE_RETRY_WITHOUT_DRIFT;              // (the slots were renumbered: start over)
            }
        }
        // row-varying coefficients with H = sigma_obs^2 I and a block-identical P0: the tv path
        // everything the constant-coefficient register path does not take: row-varying coefficients (isotropic
        // lanes), per-row H_array or a P0 that is not block-identical (full-covariance lanes)
        const bool tv_ok = !iso_ok && h->d <= 2 && !h->drift && !(d->flags & SSDE_FLAG_FORCE_DENSE) && !getenv("SSDE_NO_TV") &&
                           (double)n * (TV_RS + 64) * 8.0 < 150e9;
        if (tv_ok) {
            h->tv_dense = h->has_h || !p0_is_isotropic(d, h->p0_iso);
            int st = build_tv(d, h, starts, on_dev);
            if (st) return st;
        }
        if (h->path != PATH_TV) {

        // ---- what gets tiled: the caller's rows, or -- lattice_pad() -- the same tracks on their regular lattice -------------
        // A time grid whose intervals are small whole multiples of one step (a regular schedule with fixes MISSING FROM THE DATA,
        // not NA-padded) would run the irregular-grid kernel: a per-lane transition and an exp per row, one wave per SIMD
        // (1.1 ms per 1e8 rows).  The transition over k steps is the k-fold product of the one-step transition
        // (makeT/makeQ/makeB are the exact discretisation: nllk_ctcrw.hpp:45-91, nllk_ou_ssm.hpp:35-66, nllk_bm_ssm.hpp:33),
        // and a row whose observation is missing is exactly one prediction step (nllk_ctcrw.hpp:214-228): so the tracks are
        // laid out on the lattice with NA rows where fixes are absent, and run as a regular grid with missing rows (hoisted
        // transition, two waves per SIMD, 0.75 ms per 1e8 lattice rows; groups without a gap take the shared-covariance path).
        int64_t tn = n;
        std::vector<int64_t> tstarts_pad;
        DevBuf<double> pad_times, pad_obs;
        const double *t_times = d->times, *t_obs = d->obs;
        bool t_on_dev = on_dev;
        if (h->path == PATH_ISO && !h->drift && !(d->flags & SSDE_FLAG_NO_UNIFORM_DT) && !getenv("SSDE_NO_LATTICE")) {
            int st = lattice_pad(d, h, starts, on_dev, tstarts_pad, pad_times, pad_obs);
            if (st) return st;
            if (h->n_pad > 0) { tn = h->n_pad; t_times = pad_times.p; t_obs = pad_obs.p; t_on_dev = true; }
        }
        const std::vector<int64_t>& tstarts = h->n_pad > 0 ? tstarts_pad : starts;

        // tracks -> lanes, 64 per wavefront: tracks WITHOUT a missing row first, longest first within each class (stable).
        // A wavefront whose 64 tracks have every row runs the shared-covariance kernel, one missing row anywhere in it
        // sends all 64 to the general kernel (3x the time): dealing the tracks that have missing rows (or, on a lattice
        // layout, absent fixes) to wavefronts of their own keeps everybody else on the fast path.
        const int64_t M = h->n_seg;
        // the observations on the device (host data: staged here, freed again after tiling)
        DevBuf<double> s_obs;
        const double* p_obs = t_obs;
        if (!t_on_dev) { HIPCHK(h, stage(t_obs, (size_t)tn * d->n_dim, false, s_obs)); p_obs = s_obs.p; }
        std::vector<uint8_t> seg_dirty((size_t)M, 0);
        std::vector<int> seg_napos;
        if (h->path == PATH_ISO && !h->drift && !getenv("SSDE_NO_REGROUP")) {
            DevBuf<int64_t> sd;
            DevBuf<int> fl;
            HIPCHK(h, sd.upload(tstarts));
            HIPCHK(h, fl.alloc((size_t)M));
            HIPCHK(h, hipMemset(fl.p, 0, (size_t)M * sizeof(int)));
            HIPCHK(h, launch_seg_nan(p_obs, tn, d->n_dim, sd.p, M, fl.p, 0));
            std::vector<int> flh((size_t)M);
            HIPCHK(h, hipMemcpy(flh.data(), fl.p, (size_t)M * sizeof(int), hipMemcpyDeviceToHost));
            for (int64_t k = 0; k < M; k++) seg_dirty[k] = flh[k] != 0;
            sd.release(); fl.release();
            if (!getenv("SSDE_NO_NA_SORT")) seg_napos = flh;    // 1 + the track's last row with a missing observation
        }
        std::vector<int64_t> order(M);
        std::iota(order.begin(), order.end(), 0);
        std::stable_sort(order.begin(), order.end(), [&](int64_t a, int64_t b) {
            if (seg_dirty[a] != seg_dirty[b]) return seg_dirty[a] < seg_dirty[b];
            const int64_t la = tstarts[a + 1] - tstarts[a], lb = tstarts[b + 1] - tstarts[b];
            if (la != lb) return la > lb;
            // tracks of one length with missing rows: neighbours by WHERE they miss them share a wavefront, so that its lanes
            // leave the stationary regime together (quiet rows of the general kernel, k_iso.hip)
            if (!seg_napos.empty() && seg_dirty[a]) return seg_napos[a] < seg_napos[b];
            return false;
        });
        h->n_groups = (int)((M + WAVE - 1) / WAVE);
        const int G = h->n_groups;
        // stage the time stamps first: a GLOBALLY regular grid (every consecutive pair of rows, track boundaries
        // included, is dt apart) needs no dt channel in the tiles -- nobody would read it, and a stream with holes
        // costs HBM efficiency (2 of 3 channels read: 5.6 TB/s; contiguous: > 7 TB/s)
        DevBuf<double> s_times;
        const double* p_times = t_times;
        if (!t_on_dev) { HIPCHK(h, stage(t_times, (size_t)tn, false, s_times)); p_times = s_times.p; }
        h->c_obs = 1;
        if (!(d->flags & SSDE_FLAG_NO_UNIFORM_DT) && !getenv("SSDE_KEEP_DT_CHANNEL")) {
            const int nb = 1024;
            DevBuf<double> mm;
            HIPCHK(h, mm.alloc((size_t)nb * 2));
            HIPCHK(h, launch_dt_minmax(p_times, nullptr, tn, mm.p, nb, 0));
            std::vector<double> mmh((size_t)nb * 2);
            HIPCHK(h, hipMemcpy(mmh.data(), mm.p, mmh.size() * 8, hipMemcpyDeviceToHost));
            double lo = INFINITY, hi = -INFINITY;
            for (int b = 0; b < nb; b++) { lo = std::min(lo, mmh[2 * b]); hi = std::max(hi, mmh[2 * b + 1]); }
            mm.release();
            double rtol = 1e-12;                                    // (regular to the last bits counts as regular: lattice_pad)
            if (const char* e = getenv("SSDE_GRID_RTOL")) rtol = std::max(0.0, atof(e));
            if (std::isfinite(lo) && lo > 0.0 && hi <= lo * (1.0 + rtol)) { h->c_obs = 0; h->dt_all = 0.5 * (lo + hi); }
        }
        // (a smooth drift evaluated from tables: one channel per block -- its covariate -- instead of one per column)
        const int n_tile_cols = h->pp_drift.nb > 0 ? h->pp_drift.nb : h->n_stream_cols;
        h->C = h->c_obs + d->n_dim + (h->has_h ? d->n_dim * d->n_dim : 0) + n_tile_cols;
        std::vector<int64_t> lane_row0((size_t)G * WAVE, -1), lane_seg((size_t)G * WAVE, 0), goff(G);
        std::vector<int32_t> lane_ns((size_t)G * WAVE, 0), glen(G);
        int64_t off = 0;
        for (int g = 0; g < G; g++) {
            int32_t mx = 0;
            for (int l = 0; l < WAVE; l++) {
                int64_t t = (int64_t)g * WAVE + l;
                if (t >= M) break;
                int64_t seg = order[t];
                int64_t len = tstarts[seg + 1] - tstarts[seg];
                if (len - 1 > INT32_MAX) return fail(h, SSDE_ERR_ARG, "track too long");
                lane_row0[t] = tstarts[seg];
                lane_seg[t] = seg;
                lane_ns[t] = (int32_t)(len - 1);
                mx = std::max(mx, lane_ns[t]);
            }
            glen[g] = (mx + TILE_U - 1) / TILE_U * TILE_U;
            goff[g] = off;
            off += (int64_t)glen[g] * h->C * WAVE;
        }
        h->tile_doubles = off + (int64_t)TILE_SPARE * h->C * WAVE;  // spare rows for prefetching ahead
        HIPCHK(h, h->tiles.alloc((size_t)h->tile_doubles));
        HIPCHK(h, hipMemset(h->tiles.p, 0, (size_t)h->tile_doubles * 8));
        HIPCHK(h, h->a0.alloc((size_t)G * h->sdim * WAVE));
        HIPCHK(h, h->group_off.upload(goff));
        HIPCHK(h, h->group_len.upload(glen));
        HIPCHK(h, h->lane_row0.upload(lane_row0));
        HIPCHK(h, h->lane_nsteps.upload(lane_ns));
        h->glen_host = glen; h->lane_ns_host = lane_ns;

        // stage the caller's arrays (host data) -- freed again after tiling
        DevBuf<double> s_h, s_a0, s_cols;
        DevBuf<const double*> s_colptr;
        DevBuf<int64_t> s_lane_seg;
        const double* p_h = d->h_array;
        if (!t_on_dev) {
            if (h->has_h) { HIPCHK(h, stage(d->h_array, (size_t)tn * d->n_dim * d->n_dim, false, s_h)); p_h = s_h.p; }
        }
        std::vector<const double*> cp(n_tile_cols, nullptr);
        if (h->pp_drift.nb > 0) {
            for (int b = 0; b < h->pp_drift.nb; b++) cp[b] = h->pp[h->pp_drift_j[b]].x;      // (engine-owned copies in HBM: stage_basis_tables)
            HIPCHK(h, s_colptr.upload(cp));
        } else
        if (h->n_stream_cols > 0) {
            if (!t_on_dev) HIPCHK(h, s_cols.alloc((size_t)tn * h->n_stream_cols));
            for (auto& s : h->slots)
                if (s.col >= 0) {
                    if (t_on_dev) cp[s.col] = s.src;
                    else {
                        double* dst = s_cols.p + (size_t)s.col * tn;
                        HIPCHK(h, hipMemcpy(dst, s.src, (size_t)tn * 8, hipMemcpyDefault));   // host array or materialised basis block
                        cp[s.col] = dst;
                    }
                }
            HIPCHK(h, s_colptr.upload(cp));
        }
        const double* p_a0 = nullptr;
        if (d->a0) {
            // a0 is tiny (n_seg x sdim): always treated as a host array
            HIPCHK(h, stage(d->a0, (size_t)h->n_seg * h->sdim, false, s_a0));
            p_a0 = s_a0.p;
            HIPCHK(h, s_lane_seg.upload(lane_seg));
        }
        const int ych = ingest_ychunks(G);
        DevBuf<double> mm;
        HIPCHK(h, mm.alloc((size_t)G * ych * 3));
        IngestArgs ia;
        ia.times = p_times; ia.obs = p_obs; ia.h_array = h->has_h ? p_h : nullptr;
        ia.cols = s_colptr.p; ia.ncols = n_tile_cols; ia.d = d->n_dim; ia.n = tn;
        ia.lane_row0 = h->lane_row0.p; ia.lane_nsteps = h->lane_nsteps.p;
        ia.group_off = h->group_off.p; ia.group_len = h->group_len.p;
        ia.n_groups = G; ia.C = h->C; ia.c_obs = h->c_obs; ia.tiles = h->tiles.p; ia.a0 = h->a0.p;
        ia.a0_src = p_a0; ia.lane_seg = s_lane_seg.p; ia.n_seg = h->n_seg;
        ia.sdim = h->sdim; ia.model = d->model; ia.dt_minmax = mm.p; ia.ychunks = ych; ia.last_dt = h->last_dt;
        HIPCHK(h, launch_ingest(ia, 0));
        std::vector<double> mmh((size_t)G * ych * 3);
        HIPCHK(h, hipMemcpy(mmh.data(), mm.p, mmh.size() * 8, hipMemcpyDeviceToHost));  // also syncs
        double dmin = INFINITY, dmax = -INFINITY;
        std::vector<int32_t> gflags(G, 1);
        for (size_t k = 0; k < mmh.size(); k += 3) {
            dmin = std::min(dmin, mmh[k]); dmax = std::max(dmax, mmh[k + 1]);
            if (mmh[k + 2] != 0.0) gflags[(k / 3) / ych] = 0;   // a NaN observation somewhere in the group
        }
        h->uniform_dt = (dmin == dmax) && std::isfinite(dmin) && !(d->flags & SSDE_FLAG_NO_UNIFORM_DT);
        if (h->n_pad > 0) {
            // the lattice was verified to SSDE_GRID_RTOL of its step: every scored interval IS the step (an interpolated
            // time stamp may differ from it in the last bits)
            h->uniform_dt = true; dmin = h->pad_step;
        } else if (h->snap_dt > 0.0) {
            h->uniform_dt = true; dmin = h->snap_dt;
        }
        h->dt_uniform = h->uniform_dt ? dmin : 0.0;
        h->dt_min = std::isfinite(dmin) ? dmin : 0.0;
        h->dt_max = std::isfinite(dmax) ? dmax : 0.0;
        mm.release(); s_times.release(); s_obs.release(); s_h.release(); s_a0.release(); s_cols.release();
        s_colptr.release(); s_lane_seg.release();
        pad_times.release(); pad_obs.release();
        h->hbm_bytes = h->tile_doubles * 8;

        if (h->path == PATH_ISO) {
            const int st = plan_register_path(h, G, gflags, lane_ns, glen);
            if (st) return st;
            if (h->pp_drift.nb > 0)                                 // the tiles hold the covariates: neither the materialised blocks nor the copies are read again
                for (int b = 0; b < h->pp_drift.nb; b++) { h->pp_mat[h->pp_drift_j[b]].release(); h->pp_x[h->pp_drift_j[b]].release(); h->pp[h->pp_drift_j[b]].x = nullptr; }
        } else {
            // gradient directions of the dense kernel: free parameters that reach the data term
            std::vector<DenseDir> dirs;
            if (!h->fixed[0]) dirs.push_back({1, 0, 0, 0});
            for (size_t k = 0; k < h->slots.size(); k++)
                if (!h->fixed[h->slots[k].pidx]) dirs.push_back({2, (int16_t)k, (int16_t)h->slots[k].pidx, 0});
            while (dirs.size() % DENSE_NT) dirs.push_back({0, 0, -1, 0});
            if (dirs.empty()) dirs.resize(DENSE_NT, DenseDir{0, 0, -1, 0});
            h->dirs_host = dirs;
            h->n_dirblocks = (int)dirs.size() / DENSE_NT;
            HIPCHK(h, h->dirs.upload(dirs));
            h->partial_doubles = (size_t)h->n_dirblocks * (1 + DENSE_NT) * G;
        }
        }  // path != PATH_TV
    }

    if (h->path == PATH_ISO && h->use_shared) {
        for (int i = 0; i < 2; i++) {
            HIPCHK(h, hipStreamCreateWithFlags(&h->aux[i], hipStreamNonBlocking));
            HIPCHK(h, hipEventCreateWithFlags(&h->ev_join[i], hipEventDisableTiming));
        }
        HIPCHK(h, hipEventCreateWithFlags(&h->ev_fork, hipEventDisableTiming));
        for (int i = 0; i < PAR_RING; i++) HIPCHK(h, hipEventCreateWithFlags(&h->par_ev[i], hipEventDisableTiming));
        h->par_ev_ok = true;
    }
    if (h->path != PATH_ISO) {
        SlotTable st;
        memset(&st, 0, sizeof(st));
        st.n_slots = (int)h->slots.size();
        st.q = h->q;
        for (size_t k = 0; k < h->slots.size(); k++) {
            st.par_j[k] = (int16_t)h->slots[k].par_j;
            st.col[k] = (int16_t)h->slots[k].col;
            st.pidx[k] = (int16_t)h->slots[k].pidx;
            st.is_free[k] = h->fixed[h->slots[k].pidx] ? 0 : 1;
            st.decay[k] = (int16_t)h->slots[k].decay;
        }
        if (!h->slot_table.p) HIPCHK(h, h->slot_table.upload(std::vector<SlotTable>(1, st)));
        HIPCHK(h, h->par_ring.alloc((size_t)PAR_RING * MAX_PAR));
        HIPCHK(h, hipHostMalloc((void**)&h->par_pinned, (size_t)PAR_RING * MAX_PAR * 8, hipHostMallocDefault));
        for (int i = 0; i < PAR_RING; i++) HIPCHK(h, hipEventCreateWithFlags(&h->par_ev[i], hipEventDisableTiming));
        h->par_ev_ok = true;
    }
    HIPCHK(h, h->partials.alloc(h->partial_doubles));
    h->hbm_bytes += (int64_t)h->partial_doubles * 8;
    // evaluations run on non-blocking streams, which the null stream's work above does not order itself against
    HIPCHK(h, hipEventCreateWithFlags(&h->ev_async, hipEventDisableTiming));
    HIPCHK(h, hipDeviceSynchronize());
    return SSDE_OK;
}
}  // namespace ssde_engine
