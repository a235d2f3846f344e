// ssde_engine_tv.hip -- engine side of the lane = gradient direction path (k_tv.hip, k_tv_dense.hip): row-varying
// coefficients, per-row H_array / general P0, ESEAL_SSM.  Data staging, gradient directions, window planning from the
// pre-pass statistics, evaluation (plain and replayed from a hipGraph).
#include "ssde_engine.hpp"

namespace ssde_engine {

// fill the static part of the tv argument block from the handle
void tv_base_args(const ssde_handle* h, TvArgs& a) {
    memset(&a, 0, sizeof(a));
    a.times = h->times.p; a.obs = h->obs.p; a.colbuf = h->colbuf.p; a.col_stride = h->col_stride;
    a.scored = h->scored.p; a.n = h->n; a.d = h->d; a.model = h->model; a.any_nan = h->na_any;
    a.slots = h->slot_table.p; a.n_slots = (int)h->slots.size();
    a.rec = h->tv_rec.p; a.wdir = h->tv_wdir.p; a.ndp = h->tv_ndp; a.lpt_shift = h->tv_lpt_shift;
    a.dirs = h->tv_dirs.p; a.trk_row0 = h->tv_row0.p; a.trk_ns = h->tv_ns.p; a.a0 = h->tv_a0.p;
    a.n_tracks = h->n_seg;
    for (int i = 0; i < 3; i++) a.p0[i] = h->p0_iso[i];
    a.dense = h->tv_dense ? 1 : 0; a.has_h = h->has_h ? 1 : 0; a.h_array = h->tv_harr.p;
    for (int i = 0; i < 16; i++) a.p0f[i] = h->p0_full[i];
    a.eseal_h = h->tv_eh.p; a.eseal_R = h->tv_eR.p;
    a.bnd = h->tv_bnd.p; a.gval = h->tv_gval.p; a.gdir = h->tv_gdir.p;
    a.stats = h->tv_stats.p; a.stats_blocks = h->tv_stats_blocks;
    a.n_out = 1 + h->L.n_full;
    a.last_dt = h->last_dt;
    for (int k = 0; k < MAX_PAR; k++) a.dir_of_par[k] = h->tv_dir_of_par[k];
}

// Row-varying isotropic Kalman path (k_tv.hip): long-format copies of the data, the streamed design
// columns, one weight per (row, gradient direction), tracks sorted by length.
int build_tv(const ssde_desc* d, ssde_handle* h, const std::vector<int64_t>& starts, bool on_dev) {
    const int64_t n = d->n, M = h->n_seg;
    h->path = PATH_TV;
    h->max_chunks = 1 << 20;        // "1" means: forced to one sequential window (ssde_widen_windows / retries)
    HIPCHK(h, stage(d->times, (size_t)n, on_dev, h->times));
    HIPCHK(h, stage(d->obs, (size_t)n * d->n_dim, on_dev, h->obs));
    if (h->has_h) HIPCHK(h, stage(d->h_array, (size_t)n * d->n_dim * d->n_dim, on_dev, h->tv_harr));
    if (is_eseal(d->model)) {
        HIPCHK(h, stage(d->eseal_h, (size_t)n, on_dev, h->tv_eh));
        HIPCHK(h, stage(d->eseal_R, (size_t)n, on_dev, h->tv_eR));
        // priors (nllk_e_seal_ssm.hpp:212-216): n and the design weights of log sigma at the first row
        h->pen.eseal_n = n;
        for (auto& sl : h->slots)
            if (sl.par_j == 1) h->pen.eseal_sig0.push_back({sl.pidx, sl.col >= 0 ? sl.src[0] : 1.0});
    }
    h->col_stride = ((n + 63) / 64) * 64 + 160;
    HIPCHK(h, h->colbuf.alloc((size_t)h->col_stride * h->n_stream_cols));
    for (auto& sl : h->slots)
        if (sl.col >= 0)
            HIPCHK(h, hipMemcpy(h->colbuf.p + (size_t)sl.col * h->col_stride, sl.src, (size_t)n * 8,
                                hipMemcpyDefault));   // caller's array (host / HBM) or a materialised basis block
    {
        DevBuf<double> idb;
        const double* idp = d->id;
        if (!on_dev) { HIPCHK(h, stage(d->id, (size_t)n, false, idb)); idp = idb.p; }
        HIPCHK(h, h->scored.alloc((size_t)((n + 31) / 32)));
        HIPCHK(h, launch_scored_mask(idp, n, h->scored.p, 0));
        HIPCHK(h, hipDeviceSynchronize());
        idb.release();
    }
    // tracks, longest first (stable): the tracks of one wave ("pack") then have similar lengths
    std::vector<int64_t> order(M);
    std::iota(order.begin(), order.end(), 0);
    std::stable_sort(order.begin(), order.end(), [&](int64_t a, int64_t b) {
        return (starts[a + 1] - starts[a]) > (starts[b + 1] - starts[b]);
    });
    std::vector<int64_t> row0(M), seg(M);
    std::vector<int32_t> ns(M);
    for (int64_t t = 0; t < M; t++) {
        const int64_t sg = order[t], len = starts[sg + 1] - starts[sg];
        if (len - 1 > INT32_MAX) return fail(h, SSDE_ERR_ARG, "track too long");
        row0[t] = starts[sg]; seg[t] = sg; ns[t] = (int32_t)(len - 1);
    }
    HIPCHK(h, h->tv_row0.upload(row0));
    HIPCHK(h, h->tv_ns.upload(ns));
    h->tv_ns_host = ns;
    h->glen_max = M > 0 ? ns[0] : 0;
    // gradient directions: log_sigma_obs, then every free coefficient slot
    std::vector<TvDir> dirs;
    for (int k = 0; k < MAX_PAR; k++) h->tv_dir_of_par[k] = -1;
    if (!h->fixed[0]) { h->tv_dir_of_par[0] = (int16_t)dirs.size(); dirs.push_back({TVK_SIG, 0, 0, -1}); }   // log_sigma_obs / log_tau
    if (is_eseal(d->model)) {                                            // a1, log_a2 (nllk_e_seal_ssm.hpp:115-116)
        if (!h->fixed[1]) { h->tv_dir_of_par[1] = (int16_t)dirs.size(); dirs.push_back({TVK_A1, 0, 1, -1}); }
        if (!h->fixed[2]) { h->tv_dir_of_par[2] = (int16_t)dirs.size(); dirs.push_back({TVK_A2, 0, 2, -1}); }
    }
    for (size_t k = 0; k < h->slots.size(); k++) {
        const Slot& sl = h->slots[k];
        if (h->fixed[sl.pidx]) continue;
        TvDir t;
        t.kind = (int16_t)(sl.par_j < h->d ? TVK_MU : (sl.par_j == h->d ? TVK_P1 : TVK_P2));
        t.dim = (int16_t)(sl.par_j < h->d ? sl.par_j : 0);
        t.pidx = (int16_t)sl.pidx; t.slot = (int16_t)k;
        h->tv_dir_of_par[sl.pidx] = (int16_t)dirs.size();
        dirs.push_back(t);
    }
    h->tv_nd = (int)dirs.size();
    int shift = 0;
    while ((1 << shift) < h->tv_nd && shift < 6) shift++;
    h->tv_lpt_shift = shift;
    const int lpt = 1 << shift;
    h->tv_nb = std::max(1, (h->tv_nd + lpt - 1) / lpt);
    h->tv_ndp = h->tv_nb * lpt;
    dirs.resize(h->tv_ndp, TvDir{TVK_NONE, 0, -1, -1});
    HIPCHK(h, h->tv_dirs.upload(dirs));
    // slot table (also uploaded by the common tail; the weights kernel needs it now)
    {
        SlotTable st;
        memset(&st, 0, sizeof(st));
        st.n_slots = (int)h->slots.size(); st.q = h->q;
        for (size_t k = 0; k < h->slots.size(); k++) {
            st.par_j[k] = (int16_t)h->slots[k].par_j; st.col[k] = (int16_t)h->slots[k].col;
            st.pidx[k] = (int16_t)h->slots[k].pidx; st.is_free[k] = h->fixed[h->slots[k].pidx] ? 0 : 1;
        }
        HIPCHK(h, h->slot_table.upload(std::vector<SlotTable>(1, st)));
    }
    HIPCHK(h, h->tv_wdir.alloc((size_t)n * h->tv_ndp));
    HIPCHK(h, h->tv_rec.alloc((size_t)n * TV_RS));
    HIPCHK(h, h->tv_a0.alloc((size_t)M * h->sdim));
    h->tv_stats_blocks = (int)std::min<int64_t>((n + 255) / 256, 256);
    HIPCHK(h, h->tv_stats.alloc((size_t)h->tv_stats_blocks * TV_STATS));
    HIPCHK(h, hipHostMalloc((void**)&h->tv_stats_pinned, (size_t)h->tv_stats_blocks * TV_STATS * 8, hipHostMallocDefault));
    HIPCHK(h, hipEventCreateWithFlags(&h->tv_stats_ev, hipEventDisableTiming));
    HIPCHK(h, hipStreamCreateWithFlags(&h->tv_stream, hipStreamNonBlocking));
    HIPCHK(h, h->tv_par_dev.alloc(MAX_PAR));
    HIPCHK(h, hipHostMalloc((void**)&h->tv_par_pinned, MAX_PAR * 8, hipHostMallocDefault));
    HIPCHK(h, hipHostMalloc((void**)&h->tv_out_pinned, (MAX_PAR + 2) * 8, hipHostMallocDefault));
    HIPCHK(h, hipHostMalloc((void**)&h->tv_chk_pinned, (size_t)TV_LEAN_ITEMS * 8, hipHostMallocDefault));
    memset(h->tv_chk_pinned, 0, (size_t)TV_LEAN_ITEMS * 8);
    h->env_tv_no_lean = getenv("SSDE_TV_NO_LEAN") != nullptr;
    TvArgs a;
    tv_base_args(h, a);
    HIPCHK(h, launch_tv_weights(a, 0));
    {
        DevBuf<double> s_a0;
        DevBuf<int64_t> s_seg;
        if (d->a0) {
            HIPCHK(h, stage(d->a0, (size_t)h->n_seg * h->sdim, false, s_a0));   // a0 is tiny: always a host array
            HIPCHK(h, s_seg.upload(seg));
        }
        HIPCHK(h, launch_tv_a0(a, s_a0.p, s_seg.p, h->n_seg, h->sdim, h->tv_a0.p, 0));
        HIPCHK(h, hipDeviceSynchronize());
        s_a0.release(); s_seg.release();
    }
    h->hbm_bytes = (int64_t)(h->times.n + h->obs.n + h->colbuf.n + h->tv_wdir.n + h->tv_rec.n + h->tv_a0.n + h->tv_harr.n) * 8 +
                   (int64_t)h->scored.n * 4;
    return SSDE_OK;
}

// spectral radius of the stationary closed-loop matrix T - K Z for constant parameters (see plan_windows)
double closed_loop_rho(int model, double dt, double p1, double p2, double hobs, const double* p0) {
    if (!(dt > 0.0) || !std::isfinite(dt)) return 1.0;
    if (model == SSDE_MODEL_CTCRW) {
        const double tau = exp(p1), nu = exp(p2);
        CtcrwTrans tr;
        ctcrw_trans(dt, tau, 1.0 / tau, 2.0 * nu / sqrt(M_PI * tau), tr);
        double p11 = p0[0], p12 = p0[1], p22 = p0[2], k1 = 0, k2 = 0;
        for (int it = 0; it < 20000; it++) {
            const double F = p11 + hobs, iF = 1.0 / F;
            const double tp11 = p11 + tr.t12 * p12, tp12 = p12 + tr.t12 * p22, tp21 = tr.e * p12, tp22 = tr.e * p22;
            k1 = tp11 * iF; k2 = tp21 * iF;
            const double n11 = tp11 * (1.0 - k1) + tp12 * tr.t12 + tr.q11, n12 = -tp11 * k2 + tp12 * tr.e + tr.q12,
                         n22 = -tp21 * k2 + tp22 * tr.e + tr.q22;
            const double ch = std::fabs(n11 - p11) + std::fabs(n12 - p12) + std::fabs(n22 - p22);
            p11 = n11; p12 = n12; p22 = n22;
            if (ch <= 1e-15 * (std::fabs(p11) + std::fabs(p22))) break;
        }
        const double trc = (1.0 - k1) + tr.e, det = (1.0 - k1) * tr.e + k2 * tr.t12;
        const double disc = trc * trc - 4.0 * det;
        return disc >= 0.0 ? std::max(std::fabs(0.5 * (trc + std::sqrt(disc))), std::fabs(0.5 * (trc - std::sqrt(disc))))
                           : std::sqrt(std::fabs(det));
    }
    ScalTrans tr;
    if (model == SSDE_MODEL_OU_SSM) ou_trans(dt, exp(p1), exp(p2), tr);
    else bm_trans(dt, exp(p1), tr);
    double p = p0[0], k = 0;
    for (int it = 0; it < 20000; it++) {
        const double F = p + hobs, tp = tr.t * p;
        k = tp / F;
        const double np_ = tp * (tr.t - k) + tr.q;
        const double ch = std::fabs(np_ - p);
        p = np_;
        if (ch <= 1e-15 * std::fabs(p)) break;
    }
    return std::fabs(tr.t - k);
}

// Time windows of the tv path.  The warm-up length comes from the slowest-forgetting corner of the
// parameter ranges the LAST evaluation's pre-pass saw (dt, par[d], par[d+1]); the device-side
// hand-over check decides whether it was enough.  Rebuilds the work-item tables when the plan changes.
int tv_plan(ssde_handle* h, double hobs, hipStream_t s) {   // hobs: sigma_obs^2 (replaced by the largest diag(H) with H_array)
    int W = 0;
    if (h->max_chunks > 1 && h->tv_stats_valid && !is_eseal(h->model)) {   // ESEAL tracks: one sequential window
        double lo[4] = {INFINITY, INFINITY, INFINITY, INFINITY}, hi[4] = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
        for (int b = 0; b < h->tv_stats_blocks; b++)
            for (int k = 0; k < 4; k++) {
                lo[k] = std::min(lo[k], h->tv_stats_pinned[b * TV_STATS + 2 * k]);
                hi[k] = std::max(hi[k], h->tv_stats_pinned[b * TV_STATS + 2 * k + 1]);
            }
        double rho = 0.0;
        bool ok = std::isfinite(lo[0]) && std::isfinite(hi[0]) && std::isfinite(lo[1]) && std::isfinite(hi[1]) &&
                  std::isfinite(lo[2]) && std::isfinite(hi[2]);
        // per-row H_array: the largest observation variance forgets slowest
        // (the slowest mode sees the LARGEST EIGENVALUE of H_array[,,i], which the largest diagonal entry underestimates when the ellipse is
        //  tilted: lambda_max <= trace <= d x the largest diagonal entry.  Round 4: with the diagonal alone C1 with error ellipses sat at
        //  1e-12 of a 1e-11 tolerance, a tenth of its evaluations were retried, and 16-row windows tipped it over)
        if (h->has_h) { ok = ok && std::isfinite(hi[3]) && hi[3] > 0.0; hobs = hi[3] * h->d; }
        const double p0d[3] = {h->p0_full[0], h->model == SSDE_MODEL_CTCRW ? h->p0_full[1] : 0.0,
                               h->model == SSDE_MODEL_CTCRW ? h->p0_full[1 + h->sdim] : 0.0};
        if (ok)
            for (int c = 0; c < 8; c++) {
                const double r = closed_loop_rho(h->model, (c & 1) ? hi[0] : lo[0], (c & 2) ? hi[1] : lo[1],
                                                 (c & 4) ? hi[2] : lo[2], hobs, p0d);
                rho = std::max(rho, std::isfinite(r) ? r : 1.0);
            }
        if (ok && rho < 0.9995) {
            int64_t w = (int64_t)std::ceil(std::log(1e-18) / std::log(std::max(rho, 1e-300))) + 16;
            // (full-covariance lanes: the isotropic estimate of rho is optimistic for them -- measured on C1 with error ellipses: checks of
            //  1e-12 .. 7e-12 against the 1e-11 tolerance, two failures in 400 evaluations and 32 evaluations on a four-fold plan after each)
            if (h->tv_dense) w += WIN_ALIGN;
            w = std::max<int64_t>(w, 16);
            if (h->env_window > 0) w = h->env_window;
            w *= h->window_boost;
            w = (w + WIN_ALIGN - 1) / WIN_ALIGN * WIN_ALIGN;
            if (2 * w <= h->glen_max) W = (int)w;
        }
    }
    // keep the current plan while it is at least as careful and not wastefully so
    if (h->tv_window >= 0 && h->tv_n_items_g > 0 && ((W == 0) == (h->tv_window == 0)) && W <= h->tv_window &&
        h->tv_window <= 2 * W + WIN_ALIGN)
        return SSDE_OK;
    const int tpw = WAVE >> h->tv_lpt_shift;
    const int64_t n_packs = (h->n_seg + tpw - 1) / tpw;
    int target = 2048;
    if (h->env_tv_waves > 0) target = h->env_tv_waves;
    const int nc_cap = (int)std::max<int64_t>(1, (target + n_packs * h->tv_nb - 1) / (n_packs * h->tv_nb));
    std::vector<TvItem> ig, iv;
    int max_nc = 1;
    for (int64_t p = 0; p < n_packs; p++) {
        const int L = h->tv_ns_host[(size_t)p * tpw];
        int nc = 1;
        // As many windows as the chip has room for (nc_cap): with idle SIMDs around, a window may be much
        // shorter than its warm-up -- the redundant warm-up rows run in parallel, the serial chain of a wave
        // is what the evaluation waits for.  SSDE_TV_MINLEN: shortest scored stretch of a window (rows).
        int minlen = h->tv_dense ? 2 * WIN_ALIGN : WIN_ALIGN;   // (round 4: 16 rows, not 32, on the isotropic lanes -- C1's filter launch 41.5 -> 33 us; the
                                                                  //  full-covariance lanes cost three times as much per warm-up row: 32)
        if (h->env_tv_minlen > 0) minlen = h->env_tv_minlen;
        if (W > 0 && L >= 2 * W) nc = std::max(1, std::min(nc_cap, (L + minlen - 1) / minlen));
        max_nc = std::max(max_nc, nc);
        for (int b = 0; b < h->tv_nb; b++)
            for (int c = 0; c < nc; c++) {
                ig.push_back({(int32_t)p, c, nc, b});
                if (b == 0) iv.push_back({(int32_t)p, c, nc, 0});
            }
    }
    // earlier evaluations may still be reading the old tables
    HIPCHK(h, hipStreamSynchronize(s));
    if (ig.size() > h->tv_items_cap) {
        h->tv_items_g.release(); h->tv_items_v.release(); h->tv_bnd.release(); h->tv_chk.release();
        h->tv_gval.release(); h->tv_gdir.release();
        h->tv_items_cap = ig.size() + ig.size() / 2;
        HIPCHK(h, h->tv_items_g.alloc(h->tv_items_cap));
        HIPCHK(h, h->tv_items_v.alloc(h->tv_items_cap));
        HIPCHK(h, h->tv_bnd.alloc(h->tv_items_cap * 2 * TV_NSTATE * WAVE));
        HIPCHK(h, h->tv_chk.alloc(h->tv_items_cap));
        HIPCHK(h, h->tv_gval.alloc(h->tv_items_cap * WAVE));
        HIPCHK(h, h->tv_gdir.alloc(h->tv_items_cap * WAVE));
    }
    HIPCHK(h, hipMemcpy(h->tv_items_g.p, ig.data(), ig.size() * sizeof(TvItem), hipMemcpyHostToDevice));
    HIPCHK(h, hipMemcpy(h->tv_items_v.p, iv.data(), iv.size() * sizeof(TvItem), hipMemcpyHostToDevice));
    h->tv_n_items_g = (int)ig.size(); h->tv_n_items_v = (int)iv.size();
    h->tv_window = W; h->tv_max_nc = max_nc;
    h->tv_plan_gen++;                                    // captured graphs of the old plan are stale
    return SSDE_OK;
}

int eval_tv(ssde_handle* h, const double* par, int order, double* out_dev, hipStream_t s) {
    const double* pdev = nullptr;
    int st = push_par(h, par, s, &pdev);
    if (st) return st;
    TvArgs a;
    tv_base_args(h, a);
    a.par = pdev;
    a.out = out_dev;                                     // the pre-pass zeroes the hand-over check slot
    const double sig = exp(par[0]);                      // nllk_ctcrw.hpp:136 (unused when H_array is supplied)
    a.h = sig * sig;
    const size_t stats_bytes = (size_t)h->tv_stats_blocks * TV_STATS * 8;
    bool prepared = false;
    if (!h->tv_stats_valid) {
        // first evaluation: the planner needs the parameter ranges of THIS parameter vector
        HIPCHK(h, launch_tv_prepare(a, s));
        HIPCHK(h, hipMemcpyAsync(h->tv_stats_pinned, h->tv_stats.p, stats_bytes, hipMemcpyDeviceToHost, s));
        HIPCHK(h, hipStreamSynchronize(s));
        h->tv_stats_valid = true;
        prepared = true;
    } else {
        HIPCHK(h, hipEventSynchronize(h->tv_stats_ev));  // ranges seen by the previous evaluation
    }
    st = tv_plan(h, a.h, s);
    if (st) return st;
    tv_base_args(h, a);                                  // the plan may have re-allocated the item buffers
    a.par = pdev; a.h = sig * sig; a.out = out_dev;
    a.h_from_par = 1;                                    // same arithmetic as the graph replay: bitwise-equal results
    a.window = h->tv_window;
    if (!prepared) {
        HIPCHK(h, launch_tv_prepare(a, s));
        HIPCHK(h, hipMemcpyAsync(h->tv_stats_pinned, h->tv_stats.p, stats_bytes, hipMemcpyDeviceToHost, s));
    }
    HIPCHK(h, hipEventRecord(h->tv_stats_ev, s));
    const bool grad = order >= 1;
    a.items = grad ? h->tv_items_g.p : h->tv_items_v.p;
    a.n_items = grad ? h->tv_n_items_g : h->tv_n_items_v;
    a.out = out_dev;
    if (h->stamps) HIPCHK(h, hipEventRecord(h->ev_k0, s));
    HIPCHK(h, launch_tv_filter(a, grad, s));
    h->last_kernel_id = (h->tv_dense || is_eseal(h->model)) ? SSDE_KERNEL_TV_DENSE : SSDE_KERNEL_TV;
    if (h->stamps) HIPCHK(h, hipEventRecord(h->ev_k1, s));
    h->ev_k_valid = h->stamps; h->last_s_stat = -1;
    HIPCHK(h, launch_tv_finalize(a, s));
    h->last_chunks = h->tv_max_nc; h->last_window = h->tv_window;
    return SSDE_OK;
}

// Synchronous tv evaluation replayed from a hipGraph: the evaluation is launch-bound (C1: 40 us of kernels,
// seven stream operations), so the whole sequence is captured once per plan and direction order.  Needs the
// parameter ranges of an earlier evaluation (the planner's input), so the first evaluation takes eval_tv.
int eval_tv_graph(ssde_handle* h, const double* par, int order, double* o_host) {
    const int n_full = h->L.n_full;
    const int ord = order >= 1 ? 1 : 0;
    const double sig = exp(par[0]);
    int st = tv_plan(h, sig * sig, h->tv_stream);        // from the statistics of the previous evaluation
    if (st) return st;
    // few rows (C1: one animal): every node of the graph costs ~4 us, the three copies as much as the kernels between them -- the
    // kernels read the parameters from, and write the statistics / sums / checks to, pinned host memory themselves (TvArgs.par0_w)
    const int n_items_now = ord ? h->tv_n_items_g : h->tv_n_items_v;
    const bool lean = !h->env_tv_no_lean && h->tv_chk_pinned && h->tv_stats_blocks <= TV_LEAN_BLOCKS && n_items_now <= TV_LEAN_ITEMS;
    if (!h->tv_gexec[ord] || h->tv_graph_plan[ord] != h->tv_plan_gen || h->tv_graph_lean[ord] != lean) {
        if (h->tv_gexec[ord]) { (void)hipGraphExecDestroy(h->tv_gexec[ord]); h->tv_gexec[ord] = nullptr; }
        TvArgs a;
        tv_base_args(h, a);
        a.par = h->tv_par_dev.p; a.h_from_par = 1; a.h = 0.0; a.out = h->out.p; a.window = h->tv_window;
        a.items = ord ? h->tv_items_g.p : h->tv_items_v.p;
        a.n_items = n_items_now;
        if (lean) {
            a.par = h->tv_par_pinned; a.par0_w = h->tv_par_dev.p; a.stats = h->tv_stats_pinned;
            a.out = h->tv_out_pinned; a.chk_items = h->tv_chk_pinned;
        }
        hipStream_t s = h->tv_stream;
        hipGraph_t g = nullptr;
        HIPCHK(h, hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
        hipError_t e = hipSuccess;
        if (!lean) e = hipMemcpyAsync(h->tv_par_dev.p, h->tv_par_pinned, (size_t)n_full * 8, hipMemcpyHostToDevice, s);
        if (e == hipSuccess) e = launch_tv_prepare(a, s);
        if (e == hipSuccess && !lean) e = hipMemcpyAsync(h->tv_stats_pinned, h->tv_stats.p, (size_t)h->tv_stats_blocks * TV_STATS * 8, hipMemcpyDeviceToHost, s);
        if (e == hipSuccess) e = launch_tv_filter(a, ord == 1, s);
        h->last_kernel_id = (h->tv_dense || is_eseal(h->model)) ? SSDE_KERNEL_TV_DENSE : SSDE_KERNEL_TV;
        if (e == hipSuccess) e = launch_tv_finalize(a, s);
        if (e == hipSuccess && !lean) e = hipMemcpyAsync(h->tv_out_pinned, h->out.p, (size_t)(2 + n_full) * 8, hipMemcpyDeviceToHost, s);
        hipError_t e2 = hipStreamEndCapture(s, &g);
        if (e != hipSuccess || e2 != hipSuccess) {
            if (g) (void)hipGraphDestroy(g);
            h->err = std::string("hipGraph capture of the tv evaluation failed: ") + hipGetErrorString(e != hipSuccess ? e : e2);
            return SSDE_ERR_HIP;
        }
        e = hipGraphInstantiate(&h->tv_gexec[ord], g, nullptr, nullptr, 0);
        (void)hipGraphDestroy(g);
        if (e != hipSuccess) { h->tv_gexec[ord] = nullptr; h->err = std::string("hipGraphInstantiate: ") + hipGetErrorString(e); return SSDE_ERR_HIP; }
        h->tv_graph_plan[ord] = h->tv_plan_gen;
        h->tv_graph_lean[ord] = lean;
    }
    memcpy(h->tv_par_pinned, par, (size_t)n_full * 8);
    HIPCHK(h, hipGraphLaunch(h->tv_gexec[ord], h->tv_stream));
    HIPCHK(h, hipStreamSynchronize(h->tv_stream));
    memcpy(o_host, h->tv_out_pinned, (size_t)(2 + n_full) * 8);
    if (lean) {                                          // the largest of the items' hand-over checks (NaN was stored as +inf)
        double w = 0.0;
        for (int i = 0; i < n_items_now; i++) w = std::max(w, h->tv_chk_pinned[i]);
        o_host[1 + n_full] = w;
    }
    h->ev_k_valid = false;                               // no per-kernel timing inside a replayed graph
    h->last_chunks = h->tv_max_nc; h->last_window = h->tv_window;
    return SSDE_OK;
}

}  // namespace ssde_engine
