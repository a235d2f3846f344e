// ssde_engine.hip -- C ABI (include/ssde.h) of the MI355X nllk engine: descriptor checks,
// segment discovery, one-off upload + re-tiling, per-evaluation launch + reduction, penalty.
//
// Replaces, for the nllk/gradient path only, what TMB's MakeADFunObject / EvalADFunObject do
// for the reference (/root/reference/src/init.c:6-8, R/sde.R:656-669, 694-697).
// There is no CPU evaluation path in this library: without a gfx950 device ssde_create fails.
#include "ssde_engine.hpp"

#include <chrono>
#include <limits>

namespace ssde_engine {
thread_local std::string g_create_error;
}  // namespace ssde_engine
using namespace ssde_engine;

namespace ssde_engine {
int fail(ssde_handle* h, int code, const std::string& msg) {
    if (h) h->err = msg;
    g_create_error = msg;
    return code;
}

}  // namespace ssde_engine

namespace ssde_engine {

// everything the handle holds on the device and in pinned memory (the handle itself stays)
static void release_device(ssde_handle* h) {
    if (!h->wave_clock_file.empty() && h->wave_clock.p && h->wave_clock_items > 0) {
        std::vector<double> w((size_t)4 * h->wave_clock_items);
        if (hipSetDevice(h->device) == hipSuccess && hipDeviceSynchronize() == hipSuccess &&
            hipMemcpy(w.data(), h->wave_clock.p, w.size() * 8, hipMemcpyDeviceToHost) == hipSuccess) {
            if (FILE* f = fopen(h->wave_clock_file.c_str(), "w")) {
                fprintf(f, "# work item, start, end (100 MHz ticks), HW_ID; windows %d, warm-up %d, t0 %d\n", h->last_chunks, h->last_window, h->last_t0);
                for (int i = 0; i < h->wave_clock_items; i++)
                    if (w[4 * (size_t)i + 3] != 0.0) fprintf(f, "%d %.0f %.0f %.0f\n", i, w[4 * (size_t)i], w[4 * (size_t)i + 1], w[4 * (size_t)i + 2]);
                fclose(f);
            }
        }
    }
    h->wave_clock.release();
    h->hs_partials.release(); h->hs_hess.release(); h->hs_i16.release();
    if (h->trace && h->trace_n > 0)
        fprintf(stderr, "[ssde trace] %lld isotropic evaluations, host us per evaluation: plan %.1f | gain table %.1f | main launch %.1f | "
                        "finalize launch %.1f | read-back (blocks until the GPU is done) %.1f\n", (long long)h->trace_n,
                h->trace_us[0] / h->trace_n, h->trace_us[1] / h->trace_n, h->trace_us[2] / h->trace_n, h->trace_us[3] / h->trace_n,
                h->trace_us[4] / h->trace_n);
    destroy_dist(h);
    h->bnd.release(); h->chk.release(); h->group_flags.release(); h->gain_ring.release(); h->pad_pos.release(); h->dirty_groups.release(); h->lap_out.release();
    if (h->gain_pinned) (void)hipHostFree(h->gain_pinned);
    for (int i = 0; i < 2; i++) { if (h->aux[i]) (void)hipStreamDestroy(h->aux[i]); if (h->ev_join[i]) (void)hipEventDestroy(h->ev_join[i]); }
    if (h->ev_fork) (void)hipEventDestroy(h->ev_fork);
    if (h->ev_async) (void)hipEventDestroy(h->ev_async);
    for (auto& pr : h->ev_ring) { if (pr[0]) (void)hipEventDestroy(pr[0]); if (pr[1]) (void)hipEventDestroy(pr[1]); }
    h->tv_eh.release(); h->tv_eR.release(); h->tv_harr.release(); h->tv_rec.release(); h->tv_wdir.release(); h->tv_a0.release(); h->tv_bnd.release(); h->tv_chk.release();
    h->tv_gval.release(); h->tv_gdir.release(); h->tv_stats.release(); h->tv_dirs.release(); h->tv_row0.release();
    h->tv_ns.release(); h->tv_items_g.release(); h->tv_items_v.release();
    for (int i = 0; i < 2; i++) if (h->tv_gexec[i]) (void)hipGraphExecDestroy(h->tv_gexec[i]);
    if (h->tv_stream) (void)hipStreamDestroy(h->tv_stream);
    h->tv_par_dev.release();
    if (h->tv_par_pinned) (void)hipHostFree(h->tv_par_pinned);
    if (h->tv_out_pinned) (void)hipHostFree(h->tv_out_pinned);
    if (h->tv_stats_pinned) (void)hipHostFree(h->tv_stats_pinned);
    if (h->tv_stats_ev) (void)hipEventDestroy(h->tv_stats_ev);
    h->tiles.release(); h->a0.release(); h->group_off.release(); h->lane_row0.release();
    h->group_len.release(); h->lane_nsteps.release();
    for (int j = 0; j < MAX_Q; j++) { h->pp_x[j].release(); h->pp_knots[j].release(); h->pp_tab[j].release(); h->pp_mat[j].release(); }
    h->tdecay.release(); h->times.release(); h->obs.release(); h->colbuf.release(); h->scored.release(); h->colptr.release();
    h->slot_table.release(); h->dirs.release(); h->par_ring.release();
    h->partials.release(); h->out.release();
    if (h->par_pinned) (void)hipHostFree(h->par_pinned);
    if (h->out_pinned) (void)hipHostFree(h->out_pinned);
    if (h->pub_pinned) (void)hipHostFree(h->pub_pinned);
    h->pub_count.release();
    if (h->par_ev_ok)
        for (int i = 0; i < PAR_RING; i++) (void)hipEventDestroy(h->par_ev[i]);
}

void destroy(ssde_handle* h) {
    if (!h) return;
    release_device(h);
    delete h;
}
}  // namespace ssde_engine

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
static int build_impl(const ssde_desc* d, ssde_handle* h, const ParLayout* part_layout, bool allow_drift);
int build(const ssde_desc* d, ssde_handle* h, const ParLayout* part_layout) {
    int st = build_impl(d, h, part_layout, true);
    if (st == SSDE_RETRY_WITHOUT_DRIFT) {
        // the row-varying-drift layout needs a regular grid and no missing row, which only the tiling pass finds out:
        // start over on the path such a batch takes otherwise
        release_device(h);
        *h = ssde_handle();
        st = build_impl(d, h, part_layout, false);
    }
    return st;
}
static int build_impl(const ssde_desc* d, ssde_handle* h, const ParLayout* part_layout, bool allow_drift) {
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
    if (d->n_dim < 1 || d->n_dim > 2)
        return fail(h, SSDE_ERR_MODEL, "n_dim must be 1 or 2 (wider responses are outside this engine's kernels)");
    if (d->n_par != n_sde_par(d->model, d->n_dim)) return fail(h, SSDE_ERR_ARG, "n_par does not match model / n_dim");
    if (d->n < 2) return fail(h, SSDE_ERR_ARG, "need at least two rows");
    if (!d->id || !d->times || !d->obs || !d->ncol_fe) return fail(h, SSDE_ERR_ARG, "id/times/obs/ncol_fe must be non-NULL");
    h->model = d->model; h->d = d->n_dim; h->q = d->n_par; h->n = d->n;
    if (const char* e = getenv("SSDE_WINDOW")) h->env_window = std::max(1, atoi(e));     // testing: deliberately short overlaps
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

    // ---- design blocks given as functions of a covariate (ssde_ppbasis) ---------------------------------------
    // Fast route: a direct family whose fast kernel applies (<= 2 parameters with columns, no decay), the whole
    // random-effect block of the parameter is the table and its fixed-effect part is the intercept: the kernel
    // evaluates the block from x (8 B/row).  Everything else gets the dense block materialised once in HBM.
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

    // ---- ID segments (nllk_ctcrw.hpp:196; R/sde.R:547,574) ---------------------------------------
    std::vector<int64_t> starts;
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
    } else {
        // ---- Kalman families: pick the path, then tile ------------------------------------------------
        for (int i = 0; i < h->sdim; i++)
            for (int j = 0; j < h->sdim; j++) h->p0_full[i + j * h->sdim] = p0_entry(d, i, j);
        const bool iso_ok = !h->has_h && h->const_coeff && p0_is_isotropic(d, h->p0_iso) &&
                            !(d->flags & SSDE_FLAG_FORCE_DENSE);
        h->path = iso_ok ? PATH_ISO : PATH_DENSE;
        // Row-varying DRIFT only (design columns in the rows of mu_1 .. mu_d, everything else constant), many tracks: the
        // register path with the design columns streamed next to the observations (k_iso_drift.hip) -- on a regular grid with
        // complete tracks (which the tiling pass below finds out) the covariance half is as data-independent as with constant
        // coefficients and the shared-covariance lanes run; otherwise the lanes carry their own covariance.  Few tracks (C1:
        // one animal) stay on the lane = direction path, whose windows cut ONE track into a hundred concurrent pieces.
        if (!iso_ok && allow_drift && !h->has_h && !h->const_coeff && p0_is_isotropic(d, h->p0_iso) && !(d->flags & SSDE_FLAG_FORCE_DENSE) &&
            !getenv("SSDE_NO_DRIFT") && h->n_stream_cols <= DRIFT_KMAX) {
            bool mu_only = true;
            for (auto& sl : h->slots)
                if (sl.col >= 0 && sl.par_j >= h->d) mu_only = false;
            int min_tracks = 32;
            if (const char* e = getenv("SSDE_DRIFT_MIN_TRACKS")) min_tracks = atoi(e);
            if (mu_only && h->n_seg >= min_tracks) { h->drift = 1; h->path = PATH_ISO; }
        }
        // row-varying coefficients with H = sigma_obs^2 I and a block-identical P0: the tv path
        // everything the constant-coefficient register path does not take: row-varying coefficients (isotropic
        // lanes), per-row H_array or a P0 that is not block-identical (full-covariance lanes)
        const bool tv_ok = !iso_ok && !h->drift && !(d->flags & SSDE_FLAG_FORCE_DENSE) && !getenv("SSDE_NO_TV") &&
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
        }
        std::vector<int64_t> order(M);
        std::iota(order.begin(), order.end(), 0);
        std::stable_sort(order.begin(), order.end(), [&](int64_t a, int64_t b) {
            if (seg_dirty[a] != seg_dirty[b]) return seg_dirty[a] < seg_dirty[b];
            return (tstarts[a + 1] - tstarts[a]) > (tstarts[b + 1] - tstarts[b]);
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
        h->C = h->c_obs + d->n_dim + (h->has_h ? d->n_dim * d->n_dim : 0) + h->n_stream_cols;
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
        std::vector<const double*> cp(h->n_stream_cols, nullptr);
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
        ia.cols = s_colptr.p; ia.ncols = h->n_stream_cols; ia.d = d->n_dim; ia.n = tn;
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
        mm.release(); s_times.release(); s_obs.release(); s_h.release(); s_a0.release(); s_cols.release();
        s_colptr.release(); s_lane_seg.release();
        pad_times.release(); pad_obs.release();
        h->hbm_bytes = h->tile_doubles * 8;

        if (h->drift) {
            // regular grid and every track complete: the shared-covariance lanes; otherwise the lanes carry their own covariance
            // (SSDE_NO_DRIFT_GENERAL: back to the lane = direction path instead, for A/B)
            bool all_clean = h->uniform_dt;
            for (int g = 0; g < G; g++) all_clean = all_clean && gflags[g] != 0;
            if (!all_clean && getenv("SSDE_NO_DRIFT_GENERAL")) return SSDE_RETRY_WITHOUT_DRIFT;
            if (getenv("SSDE_NO_SHARED")) all_clean = false;                  // (testing: the general lanes on a batch the shared ones would take)
            h->drift = all_clean ? 1 : 2;
            h->drift_nstate = all_clean ? drift_nstate(h->model, h->d, h->n_stream_cols) : drift_general_nstate(h->model, h->d, h->n_stream_cols);
        }
        if (h->path == PATH_ISO) {
            if (h->drift) { h->iso_parts = 1; h->iso_masks[0] = DIR_SIG | DIR_MU | DIR_P1 | DIR_P2; h->iso_free_mask = h->iso_masks[0]; }
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
            // a lone wave already issues fp64 at the SIMD's rate, and fewer windows mean fewer warm-up rows
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
            if (const char* e = getenv("SSDE_CHUNKS")) { want = atoi(e); h->chunks_forced = true; }   // testing
            h->max_chunks = std::max(1, std::min(want + 1, std::max(1, glmax / (4 * WIN_ALIGN))));
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
            if (h->use_shared) {
                h->gain_rows_cap = (size_t)glmax + 1;
                HIPCHK(h, h->gain_ring.alloc((size_t)PAR_RING * h->gain_rows_cap * GAIN_ROW));
                HIPCHK(h, hipHostMalloc((void**)&h->gain_pinned, (size_t)PAR_RING * h->gain_rows_cap * GAIN_ROW * 8,
                                        hipHostMallocDefault));
            }
            HIPCHK(h, h->bnd.alloc((size_t)h->iso_parts * buf_chunks * G * 2 * (h->drift ? std::max(NSTATE_MAX, h->drift_nstate) : NSTATE_MAX) * WAVE));
            HIPCHK(h, h->chk.alloc((size_t)h->iso_parts * buf_chunks * G));
            h->partial_doubles = (size_t)MAX_PARTS * buf_chunks * NACC_MAX * G;
            h->hbm_bytes += (int64_t)(h->bnd.n + h->chk.n) * 8;
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
    int W = 0;
    if (!(rho < 0.9995) || !std::isfinite(rho)) return;  // no usable forgetting: sequential filter
    // The stationary CTCRW lanes run the filter as 1/D(q)^2 recursions (k_iso_shared.hip): with closed-loop poles
    // close to 1 their intermediate signals grow like 1/(1-rho)^2 and cancel in the innovation -- below rho = 0.97
    // that costs < 1e-12 relative; above, the evaluation stays on the sequential direction-form filter
    if (h->use_shared && !h->drift && h->model == SSDE_MODEL_CTCRW && rho > 0.97) return;
    W = (int)std::ceil(std::log(1e-18) / std::log(std::max(rho, 1e-300))) + 16;
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
    while (nc > 1 && (glmax / nc) < 2 * W) nc--;
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
    if (!h->sync_call || h->par_ev_pending[slot]) { HIPCHK(h, hipEventSynchronize(h->par_ev[slot])); h->par_ev_pending[slot] = false; }
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
    }
    const int rows = last + 1;
    h->last_gain_rows = rows;
    HIPCHK(h, hipMemcpyAsync(dev, host, (size_t)rows * GAIN_ROW * 8, hipMemcpyHostToDevice, s));
    if (!h->sync_call) { HIPCHK(h, hipEventRecord(h->par_ev[slot], s)); h->par_ev_pending[slot] = true; }
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
// upload the parameter vector for the dense / direct kernels; returns the device pointer
int push_par(ssde_handle* h, const double* par, hipStream_t s, const double** dev) {
    const int slot = h->par_next;
    h->par_next = (h->par_next + 1) % PAR_RING;
    HIPCHK(h, hipEventSynchronize(h->par_ev[slot]));  // the slot's previous copy has left the host buffer
    double* host = h->par_pinned + (size_t)slot * MAX_PAR;
    double* devp = h->par_ring.p + (size_t)slot * MAX_PAR;
    memcpy(host, par, (size_t)h->L.n_full * 8);
    HIPCHK(h, hipMemcpyAsync(devp, host, (size_t)h->L.n_full * 8, hipMemcpyHostToDevice, s));
    HIPCHK(h, hipEventRecord(h->par_ev[slot], s));
    *dev = devp;
    return SSDE_OK;
}
}  // namespace ssde_engine

namespace ssde_engine {

// this evaluation's stamp pair (the previous evaluations' stay readable: ssde_kernel_ms_history)
static void next_stamp_pair(ssde_handle* h) {
    if (h->ev_idx >= 0) h->ev_ring_valid[h->ev_idx % ssde_handle::EV_RING] = h->ev_k_valid;
    h->ev_idx++;
    const int slot = (int)(h->ev_idx % ssde_handle::EV_RING);
    h->ev_k0 = h->ev_ring[slot][0]; h->ev_k1 = h->ev_ring[slot][1];
    h->ev_k_valid = false; h->ev_ring_valid[slot] = false;
}

int eval_device(ssde_handle* h, const double* par, int order, double* out_dev, hipStream_t s) {
    HIPCHK(h, hipSetDevice(h->device));
    h->n_evals++;
    next_stamp_pair(h);
    h->pub_armed = false;
    if (h->path == PATH_TV) { h->pub_request = false; return eval_tv(h, par, order, out_dev, s); }
    const ParLayout& L = h->L;
    ReduceArgs ra;
    memset(&ra, 0, sizeof(ra));
    ra.n_value_parts = 1; ra.chunks_per_part = 1; ra.chk = nullptr; ra.n_chk = 0;
    for (int i = 0; i < 4; i++) { ra.add[i] = 0.0; ra.add_slot[i] = -1; }
    ra.partials = h->partials.p;
    ra.n_out = 1 + L.n_full;
    ra.out = out_dev;
    for (int k = 0; k < MAX_PAR + 16; k++) ra.map[k] = -1;
    if (h->pub_request && h->pub_ok && out_dev == h->out.p) {
        // a synchronous evaluation: the reducing launch publishes the result itself (ssde_device.hpp: ReduceArgs.pub)
        ra.pub = h->pub_pinned; ra.pub_flag = h->pub_flag; ra.pub_seq = ++h->pub_seq; ra.pub_count = h->pub_count.p;
        h->pub_armed = true;
    }
    h->pub_request = false;

    if (h->path == PATH_ISO) {
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
        a.any_nan = h->na_any;
        a.uniform_dt = h->uniform_dt ? 1 : 0;
        const double sig = exp(par[0]);                     // nllk_ctcrw.hpp:136
        a.h = sig * sig;                                    // makeH: sigma_obs * sigma_obs
        for (int i = 0; i < h->d; i++) a.mu[i] = par[L.off_fe + L.fe_off[i]];
        for (int i = 0; i < 3; i++) a.p0[i] = h->p0_iso[i];
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
        a.all_clean = ((h->use_shared && h->n_clean_groups == h->n_groups) || h->drift) ? 1 : 0;      // (drift: one dump layout for every group)
        a.nstate_clean = h->drift ? h->drift_nstate
                       : h->use_shared ? shared_nstate(h->sdim, order >= 1 ? a.part_mask[0] : 0, h->model != SSDE_MODEL_BM_SSM) : 0;
        h->last_chunks = a.n_chunks; h->last_window = a.window;
        a.group_flags = h->group_flags.p;
        a.group_mode = 0;
        double add[4] = {0, 0, 0, 0};
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
            else HIPCHK(h, launch_iso_shared(h->model, h->d, b, s, h->stamps ? h->ev_k0 : nullptr, h->stamps ? h->ev_k1 : nullptr));
            h->ev_k_valid = h->stamps;
            h->last_s_stat = h->drift ? -1 : (a.gain_last + SHARED_U - 1) / SHARED_U * SHARED_U;
            if (any_dirty) HIPCHK(h, hipStreamWaitEvent(s, h->ev_join[1], 0));
        } else {
            if (h->stamps) HIPCHK(h, hipEventRecord(h->ev_k0, s));
            if (h->drift) { a.t0 = 0; a.t0_delta = 0; HIPCHK(h, launch_iso_drift_general(h->model, h->d, a, s)); }
            else HIPCHK(h, launch_iso(h->model, h->d, a, false, s));
            if (h->stamps) HIPCHK(h, hipEventRecord(h->ev_k1, s));
            h->ev_k_valid = h->stamps;
            h->last_s_stat = -1;
        }
        if (h->trace) { const double t = tick(); h->trace_us[2] += t - tk0; tk0 = t; }
        for (int i = 0; i < 4; i++) { ra.add[i] = add[i]; ra.add_slot[i] = -1; }
        if (h->use_shared) {
            ra.add_slot[0] = 0;
            if (order >= 1) {
                const int pj[NDIRP] = {0, L.off_fe + L.fe_off[h->d], h->q > h->d + 1 ? L.off_fe + L.fe_off[h->d + 1] : 0};
                for (int j = 0; j < NDIRP; j++)
                    if (pj[j] < L.n_full && !h->fixed[pj[j]] && (j < 2 || h->q > h->d + 1)) ra.add_slot[1 + j] = (int16_t)(1 + pj[j]);
            }
        }
        const int nacc = 4 + h->d + (h->drift ? h->n_stream_cols : 0);
        const int ncr = (a.dual && a.n_chunks_d > a.n_chunks) ? a.n_chunks_d : a.n_chunks;     // windows the final sums run over
        ra.n_parts = a.n_parts * ncr; ra.nacc = nacc; ra.n_blocks = h->n_groups;
        ra.n_value_parts = ncr; ra.chunks_per_part = ncr;
        ra.chk = h->chk.p; ra.n_chk = a.n_chunks > 1 ? a.n_parts * (a.n_chunks - 1) * h->n_groups : 0;
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
        // the hand-over checks and the final sums in one launch
        HIPCHK(h, launch_iso_finalize(h->model, h->d, a, ra, s));
        if (h->trace) {
            const double t = tick(); h->trace_us[3] += t - tk0; h->trace_n++;
            if (h->trace_skip < 8) {                        // the first calls load code objects: not what is being measured
                h->trace_skip++;
                for (double& v : h->trace_us) v = 0.0;
                h->trace_n = 0;
            }
        }
        return SSDE_OK;
    } else if (h->path == PATH_DENSE) {
        const double* pdev = nullptr;
        int st = push_par(h, par, s, &pdev);
        if (st) return st;
        DenseArgs a;
        memset(&a, 0, sizeof(a));
        a.tv.tiles = h->tiles.p; a.tv.group_off = h->group_off.p; a.tv.group_len = h->group_len.p;
        a.tv.lane_nsteps = h->lane_nsteps.p; a.tv.a0 = h->a0.p; a.tv.n_groups = h->n_groups; a.tv.C = h->C; a.tv.c_obs = h->c_obs; a.tv.dt_all = h->dt_all;
        a.model = h->model; a.d = h->d; a.any_nan = h->na_any; a.has_h = h->has_h ? 1 : 0;
        a.slots = h->slot_table.p; a.par = pdev; a.n_slots = (int)h->slots.size();
        for (int i = 0; i < 16; i++) a.p0[i] = h->p0_full[i];
        a.n_dirblocks = h->n_dirblocks; a.dirs = h->dirs.p; a.partials = h->partials.p;
        a.report = nullptr; a.lane_row0 = h->lane_row0.p; a.n = h->n; a.last_dt = h->last_dt;
        if (h->stamps) HIPCHK(h, hipEventRecord(h->ev_k0, s));
        HIPCHK(h, launch_dense(a, order >= 1, s));
        if (h->stamps) HIPCHK(h, hipEventRecord(h->ev_k1, s));
        h->ev_k_valid = h->stamps; h->last_s_stat = -1;
        if (order >= 1) {
            ra.n_parts = h->n_dirblocks; ra.nacc = 1 + DENSE_NT;
            for (size_t k = 0; k < h->dirs_host.size(); k++)
                if (h->dirs_host[k].kind != 0) ra.map[k] = (int16_t)(1 + h->dirs_host[k].pidx);
        } else {
            ra.n_parts = 1; ra.nacc = 1;
        }
        ra.n_blocks = h->n_groups;
    } else {
        if (h->direct_fast) {
            DirectFastArgs f;
            memset(&f, 0, sizeof(f));
            f.times = h->times.p; f.obs = h->obs.p; f.scored = h->scored.p; f.n = h->n;
            f.d = h->d; f.model = h->model; f.any_nan = h->na_any; f.n_blocks = h->direct_blocks;
            f.partials = h->partials.p;
            for (int j = 0; j < MAX_Q; j++) {
                f.has_icpt[j] = h->df_icpt[j] >= 0;
                f.base[j] = h->df_icpt[j] >= 0 ? par[h->df_icpt[j]] : 0.0;
            }
            f.ja = h->df_ja; f.jb = h->df_jb;
            f.ncA = (int)h->df_pidxA.size(); f.ncB = (int)h->df_pidxB.size();
            f.colA = h->df_colA; f.colB = h->df_colB; f.col_stride = h->col_stride;
            for (int c = 0; c < f.ncA; c++) f.coefA[c] = par[h->df_pidxA[c]];
            for (int c = 0; c < f.ncB; c++) f.coefB[c] = par[h->df_pidxB[c]];
            f.uniform_dt = h->direct_uniform_dt ? 1 : 0;
            f.dt_uniform = h->direct_dt;
            f.tdf = h->tdf; f.tconst = h->tconst;
            if (h->df_ja >= 0 && h->pp_fast[h->df_ja]) f.ppA = h->pp[h->df_ja];
            if (h->df_jb >= 0 && h->pp_fast[h->df_jb]) f.ppB = h->pp[h->df_jb];
            if (h->stamps) HIPCHK(h, hipEventRecord(h->ev_k0, s));
            HIPCHK(h, launch_direct_fast(f, s));
            if (h->stamps) HIPCHK(h, hipEventRecord(h->ev_k1, s));
            h->ev_k_valid = h->stamps; h->last_s_stat = -1;
            ra.n_parts = 1; ra.nacc = 1 + MAX_Q + f.ncA + f.ncB; ra.n_blocks = h->direct_blocks;
            if (order >= 1) {
                for (int j = 0; j < MAX_Q; j++)
                    if (h->df_icpt[j] >= 0 && !h->fixed[h->df_icpt[j]]) ra.map[j] = (int16_t)(1 + h->df_icpt[j]);
                for (int c = 0; c < f.ncA; c++)
                    if (!h->fixed[h->df_pidxA[c]]) ra.map[MAX_Q + c] = (int16_t)(1 + h->df_pidxA[c]);
                for (int c = 0; c < f.ncB; c++)
                    if (!h->fixed[h->df_pidxB[c]]) ra.map[MAX_Q + f.ncA + c] = (int16_t)(1 + h->df_pidxB[c]);
            }
            HIPCHK(h, launch_reduce(ra, s));
            return SSDE_OK;
        }
        const double* pdev = nullptr;
        int st = push_par(h, par, s, &pdev);
        if (st) return st;
        DirectArgs a;
        memset(&a, 0, sizeof(a));
        a.times = h->times.p; a.obs = h->obs.p; a.cols = h->colptr.p; a.scored = h->scored.p;
        a.n = h->n; a.d = h->d; a.model = h->model; a.any_nan = h->na_any;
        a.slots = h->slot_table.p; a.par = pdev; a.n_slots = (int)h->slots.size();
        a.n_blocks = h->direct_blocks; a.partials = h->partials.p;
        a.tdf = h->tdf; a.tconst = h->tconst;
        a.t_decay = h->tdecay.p; a.n_decay = L.n_decay; a.off_decay = L.off_decay;
        if (a.n_slots > 64) { h->err = "direct families: more than 64 coefficients"; return SSDE_ERR_ARG; }
        if (h->stamps) HIPCHK(h, hipEventRecord(h->ev_k0, s));
        HIPCHK(h, launch_direct(a, s));
        if (h->stamps) HIPCHK(h, hipEventRecord(h->ev_k1, s));
        h->ev_k_valid = h->stamps; h->last_s_stat = -1;
        ra.n_parts = 1; ra.nacc = 1 + a.n_slots + L.n_decay; ra.n_blocks = h->direct_blocks;
        if (order >= 1) {
            for (int k = 0; k < a.n_slots; k++)
                if (!h->fixed[h->slots[k].pidx]) ra.map[k] = (int16_t)(1 + h->slots[k].pidx);
            for (int q = 0; q < L.n_decay; q++)
                if (!h->fixed[L.off_decay + q]) ra.map[a.n_slots + q] = (int16_t)(1 + L.off_decay + q);
        }
    }
    HIPCHK(h, launch_reduce(ra, s));
    return SSDE_OK;
}

}  // namespace ssde_engine

extern "C" {

int ssde_abi_version(void) { return SSDE_ABI_VERSION; }

int ssde_create(const ssde_desc* desc, ssde_handle** out) {
    if (!desc || !out) { g_create_error = "NULL argument"; return SSDE_ERR_ARG; }
    *out = nullptr;
    ssde_handle* h = new (std::nothrow) ssde_handle();
    if (!h) { g_create_error = "out of host memory"; return SSDE_ERR_ALLOC; }
    // several engines behind one handle: whole tracks over several devices, and / or a response wider than two columns
    // evaluated as pairs of columns (the likelihood is a sum over dimensions whenever P0 and H do not couple them)
    const bool sharded = desc->abi_version == SSDE_ABI_VERSION && ((desc->n_devices > 1 && desc->devices) || desc->n_dim > 2);
    int st = sharded ? create_sharded(desc, h) : build(desc, h);
    if (st != SSDE_OK) {
        g_create_error = h->err;
        destroy(h);
        return st;
    }
    *out = h;
    return SSDE_OK;
}

void ssde_destroy(ssde_handle* h) { destroy(h); }

const char* ssde_last_error(const ssde_handle* h) { return h ? h->err.c_str() : g_create_error.c_str(); }

int ssde_eval_device(ssde_handle* h, const double* par, int32_t n_par_full, int32_t order, double* out_dev,
                     void* stream) {
    if (!h || !par || !out_dev) return SSDE_ERR_ARG;
    if (n_par_full != h->L.n_full) { h->err = "parameter vector has the wrong length"; return SSDE_ERR_ARG; }
    if (!h->shards.empty()) {
        if (h->n_track_shards > 1) { h->err = "ssde_eval_device: a multi-device handle is evaluated with ssde_eval"; return SSDE_ERR_ARG; }
        // the dimension parts of a wide response, all on one device: evaluate them on the caller's stream and sum them there
        const size_t count = 2 + (size_t)h->L.n_full;
        HIPCHK(h, hipSetDevice(h->device));
        for (ssde_handle* sh : h->shards) {
            int st = eval_device(sh, par, order, sh->out.p, (hipStream_t)stream);
            if (st) { h->err = sh->err; return st; }
        }
        HIPCHK(h, hipMemcpyAsync(out_dev, h->shards[0]->out.p, count * 8, hipMemcpyDeviceToDevice, (hipStream_t)stream));
        for (size_t e = 1; e < h->shards.size(); e++)
            HIPCHK(h, launch_sum_into(out_dev, h->shards[e]->out.p, (int)count, (hipStream_t)stream));
        if (h->poison) {
            // SSDE_NA_ANY_NAN, wide response: a NaN outside column 0 of an observed row is a NaN innovation in the reference:
            // value AND every free gradient entry (what ssde_eval returns for such a handle)
            std::vector<double>& nan_vec = h->poison_vec;          // (a member: it outlives the asynchronous copy)
            if (nan_vec.size() < count) nan_vec.assign(count, 0.0);
            for (size_t k = 0; k + 1 < count; k++) nan_vec[k] = (k == 0 || !h->fixed[k - 1]) ? std::numeric_limits<double>::quiet_NaN() : 0.0;
            HIPCHK(h, hipMemcpyAsync(out_dev, nan_vec.data(), (count - 1) * 8, hipMemcpyHostToDevice, (hipStream_t)stream));
        }
        int stw = SSDE_OK;
        if (!h->comms.empty()) stw = reduce_ranks(h, out_dev, (hipStream_t)stream);
        if (stw) return stw;
        // a later synchronous ssde_eval runs the parts on their own stream and shares their work buffers with this evaluation
        for (ssde_handle* sh : h->shards) {
            HIPCHK(h, hipEventRecord(sh->ev_async, (hipStream_t)stream));
            sh->async_pending = true;
        }
        return SSDE_OK;
    }
    int st = eval_device(h, par, order, out_dev, (hipStream_t)stream);
    if (st == SSDE_OK && !h->comms.empty()) st = reduce_ranks(h, out_dev, (hipStream_t)stream);     // one ncclAllReduce of 2 + p doubles on the same stream
    if (st) return st;
    // a later synchronous ssde_eval runs on the handle's own stream and shares this evaluation's work buffers
    HIPCHK(h, hipEventRecord(h->ev_async, (hipStream_t)stream));
    h->async_pending = true;
    return SSDE_OK;
}

int ssde_penalty(ssde_handle* h, const double* par, int32_t n_par_full, double* value, double* grad) {
    if (!h || !par || !value) return SSDE_ERR_ARG;
    if (n_par_full != h->L.n_full) { h->err = "parameter vector has the wrong length"; return SSDE_ERR_ARG; }
    std::vector<double> g(h->L.n_full, 0.0);
    *value = h->pen.eval(h->L, par, grad ? g.data() : nullptr);
    if (grad)
        for (int k = 0; k < h->L.n_full; k++)
            if (!h->fixed[k]) grad[k] += g[k];
    return SSDE_OK;
}

}  // extern "C"

namespace {

// the engines one evaluation runs on: the handle itself, or the shards of a multi-device parent
template <class F>
void each_engine(ssde_handle* h, F fn) {
    if (h->shards.empty()) fn(h);
    else for (ssde_handle* s : h->shards) fn(s);
}

// Is the gradient free where the value is computed?  On the shared-covariance and direct kernels the
// sensitivities ride along in registers with the HBM stream that bounds the kernel, so an order-0 call evaluates
// order 1 and memoises it: optim's fn(x); gr(x) (R/sde.R:694-696) then costs one evaluation.
bool grad_rides_along(const ssde_handle* h) {
    // ranks of a communicator: what ALL of them can do (a rank that memoised a gradient the others never computed would
    // answer the following gr(x) from its memo while the others enter the collective alone)
    if (!h->comms.empty() && h->comm_rides >= 0 && (h->shards.empty() || h->n_track_shards <= 1)) return h->comm_rides != 0;
    const ssde_handle* e = h->shards.empty() ? h : h->shards[0];
    return e->path == PATH_DIRECT || (e->path == PATH_ISO && e->use_shared);
}

// One evaluation of every engine of the handle + the sum over shards / ranks, result [nllk_data, grad..., check]
// in host memory.  Nothing of the retry policy lives here.
int run_once(ssde_handle* h, const double* par, int order, double* o) {
    const size_t nout = 2 + (size_t)h->L.n_full;
    if (!h->shards.empty()) {
        int dev_before = 0;
        (void)hipGetDevice(&dev_before);                   // the caller's current device is left as it was found
        struct Restore { int d; ~Restore() { (void)hipSetDevice(d); } } restore{dev_before};
        for (ssde_handle* sh : h->shards) {
            if (sh->async_pending) {         // an ssde_eval_device of this handle still running on the caller's stream shares the work buffers
                HIPCHK(h, hipSetDevice(sh->device));
                HIPCHK(h, hipStreamWaitEvent(sh->own_stream, sh->ev_async, 0));
                sh->async_pending = false;
            }
            int st = eval_device(sh, par, order, sh->out.p, sh->own_stream);
            if (st) { h->err = sh->err; return st; }
        }
        int st = reduce_shards(h);
        if (st) return st;
        ssde_handle* s0 = h->shards[0];
        HIPCHK(h, hipSetDevice(s0->device));
        HIPCHK(h, hipMemcpyAsync(o, s0->out.p, nout * 8, hipMemcpyDeviceToHost, s0->own_stream));
        // every device has to be done before the next evaluation overwrites what the collective reads
        for (ssde_handle* sh : h->shards) {
            HIPCHK(h, hipSetDevice(sh->device));
            HIPCHK(h, hipStreamSynchronize(sh->own_stream));
        }
        return SSDE_OK;
    }
    // Stream discipline of the synchronous call: the NULL stream and a blocking 48-byte read-back.  The alternative --
    // the handle's own non-blocking stream, an asynchronous copy into pinned memory and one stream synchronisation --
    // saves 7.5 us in isolation (tools/microbench_graph.hip: 24 against 31.5 us of fixed overhead for copy + two
    // launches + read-back; a hipGraph of the same four operations measures the same 24 us) but NOT inside the engine:
    // same-session A/B (SSDE_SYNC_OWN_STREAM=1, tools/bench_c2.py) C2 0.0875 against 0.0872 ms; headline 0.315 against
    // 0.3035 ms in one session, 0.3025 against 0.3029 ms in another.  No gain to be had: the null stream stays.
    HIPCHK(h, hipSetDevice(h->device));
    if (h->async_pending) {              // an ssde_eval_device still running on the caller's stream shares the work buffers
        HIPCHK(h, hipStreamWaitEvent(0, h->ev_async, 0));
        if (h->tv_stream) HIPCHK(h, hipStreamWaitEvent(h->tv_stream, h->ev_async, 0));
        h->async_pending = false;
    }
    if (!h->comms.empty()) {
        // the collective sits between the finalize launch and the read-back, on the same stream
        h->sync_call = true;
        int st = eval_device(h, par, order, h->out.p, 0);
        h->sync_call = false;
        if (st) return st;
        st = reduce_ranks(h, h->out.p, 0);
        if (st) return st;
        HIPCHK(h, hipMemcpy(o, h->out.p, nout * 8, hipMemcpyDeviceToHost));
        return SSDE_OK;
    }
    if (h->path == PATH_TV && !h->env_no_graph && h->tv_stats_valid) {
        HIPCHK(h, hipSetDevice(h->device));
        h->n_evals++;
        next_stamp_pair(h);                                    // (a replayed graph carries no stamps: the slot stays invalid)
        return eval_tv_graph(h, par, order, o);
    }
    if (h->env_own_stream) {             // A/B (see above): own non-blocking stream, asynchronous read-back into pinned memory
        if (!h->own_stream) HIPCHK(h, hipStreamCreateWithFlags(&h->own_stream, hipStreamNonBlocking));
        h->sync_call = true;
        int st1 = eval_device(h, par, order, h->out.p, h->own_stream);
        h->sync_call = false;
        if (st1) return st1;
        HIPCHK(h, hipMemcpyAsync(h->out_pinned, h->out.p, nout * 8, hipMemcpyDeviceToHost, h->own_stream));
        HIPCHK(h, hipStreamSynchronize(h->own_stream));
        memcpy(o, h->out_pinned, nout * 8);
        return SSDE_OK;
    }
    h->sync_call = true;
    h->pub_request = true;
    int st = eval_device(h, par, order, h->out.p, 0);
    h->sync_call = false;
    if (st) return st;
    const auto t0 = std::chrono::steady_clock::now();
    // The reducing launch's last workgroup has been told to copy the result into pinned memory and to store this
    // evaluation's sequence number after it: spin on that word.  (tools/microbench_latency.hip, 40-us kernel: 12.7 us of
    // fixed overhead against 17.9 with the blocking 48-byte copy this replaces and 25.7 with the copy and event-stamped
    // launches.  Rounds 1 and 2 had measured a pinned mirror as SLOWER; that was with the stamps on and a host that
    // synchronised the stream first.)  In the engine the two measure the same (round 3, fence-free counting); the blocking copy
    // stays the default and SSDE_PUBLISH=1 selects the spin.
    if (h->pub_armed) {
        h->pub_armed = false;
        const unsigned long long want = h->pub_seq;
        bool seen = false;
        for (uint64_t spins = 1;; spins++) {
            if (__atomic_load_n(h->pub_flag, __ATOMIC_ACQUIRE) == want) { seen = true; break; }
            if ((spins & 0xFFFF) == 0) {
                // a launch that failed never publishes: ask the runtime now and then instead of spinning for ever
                const hipError_t q = hipStreamQuery(0);
                if (q == hipSuccess) { seen = __atomic_load_n(h->pub_flag, __ATOMIC_ACQUIRE) == want; break; }
                if (q != hipErrorNotReady) { h->err = std::string("evaluation failed on the device: ") + hipGetErrorString(q); return SSDE_ERR_HIP; }
            }
            __builtin_ia32_pause();
        }
        if (seen) memcpy(o, h->pub_pinned, nout * 8);
        else HIPCHK(h, hipMemcpy(o, h->out.p, nout * 8, hipMemcpyDeviceToHost));     // (stream drained without the word: read the device buffer)
    } else {
        HIPCHK(h, hipMemcpy(o, h->out.p, nout * 8, hipMemcpyDeviceToHost));
    }
    if (h->trace) h->trace_us[4] += std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
    return SSDE_OK;
}

// An evaluation with the window policy around it: re-evaluate with a wider warm-up until the time
// windows agree, narrow again on probation after `cooldown` calm evaluations.  Every decision is taken on the
// REDUCED check value (a sum of non-negative per-shard / per-rank maxima, compared with the single-engine
// tolerance: conservative), so the shards of a parent and the ranks of a communicator move in lockstep and the
// collective inside run_once is entered by everyone the same number of times.
int run_checked(ssde_handle* h, const double* par, int order, std::vector<double>& o) {
    const bool dist = !h->shards.empty() || !h->comms.empty();
    int attempt = 0;
    for (;; attempt++) {
        int st = run_once(h, par, order, o.data());
        if (st) return st;
        h->last_check = o[1 + h->L.n_full];
        each_engine(h, [&](ssde_handle* e) { e->last_check = h->last_check; });
        // hand-over check of the time windows (k_iso.hip): widen the warm-up and re-evaluate
        // until the windows agree with each other; 64x the estimate ends in one sequential window
        if (h->last_check <= SSDE_WINDOW_TOL) break;
        if (!dist && h->last_chunks <= 1) break;
        // a non-finite nllk is rejected by the caller whatever the windows did: no retry, and no lasting
        // widening of the plan because an optimiser probed an absurd parameter once
        if (!std::isfinite(o[0])) break;
        h->n_retries++;
        h->calm = 0;
        if (h->probing && attempt == 0) {
            // the narrower plan tried on probation does not hold here: back to the one that worked, and wait twice
            // as long before the next try
            h->probing = false;
            h->cooldown = std::min(h->cooldown * 2, 1 << 14);
            if (h->probe_from == 0) each_engine(h, [](ssde_handle* e) { e->max_chunks = 1; e->want_chunks = 1; e->gave_up = true; });
            else each_engine(h, [&](ssde_handle* e) { e->window_boost = h->probe_from; });
            if (h->probe_from == 0) h->gave_up = true;
            else h->window_boost = h->probe_from;
            continue;
        }
        // the row-varying path plans from the parameter ranges its pre-pass saw in the PREVIOUS evaluation: after a
        // jump in the parameters the first retry needs no boost, just this evaluation's own ranges
        const int path0 = h->shards.empty() ? h->path : h->shards[0]->path;
        if (path0 == PATH_TV && attempt == 0) continue;
        if (attempt >= 3) {                                            // give up on windows: sequential filter
            each_engine(h, [](ssde_handle* e) {
                if (!e->gave_up) { e->saved_max_chunks = e->max_chunks; e->saved_want_chunks = e->want_chunks; e->gave_up = true; }
                e->max_chunks = 1; e->want_chunks = 1;
            });
            h->gave_up = true;
        } else {
            each_engine(h, [](ssde_handle* e) { e->window_boost *= 4; });
            if (!h->shards.empty()) h->window_boost *= 4;
        }
    }
    if (std::isfinite(o[0]) && !(h->last_check <= h->check_max)) h->check_max = h->last_check;
    // A widened plan is not for life: one slow-forgetting parameter vector in a line search would otherwise tax every
    // later evaluation.  Every evaluation is checked, so narrowing on probation is safe -- a failure costs one retry.
    const bool forced = h->shards.empty() ? h->chunks_forced : h->shards[0]->chunks_forced;
    if (attempt == 0 && !forced) {
        h->calm++;
        if (h->probing && h->calm >= 4) h->probing = false;           // the narrower plan holds
        if (h->calm >= h->cooldown && (h->gave_up || h->window_boost > 1)) {
            h->calm = 0;
            h->probing = true;
            if (h->gave_up) {
                h->probe_from = 0;
                each_engine(h, [](ssde_handle* e) { e->max_chunks = e->saved_max_chunks; e->want_chunks = e->saved_want_chunks; e->gave_up = false; });
                h->gave_up = false;
            } else {
                h->probe_from = h->window_boost;
                each_engine(h, [](ssde_handle* e) { e->window_boost = std::max(1, e->window_boost / 2); });
                if (!h->shards.empty()) h->window_boost = std::max(1, h->window_boost / 2);
            }
        }
    }
    return SSDE_OK;
}

}  // namespace

extern "C" {

int ssde_eval(ssde_handle* h, const double* par, int32_t n_par_full, int32_t order, double* value, double* grad) {
    if (!h || !par || !value) return SSDE_ERR_ARG;
    if (n_par_full != h->L.n_full) { h->err = "parameter vector has the wrong length"; return SSDE_ERR_ARG; }
    const size_t np = (size_t)h->L.n_full;
    const bool want_grad = order >= 1 && grad;
    // memo: same bit pattern as the last evaluated vector, and what is asked for was computed then
    if (h->memo_order >= (want_grad ? 1 : 0) && memcmp(par, h->memo_par.data(), np * 8) == 0) {
        *value = h->memo_value;
        if (want_grad) memcpy(grad, h->memo_grad.data(), np * 8);
        h->n_memo_hits++;
        return SSDE_OK;
    }
    const int eval_order = (want_grad || grad_rides_along(h)) ? 1 : 0;
    std::vector<double>& o = h->eval_out;
    o.resize(2 + np);
    int st = run_checked(h, par, eval_order, o);
    if (st) return st;
    if (h->poison)                     // SSDE_NA_ANY_NAN, wide response: a NaN outside column 0 of an observed row (a NaN innovation
        for (size_t k = 0; k < 1 + np; k++) o[k] = (k == 0 || !h->fixed[k - 1]) ? std::numeric_limits<double>::quiet_NaN() : 0.0;   // in the reference)
    double pen = 0.0;
    h->memo_order = -1;
    h->memo_par.assign(par, par + np);
    h->memo_grad.assign(np, 0.0);
    if (eval_order >= 1) {
        for (size_t k = 0; k < np; k++) h->memo_grad[k] = o[1 + k];
        st = ssde_penalty(h, par, n_par_full, &pen, h->memo_grad.data());
        if (want_grad) memcpy(grad, h->memo_grad.data(), np * 8);
    } else {
        st = ssde_penalty(h, par, n_par_full, &pen, nullptr);
    }
    *value = h->memo_value = o[0] + pen;
    if (st == SSDE_OK) h->memo_order = eval_order;
    return st;
}

namespace {
void fill_single_rows(const ssde_handle* h, double* aest_all) {
    for (size_t k = 0; k < h->single_rows.size(); k++)
        for (int c = 0; c < h->sdim; c++) aest_all[h->single_rows[k] + (int64_t)c * h->n] = h->single_a0[k * h->sdim + c];
}
}  // namespace

int ssde_report(ssde_handle* h, const double* par, int32_t n_par_full, double* aest_all) {
    if (!h || !par || !aest_all) return SSDE_ERR_ARG;
    if (n_par_full != h->L.n_full) { h->err = "parameter vector has the wrong length"; return SSDE_ERR_ARG; }
    if (!is_kalman(h->model)) { h->err = "aest_all is reported by the Kalman families only (the ESEAL template has no REPORT)"; return SSDE_ERR_MODEL; }
    if (!h->shards.empty()) return report_sharded(h, par, aest_all);
    HIPCHK(h, hipSetDevice(h->device));
    if (h->path == PATH_TV) {
        // one sequential window per track, direction block 0, states written straight to the long format
        const int tpw = WAVE >> h->tv_lpt_shift;
        const int64_t n_packs = (h->n_seg + tpw - 1) / tpw;
        std::vector<TvItem> items;
        for (int64_t p = 0; p < n_packs; p++) items.push_back({(int32_t)p, 0, 1, 0});
        DevBuf<TvItem> ib;
        DevBuf<double> rep;
        HIPCHK(h, ib.upload(items));
        HIPCHK(h, rep.alloc((size_t)h->n * h->sdim));
        HIPCHK(h, hipMemset(rep.p, 0, (size_t)h->n * h->sdim * 8));
        const double* pdev = nullptr;
        int st = push_par(h, par, 0, &pdev);
        if (st) return st;
        TvArgs a;
        tv_base_args(h, a);
        a.par = pdev;
        const double sig = exp(par[0]);
        a.h = sig * sig;
        a.items = ib.p; a.n_items = (int)items.size(); a.window = 0; a.report = rep.p;
        HIPCHK(h, hipStreamSynchronize(0));          // the record buffer is shared with in-flight evaluations
        HIPCHK(h, launch_tv_prepare(a, 0));
        HIPCHK(h, launch_tv_filter(a, false, 0));
        HIPCHK(h, hipMemcpy(aest_all, rep.p, (size_t)h->n * h->sdim * 8, hipMemcpyDeviceToHost));
        fill_single_rows(h, aest_all);
        h->tv_stats_valid = false;                   // the stats buffer now describes this parameter vector
        ib.release(); rep.release();
        return SSDE_OK;
    }
    // ssde_report always runs the general kernel (value only) and un-tiles on the fly
    DevBuf<double> rep, pbuf;
    DevBuf<SlotTable> stb;
    const int64_t nt = h->n_pad > 0 ? h->n_pad : h->n;          // rows of the tiled (possibly lattice-padded) layout
    HIPCHK(h, rep.alloc((size_t)nt * h->sdim));
    HIPCHK(h, hipMemset(rep.p, 0, (size_t)nt * h->sdim * 8));
    HIPCHK(h, pbuf.upload(std::vector<double>(par, par + h->L.n_full)));
    SlotTable st;
    memset(&st, 0, sizeof(st));
    st.n_slots = (int)h->slots.size(); st.q = h->q;
    for (size_t k = 0; k < h->slots.size(); k++) {
        st.par_j[k] = (int16_t)h->slots[k].par_j; st.col[k] = (int16_t)h->slots[k].col;
        st.pidx[k] = (int16_t)h->slots[k].pidx; st.is_free[k] = 0;
    }
    HIPCHK(h, stb.upload(std::vector<SlotTable>(1, st)));
    DenseArgs a;
    memset(&a, 0, sizeof(a));
    a.tv.tiles = h->tiles.p; a.tv.group_off = h->group_off.p; a.tv.group_len = h->group_len.p;
    a.tv.lane_nsteps = h->lane_nsteps.p; a.tv.a0 = h->a0.p; a.tv.n_groups = h->n_groups; a.tv.C = h->C; a.tv.c_obs = h->c_obs; a.tv.dt_all = h->dt_all;
    a.model = h->model; a.d = h->d; a.any_nan = h->na_any; a.has_h = h->has_h ? 1 : 0;
    a.slots = stb.p; a.par = pbuf.p; a.n_slots = st.n_slots;
    for (int i = 0; i < 16; i++) a.p0[i] = h->p0_full[i];
    a.n_dirblocks = 1; a.dirs = nullptr; a.partials = nullptr;
    a.report = rep.p; a.lane_row0 = h->lane_row0.p; a.n = nt; a.last_dt = h->last_dt;
    HIPCHK(h, launch_dense(a, false, 0));
    if (h->n_pad > 0) {                                          // the caller's rows out of the lattice's
        DevBuf<double> rows;
        HIPCHK(h, rows.alloc((size_t)h->n * h->sdim));
        HIPCHK(h, launch_lattice_gather(h->pad_pos.p, rep.p, h->n, nt, h->sdim, rows.p, 0));
        HIPCHK(h, hipMemcpy(aest_all, rows.p, (size_t)h->n * h->sdim * 8, hipMemcpyDeviceToHost));
        rows.release();
    } else
    HIPCHK(h, hipMemcpy(aest_all, rep.p, (size_t)h->n * h->sdim * 8, hipMemcpyDeviceToHost));
    fill_single_rows(h, aest_all);
    rep.release(); pbuf.release(); stb.release();
    return SSDE_OK;
}

int ssde_widen_windows(ssde_handle* h, int32_t factor) {
    if (!h) return SSDE_ERR_ARG;
    for (ssde_handle* s : h->shards) ssde_widen_windows(s, factor);
    if (factor <= 0) { h->max_chunks = 1; h->want_chunks = 1; }
    else if (h->window_boost < (1 << 20)) h->window_boost *= factor;
    h->memo_order = -1;
    return SSDE_OK;
}

double ssde_last_kernel_ms(const ssde_handle* h) {
    if (!h) return 0.0;
    if (!h->shards.empty()) {
        double m = 0.0;
        for (const ssde_handle* s : h->shards) m = std::max(m, ssde_last_kernel_ms(s));
        return m;
    }
    float ms = 0.f;
    if (h->ev_k_valid && hipEventQuery(h->ev_k1) == hipSuccess && hipEventElapsedTime(&ms, h->ev_k0, h->ev_k1) == hipSuccess) return ms;
    return 0.0;
}

int ssde_kernel_ms_history(const ssde_handle* h, double* ms, int32_t n) {
    if (!h || !ms || n < 0) return SSDE_ERR_ARG;
    if (!h->shards.empty()) {                        // the slowest shard, evaluation by evaluation
        std::vector<double> tmp((size_t)n);
        for (int k = 0; k < n; k++) ms[k] = 0.0;
        for (const ssde_handle* sh : h->shards) {
            int st = ssde_kernel_ms_history(sh, tmp.data(), n);
            if (st) return st;
            for (int k = 0; k < n; k++) ms[k] = std::max(ms[k], tmp[k]);
        }
        return SSDE_OK;
    }
    // ms[0] = the last evaluation, ms[1] the one before, ...; 0 where no stamp exists (older than the ring, a replayed graph)
    for (int k = 0; k < n; k++) {
        ms[k] = 0.0;
        const int64_t idx = h->ev_idx - k;
        if (idx < 0 || k >= ssde_handle::EV_RING) continue;
        const int slot = (int)(idx % ssde_handle::EV_RING);
        const bool valid = k == 0 ? h->ev_k_valid : h->ev_ring_valid[slot];
        float f = 0.f;
        if (valid && hipEventQuery(h->ev_ring[slot][1]) == hipSuccess &&
            hipEventElapsedTime(&f, h->ev_ring[slot][0], h->ev_ring[slot][1]) == hipSuccess) ms[k] = f;
    }
    return SSDE_OK;
}

int ssde_set_option(ssde_handle* h, int32_t option, int64_t value) {
    if (!h) return SSDE_ERR_ARG;
    if (option == SSDE_OPT_KERNEL_STAMPS) {
        h->stamps = value != 0;
        for (ssde_handle* s : h->shards) s->stamps = h->stamps;
        return SSDE_OK;
    }
    h->err = "ssde_set_option: unknown option";
    return SSDE_ERR_ARG;
}

int ssde_forget(ssde_handle* h) {
    if (!h) return SSDE_ERR_ARG;
    h->memo_order = -1;
    return SSDE_OK;
}

int ssde_relax_windows(ssde_handle* h) {
    if (!h) return SSDE_ERR_ARG;
    for (ssde_handle* s : h->shards) ssde_relax_windows(s);
    h->window_boost = std::max(1, h->window_boost / 2);
    return SSDE_OK;
}

int ssde_info(const ssde_handle* h, ssde_info_t* info) {
    if (!h || !info) return SSDE_ERR_ARG;
    memset(info, 0, sizeof(*info));
    if (!h->shards.empty()) {
        // a multi-device parent: totals over the shards, plan and path of shard 0, the slowest shard's kernel time
        // Engine k = track shard k / P, dimension part k % P.  Rows and tracks are counted once per track shard; the bytes a
        // row costs add up over its dimension parts (each part streams its own columns; the time stamp is counted once).
        ssde_info_t si;
        const int P = h->n_dim_parts;
        double algo = 8.0, required = 0.0, ms_shard = 0.0;
        for (size_t k = 0; k < h->shards.size(); k++) {
            ssde_info(h->shards[k], &si);
            const bool first_part = k % P == 0;
            if (k == 0) *info = si;
            else {
                if (first_part) { info->n_tracks += si.n_tracks; info->n_rows += si.n_rows; info->n_steps += si.n_steps; info->main_kernel_rows += si.main_kernel_rows; }
                info->hbm_bytes += si.hbm_bytes;
                info->n_rows_tiled += si.n_rows_tiled; info->n_groups += si.n_groups; info->n_clean_groups += si.n_clean_groups;
                info->n_kernel_blocks += si.n_kernel_blocks;   // (n_evals: shard 0's count -- every shard runs every evaluation)
                info->uniform_dt = info->uniform_dt && si.uniform_dt;
                info->const_coeff = info->const_coeff && si.const_coeff;
            }
            // the parts of one track shard run one after the other on their device; the shards side by side
            ms_shard = first_part ? si.main_kernel_ms : ms_shard + si.main_kernel_ms;
            if (k == 0 || ms_shard > info->main_kernel_ms) info->main_kernel_ms = ms_shard;
            if (k < (size_t)P) { algo += si.algo_bytes_per_row - 8.0; required += si.required_bytes_per_row; }
        }
        if (P > 1) {
            // SURVEY 8(d)'s figure counts the time stamp once; what the parts really read counts it once per part that
            // reads it at all (none does on a globally regular grid)
            info->algo_bytes_per_row = algo;
            info->required_bytes_per_row = required;
            info->sdim = h->sdim;
        }
        info->window_check = h->last_check; info->window_retries = h->n_retries; info->window_check_max = h->check_max;
        info->n_memo_hits = h->n_memo_hits;
        info->n_devices = h->n_track_shards; info->comm_ranks = h->comm_ranks;
        return SSDE_OK;
    }
    info->n_par_full = h->L.n_full;
    info->n_free = h->n_free;
    info->sdim = h->sdim;
    info->path = h->path;
    info->const_coeff = h->const_coeff;
    info->uniform_dt = h->uniform_dt;
    info->n_tracks = h->n_seg;
    info->n_rows = h->n;
    info->n_steps = h->n_steps;
    info->hbm_bytes = h->hbm_bytes;
    info->algo_bytes_per_row = 8.0 * (h->d + 1 + (h->has_h ? h->d * h->d : 0) + h->n_stream_cols_algo);
    // what the resident layout has to read per row: the `times` stream is not even stored when the grid is globally
    // regular (Kalman tiles without a dt channel); the direct families do not read it on a regular grid either
    // (on the isotropic path with a hoisted transition nobody reads the dt slot even where it is stored -- a grid that is
    //  regular within the tracks but not across their boundaries, or a lattice layout: its rows are counted per CALLER row)
    info->required_bytes_per_row = info->algo_bytes_per_row -
        (((h->path == PATH_ISO || h->path == PATH_DENSE) && h->c_obs == 0) || (h->path == PATH_ISO && h->uniform_dt) || (h->path == PATH_DIRECT && h->direct_fast && h->direct_uniform_dt && h->df_ja != h->d && h->df_jb != h->d &&
          h->df_ja != h->d + 1 && h->df_jb != h->d + 1 && h->model != SSDE_MODEL_BM_T && h->model != SSDE_MODEL_CIR) ? 8.0 : 0.0);
    if (h->path == PATH_DIRECT && h->direct_fast) {
        // a block evaluated on the fly from its basis table is read as 8 B/row of covariate, not as its K streamed columns
        if (h->df_ja >= 0 && h->pp_fast[h->df_ja]) info->required_bytes_per_row -= 8.0 * ((double)h->df_pidxA.size() - 1.0);
        if (h->df_jb >= 0 && h->pp_fast[h->df_jb]) info->required_bytes_per_row -= 8.0 * ((double)h->df_pidxB.size() - 1.0);
    }
    if (h->n_pad > 0) info->required_bytes_per_row *= (double)h->n_pad / (double)h->n;
    if (h->path == PATH_ISO || h->path == PATH_DENSE) {
        info->n_rows_tiled = h->n_pad > 0 ? h->n_pad : h->n;
        info->n_groups = h->n_groups; info->n_clean_groups = h->path == PATH_ISO ? h->n_clean_groups : 0;
    }
    info->n_evals = h->n_evals; info->n_memo_hits = h->n_memo_hits;
    info->n_devices = 1; info->comm_ranks = h->comm_ranks; info->window_check_max = h->check_max;
    if (h->path == PATH_ISO)   // 4-wave workgroups; with a transient window the grid enumerates windows 1.. only
        info->n_kernel_blocks = ((h->n_groups + 7) / 8 * 8 * h->iso_parts * ((h->use_shared && h->last_t0 > 0) ? h->last_chunks - 1 : h->last_chunks) + WG_WAVES - 1) / WG_WAVES;
    else if (h->path == PATH_DENSE) info->n_kernel_blocks = h->n_groups * h->n_dirblocks;
    else if (h->path == PATH_TV) info->n_kernel_blocks = (h->tv_n_items_g + WG_WAVES - 1) / WG_WAVES;
    else info->n_kernel_blocks = h->direct_blocks;
    info->lanes_per_track = h->path == PATH_ISO ? h->iso_parts * h->last_chunks
                          : h->path == PATH_DENSE ? h->n_dirblocks
                          : h->path == PATH_TV ? h->tv_ndp * h->tv_max_nc : 1;
    info->window = h->last_window;
    info->window_check = h->last_check;
    info->window_retries = h->n_retries;
    info->main_kernel_ms = 0.0;
    if (h->ev_k_valid && hipEventQuery(h->ev_k1) == hipSuccess) {
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, h->ev_k0, h->ev_k1) == hipSuccess) info->main_kernel_ms = ms;
    }
    // rows scored by the dominant launch: everything, except on the shared-covariance path where the
    // windows that touch the covariance transient run in the small concurrent launch
    info->main_kernel_rows = h->n_steps;
    if (h->path == PATH_ISO && h->last_s_stat >= 0 && h->rows_key[0] == h->last_chunks &&
        h->rows_key[1] == h->last_window && h->rows_key[2] == h->last_s_stat + 100000 * h->last_t0) {
        info->main_kernel_rows = h->rows_cached;
    } else if (h->path == PATH_ISO && h->last_s_stat >= 0) {
        int64_t rows = 0;
        const int nc = h->last_chunks;
        for (int g = 0; g < h->n_groups; g++) {
            const int L = h->glen_host[g];
            for (int c = 0; c < nc; c++) {
                int s_begin, s_acc, s_end;
                window_bounds(L, nc, h->last_window, h->last_t0, c, s_begin, s_acc, s_end, h->last_t0_delta);
                for (int l = 0; l < WAVE; l++) {
                    const int ns = h->lane_ns_host[(size_t)g * WAVE + l];
                    rows += std::max(0, std::min(ns, s_end) - s_acc);
                }
            }
        }
        if (h->n_pad > 0) rows = (int64_t)((double)rows * (double)h->n / (double)h->n_pad);   // lattice rows -> caller rows
        info->main_kernel_rows = rows;
        h->rows_cached = rows;
        h->rows_key[0] = h->last_chunks; h->rows_key[1] = h->last_window; h->rows_key[2] = h->last_s_stat + 100000 * h->last_t0;
    }
    return SSDE_OK;
}

}  // extern "C"
