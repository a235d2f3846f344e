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
void release_device(ssde_handle* h) {
    if (!h->wave_clock_file.empty() && h->wave_clock.p && h->wave_clock_items > 0) {
        std::vector<double> w((size_t)4 * h->wave_clock_items);
        if (hipSetDevice(h->device) == hipSuccess && hipDeviceSynchronize() == hipSuccess &&
            hipMemcpy(w.data(), h->wave_clock.p, w.size() * 8, hipMemcpyDeviceToHost) == hipSuccess) {
            if (FILE* f = fopen(h->wave_clock_file.c_str(), "w")) {
                fprintf(f, "# work item, start, end (100 MHz ticks), HW_ID, rows; windows %d, warm-up %d, t0 %d, t0_delta %d, groups %d, longest group %d\n", h->last_chunks, h->last_window, h->last_t0, h->last_t0_delta, h->n_groups, h->glen_max);
                for (int i = 0; i < h->wave_clock_items; i++)
                    if (h->drift == 3) fprintf(f, "%d %.1f %.1f %.1f %.1f\n", i, w[4 * (size_t)i], w[4 * (size_t)i + 1], w[4 * (size_t)i + 2], w[4 * (size_t)i + 3]);
                    else if (w[4 * (size_t)i + 3] != 0.0) fprintf(f, "%d %.0f %.0f %.0f %.0f\n", i, w[4 * (size_t)i], w[4 * (size_t)i + 1], w[4 * (size_t)i + 2], w[4 * (size_t)i + 3] - 1.0);
                fclose(f);
            }
        }
    }
    h->wave_clock.release();
    for (hipEvent_t& e : h->ev_ph) { if (e) (void)hipEventDestroy(e); e = nullptr; }
    h->cv_ranges.release(); h->cv_parts.release(); h->adj_ckpt.release(); h->fuse_words.release();
    if (h->cv_ranges_pinned) { (void)hipHostFree(h->cv_ranges_pinned); h->cv_ranges_pinned = nullptr; }
    h->hs_partials.release(); h->hs_hess.release(); h->hs_i16.release();
    if (h->trace && h->trace_n > 0)
        fprintf(stderr, "[ssde trace] %lld isotropic evaluations, host us per evaluation: plan %.1f | gain table %.1f | main launch %.1f | "
                        "finalize launch %.1f | read-back (blocks until the GPU is done) %.1f\n", (long long)h->trace_n,
                h->trace_us[0] / h->trace_n, h->trace_us[1] / h->trace_n, h->trace_us[2] / h->trace_n, h->trace_us[3] / h->trace_n,
                h->trace_us[4] / h->trace_n);
    destroy_dist(h);
    h->bnd.release(); h->chk.release(); h->group_flags.release(); h->gain_ring.release(); h->pad_pos.release(); h->dirty_groups.release(); h->lap_out.release(); h->nan_bits.release(); h->quiet_flag.release();
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
    if (h->tv_chk_pinned) (void)hipHostFree(h->tv_chk_pinned);
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
    if (h->hess_companion) { destroy(h->hess_companion); h->hess_companion = nullptr; }
    release_device(h);
    delete h;
}
}  // namespace ssde_engine



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
        return eval_iso(h, par, order, out_dev, s, ra);
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
        for (int i = 0; i < 256; i++) a.p0[i] = h->p0_full[i];
        a.n_dirblocks = h->n_dirblocks; a.dirs = h->dirs.p; a.partials = h->partials.p;
        a.report = nullptr; a.lane_row0 = h->lane_row0.p; a.n = h->n; a.last_dt = h->last_dt;
        if (h->stamps) HIPCHK(h, hipEventRecord(h->ev_k0, s));
        HIPCHK(h, launch_dense(a, order >= 1, s));
        h->last_kernel_id = SSDE_KERNEL_DENSE;
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
            h->last_kernel_id = SSDE_KERNEL_DIRECT_FAST;
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
        h->last_kernel_id = SSDE_KERNEL_DIRECT;
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
    if (!sharded) attach_hess_companion(desc, h);
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
        if (!h->comms.empty() && !h->comm_defer) stw = reduce_ranks(h, out_dev, (hipStream_t)stream);
        if (stw) return stw;
        // a later synchronous ssde_eval runs the parts on their own stream and shares their work buffers with this evaluation
        for (ssde_handle* sh : h->shards) {
            HIPCHK(h, hipEventRecord(sh->ev_async, (hipStream_t)stream));
            sh->async_pending = true;
        }
        return SSDE_OK;
    }
    int st = eval_device(h, par, order, out_dev, (hipStream_t)stream);
    if (st == SSDE_OK && !h->comms.empty() && !h->comm_defer) st = reduce_ranks(h, out_dev, (hipStream_t)stream);     // one ncclAllReduce of 2 + p doubles on the same stream
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
    // (measurement passes -- SSDE_OPT_KERNEL_STAMPS on -- also mark the evaluation's first operation, the end of its finalising
    //  launch and the end of the all-reduce on the stream: ssde_last_phase_ms)
    const bool ph = h->stamps && h->path != PATH_TV;
    const auto ph_t0 = std::chrono::steady_clock::now();
    auto ph_mark = [&](int k) -> hipError_t {
        if (!ph) return hipSuccess;
        if (!h->ev_ph[k]) { hipError_t e = hipEventCreate(&h->ev_ph[k]); if (e != hipSuccess) return e; }
        return hipEventRecord(h->ev_ph[k], 0);
    };
    auto ph_ms = [&](std::chrono::steady_clock::time_point a) { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - a).count(); };
    h->ph_valid = false;
    if (!h->comms.empty()) {
        // the collective sits between the finalize launch and the read-back, on the same stream
        HIPCHK(h, ph_mark(0));
        h->sync_call = true;
        int st = eval_device(h, par, order, h->out.p, 0);
        h->sync_call = false;
        if (st) return st;
        HIPCHK(h, ph_mark(1));
        st = reduce_ranks(h, h->out.p, 0);
        if (st) return st;
        HIPCHK(h, ph_mark(2));
        h->ph_host_enq_ms = ph_ms(ph_t0);
        HIPCHK(h, hipMemcpy(o, h->out.p, nout * 8, hipMemcpyDeviceToHost));
        h->ph_host_total_ms = ph_ms(ph_t0); h->ph_valid = ph; h->ph_has_comm = true;
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
    HIPCHK(h, ph_mark(0));
    h->sync_call = true;
    h->pub_request = true;
    int st = eval_device(h, par, order, h->out.p, 0);
    h->sync_call = false;
    if (st) return st;
    HIPCHK(h, ph_mark(1));
    h->ph_host_enq_ms = ph_ms(ph_t0);
    struct PhDone {                   // (the read-back below has several exits)
        ssde_handle* h; bool ph; std::chrono::steady_clock::time_point t0;
        ~PhDone() { h->ph_host_total_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count(); h->ph_valid = ph; h->ph_has_comm = false; }
    } ph_done{h, ph, ph_t0};
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
    double seen[2] = {-1.0, -1.0};                    // the check values of the last two attempts that each made this call quadruple the warm-up
    for (;; attempt++) {
        int st = run_once(h, par, order, o.data());
        if (st) return st;
        h->last_check = o[1 + h->L.n_full];
        each_engine(h, [&](ssde_handle* e) { e->last_check = h->last_check; });
        // hand-over check of the time windows (k_iso.hip): widen the warm-up and re-evaluate
        // until the windows agree with each other; 64x the estimate ends in one sequential window
        if (h->last_check <= std::max(SSDE_WINDOW_TOL, h->check_floor)) break;
        // ROUNDING FLOOR.  What a short warm-up leaves behind decays with its length; a SMALL disagreement that stays where it is
        // while the warm-up is quadrupled TWICE (16 x the rows) is not the warm-up's: it is rounding in the states themselves (fixes
        // that are very precise against the movement between them: P11 ~ sigma_obs^2 next to P22 ~ 1, the gains 1 - O(sigma_obs^2)),
        // the same in one sequential window, where nothing would notice it.  Such a floor is accepted from here on (up to 4 x what was
        // seen, never beyond 1e-8) and the plan goes back to the warm-up it had -- instead of ending in ONE window per track, two orders
        // of magnitude slower on long tracks, for the same digits.  (A slowly forgetting mode of small amplitude is NOT flat over 16 x
        // the rows: tests/test_gpu_drift.py::test_forced_short_warm_up_is_caught_and_repaired.  The value is the reduced one: shards
        // and ranks decide alike.)
        if (seen[0] > 0.0 && seen[1] > 0.0 && seen[0] <= 1e-8 && seen[1] <= 1e-8 && h->last_check <= 1e-8 &&
            seen[1] >= 0.5 * seen[0] && h->last_check >= 0.5 * seen[1] && std::isfinite(o[0])) {
            h->check_floor = std::min(1e-8, 4.0 * std::max(h->last_check, std::max(seen[0], seen[1])));
            each_engine(h, [](ssde_handle* e) { e->window_boost = std::max(1, e->window_boost / 16); });
            if (!h->shards.empty()) h->window_boost = std::max(1, h->window_boost / 16);
            break;
        }
        // (one window has no hand-over to disagree -- unless quiet rows ran: their switch check is folded into the same
        //  value, and a longer memory, finally none at all, is the repair)
        if (!dist && h->last_chunks <= 1 && h->last_quiet_window == 0) break;
        // a non-finite nllk is rejected by the caller whatever the windows did: no retry, and no lasting
        // widening of the plan because an optimiser probed an absurd parameter once
        if (!std::isfinite(o[0])) break;
        // (a plan that has given up runs one window WITHOUT quiet rows -- eval_iso -- so nothing is left to disagree; the cap is
        //  for whatever that reasoning missed: the failure is then reported through window_check_max instead of spinning -- ADVICE r04)
        if (attempt > 6) break;
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
        const int drift0 = h->shards.empty() ? h->drift : h->shards[0]->drift;
        if ((path0 == PATH_TV || drift0 == 3) && attempt == 0) continue;       // (k_iso_colvar.hip plans the same way)
        if (attempt >= 3) {                                            // give up on windows: sequential filter
            each_engine(h, [](ssde_handle* e) {
                if (!e->gave_up) { e->saved_max_chunks = e->max_chunks; e->saved_want_chunks = e->want_chunks; e->gave_up = true; }
                e->max_chunks = 1; e->want_chunks = 1;
            });
            h->gave_up = true;
        } else {
            seen[0] = seen[1]; seen[1] = h->last_check;
            each_engine(h, [](ssde_handle* e) { e->window_boost *= 4; });
            if (!h->shards.empty()) h->window_boost *= 4;
        }
    }
    if (std::isfinite(o[0]) && !(h->last_check <= h->check_max)) h->check_max = h->last_check;
    // (the floor belongs to the regime that showed it: an evaluation whose windows agree outright ends it, so that a later, genuinely
    //  short warm-up is not waved through under an old allowance)
    if (h->last_check <= SSDE_WINDOW_TOL) h->check_floor = 0.0;
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

namespace ssde_engine {
// SSDE_FLAG_EXACT_HESS: a state-space handle whose own kernels are first-order only -- every register-path (lane = track) handle:
// constant coefficients, a smooth drift, row-varying tau / nu, with H = sigma_obs^2 I or per-row H_array -- keeps its rows a second time in the layout the
// second-order lanes read (the lane = direction path, k_tv_hess.hip), and ssde_hess / ssde_laplace_eval take exact second derivatives
// from there.  ~(TV_RS + 64) x 8 bytes per row: built only when that is at most a third of the device memory still free (a fit of
// 10^8 rows keeps differencing its gradient instead and ssde_info.exact_hess_scope says so).
void attach_hess_companion(const ssde_desc* desc, ssde_handle* h) {
    if (!(desc->flags & SSDE_FLAG_EXACT_HESS) || h->path != PATH_ISO) return;
    size_t free_b = 0, total_b = 0;
    if (hipSetDevice(h->device) != hipSuccess || hipMemGetInfo(&free_b, &total_b) != hipSuccess) return;
    const double need = (double)h->n * (TV_RS + 64) * 8.0;
    if (need > (double)free_b / 3.0) return;
    ssde_handle* c = new (std::nothrow) ssde_handle();
    if (!c) return;
    c->force_tv = true;
    const int stc = build(desc, c);
    if (stc == SSDE_OK && c->path == PATH_TV && !is_eseal(c->model)) h->hess_companion = c;
    else destroy(c);                                     // (not exact there either: ssde_hess says so when asked)
}
}  // namespace ssde_engine

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
    for (int i = 0; i < 256; i++) a.p0[i] = h->p0_full[i];
    a.n_dirblocks = 1; a.dirs = nullptr; a.partials = nullptr;
    a.pp = h->pp_drift;
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
    if (option == SSDE_OPT_COMM_DEFER) {
        // (a multi-device parent holds one communicator per shard: deferring ITS collective would leave a partial sum with no
        //  call that could finish it -- ssde_comm_allreduce speaks for one rank)
        if (!h->shards.empty() && value != 0) { h->err = "SSDE_OPT_COMM_DEFER: only for a handle that joined through ssde_comm_init_rank"; return SSDE_ERR_ARG; }
        h->comm_defer = value != 0;
        return SSDE_OK;
    }
    h->err = "ssde_set_option: unknown option";
    return SSDE_ERR_ARG;
}

int ssde_last_phase_ms(ssde_handle* h, double ms[8]) {
    if (!h || !ms) return SSDE_ERR_ARG;
    for (int k = 0; k < 8; k++) ms[k] = 0.0;
    if (!h->shards.empty()) { h->err = "ssde_last_phase_ms: single-device handles only"; return SSDE_ERR_ARG; }
    if (!h->ph_valid || !h->ev_ph[0] || !h->ev_ph[1]) return SSDE_OK;      // no stamped synchronous evaluation yet: zeros
    HIPCHK(h, hipSetDevice(h->device));
    hipEvent_t last = (h->ph_has_comm && h->ev_ph[2]) ? h->ev_ph[2] : h->ev_ph[1];
    HIPCHK(h, hipEventSynchronize(last));
    auto el = [](hipEvent_t a, hipEvent_t b) { float t = 0.f; return hipEventElapsedTime(&t, a, b) == hipSuccess ? (double)t : -1.0; };
    ms[0] = h->ph_host_total_ms; ms[1] = h->ph_host_enq_ms;
    const bool kv = h->ev_k_valid && h->ev_k0 && h->ev_k1;
    double pre = kv ? el(h->ev_ph[0], h->ev_k0) : -1.0, ker = kv ? el(h->ev_k0, h->ev_k1) : -1.0, fin = kv ? el(h->ev_k1, h->ev_ph[1]) : -1.0;
    if (pre < 0.0 || ker < 0.0 || fin < 0.0) {
        // no stamp pair on this path (or the runtime will not difference a kernel stamp and a stream marker): the whole span as one
        pre = 0.0; ker = 0.0; fin = el(h->ev_ph[0], h->ev_ph[1]);
        if (fin < 0.0) fin = 0.0;
    }
    ms[2] = pre; ms[3] = ker; ms[4] = fin;
    if (h->ph_has_comm && h->ev_ph[2]) { const double ar = el(h->ev_ph[1], h->ev_ph[2]); ms[5] = ar > 0.0 ? ar : 0.0; }
    const double rb = ms[0] - (ms[1] > (ms[2] + ms[3] + ms[4] + ms[5]) ? ms[1] : (ms[2] + ms[3] + ms[4] + ms[5]));
    ms[6] = rb > 0.0 ? rb : 0.0;
    return SSDE_OK;
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
                info->quiet_window = std::max(info->quiet_window, si.quiet_window); info->quiet_share = std::max(info->quiet_share, si.quiet_share);
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
        info->n_devices = h->n_track_shards; info->comm_ranks = h->comm_ranks; info->comm_ranks_reported = h->comm_ranks_reported;
        info->exact_hess_scope = hess_exact_scope(h);
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
    info->hbm_bytes = h->hbm_bytes + (h->hess_companion ? h->hess_companion->hbm_bytes : 0);     // (SSDE_FLAG_EXACT_HESS: the rows a second time)
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
    // row-varying tau / nu: a design column both parameters use is resident (and read) once
    if (h->drift == 3) info->required_bytes_per_row -= 8.0 * (double)(h->n_stream_cols_algo - h->n_stream_cols);
    // a smooth drift evaluated from its blocks' tables: one covariate per block is resident, not the columns
    if (h->pp_drift.nb > 0) info->required_bytes_per_row -= 8.0 * (double)(h->n_stream_cols - h->pp_drift.nb);
    if (h->n_pad > 0) info->required_bytes_per_row *= (double)h->n_pad / (double)h->n;
    if (h->path == PATH_ISO || h->path == PATH_DENSE) {
        info->n_rows_tiled = h->n_pad > 0 ? h->n_pad : h->n;
        info->n_groups = h->n_groups; info->n_clean_groups = h->path == PATH_ISO ? h->n_clean_groups : 0;
        info->quiet_window = h->last_quiet_window; info->quiet_share = h->quiet_share;
    }
    info->n_evals = h->n_evals; info->n_memo_hits = h->n_memo_hits;
    info->n_devices = 1; info->comm_ranks = h->comm_ranks; info->window_check_max = h->check_max;
    info->comm_ranks_reported = h->comm_ranks_reported; info->kernel_id = h->last_kernel_id;
    info->exact_hess_scope = hess_exact_scope(h);
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
