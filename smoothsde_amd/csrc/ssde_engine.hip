// ssde_engine.hip -- C ABI (include/ssde.h) of the MI355X nllk engine: descriptor checks,
// segment discovery, one-off upload + re-tiling, per-evaluation launch + reduction, penalty.
//
// Replaces, for the nllk/gradient path only, what TMB's MakeADFunObject / EvalADFunObject do
// for the reference (/root/reference/src/init.c:6-8, R/sde.R:656-669, 694-697).
// There is no CPU evaluation path in this library: without a gfx950 device ssde_create fails.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <numeric>
#include <string>
#include <vector>

#include "../../include/ssde.h"
#include "ssde_device.hpp"
#include "ssde_host.hpp"
#include "ssde_tv.hpp"

using namespace ssde;
using namespace ssde_host;

namespace {

thread_local std::string g_create_error;

enum { PATH_DIRECT = 0, PATH_ISO = 1, PATH_DENSE = 2, PATH_TV = 3 };
constexpr int PAR_RING = 8;
constexpr double SSDE_WINDOW_TOL = 1e-11;  // largest tolerated relative hand-over disagreement

template <class T>
struct DevBuf {
    T* p = nullptr;
    size_t n = 0;
    hipError_t alloc(size_t count) {
        n = count;
        if (count == 0) { p = nullptr; return hipSuccess; }
        return hipMalloc((void**)&p, count * sizeof(T));
    }
    hipError_t upload(const std::vector<T>& v) {
        hipError_t e = alloc(v.size());
        if (e != hipSuccess || v.empty()) return e;
        return hipMemcpy(p, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice);
    }
    void release() { if (p) (void)hipFree(p); p = nullptr; n = 0; }
};

}  // namespace

struct ssde_handle {
    std::string err;
    int model = 0, d = 0, q = 0, sdim = 0, na_any = 0, device = 0, path = 0;
    int64_t n = 0, n_seg = 0, n_steps = 0;
    bool has_h = false, const_coeff = false, uniform_dt = false;
    double dt_uniform = 0.0;
    double tdf = 0.0, tconst = 0.0;     // BM_t: degrees of freedom, normalising constant of dt(., df)
    double p0_iso[3] = {0, 0, 0};
    double p0_full[16] = {0};
    ParLayout L;
    Penalty pen;
    std::vector<Slot> slots;
    int n_stream_cols = 0;
    std::vector<uint8_t> fixed;
    int n_free = 0;

    // Kalman tiles
    DevBuf<double> tiles, a0;
    DevBuf<int64_t> group_off, lane_row0;
    DevBuf<int32_t> group_len, lane_nsteps;
    int n_groups = 0, C = 0;
    int64_t tile_doubles = 0;

    // direct families (long format, engine-owned copies)
    DevBuf<double> times, obs, colbuf, tdecay;
    DevBuf<uint32_t> scored;
    DevBuf<const double*> colptr;
    int direct_blocks = 0;

    // fast direct kernel (<= 2 parameters with streamed columns)
    bool direct_fast = false;
    int64_t col_stride = 0;                        // doubles between consecutive streamed columns
    int df_ja = -1, df_jb = -1;
    std::vector<int> df_pidxA, df_pidxB;          // full-par indices of the streamed coefficients
    int df_icpt[MAX_Q] = {-1, -1, -1, -1};         // full-par index of each parameter's intercept, or -1
    const double *df_colA = nullptr, *df_colB = nullptr;
    bool direct_uniform_dt = false;
    double direct_dt = 0.0;

    // dense / direct parameter plumbing
    DevBuf<SlotTable> slot_table;
    DevBuf<DenseDir> dirs;
    std::vector<DenseDir> dirs_host;
    int n_dirblocks = 0;
    DevBuf<double> par_ring;
    double* par_pinned = nullptr;
    hipEvent_t par_ev[PAR_RING];
    bool par_ev_ok = false;
    int par_next = 0;

    DevBuf<double> partials, out;
    size_t partial_doubles = 0;

    // iso time windows
    DevBuf<double> bnd, chk;
    int max_chunks = 1;            // allocation bound
    int want_chunks = 1;           // planned number of equal windows (the transient window comes on top)
    int glen_max = 0;              // steps of the longest track group
    double dt_min = 0.0;           // smallest interval used inside a track
    bool chunks_forced = false;    // SSDE_CHUNKS given: the window count is the tester's (1 = plain sequential filter)
    int plan_warmup = 0;           // warm-up rows the last plan_windows call found sufficient (0: no usable forgetting)
    int window_boost = 1;          // multiplies the estimated warm-up after a failed hand-over check
    int last_chunks = 1, last_window = 0;
    double last_check = 0.0;
    int n_retries = 0;

    // shared-covariance path
    DevBuf<int32_t> group_flags;
    int n_clean_groups = 0;
    bool use_shared = false;
    std::vector<std::pair<int, int64_t>> clean_ns_hist;  // (scored rows, number of tracks) over NaN-free groups
    DevBuf<double> gain_ring;
    double* gain_pinned = nullptr;
    size_t gain_rows_cap = 0;
    int last_gain_rows = 0;

    // side streams: the kernels of one evaluation that do not depend on each other run concurrently
    hipStream_t aux[2] = {nullptr, nullptr};
    hipEvent_t ev_fork = nullptr, ev_join[2] = {nullptr, nullptr};

    // timing of the dominant kernel (recorded on the stream it is launched on)
    hipEvent_t ev_k0 = nullptr, ev_k1 = nullptr;
    bool ev_k_valid = false;
    std::vector<int32_t> glen_host, lane_ns_host;
    int last_s_stat = 0, last_t0 = 0;
    mutable int rows_key[3] = {-1, -1, -1};
    mutable int64_t rows_cached = 0;

    // iso direction split
    int iso_parts = 1;
    int iso_masks[MAX_PARTS] = {0, 0, 0, 0};
    int iso_free_mask = 0;

    // row-varying isotropic path (k_tv.hip)
    DevBuf<double> tv_rec, tv_wdir, tv_a0, tv_bnd, tv_chk, tv_gval, tv_gdir, tv_stats, tv_harr, tv_eh, tv_eR;
    bool tv_dense = false;         // full-covariance lanes: per-row H_array and / or a P0 that is not block-identical
    DevBuf<TvDir> tv_dirs;
    DevBuf<int64_t> tv_row0;
    DevBuf<int32_t> tv_ns;
    DevBuf<TvItem> tv_items_g, tv_items_v;     // work items of a gradient / a value-only evaluation
    std::vector<int32_t> tv_ns_host;
    int tv_nd = 0, tv_ndp = 0, tv_lpt_shift = 0, tv_nb = 1;
    int tv_n_items_g = 0, tv_n_items_v = 0, tv_window = -1, tv_max_nc = 1;
    size_t tv_items_cap = 0;
    double* tv_stats_pinned = nullptr;
    hipEvent_t tv_stats_ev = nullptr;
    bool tv_stats_valid = false;
    int tv_stats_blocks = 0;
    int16_t tv_dir_of_par[MAX_PAR];
    // hipGraph replay of a synchronous tv evaluation (ssde_eval): upload, pre-pass, statistics read-back, filter,
    // finalize and result read-back are captured once per plan and replayed with ONE launch call
    hipStream_t tv_stream = nullptr;
    hipGraphExec_t tv_gexec[2] = {nullptr, nullptr};    // [order]
    int tv_graph_plan[2] = {-1, -1};                    // plan generation the executable was captured for
    int tv_plan_gen = 0;
    DevBuf<double> tv_par_dev;
    double* tv_par_pinned = nullptr;
    double* tv_out_pinned = nullptr;

    int64_t hbm_bytes = 0;
};

namespace {

#define HIPCHK(h, call)                                                                  \
    do {                                                                                 \
        hipError_t e__ = (call);                                                         \
        if (e__ != hipSuccess) {                                                         \
            (h)->err = std::string(#call) + ": " + hipGetErrorString(e__);               \
            return SSDE_ERR_HIP;                                                         \
        }                                                                                \
    } while (0)

int fail(ssde_handle* h, int code, const std::string& msg) {
    if (h) h->err = msg;
    g_create_error = msg;
    return code;
}

void destroy(ssde_handle* h) {
    if (!h) return;
    h->bnd.release(); h->chk.release(); h->group_flags.release(); h->gain_ring.release();
    if (h->gain_pinned) (void)hipHostFree(h->gain_pinned);
    for (int i = 0; i < 2; i++) { if (h->aux[i]) (void)hipStreamDestroy(h->aux[i]); if (h->ev_join[i]) (void)hipEventDestroy(h->ev_join[i]); }
    if (h->ev_fork) (void)hipEventDestroy(h->ev_fork);
    if (h->ev_k0) (void)hipEventDestroy(h->ev_k0);
    if (h->ev_k1) (void)hipEventDestroy(h->ev_k1);
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
    h->tdecay.release(); h->times.release(); h->obs.release(); h->colbuf.release(); h->scored.release(); h->colptr.release();
    h->slot_table.release(); h->dirs.release(); h->par_ring.release();
    h->partials.release(); h->out.release();
    if (h->par_pinned) (void)hipHostFree(h->par_pinned);
    if (h->par_ev_ok)
        for (int i = 0; i < PAR_RING; i++) (void)hipEventDestroy(h->par_ev[i]);
    delete h;
}

// copy a caller array (host or device) into a fresh device buffer
template <class T>
hipError_t stage(const T* src, size_t count, bool on_device, DevBuf<T>& dst) {
    hipError_t e = dst.alloc(count);
    if (e != hipSuccess || count == 0) return e;
    return hipMemcpy(dst.p, src, count * sizeof(T), on_device ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice);
}

int choose_iso_split(ssde_handle* h) {
    // Which gradient directions are wanted at all
    int m = 0;
    const ParLayout& L = h->L;
    if (!h->fixed[0]) m |= DIR_SIG;
    for (int a = 0; a < h->d; a++)
        if (!h->fixed[L.off_fe + a]) m |= DIR_MU;
    if (!h->fixed[L.off_fe + h->d]) m |= DIR_P1;
    if (h->q > h->d + 1 && !h->fixed[L.off_fe + h->d + 1]) m |= DIR_P2;
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

int push_par(ssde_handle* h, const double* par, hipStream_t s, const double** dev);

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
                                on_dev ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice));
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
        if (h->has_h) { ok = ok && std::isfinite(hi[3]) && hi[3] > 0.0; hobs = hi[3]; }
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
            w = std::max<int64_t>(w, 16);
            if (const char* e = getenv("SSDE_WINDOW")) w = std::max(1, atoi(e));
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
    if (const char* e = getenv("SSDE_TV_WAVES")) target = std::max(1, atoi(e));
    const int nc_cap = (int)std::max<int64_t>(1, (target + n_packs * h->tv_nb - 1) / (n_packs * h->tv_nb));
    std::vector<TvItem> ig, iv;
    int max_nc = 1;
    for (int64_t p = 0; p < n_packs; p++) {
        const int L = h->tv_ns_host[(size_t)p * tpw];
        int nc = 1;
        // As many windows as the chip has room for (nc_cap): with idle SIMDs around, a window may be much
        // shorter than its warm-up -- the redundant warm-up rows run in parallel, the serial chain of a wave
        // is what the evaluation waits for.  SSDE_TV_MINLEN: shortest scored stretch of a window (rows).
        int minlen = 2 * WIN_ALIGN;
        if (const char* e = getenv("SSDE_TV_MINLEN")) minlen = std::max(WIN_ALIGN, atoi(e) / WIN_ALIGN * WIN_ALIGN);
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
    HIPCHK(h, hipEventRecord(h->ev_k0, s));
    HIPCHK(h, launch_tv_filter(a, grad, s));
    HIPCHK(h, hipEventRecord(h->ev_k1, s));
    h->ev_k_valid = true; h->last_s_stat = -1;
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
    if (!h->tv_gexec[ord] || h->tv_graph_plan[ord] != h->tv_plan_gen) {
        if (h->tv_gexec[ord]) { (void)hipGraphExecDestroy(h->tv_gexec[ord]); h->tv_gexec[ord] = nullptr; }
        TvArgs a;
        tv_base_args(h, a);
        a.par = h->tv_par_dev.p; a.h_from_par = 1; a.h = 0.0; a.out = h->out.p; a.window = h->tv_window;
        a.items = ord ? h->tv_items_g.p : h->tv_items_v.p;
        a.n_items = ord ? h->tv_n_items_g : h->tv_n_items_v;
        hipStream_t s = h->tv_stream;
        hipGraph_t g = nullptr;
        HIPCHK(h, hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
        hipError_t e = hipMemcpyAsync(h->tv_par_dev.p, h->tv_par_pinned, (size_t)n_full * 8, hipMemcpyHostToDevice, s);
        if (e == hipSuccess) e = launch_tv_prepare(a, s);
        if (e == hipSuccess) e = hipMemcpyAsync(h->tv_stats_pinned, h->tv_stats.p, (size_t)h->tv_stats_blocks * TV_STATS * 8, hipMemcpyDeviceToHost, s);
        if (e == hipSuccess) e = launch_tv_filter(a, ord == 1, s);
        if (e == hipSuccess) e = launch_tv_finalize(a, s);
        if (e == hipSuccess) e = hipMemcpyAsync(h->tv_out_pinned, h->out.p, (size_t)(2 + n_full) * 8, hipMemcpyDeviceToHost, s);
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
    }
    memcpy(h->tv_par_pinned, par, (size_t)n_full * 8);
    HIPCHK(h, hipGraphLaunch(h->tv_gexec[ord], h->tv_stream));
    HIPCHK(h, hipStreamSynchronize(h->tv_stream));
    memcpy(o_host, h->tv_out_pinned, (size_t)(2 + n_full) * 8);
    h->ev_k_valid = false;                               // no per-kernel timing inside a replayed graph
    h->last_chunks = h->tv_max_nc; h->last_window = h->tv_window;
    return SSDE_OK;
}

int build(const ssde_desc* d, ssde_handle* h) {
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
        for (int64_t sgi = 0; sgi < d->n_seg; sgi++)
            if (d->a0[sgi] != 1.0) return fail(h, SSDE_ERR_ARG, "ESEAL_SSM: the first column of a0 must be 1 (R/sde.R:602)");
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
    h->sdim = state_dim(d->model, d->n_dim);
    h->na_any = d->na_mode == SSDE_NA_ANY_NAN;
    h->has_h = is_kalman(d->model) && d->h_array != nullptr;
    for (int j = 0; j < d->n_par; j++) {
        if (d->ncol_fe[j] < 1) return fail(h, SSDE_ERR_ARG, "every SDE parameter needs at least one fixed-effect column");
        if (!(d->x_fe && d->x_fe[j]) && d->ncol_fe[j] != 1)
            return fail(h, SSDE_ERR_ARG, "x_fe[j] == NULL means intercept-only: ncol_fe[j] must be 1");
        if (d->ncol_re && d->ncol_re[j] > 0 && !(d->x_re && d->x_re[j]))
            return fail(h, SSDE_ERR_ARG, "x_re[j] missing for a parameter with random-effect columns");
    }
    if (d->n_decay > 0) {
        if (is_kalman(d->model)) return fail(h, SSDE_ERR_ARG, "decaying terms are a feature of the direct families (nllk_sde.hpp:47-58)");
        if (d->n_decay > MAX_DECAY) return fail(h, SSDE_ERR_ARG, "more than 4 decay rates");
        if (!d->t_decay || !d->col_decay || !d->ind_decay || d->n_decay_cols < 1) return fail(h, SSDE_ERR_ARG, "t_decay / col_decay / ind_decay missing");
        for (int c = 0; c < d->n_decay_cols; c++)
            if (d->ind_decay[c] < 0 || d->ind_decay[c] >= d->n_decay) return fail(h, SSDE_ERR_ARG, "ind_decay out of range");
    }
    h->L = make_layout(d);
    if (h->L.n_full > MAX_PAR) return fail(h, SSDE_ERR_ARG, "too many parameters for the kernel argument block");
    int nsm = 0;
    for (int s = 0; s < d->n_smooth; s++) nsm += d->smooth_ncol[s];
    if (nsm != h->L.n_re) return fail(h, SSDE_ERR_ARG, "smooth_ncol does not add up to the random-effect columns");
    h->pen.setup(d);
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
    h->n_steps = n - h->n_seg;
    starts.push_back(n);

    HIPCHK(h, h->out.alloc(2 + h->L.n_full));
    HIPCHK(h, hipEventCreate(&h->ev_k0));
    HIPCHK(h, hipEventCreate(&h->ev_k1));

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
        HIPCHK(h, h->colbuf.alloc((size_t)h->col_stride * h->n_stream_cols));
        std::vector<const double*> cp(h->n_stream_cols, nullptr);
        for (auto& s : h->slots)
            if (s.col >= 0) {
                double* dst = h->colbuf.p + (size_t)s.col * h->col_stride;
                HIPCHK(h, hipMemcpy(dst, s.src, (size_t)n * 8, on_dev ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice));
                cp[s.col] = dst;
            }
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
            h->direct_uniform_dt = (dmin == dmax) && std::isfinite(dmin) && !(d->flags & SSDE_FLAG_NO_UNIFORM_DT);
            h->direct_dt = h->direct_uniform_dt ? dmin : 0.0;
            h->uniform_dt = h->direct_uniform_dt;
            mm.release();
            // which parameters have streamed columns (slots are ordered parameter by parameter)
            std::vector<int> streamed_par;
            for (auto& sl : h->slots) {
                if (sl.col < 0) { h->df_icpt[sl.par_j] = sl.pidx; continue; }
                if (streamed_par.empty() || streamed_par.back() != sl.par_j) streamed_par.push_back(sl.par_j);
            }
            bool ok = streamed_par.size() <= 2 && !getenv("SSDE_NO_DIRECT_FAST") && h->L.n_decay == 0;   // decaying columns: generic kernel
            if (ok) {
                for (auto& sl : h->slots) {
                    if (sl.col < 0) continue;
                    const bool isA = sl.par_j == streamed_par[0];
                    auto& pid = isA ? h->df_pidxA : h->df_pidxB;
                    const double*& base = isA ? h->df_colA : h->df_colB;
                    if (pid.empty()) base = h->colbuf.p + (size_t)sl.col * h->col_stride;
                    else if (h->colbuf.p + (size_t)sl.col * h->col_stride != base + pid.size() * (size_t)h->col_stride) ok = false;  // contiguous
                    pid.push_back(sl.pidx);
                }
                if ((int)h->df_pidxA.size() > DIRECT_KCAP || (int)h->df_pidxB.size() > DIRECT_KCAP) ok = false;
            }
            h->direct_fast = ok;
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
        // row-varying coefficients with H = sigma_obs^2 I and a block-identical P0: the tv path
        // everything the constant-coefficient register path does not take: row-varying coefficients (isotropic
        // lanes), per-row H_array or a P0 that is not block-identical (full-covariance lanes)
        const bool tv_ok = !iso_ok && !(d->flags & SSDE_FLAG_FORCE_DENSE) && !getenv("SSDE_NO_TV") &&
                           (double)n * (TV_RS + 64) * 8.0 < 150e9;
        if (tv_ok) {
            h->tv_dense = h->has_h || !p0_is_isotropic(d, h->p0_iso);
            int st = build_tv(d, h, starts, on_dev);
            if (st) return st;
        }
        if (h->path != PATH_TV) {

        // tracks -> lanes: longest first (stable), 64 per wavefront
        const int64_t M = h->n_seg;
        std::vector<int64_t> order(M);
        std::iota(order.begin(), order.end(), 0);
        std::stable_sort(order.begin(), order.end(), [&](int64_t a, int64_t b) {
            return (starts[a + 1] - starts[a]) > (starts[b + 1] - starts[b]);
        });
        h->n_groups = (int)((M + WAVE - 1) / WAVE);
        const int G = h->n_groups;
        h->C = 1 + d->n_dim + (h->has_h ? d->n_dim * d->n_dim : 0) + h->n_stream_cols;
        std::vector<int64_t> lane_row0((size_t)G * WAVE, -1), lane_seg((size_t)G * WAVE, 0), goff(G);
        std::vector<int32_t> lane_ns((size_t)G * WAVE, 0), glen(G);
        int64_t off = 0;
        for (int g = 0; g < G; g++) {
            int32_t mx = 0;
            for (int l = 0; l < WAVE; l++) {
                int64_t t = (int64_t)g * WAVE + l;
                if (t >= M) break;
                int64_t seg = order[t];
                int64_t len = starts[seg + 1] - starts[seg];
                if (len - 1 > INT32_MAX) return fail(h, SSDE_ERR_ARG, "track too long");
                lane_row0[t] = starts[seg];
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
        DevBuf<double> s_times, s_obs, s_h, s_a0, s_cols;
        DevBuf<const double*> s_colptr;
        DevBuf<int64_t> s_lane_seg;
        const double *p_times = d->times, *p_obs = d->obs, *p_h = d->h_array;
        if (!on_dev) {
            HIPCHK(h, stage(d->times, (size_t)n, false, s_times)); p_times = s_times.p;
            HIPCHK(h, stage(d->obs, (size_t)n * d->n_dim, false, s_obs)); p_obs = s_obs.p;
            if (h->has_h) { HIPCHK(h, stage(d->h_array, (size_t)n * d->n_dim * d->n_dim, false, s_h)); p_h = s_h.p; }
        }
        std::vector<const double*> cp(h->n_stream_cols, nullptr);
        if (h->n_stream_cols > 0) {
            if (!on_dev) HIPCHK(h, s_cols.alloc((size_t)n * h->n_stream_cols));
            for (auto& s : h->slots)
                if (s.col >= 0) {
                    if (on_dev) cp[s.col] = s.src;
                    else {
                        double* dst = s_cols.p + (size_t)s.col * n;
                        HIPCHK(h, hipMemcpy(dst, s.src, (size_t)n * 8, hipMemcpyHostToDevice));
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
        ia.cols = s_colptr.p; ia.ncols = h->n_stream_cols; ia.d = d->n_dim; ia.n = n;
        ia.lane_row0 = h->lane_row0.p; ia.lane_nsteps = h->lane_nsteps.p;
        ia.group_off = h->group_off.p; ia.group_len = h->group_len.p;
        ia.n_groups = G; ia.C = h->C; ia.tiles = h->tiles.p; ia.a0 = h->a0.p;
        ia.a0_src = p_a0; ia.lane_seg = s_lane_seg.p; ia.n_seg = h->n_seg;
        ia.sdim = h->sdim; ia.model = d->model; ia.dt_minmax = mm.p; ia.ychunks = ych;
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
        h->dt_uniform = h->uniform_dt ? dmin : 0.0;
        h->dt_min = std::isfinite(dmin) ? dmin : 0.0;
        mm.release(); s_times.release(); s_obs.release(); s_h.release(); s_a0.release(); s_cols.release();
        s_colptr.release(); s_lane_seg.release();
        h->hbm_bytes = h->tile_doubles * 8;

        if (h->path == PATH_ISO) {
            choose_iso_split(h);
            // shared-covariance path: regular grid + groups without missing rows
            HIPCHK(h, h->group_flags.upload(gflags));
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
            // time windows: enough (group, window, part) workgroups for ~2 waves on each of the 1024 SIMDs
            int glmax = 0;
            for (int g = 0; g < G; g++) glmax = std::max(glmax, glen[g]);
            h->glen_max = glmax;
            // one wave per SIMD (1024 work items INCLUDING the padding of the group count to a multiple of 8):
            // a lone wave already issues fp64 at the SIMD's rate, and fewer windows mean fewer warm-up rows
            int want = std::max(1, 1024 / (((G + 7) / 8 * 8) * h->iso_parts));
            if (const char* e = getenv("SSDE_CHUNKS")) { want = atoi(e); h->chunks_forced = true; }   // testing
            h->max_chunks = std::max(1, std::min(want + 1, std::max(1, glmax / (4 * WIN_ALIGN))));
            h->want_chunks = std::max(1, std::min(want, h->max_chunks));
            if (h->use_shared) {
                h->gain_rows_cap = (size_t)glmax + 1;
                HIPCHK(h, h->gain_ring.alloc((size_t)PAR_RING * h->gain_rows_cap * GAIN_ROW));
                HIPCHK(h, hipHostMalloc((void**)&h->gain_pinned, (size_t)PAR_RING * h->gain_rows_cap * GAIN_ROW * 8,
                                        hipHostMallocDefault));
            }
            HIPCHK(h, h->bnd.alloc((size_t)h->iso_parts * h->max_chunks * G * 2 * NSTATE_MAX * WAVE));
            HIPCHK(h, h->chk.alloc((size_t)h->iso_parts * h->max_chunks * G));
            h->partial_doubles = (size_t)MAX_PARTS * h->max_chunks * NACC_MAX * G;
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
    return SSDE_OK;
}

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
    if (h->use_shared && h->model == SSDE_MODEL_CTCRW && rho > 0.97) return;
    W = (int)std::ceil(std::log(1e-18) / std::log(std::max(rho, 1e-300))) + 16;
    W = std::max(W, 16);
    if (const char* e = getenv("SSDE_WINDOW")) W = std::max(1, atoi(e));  // testing: deliberately short overlaps
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
    HIPCHK(h, hipEventSynchronize(h->par_ev[slot]));
    double* host = h->gain_pinned + (size_t)slot * h->gain_rows_cap * GAIN_ROW;
    double* dev = h->gain_ring.p + (size_t)slot * h->gain_rows_cap * GAIN_ROW;
    const int tmax = h->glen_max;                 // rows 0 .. tmax-1 can be asked for
    std::vector<double> cum_ld, cum_g[NDIRP];
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
            r[0] = G.iF; r[1] = G.k;
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
    HIPCHK(h, hipEventRecord(h->par_ev[slot], s));
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

int eval_device(ssde_handle* h, const double* par, int order, double* out_dev, hipStream_t s) {
    HIPCHK(h, hipSetDevice(h->device));
    if (h->path == PATH_TV) return eval_tv(h, par, order, out_dev, s);
    const ParLayout& L = h->L;
    ReduceArgs ra;
    memset(&ra, 0, sizeof(ra));
    ra.n_value_parts = 1; ra.chunks_per_part = 1; ra.chk = nullptr; ra.n_chk = 0;
    for (int i = 0; i < 4; i++) { ra.add[i] = 0.0; ra.add_slot[i] = -1; }
    ra.partials = h->partials.p;
    ra.n_out = 1 + L.n_full;
    ra.out = out_dev;
    for (int k = 0; k < MAX_PAR + 16; k++) ra.map[k] = -1;

    if (h->path == PATH_ISO) {
        IsoArgs a;
        memset(&a, 0, sizeof(a));
        a.tv.tiles = h->tiles.p; a.tv.group_off = h->group_off.p; a.tv.group_len = h->group_len.p;
        a.tv.lane_nsteps = h->lane_nsteps.p; a.tv.a0 = h->a0.p; a.tv.n_groups = h->n_groups; a.tv.C = h->C;
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
        for (int i = 0; i < h->d; i++) a.mu[i] = par[L.off_fe + i];
        for (int i = 0; i < 3; i++) a.p0[i] = h->p0_iso[i];
        const double p1 = par[L.off_fe + h->d];
        const double p2 = (h->q > h->d + 1) ? par[L.off_fe + h->d + 1] : 0.0;
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
        plan_windows(h, a, &a.n_chunks, &a.window);
        a.bnd = h->bnd.p; a.chk = h->chk.p;
        a.chk_out = out_dev + (1 + L.n_full);
        a.derive = getenv("SSDE_NO_DERIVE") ? 0 : 1;
        a.nstate_clean = h->use_shared ? shared_nstate(h->sdim, order >= 1 ? a.part_mask[0] : 0, h->model != SSDE_MODEL_BM_SSM) : 0;
        h->last_chunks = a.n_chunks; h->last_window = a.window;
        a.group_flags = h->group_flags.p;
        a.group_mode = 0;
        double add[4] = {0, 0, 0, 0};
        if (h->use_shared) {
            int st = (h->d == 1) ? build_gain_table<1>(h, a, h->iso_free_mask, s, add)
                                 : build_gain_table<2>(h, a, h->iso_free_mask, s, add);
            if (st) return st;
            a.group_mode = 3;
            // the covariance transient gets its own short window [0, t0): every other window (warm-up
            // included) then lies in the stationary regime and runs the lean kernel
            // A batch with more track groups than SIMDs needs no time windows to fill the chip, but the lean
            // stationary kernel only exists for windows past the covariance transient: split every track into
            // the transient window and ONE stationary window (same wave, so no extra work items)
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
        h->last_t0 = a.t0;
        if (h->use_shared) {
            // two independent launches (NaN-free groups on the shared-covariance kernel, NaN-carrying groups on
            // the general kernel): fork onto a side stream so they share the chip, join before the hand-over check
            IsoArgs b = a;
            b.group_mode = 2;
            const bool any_dirty = h->n_clean_groups < h->n_groups;
            if (any_dirty) {
                HIPCHK(h, hipEventRecord(h->ev_fork, s));
                HIPCHK(h, hipStreamWaitEvent(h->aux[1], h->ev_fork, 0));
                HIPCHK(h, launch_iso(h->model, h->d, a, true, h->aux[1]));
                HIPCHK(h, hipEventRecord(h->ev_join[1], h->aux[1]));
            }
            HIPCHK(h, hipEventRecord(h->ev_k0, s));
            HIPCHK(h, launch_iso_shared(h->model, h->d, b, s));
            HIPCHK(h, hipEventRecord(h->ev_k1, s));
            h->ev_k_valid = true;
            h->last_s_stat = (a.gain_last + SHARED_U - 1) / SHARED_U * SHARED_U;
            if (any_dirty) HIPCHK(h, hipStreamWaitEvent(s, h->ev_join[1], 0));
        } else {
            HIPCHK(h, hipEventRecord(h->ev_k0, s));
            HIPCHK(h, launch_iso(h->model, h->d, a, false, s));
            HIPCHK(h, hipEventRecord(h->ev_k1, s));
            h->ev_k_valid = true;
            h->last_s_stat = -1;
        }
        for (int i = 0; i < 4; i++) { ra.add[i] = add[i]; ra.add_slot[i] = -1; }
        if (h->use_shared) {
            ra.add_slot[0] = 0;
            if (order >= 1) {
                const int pj[NDIRP] = {0, L.off_fe + h->d, L.off_fe + h->d + 1};
                for (int j = 0; j < NDIRP; j++)
                    if (pj[j] < L.n_full && !h->fixed[pj[j]] && (j < 2 || h->q > h->d + 1)) ra.add_slot[1 + j] = (int16_t)(1 + pj[j]);
            }
        }
        const int nacc = 4 + h->d;
        ra.n_parts = a.n_parts * a.n_chunks; ra.nacc = nacc; ra.n_blocks = h->n_groups;
        ra.n_value_parts = a.n_chunks; ra.chunks_per_part = a.n_chunks;
        ra.chk = h->chk.p; ra.n_chk = a.n_chunks > 1 ? a.n_parts * (a.n_chunks - 1) * h->n_groups : 0;
        if (order >= 1) {
            for (int p = 0; p < a.n_parts; p++)
                for (int k = 1; k < nacc; k++) {
                    const int pidx = k - 1;  // accumulators are ordered like the parameter vector
                    if (pidx < L.n_full && !h->fixed[pidx]) ra.map[p * (nacc - 1) + (k - 1)] = (int16_t)(1 + pidx);
                }
        }
        // the hand-over checks and the final sums in one launch
        HIPCHK(h, launch_iso_finalize(h->model, h->d, a, ra, s));
        return SSDE_OK;
    } else if (h->path == PATH_DENSE) {
        const double* pdev = nullptr;
        int st = push_par(h, par, s, &pdev);
        if (st) return st;
        DenseArgs a;
        memset(&a, 0, sizeof(a));
        a.tv.tiles = h->tiles.p; a.tv.group_off = h->group_off.p; a.tv.group_len = h->group_len.p;
        a.tv.lane_nsteps = h->lane_nsteps.p; a.tv.a0 = h->a0.p; a.tv.n_groups = h->n_groups; a.tv.C = h->C;
        a.model = h->model; a.d = h->d; a.any_nan = h->na_any; a.has_h = h->has_h ? 1 : 0;
        a.slots = h->slot_table.p; a.par = pdev; a.n_slots = (int)h->slots.size();
        for (int i = 0; i < 16; i++) a.p0[i] = h->p0_full[i];
        a.n_dirblocks = h->n_dirblocks; a.dirs = h->dirs.p; a.partials = h->partials.p;
        a.report = nullptr; a.lane_row0 = h->lane_row0.p; a.n = h->n;
        HIPCHK(h, hipEventRecord(h->ev_k0, s));
        HIPCHK(h, launch_dense(a, order >= 1, s));
        HIPCHK(h, hipEventRecord(h->ev_k1, s));
        h->ev_k_valid = true; h->last_s_stat = -1;
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
            HIPCHK(h, hipEventRecord(h->ev_k0, s));
            HIPCHK(h, launch_direct_fast(f, s));
            HIPCHK(h, hipEventRecord(h->ev_k1, s));
            h->ev_k_valid = true; h->last_s_stat = -1;
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
        HIPCHK(h, hipEventRecord(h->ev_k0, s));
        HIPCHK(h, launch_direct(a, s));
        HIPCHK(h, hipEventRecord(h->ev_k1, s));
        h->ev_k_valid = true; h->last_s_stat = -1;
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

}  // namespace

extern "C" {

int ssde_abi_version(void) { return SSDE_ABI_VERSION; }

int ssde_create(const ssde_desc* desc, ssde_handle** out) {
    if (!desc || !out) { g_create_error = "NULL argument"; return SSDE_ERR_ARG; }
    *out = nullptr;
    ssde_handle* h = new (std::nothrow) ssde_handle();
    if (!h) { g_create_error = "out of host memory"; return SSDE_ERR_ALLOC; }
    int st = build(desc, h);
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
    return eval_device(h, par, order, out_dev, (hipStream_t)stream);
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

int ssde_eval(ssde_handle* h, const double* par, int32_t n_par_full, int32_t order, double* value, double* grad) {
    if (!h || !par || !value) return SSDE_ERR_ARG;
    if (n_par_full != h->L.n_full) { h->err = "parameter vector has the wrong length"; return SSDE_ERR_ARG; }
    std::vector<double> o(2 + h->L.n_full);
    const bool use_graph = h->path == PATH_TV && !getenv("SSDE_NO_GRAPH");
    for (int attempt = 0;; attempt++) {
        if (use_graph && h->tv_stats_valid) {
            HIPCHK(h, hipSetDevice(h->device));
            int st = eval_tv_graph(h, par, order, o.data());
            if (st) return st;
        } else {
            int st = eval_device(h, par, order, h->out.p, 0);
            if (st) return st;
            HIPCHK(h, hipMemcpy(o.data(), h->out.p, o.size() * 8, hipMemcpyDeviceToHost));
        }
        h->last_check = o[1 + h->L.n_full];
        // hand-over check of the time windows (k_iso.hip): widen the warm-up and re-evaluate
        // until the windows agree with each other; 64x the estimate ends in one sequential window
        if (h->last_check <= SSDE_WINDOW_TOL || h->last_chunks <= 1) break;
        if (attempt >= 3) { h->max_chunks = 1; h->want_chunks = 1; }   // give up on windows: sequential filter
        else h->window_boost *= 4;
        h->n_retries++;
    }
    double pen = 0.0;
    int st;
    if (order >= 1 && grad) {
        for (int k = 0; k < h->L.n_full; k++) grad[k] = o[1 + k];
        st = ssde_penalty(h, par, n_par_full, &pen, grad);
    } else {
        st = ssde_penalty(h, par, n_par_full, &pen, nullptr);
    }
    *value = o[0] + pen;
    return st;
}

int ssde_report(ssde_handle* h, const double* par, int32_t n_par_full, double* aest_all) {
    if (!h || !par || !aest_all) return SSDE_ERR_ARG;
    if (n_par_full != h->L.n_full) { h->err = "parameter vector has the wrong length"; return SSDE_ERR_ARG; }
    if (!is_kalman(h->model)) { h->err = "aest_all is reported by the Kalman families only (the ESEAL template has no REPORT)"; return SSDE_ERR_MODEL; }
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
        h->tv_stats_valid = false;                   // the stats buffer now describes this parameter vector
        ib.release(); rep.release();
        return SSDE_OK;
    }
    // ssde_report always runs the general kernel (value only) and un-tiles on the fly
    DevBuf<double> rep, pbuf;
    DevBuf<SlotTable> stb;
    HIPCHK(h, rep.alloc((size_t)h->n * h->sdim));
    HIPCHK(h, hipMemset(rep.p, 0, (size_t)h->n * h->sdim * 8));
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
    a.tv.lane_nsteps = h->lane_nsteps.p; a.tv.a0 = h->a0.p; a.tv.n_groups = h->n_groups; a.tv.C = h->C;
    a.model = h->model; a.d = h->d; a.any_nan = h->na_any; a.has_h = h->has_h ? 1 : 0;
    a.slots = stb.p; a.par = pbuf.p; a.n_slots = st.n_slots;
    for (int i = 0; i < 16; i++) a.p0[i] = h->p0_full[i];
    a.n_dirblocks = 1; a.dirs = nullptr; a.partials = nullptr;
    a.report = rep.p; a.lane_row0 = h->lane_row0.p; a.n = h->n;
    HIPCHK(h, launch_dense(a, false, 0));
    HIPCHK(h, hipMemcpy(aest_all, rep.p, (size_t)h->n * h->sdim * 8, hipMemcpyDeviceToHost));
    rep.release(); pbuf.release(); stb.release();
    return SSDE_OK;
}

int ssde_widen_windows(ssde_handle* h, int32_t factor) {
    if (!h) return SSDE_ERR_ARG;
    if (factor <= 0) { h->max_chunks = 1; h->want_chunks = 1; }
    else if (h->window_boost < (1 << 20)) h->window_boost *= factor;
    return SSDE_OK;
}

int ssde_info(const ssde_handle* h, ssde_info_t* info) {
    if (!h || !info) return SSDE_ERR_ARG;
    memset(info, 0, sizeof(*info));
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
    info->algo_bytes_per_row = 8.0 * (h->d + 1 + (h->has_h ? h->d * h->d : 0) + h->n_stream_cols);
    if (h->path == PATH_ISO)   // 4-wave workgroups; with a transient window the grid enumerates windows 1.. only
        info->n_kernel_blocks = ((h->n_groups + 7) / 8 * 8 * h->iso_parts * (h->last_t0 > 0 ? h->last_chunks - 1 : h->last_chunks) + WG_WAVES - 1) / WG_WAVES;
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
                window_bounds(L, nc, h->last_window, h->last_t0, c, s_begin, s_acc, s_end);
                for (int l = 0; l < WAVE; l++) {
                    const int ns = h->lane_ns_host[(size_t)g * WAVE + l];
                    rows += std::max(0, std::min(ns, s_end) - s_acc);
                }
            }
        }
        info->main_kernel_rows = rows;
        h->rows_cached = rows;
        h->rows_key[0] = h->last_chunks; h->rows_key[1] = h->last_window; h->rows_key[2] = h->last_s_stat + 100000 * h->last_t0;
    }
    return SSDE_OK;
}

}  // extern "C"
