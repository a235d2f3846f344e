// ssde_engine.hpp -- internals shared by the engine's translation units (ssde_engine.hip: C ABI, constant-coefficient
// and direct paths; ssde_engine_tv.hip: the lane = gradient direction path).  Not part of the C ABI.
#ifndef SSDE_ENGINE_HPP
#define SSDE_ENGINE_HPP

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

namespace ssde_engine {

extern thread_local std::string g_create_error;   // message of the last failed ssde_create on this thread

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

}  // namespace ssde_engine

using ssde_engine::DevBuf;
using ssde_engine::PAR_RING;
using namespace ssde;
using namespace ssde_host;

struct ssde_handle {
    std::string err;
    int model = 0, d = 0, q = 0, sdim = 0, na_any = 0, device = 0, path = 0;
    int64_t n = 0, n_seg = 0, n_steps = 0;
    bool has_h = false, const_coeff = false, uniform_dt = false;
    double dt_uniform = 0.0;
    double tdf = 0.0, tconst = 0.0;     // BM_t: degrees of freedom, normalising constant of dt(., df)
    double p0_iso[3] = {0, 0, 0};
    double p0_full[256] = {0};     // sdim x sdim column-major (sdim <= 16)
    bool wide_ok = false;          // set by ssde_create for a response of 3 to 8 columns that has to run as ONE filter (coupling H_array / P0)
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
    int c_obs = 1;                 // tile channel of the first obs column (0: no dt channel, globally regular grid)
    double dt_all = 0.0;
    int64_t tile_doubles = 0;

    // direct families (long format, engine-owned copies)
    DevBuf<double> times, obs, colbuf, tdecay;
    DevBuf<uint32_t> scored;
    DevBuf<const double*> colptr;
    int direct_blocks = 0;

    // random-effect blocks given as piecewise-cubic functions of a covariate (ssde_ppbasis): tables in HBM; either
    // evaluated on the fly by the fast direct kernel (pp_fast) or materialised once into dense columns (pp_mat)
    PPRef pp[MAX_Q] = {};
    bool pp_fast[MAX_Q] = {false, false, false, false};
    DevBuf<double> pp_x[MAX_Q], pp_knots[MAX_Q], pp_tab[MAX_Q], pp_mat[MAX_Q];
    PpDrift pp_drift = {};                     // nb > 0: a smooth drift whose blocks the lanes evaluate from the tables (k_iso_drift_pp.hip); the tiles hold the covariates
    int pp_drift_j[2] = {-1, -1};              // ... the SDE parameter of each block
    bool no_drift_pp = false;                  // (a second build after SSDE_RETRY_WITHOUT_PP)
    bool force_tv = false;                     // the lane = direction path whatever the batch's size (the companion of SSDE_FLAG_EXACT_HESS)
    ssde_handle* hess_companion = nullptr;     // SSDE_FLAG_EXACT_HESS: the same problem on the lane = direction path, for k_tv_hess.hip
    int n_stream_cols_algo = 0;                    // streamed columns of the reference's data contract (algorithmic bytes)

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
    int want_chunks_d = 0;         // mixed batch: windows of the general launch's own plan (0 = one plan for everything)
    int glen_max = 0;              // steps of the longest track group
    double dt_min = 0.0;           // smallest interval used inside a track
    double dt_max = 0.0;           // ... and the largest
    bool chunks_forced = false;    // SSDE_CHUNKS given: the window count is the tester's (1 = plain sequential filter)
    bool stream_nt = true;         // the tile stream is read with non-temporal loads (a batch far larger than the Infinity Cache)
    // quiet rows of the general kernel (k_iso.hip): blocks that hold a missing observation, per group; 0 words = not in use
    DevBuf<unsigned long long> nan_bits;
    DevBuf<double> quiet_flag;
    int nan_words = 0;
    bool quiet_ok = false;
    int env_quiet_window = 0;      // SSDE_QUIET_WINDOW (testing)
    double quiet_share = 0.0;      // share of the dirty groups' blocks that qualify (nominal 128-row memory)
    int last_quiet_window = 0;     // rows of memory the last launch used (0: no quiet rows)
    bool gain_stationary = false;  // the last gain recursion reached its stationary row
    double stat_p[12] = {0}, stat_ld = 0.0, stat_gld[3] = {0, 0, 0};   // ... the covariance, log F and dF / F there
    double plan_rho = 1.0;         // spectral radius of the closed-loop matrix the last plan_windows call found
    int plan_warmup = 0;           // warm-up rows the last plan_windows call found sufficient (0: no usable forgetting)
    int window_boost = 1;          // multiplies the estimated warm-up after a failed hand-over check
    int last_chunks = 1, last_window = 0;
    double last_check = 0.0;
    int n_retries = 0;
    // testing / tuning knobs, read once at create: nothing calls getenv per evaluation
    int env_window = 0, env_tv_waves = 0, env_tv_minlen = 0;
    bool env_no_derive = false, env_no_graph = false, env_no_exact_hess = false;   // (SSDE_NO_EXACT_HESS: difference the gradient even where ssde_hess is exact)
    double env_t0_cost = 3.0;
    double env_w0_ratio = 1.2;     // (measured: 0 .. 1.45 swept, 3 % on CTCRW at 1.2, nothing on the scalar models) cost of a row of window 0 (every direction) over a row of a later window (one derived)
    // recovery from a widened plan (ssde_eval): after `cooldown` evaluations accepted at the first try the boost is
    // halved (or a given-up window plan restored) on probation; a failure on probation restores the level that worked
    // and doubles the cooldown
    int calm = 0, cooldown = 32, probe_from = 0, saved_max_chunks = 0, saved_want_chunks = 0;
    bool probing = false, gave_up = false;

    // shared-covariance path
    DevBuf<int32_t> group_flags;
    DevBuf<int32_t> dirty_groups;   // the groups that hold a track with a missing row (mixed batch: the general launch's)
    int n_dirty_groups = 0;
    int n_clean_groups = 0;
    bool use_shared = false;
    int drift = 0;                 // row-varying drift on the register path (k_iso_drift.hip): 1 = shared-covariance lanes (regular grid, complete
                                   // tracks), 2 = lanes with their own covariance (missing rows / irregular grid)
    int drift_nstate = 0;          // components of its hand-over dumps
    DevBuf<ssde::CvPart> cv_parts;  // drift == 3 (row-varying tau / nu on lane = track lanes, k_iso_colvar.hip): the columns of the four parts
    std::vector<int> cv_pidx;      // [CV_WAVES][CV_KC] full-parameter index of a part's column, -1 = unused slot
    int cv_mu_part = -1, cv_sig_part = -1;   // the parts that carry the drift-intercept / log sigma_obs direction
    int cv_kc = 0;                 // the widest part's column count
    bool cv_full = false;          // 4 x 4 covariance lanes (CTCRW, d = 2, per-row H_array)
    bool cv_single = false;        // ... with constant tau / nu: one wave per (group, window) runs filter and tangents (iso_full_kernel)
    bool cv_few = false;           // few design columns, H = sigma_obs^2 I: one wave per (group, window) too (iso_few_kernel)
    DevBuf<unsigned> fuse_words;   // the fused finalising work of iso_shared_kernel: [0] finished work items, [2..3] the check word, [4..] arrivals per (boundary, group)
    bool env_no_fused = false, last_fused = false;
    int env_adj_tail = 0;          // testing (SSDE_ADJ_TAIL): rows past a window's end before its backward recursion starts
    bool cv_adj = false;           // gradient by a reverse sweep: one wave per (group, window), two passes (iso_adj_kernel)
    DevBuf<double> adj_ckpt;       // ... the state entering every adj_ckpt_rows-th row of every window (grown on demand)
    bool cv_one_wave() const { return cv_single || cv_few || cv_adj; }
    double cv_hmax = 0.0;          // ... the largest diagonal entry of H_array over the batch (the window planner's observation variance)
    std::vector<double> cv_col_lo, cv_col_hi;   // range of every streamed column over the batch (found at create)
    DevBuf<double> cv_ranges;      // [workgroup][4] ranges of the linear predictors seen by the last launch
    double* cv_ranges_pinned = nullptr;   // ... reduced over the launch: min / max of p1, min / max of p2 (host-visible; +inf / -inf before the first launch)
    double cv_eta_lo[2] = {0, 0}, cv_eta_hi[2] = {0, 0};   // range the linear predictors of par[d], par[d + 1] can reach at the last parameters
    // exact Hessian over the drift coefficients (ssde_hess.hip): eval_device launches the Hessian kernels instead of an evaluation
    DevBuf<double> hs_partials, hs_hess;      // scratch of the Hessian passes, kept between calls (allocation costs more than the pass)
    DevBuf<int16_t> hs_i16;
    bool hess_req = false;
    DriftHessArgs hess_args;
    int hess_tiles = 0;
    // one-row tracks never reach a kernel; REPORT(aest_all) still shows their a0 (nllk_ctcrw.hpp:196-200, 246)
    std::vector<int64_t> single_rows;
    std::vector<double> single_a0;       // [single_rows.size()][sdim]
    std::vector<std::pair<int, int64_t>> clean_ns_hist;  // (scored rows, number of tracks) over NaN-free groups
    DevBuf<double> gain_ring;
    double* gain_pinned = nullptr;
    size_t gain_rows_cap = 0;
    int last_gain_rows = 0;
    bool par_ev_pending[PAR_RING] = {false, false, false, false, false, false, false, false};   // slot last used by an asynchronous call
    bool sync_call = false;        // set around the evaluation of a synchronous ssde_eval (single engine, null stream)

    // side streams: the kernels of one evaluation that do not depend on each other run concurrently
    hipStream_t aux[2] = {nullptr, nullptr};
    hipEvent_t ev_fork = nullptr, ev_join[2] = {nullptr, nullptr};
    hipEvent_t ev_async = nullptr;            // end of the last ssde_eval_device on the caller's stream: a synchronous call waits for it
    bool async_pending = false;
    bool env_own_stream = false;

    // timing of the dominant kernel (recorded on the stream it is launched on)
    hipEvent_t ev_k0 = nullptr, ev_k1 = nullptr;   // the CURRENT evaluation's pair: ev_ring[ev_idx % EV_RING]
    bool ev_k_valid = false;
    bool stamps = true;                            // SSDE_OPT_KERNEL_STAMPS: 0 = plain launches, no event pair per evaluation
    // Every evaluation stamps its dominant kernel with a pair of its own, so that a caller can time K evaluations and read
    // the K kernel durations AFTERWARDS (ssde_kernel_ms_history) instead of paying an event query between them.
    static constexpr int EV_RING = 64;
    hipEvent_t ev_ring[EV_RING][2] = {};
    bool ev_ring_valid[EV_RING] = {};
    int64_t ev_idx = -1;
    std::vector<int32_t> glen_host, lane_ns_host;
    int last_s_stat = 0, last_t0 = 0, last_t0_delta = 0;
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
    double* tv_chk_pinned = nullptr;          // [TV_LEAN_ITEMS] the items' hand-over checks of a lean replay (eval_tv_graph)
    bool tv_graph_lean[2] = {false, false};
    bool env_tv_no_lean = false;
    DevBuf<double> lap_out;                   // ssde_laplace_eval: result vectors of a batch of asynchronous evaluations
    double* out_pinned = nullptr;             // read-back target of the synchronous ssde_eval (2 + n_full doubles)
    // publication of a synchronous evaluation's result by its reducing launch (ReduceArgs.pub, ssde_device.hpp): the host spins
    // on a sequence word in pinned memory instead of issuing a read-back copy
    double* pub_pinned = nullptr;             // [2 + n_full] result, then (128-byte aligned) the sequence word
    unsigned long long* pub_flag = nullptr;
    unsigned long long pub_seq = 0;
    DevBuf<unsigned int> pub_count;
    bool pub_ok = false;                      // buffers exist and SSDE_PUBLISH is set (opt-in)
    DevBuf<double> wave_clock;                // SSDE_WAVE_CLOCK=file: per-wave stamps of the last shared-covariance launch, written at destroy
    std::string wave_clock_file;
    int wave_clock_items = 0;
    bool pub_request = false;                 // run_once asks the next eval_device to publish
    std::vector<double> gain_cum[4];          // build_gain_table's running sums
    bool pub_armed = false;                   // the evaluation just enqueued will publish (set by eval_device, consumed by run_once)
    std::vector<double> eval_out;             // ssde_eval's result vector (no allocation per call)

    int64_t hbm_bytes = 0;
    // dtimes(n-1) of the reference is 1 (nllk_ctcrw.hpp:126-129); a shard of a multi-device handle that is not the last
    // one carries the real interval to the next shard's first row there (only REPORT(aest_all) ever shows it)
    double last_dt = 1.0;
    // lattice padding (ssde_engine.hip: lattice_pad): the tiles hold n_pad > n rows; pad_pos[i] = the lattice row whose
    // reported state is caller row i's (ssde_report)
    int64_t n_pad = 0;
    DevBuf<int64_t> pad_pos;
    double pad_step = 0.0;
    double snap_dt = 0.0;          // > 0: a grid that is regular to SSDE_GRID_RTOL (last-bit jitter of decimal steps): this step is hoisted

    // ---- distributed evaluation (ssde_engine_dist.hip) -------------------------------------------------------------
    // single-process multi-GPU parent (ssde_desc.n_devices > 1): one engine per device, this handle owns no device data
    std::vector<ssde_handle*> shards;
    std::vector<int64_t> shard_row0;          // first global row of each engine
    std::vector<int64_t> shard_nrows;         // its rows
    std::vector<int> shard_col0;              // first column of aest_all it reports (dimension parts: 2 or 4 columns each)
    std::vector<int> shard_leader;            // index of the first engine on the same device (its stream and out buffer
                                              // collect that device's engines before the collective)
    int n_track_shards = 1, n_dim_parts = 1;
    std::vector<double> poison_vec;           // the NaN result vector of a poisoned handle (ssde_eval_device)
    bool poison = false;                      // SSDE_NA_ANY_NAN, n_dim > 2: an observed row with a NaN outside column 0
    std::vector<void*> comms;                 // ncclComm_t: one per shard (parent), or one (ssde_comm_init_rank)
    bool shards_share_device = false;         // rehearsal on a one-GPU machine: shards summed by a kernel, not RCCL
    int comm_ranks = 1;                       // ranks of a multi-process communicator (the caller's argument)
    int comm_ranks_reported = 0;              // ... what ncclCommCount says about it (0: no communicator)
    bool comm_defer = false;                  // SSDE_OPT_COMM_DEFER: ssde_eval_device leaves the rank's partial result to the caller's own collective
    int last_kernel_id = 0;                   // SSDE_KERNEL_*: the family that ran the last evaluation's rows
    // where a stamped synchronous evaluation spent its time (ssde_last_phase_ms): events on the evaluation's stream
    hipEvent_t ev_ph[3] = {nullptr, nullptr, nullptr};   // first operation | end of the finalising launch | end of the all-reduce
    bool ph_valid = false, ph_has_comm = false;
    double ph_host_total_ms = 0.0, ph_host_enq_ms = 0.0;
    // Decisions that shape the SEQUENCE of collectives must be the same on every rank, whatever each rank's own data look
    // like (one rank's shard on a regular grid, another's with missing rows): agreed on once, at ssde_comm_init_rank
    // (min over ranks), and used instead of the local facts while a communicator is joined.  -1 = no communicator.
    int comm_rides = -1;                      // an order-0 ssde_eval evaluates order 1 (grad_rides_along) on every rank, or on none
    int comm_async_ok = -1;                   // ssde_laplace_eval batches its evaluations through ssde_eval_device on every rank, or on none
    hipStream_t own_stream = nullptr;         // stream of the synchronous evaluation when a collective follows it
    // ---- memo of the last ssde_eval (include/ssde.h) ---------------------------------------------------------------
    std::vector<double> memo_par, memo_grad;
    double memo_value = 0.0;
    int memo_order = -1;                      // -1 = nothing memoised
    int64_t n_evals = 0, n_memo_hits = 0;
    double check_max = 0.0;                   // largest accepted hand-over disagreement since create
    double check_floor = 0.0;                 // > 0: a disagreement that a 4x longer warm-up did NOT reduce -- rounding in the states, not a short warm-up; accepted up to here
    // host-side phase clock of the isotropic path (SSDE_TRACE=1 at create; printed at destroy): plan, gain table,
    // main launch(es), finalize launch, read-back
    bool trace = false;
    double trace_us[6] = {0, 0, 0, 0, 0, 0};
    int64_t trace_n = 0;
    int trace_skip = 0;
};


namespace ssde_engine {

#define HIPCHK(h, call)                                                                  \
    do {                                                                                 \
        hipError_t e__ = (call);                                                         \
        if (e__ != hipSuccess) {                                                         \
            (h)->err = std::string(#call) + ": " + hipGetErrorString(e__);               \
            return SSDE_ERR_HIP;                                                         \
        }                                                                                \
    } while (0)


int fail(ssde_handle* h, int code, const std::string& msg);

// ssde_engine_build.hip
int build(const ssde_desc* d, ssde_handle* h, const ParLayout* part_layout = nullptr);
void destroy(ssde_handle* h);
void attach_hess_companion(const ssde_desc* desc, ssde_handle* h);     // SSDE_FLAG_EXACT_HESS (ssde_engine.hip)
void release_device(ssde_handle* h);          // everything the handle holds on the device / in pinned memory (the handle stays)
int eval_iso(ssde_handle* h, const double* par, int order, double* out_dev, hipStream_t s, ReduceArgs& ra);   // ssde_engine_iso.hip
// spectral radius of the stationary closed-loop matrix T - K Z at constant parameters p1, p2 (ssde_engine_tv.hip); p0 = {p11, p12, p22} or {p}
double closed_loop_rho(int model, double dt, double p1, double p2, double hobs, const double* p0);
int eval_device(ssde_handle* h, const double* par, int order, double* out_dev, hipStream_t s);

// ---- distributed evaluation (ssde_engine_dist.hip) --------------------------------------------------------------------
int create_sharded(const ssde_desc* d, ssde_handle* parent);
// sum the shards' [nllk, grad..., check] buffers (RCCL all-reduce, or the rehearsal kernel) -- enqueue only
int reduce_shards(ssde_handle* parent);
// all-reduce one handle's out buffer over its multi-process communicator -- enqueue only
int reduce_ranks(ssde_handle* h, double* buf, hipStream_t s);
int report_sharded(ssde_handle* parent, const double* par, double* aest_all);
void destroy_dist(ssde_handle* h);
hipError_t launch_sum_into(double* dst, const double* src, int n, hipStream_t s);   // k_reduce.hip

// ---- exact second derivatives, direct families BM / OU (ssde_hess.hip) -------------------------------------------------
// 0: no exact second derivatives; 1: over the drift coefficients of a shared-covariance drift handle (and log_lambda);
// 2: over every coefficient and log_lambda (direct families BM / OU)
int hess_exact_scope(const ssde_handle* h);
inline bool hess_exact_available(const ssde_handle* h) { return hess_exact_scope(h) == 2; }
int hess_exact(ssde_handle* h, const double* par, const int32_t* idx, int n_idx, double* H);

// copy a caller array (host or device) into a fresh device buffer
template <class T>
inline hipError_t stage(const T* src, size_t count, bool on_device, DevBuf<T>& dst) {
    hipError_t e = dst.alloc(count);
    if (e != hipSuccess || count == 0) return e;
    return hipMemcpy(dst.p, src, count * sizeof(T), on_device ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice);
}


// upload the parameter vector for the dense / direct / tv kernels; returns the device pointer (ssde_engine.hip)
int push_par(ssde_handle* h, const double* par, hipStream_t s, const double** dev);

// ---- lane = gradient direction path (ssde_engine_tv.hip) ------------------------------------------------------------
void tv_base_args(const ssde_handle* h, TvArgs& a);
int build_tv(const ssde_desc* d, ssde_handle* h, const std::vector<int64_t>& starts, bool on_dev);
int tv_plan(ssde_handle* h, double hobs, hipStream_t s);
int eval_tv(ssde_handle* h, const double* par, int order, double* out_dev, hipStream_t s);
int eval_tv_graph(ssde_handle* h, const double* par, int order, double* o_host);

}  // namespace ssde_engine
#endif
