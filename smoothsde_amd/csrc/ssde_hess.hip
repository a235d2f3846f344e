// ssde_hess.hip -- ssde_hess: second derivatives of the joint penalised nllk, the counterpart of tmb_obj_joint$he(x)
// (MakeADHessObject2, src/init.c:13; used at R/sde.R:1363) and of the H_uu block TMB's Laplace approximation needs
// (random = "coeff_re", R/sde.R:510-525, 656-658).
//
// EXACT for the Gaussian direct families BM and OU: their SDE parameters are linear in the coefficients, so the Hessian
// of the data term is X' D X with the closed-form per-row D of k_direct_hess.hip, and the smoothing penalty is a
// quadratic form in coeff_re times exp(log_lambda) (nllk_sde.hpp:91-124): nothing is differenced.  EXACT as well over the
// drift coefficients of a state-space batch on the shared-covariance lanes (a smooth mu, everything else constant, regular
// grid, complete tracks: k_iso_drift.hip): the innovations are linear in them, so the Hessian is sum_i F_i^-1 mx_k mx_l of
// the column recursions the evaluation already runs.  Every other model / entry returns SSDE_ERR_MODEL / SSDE_ERR_ARG: the documented route there is central differences of ssde_eval's gradient (what
// ssde_laplace_eval, smoothsde_amd/report.py and R_glue's he() do).
#include <cmath>
#include <cstring>
#include <vector>

#include "ssde_comm.hpp"
#include "ssde_engine.hpp"

namespace ssde_engine {

// one engine by itself: 3 = every free entry (hyper-dual lanes, k_tv_hess.hip -- its own rows or those of its companion),
// 2 = the coefficients of a direct family (k_direct_hess.hip), 1 = the drift coefficients of a shared-covariance batch, 0 = none
static int scope_one(const ssde_handle* e) {
    if (e->env_no_exact_hess) return 0;
    // SSDE_FLAG_EXACT_HESS: the same rows on the lane = direction path next to a handle whose own kernels are first-order only
    if (e->hess_companion) return scope_one(e->hess_companion);
    if (e->path == PATH_TV) {
        // row-varying (or, as the companion of a constant-coefficient handle, intercept-only) coefficients on the lane = direction
        // path: second-order forward mode over coefficient pairs -- the isotropic lanes, or (per-row H_array, a general P0) the
        // full-covariance step in the same hyper-dual arithmetic; ESEAL_SSM: its scalar lipid-mass filter (HessLaneEseal).
        return 3;
    }
    if (e->path == PATH_ISO) return e->drift == 1 ? 1 : 0;          // a smooth drift on the shared-covariance lanes: QUADRATIC in its coefficients
    if (e->path != PATH_DIRECT) return 0;                           // BM, BM_t, OU, CIR: closed-form (CIR: series + hyper-dual) per-row D
    for (auto& sl : e->slots) if (sl.col == -2) return 0;           // a block evaluated from its basis table has no columns to read
    return 2;
}

int hess_exact_scope(const ssde_handle* h) {
    if (h->n_dim_parts > 1) return 0;
    if (h->shards.empty()) return scope_one(h);
    // whole-track shards over several devices: the batch's Hessian is the sum of the shards' (tracks are independent,
    // nllk_ctcrw.hpp:196-200, 234) -- if every shard has one of the same kind
    const int s0 = scope_one(h->shards[0]);
    for (const ssde_handle* s : h->shards) if (scope_one(s) != s0) return 0;
    return s0;
}

// scratch of a Hessian pass, kept in the handle between calls: three small index arrays (one upload), partials, the result
static int hess_scratch(ssde_handle* h, const std::vector<int16_t>& a, const std::vector<int16_t>& ti, const std::vector<int16_t>& tj,
                        size_t n_partials, size_t n_hess, const int16_t** pa, const int16_t** pti, const int16_t** ptj) {
    std::vector<int16_t> pack(a);
    pack.insert(pack.end(), ti.begin(), ti.end());
    pack.insert(pack.end(), tj.begin(), tj.end());
    if (h->hs_i16.n < pack.size()) { h->hs_i16.release(); HIPCHK(h, h->hs_i16.alloc(std::max<size_t>(pack.size(), 512))); }
    HIPCHK(h, hipMemcpy(h->hs_i16.p, pack.data(), pack.size() * sizeof(int16_t), hipMemcpyHostToDevice));
    if (h->hs_partials.n < n_partials) { h->hs_partials.release(); HIPCHK(h, h->hs_partials.alloc(n_partials)); }
    if (h->hs_hess.n < n_hess) { h->hs_hess.release(); HIPCHK(h, h->hs_hess.alloc(std::max<size_t>(n_hess, 1024))); }
    *pa = h->hs_i16.p; *pti = h->hs_i16.p + a.size(); *ptj = h->hs_i16.p + a.size() + ti.size();
    return SSDE_OK;
}

// drift handle: Hessian of the data term over the mu coefficients idx[] (streamed columns and intercepts), device buffer
static int hess_drift_device(ssde_handle* h, const double* par, const std::vector<int>& idx, const double** hess_dev, hipStream_t s,
                             std::vector<int>& dim_of) {
    HIPCHK(h, hipSetDevice(h->device));
    const int n = (int)idx.size();
    std::vector<int16_t> chan(n, 0);
    dim_of.assign(n, 0);
    for (int k = 0; k < n; k++) {
        const Slot* sl = nullptr;
        for (auto& t : h->slots) if (t.pidx == idx[k]) sl = &t;
        // (SSDE_ERR_MODEL = "not exact here": R's he() and the Laplace layer fall back to differencing the gradient -- an argument
        //  error would stop a fit whose he(x) runs over ALL free parameters, tau / nu intercepts included)
        if (!sl || sl->par_j >= h->d) { h->err = "ssde_hess: on a state-space model only the drift coefficients (and log_lambda) have exact second derivatives"; return SSDE_ERR_MODEL; }
        chan[k] = sl->col >= 0 ? (int16_t)(h->c_obs + h->d + sl->col) : (int16_t)-1;
        dim_of[k] = sl->par_j;
    }
    // the window plan is the evaluation's: evaluate once at these parameters first (answered from the memo after an
    // ssde_eval at the same point), so that a hand-over failure widens the plan before the Hessian pass uses it
    {
        double v;
        std::vector<double> g((size_t)h->L.n_full);
        int st = ssde_eval(h, par, h->L.n_full, 1, &v, g.data());
        if (st) return st;
    }
    const int nt = (n + HESS_T - 1) / HESS_T;
    std::vector<int16_t> ti, tj;
    for (int a = 0; a < nt; a++) for (int b = a; b < nt; b++) { ti.push_back((int16_t)a); tj.push_back((int16_t)b); }
    const int n_tiles = (int)ti.size();
    const int16_t *p_ch, *p_ti, *p_tj;
    int sst = hess_scratch(h, chan, ti, tj, (size_t)n_tiles * HESS_T * HESS_T * (size_t)(h->max_chunks + 1) * h->n_groups, (size_t)n * n, &p_ch, &p_ti, &p_tj);
    if (sst) return sst;
    h->hess_args.n = n; h->hess_args.chan = p_ch; h->hess_args.tile_i = p_ti; h->hess_args.tile_j = p_tj;
    h->hess_args.partials = h->hs_partials.p; h->hess_args.hess = h->hs_hess.p;
    *hess_dev = h->hs_hess.p;
    h->hess_tiles = n_tiles;
    h->hess_req = true;
    if (h->async_pending) { HIPCHK(h, hipStreamWaitEvent(s, h->ev_async, 0)); h->async_pending = false; }
    int st = eval_device(h, par, 1, h->out.p, s);
    h->hess_req = false;
    if (st) return st;
    HIPCHK(h, hipStreamSynchronize(s));
    return SSDE_OK;
}

// data-term Hessian of ONE engine over the coefficients with full-parameter indices idx[] (slots only), left in the
// device buffer `hess_dev` (nu x nu, column-major) on stream s
static int hess_data_device(ssde_handle* h, const double* par, const std::vector<int>& idx, const double** hess_dev, hipStream_t s) {
    HIPCHK(h, hipSetDevice(h->device));
    const int nu = (int)idx.size();
    std::vector<int16_t> uslot(nu, 0);
    for (int k = 0; k < nu; k++) {
        int found = -1;
        for (size_t t = 0; t < h->slots.size(); t++) if (h->slots[t].pidx == idx[k]) found = (int)t;
        if (found < 0 && idx[k] >= h->L.off_decay && idx[k] < h->L.off_decay + h->L.n_decay) { uslot[k] = (int16_t)(-1 - (idx[k] - h->L.off_decay)); continue; }
        if (found < 0) { h->err = "ssde_hess: internal: index without a coefficient slot"; return SSDE_ERR_ARG; }
        uslot[k] = (int16_t)found;
    }
    const int nt = (nu + HESS_T - 1) / HESS_T;
    std::vector<int16_t> ti, tj;
    for (int a = 0; a < nt; a++) for (int b = a; b < nt; b++) { ti.push_back((int16_t)a); tj.push_back((int16_t)b); }
    const int n_tiles = (int)ti.size();
    const int n_blocks = (int)std::max<int64_t>(1, std::min<int64_t>(1024, (h->n + 2047) / 2048));
    const int16_t *p_us, *p_ti, *p_tj;
    int sst = hess_scratch(h, uslot, ti, tj, (size_t)n_tiles * HESS_T * HESS_T * n_blocks, (size_t)nu * nu, &p_us, &p_ti, &p_tj);
    if (sst) return sst;
    *hess_dev = h->hs_hess.p;
    const double* pdev = nullptr;
    int st = push_par(h, par, s, &pdev);
    if (st) return st;
    DirectHessArgs a;
    memset(&a, 0, sizeof(a));
    a.times = h->times.p; a.obs = h->obs.p; a.cols = h->colptr.p; a.scored = h->scored.p; a.n = h->n;
    a.d = h->d; a.model = h->model; a.any_nan = h->na_any; a.slots = h->slot_table.p; a.par = pdev; a.n_slots = (int)h->slots.size();
    a.tdf = h->tdf;
    a.t_decay = h->tdecay.p; a.n_decay = h->L.n_decay; a.off_decay = h->L.off_decay;
    a.nu = nu; a.uslot = p_us; a.tile_i = p_ti; a.tile_j = p_tj; a.partials = h->hs_partials.p; a.hess = h->hs_hess.p;
    HIPCHK(h, launch_direct_hess(a, n_tiles, n_blocks, s));
    HIPCHK(h, hipStreamSynchronize(s));
    return SSDE_OK;
}

// lane = direction handle (row-varying coefficients): data-term Hessian over the DIRECTIONS dk[] (indices into the handle's gradient
// directions: log_sigma_obs and every free coefficient), host matrix Hd (n x n, column-major).  One lane per pair, hyper-dual
// arithmetic (k_tv_hess.hip); the time windows are the evaluation's, with a warm-up of their own and a hand-over check on every
// component -- widened until it passes, finally one sequential window.
static int hess_tv_device(ssde_handle* h, const double* par, const std::vector<int>& dk, std::vector<double>& Hd) {
    HIPCHK(h, hipSetDevice(h->device));
    const int n = (int)dk.size();
    // the plan (and the parameter ranges it is made from) is the evaluation's: evaluate once at these parameters (memoised after an ssde_eval)
    {
        double v;
        std::vector<double> g((size_t)h->L.n_full);
        int st = ssde_eval(h, par, h->L.n_full, 1, &v, g.data());
        if (st) return st;
    }
    std::vector<int16_t> pa, pb;
    for (int a = 0; a < n; a++) for (int b = a; b < n; b++) { pa.push_back((int16_t)dk[a]); pb.push_back((int16_t)dk[b]); }
    const int n_pairs = (int)pa.size(), n_pb = (n_pairs + WAVE - 1) / WAVE;
    hipStream_t s = h->tv_stream ? h->tv_stream : 0;
    const double* pdev = nullptr;
    int st = push_par(h, par, s, &pdev);
    if (st) return st;
    struct Scratch {                 // (freed on every exit)
        DevBuf<int16_t> pairs; DevBuf<double> rec, bnd, part, out; DevBuf<TvItem> items;
        ~Scratch() { pairs.release(); rec.release(); bnd.release(); part.release(); out.release(); items.release(); }
    } sc;
    DevBuf<int16_t>& d_pairs = sc.pairs;
    DevBuf<double>&d_rec = sc.rec, &d_bnd = sc.bnd, &d_part = sc.part, &d_out = sc.out;
    DevBuf<TvItem>& d_items = sc.items;
    {
        std::vector<int16_t> both(pa);
        both.insert(both.end(), pb.begin(), pb.end());
        HIPCHK(h, d_pairs.upload(both));
    }
    HIPCHK(h, d_rec.alloc((size_t)h->n * HESS_RS));
    HIPCHK(h, d_out.alloc((size_t)4 * n_pb * WAVE + 1));
    const int64_t M = h->n_seg;
    int W = h->tv_window > 0 ? 2 * h->tv_window + WIN_ALIGN : 0;      // second-order tangents forget like t^2 rho^t
    if (is_eseal(h->model)) W = 0;                                    // (one sequential window per track, as its evaluation)
    Hd.assign((size_t)n * n, 0.0);
    for (int attempt = 0;; attempt++) {
        // windows per track: as many as keep ~2048 waves busy, each at least two alignment units of scored rows
        std::vector<TvItem> items;
        const int target = h->env_tv_waves > 0 ? h->env_tv_waves : 2048;
        const int nc_cap = (int)std::max<int64_t>(1, (target + M * n_pb - 1) / (M * n_pb));
        for (int64_t t = 0; t < M; t++) {
            const int L = h->tv_ns_host[(size_t)t];
            int nc = 1;
            if (W > 0 && L >= 2 * W) nc = std::max(1, std::min(nc_cap, (L + 2 * WIN_ALIGN - 1) / (2 * WIN_ALIGN)));
            for (int b = 0; b < n_pb; b++)
                for (int c = 0; c < nc; c++) items.push_back({(int32_t)t, c, nc, b});
        }
        d_items.release(); d_bnd.release(); d_part.release();
        HIPCHK(h, d_items.upload(items));
        HIPCHK(h, d_bnd.alloc(items.size() * 2 * HESS_NSTATE * WAVE));
        HIPCHK(h, d_part.alloc(items.size() * 4 * WAVE));
        TvHessArgs a;
        memset(&a, 0, sizeof(a));
        a.times = h->times.p; a.obs = h->obs.p; a.colbuf = h->colbuf.p; a.col_stride = h->col_stride; a.n = h->n;
        a.d = h->d; a.model = h->model; a.any_nan = h->na_any; a.slots = h->slot_table.p; a.n_slots = (int)h->slots.size();
        a.par = pdev; a.rec = d_rec.p; a.wdir = h->tv_wdir.p; a.ndp = h->tv_ndp; a.dirs = h->tv_dirs.p;
        a.trk_row0 = h->tv_row0.p; a.trk_ns = h->tv_ns.p; a.a0 = h->tv_a0.p;
        a.items = d_items.p; a.n_items = (int)items.size(); a.window = W;
        a.pair_a = d_pairs.p; a.pair_b = d_pairs.p + n_pairs; a.n_pairs = n_pairs; a.n_pb = n_pb;
        for (int i = 0; i < 3; i++) a.p0[i] = h->p0_iso[i];
        a.dense = h->tv_dense ? 1 : 0; a.has_h = h->has_h ? 1 : 0; a.h_array = h->tv_harr.p;
        a.eseal_h = h->tv_eh.p; a.eseal_R = h->tv_eR.p;
        for (int i = 0; i < 16; i++) a.p0_full[i] = h->p0_full[i];
        a.last_dt = h->last_dt;
        a.bnd = d_bnd.p; a.part = d_part.p; a.out = d_out.p;
        HIPCHK(h, launch_tv_hess(a, s));
        std::vector<double> out((size_t)4 * n_pb * WAVE + 1);
        HIPCHK(h, hipMemcpyAsync(out.data(), d_out.p, out.size() * 8, hipMemcpyDeviceToHost, s));
        HIPCHK(h, hipStreamSynchronize(s));
        const double check = out.back();
        if (check <= 1e-10 || W == 0) {
            int p = 0;
            for (int a2 = 0; a2 < n; a2++)
                for (int b2 = a2; b2 < n; b2++, p++) {
                    const double v = out[p];                                    // [0][pb][lane]: the mixed second derivative
                    Hd[a2 + (size_t)b2 * n] = v; Hd[b2 + (size_t)a2 * n] = v;
                }
            break;
        }
        // the hand-over of some component disagrees: a longer warm-up, finally one sequential window
        W = attempt >= 3 ? 0 : 4 * W;
        if ((int64_t)2 * W > h->glen_max) W = 0;
    }
    return SSDE_OK;
}

// H (n_idx x n_idx, column-major, host) of the joint penalised nllk over the full-parameter indices idx[]
int hess_exact(ssde_handle* h, const double* par, const int32_t* idx, int n_idx, double* H) {
    const int scope = hess_exact_scope(h);
    if (scope == 0) { h->err = "ssde_hess: exact second derivatives exist for the direct families BM, OU and BM_t (resident design columns, no decaying terms), for the drift coefficients of a smooth-drift state-space batch on a regular grid without missing rows, and for every free parameter of a state-space model evaluated on the lane = direction path (row-varying coefficients, ESEAL_SSM) or created with SSDE_FLAG_EXACT_HESS"; return SSDE_ERR_MODEL; }
    const ParLayout& L = h->L;
    const int np = L.n_full;
    for (int k = 0; k < n_idx; k++)
        if (idx[k] < 0 || idx[k] >= np) { h->err = "ssde_hess: index out of range"; return SSDE_ERR_ARG; }
    {
        // a duplicated entry would get the penalty in one copy and the data term in both
        std::vector<char> seen((size_t)np, 0);
        for (int k = 0; k < n_idx; k++) {
            if (seen[idx[k]]) { h->err = "ssde_hess: duplicate index"; return SSDE_ERR_ARG; }
            seen[idx[k]] = 1;
        }
    }
    for (size_t k = 0; k < (size_t)n_idx * n_idx; k++) H[k] = 0.0;
    // ---- data term: the entries that are coefficients of the linear predictor -------------------------------------
    std::vector<int> cidx, cpos;
    for (int k = 0; k < n_idx; k++) {
        const bool is_coef = (idx[k] >= L.off_fe && idx[k] < L.off_fe + L.n_fe) || (idx[k] >= L.off_re && idx[k] < L.off_re + L.n_re);
        const bool is_lambda = idx[k] >= L.off_lambda && idx[k] < L.off_lambda + L.n_lambda;
        if (is_coef) { cidx.push_back(idx[k]); cpos.push_back(k); }
        else if (scope == 3 && idx[k] == L.off_sig && L.off_sig >= 0) { cidx.push_back(idx[k]); cpos.push_back(k); }   // log_sigma_obs is a direction like any other there
        else if (scope == 3 && is_eseal(h->model) && idx[k] >= 0 && idx[k] <= 2) { cidx.push_back(idx[k]); cpos.push_back(k); }   // log_tau, a1, log_a2 (nllk_e_seal_ssm.hpp:114-116)
        else if (scope == 2 && idx[k] >= L.off_decay && idx[k] < L.off_decay + L.n_decay) { cidx.push_back(idx[k]); cpos.push_back(k); }   // log_decay: an unknown of the direct-family kernel
        else if (!is_lambda) {
            // log_sigma_obs, log_decay: no closed form here -- say so instead of returning zero rows and columns
            h->err = "ssde_hess: exact second derivatives cover coefficients of the linear predictor and log_lambda only (not log_sigma_obs / log_decay)";
            return SSDE_ERR_MODEL;
        }
    }
    const int nu = (int)cidx.size();
    if (nu > MAX_COLS) { h->err = "ssde_hess: more than 96 coefficients"; return SSDE_ERR_ARG; }
    if (nu > 0 && scope == 3) {
        // every requested entry must be one of the handle's gradient directions (a free parameter that reaches the data term);
        // shards / ranks: the sum of theirs
        std::vector<double> Hsum((size_t)nu * nu, 0.0), Hd;
        auto one = [&](ssde_handle* e) -> int {
            ssde_handle* t = e->hess_companion ? e->hess_companion : e;
            std::vector<int> dk(nu);
            for (int k = 0; k < nu; k++) {
                dk[k] = t->tv_dir_of_par[cidx[k]];
                if (dk[k] < 0) { h->err = "ssde_hess: an entry held fixed (par_fixed) has no second derivatives on the row-varying path"; return SSDE_ERR_MODEL; }
            }
            int st = hess_tv_device(t, par, dk, Hd);
            if (st) { h->err = t->err; return st; }
            for (size_t k = 0; k < Hsum.size(); k++) Hsum[k] += Hd[k];
            return SSDE_OK;
        };
        if (h->shards.empty()) {
            int st = one(h);
            if (st) return st;
            if (!h->comms.empty()) {
                // ranks of a communicator: every rank calls ssde_hess (a collective, like ssde_eval)
                HIPCHK(h, hipSetDevice(h->device));
                if (h->hs_hess.n < Hsum.size()) { h->hs_hess.release(); HIPCHK(h, h->hs_hess.alloc(std::max<size_t>(Hsum.size(), 1024))); }
                HIPCHK(h, hipMemcpy(h->hs_hess.p, Hsum.data(), Hsum.size() * 8, hipMemcpyHostToDevice));
                ncclResult_t r = rccl().AllReduce(h->hs_hess.p, h->hs_hess.p, Hsum.size(), ncclDouble, ncclSum, (ncclComm_t)h->comms[0], 0);
                if (r != ncclSuccess) { h->err = std::string("ncclAllReduce: ") + rccl().GetErrorString(r); return SSDE_ERR_HIP; }
                HIPCHK(h, hipMemcpy(Hsum.data(), h->hs_hess.p, Hsum.size() * 8, hipMemcpyDeviceToHost));
            }
        } else {
            int dev_before = 0;
            (void)hipGetDevice(&dev_before);
            for (ssde_handle* sh : h->shards) {
                int st = one(sh);
                if (st) { (void)hipSetDevice(dev_before); return st; }
            }
            (void)hipSetDevice(dev_before);
        }
        for (int a = 0; a < nu; a++)
            for (int b = 0; b < nu; b++) H[cpos[a] + (size_t)cpos[b] * n_idx] = Hsum[a + (size_t)b * nu];
    } else
    if (nu > 0) {
        std::vector<double> Hd((size_t)nu * nu, 0.0), tmp((size_t)nu * nu);
        auto one = [&](ssde_handle* e, bool all_reduce) -> int {
            const double* hd = nullptr;
            hipStream_t s = e->own_stream ? e->own_stream : 0;
            std::vector<int> dim_of;
            int st = scope == 2 ? hess_data_device(e, par, cidx, &hd, s) : hess_drift_device(e, par, cidx, &hd, s, dim_of);
            if (st) { h->err = e->err; return st; }
            if (all_reduce) {
                // ranks of a communicator: the batch's Hessian is the sum of the ranks' (tracks are independent)
                ncclResult_t r = rccl().AllReduce(hd, (void*)hd, (size_t)nu * nu, ncclDouble, ncclSum, (ncclComm_t)h->comms[0], s);
                if (r != ncclSuccess) { h->err = std::string("ncclAllReduce: ") + rccl().GetErrorString(r); return SSDE_ERR_HIP; }
                if (hipStreamSynchronize(s) != hipSuccess) { h->err = "ssde_hess: stream synchronisation failed"; return SSDE_ERR_HIP; }
            }
            hipError_t ce = hipMemcpy(tmp.data(), hd, (size_t)nu * nu * 8, hipMemcpyDeviceToHost);
            if (ce != hipSuccess) { h->err = "ssde_hess: read-back failed"; return SSDE_ERR_HIP; }
            if (scope == 1)                                 // columns that feed different dimensions do not meet in the likelihood
                for (int a = 0; a < nu; a++)
                    for (int b = 0; b < nu; b++) if (dim_of[a] != dim_of[b]) tmp[a + (size_t)b * nu] = 0.0;
            for (size_t k = 0; k < tmp.size(); k++) Hd[k] += tmp[k];
            return SSDE_OK;
        };
        if (h->shards.empty()) {
            int st = one(h, !h->comms.empty());
            if (st) return st;
        } else {
            int dev_before = 0;
            (void)hipGetDevice(&dev_before);
            for (ssde_handle* sh : h->shards) {
                int st = one(sh, false);
                if (st) { (void)hipSetDevice(dev_before); return st; }
            }
            (void)hipSetDevice(dev_before);
        }
        for (int a = 0; a < nu; a++)
            for (int b = 0; b < nu; b++) H[cpos[a] + (size_t)cpos[b] * n_idx] = Hd[a + (size_t)b * nu];
    }
    // ---- smoothing penalty (nllk_sde.hpp:91-124): sum_s [ -Sn/2 log_lambda_s + exp(log_lambda_s)/2 b_s' S_s b_s ] -----
    const Penalty& P = h->pen;
    if (is_eseal(h->model)) {
        // the inverse-gamma priors on sigma(0)^2 and tau^2 (nllk_e_seal_ssm.hpp:212-216, Penalty::eseal_priors):
        // -log prior = ... + 2 (shape + 1) l + scale exp(-2 l) in l = log sigma(0) = sum_k w_k coef_k, resp. l = log tau
        std::vector<int> where(np, -1);
        for (int k = 0; k < n_idx; k++) where[idx[k]] = k;
        const double n = (double)P.eseal_n, nh = (double)(P.eseal_n / 2);
        double ls0 = 0.0;
        for (auto& e : P.eseal_sig0) ls0 += e.second * par[e.first];
        const double c1 = 4.0 * (4.0 * (10.0 * n - 1.0)) * std::exp(-2.0 * ls0), c2 = 4.0 * (nh - 1.0) * std::exp(-2.0 * par[0]);
        for (auto& ea : P.eseal_sig0)
            for (auto& eb : P.eseal_sig0)
                if (where[ea.first] >= 0 && where[eb.first] >= 0) H[where[ea.first] + (size_t)where[eb.first] * n_idx] += c1 * (ea.second * eb.second);
        if (where[0] >= 0) H[where[0] + (size_t)where[0] * n_idx] += c2;
    }
    if (!P.ncol.empty() && (P.include_penalty || is_kalman(h->model) || is_eseal(h->model))) {       // (the Kalman templates ignore include_penalty: Q7)
        std::vector<int> where(np, -1);
        for (int k = 0; k < n_idx; k++) where[idx[k]] = k;
        int start = 0;
        for (size_t s = 0; s < P.ncol.size(); s++) {
            const int n = P.ncol[s];
            const double lam = std::exp(par[L.off_lambda + s]);
            const double* b = par + L.off_re + start;
            const int wl = where[L.off_lambda + (int)s];
            double quad = 0.0;
            std::vector<double> Sb(n, 0.0);                 // (S + S') b / 2
            for (int a = 0; a < n; a++) {
                double sx = 0.0, stx = 0.0;
                for (int c = 0; c < n; c++) { sx += P.S[s][a + (size_t)c * n] * b[c]; stx += P.S[s][c + (size_t)a * n] * b[c]; }
                quad += b[a] * sx;
                Sb[a] = 0.5 * (sx + stx);
            }
            if (wl >= 0) H[wl + (size_t)wl * n_idx] += 0.5 * lam * quad;
            for (int a = 0; a < n; a++) {
                const int wa = where[L.off_re + start + a];
                if (wa < 0) continue;
                if (wl >= 0) { H[wa + (size_t)wl * n_idx] += lam * Sb[a]; H[wl + (size_t)wa * n_idx] += lam * Sb[a]; }
                for (int c = 0; c < n; c++) {
                    const int wc = where[L.off_re + start + c];
                    if (wc >= 0) H[wa + (size_t)wc * n_idx] += 0.5 * lam * (P.S[s][a + (size_t)c * n] + P.S[s][c + (size_t)a * n]);
                }
            }
            start += n;
        }
    }
    return SSDE_OK;
}

}  // namespace ssde_engine

extern "C" int ssde_hess(ssde_handle* h, const double* par, int32_t n_par_full, const int32_t* idx, int32_t n_idx, double* hess) {
    if (!h || !par || !idx || !hess || n_idx < 1) return SSDE_ERR_ARG;
    if (n_par_full != h->L.n_full) { h->err = "parameter vector has the wrong length"; return SSDE_ERR_ARG; }
    return ssde_engine::hess_exact(h, par, idx, n_idx, hess);
}
