// ssde_engine_dist.hip -- the engine over several GPUs.
//
// The path shards by tracks and by nothing else: the filter state is re-initialised at every ID change and the
// log-likelihood is a plain sum over rows (/root/reference/src/nllk/nllk_ctcrw.hpp:196-200, 234), so every device
// owns a contiguous block of whole tracks for good and contributes 2 + p doubles per evaluation -- [nllk_data,
// gradient, window check] -- which ONE all-reduce (RCCL over xGMI) sums.  The parameter-only terms (smoothing
// penalty, ESEAL priors) are added once, on the host, after the sum.  Two hosts are served:
//
//   * ONE process, several devices (the reference's host is a single R process, R/sde.R:656-669):
//     ssde_desc.n_devices > 1 -> create_sharded() builds one engine per device from the caller's host arrays,
//     communicators come from ncclCommInitAll, and ssde_eval enqueues every shard's evaluation on its own stream,
//     then one ncclAllReduce per device inside ncclGroupStart/End on those streams, then reads device 0.
//   * one process per GPU (torchrun-style hosts, bench.py --gpus N): every rank creates its own single-device engine
//     and joins them with ssde_comm_unique_id / ssde_comm_init_rank (ncclCommInitRank); ssde_eval and
//     ssde_eval_device then all-reduce on the evaluation's stream before anything is read back.
//
// A one-GPU machine can rehearse the first mode with the same device listed several times: RCCL refuses two ranks on
// one device, so those shards share one stream and a small kernel sums their result vectors in shard order.
#include "ssde_comm.hpp"
#include "ssde_engine.hpp"

using namespace ssde_engine;

namespace ssde_engine {

RcclApi& rccl() {
    static RcclApi api;
    return api;
}

namespace {

#define NCCLCHK(h, call)                                                                        \
    do {                                                                                        \
        ncclResult_t r__ = (call);                                                              \
        if (r__ != ncclSuccess) {                                                               \
            (h)->err = std::string(#call) + ": " + rccl().GetErrorString(r__);                  \
            return SSDE_ERR_HIP;                                                                \
        }                                                                                       \
    } while (0)

// host copies of one shard's rows of every per-row array whose layout is not row-contiguous
struct ShardArrays {
    std::vector<double> obs, a0, t_decay;
    std::vector<std::vector<double>> fe, re;
    std::vector<const double*> fe_ptr, re_ptr;
    std::vector<ssde_ppbasis> pp;
    std::vector<const ssde_ppbasis*> pp_ptr;
};

void copy_cols(const double* src, int64_t n, int64_t lo, int64_t ns, int ncol, std::vector<double>& dst) {
    dst.resize((size_t)ns * ncol);
    for (int c = 0; c < ncol; c++) memcpy(dst.data() + (size_t)c * ns, src + (size_t)c * n + lo, (size_t)ns * 8);
}

// what a dimension part's descriptor points at that is not a plain view into the shard's arrays
struct PartArrays {
    std::vector<int32_t> ncol_fe, ncol_re;
    std::vector<const double*> fe_ptr, re_ptr;
    std::vector<const ssde_ppbasis*> pp_ptr;
    std::vector<double> p0, obs_host, t_decay_host, h_block;
    DevBuf<double> obs_dev, t_decay_dev, h_block_dev;
    DevBuf<int> poison_dev;
};

}  // namespace

// Several engines behind one handle.  Two independent reasons, freely combined:
//   * whole tracks over several devices (ssde_desc.n_devices > 1), as described at the top of this file;
//   * a response wider than two columns: every kernel's register layout is sized for d <= 2, and the likelihood is a
//     plain sum over dimensions whenever nothing couples them -- T, Q, B are block-diagonal in the dimension
//     (nllk_ctcrw.hpp:49-53, 68-72, 86-89; nllk_ou_ssm.hpp:35-66; nllk_bm_ssm.hpp:33-139), H = sigma_obs^2 I, so with a P0
//     that has no cross-dimension entries F is diagonal, log det F = sum_d log F_d and u' F^-1 u = sum_d u_d^2 / F_d; the
//     direct families loop over the dimensions themselves (nllk_sde.hpp:77-84, tr_dens.hpp:27-76).  Columns (2k, 2k+1)
//     become "dimension part" k: an ordinary engine over those columns, its own mu's and the shared parameters, indexed
//     inside the WHOLE parameter vector (child_layout), so that summing the parts' [nllk, gradient, check] vectors -- the
//     very reduction the track shards already use -- gives the whole problem's.  (The reference's det F for n_dim > 2
//     is exp(logdet F), nllk_ctcrw.hpp:20-22: the same number unless the product of the F_d overflows.)
// All engines of one device share one stream and are summed into the first of them (the device's "leader") by a small
// kernel; leaders of different devices are then all-reduced by RCCL.
int create_sharded(const ssde_desc* d, ssde_handle* parent) {
    if (d->abi_version != SSDE_ABI_VERSION) return fail(parent, SSDE_ERR_ARG, "ssde_desc.abi_version mismatch");
    if (d->model < SSDE_MODEL_BM || d->model > SSDE_MODEL_CIR) return fail(parent, SSDE_ERR_MODEL, "Unknown SDE type");
    const bool multi = d->n_devices > 1 && d->devices;
    const bool on_dev = (d->flags & SSDE_FLAG_DEVICE_DATA) != 0;
    if (multi && on_dev)
        return fail(parent, SSDE_ERR_ARG, "a multi-device engine is created from host arrays (no SSDE_FLAG_DEVICE_DATA)");
    if (multi && d->n_devices > 64) return fail(parent, SSDE_ERR_ARG, "n_devices > 64");
    if (d->n < 2 || !d->id || !d->times || !d->obs || !d->ncol_fe) return fail(parent, SSDE_ERR_ARG, "id/times/obs/ncol_fe must be non-NULL");
    if (d->n_dim < 1 || d->n_dim > 64) return fail(parent, SSDE_ERR_MODEL, "n_dim must be between 1 and 64");
    if (d->n_par != n_sde_par(d->model, d->n_dim)) return fail(parent, SSDE_ERR_ARG, "n_par does not match model / n_dim");
    const int D = d->n_dim, q = d->n_par, n_shared = q - D, sdim = state_dim(d->model, D);
    int P = D > 2 ? (D + 1) / 2 : 1;                        // dimension parts (1 again below when the columns couple and every shard runs them as one filter)
    const int per_dim = d->model == SSDE_MODEL_CTCRW ? 2 : 1;   // state components per dimension
    // A measurement covariance or a P0 that couples response columns of different pairs makes F a full matrix: the reference
    // evaluates it through atomic::logdet and F.inverse() (nllk_ctcrw.hpp:12-24, 203-205, 231-241).  For three to eight columns
    // (host arrays, one device) the whole response then runs as ONE filter on the lane = track general kernel (k_dense.hip:
    // F by LU with partial pivoting, ssde_dense.hpp) instead of pair by pair.
    // (several devices: whole-track shards as ever, every shard one filter over all columns -- round 5)
    // (device-resident arrays: one device by construction -- checked above -- and the same route; H is scanned by a kernel)
    const bool can_run_whole = is_kalman(d->model) && D <= DENSE_MAXD;
    bool whole_shards = false;
    auto run_whole = [&]() -> int {
        if (multi) { whole_shards = true; return SSDE_OK; }
        parent->wide_ok = true;
        return build(d, parent);
    };
    if (P > 1) {
        if (is_eseal(d->model)) return fail(parent, SSDE_ERR_MODEL, "ESEAL_SSM takes one response variable");
        if (d->model == SSDE_MODEL_BM_T) return fail(parent, SSDE_ERR_MODEL, "BM_t takes one response variable");
        if (is_kalman(d->model) && d->h_array) {
            // a per-row measurement covariance that is block-diagonal in the column pairs (independent axis errors, one error
            // ellipse per pair) keeps F block-diagonal: log det F and u' F^-1 u stay sums over the pairs, and every part gets
            // its own block of every row.  Anything else couples the parts.
            for (int64_t r = 0; r < (on_dev ? 0 : d->n); r++) {          // (device-resident: checked where the blocks are cut, below)
                const double* Hr = d->h_array + (size_t)r * D * D;
                for (int i = 0; i < D; i++)
                    for (int j = 0; j < D; j++)
                        if (i / 2 != j / 2 && Hr[i + (size_t)j * D] != 0.0) {     // (a NaN entry counts as coupling)
                            if (can_run_whole) { int st = run_whole(); if (!whole_shards) return st; goto coupled_done; }
                            return fail(parent, SSDE_ERR_MODEL, "n_dim > 2 with H_array: H_array[,, i] must not couple response columns of "
                                                                "different pairs (2k, 2k+1), which this engine evaluates side by side "
                                                                "(three to eight columns from host arrays run as one filter)");
                        }
            }
        }
        if (is_kalman(d->model) && d->h_array && on_dev && can_run_whole) {
            // device-resident H_array: the same test by a kernel (k_ingest.hip), before anything is cut into column pairs
            int dev_was = 0;
            (void)hipGetDevice(&dev_was);
            if (d->device >= 0) HIPCHK(parent, hipSetDevice(d->device));
            DevBuf<int> couples;
            HIPCHK(parent, couples.alloc(1));
            HIPCHK(parent, hipMemset(couples.p, 0, sizeof(int)));
            HIPCHK(parent, launch_h_couples(d->h_array, d->n, D, couples.p, 0));
            int flag = 0;
            HIPCHK(parent, hipMemcpy(&flag, couples.p, sizeof(int), hipMemcpyDeviceToHost));
            couples.release();
            (void)hipSetDevice(dev_was);
            if (flag) return run_whole();
        }
        if (is_kalman(d->model) && d->p0)
            for (int i = 0; i < sdim; i++)
                for (int j = 0; j < sdim; j++)
                    if (i / (2 * per_dim) != j / (2 * per_dim) && d->p0[i + (size_t)j * sdim] != 0.0) {
                        if (can_run_whole) { int st = run_whole(); if (!whole_shards) return st; goto coupled_done; }
                        return fail(parent, SSDE_ERR_MODEL, "n_dim > 2: P0 must not couple response columns of different pairs (2k, 2k+1) "
                                                            "(three to eight columns from host arrays run as one filter)");
                    }
    }
coupled_done:
    if (whole_shards) P = 1;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0)
        return fail(parent, SSDE_ERR_NODEVICE, "no HIP device visible: this engine has no CPU fallback");
    int dev_before = 0;
    (void)hipGetDevice(&dev_before);                        // the caller's current device is left as it was found
    struct Restore { int d; ~Restore() { (void)hipSetDevice(d); } } restore{dev_before};
    std::vector<int> devs;
    if (multi) devs.assign(d->devices, d->devices + d->n_devices);
    else devs.push_back(d->device >= 0 ? d->device : dev_before);
    bool distinct = true, same = true;
    for (size_t i = 0; i < devs.size(); i++) {
        if (devs[i] < 0 || devs[i] >= ndev) return fail(parent, SSDE_ERR_ARG, multi ? "devices[]: no such HIP device" : "device: no such HIP device");
        for (size_t k = 0; k < i; k++) distinct = distinct && devs[k] != devs[i];
        same = same && devs[i] == devs[0];
    }
    if (!distinct && !same)
        return fail(parent, SSDE_ERR_ARG, "devices[]: either all different (RCCL) or all the same (one-GPU rehearsal)");

    // ---- the whole problem's parameter layout: what the C ABI reads off the parent itself ----------------------------
    const ParLayout PL = make_layout(d);
    if (PL.n_full > MAX_PAR) return fail(parent, SSDE_ERR_ARG, "too many parameters for the kernel argument block");
    if (P > 1) {
        for (int j = 0; j < q; j++)
            if (d->ncol_fe[j] < 1) return fail(parent, SSDE_ERR_ARG, "every SDE parameter needs at least one fixed-effect column");
        if (d->n_decay > 0) {
            if (is_kalman(d->model)) return fail(parent, SSDE_ERR_ARG, "decaying terms are a feature of the direct families (nllk_sde.hpp:47-58)");
            if (d->n_decay > MAX_DECAY) return fail(parent, SSDE_ERR_ARG, "more than 4 decay rates");
            if (!d->t_decay || !d->col_decay || !d->ind_decay || d->n_decay_cols < 1) return fail(parent, SSDE_ERR_ARG, "t_decay / col_decay / ind_decay missing");
            std::vector<uint8_t> seen((size_t)std::max(PL.n_re, 1), 0);
            for (int c = 0; c < d->n_decay_cols; c++) {
                const int k = d->col_decay[c];
                if (d->ind_decay[c] < 0 || d->ind_decay[c] >= d->n_decay) return fail(parent, SSDE_ERR_ARG, "ind_decay out of range");
                if (k < 0 || k >= PL.n_re) return fail(parent, SSDE_ERR_ARG, "col_decay out of range (0-based index into coeff_re)");
                if (seen[k]) return fail(parent, SSDE_ERR_ARG, "col_decay names a column twice");
                seen[k] = 1;
            }
        }
        int nsm = 0;
        for (int sm = 0; sm < d->n_smooth; sm++) nsm += d->smooth_ncol[sm];
        if (nsm != PL.n_re) return fail(parent, SSDE_ERR_ARG, "smooth_ncol does not add up to the random-effect columns");
    }

    // ---- whole tracks per shard, balanced by rows: cut k is the first ID segment start at or beyond k n / S ----------
    const int64_t n = d->n;
    std::vector<int64_t> starts;
    int64_t n_seg = 0;
    std::vector<int64_t> cut_seg{0};     // segment index where each shard begins
    if (multi) {
        for (int64_t i = 0; i < n; i++)
            if (i == 0 || d->id[i] != d->id[i - 1]) starts.push_back(i);
        n_seg = (int64_t)starts.size();
        if (d->a0 && d->n_seg != n_seg) return fail(parent, SSDE_ERR_ARG, "a0 rows do not match the number of ID segments");
        starts.push_back(n);
        for (int k = 1; k < d->n_devices; k++) {
            const int64_t target = (int64_t)((__int128)n * k / d->n_devices);
            const int64_t sgi = std::lower_bound(starts.begin(), starts.begin() + n_seg, target) - starts.begin();
            if (sgi > cut_seg.back() && sgi < n_seg) cut_seg.push_back(sgi);          // fewer tracks than devices: no empty shards
        }
        cut_seg.push_back(n_seg);
        for (size_t k = 0; k + 1 < cut_seg.size();) {                                 // an engine needs two rows: merge smaller shards
            const int64_t rows = starts[cut_seg[k + 1]] - starts[cut_seg[k]];
            if (rows < 2 && cut_seg.size() > 2) cut_seg.erase(cut_seg.begin() + (k + 2 < cut_seg.size() ? k + 1 : k));
            else k++;
        }
        cut_seg.pop_back();
    }
    const int S = (int)cut_seg.size();
    parent->n_track_shards = S; parent->n_dim_parts = P;

    for (int k = 0; k < S; k++) {
        // ---- the shard's rows: views where the layout allows, copies of the column-major arrays otherwise -----------------
        ShardArrays A;
        ssde_desc sd = *d;
        sd.n_devices = 0; sd.devices = nullptr;
        sd.device = devs[multi ? k : 0];
        int64_t lo = 0, hi = n;
        if (multi) {
            const int64_t seg_lo = cut_seg[k], seg_hi = k + 1 < S ? cut_seg[k + 1] : n_seg;
            lo = starts[seg_lo]; hi = starts[seg_hi];
            const int64_t ns = hi - lo;
            sd.n = ns;
            sd.id = d->id + lo; sd.times = d->times + lo;
            copy_cols(d->obs, n, lo, ns, D, A.obs);
            sd.obs = A.obs.data();
            A.fe.resize(q); A.re.resize(q); A.fe_ptr.assign(q, nullptr); A.re_ptr.assign(q, nullptr);
            for (int j = 0; j < q; j++) {
                if (d->x_fe && d->x_fe[j]) { copy_cols(d->x_fe[j], n, lo, ns, d->ncol_fe[j], A.fe[j]); A.fe_ptr[j] = A.fe[j].data(); }
                if (d->x_re && d->x_re[j] && d->ncol_re && d->ncol_re[j] > 0) {
                    copy_cols(d->x_re[j], n, lo, ns, d->ncol_re[j], A.re[j]); A.re_ptr[j] = A.re[j].data();
                }
            }
            sd.x_fe = d->x_fe ? A.fe_ptr.data() : nullptr;
            sd.x_re = d->x_re ? A.re_ptr.data() : nullptr;
            if (d->basis_re) {
                A.pp.resize(q); A.pp_ptr.assign(q, nullptr);
                for (int j = 0; j < q; j++)
                    if (d->basis_re[j]) { A.pp[j] = *d->basis_re[j]; if (A.pp[j].x) A.pp[j].x += lo; A.pp_ptr[j] = &A.pp[j]; }
                sd.basis_re = A.pp_ptr.data();
            }
            if (d->a0) {
                A.a0.resize((size_t)(seg_hi - seg_lo) * sdim);
                for (int c = 0; c < sdim; c++)
                    for (int64_t g = seg_lo; g < seg_hi; g++) A.a0[(size_t)c * (seg_hi - seg_lo) + (g - seg_lo)] = d->a0[g + (int64_t)c * n_seg];
                sd.a0 = A.a0.data();
            }
            sd.n_seg = seg_hi - seg_lo;
            if (d->h_array) sd.h_array = d->h_array + (size_t)lo * D * D;
            if (d->eseal_h) sd.eseal_h = d->eseal_h + lo;
            if (d->eseal_R) sd.eseal_R = d->eseal_R + lo;
            if (d->n_decay > 0 && d->t_decay) { copy_cols(d->t_decay, n, lo, ns, q, A.t_decay); sd.t_decay = A.t_decay.data(); }
        }
        const int64_t ns = sd.n;

        for (int p = 0; p < P; p++) {
            ssde_handle* sh = new (std::nothrow) ssde_handle();
            if (!sh) return fail(parent, SSDE_ERR_ALLOC, "out of host memory");
            const int e = (int)parent->shards.size();
            parent->shards.push_back(sh);
            if (hi < n) sh->last_dt = d->times[hi] - d->times[hi - 1];   // the reference's dtimes at this row: the next track's first time
            parent->shard_row0.push_back(lo);
            parent->shard_nrows.push_back(ns);
            int leader = e;
            for (int k2 = 0; k2 < e; k2++)
                if (parent->shards[k2]->device == sd.device || (same && k2 == 0)) { leader = parent->shard_leader[k2]; break; }
            int st;
            if (P == 1) {
                parent->shard_col0.push_back(0);
                sh->wide_ok = whole_shards;
                st = build(&sd, sh);
                if (st == SSDE_OK) attach_hess_companion(&sd, sh);      // (SSDE_FLAG_EXACT_HESS: every track shard keeps its own second copy)
            } else {
                // ---- dimension part p: response columns [dlo, dlo + cnt), SDE parameters mu_dlo.., then the shared ones ------
                const int dlo = 2 * p, cnt = std::min(2, D - dlo), qc = cnt + n_shared;
                parent->shard_col0.push_back(dlo * per_dim);
                std::vector<int> jmap;
                for (int a = 0; a < cnt; a++) jmap.push_back(dlo + a);
                for (int j = D; j < q; j++) jmap.push_back(j);
                PartArrays B;
                ssde_desc cd = sd;
                cd.n_dim = cnt; cd.n_par = qc;
                cd.obs = sd.obs + (size_t)dlo * ns;
                for (int j : jmap) {
                    B.ncol_fe.push_back(sd.ncol_fe[j]);
                    B.fe_ptr.push_back(sd.x_fe ? sd.x_fe[j] : nullptr);
                    B.ncol_re.push_back(sd.ncol_re ? sd.ncol_re[j] : 0);
                    B.re_ptr.push_back(sd.x_re ? sd.x_re[j] : nullptr);
                    B.pp_ptr.push_back(sd.basis_re ? sd.basis_re[j] : nullptr);
                }
                cd.ncol_fe = B.ncol_fe.data(); cd.x_fe = sd.x_fe ? B.fe_ptr.data() : nullptr;
                cd.ncol_re = sd.ncol_re ? B.ncol_re.data() : nullptr; cd.x_re = sd.x_re ? B.re_ptr.data() : nullptr;
                cd.basis_re = sd.basis_re ? B.pp_ptr.data() : nullptr;
                const int sc = cnt * per_dim, s0c = dlo * per_dim;               // the part's state columns inside the whole state
                if (sd.a0) cd.a0 = sd.a0 + (size_t)s0c * sd.n_seg;                // (a0 is column-major: a contiguous range of columns)
                if (is_kalman(d->model) && d->p0) {
                    B.p0.resize((size_t)sc * sc);
                    for (int i = 0; i < sc; i++)
                        for (int j = 0; j < sc; j++) B.p0[i + (size_t)j * sc] = d->p0[(s0c + i) + (size_t)(s0c + j) * sdim];
                    cd.p0 = B.p0.data();
                }
                if (is_kalman(d->model) && sd.h_array && on_dev) {                 // the same on the device
                    HIPCHK(parent, hipSetDevice(sd.device));
                    if (p == 0) {
                        DevBuf<int> couples;
                        HIPCHK(parent, couples.alloc(1));
                        HIPCHK(parent, hipMemset(couples.p, 0, sizeof(int)));
                        HIPCHK(parent, launch_h_couples(sd.h_array, ns, D, couples.p, 0));
                        int flag = 0;
                        HIPCHK(parent, hipMemcpy(&flag, couples.p, sizeof(int), hipMemcpyDeviceToHost));
                        couples.release();
                        if (flag) return fail(parent, SSDE_ERR_MODEL, "n_dim > 2 with H_array: H_array[,, i] must not couple response columns of "
                                                                      "different pairs (2k, 2k+1), which this engine evaluates side by side");
                    }
                    HIPCHK(parent, B.h_block_dev.alloc((size_t)cnt * cnt * ns));
                    HIPCHK(parent, launch_h_block(sd.h_array, ns, D, dlo, cnt, B.h_block_dev.p, 0));
                    HIPCHK(parent, hipDeviceSynchronize());
                    cd.h_array = B.h_block_dev.p;
                } else
                if (is_kalman(d->model) && sd.h_array) {                           // the part's cnt x cnt block of every row
                    B.h_block.resize((size_t)cnt * cnt * ns);
                    for (int64_t r = 0; r < ns; r++)
                        for (int jj = 0; jj < cnt; jj++)
                            for (int ii = 0; ii < cnt; ii++)
                                B.h_block[(size_t)r * cnt * cnt + ii + (size_t)jj * cnt] = sd.h_array[(size_t)r * D * D + (dlo + ii) + (size_t)(dlo + jj) * D];
                    cd.h_array = B.h_block.data();
                }
                HIPCHK(parent, hipSetDevice(sd.device));
                if (d->n_decay > 0 && sd.t_decay) {                               // [q x n] -> the part's [qc x n]
                    if (on_dev) {
                        HIPCHK(parent, B.t_decay_dev.alloc((size_t)qc * ns));
                        for (int jc = 0; jc < qc; jc++)
                            HIPCHK(parent, hipMemcpy(B.t_decay_dev.p + (size_t)jc * ns, sd.t_decay + (size_t)jmap[jc] * ns, (size_t)ns * 8, hipMemcpyDeviceToDevice));
                        cd.t_decay = B.t_decay_dev.p;
                    } else {
                        B.t_decay_host.resize((size_t)qc * ns);
                        for (int jc = 0; jc < qc; jc++) memcpy(B.t_decay_host.data() + (size_t)jc * ns, sd.t_decay + (size_t)jmap[jc] * ns, (size_t)ns * 8);
                        cd.t_decay = B.t_decay_host.data();
                    }
                }
                if (is_kalman(d->model) && p > 0) {
                    // "missing" is decided on column 0 of the WHOLE response (nllk_ctcrw.hpp:214; na_follow_kernel, k_ingest.hip)
                    const int any_nan = d->na_mode == SSDE_NA_ANY_NAN;
                    if (on_dev) {
                        HIPCHK(parent, B.obs_dev.alloc((size_t)cnt * ns));
                        HIPCHK(parent, hipMemcpy(B.obs_dev.p, cd.obs, (size_t)cnt * ns * 8, hipMemcpyDeviceToDevice));
                        HIPCHK(parent, B.poison_dev.alloc(1));
                        HIPCHK(parent, hipMemset(B.poison_dev.p, 0, sizeof(int)));
                        HIPCHK(parent, launch_na_follow(sd.id, sd.obs, B.obs_dev.p, ns, any_nan, B.poison_dev.p, 0));
                        int flag = 0;
                        HIPCHK(parent, hipMemcpy(&flag, B.poison_dev.p, sizeof(int), hipMemcpyDeviceToHost));
                        parent->poison = parent->poison || flag != 0;
                        cd.obs = B.obs_dev.p;
                    } else {
                        B.obs_host.assign(cd.obs, cd.obs + (size_t)cnt * ns);
                        uint64_t na_bits = 0x7FF00000000007A2ull, nan_bits = 0x7FF8000000000000ull;
                        double na_r, nan_plain;
                        memcpy(&na_r, &na_bits, 8); memcpy(&nan_plain, &nan_bits, 8);
                        for (int64_t i = 1; i < ns; i++) {
                            if (sd.id[i] != sd.id[i - 1]) continue;
                            const double v = B.obs_host[i];
                            if (is_na(sd.obs[i], any_nan)) B.obs_host[i] = na_r;
                            else if (v != v) {
                                if (any_nan) parent->poison = true;
                                else B.obs_host[i] = nan_plain;
                            }
                        }
                        cd.obs = B.obs_host.data();
                    }
                }
                const ParLayout CL = child_layout(PL, jmap);
                st = build(&cd, sh, &CL);
                B.obs_dev.release(); B.t_decay_dev.release(); B.poison_dev.release(); B.h_block_dev.release();
            }
            if (st != SSDE_OK)
                return fail(parent, st, (S > 1 ? "shard " + std::to_string(k) + " " : std::string()) + (P > 1 ? "dimension part " + std::to_string(p) + " " : std::string()) +
                                        "(device " + std::to_string(sd.device) + "): " + sh->err);
            parent->shard_leader.push_back(leader);
            if (leader != e) sh->own_stream = parent->shards[leader]->own_stream;       // one stream per device orders its engines and their sum
            else HIPCHK(parent, hipStreamCreateWithFlags(&sh->own_stream, hipStreamNonBlocking));
        }
    }
    parent->shards_share_device = same;

    // what the C ABI reads off the parent itself: parameter layout, penalty, fixed mask, sizes
    const ssde_handle* s0 = parent->shards[0];
    parent->model = s0->model; parent->d = D; parent->q = q; parent->sdim = sdim; parent->path = s0->path;
    parent->L = PL;
    if (P == 1) parent->pen = s0->pen;
    else parent->pen.setup(d);
    parent->fixed = s0->fixed; parent->n_free = s0->n_free;         // (every engine holds the whole vector's mask)
    parent->pen.eseal_n = is_eseal(d->model) ? n : parent->pen.eseal_n;   // the ESEAL priors count ALL rows (nllk_e_seal_ssm.hpp:212-216)
    parent->n = n;
    parent->n_seg = 0;
    for (int e = 0; e < (int)parent->shards.size(); e += P) parent->n_seg += parent->shards[e]->n_seg;
    parent->n_steps = n - parent->n_seg;
    parent->device = s0->device;

    if (!same) {
        if (!rccl().load()) return fail(parent, SSDE_ERR_HIP, rccl().err);
        std::vector<ncclComm_t> cs(S);
        std::vector<int> ldev(S);
        for (int k = 0; k < S; k++) ldev[k] = parent->shards[(size_t)k * P]->device;
        NCCLCHK(parent, rccl().CommInitAll(cs.data(), S, ldev.data()));
        for (int k = 0; k < S; k++) parent->comms.push_back((void*)cs[k]);
    }
    return SSDE_OK;
}

int reduce_shards(ssde_handle* parent) {
    const size_t count = 2 + (size_t)parent->L.n_full;
    // the engines of one device into its leader (they share the leader's stream: everything is ordered behind their kernels)
    for (size_t e = 0; e < parent->shards.size(); e++) {
        const int ld = parent->shard_leader[e];
        if (ld == (int)e) continue;
        ssde_handle* lead = parent->shards[ld];
        HIPCHK(parent, hipSetDevice(lead->device));
        HIPCHK(parent, launch_sum_into(lead->out.p, parent->shards[e]->out.p, (int)count, lead->own_stream));
    }
    if (parent->shards_share_device) {
        // a rank communicator joined on top of the dimension parts of ONE track shard (process-per-GPU hosts with a wide response)
        if (!parent->comms.empty()) return reduce_ranks(parent, parent->shards[0]->out.p, parent->shards[0]->own_stream);
        return SSDE_OK;
    }
    if (parent->comms.size() < 2) return SSDE_OK;
    NCCLCHK(parent, rccl().GroupStart());
    for (size_t k = 0; k < parent->comms.size(); k++) {
        ssde_handle* sh = parent->shards[k * (size_t)parent->n_dim_parts];
        NCCLCHK(parent, rccl().AllReduce(sh->out.p, sh->out.p, count, ncclDouble, ncclSum, (ncclComm_t)parent->comms[k], sh->own_stream));
    }
    NCCLCHK(parent, rccl().GroupEnd());
    return SSDE_OK;
}

int reduce_ranks(ssde_handle* h, double* buf, hipStream_t s) {
    if (h->comms.empty()) return SSDE_OK;
    NCCLCHK(h, rccl().AllReduce(buf, buf, 2 + (size_t)h->L.n_full, ncclDouble, ncclSum, (ncclComm_t)h->comms[0], s));
    return SSDE_OK;
}

int report_sharded(ssde_handle* parent, const double* par, double* aest_all) {
    const int64_t n = parent->n;
    for (size_t k = 0; k < parent->shards.size(); k++) {
        ssde_handle* sh = parent->shards[k];
        const int64_t lo = parent->shard_row0[k], ns = parent->shard_nrows[k];
        std::vector<double> tmp((size_t)ns * sh->sdim);
        int st = ssde_report(sh, par, parent->L.n_full, tmp.data());
        if (st) { parent->err = sh->err; return st; }
        for (int c = 0; c < sh->sdim; c++)
            memcpy(aest_all + (size_t)(parent->shard_col0[k] + c) * n + lo, tmp.data() + (size_t)c * ns, (size_t)ns * 8);
    }
    return SSDE_OK;
}

void destroy_dist(ssde_handle* h) {
    for (void* c : h->comms)
        if (c && rccl().CommDestroy) (void)rccl().CommDestroy((ncclComm_t)c);
    h->comms.clear();
    for (size_t k = 0; k < h->shards.size(); k++)
        if (k < h->shard_leader.size() && h->shard_leader[k] != (int)k) h->shards[k]->own_stream = nullptr;   // owned by the device's leader
    for (ssde_handle* sh : h->shards) destroy(sh);
    h->shards.clear();
    if (h->own_stream) { (void)hipSetDevice(h->device); (void)hipStreamDestroy(h->own_stream); h->own_stream = nullptr; }
}

}  // namespace ssde_engine

extern "C" {

int ssde_comm_unique_id(void* id128) {
    if (!id128) return SSDE_ERR_ARG;
    if (!rccl().load()) { g_create_error = rccl().err; return SSDE_ERR_HIP; }
    ncclUniqueId id;
    ncclResult_t r = rccl().GetUniqueId(&id);
    if (r != ncclSuccess) { g_create_error = std::string("ncclGetUniqueId: ") + rccl().GetErrorString(r); return SSDE_ERR_HIP; }
    static_assert(sizeof(ncclUniqueId) == SSDE_COMM_ID_BYTES, "ncclUniqueId size");
    memcpy(id128, &id, sizeof(id));
    return SSDE_OK;
}

int ssde_comm_allreduce(ssde_handle* h, double* buf_dev, int64_t count, void* stream) {
    if (!h || !buf_dev || count < 1) return SSDE_ERR_ARG;
    if (h->comms.empty()) { h->err = "ssde_comm_allreduce: the handle has joined no communicator"; return SSDE_ERR_ARG; }
    if (!h->shards.empty()) { h->err = "ssde_comm_allreduce: a multi-device parent reduces inside ssde_eval (one communicator per shard)"; return SSDE_ERR_ARG; }
    HIPCHK(h, hipSetDevice(h->device));
    NCCLCHK(h, rccl().AllReduce(buf_dev, buf_dev, (size_t)count, ncclDouble, ncclSum, (ncclComm_t)h->comms[0], (hipStream_t)stream));
    return SSDE_OK;
}

int ssde_comm_init_rank(ssde_handle* h, int32_t n_ranks, int32_t rank, const void* id128) {
    if (!h || !id128 || n_ranks < 1 || rank < 0 || rank >= n_ranks) return SSDE_ERR_ARG;
    if ((!h->shards.empty() && h->n_track_shards > 1) || !h->comms.empty()) { h->err = "ssde_comm_init_rank: the handle already evaluates over several devices"; return SSDE_ERR_ARG; }
    if (!rccl().load()) { h->err = rccl().err; return SSDE_ERR_HIP; }
    HIPCHK(h, hipSetDevice(h->device));
    ncclUniqueId id;
    memcpy(&id, id128, sizeof(id));
    ncclComm_t c = nullptr;
    NCCLCHK(h, rccl().CommInitRank(&c, n_ranks, id, rank));
    h->comms.push_back((void*)c);
    h->comm_ranks = n_ranks;
    {
        // what the communicator itself says (ssde_info.comm_ranks_reported): a bench line that claims N ranks shows RCCL's count
        int cnt = 0;
        NCCLCHK(h, rccl().CommCount(c, &cnt));
        h->comm_ranks_reported = cnt;
    }
    if (h->shards.empty() && !h->own_stream) HIPCHK(h, hipStreamCreateWithFlags(&h->own_stream, hipStreamNonBlocking));
    h->memo_order = -1;
    // ---- what the ranks have to agree on (collective: every rank is here) ------------------------------------------
    // [0] the gradient rides along with the value on this rank's kernels (grad_rides_along, ssde_engine.hip)
    // [1] this rank can batch evaluations through ssde_eval_device (ssde_laplace.hip: joint_batch)
    // -> min over ranks.  ESEAL_SSM's priors are functions of the WHOLE data (nllk_e_seal_ssm.hpp:212-216: the total row
    // count n and sigma(0), the first row's sigma): [2] rows -> sum; [3..] rank 0's first design row of sigma -> sum of
    // (rank 0's row, zeros elsewhere).
    {
        const ssde_handle* e = h->shards.empty() ? h : h->shards[0];
        const size_t ns0 = e->pen.eseal_sig0.size();
        std::vector<double> fl = {(e->path == PATH_DIRECT || (e->path == PATH_ISO && e->use_shared)) ? 1.0 : 0.0,
                                  e->path == PATH_TV ? 0.0 : 1.0};
        std::vector<double> sm(1 + ns0, 0.0);
        sm[0] = (double)h->n;
        if (rank == 0) for (size_t k = 0; k < ns0; k++) sm[1 + k] = e->pen.eseal_sig0[k].second;
        DevBuf<double> dfl, dsm;
        HIPCHK(h, dfl.upload(fl));
        HIPCHK(h, dsm.upload(sm));
        hipStream_t s = h->shards.empty() ? h->own_stream : h->shards[0]->own_stream;
        NCCLCHK(h, rccl().AllReduce(dfl.p, dfl.p, fl.size(), ncclDouble, ncclMin, c, s));
        if (is_eseal(h->model)) NCCLCHK(h, rccl().AllReduce(dsm.p, dsm.p, sm.size(), ncclDouble, ncclSum, c, s));
        HIPCHK(h, hipStreamSynchronize(s));
        HIPCHK(h, hipMemcpy(fl.data(), dfl.p, fl.size() * 8, hipMemcpyDeviceToHost));
        HIPCHK(h, hipMemcpy(sm.data(), dsm.p, sm.size() * 8, hipMemcpyDeviceToHost));
        dfl.release(); dsm.release();
        h->comm_rides = fl[0] > 0.5 ? 1 : 0;
        h->comm_async_ok = fl[1] > 0.5 ? 1 : 0;
        if (is_eseal(h->model)) {
            h->pen.eseal_n = (int64_t)llround(sm[0]);
            for (size_t k = 0; k < ns0 && k < h->pen.eseal_sig0.size(); k++) h->pen.eseal_sig0[k].second = sm[1 + k];
        }
    }
    return SSDE_OK;
}

}  // extern "C"
