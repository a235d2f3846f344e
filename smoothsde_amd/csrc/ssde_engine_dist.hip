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

}  // namespace

int create_sharded(const ssde_desc* d, ssde_handle* parent) {
    if (d->flags & SSDE_FLAG_DEVICE_DATA)
        return fail(parent, SSDE_ERR_ARG, "a multi-device engine is created from host arrays (no SSDE_FLAG_DEVICE_DATA)");
    if (d->n_devices > 64) return fail(parent, SSDE_ERR_ARG, "n_devices > 64");
    if (d->n < 2 || !d->id || !d->times || !d->obs || !d->ncol_fe) return fail(parent, SSDE_ERR_ARG, "id/times/obs/ncol_fe must be non-NULL");
    if (d->n_dim < 1 || d->n_dim > 2) return fail(parent, SSDE_ERR_MODEL, "n_dim must be 1 or 2 (wider responses are outside this engine's kernels)");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0)
        return fail(parent, SSDE_ERR_NODEVICE, "no HIP device visible: this engine has no CPU fallback");
    int dev_before = 0;
    (void)hipGetDevice(&dev_before);                        // the caller's current device is left as it was found
    struct Restore { int d; ~Restore() { (void)hipSetDevice(d); } } restore{dev_before};
    bool distinct = true, same = true;
    for (int i = 0; i < d->n_devices; i++) {
        if (d->devices[i] < 0 || d->devices[i] >= ndev) return fail(parent, SSDE_ERR_ARG, "devices[]: no such HIP device");
        for (int k = 0; k < i; k++) distinct = distinct && d->devices[k] != d->devices[i];
        same = same && d->devices[i] == d->devices[0];
    }
    if (!distinct && !same)
        return fail(parent, SSDE_ERR_ARG, "devices[]: either all different (RCCL) or all the same (one-GPU rehearsal)");

    // ---- whole tracks per shard, balanced by rows: cut k is the first ID segment start at or beyond k n / S ----------
    const int64_t n = d->n;
    std::vector<int64_t> starts;
    for (int64_t i = 0; i < n; i++)
        if (i == 0 || d->id[i] != d->id[i - 1]) starts.push_back(i);
    const int64_t n_seg = (int64_t)starts.size();
    if (d->a0 && d->n_seg != n_seg) return fail(parent, SSDE_ERR_ARG, "a0 rows do not match the number of ID segments");
    starts.push_back(n);
    std::vector<int64_t> cut_seg{0};     // segment index where each shard begins
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
    const int S = (int)cut_seg.size();
    cut_seg.push_back(n_seg);

    const int q = d->n_par, sdim = state_dim(d->model, d->n_dim);
    for (int k = 0; k < S; k++) {
        const int64_t seg_lo = cut_seg[k], seg_hi = cut_seg[k + 1];
        const int64_t lo = starts[seg_lo], hi = starts[seg_hi], ns = hi - lo;
        ShardArrays A;
        ssde_desc sd = *d;
        sd.n_devices = 0; sd.devices = nullptr;
        sd.device = d->devices[k];
        sd.n = ns;
        sd.id = d->id + lo; sd.times = d->times + lo;
        copy_cols(d->obs, n, lo, ns, d->n_dim, A.obs);
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
        if (d->h_array) sd.h_array = d->h_array + (size_t)lo * d->n_dim * d->n_dim;
        if (d->eseal_h) sd.eseal_h = d->eseal_h + lo;
        if (d->eseal_R) sd.eseal_R = d->eseal_R + lo;
        if (d->n_decay > 0 && d->t_decay) { copy_cols(d->t_decay, n, lo, ns, q, A.t_decay); sd.t_decay = A.t_decay.data(); }

        ssde_handle* sh = new (std::nothrow) ssde_handle();
        if (!sh) return fail(parent, SSDE_ERR_ALLOC, "out of host memory");
        parent->shards.push_back(sh);
        if (hi < n) sh->last_dt = d->times[hi] - d->times[hi - 1];   // the reference's dtimes at this row: the next track's first time
        parent->shard_row0.push_back(lo);
        int st = build(&sd, sh);
        if (st != SSDE_OK) return fail(parent, st, "shard " + std::to_string(k) + " (device " + std::to_string(sd.device) + "): " + sh->err);
        if (same && k > 0) sh->own_stream = parent->shards[0]->own_stream;      // rehearsal: one stream orders everything
        else HIPCHK(parent, hipStreamCreateWithFlags(&sh->own_stream, hipStreamNonBlocking));
    }
    parent->shard_row0.push_back(n);
    parent->shards_share_device = same;

    // what the C ABI reads off the parent itself: parameter layout, penalty, fixed mask, sizes
    const ssde_handle* s0 = parent->shards[0];
    parent->model = s0->model; parent->d = s0->d; parent->q = s0->q; parent->sdim = s0->sdim; parent->path = s0->path;
    parent->L = s0->L; parent->pen = s0->pen; parent->fixed = s0->fixed; parent->n_free = s0->n_free;
    parent->pen.eseal_n = is_eseal(d->model) ? n : parent->pen.eseal_n;   // the ESEAL priors count ALL rows (nllk_e_seal_ssm.hpp:212-216)
    parent->n = n; parent->n_seg = n_seg; parent->n_steps = n - n_seg;
    parent->device = s0->device;

    if (!same) {
        if (!rccl().load()) return fail(parent, SSDE_ERR_HIP, rccl().err);
        std::vector<ncclComm_t> cs(S);
        std::vector<int> devs(S);
        for (int k = 0; k < S; k++) devs[k] = parent->shards[k]->device;
        NCCLCHK(parent, rccl().CommInitAll(cs.data(), S, devs.data()));
        for (int k = 0; k < S; k++) parent->comms.push_back((void*)cs[k]);
    }
    return SSDE_OK;
}

int reduce_shards(ssde_handle* parent) {
    const size_t count = 2 + (size_t)parent->L.n_full;
    if (parent->shards_share_device) {
        ssde_handle* s0 = parent->shards[0];
        HIPCHK(parent, hipSetDevice(s0->device));
        for (size_t k = 1; k < parent->shards.size(); k++)
            HIPCHK(parent, launch_sum_into(s0->out.p, parent->shards[k]->out.p, (int)count, s0->own_stream));
        return SSDE_OK;
    }
    if (parent->shards.size() < 2) return SSDE_OK;
    NCCLCHK(parent, rccl().GroupStart());
    for (size_t k = 0; k < parent->shards.size(); k++) {
        ssde_handle* sh = parent->shards[k];
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
        const int64_t lo = parent->shard_row0[k], ns = parent->shard_row0[k + 1] - lo;
        std::vector<double> tmp((size_t)ns * parent->sdim);
        int st = ssde_report(sh, par, parent->L.n_full, tmp.data());
        if (st) { parent->err = sh->err; return st; }
        for (int c = 0; c < parent->sdim; c++) memcpy(aest_all + (size_t)c * n + lo, tmp.data() + (size_t)c * ns, (size_t)ns * 8);
    }
    return SSDE_OK;
}

void destroy_dist(ssde_handle* h) {
    for (void* c : h->comms)
        if (c && rccl().CommDestroy) (void)rccl().CommDestroy((ncclComm_t)c);
    h->comms.clear();
    for (size_t k = 0; k < h->shards.size(); k++) {
        if (h->shards_share_device && k > 0) h->shards[k]->own_stream = nullptr;   // owned by shard 0
    }
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

int ssde_comm_init_rank(ssde_handle* h, int32_t n_ranks, int32_t rank, const void* id128) {
    if (!h || !id128 || n_ranks < 1 || rank < 0 || rank >= n_ranks) return SSDE_ERR_ARG;
    if (!h->shards.empty() || !h->comms.empty()) { h->err = "ssde_comm_init_rank: the handle already evaluates over several devices"; return SSDE_ERR_ARG; }
    if (!rccl().load()) { h->err = rccl().err; return SSDE_ERR_HIP; }
    HIPCHK(h, hipSetDevice(h->device));
    ncclUniqueId id;
    memcpy(&id, id128, sizeof(id));
    ncclComm_t c = nullptr;
    NCCLCHK(h, rccl().CommInitRank(&c, n_ranks, id, rank));
    h->comms.push_back((void*)c);
    h->comm_ranks = n_ranks;
    if (!h->own_stream) HIPCHK(h, hipStreamCreateWithFlags(&h->own_stream, hipStreamNonBlocking));
    h->memo_order = -1;
    return SSDE_OK;
}

}  // extern "C"
