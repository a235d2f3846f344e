// oracle/oracle_capi.cpp -- C entry points of the CPU oracle (ctypes-loadable).
//
// TEST INFRASTRUCTURE ONLY: loaded by tests/, __graft_entry__.smoke() and the
// cpu_baseline leg of bench.py.  The product library (smoothsde_amd/lib/libssde_hip.so)
// never links or calls it.  PARITY UNPINNED (see ssde_oracle.hpp / README.md).
//
//   oracle_eval(desc, par, order, &value, grad, aest_all, n_threads)
//     value = nllk_data + penalty, exactly what objective_function<Type>::operator()
//     returns (/root/reference/src/smoothSDE.cpp:9-28); grad over the FULL parameter
//     vector (0 for par_fixed entries), obtained by forward-mode duals in batches.
//     n_threads > 1 shards whole ID segments over std::threads (the reference itself is
//     single-threaded; sharding only changes the order of the final sum).
#include <algorithm>
#include <thread>
#include <vector>

#include "ssde_oracle.hpp"

using namespace ssde_oracle;

namespace {

constexpr int NB = 8;  // dual directions per pass

struct Shard {
    int64_t row_lo, row_hi, seg_lo;
};

std::vector<Shard> make_shards(const ssde_desc* d, int n_threads) {
    std::vector<int64_t> starts;
    for (int64_t i = 0; i < d->n; i++)
        if (i == 0 || d->id[i] != d->id[i - 1]) starts.push_back(i);
    int64_t nseg = (int64_t)starts.size();
    starts.push_back(d->n);
    int T = (int)std::max<int64_t>(1, std::min<int64_t>(n_threads, nseg));
    std::vector<Shard> out;
    // balance by rows
    int64_t seg = 0;
    for (int t = 0; t < T; t++) {
        int64_t target = d->n * (t + 1) / T;
        int64_t s0 = seg;
        while (seg < nseg && (starts[seg + 1] <= target || seg == s0)) seg++;
        if (t == T - 1) seg = nseg;
        if (seg > s0) out.push_back({starts[s0], starts[seg], s0});
    }
    return out;
}

template <class Type>
Type eval_shard(const ssde_desc* d, const Shard& s, const Type* par, double* aest_all) {
    Problem p = make_problem(d);
    p.row_lo = s.row_lo;
    p.row_hi = s.row_hi;
    p.seg_lo = s.seg_lo;
    return nllk_data<Type>(p, par, aest_all);
}

}  // namespace

extern "C" {

int oracle_n_par_full(const ssde_desc* d) { return make_problem(d).n_par_full; }

static int eval_impl(const ssde_desc* d, const double* par, int order, double* value, double* grad,
                     double* aest_all, int n_threads, bool with_penalty) {
    Problem p0 = make_problem(d);
    const int np = p0.n_par_full;
    std::vector<Shard> shards = make_shards(d, n_threads);
    const int S = (int)shards.size();

    // value
    {
        std::vector<double> part(S, 0.0);
        std::vector<std::thread> th;
        for (int s = 0; s < S; s++)
            th.emplace_back([&, s]() { part[s] = eval_shard<double>(d, shards[s], par, aest_all); });
        for (auto& t : th) t.join();
        double v = 0.0;
        for (int s = 0; s < S; s++) v += part[s];
        if (with_penalty) v += penalty<double>(p0, par);
        *value = v;
    }
    if (order < 1 || !grad) return 0;

    std::vector<int> free_idx;
    for (int k = 0; k < np; k++) {
        grad[k] = 0.0;
        if (!(d->par_fixed && d->par_fixed[k])) free_idx.push_back(k);
    }
    for (size_t b0 = 0; b0 < free_idx.size(); b0 += NB) {
        int nb = (int)std::min<size_t>(NB, free_idx.size() - b0);
        std::vector<Dual<NB>> dp(np);
        for (int k = 0; k < np; k++) dp[k] = Dual<NB>(par[k]);
        for (int j = 0; j < nb; j++) dp[free_idx[b0 + j]].d[j] = 1.0;
        std::vector<Dual<NB>> part(S);
        std::vector<std::thread> th;
        for (int s = 0; s < S; s++)
            th.emplace_back([&, s]() { part[s] = eval_shard<Dual<NB>>(d, shards[s], dp.data(), nullptr); });
        for (auto& t : th) t.join();
        Dual<NB> tot(0.0);
        for (int s = 0; s < S; s++) tot = tot + part[s];
        if (with_penalty) tot = tot + penalty<Dual<NB>>(p0, dp.data());
        for (int j = 0; j < nb; j++) grad[free_idx[b0 + j]] = tot.d[j];
    }
    return 0;
}

int oracle_eval(const ssde_desc* d, const double* par, int order, double* value, double* grad,
                double* aest_all, int n_threads) {
    return eval_impl(d, par, order, value, grad, aest_all, n_threads, true);
}

// data term only (no penalty): what one GPU shard contributes before the all-reduce
int oracle_eval_data(const ssde_desc* d, const double* par, int order, double* value, double* grad,
                     int n_threads) {
    return eval_impl(d, par, order, value, grad, nullptr, n_threads, false);
}

}  // extern "C"

// arbiter mode of the restatement (ssde_oracle.hpp: keep_P_symmetric): 0 = the literal recursion (default), 1 = P <- (P + P') / 2
extern "C" void ssde_oracle_keep_P_symmetric(int on) { ssde_oracle::keep_P_symmetric() = on != 0; }

