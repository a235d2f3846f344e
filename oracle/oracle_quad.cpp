// oracle/oracle_quad.cpp -- the restated nllk templates evaluated in IEEE binary128 (liboracle_quad.so).
//
// TEST INFRASTRUCTURE ONLY (tools/extreme_triage.py, tests/test_oracle_quad.py).  Same templates as liboracle.so
// (ssde_oracle.hpp, every function citing /root/reference/src/nllk/*.hpp), instantiated with ssde_oracle::Quad:
//
//   oracle_eval_quad(desc, par, order, &value, grad, fd_step)
//     value = nllk + penalty of the double-precision inputs, evaluated with a 113-bit mantissa, rounded to double once
//     grad  = central differences of that binary128 function with step fd_step (default 1e-10): truncation error
//             O(step^2) ~ 1e-20 relative, rounding 1e-34 / step ~ 1e-24 -- far below any double-precision question.
//             No dual numbers: nothing of the gradient path of liboracle.so is shared.
#include <vector>

#include "quad.hpp"
#include "ssde_oracle.hpp"

using namespace ssde_oracle;

namespace {
Quad total(const Problem& p, const std::vector<Quad>& par) {
    return nllk_data<Quad>(p, par.data(), nullptr) + penalty<Quad>(p, par.data());
}
}  // namespace

extern "C" int oracle_eval_quad(const ssde_desc* d, const double* par, int order, double* value, double* grad,
                                double fd_step) {
    Problem p = make_problem(d);
    const int np = p.n_par_full;
    std::vector<Quad> x(np);
    for (int k = 0; k < np; k++) x[k] = Quad(par[k]);
    *value = asDouble(total(p, x));
    if (order < 1 || !grad) return 0;
    const __float128 h = fd_step > 0.0 ? (__float128)fd_step : (__float128)1e-10;
    for (int k = 0; k < np; k++) {
        grad[k] = 0.0;
        if (d->par_fixed && d->par_fixed[k]) continue;
        const Quad keep = x[k];
        x[k] = q128(keep.v + h);
        const Quad fp = total(p, x);
        x[k] = q128(keep.v - h);
        const Quad fm = total(p, x);
        x[k] = keep;
        grad[k] = (double)((fp.v - fm.v) / (2 * h));
    }
    return 0;
}

// arbiter mode of the restatement (ssde_oracle.hpp: keep_P_symmetric): 0 = the literal recursion (default), 1 = P <- (P + P') / 2
extern "C" void ssde_oracle_keep_P_symmetric(int on) { ssde_oracle::keep_P_symmetric() = on != 0; }

