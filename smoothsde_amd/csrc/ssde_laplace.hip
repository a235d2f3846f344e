// ssde_laplace.hip -- ssde_laplace_eval: the marginal negative log-likelihood over the random-effect coefficients.
//
// In the reference `random = "coeff_re"` (/root/reference/R/sde.R:510-525, 656-658) makes TMB integrate coeff_re out of
// the joint nllk g(theta, u) by the Laplace approximation
//
//     f(theta) = g(theta, u^) + 1/2 log det H_uu(theta, u^) - n_u/2 log(2 pi),      u^ = argmin_u g(theta, u),
//
// and tmb_obj$fn / $gr (R/sde.R:694-696) ARE this f and its gradient.  TMB gets the derivatives from CppAD tapes; this
// engine has the joint value and its full gradient from the device (ssde_eval, 0.1-1 ms), and builds f on top of it:
//
//   * H_uu: central differences of the device gradient in the coeff_re coordinates (2 n_u evaluations; n_u ~ 10-40);
//   * inner problem: Newton on u with that Hessian (Cholesky; shifted to positive definite away from the optimum),
//     backtracking on the joint value, warm start from the coeff_re the caller passes in (and receives back);
//   * gradient:  df/dtheta_k = dg/dtheta_k (EXACT: the device gradient at (theta, u^); dg/du = 0 there)
//                            + 1/2 d/dtheta_k log det H_uu(theta, u^(theta)),
//     the second term -- TMB's third-order term -- by a central difference along theta_k with u moved along the
//     implicit-function tangent du^/dtheta_k = -H_uu^-1 H_u,theta_k (H_u,theta_k: central difference of the device
//     gradient), i.e. 2 + 4 n_u evaluations per outer parameter and no further inner solves.
//
// Host arithmetic on top of the C ABI's own ssde_eval: works for single-device, multi-device and communicator handles
// alike (every rank runs the same deterministic host sequence on the same all-reduced numbers).
#include <algorithm>
#include <cmath>
#include <cstring>
#include <limits>
#include <vector>

#include "ssde_engine.hpp"

namespace {

// in-place Cholesky A = L L' of a column-major n x n matrix (lower triangle); false if not positive definite
bool cholesky(std::vector<double>& A, int n) {
    for (int j = 0; j < n; j++) {
        double d = A[j + (size_t)j * n];
        for (int k = 0; k < j; k++) d -= A[j + (size_t)k * n] * A[j + (size_t)k * n];
        if (!(d > 0.0) || !std::isfinite(d)) return false;
        d = std::sqrt(d);
        A[j + (size_t)j * n] = d;
        for (int i = j + 1; i < n; i++) {
            double s = A[i + (size_t)j * n];
            for (int k = 0; k < j; k++) s -= A[i + (size_t)k * n] * A[j + (size_t)k * n];
            A[i + (size_t)j * n] = s / d;
        }
    }
    return true;
}
void chol_solve(const std::vector<double>& Lm, int n, std::vector<double>& b) {
    for (int i = 0; i < n; i++) {
        double s = b[i];
        for (int k = 0; k < i; k++) s -= Lm[i + (size_t)k * n] * b[k];
        b[i] = s / Lm[i + (size_t)i * n];
    }
    for (int i = n - 1; i >= 0; i--) {
        double s = b[i];
        for (int k = i + 1; k < n; k++) s -= Lm[k + (size_t)i * n] * b[k];
        b[i] = s / Lm[i + (size_t)i * n];
    }
}
double chol_logdet(const std::vector<double>& Lm, int n) {
    double s = 0.0;
    for (int i = 0; i < n; i++) s += std::log(Lm[i + (size_t)i * n]);
    return 2.0 * s;
}

struct Laplace {
    ssde_handle* h;
    int np, nu;
    std::vector<int> ir;               // coeff_re entries that are not fixed
    double hess_step, fd_step, newton_tol;
    int max_newton;
    bool exact = false;                // exact second derivatives over EVERY free parameter (ssde_engine::hess_exact_scope == 2 or 3)
    bool exact_uu = false;             // ... over the random-effect coefficients at least (scope >= 1)
    std::vector<double> work, grad;

    int joint(const std::vector<double>& par, double* v, std::vector<double>* g) {
        grad.assign(np, 0.0);
        int st = ssde_eval(h, par.data(), np, 1, v, grad.data());
        if (g) *g = grad;
        return st;
    }
    // Joint value + gradient at SEVERAL parameter vectors that do not depend on each other (the 2 n_u points of a
    // differenced Hessian): enqueued back to back through ssde_eval_device and read back with ONE copy, so that the host's
    // launch work for evaluation k + 1 overlaps the GPU's work for evaluation k -- on a latency-bound problem (one track,
    // 0.1 ms per synchronous evaluation) that is most of the time.  An evaluation whose window hand-over check does not
    // pass is repeated through ssde_eval, which owns the retry policy; a handle that cannot evaluate asynchronously
    // (track shards over several devices) takes the plain loop.
    int joint_batch(const std::vector<std::vector<double>>& P, std::vector<double>& V, std::vector<std::vector<double>>& G) {
        const size_t K = P.size(), nout = 2 + (size_t)np;
        V.assign(K, 0.0); G.assign(K, std::vector<double>());
        bool async_ok = h->shards.empty() || h->n_track_shards <= 1;
        // (the row-varying path replays a hipGraph in its synchronous call -- one launch per evaluation -- and needs the previous
        //  evaluation's statistics before it can plan the next: measured, 0.137 ms per evaluation batched against 0.106 ms one by one)
        const ssde_handle* e0 = h->shards.empty() ? h : h->shards[0];
        if (e0->path == ssde_engine::PATH_TV) async_ok = false;
        // ranks of a communicator take the same route (agreed on at ssde_comm_init_rank): the batched route pairs its
        // k-th all-reduce with the other ranks' k-th, the one-by-one route interleaves retries
        if (!h->comms.empty() && h->comm_async_ok >= 0 && (h->shards.empty() || h->n_track_shards <= 1)) async_ok = h->comm_async_ok != 0;
        if (async_ok && K > 1) {
            if (h->lap_out.n < K * nout) {
                h->lap_out.release();
                if (h->lap_out.alloc(K * nout) != hipSuccess) async_ok = false;
            }
        }
        if (!async_ok || K <= 1) {
            for (size_t k = 0; k < K; k++) { int st = joint(P[k], &V[k], &G[k]); if (st) return st; }
            return SSDE_OK;
        }
        for (size_t k = 0; k < K; k++) {
            int st = ssde_eval_device(h, P[k].data(), np, 1, h->lap_out.p + k * nout, nullptr);
            if (st) return st;
        }
        std::vector<double> host(K * nout);
        if (hipMemcpy(host.data(), h->lap_out.p, K * nout * 8, hipMemcpyDeviceToHost) != hipSuccess) { h->err = "ssde_laplace_eval: read-back failed"; return SSDE_ERR_HIP; }
        for (size_t k = 0; k < K; k++) {
            const double* o = host.data() + k * nout;
            if (!(o[1 + np] <= ssde_engine::SSDE_WINDOW_TOL)) {               // the windows did not agree: the synchronous call repeats it with its policy
                int st = joint(P[k], &V[k], &G[k]);
                if (st) return st;
                continue;
            }
            G[k].assign(o + 1, o + 1 + np);
            double pen = 0.0;
            int st = ssde_penalty(h, P[k].data(), np, &pen, G[k].data());
            if (st) return st;
            V[k] = o[0] + pen;
        }
        return SSDE_OK;
    }
    // H_uu at (par): central differences of the gradient, symmetrised
    int hess_uu(const std::vector<double>& par, std::vector<double>& H) {
        H.assign((size_t)nu * nu, 0.0);
        if (exact_uu) {                              // direct families BM / OU, smooth-drift state-space batches: nothing differenced (ssde_hess.hip)
            std::vector<int32_t> ix(ir.begin(), ir.end());
            const int st = ssde_engine::hess_exact(h, par.data(), ix.data(), nu, H.data());
            if (st != SSDE_ERR_ARG && st != SSDE_ERR_MODEL) return st;
            exact_uu = false;                        // (an entry without exact second derivatives after all: difference the gradient)
            H.assign((size_t)nu * nu, 0.0);
        }
        std::vector<std::vector<double>> P;
        std::vector<double> steps((size_t)nu);
        for (int k = 0; k < nu; k++) {
            steps[k] = hess_step * std::max(1.0, std::fabs(par[ir[k]]));
            P.push_back(par); P.back()[ir[k]] = par[ir[k]] + steps[k];
            P.push_back(par); P.back()[ir[k]] = par[ir[k]] - steps[k];
        }
        std::vector<double> V;
        std::vector<std::vector<double>> G;
        int st = joint_batch(P, V, G);
        if (st) return st;
        for (int k = 0; k < nu; k++) {
            const std::vector<double>&gp = G[2 * (size_t)k], &gm = G[2 * (size_t)k + 1];
            for (int i = 0; i < nu; i++) H[i + (size_t)k * nu] = (gp[ir[i]] - gm[ir[i]]) / (2.0 * steps[k]);
        }
        for (int i = 0; i < nu; i++)
            for (int k = i + 1; k < nu; k++) {
                const double m = 0.5 * (H[i + (size_t)k * nu] + H[k + (size_t)i * nu]);
                H[i + (size_t)k * nu] = H[k + (size_t)i * nu] = m;
            }
        return SSDE_OK;
    }
    // Newton solve for u^ from par's coeff_re; returns the joint value, its gradient and H_uu there
    int inner(std::vector<double>& par, double* val, std::vector<double>* g_at, std::vector<double>& H) {
        std::vector<double> g;
        int st = joint(par, val, &g);
        if (st) return st;
        bool h_current = false;
        for (int it = 0; it < max_newton; it++) {
            if (!std::isfinite(*val)) break;
            st = hess_uu(par, H);
            if (st) return st;
            h_current = true;
            // Away from the inner optimum the joint nllk need not be convex in u (a penalty with a null space leaves
            // those directions to the data): shift an indefinite Hessian to positive definite for the STEP
            // (Levenberg), so that the iteration keeps descending towards a minimum, where H itself is positive
            double dmax = 0.0;
            for (int i = 0; i < nu; i++) dmax = std::max(dmax, std::fabs(H[i + (size_t)i * nu]));
            std::vector<double> Lm = H;
            double shift = 0.0;
            while (!cholesky(Lm, nu)) {
                shift = shift == 0.0 ? 1e-3 * std::max(1.0, dmax) : 4.0 * shift;
                if (!(shift < 1e12 * std::max(1.0, dmax))) { *val = INFINITY; return SSDE_OK; }
                Lm = H;
                for (int i = 0; i < nu; i++) Lm[i + (size_t)i * nu] += shift;
            }
            std::vector<double> step(nu);
            for (int i = 0; i < nu; i++) step[i] = g[ir[i]];
            chol_solve(Lm, nu, step);
            // backtracking: the joint is close to quadratic in u, a full step almost always passes
            double t = 1.0, v_new = *val;
            std::vector<double> p_new = par, g_new;
            bool ok = false;
            for (int bt = 0; bt < 20; bt++) {
                for (int i = 0; i < nu; i++) p_new[ir[i]] = par[ir[i]] - t * step[i];
                st = joint(p_new, &v_new, &g_new);
                if (st) return st;
                if (std::isfinite(v_new) && v_new <= *val + 1e-12 * std::fabs(*val)) { ok = true; break; }
                t *= 0.5;
            }
            if (!ok) break;                          // no descent along the Newton direction: u is as good as it gets
            double smax = 0.0, umax = 0.0;
            for (int i = 0; i < nu; i++) { smax = std::max(smax, std::fabs(t * step[i])); umax = std::max(umax, std::fabs(p_new[ir[i]])); }
            par = p_new; *val = v_new; g = g_new;
            h_current = smax <= (exact_uu ? 1e-12 : 1e-3 * hess_step) * std::max(1.0, umax);   // H moved by less than its own (differencing) error
            if (smax <= newton_tol * std::max(1.0, umax)) break;
        }
        if (!h_current && std::isfinite(*val)) {
            st = hess_uu(par, H);
            if (st) return st;
        }
        if (g_at) *g_at = g;
        return SSDE_OK;
    }
};

}  // namespace

extern "C" int ssde_laplace_eval(ssde_handle* h, double* par, int32_t n_par_full, int32_t order, double* value,
                                 double* grad, double* hess_uu, const ssde_laplace_opts* opts) {
    if (!h || !par || !value) return SSDE_ERR_ARG;
    if (n_par_full != h->L.n_full) { h->err = "parameter vector has the wrong length"; return SSDE_ERR_ARG; }
    Laplace lp;
    lp.h = h; lp.np = n_par_full;
    lp.exact = ssde_engine::hess_exact_scope(h) == 2 || ssde_engine::hess_exact_scope(h) == 3;   // (3: row-varying coefficients, k_tv_hess.hip -- every free entry)
    lp.exact_uu = ssde_engine::hess_exact_scope(h) >= 1;
    lp.hess_step = (opts && opts->hess_step > 0) ? opts->hess_step : 1e-4;
    // the log-determinant term differences H_uu: with exact Hessians a shorter step is affordable (truncation e^2, rounding 1e-16 / e)
    lp.fd_step = (opts && opts->fd_step > 0) ? opts->fd_step : (lp.exact_uu ? 1e-5 : 1e-4);
    lp.newton_tol = (opts && opts->newton_tol > 0) ? opts->newton_tol : (lp.exact_uu ? 1e-10 : 1e-8);   // (exact Hessians: Newton converges quadratically)
    lp.max_newton = (opts && opts->max_newton > 0) ? opts->max_newton : 30;
    for (int k = 0; k < h->L.n_re; k++)
        if (!h->fixed[h->L.off_re + k]) lp.ir.push_back(h->L.off_re + k);
    lp.nu = (int)lp.ir.size();
    const int np = n_par_full, nu = lp.nu;
    if (grad) for (int k = 0; k < np; k++) grad[k] = 0.0;
    std::vector<double> p(par, par + np), g, H;
    if (nu == 0) {                                   // nothing to integrate out: the joint IS the marginal
        int st = lp.joint(p, value, &g);
        if (st == SSDE_OK && order >= 1 && grad) memcpy(grad, g.data(), (size_t)np * 8);
        return st;
    }
    double val = 0.0;
    int st = lp.inner(p, &val, &g, H);
    if (st) return st;
    if (hess_uu) memcpy(hess_uu, H.data(), (size_t)nu * nu * 8);
    std::vector<double> Lm = H;
    // not a minimum: a rejected step -- and the caller's coeff_re stay what they were (a line-search probe must not move the warm start)
    if (!std::isfinite(val) || !cholesky(Lm, nu)) { *value = INFINITY; return SSDE_OK; }
    for (int i = 0; i < nu; i++) par[lp.ir[i]] = p[lp.ir[i]];          // u^ back to the caller (warm start / par.random)
    const double half_ld0 = 0.5 * chol_logdet(Lm, nu);
    *value = val + half_ld0 - 0.5 * nu * std::log(2.0 * M_PI);
    if (order < 1 || !grad) return SSDE_OK;

    std::vector<uint8_t> is_u(np, 0);
    for (int i = 0; i < nu; i++) is_u[lp.ir[i]] = 1;
    std::vector<double> pp, gp, gm, Hk, du(nu);
    // H_u,theta for every outer parameter at once where second derivatives are exact
    std::vector<int> outer;
    for (int k = 0; k < np; k++) if (!h->fixed[k] && !is_u[k]) outer.push_back(k);
    std::vector<double> Hfull;
    const int no = (int)outer.size(), nf = nu + no;
    if (lp.exact && no > 0) {
        std::vector<int32_t> ix;
        for (int i = 0; i < nu; i++) ix.push_back(lp.ir[i]);
        for (int k : outer) ix.push_back(k);
        Hfull.assign((size_t)nf * nf, 0.0);
        st = ssde_engine::hess_exact(h, p.data(), ix.data(), nf, Hfull.data());
        if (st) return st;
    }
    for (int ko = 0; ko < no; ko++) {
        const int k = outer[ko];
        const double e = lp.fd_step * std::max(1.0, std::fabs(p[k]));
        // H_u,theta_k and the tangent of u^(theta)
        if (lp.exact) {
            for (int i = 0; i < nu; i++) du[i] = Hfull[i + (size_t)(nu + ko) * nf];
        } else {
            double v;
            pp = p; pp[k] = p[k] + e;
            st = lp.joint(pp, &v, &gp);
            if (st) return st;
            pp[k] = p[k] - e;
            st = lp.joint(pp, &v, &gm);
            if (st) return st;
            for (int i = 0; i < nu; i++) du[i] = (gp[lp.ir[i]] - gm[lp.ir[i]]) / (2.0 * e);
        }
        chol_solve(Lm, nu, du);                                       // du = H^-1 H_u,theta_k;  du^/dtheta_k = -du
        double half_ld[2] = {0.0, 0.0};
        bool okk = true;
        for (int sgn = 0; sgn < 2 && okk; sgn++) {
            const double sg = sgn == 0 ? 1.0 : -1.0;
            pp = p; pp[k] = p[k] + sg * e;
            for (int i = 0; i < nu; i++) pp[lp.ir[i]] = p[lp.ir[i]] - sg * e * du[i];
            st = lp.hess_uu(pp, Hk);
            if (st) return st;
            okk = cholesky(Hk, nu);
            if (okk) half_ld[sgn] = 0.5 * chol_logdet(Hk, nu);
        }
        grad[k] = g[k] + (okk ? (half_ld[0] - half_ld[1]) / (2.0 * e) : NAN);
    }
    return SSDE_OK;
}
