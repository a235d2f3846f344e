/* ssde_rcall.c -- .Call shim between R and libssde_hip.so (include/ssde.h).
 *
 * Counterpart of the reference's src/init.c (registration of TMB's generic entry points,
 * /root/reference/src/init.c:6-35): same pattern -- an R_CallMethodDef table registered in
 * R_init_<pkg> with R_useDynamicSymbols(dll, FALSE) -- but the entry points are this engine's.
 *
 * NOT compiled in the build image (no R headers there); it is deliberately thin: every
 * function only unpacks SEXPs into plain C arrays, calls one ssde_* function and turns a
 * non-zero status into Rf_error() after all C resources are released.
 *
 *   ssdeR_create(spec)              -> external pointer        (replaces MakeADFunObject)
 *   ssdeR_eval(ptr, par, order)     -> list(value=, gradient=) (replaces EvalADFunObject)
 *   ssdeR_report(ptr, par)          -> n x sdim matrix aest_all (replaces obj$report()$aest_all)
 *   ssdeR_laplace(ptr, par, order)  -> list(value=, gradient=, par=, hessian.random=): the Laplace marginal over
 *                                      coeff_re, i.e. what fn / gr ARE when MakeADFun gets random = "coeff_re"
 *                                      (R/sde.R:510-525, 656-658); par comes back with coeff_re at u_hat
 *   ssdeR_info(ptr)                 -> named list
 */
#include <R.h>
#include <Rinternals.h>
#include <R_ext/Rdynload.h>
#include <stdlib.h>
#include <string.h>

#include "ssde.h"

static SEXP get_elt(SEXP list, const char *name) {
    SEXP names = Rf_getAttrib(list, R_NamesSymbol);
    for (R_xlen_t i = 0; i < Rf_xlength(list); i++)
        if (strcmp(CHAR(STRING_ELT(names, i)), name) == 0) return VECTOR_ELT(list, i);
    return R_NilValue;
}

static void handle_finalizer(SEXP ptr) {
    ssde_handle *h = (ssde_handle *)R_ExternalPtrAddr(ptr);
    if (h) ssde_destroy(h);
    R_ClearExternalPtr(ptr);
}

static int model_code(const char *type) {
    if (!strcmp(type, "BM")) return SSDE_MODEL_BM;
    if (!strcmp(type, "OU")) return SSDE_MODEL_OU;
    if (!strcmp(type, "BM_SSM")) return SSDE_MODEL_BM_SSM;
    if (!strcmp(type, "OU_SSM")) return SSDE_MODEL_OU_SSM;
    if (!strcmp(type, "CTCRW")) return SSDE_MODEL_CTCRW;
    if (!strcmp(type, "BM_t")) return SSDE_MODEL_BM_T;
    if (!strcmp(type, "ESEAL_SSM")) return SSDE_MODEL_ESEAL_SSM;
    if (!strcmp(type, "CIR")) return SSDE_MODEL_CIR;
    return -1;
}

/* spec: list(type, ID (numeric codes), times, obs (n x d matrix), X_list_fe, X_list_re (lists of
 * matrices or NULL), S_list (list of matrices), a0, P0, H (d x d x n array) or NULL, par_fixed (logical),
 * include_penalty, device, basis_re) -- exactly the objects SDE$setup already has in hand (R/sde.R:496-598). */
SEXP ssdeR_create(SEXP spec) {
    ssde_desc d;
    memset(&d, 0, sizeof(d));
    d.abi_version = SSDE_ABI_VERSION;
    d.model = model_code(CHAR(STRING_ELT(get_elt(spec, "type"), 0)));
    if (d.model < 0) Rf_error("Unknown SDE type");                 /* src/smoothSDE.cpp:25 */
    SEXP obs = get_elt(spec, "obs");
    d.n = Rf_nrows(obs);
    d.n_dim = Rf_ncols(obs);
    d.n_par = (d.model == SSDE_MODEL_BM || d.model == SSDE_MODEL_BM_SSM || d.model == SSDE_MODEL_BM_T ||
               d.model == SSDE_MODEL_ESEAL_SSM) ? d.n_dim + 1 : d.n_dim + 2;
    d.id = REAL(get_elt(spec, "ID"));
    d.times = REAL(get_elt(spec, "times"));
    d.obs = REAL(obs);
    SEXP xfe = get_elt(spec, "X_list_fe"), xre = get_elt(spec, "X_list_re"), sl = get_elt(spec, "S_list");
    int q = d.n_par;
    int32_t *ncol_fe = (int32_t *)R_alloc(q, sizeof(int32_t)), *ncol_re = (int32_t *)R_alloc(q, sizeof(int32_t));
    const double **pfe = (const double **)R_alloc(q, sizeof(double *));
    const double **pre = (const double **)R_alloc(q, sizeof(double *));
    for (int j = 0; j < q; j++) {
        SEXP f = VECTOR_ELT(xfe, j), r = VECTOR_ELT(xre, j);
        ncol_fe[j] = Rf_ncols(f);
        /* an intercept-only block is passed as NULL: the engine broadcasts */
        int ones = ncol_fe[j] == 1;
        for (R_xlen_t i = 0; ones && i < d.n; i++) ones = REAL(f)[i] == 1.0;
        pfe[j] = ones ? NULL : REAL(f);
        ncol_re[j] = (r == R_NilValue) ? 0 : Rf_ncols(r);
        pre[j] = ncol_re[j] > 0 ? REAL(r) : NULL;
    }
    d.ncol_fe = ncol_fe; d.x_fe = pfe; d.ncol_re = ncol_re; d.x_re = pre;
    d.n_smooth = (sl == R_NilValue) ? 0 : (int32_t)Rf_xlength(sl);
    int32_t *sn = (int32_t *)R_alloc(d.n_smooth > 0 ? d.n_smooth : 1, sizeof(int32_t));
    size_t tot = 0;
    for (int s = 0; s < d.n_smooth; s++) { sn[s] = Rf_ncols(VECTOR_ELT(sl, s)); tot += (size_t)sn[s] * sn[s]; }
    double *sb = (double *)R_alloc(tot > 0 ? tot : 1, sizeof(double));
    size_t off = 0;
    for (int s = 0; s < d.n_smooth; s++) {
        memcpy(sb + off, REAL(VECTOR_ELT(sl, s)), (size_t)sn[s] * sn[s] * sizeof(double));
        off += (size_t)sn[s] * sn[s];
    }
    d.smooth_ncol = sn; d.s_blocks = sb;
    d.include_penalty = Rf_asInteger(get_elt(spec, "include_penalty"));
    SEXP a0 = get_elt(spec, "a0"), p0 = get_elt(spec, "P0"), H = get_elt(spec, "H"), fx = get_elt(spec, "par_fixed");
    if (a0 != R_NilValue) { d.a0 = REAL(a0); d.n_seg = Rf_nrows(a0); }
    if (p0 != R_NilValue) d.p0 = REAL(p0);
    if (H != R_NilValue) d.h_array = REAL(H);
    uint8_t *fixed = NULL;
    if (fx != R_NilValue) {
        fixed = (uint8_t *)R_alloc(Rf_xlength(fx), 1);
        for (R_xlen_t k = 0; k < Rf_xlength(fx); k++) fixed[k] = LOGICAL(fx)[k] != 0;
    }
    d.par_fixed = fixed;
    SEXP od = get_elt(spec, "other_data");                          /* tmb_dat$other_data: df of BM_t (R/sde.R:539-541) */
    if (od != R_NilValue && Rf_xlength(od) > 0) { d.other_data = REAL(od); d.n_other_data = (int32_t)Rf_xlength(od); }
    SEXP eh = get_elt(spec, "eseal_h"), eR = get_elt(spec, "eseal_R");     /* tmb_dat$h, tmb_dat$R (R/sde.R:611-614) */
    if (eh != R_NilValue) d.eseal_h = REAL(eh);
    if (eR != R_NilValue) d.eseal_R = REAL(eR);
    /* decaying response model (R/sde.R:635-644): t_decay (q*n), col_decay / ind_decay (1-based in R) */
    SEXP td = get_elt(spec, "t_decay"), cd = get_elt(spec, "col_decay"), idd = get_elt(spec, "ind_decay");
    if (td != R_NilValue && Rf_xlength(td) > 1) {
        int nc = (int)Rf_xlength(cd);
        int32_t *c0 = (int32_t *)R_alloc(nc, sizeof(int32_t)), *i0 = (int32_t *)R_alloc(nc, sizeof(int32_t));
        int nrate = 0;
        for (int k = 0; k < nc; k++) {
            c0[k] = INTEGER(cd)[k] - 1; i0[k] = INTEGER(idd)[k] - 1;
            if (i0[k] + 1 > nrate) nrate = i0[k] + 1;
        }
        d.t_decay = REAL(td); d.n_decay_cols = nc; d.col_decay = c0; d.ind_decay = i0; d.n_decay = nrate;
    }
    /* basis_re: list of length q, element j NULL or list(x = covariate (n), knots = breakpoints, coef = array
     * 4 x K x (n_knots - 1)): the random-effect block of parameter j as a piecewise-cubic table (ssde_ppbasis) */
    SEXP bre = get_elt(spec, "basis_re");
    if (bre != R_NilValue) {
        const ssde_ppbasis **pb = (const ssde_ppbasis **)R_alloc(q, sizeof(ssde_ppbasis *));
        for (int j = 0; j < q; j++) {
            SEXP b = VECTOR_ELT(bre, j);
            pb[j] = NULL;
            if (b == R_NilValue) continue;
            ssde_ppbasis *t = (ssde_ppbasis *)R_alloc(1, sizeof(ssde_ppbasis));
            SEXP kn = get_elt(b, "knots"), cf = get_elt(b, "coef");
            t->x = REAL(get_elt(b, "x"));
            t->n_knots = (int32_t)Rf_xlength(kn);
            t->n_cols = ncol_re[j];
            t->knots = REAL(kn);
            if (Rf_xlength(cf) != (R_xlen_t)4 * t->n_cols * (t->n_knots - 1)) Rf_error("basis_re[[%d]]$coef has the wrong size", j + 1);
            t->coef = REAL(cf);          /* [iv][k][m] with m fastest == R array dim c(4, K, n_knots - 1) */
            pb[j] = t;
        }
        d.basis_re = pb;
    }
    d.na_mode = SSDE_NA_R_ONLY;                                     /* R_IsNA semantics (nllk_ctcrw.hpp:214) */
    /* exact_hess = TRUE (what make_hip_obj sets with random = "coeff_re"): a batch that ssde_create puts on the lane = track
     * kernels keeps its rows a second time for the second-order pass (include/ssde.h: SSDE_FLAG_EXACT_HESS) */
    SEXP eh2 = get_elt(spec, "exact_hess");
    if (eh2 != R_NilValue && Rf_asInteger(eh2) == 1) d.flags |= SSDE_FLAG_EXACT_HESS;
    SEXP dev = get_elt(spec, "device");
    d.device = (dev == R_NilValue) ? -1 : Rf_asInteger(dev);
    /* devices = c(0, 1, ..., 7): one R process, several GPUs -- whole tracks are sharded over them inside the engine
     * and every evaluation ends in one ncclAllReduce of the 2 + p doubles (include/ssde.h: n_devices) */
    SEXP devs = get_elt(spec, "devices");
    if (devs != R_NilValue && Rf_xlength(devs) > 1) {
        int nd = (int)Rf_xlength(devs);
        int32_t *dl = (int32_t *)R_alloc(nd, sizeof(int32_t));
        for (int k = 0; k < nd; k++) dl[k] = INTEGER(devs)[k];
        d.n_devices = nd; d.devices = dl;
    }
    ssde_handle *h = NULL;
    int st = ssde_create(&d, &h);
    if (st != SSDE_OK) Rf_error("ssde_create failed (%d): %s", st, ssde_last_error(NULL));
    SEXP ptr = PROTECT(R_MakeExternalPtr(h, R_NilValue, R_NilValue));
    R_RegisterCFinalizerEx(ptr, handle_finalizer, TRUE);
    UNPROTECT(1);
    return ptr;
}

SEXP ssdeR_eval(SEXP ptr, SEXP par, SEXP order) {
    ssde_handle *h = (ssde_handle *)R_ExternalPtrAddr(ptr);
    if (!h) Rf_error("engine handle was destroyed (call $setup() again)");
    int np = (int)Rf_xlength(par), ord = Rf_asInteger(order);
    SEXP val = PROTECT(Rf_allocVector(REALSXP, 1)), grad = PROTECT(Rf_allocVector(REALSXP, np));
    int st = ssde_eval(h, REAL(par), np, ord, REAL(val), REAL(grad));
    if (st != SSDE_OK) { UNPROTECT(2); Rf_error("ssde_eval failed (%d): %s", st, ssde_last_error(h)); }
    SEXP out = PROTECT(Rf_allocVector(VECSXP, 2)), nm = PROTECT(Rf_allocVector(STRSXP, 2));
    SET_STRING_ELT(nm, 0, Rf_mkChar("value")); SET_STRING_ELT(nm, 1, Rf_mkChar("gradient"));
    SET_VECTOR_ELT(out, 0, val); SET_VECTOR_ELT(out, 1, grad);
    Rf_setAttrib(out, R_NamesSymbol, nm);
    UNPROTECT(4);
    return out;
}

SEXP ssdeR_laplace(SEXP ptr, SEXP par, SEXP order) {
    ssde_handle *h = (ssde_handle *)R_ExternalPtrAddr(ptr);
    if (!h) Rf_error("engine handle was destroyed (call $setup() again)");
    int np = (int)Rf_xlength(par), ord = Rf_asInteger(order);
    ssde_info_t inf;
    ssde_info(h, &inf);
    SEXP val = PROTECT(Rf_allocVector(REALSXP, 1)), grad = PROTECT(Rf_allocVector(REALSXP, np));
    SEXP pout = PROTECT(Rf_duplicate(par));                         /* in/out: coeff_re <- u_hat */
    /* n_u <= number of coeff_re entries; the engine fills the leading n_u x n_u block */
    SEXP hess = PROTECT(Rf_allocMatrix(REALSXP, np, np));
    memset(REAL(hess), 0, sizeof(double) * (size_t)np * np);
    int st = ssde_laplace_eval(h, REAL(pout), np, ord, REAL(val), REAL(grad), REAL(hess), NULL);
    if (st != SSDE_OK) { UNPROTECT(4); Rf_error("ssde_laplace_eval failed (%d): %s", st, ssde_last_error(h)); }
    SEXP out = PROTECT(Rf_allocVector(VECSXP, 4)), nm = PROTECT(Rf_allocVector(STRSXP, 4));
    SET_STRING_ELT(nm, 0, Rf_mkChar("value")); SET_STRING_ELT(nm, 1, Rf_mkChar("gradient"));
    SET_STRING_ELT(nm, 2, Rf_mkChar("par")); SET_STRING_ELT(nm, 3, Rf_mkChar("hessian.random.packed"));
    SET_VECTOR_ELT(out, 0, val); SET_VECTOR_ELT(out, 1, grad); SET_VECTOR_ELT(out, 2, pout); SET_VECTOR_ELT(out, 3, hess);
    Rf_setAttrib(out, R_NamesSymbol, nm);
    UNPROTECT(6);
    return out;
}

SEXP ssdeR_report(SEXP ptr, SEXP par) {
    ssde_handle *h = (ssde_handle *)R_ExternalPtrAddr(ptr);
    if (!h) Rf_error("engine handle was destroyed (call $setup() again)");
    ssde_info_t inf;
    ssde_info(h, &inf);
    SEXP out = PROTECT(Rf_allocMatrix(REALSXP, (int)inf.n_rows, inf.sdim));
    int st = ssde_report(h, REAL(par), (int)Rf_xlength(par), REAL(out));
    if (st != SSDE_OK) { UNPROTECT(1); Rf_error("ssde_report failed (%d): %s", st, ssde_last_error(h)); }
    UNPROTECT(1);
    return out;
}

/* he(x): exact second derivatives of the joint penalised nllk over the 0-based full-parameter indices `idx` (ssde_hess:
 * direct families BM / OU).  Returns NULL where the engine has no exact Hessian: the R side then differences gr(). */
SEXP ssdeR_hess(SEXP ptr, SEXP par, SEXP idx) {
    ssde_handle *h = (ssde_handle *)R_ExternalPtrAddr(ptr);
    if (!h) Rf_error("engine handle was destroyed (call $setup() again)");
    int n = (int)Rf_xlength(idx);
    SEXP out = PROTECT(Rf_allocMatrix(REALSXP, n, n));
    int st = ssde_hess(h, REAL(par), (int)Rf_xlength(par), INTEGER(idx), n, REAL(out));
    UNPROTECT(1);
    if (st == SSDE_ERR_MODEL) return R_NilValue;
    if (st != SSDE_OK) Rf_error("ssde_hess failed (%d): %s", st, ssde_last_error(h));
    return out;
}

SEXP ssdeR_info(SEXP ptr) {
    ssde_handle *h = (ssde_handle *)R_ExternalPtrAddr(ptr);
    if (!h) Rf_error("engine handle was destroyed");
    ssde_info_t inf;
    ssde_info(h, &inf);
    /* kernel_id: SSDE_KERNEL_* -- which kernel family ran the last evaluation's rows (ABI 10) */
    const char *nms[] = {"n_par_full", "n_free", "path", "uniform_dt", "n_tracks", "n_rows", "window", "window_check",
                         "n_rows_tiled", "n_groups", "n_clean_groups", "n_devices", "kernel_id", "comm_ranks", "comm_ranks_reported",
                         "exact_hess_scope"};
    double vals[] = {inf.n_par_full, inf.n_free, inf.path, inf.uniform_dt, (double)inf.n_tracks, (double)inf.n_rows,
                     inf.window, inf.window_check, (double)inf.n_rows_tiled, inf.n_groups, inf.n_clean_groups, inf.n_devices,
                     inf.kernel_id, inf.comm_ranks, inf.comm_ranks_reported, inf.exact_hess_scope};
    const int nv = (int)(sizeof(vals) / sizeof(vals[0]));
    SEXP out = PROTECT(Rf_allocVector(REALSXP, nv)), nm = PROTECT(Rf_allocVector(STRSXP, nv));
    for (int i = 0; i < nv; i++) { REAL(out)[i] = vals[i]; SET_STRING_ELT(nm, i, Rf_mkChar(nms[i])); }
    Rf_setAttrib(out, R_NamesSymbol, nm);
    UNPROTECT(2);
    return out;
}

static const R_CallMethodDef ssde_calldefs[] = {
    {"ssdeR_create", (DL_FUNC)&ssdeR_create, 1},
    {"ssdeR_eval", (DL_FUNC)&ssdeR_eval, 3},
    {"ssdeR_laplace", (DL_FUNC)&ssdeR_laplace, 3},
    {"ssdeR_report", (DL_FUNC)&ssdeR_report, 2},
    {"ssdeR_hess", (DL_FUNC)&ssdeR_hess, 3},
    {"ssdeR_info", (DL_FUNC)&ssdeR_info, 1},
    {NULL, NULL, 0}};

/* In the package this table is appended to R_CallDef[] of src/init.c (line 22-26); as a separate
 * shared object it registers itself: */
void R_init_ssdehip(DllInfo *dll) {
    R_registerRoutines(dll, NULL, ssde_calldefs, NULL, NULL);
    R_useDynamicSymbols(dll, FALSE);
}
