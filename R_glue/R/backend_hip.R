# backend_hip.R -- R glue for the MI355X engine (drop-in for TMB::MakeADFun inside SDE$setup).
#
# NOT run in the build image (no R there).  A maintainer of smoothSDE adds this file to R/, the shim
# R_glue/src/ssde_rcall.c to src/, links libssde_hip.so, and changes ONE call in R/sde.R (see
# INTEGRATION.md): where $setup() calls MakeADFun(...) at R/sde.R:656-658 and :666-668 it calls
# make_hip_obj(...) instead when other_data$backend == "hip".  $fit() (optim BFGS on obj$fn / obj$gr,
# R/sde.R:694-697) and logLik.SDE (obj_joint$fn, R/utility.R:118) stay as they are.

#' Piecewise-cubic table of a univariate regression-spline smooth (ssde_ppbasis, include/ssde.h)
#'
#' mgcv's "cr", "cs", "bs" (m = 3) and "ps" bases are cubic polynomials of the covariate on every knot interval, also
#' after the identifiability constraint has been absorbed (a linear map of the columns).  Four evaluations per interval
#' therefore determine the block exactly; the engine then reads the covariate (8 B/row) instead of the n x K block.
#' @param sm smooth object from smoothCon(..., absorb.cons = TRUE) -- gam_setup$smooth[[i]] in SDE$make_mat
#' @param x covariate values of the rows (the column sm$term of the data)
#' @return list(x, knots, coef) for spec$basis_re[[j]], or NULL when the smooth is not of that kind
pp_table <- function(sm, x) {
    if(length(sm$term) != 1 || sm$by != "NA") return(NULL)
    br <- if(inherits(sm, "cr.smooth") || inherits(sm, "cs.smooth")) sm$xp
          else if(inherits(sm, "Bspline.smooth") || inherits(sm, "pspline.smooth")) sort(unique(sm$knots))
          else return(NULL)
    if(!is.null(sm$m) && inherits(sm, c("Bspline.smooth", "pspline.smooth")) && sm$m[1] != 3 && sm$m[1] != 2) return(NULL)
    br <- br[br > min(x) & br < max(x)]
    br <- c(min(x), br, max(x))                       # "cr" extrapolates linearly outside xp: never needed inside range(x)
    K <- ncol(mgcv::PredictMat(sm, setNames(data.frame(x[1]), sm$term)))
    coef <- array(0, dim = c(4, K, length(br) - 1))
    for(iv in seq_len(length(br) - 1)) {
        t <- (br[iv + 1] - br[iv]) * c(0.1, 0.37, 0.63, 0.9)
        B <- mgcv::PredictMat(sm, setNames(data.frame(br[iv] + t), sm$term))      # 4 x K
        coef[, , iv] <- solve(outer(t, 0:3, "^"), B)                               # B = V c, V = Vandermonde in t
    }
    list(x = as.numeric(x), knots = as.numeric(br), coef = coef)
}

#' Build a MakeADFun-like object backed by the HIP engine
#'
#' @param sde SDE object (after initialize)
#' @param tmb_dat,tmb_par,map the lists SDE$setup has just assembled (R/sde.R:504-536, 621-632)
#' @return list(par, fn, gr, he, report, env) shaped like TMB::MakeADFun's value
#' @param random NULL (joint objective: every free entry is optimised, what tmb_obj_joint is, R/sde.R:666-668) or
#'   "coeff_re": the returned fn / gr are the Laplace marginal over coeff_re (ssde_laplace_eval), log_lambda is a free
#'   outer parameter, and env$last.par holds c(theta, u_hat) after every call -- the semantics of
#'   MakeADFun(..., random = "coeff_re") at R/sde.R:656-658
#' @param free_lambda joint object only: leave log_lambda free (what sdreport_hip needs for the joint Hessian)
#' @param devices integer vector of HIP device ordinals: one R process, several GPUs (tracks sharded inside the engine,
#'   one RCCL all-reduce per evaluation); NULL = the single device `device`
make_hip_obj <- function(sde, tmb_dat, tmb_par, map, device = NULL, random = NULL, devices = NULL, free_lambda = FALSE) {
    mats <- sde$make_mat()                      # X_list_fe / X_list_re / S_list (R/sde.R:452-454)
    kalman <- sde$type() %in% c("BM_SSM", "OU_SSM", "CTCRW")
    eseal <- sde$type() == "ESEAL_SSM"          # leading parameters log_tau, a1, log_a2 (nllk_e_seal_ssm.hpp:114-116)
    # full parameter vector in template order (nllk_ctcrw.hpp:135-140, nllk_sde.hpp:42-45);
    # without random effects TMB carries dummy log_lambda / coeff_re entries: the engine has none
    has_re <- !is.null(mats$S)
    has_decay <- length(tmb_dat$t_decay) > 1    # decaying response model: log_decay sits between log_lambda and coeff_re
    par_full <- c(if(kalman) tmb_par$log_sigma_obs, if(eseal) c(tmb_par$log_tau, tmb_par$a1, tmb_par$log_a2), tmb_par$coeff_fe,
                  if(has_re) tmb_par$log_lambda, if(has_decay) tmb_par$log_decay, if(has_re) tmb_par$coeff_re)
    fixed <- rep(FALSE, length(par_full))
    off_fe <- if(kalman) 1 else if(eseal) 3 else 0
    if(!is.null(map$coeff_fe)) fixed[off_fe + which(is.na(map$coeff_fe))] <- TRUE
    if(kalman && !is.null(map$log_sigma_obs)) fixed[1] <- TRUE
    laplace <- has_re && identical(random, "coeff_re")
    off_l <- off_fe + length(tmb_par$coeff_fe)
    off_re <- length(par_full) - length(tmb_par$coeff_re)
    if(has_re && !laplace && !free_lambda) {
        # joint objective at fixed smoothing parameters (log_lambda is not identifiable from the joint likelihood);
        # free_lambda = TRUE keeps it free: the joint object sdreport_hip differentiates at (theta_hat, u_hat)
        fixed[off_l + seq_along(tmb_par$log_lambda)] <- TRUE
    }
    if(has_re && !is.null(map$log_lambda)) fixed[off_l + which(is.na(map$log_lambda))] <- TRUE
    spec <- list(type = sde$type(), ID = as.numeric(sde$data()$ID), times = as.numeric(sde$data()$time),
                 obs = as.matrix(sde$obs()),
                 X_list_fe = lapply(seq_along(sde$formulas()), function(j) {
                     i1 <- sum(sde$terms()$ncol_fe[seq_len(j - 1)]); nj <- sde$terms()$ncol_fe[j]
                     as.matrix(sde$mats()$X_fe[(j - 1) * nrow(sde$data()) + seq_len(nrow(sde$data())),
                                               i1 + seq_len(nj), drop = FALSE])
                 }),
                 X_list_re = lapply(mats$X_list_re, function(x) if(ncol(x) > 0) as.matrix(x) else NULL),
                 a0 = tmb_dat$a0, P0 = tmb_dat$P0,
                 H = if(length(tmb_dat$H_array) > 1) tmb_dat$H_array else NULL,
                 par_fixed = fixed, include_penalty = tmb_dat$include_penalty, device = device,
                 exact_hess = isTRUE(laplace) && is.null(devices),   # SSDE_FLAG_EXACT_HESS: second derivatives wherever the batch is evaluated
                 devices = if(is.null(devices)) NULL else as.integer(devices),
                 other_data = if(sde$type() == "BM_t") as.numeric(tmb_dat$other_data) else NULL,
                 eseal_h = if(eseal) as.numeric(tmb_dat$h) else NULL, eseal_R = if(eseal) as.numeric(tmb_dat$R) else NULL,
                 t_decay = if(length(tmb_dat$t_decay) > 1) as.numeric(tmb_dat$t_decay) else NULL,
                 col_decay = as.integer(tmb_dat$col_decay), ind_decay = as.integer(tmb_dat$ind_decay),
                 # optional: list of length q (NULL entries = streamed block); the caller builds entry j with
                 # pp_table(gam_setup$smooth[[i]], data[[term]]) when parameter j has exactly one such smooth
                 basis_re = sde$other_data()$basis_re)
    # one penalty matrix per smooth: the diagonal blocks of S, sizes terms()$ncol_re (R/sde.R:424-447)
    if(has_re) {
        ncol_re <- sde$terms()$ncol_re; off <- cumsum(c(0, ncol_re)); S <- as.matrix(sde$mats()$S)
        spec$S_list <- lapply(seq_along(ncol_re), function(s) S[off[s] + seq_len(ncol_re[s]), off[s] + seq_len(ncol_re[s]), drop = FALSE])
    }
    ptr <- .Call("ssdeR_create", spec, PACKAGE = "smoothSDE")
    env <- new.env()
    # with random = "coeff_re" the optimiser sees the outer parameters only; the engine integrates coeff_re out
    is_u <- rep(FALSE, length(par_full))
    if(laplace) is_u[off_re + seq_along(tmb_par$coeff_re)] <- !fixed[off_re + seq_along(tmb_par$coeff_re)]
    free <- which(!fixed & !is_u)
    last <- new.env()
    eval_at <- function(x) {                    # fn(x) and gr(x) arrive separately with the same x
        if(is.null(last$x) || !identical(x, last$x)) {
            full <- par_full; full[free] <- x
            if(laplace) {
                if(!is.null(last$u)) full[is_u] <- last$u                          # warm start of the inner Newton solve
                last$res <- .Call("ssdeR_laplace", ptr, full, 1L, PACKAGE = "smoothSDE")
                # (a rejected probe -- value +Inf -- leaves the engine's coeff_re untouched; keep the last accepted u as the warm start)
                if(is.finite(last$res$value)) last$u <- last$res$par[is_u]
                env$last.par <- c(x, if(is.null(last$u)) full[is_u] else last$u)   # TMB's env$last.par: fixed, then random
                if(is.null(env$value.best) || last$res$value < env$value.best) {
                    env$value.best <- last$res$value; env$last.par.best <- env$last.par
                }
            } else {
                last$res <- .Call("ssdeR_eval", ptr, full, 1L, PACKAGE = "smoothSDE")  # (the engine memoises on x as well)
            }
            last$x <- x
        }
        last$res
    }
    env$last.par.best <- c(par_full[free], par_full[is_u])
    # TMB names obj$par by parameter block
    nm_full <- c(if(kalman) "log_sigma_obs", if(eseal) c("log_tau", "a1", "log_a2"), rep("coeff_fe", length(tmb_par$coeff_fe)),
                 if(has_re) rep("log_lambda", length(tmb_par$log_lambda)), if(has_decay) rep("log_decay", length(tmb_par$log_decay)),
                 if(has_re) rep("coeff_re", length(tmb_par$coeff_re)))
    list(par = setNames(par_full[free], nm_full[free]),
         fn = function(x = par_full[free]) eval_at(x)$value,
         gr = function(x = par_full[free]) matrix(eval_at(x)$gradient[free], nrow = 1),
         he = function(x = par_full[free]) {     # exact where the engine has second derivatives (ssde_hess; info()["exact_hess_scope"]), else
             if(!laplace) {                      # finite differences of the GPU gradient
                 full <- par_full; full[free] <- x
                 He <- .Call("ssdeR_hess", ptr, full, as.integer(free - 1L), PACKAGE = "smoothSDE")
                 if(!is.null(He)) return(He)
             }
             h <- 1e-5; p <- length(x)
             H <- matrix(0, p, p)
             for(k in seq_len(p)) {
                 e <- rep(0, p); e[k] <- h
                 H[, k] <- (eval_at(x + e)$gradient[free] - eval_at(x - e)$gradient[free]) / (2 * h)
             }
             (H + t(H)) / 2
         },
         report = function(x = par_full[free]) {
             full <- par_full; full[free] <- x
             list(aest_all = .Call("ssdeR_report", ptr, full, PACKAGE = "smoothSDE"))
         },
         env = env, ptr = ptr)
}

#' sdreport() for an object made by make_hip_obj (TMB::sdreport needs the tape inside a MakeADFun object)
#'
#' Returns the parts of TMB's report that the package reads (R/sde.R:707-719, 871-882, 1360-1375; R/utility.R:115-123):
#' par.fixed, par.random, cov.fixed, jointPrecision (names = TMB's parameter block names), plus value / gradient.fixed.
#' The assembly is TMB's own (sdreport with getJointPrecision = TRUE), with finite differences of the DEVICE gradient
#' where TMB has tapes (the same code as smoothsde_amd/report.py, which the GPU tests check against autograd Hessians):
#'   H       = d2 g / d(theta, u)^2 of the JOINT penalised nllk at (theta_hat, u_hat): central differences of obj_joint$gr
#'   Hfix    = Hessian of the marginal: stats::optimHess(par, obj$fn, obj$gr)  (obj = the random = "coeff_re" object)
#'   cov.fixed = Hfix^-1
#'   jointPrecision = [ Hfix + Htu Huu^-1 Hut , Htu ; Hut , Huu ]
#' @param obj        make_hip_obj(..., random = "coeff_re") after optim (obj$env$last.par.best = c(theta_hat, u_hat)),
#'                   or the joint object when the model has no random effects
#' @param obj_joint  make_hip_obj(..., random = NULL, free_lambda = TRUE) on the SAME tmb_dat as `obj` (penalty included),
#'                   or NULL without random effects
#' @param par_fixed  optim()$par
#' @param names_fixed,names_random  TMB block names of the entries ("coeff_fe", "log_lambda", ..., "coeff_re")
sdreport_hip <- function(obj, obj_joint = NULL, par_fixed, names_fixed, names_random = character(0), rel_step = 1e-4) {
    fd_hessian <- function(gr, x) {
        p <- length(x); H <- matrix(0, p, p)
        for(k in seq_len(p)) {
            e <- rep(0, p); e[k] <- rel_step * max(1, abs(x[k]))
            H[, k] <- (as.numeric(gr(x + e)) - as.numeric(gr(x - e))) / (2 * e[k])
        }
        (H + t(H)) / 2
    }
    nf <- length(par_fixed)
    if(is.null(obj_joint) || length(names_random) == 0) {
        H <- fd_hessian(obj$gr, par_fixed)
        cov <- solve(H); dimnames(cov) <- list(names_fixed, names_fixed)
        return(list(par.fixed = setNames(par_fixed, names_fixed), par.random = NULL, cov.fixed = cov, jointPrecision = NULL,
                    value = obj$fn(par_fixed), gradient.fixed = obj$gr(par_fixed), pdHess = all(eigen(H, TRUE, TRUE)$values > 0)))
    }
    u_hat <- obj$env$last.par.best[nf + seq_along(names_random)]
    x_joint <- c(par_fixed, u_hat)                      # obj_joint's free vector is ordered (theta, u) like TMB's last.par
    H <- fd_hessian(obj_joint$gr, x_joint)
    it <- seq_len(nf); iu <- nf + seq_along(u_hat)
    Htu <- H[it, iu, drop = FALSE]; Huu <- H[iu, iu, drop = FALSE]
    Hfix <- stats::optimHess(par_fixed, obj$fn, function(x) as.numeric(obj$gr(x)))
    cov <- solve(Hfix); dimnames(cov) <- list(names_fixed, names_fixed)
    Q <- H
    Q[it, it] <- Hfix + Htu %*% solve(Huu, t(Htu))
    nm <- c(names_fixed, names_random); dimnames(Q) <- list(nm, nm)
    list(par.fixed = setNames(par_fixed, names_fixed), par.random = setNames(u_hat, names_random), cov.fixed = cov,
         jointPrecision = Matrix::Matrix(Q, sparse = TRUE), value = obj$fn(par_fixed), gradient.fixed = obj$gr(par_fixed),
         pdHess = all(eigen(Hfix, TRUE, TRUE)$values > 0))
}
