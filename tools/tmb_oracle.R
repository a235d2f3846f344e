#!/usr/bin/env Rscript
# tools/tmb_oracle.R -- dump the TRUE reference numbers (TMB fn / gr) for the committed fixtures.
#
# Written blind: this image has no R / TMB / smoothSDE.  On a machine that has them
# (install.packages(c("TMB","mgcv","R6","jsonlite")); devtools::install_github("TheoMichelot/smoothSDE")):
#     Rscript tools/tmb_oracle.R tests/golden/cases.json tmb_dump.json
# then compare tmb_dump.json with the "expected" blocks of cases.json (value rel <= 1e-8, gradient
# |d| <= 1e-8 * max|g| + 1e-10).  Only constant-coefficient cases are dumped (the fixtures' spline
# blocks are this repository's B-spline stand-in, not an mgcv basis).
suppressMessages({library(smoothSDE); library(jsonlite)})
args <- commandArgs(trailingOnly = TRUE)
cases <- fromJSON(args[1], simplifyVector = FALSE)
hex2dbl <- function(h) {
    raw <- as.raw(strtoi(substring(h, seq(1, 15, 2), seq(2, 16, 2)), 16L))
    readBin(rev(raw), "double", n = 1, size = 8, endian = "little")
}
dec <- function(x) { v <- vapply(x$hex, hex2dbl, 0.0); if(length(x$shape) == 2) matrix(v, x$shape[[1]], x$shape[[2]], byrow = TRUE) else v }
out <- list()
for(cs in cases) {
    if(!is.null(cs$X_fe) || !is.null(cs$H) || !is.null(cs$P0)) next
    obs <- dec(cs$obs); d <- ncol(obs)
    resp <- paste0("z", seq_len(d))
    dat <- data.frame(ID = dec(cs$ID), time = dec(cs$times)); dat[resp] <- obs
    fix <- NULL
    sde <- SDE$new(data = dat, type = cs$model, response = resp)
    sde$setup()
    obj <- sde$tmb_obj()
    par <- dec(cs$par)
    if(!is.null(cs$par_fixed)) {   # re-setup with the same fixed coefficients
        nm <- names(sde$formulas())[which(unlist(cs$par_fixed$u8)[-1][seq_along(sde$formulas())] == 1)]
        sde <- SDE$new(data = dat, type = cs$model, response = resp, fixpar = nm,
                       par0 = NULL); sde$setup(); obj <- sde$tmb_obj()
    }
    free <- if(is.null(cs$par_fixed)) seq_along(par) else which(unlist(cs$par_fixed$u8) == 0)
    x <- par[free]
    out[[cs$name]] <- list(value = obj$fn(x), gradient = as.numeric(obj$gr(x)), free = free)
}
writeLines(toJSON(out, digits = 17, auto_unbox = TRUE), args[2])
