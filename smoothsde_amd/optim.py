"""optim(method = "BFGS") as the reference calls it (`optim(par, fn, gr, method = "BFGS")`, /root/reference/R/sde.R:694-697).

R's BFGS is `vmmin` (Nash, Compact Numerical Methods, algorithm 21): inverse-Hessian update, a pure BACKTRACKING line
search (step 1, reduced by 0.2 until the Armijo test with 1e-4 passes; a non-finite value simply fails the test),
restart with the identity when the update would lose positive definiteness, relative tolerance sqrt(eps) on the
decrease, at most 100 iterations.  Restated here from the published algorithm because the behaviour matters for a
drop-in: a line search that extrapolates (scipy's strong-Wolfe search) walks log-scale parameters into overflow from
the reference's own starting values, where vmmin backs off."""
from __future__ import annotations

from typing import Callable, Dict

import numpy as np


def optim_bfgs(fn: Callable[[np.ndarray], float], gr: Callable[[np.ndarray], np.ndarray], par, maxit: int = 100,
               reltol: float = float(np.sqrt(np.finfo(float).eps)), abstol: float = -np.inf) -> Dict:
    """Returns dict(par, value, counts=(fn, gr), convergence) with optim()'s meaning: convergence 0 = converged,
    1 = iteration limit reached."""
    stepredn, acctol, reltest = 0.2, 1e-4, 10.0
    b = np.array(par, dtype=np.float64)
    n = len(b)
    if n == 0:
        return dict(par=b, value=float(fn(b)), counts=(1, 0), convergence=0)
    f = float(fn(b))
    if not np.isfinite(f):
        raise ValueError("initial value in 'vmmin' is not finite")            # optim()'s own error
    fmin = f
    funcount = gradcount = 1
    g = np.array(gr(b), dtype=np.float64)
    it = 1
    ilast = gradcount
    B = np.eye(n)
    count = 0
    while True:
        if ilast == gradcount:
            B = np.eye(n)
        X = b.copy()
        c = g.copy()
        t = -B @ g
        gradproj = float(t @ g)
        if gradproj < 0.0:                                   # a descent direction
            steplength = 1.0
            accpoint = False
            while True:
                b = X + steplength * t
                count = int(np.sum(reltest + X == reltest + b))
                if count < n:
                    f = float(fn(b))
                    funcount += 1
                    accpoint = np.isfinite(f) and f <= fmin + gradproj * steplength * acctol
                    if not accpoint:
                        steplength *= stepredn
                if count == n or accpoint:
                    break
            enough = f > abstol and abs(f - fmin) > reltol * (abs(fmin) + reltol)
            if not enough:                                   # no progress worth the name: stop
                count = n
                fmin = f if accpoint else fmin
            if count < n:
                fmin = f
                g = np.array(gr(b), dtype=np.float64)
                gradcount += 1
                it += 1
                t = steplength * t
                c = g - c
                d1 = float(t @ c)
                if d1 > 0.0:
                    Xc = B @ c
                    d2 = 1.0 + float(Xc @ c) / d1
                    B = B + (d2 * np.outer(t, t) - np.outer(Xc, t) - np.outer(t, Xc)) / d1
                else:                                        # curvature condition lost: restart from the identity
                    ilast = gradcount
            else:
                if not accpoint:
                    b = X                                    # no acceptable point on this line: stay
                if ilast < gradcount:
                    count = 0
                    ilast = gradcount
        else:                                                # not a descent direction: restart, or stop if just restarted
            count = 0
            if ilast == gradcount:
                count = n
            else:
                ilast = gradcount
        if it >= maxit:
            break
        if gradcount - ilast > 2 * n:
            ilast = gradcount                                # periodic restart
        if count == n and ilast == gradcount:
            break
    return dict(par=b, value=fmin, counts=(funcount, gradcount), convergence=int(it >= maxit))
