#!/usr/bin/env python3
"""tests/golden/gen_unstable.py -- writes tests/golden/unstable_cases.json: four small CTCRW fixtures (2 tracks x 200 rows, d = 2) in the
format of cases.json, for `Rscript tools/tmb_oracle.R tests/golden/unstable_cases.json dump.json` on a machine with R + TMB + smoothSDE:
a measurement covariance that COUPLES the response columns, a P0 with entries between the dimensions, and their diagonal / default
counterparts.  `expected` holds the literal restatement in double (what TMB's arithmetic should give up to the order of its own
floating-point operations -- i.e. NOT reproducibly, in the two coupling cases), the same in binary128, and the restatement in arbiter
mode (P kept symmetric); DESIGN.md 5c, tests/test_oracle_golden.py::test_reference_form_loses_...
`python tools/compare_tmb_dump.py dump.json tests/golden/unstable_cases.json` prints which of the three TMB agrees with."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from gen_golden import enc  # noqa: E402
from oracle_lib import keep_P_symmetric, oracle_eval, oracle_eval_quad  # noqa: E402
from smoothsde_amd import capi  # noqa: E402
from smoothsde_amd.synth import simulate  # noqa: E402


def main():
    ID, times, obs = simulate("CTCRW", 2, 200, 2, tau=2.0, nu=1.0, sigma_obs=0.1, seed=13)
    n = len(ID)
    A = np.random.default_rng(3).standard_normal((4, 4))
    H = lambda m: np.ascontiguousarray(np.transpose(np.tile(np.array(m), (n, 1, 1)), (1, 2, 0)))      # noqa: E731
    cases = [("unstable_H_coupling", dict(H=H([[0.005, 0.002], [0.002, 0.004]])), [0.0, 0.0, 0.0, np.log(2.0), 0.0]),
             ("stable_H_diagonal", dict(H=H([[0.005, 0.0], [0.0, 0.004]])), [0.0, 0.0, 0.0, np.log(2.0), 0.0]),
             ("unstable_P0_coupling", dict(P0=A @ A.T + np.eye(4)), [np.log(0.07), 0.0, 0.0, np.log(2.0), 0.0]),
             ("stable_P0_default", dict(), [np.log(0.07), 0.0, 0.0, np.log(2.0), 0.0])]
    out = []
    for name, kw, par in cases:
        par = np.array(par)
        fixed = np.array([1 if "H" in kw else 0, 1, 1, 0, 0], dtype=np.uint8)
        pb = capi.Problem("CTCRW", ID, times, obs, par_fixed=fixed, **kw)
        lit, lit_g = oracle_eval(pb, par, order=1)
        quad = oracle_eval_quad(pb, par, order=0)
        keep_P_symmetric(True)
        try:
            arb, arb_g = oracle_eval(pb, par, order=1)
        finally:
            keep_P_symmetric(False)
        spec = dict(name=name, model="CTCRW", n_dim=2, ID=ID, times=times, obs=obs, X_fe=None, X_re=None, S_list=None, a0=None,
                    P0=kw.get("P0"), H=kw.get("H"), par_fixed=fixed, na_mode=1, include_penalty=1, par=par)
        rec = {k: enc(v) for k, v in spec.items()}
        rec["expected"] = dict(value=lit, grad=enc(lit_g), binary128_value=quad, arbiter_value=arb, arbiter_grad=enc(arb_g))
        out.append(rec)
        print(f"{name:24s} literal double {lit:.12f}  binary128 {quad:.12f}  arbiter {arb:.12f}  |literal - binary128| / |.| = {abs(lit - quad) / abs(quad):.1e}")
    path = os.path.join(ROOT, "tests", "golden", "unstable_cases.json")
    with open(path, "w") as f:
        json.dump(out, f)
    print("wrote", path, os.path.getsize(path) // 1024, "kB")


if __name__ == "__main__":
    main()
