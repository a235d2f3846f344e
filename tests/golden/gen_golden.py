#!/usr/bin/env python3
"""Generate tests/golden/cases.json -- run from the repo root: python tests/golden/gen_golden.py

For every case of tests/cases.py this script
  1. evaluates the C++ oracle (oracle/liboracle.so: value, gradient, aest_all),
  2. evaluates the INDEPENDENT restatement of tests/refimpl.py (dense joint Gaussian /
     torch.distributions + autograd),
  3. refuses to write the fixture unless both agree (value rel 1e-10, gradient
     |d| <= 1e-8 * max|g| + 1e-10), and
  4. stores inputs + oracle outputs + the independent outputs.

The reference package itself cannot produce these numbers here (no R / TMB in the image);
tools/tmb_oracle.R dumps the same quantities on a machine that has them.
"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from cases import all_specs, problem_from_spec  # noqa: E402
from oracle_lib import oracle_eval  # noqa: E402
from refimpl import ref_eval  # noqa: E402


def enc(x):
    if x is None:
        return None
    if isinstance(x, (list, tuple)):
        return [enc(v) for v in x]
    if isinstance(x, np.ndarray):
        a = np.asarray(x)
        if a.dtype == np.uint8:
            return {"u8": a.tolist()}
        bits = a.astype(np.float64).view(np.uint64)
        # NaN payloads (R's NA_real_) do not survive JSON floats: keep exact bit patterns as hex
        return {"shape": list(a.shape), "hex": [format(int(b), "016x") for b in bits.flatten(order="C")]}
    if isinstance(x, (np.floating, np.integer)):
        return x.item()
    return x


def main():
    out = []
    worst_v, worst_g = 0.0, 0.0
    for spec in all_specs():
        pb = problem_from_spec(spec)
        par = spec["par"]
        if pb.kalman and pb.model != "ESEAL_SSM":     # (the ESEAL template has no REPORT)
            val, grad, aest = oracle_eval(pb, par, order=1, report=True)
        else:
            val, grad = oracle_eval(pb, par, order=1)
            aest = None
        rval, rgrad = ref_eval(pb, par)
        ev = abs(val - rval) / max(1.0, abs(rval))
        eg = np.max(np.abs(grad - rgrad)) / (1e-8 * np.max(np.abs(rgrad)) + 1e-10)
        worst_v, worst_g = max(worst_v, ev), max(worst_g, eg)
        print(f"{spec['name']:36s} n={pb.n:3d} p={pb.n_par_full:2d}  nllk={val: .12e}  dv={ev:.1e}  dg/tol={eg:.2e}")
        assert ev < 1e-10, (spec["name"], val, rval)
        assert eg < 1.0, (spec["name"], grad, rgrad)
        rec = {k: enc(v) for k, v in spec.items()}
        rec["expected"] = dict(value=val, grad=enc(grad), aest_all=enc(aest),
                               indep_value=rval, indep_grad=enc(rgrad))
        out.append(rec)
    path = os.path.join(ROOT, "tests", "golden", "cases.json")
    with open(path, "w") as f:
        json.dump(out, f)
    print(f"wrote {len(out)} cases to {path} ({os.path.getsize(path)/1024:.0f} kB); "
          f"worst value rel diff {worst_v:.2e}, worst grad diff/tol {worst_g:.2e}")


if __name__ == "__main__":
    main()
