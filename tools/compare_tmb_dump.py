#!/usr/bin/env python3
"""tools/compare_tmb_dump.py tmb_dump.json [tests/golden/unstable_cases.json] -- compare the TRUE TMB numbers written by tools/tmb_oracle.R (on a machine with
R + TMB + smoothSDE) with the committed expectations of tests/golden/cases.json and with the CPU oracle.
Tolerances: value 1e-8 relative (north-star bar), gradient 1e-8 * max|g| + 1e-10.  Exit code 1 on any miss.  This is
the step that turns "parity unpinned" into "pinned" -- it cannot run in the build image (no R)."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from cases import problem_from_spec          # noqa: E402
from golden_io import load_golden            # noqa: E402
from oracle_lib import oracle_eval           # noqa: E402

dump = json.load(open(sys.argv[1]))
bad = 0
if len(sys.argv) > 2:
    # tests/golden/unstable_cases.json (tests/golden/gen_unstable.py): fixtures where the reference's covariance recursion amplifies
    # rounding.  No pass / fail here -- the point is WHICH number TMB's double arithmetic lands near: the literal
    # restatement in double (both are rounding noise of the same unstable recursion: expect agreement to a few digits only in the two
    # `unstable_*` cases, to 1e-12 in the `stable_*` ones), its binary128 evaluation, or the restatement with P kept symmetric.
    from golden_io import dec
    for raw in json.load(open(sys.argv[2])):
        rec = dec(raw)
        t = dump.get(rec["name"])
        if t is None:
            print(f"{rec['name']:24s} missing from the dump")
            continue
        tv, e = float(t["value"]), rec["expected"]
        rel = lambda a: abs(tv - a) / abs(a)        # noqa: E731
        print(f"{rec['name']:24s} TMB {tv:.12f}: from the literal double restatement {rel(e['value']):.1e}, from binary128 {rel(e['binary128_value']):.1e}, "
              f"from the arbiter (P symmetric) {rel(e['arbiter_value']):.1e}")
    sys.exit(0)
for rec in load_golden():
    t = dump.get(rec["name"])
    if t is None:
        print(f"{rec['name']:32s} missing from the dump")
        bad += 1
        continue
    tv, tg = float(t["value"]), np.asarray(t["gradient"], dtype=float)
    ev, eg = rec["expected"]["value"], np.asarray(rec["expected"]["grad"], dtype=float)
    ov, og = oracle_eval(problem_from_spec(rec), rec["par"], order=1)
    ok = (abs(ev - tv) <= 1e-8 * max(1.0, abs(tv)) and np.max(np.abs(eg - tg)) <= 1e-8 * np.max(np.abs(tg)) + 1e-10 and
          abs(ov - tv) <= 1e-8 * max(1.0, abs(tv)) and np.max(np.abs(og - tg)) <= 1e-8 * np.max(np.abs(tg)) + 1e-10)
    print(f"{rec['name']:32s} TMB {tv:.15g}  fixture {ev:.15g}  oracle {ov:.15g}  "
          f"max|dgrad| fixture {np.max(np.abs(eg - tg)):.2e} oracle {np.max(np.abs(og - tg)):.2e}  {'ok' if ok else 'MISMATCH'}")
    bad += 0 if ok else 1
print("mismatches", bad)
sys.exit(1 if bad else 0)
