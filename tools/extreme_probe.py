#!/usr/bin/env python3
"""Probe (GPU box): random problems of the fuzz suite evaluated with one coordinate pushed to an extreme or NaN.

Every disagreement with the double-precision oracle (finiteness, value 1e-9, gradient 1e-7) is ARBITRATED by the same
restated templates evaluated in IEEE binary128 (oracle/oracle_quad.cpp: the exact value of the reference's formulas at
these inputs, to 1e-30).  At such parameters the reference's own formulas cancel catastrophically (P (T - K Z)' once
P >> H, makeQ_ctcrw at beta dt -> 0, 1 - exp(-2 dt / tau) at dt / tau -> 0), so a literal double evaluation -- the
oracle, and TMB itself -- is noise of some relative size e_oracle.  A disagreement counts as an ISSUE only when the
engine is further from the exact value than that noise allows:
    e_engine > max(10 * e_oracle, 1e-9)      (value; same rule on the gradient with 1e-7)
Otherwise it is printed as ILL-CONDITIONED with both errors.  Progress goes to stdout line by line."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from smoothsde_amd import capi                      # noqa: E402
from oracle_lib import oracle_eval, oracle_eval_quad  # noqa: E402
from test_gpu_fuzz import random_problem            # noqa: E402

bad = 0
table = []
t00 = time.time()
for seed in range(int(sys.argv[1]) if len(sys.argv) > 1 else 64):
    pb, par0 = random_problem(seed)
    eng = capi.Engine(pb)
    rng = np.random.default_rng(1000 + seed)
    free = np.flatnonzero(pb.par_fixed == 0)
    for trial in range(6):
        k = int(rng.choice(free))
        delta = [-100.0, -30.0, 30.0, 100.0, np.nan, -12.0][trial]
        par = par0.copy()
        par[k] = delta
        t0 = time.time()
        v, g = eng.eval(par)
        tg = time.time() - t0
        ov, og = oracle_eval(pb, par, order=1, threads=4)
        okv = (np.isfinite(v) == np.isfinite(ov)) and (not np.isfinite(ov) or abs(v - ov) <= 1e-9 * max(1, abs(ov)))
        gf = np.all(np.isfinite(g)) == np.all(np.isfinite(og))
        okg = gf and (not np.all(np.isfinite(og)) or np.max(np.abs(g - og)) <= 1e-7 * np.max(np.abs(og)) + 1e-9)
        if okv and okg and tg <= 2.0:
            continue
        if tg > 2.0:
            bad += 1
            print("SLOW", seed, pb.model, pb.n, "par", k, delta, "t", round(tg, 2), flush=True)
            continue
        finite = np.isfinite(ov) and np.all(np.isfinite(og)) and np.isfinite(v) and np.all(np.isfinite(g))
        verdict = "ISSUE"
        detail = ""
        if finite:
            qv, qg = oracle_eval_quad(pb, par)
            sv, sg = max(1.0, abs(qv)), max(np.max(np.abs(qg)), 1e-300)
            ev_e, ev_o = abs(v - qv) / sv, abs(ov - qv) / sv
            eg_e, eg_o = np.max(np.abs(g - qg)) / sg, np.max(np.abs(og - qg)) / sg
            detail = f"value err engine {ev_e:.2e} oracle {ev_o:.2e} | grad err engine {eg_e:.2e} oracle {eg_o:.2e} | exact {qv!r}"
            if ev_e <= max(10 * ev_o, 1e-9) and eg_e <= max(10 * eg_o, 1e-7):
                verdict = "ILL-CONDITIONED"
            table.append((seed, pb.model, pb.n, k, delta, ev_e, ev_o, eg_e, eg_o))
        if verdict == "ISSUE":
            bad += 1
        print(verdict, seed, pb.model, pb.n, "par", k, delta, "gpu", v, "oracle", ov, "path", eng.info()["path"], detail,
              "\n   g", g, "\n  og", og, flush=True)
    eng.close()
    print("seed", seed, pb.model, "done", round(time.time() - t00, 1), flush=True)
print("| seed | model | rows | par | value | engine value err | oracle value err | engine grad err | oracle grad err |")
print("|---|---|---|---|---|---|---|---|---|")
for r in table:
    print(f"| {r[0]} | {r[1]} | {r[2]} | {r[3]} | {r[4]:g} | {r[5]:.1e} | {r[6]:.1e} | {r[7]:.1e} | {r[8]:.1e} |")
print("issues", bad)
