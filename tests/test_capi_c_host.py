"""The drop-in boundary from PLAIN C: tests/capi_c/capi_smoke.c is compiled with `gcc -std=c99 -pedantic` against
include/ssde.h and linked with libssde_hip.so alone (no C++ runtime of its own, no Python, no torch) -- what the R
`.Call` shim of R_glue/ or any C host does.

CPU suite: it compiles without a warning, links, and fails loudly with SSDE_ERR_NODEVICE (no CPU fallback).
GPU suite: its numbers (ssde_eval value + gradient, ssde_laplace_eval, ssde_report, the fn-then-gr memo) equal what the
Python host gets from the same arrays through ctypes."""
import os
import subprocess

import numpy as np
import pytest

from smoothsde_amd import capi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "tests", "capi_c", "capi_smoke.c")
EXE = os.path.join(ROOT, "tests", "capi_c", "capi_smoke")


def _build():
    if not os.path.exists(capi.lib_path()):
        import __graft_entry__ as g
        g.build()
    libdir = os.path.dirname(capi.lib_path())
    r = subprocess.run(["gcc", "-std=c99", "-Wall", "-Wextra", "-pedantic", "-Werror", "-I", os.path.join(ROOT, "include"), SRC,
                        "-L", libdir, "-lssde_hip", "-lm", f"-Wl,-rpath,{libdir}", "-o", EXE], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    return EXE


def test_c_host_compiles_links_and_refuses_without_a_device():
    import torch
    exe = _build()
    if torch.cuda.device_count() > 0:
        pytest.skip("a GPU is present: the GPU suite runs the program")
    r = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert r.returncode == 3 and "no CPU fallback" in r.stdout, (r.returncode, r.stdout, r.stderr)


def _lcg_stream(n_values, seed=12345):
    out = np.empty(n_values)
    s = seed
    for i in range(n_values):
        s = (s * 6364136223846793005 + 1442695040888963407) & 0xFFFFFFFFFFFFFFFF
        out[i] = ((s >> 11) + 1) / 9007199254740994.0
    return out


@pytest.mark.gpu
@pytest.mark.parametrize("with_re", [1, 0])
def test_c_host_numbers_equal_the_python_host(with_re):
    exe = _build()
    n_tracks, rows = 9, 70
    r = subprocess.run([exe, str(n_tracks), str(rows), str(with_re)], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, (r.stdout, r.stderr)
    line = r.stdout.strip().splitlines()[-1]
    head, lap, counts, tail = [p.strip() for p in line.split("|")]
    nums = np.array(head.split(), dtype=float)
    c_val, c_grad = nums[0], nums[1:]
    n = n_tracks * rows
    u = _lcg_stream(4 * n)
    obs = np.zeros((n, 2))
    xre = np.zeros((n, 3))
    k = 0
    for m in range(n_tracks):
        x = y = 0.0
        for s in range(rows):
            i = m * rows + s
            x += u[k] - 0.5; y += u[k + 1] - 0.5
            obs[i, 0] = x + 0.1 * (u[k + 2] - 0.5)
            obs[i, 1] = y + 0.1 * (u[k + 3] - 0.5)
            k += 4
            for c in range(3):
                xre[i, c] = np.sin(0.05 * (s + 1) * (c + 1)) / (c + 1)
    ID = np.repeat(np.arange(n_tracks), rows).astype(float)
    times = np.arange(1.0, n + 1)
    S = np.array([[2.0, -1, 0], [-1, 2, -1], [0, -1, 2]])
    kw = dict(X_re=[None, None, xre, None], S_list=[S]) if with_re else {}
    pb = capi.Problem("CTCRW", ID, times, obs, **kw)
    par = np.zeros(pb.n_par_full)
    par[0], par[3], par[4] = np.log(0.2), 0.3, -0.1
    if with_re:
        par[-3:] = 0.05 * np.arange(1, 4)
    eng = capi.Engine(pb)
    v, g = eng.eval(par)
    assert c_val == v and np.array_equal(c_grad, g), (c_val, v)          # same library, same arrays: bitwise
    lv, _, _ = eng.laplace_eval(par, order=1)
    assert abs(float(lap) - lv) <= 1e-12 * max(1.0, abs(lv))
    n_evals, n_hits = (int(t) for t in counts.split())
    assert n_hits >= 1                                                   # gr(x) after fn(x) ... (or the value-only call on the row-varying path)
    a_last, same = tail.split()
    assert abs(float(a_last) - eng.report(par)[n - 1, 0]) <= 1e-12 * max(1.0, abs(float(a_last))) and same == "1"
    eng.close()
