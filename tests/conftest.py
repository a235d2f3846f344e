import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_specs():
    from golden_io import load_golden
    return load_golden()


# ---- kernel-coverage ledger (GPU suite) ---------------------------------------------------------------------------------------
# Dispatch is by the batch's rows over 17 kernel families: a test that means to check one family can silently be sent to another
# (round 4: a double-counted intercept lived a round because the size rules kept every fuzz batch off iso_colvar_kernel).  Every
# Engine.eval of a GPU test is recorded with the kernel family that ran it (ssde_info.kernel_id) and whether it ran several time
# windows; a test that also evaluated the oracle (oracle_lib.oracle_eval / oracle_report) or built its problem from a golden record
# (cases.problem_from_spec) counts as one oracle comparison for every family it evaluated.  tests/test_zz_gpu_ledger.py, which sorts
# last, asserts that every family was compared often enough and prints the table.
LEDGER = {}          # kernel_id -> dict(evals, compared, compared_windowed, tests=set())
_CUR = {"evals": set(), "files": set()}
_HOOKED = [False]


def _hook_once():
    if _HOOKED[0]:
        return
    _HOOKED[0] = True
    from smoothsde_amd import capi
    orig_eval = capi.Engine.eval

    def eval_recorded(self, par, order=1):
        out = orig_eval(self, par, order)
        try:
            inf = self.info()
            kid = int(inf["kernel_id"])
            _CUR["evals"].add((kid, bool(inf["window"] > 0 and inf["lanes_per_track"] > 1)))
            LEDGER.setdefault(kid, dict(evals=0, compared=0, compared_windowed=0, tests=set()))["evals"] += 1
        except Exception:
            pass
        return out

    capi.Engine.eval = eval_recorded


def _oracle_calls():
    import cases
    import oracle_lib
    return oracle_lib.CALLS[0] + cases.SPEC_CALLS[0]


@pytest.fixture(autouse=True)
def _kernel_ledger(request):
    if request.node.get_closest_marker("gpu") is None:
        yield
        return
    _hook_once()
    _CUR["evals"] = set()
    before = _oracle_calls()
    yield
    _CUR["files"].add(request.node.fspath.basename)
    if _oracle_calls() > before:
        for kid, windowed in _CUR["evals"]:
            e = LEDGER.setdefault(kid, dict(evals=0, compared=0, compared_windowed=0, tests=set()))
            e["compared"] += 1
            e["compared_windowed"] += 1 if windowed else 0
            e["tests"].add(request.node.fspath.basename)
