"""GPU suite, LAST file (it sorts after every other test file): the kernel-coverage ledger of tests/conftest.py.

Every kernel family of include/ssde.h (SSDE_KERNEL_*) must have been compared with the oracle or a golden vector by at least ten
tests of this run, and -- for the families that cut tracks into time windows -- with several windows in at least three of them.
Dispatch is by the batch's rows; without this a test written for one family can silently run on another."""
import pytest

import conftest
from smoothsde_amd import capi

pytestmark = pytest.mark.gpu

N_FAMILIES = 17
# families that run time windows with a verified hand-over (the direct families score rows independently; the dense lanes walk a track)
WINDOWED = {3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 15, 16, 17}
MIN_COMPARED, MIN_WINDOWED = 10, 3


def test_every_kernel_family_was_compared_with_the_oracle():
    files = conftest._CUR["files"]
    if len([f for f in files if f.startswith("test_gpu_")]) < 12:
        pytest.skip("the ledger needs the whole GPU suite in one run (this run covered %d of its files)" % len(files))
    rows, bad = [], []
    for kid in range(1, N_FAMILIES + 1):
        e = conftest.LEDGER.get(kid, dict(evals=0, compared=0, compared_windowed=0, tests=set()))
        rows.append("%2d %-48s evals %6d  compared %4d  with windows %4d  files %s" % (
            kid, capi.KERNEL_NAMES.get(kid, "?"), e["evals"], e["compared"], e["compared_windowed"], ",".join(sorted(e["tests"]))))
        if e["compared"] < MIN_COMPARED:
            bad.append("%s: %d oracle comparisons" % (capi.KERNEL_NAMES.get(kid, kid), e["compared"]))
        if kid in WINDOWED and e["compared_windowed"] < MIN_WINDOWED:
            bad.append("%s: %d oracle comparisons with several windows" % (capi.KERNEL_NAMES.get(kid, kid), e["compared_windowed"]))
    table = "\n".join(rows)
    print("\nkernel-coverage ledger (id, family, evaluations, tests that compared it with the oracle, ... with several windows)\n" + table)
    import os
    out = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
    if os.path.isdir(out):
        with open(os.path.join(out, "kernel_ledger.txt"), "w") as fh:
            fh.write(table + "\n")
    assert not bad, "kernel families the suite did not check against the oracle:\n  " + "\n  ".join(bad) + "\n" + table
