"""The R .Call shim (R_glue/src/ssde_rcall.c) cannot be compiled against R in this image (no R, no Rinternals.h).  This
test only PARSES it: `gcc -fsyntax-only -Wall -Werror` against include/ssde.h and a test-only declaration header of the R C-API
symbols it uses (tests/r_stub/, which is NOT R and proves nothing about R).  What it does catch: drift between the C ABI --
ssde_desc, ssde_info_t, the entry points' signatures -- and the shim at every ABI bump (src/init.c:22-35 is the registration
pattern the shim follows)."""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SHIM = os.path.join(ROOT, "R_glue", "src", "ssde_rcall.c")


@pytest.mark.skipif(shutil.which("gcc") is None, reason="no gcc")
def test_the_r_shim_parses_against_the_c_abi():
    r = subprocess.run(["gcc", "-fsyntax-only", "-Wall", "-Werror", "-std=c99", "-I", os.path.join(ROOT, "tests", "r_stub"),
                        "-I", os.path.join(ROOT, "include"), SHIM], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr


def test_every_registered_entry_point_is_defined_with_that_many_arguments():
    src = open(SHIM).read()
    table = re.findall(r'\{"(ssdeR_\w+)",\s*\(DL_FUNC\)&(\w+),\s*(\d+)\}', src)
    assert len(table) >= 6
    for name, fun, nargs in table:
        assert name == fun
        m = re.search(r"^SEXP " + fun + r"\(([^)]*)\)\s*\{", src, re.M)
        assert m, f"{fun} is registered but not defined"
        assert len([a for a in m.group(1).split(",") if a.strip()]) == int(nargs), fun
    # and the R glue calls nothing that is not registered
    glue = open(os.path.join(ROOT, "R_glue", "R", "backend_hip.R")).read()
    called = set(re.findall(r'\.Call\("?(ssdeR_\w+)"?', glue))
    assert called and called <= {n for n, _, _ in table}, called - {n for n, _, _ in table}


def test_the_stub_headers_say_what_they_are():
    for f in ("R.h", "Rinternals.h", os.path.join("R_ext", "Rdynload.h")):
        head = open(os.path.join(ROOT, "tests", "r_stub", f)).read(300)
        assert "TEST-ONLY" in head and "NOT R" in head
