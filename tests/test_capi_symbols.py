"""CPU suite: the C-ABI library loads and exports every symbol include/ssde.h declares
(no compute calls: there is no GPU here), and creating an engine without a device fails loudly."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from smoothsde_amd import capi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    src = open(os.path.join(ROOT, "include", "ssde.h")).read()
    return sorted(set(re.findall(r"\b(ssde_[a-z_]+)\s*\(", src)))


def test_header_symbols_are_exported():
    if not os.path.exists(capi.lib_path()):
        import __graft_entry__ as g
        g.build()
    lib = C.CDLL(capi.lib_path())
    declared = _declared_symbols()
    assert set(declared) == set(capi.EXPORTED_SYMBOLS), declared
    for name in declared:
        assert hasattr(lib, name), name
    assert capi.load_library().ssde_abi_version() == capi.ABI_VERSION


def test_desc_struct_matches_header_layout():
    # field order of the ctypes mirror == field order of the C struct
    src = open(os.path.join(ROOT, "include", "ssde.h")).read()
    body = src[src.index("typedef struct ssde_desc {"):src.index("} ssde_desc;")]
    body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
    fields = re.findall(r"\b([A-Za-z_0-9]+)\s*;", body)
    assert fields == [f for f, _ in capi.SsdeDesc._fields_]
    body = src[src.index("typedef struct ssde_info_t {"):src.index("} ssde_info_t;")]
    body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
    fields = re.findall(r"\b([A-Za-z_0-9]+)\s*;", body)
    assert fields == [f for f, _ in capi.SsdeInfo._fields_]


def test_no_cpu_fallback():
    """Without a GPU the product path must fail, never silently fall back to a CPU evaluation."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    pb = capi.Problem("BM_SSM", np.zeros(4), np.arange(4.0), np.arange(4.0)[:, None])
    with pytest.raises(capi.EngineError) as e:
        capi.Engine(pb)
    assert "no CPU fallback" in str(e.value) or "HIP" in str(e.value) or "device" in str(e.value)


def test_unknown_and_unsupported_types():
    with pytest.raises(ValueError, match="Unknown SDE type"):
        capi.Problem("XYZ", np.zeros(3), np.arange(3.0), np.zeros((3, 1)))
    for t in capi.UNSUPPORTED_MODELS:
        with pytest.raises(NotImplementedError):
            capi.Problem(t, np.zeros(3), np.arange(3.0), np.zeros((3, 1)))


def test_product_never_imports_oracle():
    pkg = os.path.join(ROOT, "smoothsde_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".cpp", ".h")) or f == "Makefile":
                txt = open(os.path.join(dirpath, f)).read()
                assert "oracle_lib" not in txt and "liboracle" not in txt and "oracle/" not in txt.replace(
                    "oracle/ is test", "").replace("under oracle/", ""), os.path.join(dirpath, f)


def test_scripts_compile():
    """bench.py, __graft_entry__.py and every tools/*.py at least parse (they only run on the GPU box)."""
    import glob
    import py_compile
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    files = [os.path.join(root, "bench.py"), os.path.join(root, "__graft_entry__.py")] + sorted(glob.glob(os.path.join(root, "tools", "*.py")))
    assert len(files) > 8
    for f in files:
        py_compile.compile(f, doraise=True)


def test_descriptor_checks_precede_the_device():
    """Malformed descriptors are rejected before anything touches a GPU (so this runs on the CPU box too): a decaying
    column index outside coeff_re would otherwise silently not decay while log_decay stayed a free parameter with a
    zero gradient (the reference stops on an unknown name, R/sde.R:637-640); a duplicate likewise."""
    rng = np.random.default_rng(0)
    n = 40
    ID = np.repeat([0, 1], 20)
    X = rng.standard_normal((n, 3))
    S = np.eye(3)
    td = np.tile(np.linspace(0, 1, n), 3)
    pb = capi.Problem("OU", ID, np.arange(1.0, n + 1), rng.standard_normal((n, 1)), X_re=[X, None, None], S_list=[S],
                      t_decay=td, col_decay=[0, 1], ind_decay=[0, 0])
    pb.col_decay = np.array([0, 7], dtype=np.int32)            # past the three columns of coeff_re
    with pytest.raises(capi.EngineError, match="col_decay out of range"):
        capi.Engine(pb)
    pb.col_decay = np.array([1, 1], dtype=np.int32)
    with pytest.raises(capi.EngineError, match="twice"):
        capi.Engine(pb)
