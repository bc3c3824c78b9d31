"""The C-ABI library loads and exports every symbol include/proslam_hip.h declares (no GPU needed)."""
import ctypes as C
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    names = []
    inc = os.path.join(ROOT, "include")
    for fn in sorted(os.listdir(inc)):
        if fn.endswith(".h"):
            text = open(os.path.join(inc, fn)).read()
            names += re.findall(r"PRS_API\s+[\w\s\*]+?\b(prs_\w+)\s*\(", text)
    return sorted(set(names))


@pytest.fixture(scope="module")
def built():
    import __graft_entry__ as g
    g.build()
    from srrg2_proslam_amd import _lib
    return _lib


def test_header_declares_entry_points():
    names = declared_symbols()
    for must in ("prs_context_create", "prs_stereo_match", "prs_stereo_match_batch", "prs_triangulate"):
        assert must in names


def test_library_exports_every_declared_symbol(built):
    lib = C.CDLL(built.LIB_PATH)
    missing = [n for n in declared_symbols() if not hasattr(lib, n)]
    assert not missing, "declared in include/*.h but not exported: %s" % missing


def test_python_binding_covers_header(built):
    assert sorted(built.SYMBOLS.keys()) == declared_symbols()
    lib = built.load()
    assert lib.prs_version() == built.ABI_VERSION
    header = open(os.path.join(ROOT, "include", "proslam_hip.h")).read()
    assert "#define PRS_ABI_VERSION %d" % built.ABI_VERSION in header
    # a client built against another header, or with a shorter parameter struct, is refused instead of read past
    sizes = [C.sizeof(built.StereoParams), C.sizeof(built.PcfParams), C.sizeof(built.AlignerParams), C.sizeof(built.AlignBatch)]
    assert lib.prs_abi_check(built.ABI_VERSION, *sizes) == 0
    assert lib.prs_abi_check(built.ABI_VERSION - 1, *sizes) < 0
    assert lib.prs_abi_check(built.ABI_VERSION, sizes[0], sizes[1], sizes[2] - 12, sizes[3]) < 0
    assert lib.prs_status_string(-1).decode().startswith("required")


def test_context_create_without_gpu_fails_loudly(built):
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    lib = built.load()
    h = C.c_void_p()
    rc = lib.prs_context_create(0, C.byref(h))
    assert rc < 0  # PRS_ERR_NO_DEVICE (or HIP error): never a silent CPU path
    from srrg2_proslam_amd import ops
    with pytest.raises(built.ProslamHipError):
        ops.Context(0)


def test_product_package_never_touches_the_oracle():
    """the oracle is test infrastructure: nothing under srrg2_proslam_amd/ may reference it"""
    pkg = os.path.join(ROOT, "srrg2_proslam_amd")
    bad = []
    for dirpath, _, files in os.walk(pkg):
        for fn in files:
            if fn.endswith((".py", ".hip", ".h", ".cpp", ".hpp")) or fn == "Makefile":
                text = open(os.path.join(dirpath, fn), errors="ignore").read()
                if re.search(r"\boracle\b", text):
                    bad.append(os.path.relpath(os.path.join(dirpath, fn), ROOT))
    assert not bad, "product files mention the oracle: %s" % bad
