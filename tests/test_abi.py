"""The C-ABI shared library loads without a GPU and exports every symbol that
include/markovmodels_amd.h declares (no compute calls here)."""
import ctypes as C
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "markovmodels_amd.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(mm_[a-z0-9_]+)\s*\(", text)))


def test_header_symbols_are_exported(mm):
    syms = declared_symbols()
    assert len(syms) >= 14
    lib = C.CDLL(mm.LIB_PATH)
    for s in syms:
        assert hasattr(lib, s), f"{s} declared in the header but not exported"
    assert sorted(mm.SYMBOLS) == syms, "the ctypes binding must bind exactly the header's entry points"
    lib.mm_abi_version.restype = C.c_int
    assert lib.mm_abi_version() == 3


def test_no_torch_types_in_the_abi():
    text = open(os.path.join(ROOT, "include", "markovmodels_amd.h")).read()
    code = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    assert "torch" not in code.lower() and "at::" not in code and "std::" not in code
    assert 'extern "C"' in text


def test_product_does_not_import_the_oracle():
    """The product package must never route through oracle/ (no CPU fallback)."""
    pkg = os.path.join(ROOT, "markovmodels.jl_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".cpp", ".h")):
                src = open(os.path.join(dirpath, f)).read()
                assert "mm_oracle" not in src and "libmm_oracle" not in src, f
