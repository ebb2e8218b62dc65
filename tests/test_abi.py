"""The C-ABI library loads and exports every entry point include/mcensus.h declares (no compute calls: this
runs without a GPU).  mc_open must fail loudly - not fall back - when no HIP device is visible."""
import ctypes as C
import os
import re

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    import __graft_entry__ as g
    if not os.path.exists(g.LIB):
        g.build()
    return C.CDLL(g.LIB)


def declared_functions():
    text = open(os.path.join(REPO, "include", "mcensus.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(mc_[a-z0-9_]+)\s*\(", text)))


def test_every_declared_symbol_is_exported(lib):
    names = declared_functions()
    assert len(names) >= 14
    for n in names:
        assert hasattr(lib, n), n


def test_python_binding_lists_the_same_symbols():
    from microbecensus_amd import _native
    assert sorted(_native.EXPORTED_SYMBOLS) == declared_functions()


def test_no_cpu_fallback(lib):
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is visible")
    lib.mc_open.restype = C.c_void_p
    lib.mc_last_error.restype = C.c_char_p
    names = (C.c_char_p * 1)(b"m0")
    seqs = (C.c_char_p * 1)(b"MKTAYIAKQRQISFVKSHFSRQ")
    fam = (C.c_int32 * 1)(0)
    h = lib.mc_open(names, seqs, 1, fam, 1, 0)
    assert not h
    assert b"no HIP device" in lib.mc_last_error()
