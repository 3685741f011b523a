"""C-ABI checks that need no GPU: the library loads, exports every symbol the header declares,
fails loudly without a device, and the host-only tree utility works."""
import ctypes as C
import os
import re

import pytest

from pepr_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _no_gpu():
    import torch
    return not torch.cuda.is_available()


def test_header_symbols_exported():
    hdr = open(os.path.join(ROOT, "include", "peprml.h")).read()
    declared = set(re.findall(r"\b(pml_[a-z_0-9]+)\s*\(", hdr))
    assert declared == set(_lib.SYMBOLS), declared ^ set(_lib.SYMBOLS)
    L = _lib.load()
    for s in _lib.SYMBOLS:
        assert hasattr(L, s), s


def test_struct_layouts_match_header():
    assert C.sizeof(_lib.Alignment) == 24 and C.sizeof(_lib.Model) == 24
    assert C.sizeof(_lib.Result) == 56 and C.sizeof(_lib.SearchOpts) == 56


def test_strerror_and_version():
    L = _lib.load()
    assert L.pml_strerror(0) == b"ok" and b"device" in L.pml_strerror(-3)
    assert b"gfx950" in L.pml_version()


def test_create_fails_loudly_without_gpu():
    if not _no_gpu():
        pytest.skip("GPU present")
    from pepr_amd import engine
    with pytest.raises(engine.PmlError) as ei:
        engine.Context(0)
    assert ei.value.code == -3          # PML_ENODEVICE: no CPU fallback exists


def test_rf_distance_host_only():
    from pepr_amd import engine
    a = "((a:1,b:1):1,(c:1,d:1):1,(e:1,f:1):1);"
    b = "((a:1,c:1)90:1,(b:1,d:1):1[7],(e:1,f:1):1)"
    c = "(a:1,(b:1,(c:1,(d:1,(e:1,f:1):1):1):1):1);"
    assert engine.rf_distance(a, a) == 0 and engine.rf_distance(a, b) == 2 and engine.rf_distance(a, c) == 1
    with pytest.raises(engine.PmlError):
        engine.rf_distance(a, "(a:1,(b:1,c:1):1,(d:1,x:1):1);")


def test_null_arguments_rejected():
    L = _lib.load()
    assert L.pml_create(None, None) == -1
    assert L.pml_batch_score(None, None) == -1
    assert L.pml_kernel_stats(None, 0, None, None, None) == -1
