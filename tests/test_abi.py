"""The C-ABI library loads on a CPU-only box and exports every symbol include/grove_hip.h declares;
the ctypes mirror of every params struct has the size the library was compiled with."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    from grove_amd import _lib
    if not os.path.exists(_lib.LIB_PATH):
        import __graft_entry__
        __graft_entry__.build()
    return _lib.lib()


def header_symbols():
    text = open(os.path.join(ROOT, "include", "grove_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\bint\s+(grove_[a-z0-9_]+)\s*\(", text)))


def test_exports_every_declared_symbol(lib):
    from grove_amd import _lib
    declared = header_symbols()
    assert len(declared) >= 30
    for name in declared:
        assert hasattr(lib, name), f"libgrove_hip.so does not export {name}"
    assert sorted(_lib.SYMBOLS) == declared, set(_lib.SYMBOLS) ^ set(declared)


def test_struct_sizes_match(lib):
    from grove_amd import _lib
    for name, st in _lib.STRUCTS.items():
        assert lib.grove_sizeof(name.encode()) == ctypes.sizeof(st), name


def test_errors_are_loud(lib):
    from grove_amd import _lib
    p = _lib.GemmParams()
    p.M, p.N, p.K = 4, 4, 7  # K not a multiple of 32 -> rejected before any launch
    rc = lib.grove_gemm_bf16(ctypes.byref(p), None)
    assert rc == -1 and "multiple of 32" in _lib.last_error()
    with pytest.raises(RuntimeError):
        _lib.check(rc, "grove_gemm_bf16")


def test_ops_refuse_cpu_tensors():
    import torch
    from grove_amd import ops
    with pytest.raises(RuntimeError):
        ops.linear(torch.zeros(32, 32, dtype=torch.bfloat16), torch.zeros(32, 32, dtype=torch.bfloat16))
