"""The C-ABI library loads without a GPU and exports every symbol include/cellector_ffi.h declares."""
import ctypes
import os
import re

from cellector_amd import ffi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    text = open(os.path.join(ROOT, "include", "cellector_ffi.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(cellector_[a-z0-9_]+)\s*\(", text)))


def test_every_declared_symbol_is_exported_and_bound(hip_lib_path):
    lib = ctypes.CDLL(hip_lib_path)
    names = _declared()
    assert len(names) >= 30
    for n in names:
        assert hasattr(lib, n), f"{n} declared in cellector_ffi.h but not exported"
        assert n in ffi.SIGNATURES, f"{n} has no ctypes signature in cellector_amd/ffi.py"
    assert sorted(ffi.SIGNATURES) == names


def test_library_loads_and_reports_version(hip_lib_path):
    lib = ffi.load_library()
    assert b"gfx950" in lib.cellector_version()
    assert lib.cellector_last_error(None) == b"null ctx"


def test_status_struct_layouts():
    assert ctypes.sizeof(ffi.Dims) == 48
    assert ctypes.sizeof(ffi.IterSummary) == 72
    assert ffi.IterSummary.median.offset == 40 and ffi.IterSummary.n_near_threshold.offset == 64
