"""CPU tests: the C-ABI library loads and exports every symbol include/mvs_hip.h declares."""
import ctypes
import os
import re

from multiview_stitcher_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "mvs_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(mvs_[a-z0-9_]+)\s*\(", text)))


def test_header_symbols_exported_and_bound():
    lib = _lib.load()
    syms = declared_symbols()
    assert len(syms) >= 15
    for s in syms:
        assert hasattr(lib, s), f"{s} declared in mvs_hip.h but not exported"
        assert s in _lib.SIGNATURES, f"{s} has no ctypes signature"
    assert set(_lib.SIGNATURES) == set(syms)


def test_struct_sizes_match_header_layout():
    # mvs_view_t: ptr + 2*i32 + 3*i64 + 3*i64 + 9+3+9+3 doubles + 125 floats + i32 + 3*i64 (index_offset)
    assert ctypes.sizeof(_lib.mvs_view_t) == 8 + 8 + 24 + 24 + 24 * 8 + 125 * 4 + 4 + 24
    assert ctypes.sizeof(_lib.mvs_fuse_opts_t) == 16 + 24 + 24 + 8 + 8 + 24


def test_version_and_no_device_error_path():
    lib = _lib.load()
    assert b"gfx950" in lib.mvs_version()
    if lib.mvs_device_count() == 0:
        assert lib.mvs_init(0) != 0          # fails loudly, no CPU fallback
        assert lib.mvs_last_error(0)
