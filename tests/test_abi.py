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


def test_pair_job_layout_and_host_only_entry_points_refuse_bad_arguments():
    """mvs_pair_job_t = two views + output grid + tickets + bin (include/mvs_hip.h); the host-only entry points of round 5 return
    error codes (never crash) on malformed input and need no device."""
    import numpy as np

    V = ctypes.sizeof(_lib.mvs_view_t)
    assert ctypes.sizeof(_lib.mvs_pair_job_t) == 2 * V + 24 + 16 + 12 + 4
    assert _lib.mvs_pair_job_t.bin.offset == 2 * V + 40 and _lib.mvs_pair_job_t.wait_ticket.offset == 2 * V + 24
    lib = _lib.load()
    C = ctypes
    ptr = lambda a, ty: a.ctypes.data_as(C.POINTER(ty))
    lo = np.zeros((2, 2))
    hi = np.ones((2, 2))
    edges = np.zeros((4, 2), np.int32)
    ovl = np.zeros(4)
    ne = C.c_int32(0)
    bad = np.array([[0, 5]], np.int32)                    # a view index out of range
    assert lib.mvs_view_graph_prune(2, 2, ptr(lo, C.c_double), ptr(hi, C.c_double), 1, ptr(bad, C.c_int32), 1, 2, ptr(edges, C.c_int32),
                                    ptr(ovl, C.c_double), C.byref(ne), None) < 0
    assert lib.mvs_view_graph_prune(4, 2, ptr(lo, C.c_double), ptr(hi, C.c_double), 0, None, 1, 2, ptr(edges, C.c_int32),
                                    ptr(ovl, C.c_double), C.byref(ne), None) < 0        # ndim 4
    ok = np.array([[0, 1], [1, 0]], np.int32)             # identical boxes overlap: one edge, reported once
    assert lib.mvs_view_graph_prune(2, 2, ptr(lo, C.c_double), ptr(hi, C.c_double), 2, ptr(ok, C.c_int32), 1, 2, ptr(edges, C.c_int32),
                                    ptr(ovl, C.c_double), C.byref(ne), None) == 0
    assert ne.value == 1 and edges[0].tolist() == [0, 1] and ovl[0] == 1.0
    # mvs_resolve_translations: an edge list that is not sorted / a disconnected view -> "not covered" (the caller takes the Python form)
    t, q, blo, bhi, sp = np.zeros((1, 2)), np.ones(1), np.zeros((1, 2)), np.ones((1, 2)), np.ones((3, 2))
    tout, rms, mh, xh, nit = np.zeros((3, 2)), np.zeros(1), np.zeros(10), np.zeros(10), C.c_int32(0)
    e = np.array([[1, 0]], np.int32)
    args = lambda en, nviews: (2, nviews, 1, ptr(en, C.c_int32), ptr(t, C.c_double), ptr(q, C.c_double), ptr(blo, C.c_double), ptr(bhi, C.c_double),
                               ptr(sp, C.c_double), -1, 10, 1e-4, -1.0, ptr(tout, C.c_double), ptr(rms, C.c_double), ptr(mh, C.c_double),
                               ptr(xh, C.c_double), C.byref(nit), None)
    assert lib.mvs_resolve_translations(*args(e, 2)) == -4                 # MVS_ERR_UNSUPPORTED
    e = np.array([[0, 1]], np.int32)
    assert lib.mvs_resolve_translations(*args(e, 3)) == -4                 # view 2 has no pair
    assert lib.mvs_resolve_translations(*args(e, 2)) == 0 and nit.value >= 1
    # mvs_plan_pairs: NULL coordinate arrays / bad pair indices
    st = np.zeros(1, np.int32)
    w, oo, osp, osh, md, of = np.zeros((1, 2, 3, 2), np.int64), np.zeros((1, 3)), np.zeros((1, 3)), np.ones((1, 3), np.int64), np.zeros((1, 2, 3)), np.zeros((1, 2, 3))
    c0 = np.arange(8, dtype=np.float64)
    cptr = (C.c_void_p * 4)(c0.ctypes.data, c0.ctypes.data, c0.ctypes.data, c0.ctypes.data)
    clen = np.array([8, 8, 8, 8], np.int64)
    tr = np.array([[0.0, 0.0], [0.0, 5.0]])
    pr = np.array([[0, 7]], np.int32)
    call = lambda pairs: lib.mvs_plan_pairs(2, 2, cptr, ptr(clen, C.c_int64), ptr(tr, C.c_double), None, 1, ptr(pairs, C.c_int32), ptr(w, C.c_int64),
                                            ptr(oo, C.c_double), ptr(osp, C.c_double), ptr(osh, C.c_int64), ptr(md, C.c_double), ptr(of, C.c_double),
                                            ptr(st, C.c_int32))
    assert call(pr) < 0
    assert call(np.array([[0, 1]], np.int32)) == 0 and st[0] == 0 and osh[0, :2].tolist() == [8, 3]
    tr[1, 1] = 50.0                                        # no overlap: status 1, not an error
    assert call(np.array([[0, 1]], np.int32)) == 0 and st[0] == 1
