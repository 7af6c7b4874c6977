"""ctypes binding of libmvs_hip.so (C ABI: include/mvs_hip.h).

The library is looked up next to this file (built in-tree by
``__graft_entry__.build()`` / ``csrc/Makefile``).  There is no fallback: if it
cannot be loaded every compute entry point raises ``RuntimeError``.
"""

from __future__ import annotations

import ctypes as C
import os
import threading

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libmvs_hip.so")

MVS_U8, MVS_U16, MVS_F32 = 0, 1, 2
MVS_MEM_HOST, MVS_MEM_DEVICE = 0, 1
MVS_FUSE_WEIGHTED_AVERAGE, MVS_FUSE_MAX, MVS_FUSE_SIMPLE_AVERAGE = 0, 1, 2
MVS_WEIGHTS_NONE, MVS_WEIGHTS_CONTENT_BASED = 0, 1

DTYPE_CODES = {np.dtype(np.uint8): MVS_U8, np.dtype(np.uint16): MVS_U16, np.dtype(np.float32): MVS_F32}
CODE_DTYPES = {v: k for k, v in DTYPE_CODES.items()}


class mvs_view_t(C.Structure):
    _fields_ = [
        ("data", C.c_void_p),
        ("dtype", C.c_int32),
        ("mem", C.c_int32),
        ("shape", C.c_int64 * 3),
        ("stride", C.c_int64 * 3),
        ("matrix", C.c_double * 9),
        ("offset", C.c_double * 3),
        ("w_matrix", C.c_double * 9),
        ("w_offset", C.c_double * 3),
        ("edt", C.c_float * 125),
        ("reserved", C.c_int32),
        ("index_offset", C.c_int64 * 3),
    ]


class mvs_fuse_opts_t(C.Structure):
    _fields_ = [
        ("ndim", C.c_int32),
        ("order", C.c_int32),
        ("fusion", C.c_int32),
        ("weights", C.c_int32),
        ("out_shape", C.c_int64 * 3),
        ("trim", C.c_int64 * 3),
        ("sigma_1", C.c_float),
        ("sigma_2", C.c_float),
        ("out_dtype", C.c_int32),
        ("out_mem", C.c_int32),
        ("index_origin", C.c_int64 * 3),
    ]


class mvs_pair_job_t(C.Structure):
    _fields_ = [
        ("fixed", mvs_view_t),
        ("moving", mvs_view_t),
        ("out_shape", C.c_int64 * 3),
        ("wait_ticket", C.c_uint64 * 2),
        ("bin", C.c_int32 * 3),
        ("flags", C.c_int32),
    ]


# name -> (restype, argtypes); every symbol include/mvs_hip.h declares
SIGNATURES = {
    "mvs_version": (C.c_char_p, []),
    "mvs_device_count": (C.c_int, []),
    "mvs_init": (C.c_int, [C.c_int]),
    "mvs_shutdown": (None, [C.c_int]),
    "mvs_last_error": (C.c_char_p, [C.c_int]),
    "mvs_set_stream": (C.c_int, [C.c_int, C.c_void_p]),
    "mvs_synchronize": (C.c_int, [C.c_int]),
    "mvs_set_option": (C.c_int, [C.c_int, C.c_char_p, C.c_int64]),
    "mvs_get_counter": (C.c_int, [C.c_int, C.c_char_p, C.c_int32, C.POINTER(C.c_double)]),
    "mvs_memcpy_peer": (C.c_int, [C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_uint64]),
    "mvs_last_kernel_ms": (C.c_double, [C.c_int]),
    "mvs_malloc": (C.c_int, [C.c_int, C.c_uint64, C.POINTER(C.c_void_p)]),
    "mvs_free": (C.c_int, [C.c_int, C.c_void_p]),
    "mvs_mem_info": (C.c_int, [C.c_int, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]),
    "mvs_memcpy_h2d": (C.c_int, [C.c_int, C.c_void_p, C.c_void_p, C.c_uint64]),
    "mvs_memcpy_d2h": (C.c_int, [C.c_int, C.c_void_p, C.c_void_p, C.c_uint64]),
    "mvs_upload_tile": (C.c_int, [C.c_int, C.c_void_p, C.c_int32, C.POINTER(C.c_int64), C.POINTER(C.c_void_p)]),
    "mvs_memset": (C.c_int, [C.c_int, C.c_void_p, C.c_int32, C.c_uint64]),
    "mvs_copy_into": (C.c_int, [C.c_int, C.c_void_p, C.c_int32, C.POINTER(C.c_int64), C.c_void_p, C.POINTER(C.c_int64),
                                C.POINTER(C.c_int64)]),
    "mvs_fuse_chunk": (C.c_int, [C.c_int, C.POINTER(mvs_view_t), C.c_int32, C.POINTER(mvs_fuse_opts_t), C.c_void_p]),
    "mvs_resample": (C.c_int, [C.c_int, C.POINTER(mvs_view_t), C.POINTER(C.c_int64), C.c_int32, C.c_float, C.c_void_p, C.c_int32]),
    "mvs_blend_weights": (C.c_int, [C.c_int, C.POINTER(mvs_view_t), C.c_int32, C.POINTER(C.c_int64), C.c_void_p, C.c_int32]),
    "mvs_phasecorr": (
        C.c_int,
        [C.c_int, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.POINTER(C.c_int64), C.c_int32, C.c_int32,
         C.POINTER(C.c_double), C.POINTER(C.c_int64), C.POINTER(C.c_float)],
    ),
    "mvs_fft_c2c": (C.c_int, [C.c_int, C.c_void_p, C.c_int32, C.c_int32, C.POINTER(C.c_int64), C.c_int32]),
    "mvs_phasecorr_multi": (
        C.c_int,
        [C.c_int, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.POINTER(C.c_int64), C.POINTER(C.c_int32), C.c_int32,
         C.c_int32, C.POINTER(C.c_double), C.POINTER(C.c_int64), C.POINTER(C.c_float)],
    ),
    "mvs_rescale_intensity": (
        C.c_int,
        [C.c_int, C.c_void_p, C.c_int32, C.c_int64, C.c_void_p, C.c_int32, C.POINTER(C.c_float), C.POINTER(C.c_float),
         C.POINTER(C.c_int64)],
    ),
    "mvs_bin_mean": (
        C.c_int,
        [C.c_int, C.c_void_p, C.c_int32, C.c_int32, C.POINTER(C.c_int64), C.POINTER(C.c_int64), C.POINTER(C.c_int64),
         C.c_void_p, C.c_int32],
    ),
    "mvs_bin_mean_async": (
        C.c_int,
        [C.c_int, C.c_void_p, C.c_int32, C.POINTER(C.c_int64), C.POINTER(C.c_int64), C.POINTER(C.c_int64), C.c_void_p],
    ),
    "mvs_bin_mean_batch_async": (
        C.c_int,
        [C.c_int, C.c_int32, C.POINTER(C.c_void_p), C.c_int32, C.POINTER(C.c_int64), C.POINTER(C.c_int64), C.POINTER(C.c_int64),
         C.POINTER(C.c_void_p)],
    ),
    "mvs_event_record": (C.c_int, [C.c_int, C.POINTER(C.c_uint64)]),
    "mvs_event_wait": (C.c_int, [C.c_int, C.c_uint64]),
    "mvs_host_alloc": (C.c_int, [C.c_uint64, C.POINTER(C.c_void_p)]),
    "mvs_host_free": (C.c_int, [C.c_void_p]),
    "mvs_copy_async": (C.c_int, [C.c_int, C.c_void_p, C.c_void_p, C.c_uint64, C.c_int32, C.c_uint64, C.POINTER(C.c_uint64)]),
    "mvs_mark": (C.c_int, [C.c_int, C.POINTER(C.c_uint64)]),
    "mvs_copy_box": (C.c_int, [C.c_int, C.c_void_p, C.POINTER(C.c_int64), C.c_void_p, C.POINTER(C.c_int64), C.POINTER(C.c_int64)]),
    "mvs_ticket_sync": (C.c_int, [C.c_uint64]),
    "mvs_ticket_elapsed_ms": (C.c_int, [C.c_uint64, C.c_uint64, C.POINTER(C.c_double)]),
    "mvs_plan_pairs": (
        C.c_int,
        [C.c_int32, C.c_int32, C.POINTER(C.c_void_p), C.POINTER(C.c_int64), C.POINTER(C.c_double), C.POINTER(C.c_double), C.c_int32,
         C.POINTER(C.c_int32), C.POINTER(C.c_int64), C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_int64),
         C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_int32)],
    ),
    "mvs_register_pairs": (
        C.c_int,
        [C.c_int, C.c_int32, C.POINTER(mvs_pair_job_t), C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.POINTER(C.c_double),
         C.POINTER(C.c_double), C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.POINTER(C.c_int32)],
    ),
    "mvs_register_views": (
        C.c_int,
        [C.c_int, C.POINTER(mvs_view_t), C.POINTER(mvs_view_t), C.c_int32, C.POINTER(C.c_int64), C.c_int32, C.c_int32, C.c_int32,
         C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_int32), C.POINTER(C.c_int32)],
    ),
    "mvs_beads_translation_sweeps": (
        C.c_int,
        [C.c_int32, C.c_int32, C.c_int32, C.POINTER(C.c_int32), C.POINTER(C.c_double), C.POINTER(C.c_double), C.c_int32,
         C.POINTER(C.c_int32), C.c_int32, C.c_int32, C.c_double, C.POINTER(C.c_double), C.POINTER(C.c_double),
         C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_int32)],
    ),
    "mvs_fuse_plan": (
        C.c_int,
        [C.c_int32, C.c_int32, C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_int64), C.POINTER(C.c_double),
         C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_int64), C.POINTER(C.c_int64),
         C.POINTER(C.c_int64), C.c_int32, C.c_void_p, C.c_int64, C.POINTER(C.c_int64), C.POINTER(C.c_int32)],
    ),
    "mvs_view_graph_prune": (
        C.c_int,
        [C.c_int32, C.c_int32, C.POINTER(C.c_double), C.POINTER(C.c_double), C.c_int64, C.POINTER(C.c_int32), C.c_int32, C.c_int32,
         C.POINTER(C.c_int32), C.POINTER(C.c_double), C.POINTER(C.c_int32), C.POINTER(C.c_int32)],
    ),
    "mvs_resolve_translations": (
        C.c_int,
        [C.c_int32, C.c_int32, C.c_int32, C.POINTER(C.c_int32), C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_double),
         C.POINTER(C.c_double), C.POINTER(C.c_double), C.c_int32, C.c_int32, C.c_double, C.c_double, C.POINTER(C.c_double),
         C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_int32), C.POINTER(C.c_int32)],
    ),
    "mvs_edge_betweenness": (
        C.c_int,
        [C.c_int32, C.c_int32, C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.POINTER(C.c_double)],
    ),
    "mvs_register_crops": (
        C.c_int,
        [C.c_int, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.POINTER(C.c_int64), C.c_int32, C.c_int32, C.c_int32,
         C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_int32), C.POINTER(C.c_int32)],
    ),
    "mvs_score_candidates": (
        C.c_int,
        [C.c_int, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.POINTER(C.c_int64), C.POINTER(C.c_double),
         C.c_int32, C.c_int32, C.c_double, C.c_double, C.c_int32, C.POINTER(C.c_double), C.POINTER(C.c_double),
         C.POINTER(C.c_int32)],
    ),
}

_lib = None
_lock = threading.Lock()


def load():
    """Load libmvs_hip.so (once). Raises RuntimeError if it is missing/unloadable."""
    global _lib
    if _lib is not None:
        return _lib
    with _lock:
        if _lib is not None:
            return _lib
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"libmvs_hip.so not found at {LIB_PATH}: build it with "
                "`python -c 'import __graft_entry__ as g; g.build()'` (needs hipcc). "
                "multiview_stitcher_amd has no CPU fallback."
            )
        try:
            lib = C.CDLL(LIB_PATH)
        except OSError as e:  # pragma: no cover
            raise RuntimeError(f"cannot load {LIB_PATH}: {e}") from e
        for name, (restype, argtypes) in SIGNATURES.items():
            fn = getattr(lib, name)
            fn.restype = restype
            fn.argtypes = argtypes
        _lib = lib
    return _lib


ERR_OUT_OF_MEMORY = -5          # MVS_ERR_OUT_OF_MEMORY (include/mvs_hip.h): a HIP call failed with hipErrorOutOfMemory


class MvsError(RuntimeError):
    """A library call returned a non-zero code; ``code`` is the MVS_ERR_* value of include/mvs_hip.h."""

    def __init__(self, message, code):
        super().__init__(message)
        self.code = int(code)


class DeviceMemoryError(MvsError):
    """MVS_ERR_OUT_OF_MEMORY: the device (or pinned host) allocation of a call did not fit."""


def check(rc, device=0, what="mvs call"):
    if rc != 0:
        msg = load().mvs_last_error(int(device))
        cls = DeviceMemoryError if rc == ERR_OUT_OF_MEMORY else MvsError
        raise cls(f"{what} failed (code {rc}): {msg.decode() if msg else ''}", rc)


def init(device=0):
    lib = load()
    check(lib.mvs_init(int(device)), device, "mvs_init")
    return lib


def device_count():
    return int(load().mvs_device_count())


def i64x3(vals):
    return (C.c_int64 * 3)(*[int(v) for v in vals])


def synchronize(device=0):
    """Block until all work queued on the context's stream has finished (mvs_synchronize)."""
    check(init(device).mvs_synchronize(int(device)), device, "mvs_synchronize")


def mem_info(device=0):
    """(free, total) bytes of the GPU behind ``device`` (free includes this context's allocation cache)."""
    f, t = C.c_uint64(0), C.c_uint64(0)
    check(init(device).mvs_mem_info(int(device), C.byref(f), C.byref(t)), device, "mvs_mem_info")
    return int(f.value), int(t.value)


def last_kernel_ms(device=0):
    return float(load().mvs_last_kernel_ms(int(device)))


class DeviceBuffer:
    """A device allocation owned by the library (mvs_malloc / mvs_free)."""

    def __init__(self, device, nbytes):
        lib = init(device)
        self.device = int(device)
        self.nbytes = int(nbytes)
        p = C.c_void_p()
        check(lib.mvs_malloc(self.device, self.nbytes, C.byref(p)), device, "mvs_malloc")
        self.ptr = p.value
        self.version = 0           # bumped by every write into the allocation: peer copies on other GPUs are stamped with it

    def mark_written(self):
        self.version += 1

    def upload(self, host: np.ndarray):
        host = np.ascontiguousarray(host)
        assert host.nbytes <= self.nbytes
        check(load().mvs_memcpy_h2d(self.device, self.ptr, host.ctypes.data, host.nbytes), self.device, "h2d")
        self.mark_written()
        return self

    def download(self, shape, dtype):
        out = np.empty(shape, dtype=dtype)
        assert out.nbytes <= self.nbytes
        check(load().mvs_memcpy_d2h(self.device, out.ctypes.data, self.ptr, out.nbytes), self.device, "d2h")
        return out

    def free(self):
        if self.ptr:
            load().mvs_free(self.device, self.ptr)
            self.ptr = None

    def __del__(self):  # pragma: no cover
        try:
            self.free()
        except Exception:
            pass


def get_counter(key, device=0, reset=False):
    """mvs_get_counter: a measurement counter of context ``device`` (see include/mvs_hip.h)."""
    v = C.c_double()
    check(init(device).mvs_get_counter(int(device), key.encode(), 1 if reset else 0, C.byref(v)), device, "mvs_get_counter")
    return float(v.value)


def set_option(key, value, device=0):
    check(init(device).mvs_set_option(int(device), key.encode(), int(value)), device, "mvs_set_option")
