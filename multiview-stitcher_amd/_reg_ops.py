"""Thin array-level wrappers of the registration entry points of libmvs_hip.so."""

from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib
from .device import DeviceArray, is_device_array
from .transformation import shape3


def _ptr_mem(a):
    if is_device_array(a):
        if not a.is_contiguous() or a.dtype != np.float32:
            raise TypeError("registration kernels need contiguous float32 device arrays")
        return a.ptr, _lib.MVS_MEM_DEVICE, a
    a = np.ascontiguousarray(a, dtype=np.float32)
    return a.ctypes.data, _lib.MVS_MEM_HOST, a


def rescale_intensity(im, device=0, out_on_device=False):
    """skimage.exposure.rescale_intensity(im, in_range=(nanmin, nanmax), out_range=(0, 1)) on the GPU.
    Returns (rescaled float32, nanmin, nanmax, n_valid)."""
    lib = _lib.init(device)
    ptr, mem, keep = _ptr_mem(im)
    n = int(np.prod(im.shape))
    mn, mx, nv = C.c_float(), C.c_float(), C.c_int64()
    if out_on_device:
        out = DeviceArray.empty(im.shape, np.float32, device)
        optr, omem = out.ptr, _lib.MVS_MEM_DEVICE
    else:
        out = np.empty(im.shape, dtype=np.float32)
        optr, omem = out.ctypes.data, _lib.MVS_MEM_HOST
    rc = lib.mvs_rescale_intensity(device, ptr, mem, n, optr, omem, C.byref(mn), C.byref(mx), C.byref(nv))
    _lib.check(rc, device, "mvs_rescale_intensity")
    return out, float(mn.value), float(mx.value), int(nv.value)


def phase_cross_correlation(reference_image, moving_image, upsample_factor=1, normalization="phase", device=0,
                            return_debug=False):
    """skimage.registration.phase_cross_correlation(ref, mov, upsample_factor=, normalization=,
    disambiguate=False)[0] on the GPU (NaN-free float32 inputs of equal shape)."""
    lib = _lib.init(device)
    if tuple(reference_image.shape) != tuple(moving_image.shape):
        raise ValueError("images must be same shape")
    if normalization not in ("phase", None):
        raise ValueError("normalization must be either phase or None")
    shape = tuple(int(s) for s in reference_image.shape)
    ndim = len(shape)
    p0, m0, k0 = _ptr_mem(reference_image)
    p1, m1, k1 = _ptr_mem(moving_image)
    if m0 != m1:
        raise TypeError("both images must live on the same side (host or device)")
    shift = (C.c_double * 3)()
    peak = (C.c_int64 * 3)()
    pabs = C.c_float()
    rc = lib.mvs_phasecorr(device, p0, p1, m0, ndim, _lib.i64x3(shape3(shape)), 1 if normalization == "phase" else 0,
                           int(upsample_factor), shift, peak, C.byref(pabs))
    _lib.check(rc, device, "mvs_phasecorr")
    s = np.array(list(shift)[3 - ndim:], dtype=np.float32)
    if return_debug:
        return s, {"peak_index": np.array(list(peak)[3 - ndim:]), "peak_abs": float(pabs.value)}
    return s


def fftn(a, inverse=False, device=0):
    """numpy.fft.fftn / ifftn (the inverse WITHOUT its 1/N) of a 2D / 3D complex64 host array on the GPU (mvs_fft_c2c)."""
    lib = _lib.init(device)
    out = np.ascontiguousarray(a, dtype=np.complex64).copy()
    rc = lib.mvs_fft_c2c(device, out.ctypes.data, _lib.MVS_MEM_HOST, out.ndim, _lib.i64x3(shape3(out.shape)), 1 if inverse else 0)
    _lib.check(rc, device, "mvs_fft_c2c")
    return out


def phase_cross_correlation_multi(reference_image, moving_image, upsample_factor=1, normalizations=("phase", None), device=0):
    """``phase_cross_correlation`` for several normalisations of one image pair; the forward transforms are
    shared (mvs_phasecorr_multi).  Returns a list of (shift, debug) like ``phase_cross_correlation(return_debug=True)``."""
    lib = _lib.init(device)
    if tuple(reference_image.shape) != tuple(moving_image.shape):
        raise ValueError("images must be same shape")
    for normalization in normalizations:
        if normalization not in ("phase", None):
            raise ValueError("normalization must be either phase or None")
    shape = tuple(int(s) for s in reference_image.shape)
    ndim = len(shape)
    p0, m0, k0 = _ptr_mem(reference_image)
    p1, m1, k1 = _ptr_mem(moving_image)
    if m0 != m1:
        raise TypeError("both images must live on the same side (host or device)")
    nn = len(normalizations)
    norms = (C.c_int32 * nn)(*[1 if v == "phase" else 0 for v in normalizations])
    shift = (C.c_double * (3 * nn))()
    peak = (C.c_int64 * (3 * nn))()
    pabs = (C.c_float * nn)()
    rc = lib.mvs_phasecorr_multi(device, p0, p1, m0, ndim, _lib.i64x3(shape3(shape)), norms, nn, int(upsample_factor),
                                 shift, peak, pabs)
    _lib.check(rc, device, "mvs_phasecorr_multi")
    out = []
    for i in range(nn):
        s = np.array(list(shift)[3 * i + 3 - ndim:3 * i + 3], dtype=np.float32)
        out.append((s, {"peak_index": np.array(list(peak)[3 * i + 3 - ndim:3 * i + 3]), "peak_abs": float(pabs[i])}))
    return out


def register_crops(im0, im1, upsample_factor, region_mode=None, constant_check=False, device=0):
    """registration.phase_correlation_registration in one library call (mvs_register_crops).  Returns
    (t (ndim,) float64, quality, status, n_candidates); status as documented in include/mvs_hip.h."""
    lib = _lib.init(device)
    shape = tuple(int(s) for s in im0.shape)
    ndim = len(shape)
    p0, m0, k0 = _ptr_mem(im0)
    p1, m1, k1 = _ptr_mem(im1)
    if m0 != m1:
        raise TypeError("both images must live on the same side (host or device)")
    t = (C.c_double * 3)()
    q = C.c_double()
    status = C.c_int32()
    ncand = C.c_int32()
    mode = -1 if region_mode is None else {"union": 0, "intersection": 1}[region_mode]
    rc = lib.mvs_register_crops(device, p0, p1, m0, ndim, _lib.i64x3(shape3(shape)), int(upsample_factor), mode, int(bool(constant_check)),
                                t, C.byref(q), C.byref(status), C.byref(ncand))
    _lib.check(rc, device, "mvs_register_crops")
    return np.array(list(t)[3 - ndim:], dtype=np.float64), float(q.value), int(status.value), int(ncand.value)


def register_views(data0, matrix0, offset0, data1, matrix1, offset1, out_shape, upsample_factor, region_mode=None, constant_check=False,
                   device=0):
    """Resample both overlap crops and register them in one library call (mvs_register_views).  ``data*``: DeviceArray slabs
    (strided windows allowed), ``matrix*`` (diagonal given as a list) / ``offset*``: the pixel affines of
    transformation.get_pixel_affine, ``out_shape``: the fixed view's overlap grid.  Returns like ``register_crops``."""
    from .transformation import shape3

    lib = _lib.init(device)
    ndim = len(out_shape)
    views = (_lib.mvs_view_t * 2)()
    for v, data, mdiag, off in ((views[0], data0, matrix0, offset0), (views[1], data1, matrix1, offset1)):
        if not is_device_array(data) or data.dtype not in _lib.DTYPE_CODES:
            raise TypeError("register_views needs DeviceArray slabs of a supported dtype")
        k = 3 - ndim
        m = [1.0, 0.0, 0.0, 0.0, 1.0, 0.0, 0.0, 0.0, 1.0]
        o = [0.0, 0.0, 0.0]
        for a in range(ndim):
            m[(k + a) * 4] = float(mdiag[a])
            o[k + a] = float(off[a])
        shp, st = [int(x) for x in data.shape], [int(x) for x in data.strides]
        if ndim == 2:
            shp, st = [1] + shp, [st[0] * shp[0]] + st
        v.data = data.ptr
        v.dtype = _lib.DTYPE_CODES[data.dtype]
        v.mem = _lib.MVS_MEM_DEVICE
        v.shape[:] = shp
        v.stride[:] = st
        v.matrix[:] = m
        v.offset[:] = o
    t = (C.c_double * 3)()
    q = C.c_double()
    status = C.c_int32()
    ncand = C.c_int32()
    mode = -1 if region_mode is None else {"union": 0, "intersection": 1}[region_mode]
    rc = lib.mvs_register_views(device, C.byref(views[0]), C.byref(views[1]), ndim, _lib.i64x3(shape3(out_shape)), int(upsample_factor), mode,
                                int(bool(constant_check)), t, C.byref(q), C.byref(status), C.byref(ncand))
    _lib.check(rc, device, "mvs_register_views")
    return np.array(list(t)[3 - ndim:], dtype=np.float64), float(q.value), int(status.value), int(ncand.value)


def score_candidates(im0, im1, t_candidates, region_mode, data_range, im1_min, device=0, quality_for_all=True):
    """The candidate loop of registration.py:493-556 on the GPU.  im0 / im1: rescaled float32 images
    (NaN = outside).  Returns (ssim, spearman, code) arrays; code 1 = (-1,-1) appended, 2 = `continue`.
    ``quality_for_all=False`` evaluates the Spearman coefficient only for the best-SSIM candidate(s) (NaN
    elsewhere) -- the only value the reference's result uses."""
    lib = _lib.init(device)
    shape = tuple(int(s) for s in im0.shape)
    ndim = len(shape)
    p0, m0, k0 = _ptr_mem(im0)
    p1, m1, k1 = _ptr_mem(im1)
    if m0 != m1:
        raise TypeError("both images must live on the same side (host or device)")
    t_all = np.asarray(t_candidates, dtype=np.float64).reshape(-1, ndim)
    # both phase-correlation variants usually agree, so the enumeration holds exact duplicates; scoring is a
    # pure function of t: evaluate each distinct vector once and scatter (list order / nanargmax unchanged)
    t_uniq, inverse = np.unique(t_all, axis=0, return_inverse=True)
    t = np.ascontiguousarray(t_uniq)
    inverse = np.asarray(inverse).reshape(-1)
    n = t.shape[0]
    ssim = np.empty(n, dtype=np.float64)
    spear = np.empty(n, dtype=np.float64)
    code = np.empty(n, dtype=np.int32)
    rc = lib.mvs_score_candidates(
        device, p0, p1, m0, ndim, _lib.i64x3(shape3(shape)), t.ctypes.data_as(C.POINTER(C.c_double)), n,
        {"union": 0, "intersection": 1}[region_mode], float(data_range), float(im1_min), int(bool(quality_for_all)),
        ssim.ctypes.data_as(C.POINTER(C.c_double)), spear.ctypes.data_as(C.POINTER(C.c_double)),
        code.ctypes.data_as(C.POINTER(C.c_int32)),
    )
    _lib.check(rc, device, "mvs_score_candidates")
    return ssim[inverse], spear[inverse], code[inverse]


def bin_mean(data, bins, device=0, wait=True, out=None):
    """``coarsen(bins, boundary="trim").mean().astype(dtype)`` (registration.py:1732-1741) on the GPU.
    ``data``: numpy array or (strided) DeviceArray, spatial dims only; ``bins``: per-axis ints.  ``wait=False`` (device
    arrays only) queues the kernel and returns: ``_lib.synchronize(device)`` before the result is used; ``out``: a contiguous
    DeviceArray of the binned shape to write into (one allocation for many tiles)."""
    lib = _lib.init(device)
    nd = data.ndim
    bins = [int(b) for b in bins]
    shape = [int(s) for s in data.shape]
    oshape = [s // b for s, b in zip(shape, bins)]
    dtype = np.dtype(data.dtype)
    if dtype not in _lib.DTYPE_CODES:
        raise TypeError(f"unsupported dtype {dtype}")
    if is_device_array(data):
        ptr, mem, strides = data.ptr, _lib.MVS_MEM_DEVICE, list(data.strides)
        if out is None:
            out = DeviceArray.empty(oshape, dtype, device)
        elif tuple(out.shape) != tuple(oshape) or out.dtype != dtype or not out.is_contiguous():
            raise ValueError("bin_mean: out must be a contiguous DeviceArray of the binned shape and the input dtype")
        optr, omem = out.ptr, _lib.MVS_MEM_DEVICE
    else:
        data = np.ascontiguousarray(data)
        ptr, mem, strides = data.ctypes.data, _lib.MVS_MEM_HOST, [int(np.prod(data.shape[k + 1:])) for k in range(nd)]
        out = np.empty(oshape, dtype=dtype)
        optr, omem = out.ctypes.data, _lib.MVS_MEM_HOST
    s3 = shape3(shape)
    st3 = strides if nd == 3 else [strides[0] * s3[1], strides[0], strides[1]]
    b3 = [1] * (3 - nd) + bins
    if not wait and mem == _lib.MVS_MEM_DEVICE:
        rc = lib.mvs_bin_mean_async(device, ptr, _lib.DTYPE_CODES[dtype], _lib.i64x3(s3), _lib.i64x3(st3), _lib.i64x3(b3), optr)
        _lib.check(rc, device, "mvs_bin_mean_async")
        out.mark_written()
        return out
    rc = lib.mvs_bin_mean(device, ptr, _lib.DTYPE_CODES[dtype], mem, _lib.i64x3(s3), _lib.i64x3(st3), _lib.i64x3(b3), optr, omem)
    _lib.check(rc, device, "mvs_bin_mean")
    if is_device_array(out):
        out.mark_written()
    return out
