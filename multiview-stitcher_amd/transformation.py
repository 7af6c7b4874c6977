"""Affine resampling on the HIP backend (mirror of the reference's transformation.py).

``get_pixel_affine`` restates the host-side parameter derivation of
``transform_sim`` (src/multiview_stitcher/transformation.py:37-83); the
resampling itself (transformation.py:136-139 -> scipy.ndimage.affine_transform)
runs in ``mvs_resample`` (csrc/mvs_fuse.hip).
"""

from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib, param_utils
from . import spatial_image_utils as si_utils


def _as_zyx(value, sdims):
    """dict keyed by dim name (reference style) or sequence -> float64 array in sdims order."""
    if isinstance(value, dict):
        return np.array([value[d] for d in sdims], dtype=np.float64)
    return np.asarray(value, dtype=np.float64)


def get_pixel_affine(p, input_origin, input_spacing, output_origin, output_spacing):
    """(matrix', offset') mapping OUTPUT pixel indices to INPUT pixel coordinates.

    transformation.py:37-83: matrix' = Sy^-1 M Sx; offset' = Sy^-1 (t + (M - I) Ox - (Oy - Ox)),
    both origins expressed relative to the output origin; rounded to 10 decimals;
    offsets within 1e-6 of an integer are snapped to it.
    """
    p = np.asarray(p, dtype=np.float64)
    ndim = p.shape[0] - 1
    M, t = p[:ndim, :ndim], p[:ndim, ndim]
    sx = np.asarray(output_spacing, dtype=np.float64)
    sy = np.asarray(input_spacing, dtype=np.float64)
    Ox = np.asarray(output_origin, dtype=np.float64)
    Oy = np.asarray(input_origin, dtype=np.float64)
    # np.linalg.solve(diag(sy), X) is a row-wise division (LU of a diagonal matrix: one exact division per element), and
    # M @ diag(sx) a column-wise product (the other terms of each dot product are exact zeros): same bits, no LAPACK call
    matrix_prime = (M * sx[None, :]) / sy[:, None]
    local_offset = t + np.dot(M - np.eye(ndim), Ox)
    offset_prime = (local_offset - (Oy - Ox)) / sy
    matrix_prime = np.around(matrix_prime, decimals=10)
    offset_prime = np.around(offset_prime, decimals=10)
    nearest = np.round(offset_prime)
    snap = np.abs(offset_prime - nearest) <= 1e-6      # np.isclose(offset', nearest, rtol=0, atol=1e-6)
    offset_prime[snap] = nearest[snap]
    return matrix_prime, offset_prime


def get_pixel_affines(p_stack, input_origins, input_spacings, output_origin, output_spacing):
    """``get_pixel_affine`` for a stack of views in one pass: (matrices (n, ndim, ndim), offsets (n, ndim)).  The same
    elementwise operations on stacked arrays (same bits); the one reduction, ``np.dot(M - I, Ox)``, is an exact zero for a
    pure translation and is taken per view through ``np.dot`` otherwise."""
    p_stack = np.asarray(p_stack, dtype=np.float64)
    ndim = p_stack.shape[1] - 1
    M, t = p_stack[:, :ndim, :ndim], p_stack[:, :ndim, ndim]
    sx = np.asarray(output_spacing, dtype=np.float64)
    sy = np.asarray(input_spacings, dtype=np.float64)
    Ox = np.asarray(output_origin, dtype=np.float64)
    Oy = np.asarray(input_origins, dtype=np.float64)
    matrix_prime = (M * sx[None, None, :]) / sy[:, :, None]
    lin = M - np.eye(ndim)[None]
    shift = np.zeros_like(t)
    for i in np.nonzero(np.any(lin != 0, axis=(1, 2)))[0]:
        shift[i] = np.dot(lin[i], Ox)
    offset_prime = ((t + shift) - (Oy - Ox[None, :])) / sy
    matrix_prime = np.around(matrix_prime, decimals=10)
    offset_prime = np.around(offset_prime, decimals=10)
    nearest = np.round(offset_prime)
    snap = np.abs(offset_prime - nearest) <= 1e-6
    offset_prime[snap] = nearest[snap]
    return matrix_prime, offset_prime


def embed3_stack(matrices, offsets):
    """``embed3`` for stacks: (n, 9) row-major matrices and (n, 3) offsets."""
    n, ndim = offsets.shape
    m3 = np.tile(np.eye(3), (n, 1, 1))
    o3 = np.zeros((n, 3))
    k = 3 - ndim
    m3[:, k:, k:] = matrices
    o3[:, k:] = offsets
    return m3.reshape(n, 9), o3


def embed3(matrix, offset):
    """Embed an ndim-D pixel affine into the 3D (z,y,x) form the C ABI takes (2D: z -> z)."""
    ndim = len(offset)
    m3 = np.eye(3)
    o3 = np.zeros(3)
    k = 3 - ndim
    m3[k:, k:] = matrix
    o3[k:] = offset
    return m3, o3


def shape3(shape):
    shape = [int(s) for s in shape]
    return [1] * (3 - len(shape)) + shape


def fill_view_geometry(view, data_ptr, dtype_code, mem, shape, strides_elems, matrix, offset):
    """Fill the data/geometry half of an ``mvs_view_t``."""
    m3, o3 = embed3(matrix, offset)
    s3 = shape3(shape)
    st3 = [int(s) for s in strides_elems]
    if len(st3) == 2:  # 2D slab: a single z plane
        st3 = [st3[0] * s3[1], st3[0], st3[1]]
    view.data = data_ptr
    view.dtype = dtype_code
    view.mem = mem
    view.shape[:] = s3
    view.stride[:] = st3
    view.offset[:] = o3.tolist()
    view.matrix[:] = m3.reshape(-1).tolist()


def check_interpolation_order(order, name):
    """The reference forwards ``order`` to scipy.ndimage.affine_transform unchanged (transformation.py:85-94,
    fusion/_core.py:797, 1627); the HIP resampler has nearest (0) and bi / trilinear (1) taps only.  Higher spline orders
    are refused HERE, at call time and under the reference's parameter name, not from inside a chunk."""
    if isinstance(order, bool) or not isinstance(order, (int, np.integer)) or int(order) not in (0, 1):
        raise NotImplementedError(
            f"{name}={order!r}: the HIP backend resamples with order 0 (nearest) or 1 (linear) only; spline orders 2-5 of "
            "scipy.ndimage.affine_transform (prefiltered B-splines) are not implemented")


def resample_array(data, matrix, offset, output_shape, order=1, cval=0.0, device=0, out_on_device=None):
    """scipy.ndimage.affine_transform(data, matrix, offset, output_shape, order, 'constant', cval)
    for order 0|1 on the GPU; float32 result (transformation.py:136-139).  ``data`` may be a numpy
    array or a (possibly strided) DeviceArray; the result stays on the device in the latter case."""
    from .device import DeviceArray, is_device_array

    lib = _lib.init(device)
    view = _lib.mvs_view_t()
    on_dev = is_device_array(data)
    if out_on_device is None:
        out_on_device = on_dev
    if on_dev:
        if data.dtype not in _lib.DTYPE_CODES:
            raise TypeError(f"unsupported dtype {data.dtype}")
        fill_view_geometry(view, data.ptr, _lib.DTYPE_CODES[data.dtype], _lib.MVS_MEM_DEVICE, data.shape, data.strides, matrix, offset)
        keep = data
    else:
        data = np.ascontiguousarray(data)
        if data.dtype not in _lib.DTYPE_CODES:
            data = data.astype(np.float32)
        strides = [int(np.prod(data.shape[k + 1:])) for k in range(data.ndim)]   # C order; size-1 axes have arbitrary numpy strides
        fill_view_geometry(view, data.ctypes.data, _lib.DTYPE_CODES[data.dtype], _lib.MVS_MEM_HOST, data.shape, strides, matrix, offset)
        keep = data
    oshape = tuple(int(s) for s in output_shape)
    if out_on_device:
        out = DeviceArray.empty(oshape, np.float32, device)
        optr, omem = out.ptr, _lib.MVS_MEM_DEVICE
    else:
        out = np.empty(oshape, dtype=np.float32)
        optr, omem = out.ctypes.data, _lib.MVS_MEM_HOST
    rc = lib.mvs_resample(device, C.byref(view), _lib.i64x3(shape3(output_shape)), int(order), float(cval), optr, omem)
    _lib.check(rc, device, "mvs_resample")
    del keep
    return out


def transform_sim(
    sim,
    p=None,
    output_stack_properties=None,
    keep_transform_keys=False,
    input_spacing=None,
    device=0,
    allow_noop=True,
    **affine_transform_kwargs,
):
    """transformation.transform_sim (transformation.py:15-148) on the HIP backend.

    Same arguments and result conventions; ``order`` 0|1, ``cval`` and
    ``mode="constant"`` are honoured, the output is float32 unless the no-op
    shortcut (transformation.py:102-119) returns the input data unchanged."""
    sdims = si_utils.get_spatial_dims_from_sim(sim)
    ndim = len(sdims)
    if p is None:
        p = param_utils.identity_transform(ndim)
    if input_spacing is None:
        input_spacing = si_utils.get_spacing_from_sim(sim)
    matrix_prime, offset_prime = get_pixel_affine(
        p,
        si_utils.get_origin_from_sim(sim, asarray=True),
        _as_zyx(input_spacing, sdims),
        _as_zyx(output_stack_properties["origin"], sdims),
        _as_zyx(output_stack_properties["spacing"], sdims),
    )
    kwargs = {"mode": "constant", "cval": 0.0, "order": 1} | affine_transform_kwargs
    if kwargs["mode"] != "constant":
        raise NotImplementedError("HIP resampler implements mode='constant' only")
    check_interpolation_order(kwargs["order"], "order")
    out_shape = tuple(int(output_stack_properties["shape"][d]) for d in sdims) if isinstance(
        output_stack_properties["shape"], dict) else tuple(int(s) for s in output_stack_properties["shape"])
    in_shape = tuple(si_utils.get_shape_from_sim(sim, asarray=True))
    is_noop = (
        allow_noop
        and out_shape == in_shape
        and np.allclose(matrix_prime, np.eye(ndim), rtol=0, atol=1e-10)
        and np.allclose(offset_prime, 0, rtol=0, atol=1e-10)
    )
    if is_noop:
        out_data = sim.data
    else:
        out_data = resample_array(sim.data, matrix_prime, offset_prime, out_shape, kwargs["order"], kwargs["cval"], device)
    scale = output_stack_properties["spacing"]
    trans = output_stack_properties["origin"]
    if not isinstance(scale, dict):
        scale = dict(zip(sdims, scale))
        trans = dict(zip(sdims, trans))
    return si_utils.to_spatial_image(out_data, dims=sim.dims, scale=scale, translation=trans)


def transform_pts(pts, affine):
    """transformation.transform_pts (transformation.py:150-161)."""
    pts = np.asarray(pts, dtype=np.float64)
    pts = np.concatenate([pts, np.ones((pts.shape[0], 1))], axis=1)
    return np.dot(pts, np.asarray(affine).T)[:, :-1]
