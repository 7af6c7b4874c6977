"""Chunk-wise fusion on the HIP backend (mirror of the reference's ``fusion`` API).

``fuse_np``  == fusion.fuse_np  (src/multiview_stitcher/fusion/_core.py:1513-1733)
``fuse``     == fusion.fuse     (_core.py:782-1501), eager, numpy- or device-backed
planner      == _build_spatial_fusion_plan and helpers (_core.py:354-722, 1736-1992)

The per-chunk arithmetic (resample + blend weights + normalise + fuse + cast)
runs as one fused kernel behind ``mvs_fuse_chunk``; this module only derives
the kernel arguments the way the reference derives scipy's.
"""

from __future__ import annotations

import ctypes as C
from itertools import product

import os
import shutil

import threading
import time

import numpy as np

from . import _lib, mv_graph, param_utils, weights
from . import spatial_image_utils as si_utils
from .device import DeviceArray, is_device_array
from .transformation import _as_zyx, embed3_stack, fill_view_geometry, get_pixel_affine, get_pixel_affines, shape3

BoundingBox = dict


# --- built-in fusion / weight functions: markers dispatched to kernel modes ------------------
def weighted_average_fusion(transformed_views=None, blending_weights=None, fusion_weights=None):
    """fusion.weighted_average_fusion (_core.py:61-94) -- computed inside mvs_fuse_chunk."""
    raise RuntimeError("weighted_average_fusion is a kernel mode of mvs_fuse_chunk; pass it as fusion_func")


def max_fusion(transformed_views=None):
    """fusion.max_fusion (_core.py:42-58) -- kernel mode."""
    raise RuntimeError("max_fusion is a kernel mode of mvs_fuse_chunk; pass it as fusion_func")


def simple_average_fusion(transformed_views=None):
    """fusion.simple_average_fusion (_core.py:97-131) -- kernel mode."""
    raise RuntimeError("simple_average_fusion is a kernel mode of mvs_fuse_chunk; pass it as fusion_func")


def content_based(transformed_views=None, blending_weights=None, sigma_1=5, sigma_2=11):
    """weights.content_based (weights.py:22-74) -- kernel mode (weights_func)."""
    raise RuntimeError("content_based is a kernel mode of mvs_fuse_chunk; pass it as weights_func")


content_based.required_overlap = lambda kwargs: 2 * (kwargs or {}).get("sigma_2", 11)

_FUSION_CODES = {
    weighted_average_fusion: _lib.MVS_FUSE_WEIGHTED_AVERAGE,
    max_fusion: _lib.MVS_FUSE_MAX,
    simple_average_fusion: _lib.MVS_FUSE_SIMPLE_AVERAGE,
    "weighted_average": _lib.MVS_FUSE_WEIGHTED_AVERAGE,
    "max": _lib.MVS_FUSE_MAX,
    "simple_average": _lib.MVS_FUSE_SIMPLE_AVERAGE,
}


# The reference's own function objects (multiview_stitcher.fusion.weighted_average_fusion, ...) select the kernel
# mode of the same name: `builtin(reference_func)` is what a backend="hip" branch inside the reference would pass on
# (INTEGRATION.md section 1).
BUILTIN = {
    "weighted_average_fusion": weighted_average_fusion,
    "max_fusion": max_fusion,
    "simple_average_fusion": simple_average_fusion,
    "content_based": content_based,
}


def builtin(func):
    """Map one of the reference's built-in fusion / weight functions (callable or name; None stays None) onto the
    kernel mode of the same name; anything else raises, custom callables cannot run inside mvs_fuse_chunk."""
    if func is None:
        return None
    name = func if isinstance(func, str) else getattr(func, "__name__", None)
    if name not in BUILTIN:
        raise NotImplementedError(f"{func!r} is not a built-in fusion/weights function of the hip backend")
    return BUILTIN[name]


def _fusion_code(fusion_func):
    try:
        return _FUSION_CODES[fusion_func]
    except (KeyError, TypeError):
        raise NotImplementedError(
            "backend='hip' fuses with the built-in fusion functions (weighted_average_fusion, "
            "max_fusion, simple_average_fusion); custom callables are not supported"
        ) from None


def _weights_code(weights_func):
    if weights_func is None:
        return _lib.MVS_WEIGHTS_NONE
    if weights_func is content_based or weights_func == "content_based":
        return _lib.MVS_WEIGHTS_CONTENT_BASED
    raise NotImplementedError("backend='hip' supports weights_func None or content_based")


def _bb_dicts(bb, sdims):
    """Accept dict-of-dicts (reference) or dict-of-arrays; return dict-of-dicts keyed by sdims."""
    out = {}
    for k in ("origin", "spacing", "shape"):
        v = bb[k]
        out[k] = dict(v) if isinstance(v, dict) else dict(zip(sdims, np.asarray(v).tolist()))
    return out


def _cb_overflowed(device):
    """True when a chunk of the fast content-based path since the last check could not list the voxels its mask lacks
    (counter ``cb_overflow``: waits for the context's stream, clears the flag)."""
    return _lib.get_counter("cb_overflow", device, reset=True) > 0


class _cb_exact:
    """Context manager: content-based weights through the bit-faithful passes (option ``cb_exact``) on ``device``."""

    def __init__(self, device):
        self.device = device

    def __enter__(self):
        _lib.set_option("cb_exact", 1, self.device)

    def __exit__(self, *exc):
        _lib.set_option("cb_exact", 0, self.device)
        return False


def fuse_np(
    sims,
    params,
    output_properties,
    fusion_func=weighted_average_fusion,
    fusion_func_kwargs=None,
    weights_func=None,
    weights_func_kwargs=None,
    trim_overlap_in_pixels=0,
    interpolation_order=1,
    full_view_bbs=None,
    spacings=None,
    origins=None,
    blending_widths=None,
    shrink_distance=0,
    backend="hip",
    output_on_backend=False,
    device=0,
    out=None,
    frame_origin=None,
    _record=None,
    _cb_check=True,
):
    """Fuse the slabs ``sims`` of one output chunk (fusion.fuse_np, _core.py:1513-1733).

    ``sims[i]`` is a SpatialImage slab (numpy- or DeviceArray-backed, spatial dims
    only), ``params[i]`` its view->world affine, ``output_properties`` the chunk
    bounding box including halo, ``full_view_bbs[i]`` the whole view's bounding
    box (for blending weights and spacing, _core.py:1611-1646).  Returns an
    array of the chunk shape minus the trimmed halo in the input dtype; a
    ``DeviceArray`` when ``output_on_backend`` (or ``out``) is given.

    ``frame_origin`` (dict per spatial dim, optional): the INDEX FRAME of include/mvs_hip.h.  The reference derives the
    pixel offsets of every view -- and of its blend-weight support grid -- from the chunk's and the slab's origins and
    rounds them to 10 decimals (transformation.py:72-83), so two chunkings of one stack differ by ~1e-9 px in the
    weights.  With a frame origin (``fuse`` passes the output stack's) the parameters are derived once per view, for that
    origin and the WHOLE view, and the chunk / slab enter as integer index shifts: a voxel gets the same result whatever
    chunk, launch block or shard it is computed in.  Needs chunk and slab origins on the frame's grids for ALL views of the
    chunk; otherwise an ``IndexFrameWarning`` is issued and the chunk falls back to per-chunk parameters (``fuse_shard``
    turns that warning into an error: its guarantee would be lost).  Voxel-exact agreement across chunkings holds on the
    translation fast path (identity pixel matrices); the generic kernel folds ``M @ origin`` into its offsets in floating
    point, so rotated / scaled views agree across chunkings to rounding (~1e-9 px in the coordinates), not bit for bit.
    """
    if backend not in ("hip", None):
        raise ValueError("multiview_stitcher_amd.fusion.fuse_np only implements backend='hip'")
    from .transformation import check_interpolation_order

    check_interpolation_order(interpolation_order, "interpolation_order")
    lib = _lib.init(device)
    sdims = si_utils.get_spatial_dims_from_sim(sims[0])
    ndim = len(sdims)
    out_bb = _bb_dicts(output_properties, sdims)
    out_origin = _as_zyx(out_bb["origin"], sdims)
    out_spacing = _as_zyx(out_bb["spacing"], sdims)
    out_shape = [int(out_bb["shape"][d]) for d in sdims]
    input_dtype = np.dtype(sims[0].dtype)
    if input_dtype not in _lib.DTYPE_CODES:
        raise TypeError(f"unsupported dtype {input_dtype} (uint8/uint16/float32)")
    if fusion_func not in _FUSION_CODES or (weights_func is not None and weights_func is not content_based):
        # user callables (docs/extension_api_fusion.md): they run after the resample, so the chunk cannot be fused in one
        # kernel; the voxel work that is ours (resample, blending weights) still runs on the device
        return _fuse_np_with_callables(
            sims, params, output_properties, fusion_func, fusion_func_kwargs, weights_func, weights_func_kwargs,
            trim_overlap_in_pixels, interpolation_order, full_view_bbs, spacings, blending_widths, shrink_distance,
            output_on_backend, device)
    fusion_code = _fusion_code(fusion_func)
    weights_code = _weights_code(weights_func)

    if spacings is None:
        spacings = [fvb["spacing"] for fvb in full_view_bbs] if full_view_bbs is not None else [None] * len(sims)
    if full_view_bbs is None:
        full_view_bbs = [si_utils.get_stack_properties_from_sim(s) for s in sims]

    n = len(sims)
    views = (_lib.mvs_view_t * n)()
    keep = []
    # the view records are computed for all views at once (stacked arrays: the per-view form of the same arithmetic costs
    # ~70 us of interpreter time per view) and written into the ctypes array through a byte view
    p_stack = np.stack([np.asarray(p, dtype=np.float64) for p in params])
    p_inv = np.linalg.inv(p_stack)
    in_spacings = np.stack([_as_zyx(sp if sp is not None else si_utils.get_spacing_from_sim(sim), sdims) for sim, sp in zip(sims, spacings)])
    in_origins = np.stack([si_utils.get_origin_from_sim(sim, asarray=True) for sim in sims])
    fv = [_bb_dicts(b, sdims) for b in full_view_bbs]
    full_origins = np.stack([_as_zyx(b["origin"], sdims) for b in fv])
    # index frame: chunk and slabs as integer shifts of parameters derived for (frame origin, whole view)
    index_origin, index_offsets = np.zeros(3, np.int64), np.zeros((n, 3), np.int64)
    ref_out_origin, ref_in_origins = out_origin, in_origins
    if frame_origin is not None:
        f_origin = _as_zyx(frame_origin, sdims)
        io_ = (out_origin - f_origin) / out_spacing
        so_ = (in_origins - full_origins) / in_spacings
        if np.all(np.abs(io_ - np.round(io_)) < 1e-6) and np.all(np.abs(so_ - np.round(so_)) < 1e-6):
            index_origin[3 - ndim:] = np.round(io_).astype(np.int64)
            index_offsets[:, 3 - ndim:] = np.round(so_).astype(np.int64)
            ref_out_origin, ref_in_origins = f_origin, full_origins
        else:
            import warnings

            off_c = float(np.abs(io_ - np.round(io_)).max()) if io_.size else 0.0
            off_s = float(np.abs(so_ - np.round(so_)).max()) if so_.size else 0.0      # (a chunk without views has no slabs)
            warnings.warn(
                "fuse_np: frame_origin cannot be applied -- the chunk origin or a slab origin is not on the frame's grid "
                f"(largest distance from it: chunk {off_c:.3g} px, slabs {off_s:.3g} px); the parameters of this chunk are derived per chunk, so its "
                "voxels may differ in the last bit from the same voxels fused through another chunk, launch block or shard",
                IndexFrameWarning, stacklevel=2)
            if _record is not None:
                _record["no_replay"] = True      # (a replayed call would not repeat the warning)
    matrices, offsets = get_pixel_affines(p_inv, ref_in_origins, in_spacings, ref_out_origin, out_spacing)
    tables, sup_origins, sup_spacings = weights.blending_supports(
        full_origins, np.stack([_as_zyx(b["spacing"], sdims) for b in fv]),
        np.stack([_as_zyx(b["shape"], sdims) for b in fv]), sdims, blending_widths, shrink_distance)
    w_matrices, w_offsets = get_pixel_affines(p_inv, sup_origins, sup_spacings, ref_out_origin, out_spacing)
    ptrs, shapes, strides = np.zeros(n, np.uint64), np.ones((n, 3), np.int64), np.zeros((n, 3), np.int64)
    mems = np.full(n, _lib.MVS_MEM_DEVICE, np.int32)
    for i, sim in enumerate(sims):
        data = sim.data
        if is_device_array(data):
            if data.dtype != input_dtype:
                raise TypeError("all views of a chunk must share one dtype")
            data = data.on_device(device)     # a tile resident on another GPU: peer copy, cached per (tile, device)
            data.wait_ready(device)           # a tile still on its way (device.to_device_async): this call's stream waits for the upload
            ptrs[i], st = data.ptr, [int(v) for v in data.strides]
        else:
            data = np.ascontiguousarray(data, dtype=input_dtype)
            ptrs[i] = data.ctypes.data
            st = [int(np.prod(data.shape[k + 1:])) for k in range(data.ndim)]   # numpy's strides of size-1 axes are arbitrary
            mems[i] = _lib.MVS_MEM_HOST
        s3 = shape3(data.shape)
        if len(st) == 2:  # 2D slab: a single z plane
            st = [st[0] * s3[1], st[0], st[1]]
        shapes[i], strides[i] = s3, st
        keep.append(data)
    V = _lib.mvs_view_t
    rec = np.frombuffer(views, dtype=np.uint8).reshape(n, C.sizeof(V))

    def field(name, dtype, count):
        off = getattr(V, name).offset
        return rec[:, off:off + count * np.dtype(dtype).itemsize].view(dtype)

    m3, o3 = embed3_stack(matrices, offsets)
    wm3, wo3 = embed3_stack(w_matrices, w_offsets)
    field("data", np.uint64, 1)[:, 0] = ptrs
    field("dtype", np.int32, 1)[:, 0] = _lib.DTYPE_CODES[input_dtype]
    field("mem", np.int32, 1)[:, 0] = mems
    field("shape", np.int64, 3)[:] = shapes
    field("stride", np.int64, 3)[:] = strides
    field("matrix", np.float64, 9)[:] = m3
    field("offset", np.float64, 3)[:] = o3
    field("w_matrix", np.float64, 9)[:] = wm3
    field("w_offset", np.float64, 3)[:] = wo3
    field("index_offset", np.int64, 3)[:] = index_offsets
    edt = field("edt", np.float32, 125)
    edt[:] = 0
    for i, table in enumerate(tables):
        edt[i, : table.size] = table.reshape(-1)

    if not isinstance(trim_overlap_in_pixels, dict):
        trim = {d: int(trim_overlap_in_pixels) for d in sdims}
    else:
        trim = {d: int(trim_overlap_in_pixels.get(d, 0)) for d in sdims}
    res_shape = [out_shape[i] - 2 * trim[d] for i, d in enumerate(sdims)]

    opts = _lib.mvs_fuse_opts_t()
    opts.ndim = ndim
    opts.order = int(interpolation_order)
    opts.fusion = fusion_code
    opts.weights = weights_code
    s3, t3 = shape3(out_shape), [0] * (3 - ndim) + [trim[d] for d in sdims]
    for k in range(3):
        opts.out_shape[k] = s3[k]
        opts.trim[k] = t3[k]
    wk = weights_func_kwargs or {}
    opts.sigma_1 = float(wk.get("sigma_1", 5))
    opts.sigma_2 = float(wk.get("sigma_2", 11))
    opts.out_dtype = _lib.DTYPE_CODES[input_dtype]
    for k in range(3):
        opts.index_origin[k] = int(index_origin[k])

    if out is not None or output_on_backend:
        if out is None:
            out = DeviceArray.empty(res_shape, input_dtype, device)
        if tuple(out.shape) != tuple(res_shape) or not out.is_contiguous():
            raise ValueError("out must be a contiguous DeviceArray of the result shape")
        opts.out_mem = _lib.MVS_MEM_DEVICE
        rc = lib.mvs_fuse_chunk(device, views, n, C.byref(opts), C.c_void_p(out.ptr))
        _lib.check(rc, device, "mvs_fuse_chunk")
        if weights_code and _cb_check and _cb_overflowed(device):
            # the fast content-based path lists the voxels its box-shaped mask lacks; a list that did not fit raised a flag
            # (a result left on the device is not waited for inside the call): this chunk again through the bit-faithful passes
            with _cb_exact(device):
                rc = lib.mvs_fuse_chunk(device, views, n, C.byref(opts), C.c_void_p(out.ptr))
                _lib.check(rc, device, "mvs_fuse_chunk")
        out.mark_written()
        if _record is not None and bool(np.all(mems == _lib.MVS_MEM_DEVICE)):
            # (fuse()'s geometry-keyed replay: the view records without their data pointers, the options, the result shape)
            _record.update(views=bytes(views), opts=bytes(opts), n=n, res_shape=tuple(int(v) for v in res_shape), ptrs=ptrs.copy())
        return out
    result = np.empty(tuple(res_shape), dtype=input_dtype)
    opts.out_mem = _lib.MVS_MEM_HOST
    rc = lib.mvs_fuse_chunk(device, views, n, C.byref(opts), result.ctypes.data)
    _lib.check(rc, device, "mvs_fuse_chunk")
    return result


def _host_weighted_average_fusion(transformed_views, blending_weights, fusion_weights=None):
    """weighted_average_fusion on host arrays (_core.py:61-94) -- used when a custom weights_func supplies fusion_weights."""
    if fusion_weights is None:
        additive = blending_weights
    else:
        additive = blending_weights * fusion_weights
        wsum = np.nansum(additive, axis=0)            # weights.normalize_weights (weights.py:325-345)
        wsum[wsum == 0] = 1
        additive = additive / wsum
    return np.nansum(transformed_views * additive, axis=0).astype(transformed_views[0].dtype)


def _host_max_fusion(transformed_views):
    """max_fusion on host arrays (_core.py:42-58)."""
    return np.nanmax(transformed_views, axis=0)


def _host_simple_average_fusion(transformed_views):
    """simple_average_fusion on host arrays (_core.py:97-131)."""
    nvalid = np.sum(~np.isnan(transformed_views), axis=0).astype(np.float32)
    nvalid[nvalid == 0] = np.nan
    return (np.nansum(transformed_views, axis=0) / nvalid).astype(transformed_views[0].dtype)


_HOST_FUSION = {weighted_average_fusion: _host_weighted_average_fusion, max_fusion: _host_max_fusion,
                simple_average_fusion: _host_simple_average_fusion}


def has_keyword(func, keyword):
    """misc_utils.has_keyword (misc_utils.py:69-80): does ``func`` accept ``keyword``?"""
    import inspect

    try:
        return keyword in inspect.signature(func).parameters
    except (TypeError, ValueError):
        return False


def _fuse_np_with_callables(sims, params, output_properties, fusion_func, fusion_func_kwargs, weights_func,
                            weights_func_kwargs, trim_overlap_in_pixels, interpolation_order, full_view_bbs, spacings,
                            blending_widths, shrink_distance, output_on_backend, device):
    """fuse_np for user-supplied ``fusion_func`` / ``weights_func`` callables (_core.py:1608-1733): every view is
    resampled with mvs_resample (float32, NaN outside), blending weights come from mvs_blend_weights and are
    normalised like weights.normalize_weights (weights.py:325-345); the callables then receive host float32
    arrays exactly as in the reference (``transformed_views`` (V, *S), ``blending_weights``, ``fusion_weights``,
    ``params``, ``output_spacing`` / ``output_chunksize`` when they ask for them)."""
    from .transformation import transform_sim

    # A custom weights_func with one of the built-in fusion functions (the documented extension case, _core.py:1663-1690):
    # the built-ins are kernel modes here, so their host form takes over behind the user's weights.
    fusion_func = _HOST_FUSION.get(fusion_func, fusion_func) if not isinstance(fusion_func, str) else _HOST_FUSION[BUILTIN[fusion_func + "_fusion"]]
    fusion_func_kwargs = dict(fusion_func_kwargs or {})
    weights_func_kwargs = dict(weights_func_kwargs or {})
    sdims = si_utils.get_spatial_dims_from_sim(sims[0])
    input_dtype = np.dtype(sims[0].dtype)
    if spacings is None:
        spacings = [fvb["spacing"] for fvb in full_view_bbs] if full_view_bbs is not None else [None] * len(sims)
    if full_view_bbs is None:
        full_view_bbs = [si_utils.get_stack_properties_from_sim(s) for s in sims]
    out_bb = _bb_dicts(output_properties, sdims)

    def host(a):
        return a.get() if is_device_array(a) else np.asarray(a)

    views_t = np.stack([
        host(transform_sim(sim, np.linalg.inv(np.asarray(param, dtype=np.float64)), output_stack_properties=out_bb,
                           input_spacing=spacing, order=interpolation_order, cval=np.nan, device=device,
                           allow_noop=False).data).astype(np.float32, copy=False)
        for sim, param, spacing in zip(sims, params, spacings)
    ])
    needs_blending = has_keyword(fusion_func, "blending_weights") or (
        weights_func is not None and has_keyword(weights_func, "blending_weights"))
    blend = None
    if needs_blending:
        blend = np.stack([
            weights.get_blending_weights(out_bb, _bb_dicts(full_view_bbs[i], sdims), params[i], blending_widths,
                                         shrink_distance, device)
            for i in range(len(sims))
        ])
        blend = blend * ~np.isnan(views_t)
        wsum = np.nansum(blend, axis=0)
        wsum[wsum == 0] = 1
        blend = blend / wsum
    fusion_func_kwargs["transformed_views"] = views_t
    if has_keyword(fusion_func, "params"):
        fusion_func_kwargs["params"] = params
    if has_keyword(fusion_func, "blending_weights"):
        fusion_func_kwargs["blending_weights"] = blend
    if has_keyword(fusion_func, "output_spacing") and "output_spacing" not in fusion_func_kwargs:
        fusion_func_kwargs["output_spacing"] = out_bb["spacing"]
    if weights_func is not None and has_keyword(fusion_func, "fusion_weights"):
        if weights_func is content_based:
            raise NotImplementedError("content_based weights with a custom fusion_func: pass a callable weights_func")
        weights_func_kwargs["transformed_views"] = views_t
        if has_keyword(weights_func, "params"):
            weights_func_kwargs["params"] = params
        if has_keyword(weights_func, "blending_weights"):
            weights_func_kwargs["blending_weights"] = blend
        if has_keyword(weights_func, "output_chunksize") and "output_chunksize" not in weights_func_kwargs:
            weights_func_kwargs["output_chunksize"] = out_bb["shape"]
        fusion_func_kwargs["fusion_weights"] = weights_func(**weights_func_kwargs)
    import warnings

    with warnings.catch_warnings():
        warnings.simplefilter("ignore", category=RuntimeWarning)   # func_ignore_nan_warning (_core.py:1684-1687)
        fused = np.asarray(fusion_func(**fusion_func_kwargs))
    if not isinstance(trim_overlap_in_pixels, dict):
        trim = {d: int(trim_overlap_in_pixels) for d in sdims}
    else:
        trim = {d: int(trim_overlap_in_pixels.get(d, 0)) for d in sdims}
    if any(trim[d] > 0 for d in sdims):
        fused = fused[tuple(slice(trim[d], -trim[d]) if trim[d] > 0 else slice(None) for d in sdims)]
    fused = np.nan_to_num(fused).astype(input_dtype)
    if output_on_backend:
        return DeviceArray.from_host(np.ascontiguousarray(fused), device)
    return fused


# --- output stack properties (_core.py:1736-1992) ---------------------------------------------
def calc_stack_properties_from_volume(volume, spacing):
    """_core.py:1972-1992: shape = floor(extent/spacing + 1e-9) + 1."""
    origin = volume[0]
    shape = np.floor((volume[1] - volume[0]) / spacing + 1e-9).astype(np.uint64) + 1
    return {"shape": shape, "spacing": spacing, "origin": origin}


def get_transformed_stack_vertices(stack_keypoints, stack_properties_list, params):
    """_core.py:1947-1969."""
    ndim = len(stack_properties_list[0]["spacing"])
    vertices = np.zeros((len(stack_properties_list), len(stack_keypoints), ndim))
    for iim, sp in enumerate(stack_properties_list):
        tmp = stack_keypoints * (np.array(sp["shape"]) - 1) * np.array(sp["spacing"]) + np.array(sp["origin"])
        vertices[iim] = np.dot(params[iim][:ndim, :ndim], tmp.T).T + params[iim][:ndim, ndim]
    return vertices


def calc_stack_properties_from_view_properties_and_params(views_props, params, spacing, mode="union"):
    """_core.py:1821-1899 (modes union / intersection / sample)."""
    sdims = ["z", "y", "x"][-len(spacing):]
    spacing = np.array([spacing[d] for d in sdims]).astype(float)
    views_props = [{k: np.array([v[d] for d in sdims]) for k, v in vp.items()} for vp in views_props]
    ndim = len(spacing)
    stack_vertices = np.array(list(np.ndindex(tuple([2] * ndim)))).astype(float)
    if mode == "sample":
        zface = stack_vertices[np.where(stack_vertices[:, 0] == 1)]
        zface[:, 2] = np.mean(zface[:, 2])
        tv = get_transformed_stack_vertices(zface, views_props, params)
        volume = np.min(np.min(tv, 1), 0), np.max(np.max(tv, 1), 0)
    elif mode == "union":
        tv = get_transformed_stack_vertices(stack_vertices, views_props, params)
        volume = np.min(np.min(tv, 1), 0), np.max(np.max(tv, 1), 0)
    elif mode == "intersection":
        tv = get_transformed_stack_vertices(stack_vertices, views_props, params)
        volume = np.max(np.min(tv, 1), 0), np.min(np.max(tv, 1), 0)
    else:
        raise ValueError(f"unknown output_stack_mode {mode!r}")
    return calc_stack_properties_from_volume(volume, spacing)


def combine_stack_props(stack_props_list):
    """_core.py:1902-1944."""
    origin = np.min([sp["origin"] for sp in stack_props_list], axis=0)
    spacing = np.min([sp["spacing"] for sp in stack_props_list], axis=0)
    shape = (
        np.max(
            [np.floor((sp["origin"] + (sp["shape"] - 1) * sp["spacing"] - origin) / spacing + 1e-9) for sp in stack_props_list],
            axis=0,
        ).astype(np.uint64)
        + 1
    )
    return {"origin": origin, "spacing": spacing, "shape": shape}


def calc_fusion_stack_properties(sims, params, spacing, mode="union"):
    """fusion.calc_fusion_stack_properties (_core.py:1736-1818); params may be t-stacked."""
    sdims = si_utils.get_spatial_dims_from_sim(sims[0])
    views_props = [si_utils.get_stack_properties_from_sim(sim) for sim in sims]
    params = [np.asarray(p, dtype=np.float64) for p in params]
    nt = max([p.shape[0] for p in params if p.ndim == 3] + [0])
    if nt:
        sp = combine_stack_props(
            [
                calc_stack_properties_from_view_properties_and_params(
                    views_props, [param_utils.select_time(p, it) for p in params], spacing, mode
                )
                for it in range(nt)
            ]
        )
    else:
        sp = calc_stack_properties_from_view_properties_and_params(views_props, params, spacing, mode)
    return {k: {d: (int(v[i]) if k == "shape" else float(v[i])) for i, d in enumerate(sdims)} for k, v in sp.items()}


def process_output_stack_properties(
    sims, output_spacing=None, output_origin=None, output_shape=None, output_stack_properties=None,
    output_stack_mode="union", transform_key=None,
):
    """_core.py:296-333."""
    if transform_key is None:
        raise ValueError("transform_key must be provided to determine transformation parameters")
    params = [si_utils.get_affine_from_sim(sim, transform_key) for sim in sims]
    if output_stack_properties is None:
        if output_spacing is None:
            output_spacing = si_utils.get_spacing_from_sim(sims[0])
        output_stack_properties = calc_fusion_stack_properties(sims, params, output_spacing, output_stack_mode)
        if output_origin is not None:
            output_stack_properties["origin"] = output_origin
        if output_shape is not None:
            output_stack_properties["shape"] = output_shape
    return output_stack_properties


def process_output_chunksize(sims, output_chunksize):
    """_core.py:248-277 (numpy-backed tiles -> the spatial_image_utils defaults)."""
    ndim = si_utils.get_ndim_from_sim(sims[0])
    sdims = si_utils.get_spatial_dims_from_sim(sims[0])
    if output_chunksize is None:
        output_chunksize = si_utils.get_default_spatial_chunksizes(ndim)
    elif isinstance(output_chunksize, int):
        output_chunksize = {dim: output_chunksize for dim in sdims}
    return output_chunksize


MAX_LAUNCH_BYTES = int(os.environ.get("MVS_MAX_LAUNCH_BYTES", 32 << 30))
MAX_STREAM_BYTES = int(os.environ.get("MVS_MAX_STREAM_BYTES", 1 << 30))      # launch block of a fuse() that reads or writes Zarr stores


def _merged_chunksize(chunksize, shape, sdims, itemsize, max_bytes=None):
    """Launch-block size for fuse(merge_chunks=True): whole multiples of the requested chunk size, as large as fits
    ``max_bytes`` of output -- the whole stack if possible, otherwise the block count is doubled along the axis with
    the longest blocks (first axis on ties) until a block fits."""
    max_bytes = MAX_LAUNCH_BYTES if max_bytes is None else int(max_bytes)
    cs = {d: max(int(chunksize[d]), 1) for d in sdims}
    nchunks = {d: -(-int(shape[d]) // cs[d]) for d in sdims}
    parts = {d: 1 for d in sdims}

    def block(d):
        return min(-(-nchunks[d] // parts[d]) * cs[d], int(shape[d]))

    while int(np.prod([block(d) for d in sdims])) * itemsize > max_bytes:
        cand = [d for d in sdims if parts[d] < nchunks[d]]
        if not cand:
            break
        d = max(cand, key=lambda d_: block(d_))
        parts[d] = min(parts[d] * 2, nchunks[d])
    return {d: block(d) for d in sdims}


def _launch_budget(sims, out_shape, sdims, itemsize, device, cap):
    """Output bytes one launch block of fuse(merge_chunks=True) may take so that the launch fits the device: the block
    itself plus the view slabs that have to be STAGED for it (host- or Zarr-backed views, and views resident on another
    GPU, are copied into device scratch; views already on ``device`` cost nothing) must stay below 90 % of the free
    device memory (``mvs_mem_info``).  The staged share is estimated per output byte: twice the mosaic-average (overlap
    zones hold every voxel two to eight times) plus one whole view.  Never above ``cap`` (MVS_MAX_LAUNCH_BYTES /
    MVS_MAX_STREAM_BYTES)."""
    try:
        free, _ = _lib.mem_info(device)
    except (RuntimeError, OSError, AttributeError):
        return cap
    staged, largest = 0, 0
    for s in sims:
        data = s.data
        if is_device_array(data) and (data.device & 0xff) == (int(device) & 0xff):
            continue
        nb = int(np.prod([s.sizes[d] for d in sdims])) * itemsize
        staged += nb
        largest = max(largest, nb)
    out_total = max(int(np.prod([int(out_shape[d]) for d in sdims])) * itemsize, 1)
    avail = int(0.9 * free) - largest
    per_out_byte = 1.0 + 2.0 * staged / out_total
    return int(max(min(cap, avail / per_out_byte), 1))


class IndexFrameWarning(RuntimeWarning):
    """fuse_np was given a frame_origin it could not apply (an origin off the frame's grid)."""


def _is_device_memory_error(exc):
    """Out of device memory, by the library's error CODE (MVS_ERR_OUT_OF_MEMORY <- hipErrorOutOfMemory), not by message text."""
    return isinstance(exc, _lib.DeviceMemoryError)


# --- chunk -> view-slab plan (_core.py:354-722): computed by the library ---------------------------------
_PLAN_ENTRY = np.dtype([("block", "<i8", (3,)), ("view", "<i4"), ("planewise", "<i4"), ("lo", "<i8", (3,)), ("n", "<i8", (3,))])


def _plan_chunks(sparams, views_bb, output_stack_properties, output_chunksize, overlap_in_pixels, interpolation_order, sdims):
    """Which views, and which index window of each, feed which output chunk: one ``mvs_fuse_plan`` call (host code in the
    library, csrc/mvs_plan.hip) for the whole chunk grid.  Returns ``(by_block, info)``: ``by_block[block_index]`` =
    ``(planewise, [(iview, lo, n), ...])`` (views ascending; blocks without contributing views are absent) and ``info`` with
    the axes that are pure translations / lie on the views' sampling grid."""
    nd, nv = len(sdims), len(sparams)
    f64 = lambda rows: np.ascontiguousarray(rows, dtype=np.float64)
    i64 = lambda rows: np.ascontiguousarray(rows, dtype=np.int64)
    vo = f64([[vbb["origin"][d] for d in sdims] for vbb in views_bb]).reshape(nv, nd)
    vs = f64([[vbb["spacing"][d] for d in sdims] for vbb in views_bb]).reshape(nv, nd)
    vn = i64([[vbb["shape"][d] for d in sdims] for vbb in views_bb]).reshape(nv, nd)
    P = f64(np.stack([np.asarray(p, dtype=np.float64) for p in sparams])) if nv else np.zeros((0, nd + 1, nd + 1))
    try:
        Pinv = f64(np.linalg.inv(P)) if nv else P
    except np.linalg.LinAlgError:
        raise ValueError("a view's affine is singular") from None
    oo = f64([output_stack_properties["origin"][d] for d in sdims])
    osp = f64([output_stack_properties["spacing"][d] for d in sdims])
    on = i64([output_stack_properties["shape"][d] for d in sdims])
    cs = i64([output_chunksize[d] for d in sdims])
    halo = i64([overlap_in_pixels[d] for d in sdims])
    lib = _lib.load()
    dp = lambda a: a.ctypes.data_as(C.POINTER(C.c_double))
    ip = lambda a: a.ctypes.data_as(C.POINTER(C.c_int64))
    count = C.c_int64(0)
    masks = (C.c_int32 * 2)()

    def call(buf, capacity):
        rc = lib.mvs_fuse_plan(nd, nv, dp(vo), dp(vs), ip(vn), dp(P), dp(Pinv), dp(oo), dp(osp), ip(on), ip(cs), ip(halo),
                               int(interpolation_order), buf, capacity, C.byref(count), masks)
        if rc != 0:
            raise RuntimeError(f"mvs_fuse_plan failed (code {rc})")

    call(None, 0)
    recs = np.zeros(max(int(count.value), 1), dtype=_PLAN_ENTRY)
    call(recs.ctypes.data_as(C.c_void_p), len(recs))
    recs = recs[: int(count.value)]
    by_block = {}
    blocks, views, pw = recs["block"][:, :nd].tolist(), recs["view"].tolist(), recs["planewise"].tolist()
    los, ns = recs["lo"][:, :nd].tolist(), recs["n"][:, :nd].tolist()
    for blk, v, p, lo, n in zip(blocks, views, pw, los, ns):
        by_block.setdefault(tuple(blk), (bool(p), []))[1].append((v, tuple(lo), tuple(n)))
    info = {"axis_aligned_translation_dims": [d for k, d in enumerate(sdims) if (masks[0] >> k) & 1],
            "grid_aligned_translation_dims": [d for k, d in enumerate(sdims) if (masks[1] >> k) & 1]}
    return by_block, info


_REPLAY = [True]           # tests / A-B: derive everything on every call
_HOST_STREAM = [os.environ.get("MVS_HOST_STREAM", "1") != "0"]            # fuse() of plain host arrays (>= 256 MiB) through the block pipeline
_HOST_STREAM_MIN_BYTES = 256 << 20
_STREAM_TILES = [os.environ.get("MVS_STREAM_TILES", "1") != "0"]            # fused blocks re-tiled on the device into chunk-major order before the download
_STREAM_PIPELINE = [os.environ.get("MVS_STREAM_PIPELINE", "1") != "0"]      # streaming.BlockPipeline around the launch blocks of a streamed fuse()
_REPLAY_MEMO = {}
_REPLAY_CAP = 32                     # geometries kept (8 ranks x a few mosaics; an entry is a few KB of view records)
_REPLAY_LOCK = threading.Lock()      # fuse() is also driven from one thread per GPU: lookups, evictions and inserts are serialised


def _hashable(v):
    if v is None or isinstance(v, (int, float, str, bool)):
        return v
    if isinstance(v, dict):
        return tuple(sorted((k, _hashable(x)) for k, x in v.items()))
    if isinstance(v, (list, tuple)):
        return tuple(_hashable(x) for x in v)
    if isinstance(v, np.ndarray):
        return (v.shape, str(v.dtype), v.tobytes())
    if isinstance(v, (np.integer, np.floating)):
        return v.item()
    raise TypeError


def _replay_key(images, transform_key, fusion_func, device, *args):
    """Everything fuse()'s host work depends on for device-resident plain images, as a hashable -- or None when the call is not
    of that kind.  Per image: the spatial dims, first / second coordinate and length of every axis (origin, spacing, shape as
    the stack-property getters read them), the transform under ``transform_key``, dtype, strides and device of the tile."""
    from . import msi_utils

    try:
        first = images[0]
        if msi_utils.is_msim(first) or fusion_func not in _FUSION_CODES:
            return None
        dims = tuple(first.dims)
        if any(d not in ("z", "y", "x") for d in dims):
            return None
        dtype = np.dtype(first.dtype)
        geo = np.empty((len(images), len(dims), 3))
        meta, trs = [], []
        for i, im in enumerate(images):
            data = im.data
            if msi_utils.is_msim(im) or tuple(im.dims) != dims or np.dtype(data.dtype) != dtype:
                return None
            if type(data).__name__ == "RemoteArray":
                # a tile this rank does not hold (sharding.RemoteArray: shape and dtype only): it takes part in the geometry; a
                # record is only made -- and replayed -- when none of the contributing views is one of these
                meta.append(("remote", tuple(data.shape)))
            elif is_device_array(data) and (data.device & 0xff) == (int(device) & 0xff):
                meta.append((data.shape, data.strides, data.device))
            else:
                return None
            co = im.coords
            for k, d in enumerate(dims):
                c = co[d]
                n = len(c)
                geo[i, k, 0], geo[i, k, 1], geo[i, k, 2] = c[0], (c[1] if n > 1 else c[0]), n
            trs.append(np.asarray(im.attrs["transforms"][transform_key], dtype=np.float64))
        shape0 = trs[0].shape
        if any(t.shape != shape0 for t in trs):
            return None
        return (dims, str(dtype), geo.tobytes(), np.stack(trs).tobytes(), shape0, tuple(meta), transform_key, fusion_func, int(device),
                tuple(_hashable(a) for a in args))
    except (KeyError, TypeError, AttributeError, IndexError):
        return None


def _replay_fuse(rec, images, transform_key, device):
    """fuse() from a remembered derivation: the view records with the CURRENT tiles' pointers, one mvs_fuse_chunk launch."""
    n = rec["n"]
    views = (_lib.mvs_view_t * n).from_buffer_copy(rec["views"])
    opts = _lib.mvs_fuse_opts_t.from_buffer_copy(rec["opts"])
    for iv in rec["view_index"]:
        images[iv].data.wait_ready(device)
    ptrs = np.array([images[iv].data.ptr for iv in rec["view_index"]], dtype=np.uint64) + rec["byte_offsets"]
    flat = np.frombuffer(views, dtype=np.uint8).reshape(n, C.sizeof(_lib.mvs_view_t))
    off = _lib.mvs_view_t.data.offset
    flat[:, off:off + 8].view(np.uint64)[:, 0] = ptrs
    out = DeviceArray.empty(rec["res_shape"], rec["dtype"], device)
    lib = _lib.init(device)
    rc = lib.mvs_fuse_chunk(device, views, n, C.byref(opts), C.c_void_p(out.ptr))
    _lib.check(rc, device, "mvs_fuse_chunk")
    out.mark_written()
    res = si_utils.to_spatial_image(out, dims=list(rec["dims"]), scale=rec["spacing"], translation=rec["origin"])
    si_utils.set_sim_affine(res, param_utils.identity_transform(rec["ndim"]), transform_key)
    return res


def _fuse_once(
    images=None,
    transform_key=None,
    fusion_func=weighted_average_fusion,
    fusion_func_kwargs=None,
    weights_func=None,
    weights_func_kwargs=None,
    output_spacing=None,
    output_stack_mode="union",
    output_origin=None,
    output_shape=None,
    output_stack_properties=None,
    output_chunksize=None,
    overlap_in_pixels=None,
    trim_overlap=True,
    interpolation_order=1,
    blending_widths=None,
    output_zarr_url=None,
    zarr_options=None,
    batch_options=None,
    backend="hip",
    output_on_backend=False,
    sims=None,
    device=0,
    chunk_filter=None,
    merge_chunks=True,
    frame_origin=None,
):
    """Fuse input views (fusion.fuse, _core.py:782-1501), eagerly, on the HIP backend.

    Same arguments as the reference.  Differences forced by the environment:
    evaluation is eager (there is no dask) and ``images`` are numpy-,
    DeviceArray- or zarr-backed SpatialImages (``ngff_utils.read_sim_from_ome_zarr``;
    only the slab a chunk needs is read).  With ``output_zarr_url`` every
    fused chunk is written into a Zarr v2 array (``zarr_options``: ``ome_zarr``,
    ``ngff_version`` "0.4", ``overwrite``, ``zarr_array_creation_kwargs``) and
    the returned image is backed by that array.  Output chunks are fused one ``mvs_fuse_chunk`` call each,
    following the reference's chunk grid, halo and slab windows; the result is
    a SpatialImage with identity affine under ``transform_key``.
    ``chunk_filter(block_index) -> bool`` restricts the work to a subset of
    chunks (used by the multi-GPU farm); untouched chunks stay zero.
    ``merge_chunks``: with the built-in fusion functions and no halo the chunk grid is only the reference's unit of dask
    scheduling (and the chunk grid of a Zarr output) -- every output voxel is the same function of the views whichever
    chunk it falls in -- so the requested chunks are merged into launch blocks of whole chunks: up to ``MAX_LAUNCH_BYTES``
    of output for in-memory results (the whole mosaic when it fits: one ``mvs_fuse_chunk`` launch instead of hundreds), up to
    ``MAX_STREAM_BYTES`` when tiles are read from or the result is written to a Zarr store (a block is fused in one launch
    and written into its chunk files).  ``batch_options`` and ``chunk_filter`` address single chunks and switch this off.
    ``frame_origin``: origin of the index frame all chunks are fused in (see ``fuse_np``); default: the output stack's
    origin, so chunked, merged and unchunked runs of one stack agree voxel for voxel.  ``sharding.fuse_shard`` passes the
    origin of the WHOLE mosaic, so that every rank's sub-box equals the corresponding part of the single-GPU result.
    """
    if images is None:
        if sims is None:
            raise TypeError("fuse() missing 1 required positional argument: 'images'")
        images = sims
    elif sims is not None:
        raise TypeError("fuse() got both 'images' and deprecated 'sims'. Use only 'images'.")
    if not images:
        raise ValueError("images must contain at least one image.")
    # Device-resident tiles fused into one device-resident launch block: everything the interpreter derives for the call -- output
    # stack, chunk plan, slab windows, the view records of mvs_fuse_chunk -- is a function of the views' geometry and the
    # arguments, not of the voxels.  It is derived once per geometry and replayed with the current data pointers afterwards
    # (a register + fuse loop over time points or channels of one mosaic pays the ~2 ms of host work once).
    fast_key, record = None, None
    if (_REPLAY[0] and output_on_backend and output_zarr_url is None and not batch_options and chunk_filter is None and merge_chunks
            and weights_func is None and not fusion_func_kwargs and not weights_func_kwargs and not zarr_options and backend in ("hip", None)):
        fast_key = _replay_key(images, transform_key, fusion_func, device, output_spacing, output_stack_mode, output_origin, output_shape,
                               output_stack_properties, output_chunksize, overlap_in_pixels, trim_overlap, interpolation_order,
                               blending_widths, frame_origin)
        if fast_key is not None:
            with _REPLAY_LOCK:
                hit = _REPLAY_MEMO.get(fast_key)
            if hit is not None:
                return _replay_fuse(hit, images, transform_key, device)
            record = {}
    if output_zarr_url is not None and output_on_backend:
        raise ValueError("output_zarr_url streams chunks to disk; it cannot be combined with output_on_backend")
    if backend not in ("hip", None):
        raise ValueError("multiview_stitcher_amd.fusion.fuse only implements backend='hip'")
    from . import msi_utils

    is_ms = [msi_utils.is_msim(im) for im in images]
    if any(is_ms) and not all(is_ms):
        raise ValueError("All input images must be of the same kind: either all SpatialImages or all MultiscaleSpatialImages.")
    if all(is_ms):
        # MultiscaleSpatialImages in, a multiscale result out (fusion/_core.py:939-1064): scale0 defines the finest output
        # geometry; every output level is FUSED from the coarsest input level that is still fine enough for it (not
        # downsampled from the level above); a Zarr output is one level fused from the matching input level.
        msims = list(images)
        common = dict(transform_key=transform_key, fusion_func=fusion_func, fusion_func_kwargs=fusion_func_kwargs,
                      weights_func=weights_func, weights_func_kwargs=weights_func_kwargs, output_stack_mode=output_stack_mode,
                      output_chunksize=output_chunksize, overlap_in_pixels=overlap_in_pixels, trim_overlap=trim_overlap,
                      interpolation_order=interpolation_order, blending_widths=blending_widths, backend=backend, device=device,
                      chunk_filter=chunk_filter, merge_chunks=merge_chunks)
        scale0 = [msi_utils.get_sim_from_msim(m, scale="scale0") for m in msims]
        sdims0 = si_utils.get_spatial_dims_from_sim(scale0[0])
        osp0 = _bb_dicts(process_output_stack_properties(scale0, output_spacing, output_origin, output_shape, output_stack_properties,
                                                         output_stack_mode, transform_key), sdims0)

        def level_sims(spacing):
            return [msi_utils.get_sim_from_msim(m, scale="scale%d" % msi_utils.get_res_level_from_spacing(m, spacing)) for m in msims]

        if output_zarr_url is not None:
            fused = fuse(images=level_sims(osp0["spacing"]), output_stack_properties=osp0, output_zarr_url=output_zarr_url,
                               zarr_options=zarr_options, batch_options=batch_options, frame_origin=frame_origin, **common)
            if (zarr_options or {}).get("ome_zarr", False) and chunk_filter is None:
                from . import ngff_utils

                return ngff_utils.read_msim_from_ome_zarr(output_zarr_url, transform_key=transform_key if transform_key is not None
                                                          else si_utils.DEFAULT_TRANSFORM_KEY)
            return msi_utils.get_msim_from_sim(fused, scale_factors=[])
        shapes, _, abs_factors = msi_utils.calc_resolution_levels({d: int(osp0["shape"][d]) for d in sdims0})
        fused_levels = []
        for shape, f in zip(shapes, abs_factors):
            props = {"shape": dict(shape), "spacing": {d: osp0["spacing"][d] * f[d] for d in sdims0},
                     # centre-of-pixel convention of downsampled levels (as in the OME-Zarr pyramid)
                     "origin": {d: osp0["origin"][d] + (f[d] - 1) * osp0["spacing"][d] / 2 for d in sdims0}}
            # (a caller's frame_origin refers to the scale0 grid: a coarser level's grid is displaced by (f - 1) * spacing / 2
            # against it, so the level keeps its own frame = its own stack origin)
            level0 = all(int(f[d]) == 1 for d in sdims0)
            fused_levels.append(fuse(images=level_sims(props["spacing"]), output_stack_properties=props,
                                           output_on_backend=output_on_backend, frame_origin=frame_origin if level0 else None, **common))
        return msi_utils.get_msim_from_sims(fused_levels)
    sims_ = list(images)

    from .transformation import check_interpolation_order

    check_interpolation_order(interpolation_order, "interpolation_order")
    output_chunksize = process_output_chunksize(sims_, output_chunksize)
    output_stack_properties = process_output_stack_properties(
        sims_, output_spacing, output_origin, output_shape, output_stack_properties, output_stack_mode, transform_key
    )
    sdims = si_utils.get_spatial_dims_from_sim(sims_[0])
    nsdims = si_utils.get_nonspatial_dims_from_sim(sims_[0])
    output_stack_properties = _bb_dicts(output_stack_properties, sdims)
    output_stack_properties["shape"] = {d: int(v) for d, v in output_stack_properties["shape"].items()}
    params = [si_utils.get_affine_from_sim(sim, transform_key) for sim in sims_]

    # halo (_core.py:1194-1222)
    overlap_in_pixels = overlap_in_pixels or 0
    if not isinstance(overlap_in_pixels, dict):
        overlap_in_pixels = {d: overlap_in_pixels for d in sdims}
    shrink_distance = 0
    for func, kw in [(weights_func, weights_func_kwargs), (fusion_func, fusion_func_kwargs)]:
        if func is not None and hasattr(func, "required_overlap"):
            cur = func.required_overlap(dict(kw or {}))
            if not isinstance(cur, dict):
                cur = {d: cur for d in sdims}
            overlap_in_pixels = {d: max(overlap_in_pixels[d], cur[d]) for d in sdims}

    store_chunksize = dict(output_chunksize)          # the chunk grid of a Zarr output stays the requested one
    requested_chunksize = dict(output_chunksize)
    merged = False
    streamed = output_zarr_url is not None or any(type(s_.data).__name__ in ("ZarrArray", "ZarrView") for s_ in sims_)
    if not streamed and _HOST_STREAM[0] and _STREAM_PIPELINE[0] and not output_on_backend and not batch_options and chunk_filter is None \
            and fusion_func in _FUSION_CODES and (weights_func is None or weights_func is content_based) \
            and all(isinstance(s_.data, np.ndarray) or is_device_array(s_.data) for s_ in sims_) \
            and sum(int(np.prod(s_.data.shape)) * np.dtype(s_.dtype).itemsize for s_ in sims_) >= _HOST_STREAM_MIN_BYTES \
            and _lib.device_count() > 0:
        # plain host arrays (or resident tiles) in, host array out -- what a user of the reference calls: launch blocks of <= 1 GiB through the block
        # pipeline (slabs copied into pinned staging buffers by the I/O pool, asynchronous transfers under the launch blocks, results
        # copied out by the pool) instead of ONE launch block whose views mvs_fuse_chunk uploads from pageable memory, fuses and
        # downloads one after the other: the north star 0.98 -> 0.41 s (2.26 -> 1.03 s for the first call of a process); resident tiles
        # with a host result 0.64 -> 0.24 s
        streamed = True
    if (merge_chunks and not batch_options and chunk_filter is None
            and weights_func is None and fusion_func in _FUSION_CODES and not any(overlap_in_pixels[d] for d in sdims)
            and not ("z" in sdims and int(output_chunksize["z"]) == 1 and output_stack_properties["shape"]["z"] > 1)):
        # streamed inputs / outputs pass through host memory block by block: a smaller budget per launch block
        itemsize = np.dtype(sims_[0].dtype).itemsize
        budget = _launch_budget(sims_, output_stack_properties["shape"], sdims, itemsize, device,
                                MAX_STREAM_BYTES if streamed else MAX_LAUNCH_BYTES)
        output_chunksize = _merged_chunksize(output_chunksize, output_stack_properties["shape"], sdims, itemsize, budget)
        merged = any(int(output_chunksize[d]) != int(requested_chunksize[d]) for d in sdims)

    chunk_bbs, block_indices = mv_graph.get_chunk_bbs(output_stack_properties, output_chunksize)
    chunk_bbs_ov = [
        cb
        | {"origin": {d: cb["origin"][d] - overlap_in_pixels[d] * output_stack_properties["spacing"][d] for d in sdims}}
        | {"shape": {d: cb["shape"][d] + 2 * overlap_in_pixels[d] for d in sdims}}
        for cb in chunk_bbs
    ]
    chunk_bbs_res = chunk_bbs if trim_overlap else chunk_bbs_ov
    views_bb = [si_utils.get_stack_properties_from_sim(sim) for sim in sims_]
    norm_chunks = mv_graph.normalize_chunks([output_chunksize[d] for d in sdims], [output_stack_properties["shape"][d] for d in sdims])
    untrimmed = (not trim_overlap) and any(overlap_in_pixels[d] for d in sdims)
    if untrimmed:
        # trim_overlap=False (_core.py:1252-1254, 1687-1711): every chunk keeps its halo and the result is the block
        # assembly of the untrimmed chunks side by side (da.block of chunks of shape chunk + 2 * halo), i.e. an array
        # that is larger than the output stack by 2 * halo per chunk and axis
        if output_zarr_url is not None:
            raise NotImplementedError("trim_overlap=False assembles untrimmed chunks in memory; it cannot stream to a Zarr store")
        norm_chunks = [tuple(int(c) + 2 * int(overlap_in_pixels[d]) for c in cs_) for cs_, d in zip(norm_chunks, sdims)]
    block_offsets = [np.cumsum((0,) + c[:-1]) for c in norm_chunks]

    out_shape_sp = tuple(int(sum(c)) for c in norm_chunks) if untrimmed else tuple(output_stack_properties["shape"][d] for d in sdims)
    ns_shape = tuple(sims_[0].sizes[d] for d in nsdims)
    dtype = np.dtype(sims_[0].dtype)
    on_device = output_on_backend
    zarr_out = None
    if output_zarr_url is not None:
        # streaming output (_core.py:1068-1171, 2044-2156): every fused chunk goes straight into its region of a Zarr v2
        # array, the mosaic never exists in host memory; with ome_zarr=True the array is level "0" of an NGFF image
        from . import ngff_utils, zarr_io

        zarr_options = dict(zarr_options or {})
        ome_zarr = bool(zarr_options.get("ome_zarr", False))
        ngff_version = zarr_options.get("ngff_version", "0.4")
        create_kw = dict(zarr_options.get("zarr_array_creation_kwargs") or {})
        if create_kw.get("chunks") is not None:
            # the store's chunk grid: full rank (c, t, spatial) or spatial dims only, as write_sim_to_ome_zarr takes it.  Every
            # fused block is written into its region, so the fuse chunk grid must be made of whole store chunks.
            req = [int(v) for v in create_kw["chunks"]]
            if len(req) == len(nsdims) + len(sdims):
                req = req[len(nsdims):]
            if len(req) != len(sdims) or min(req) < 1:
                raise ValueError(f"zarr_array_creation_kwargs['chunks'] {create_kw['chunks']} does not match dims {list(nsdims) + list(sdims)}")
            for d, c in zip(sdims, req):
                if int(requested_chunksize[d]) % c and int(requested_chunksize[d]) < int(output_stack_properties["shape"][d]):
                    raise ValueError(f"store chunks {req} do not tile the fuse chunks {[int(requested_chunksize[d_]) for d_ in sdims]}")
            store_chunksize = dict(zip(sdims, req))
        create_kw.pop("chunks", None)
        if ome_zarr_requested := bool(zarr_options.get("ome_zarr", False)):
            want_fmt = 3 if str(zarr_options.get("ngff_version", "0.4")) == "0.5" else 2
            if int(create_kw.get("zarr_format", want_fmt)) != want_fmt:
                raise ValueError(f"zarr_format {create_kw['zarr_format']} conflicts with NGFF {zarr_options.get('ngff_version', '0.4')} "
                                 f"(which stores Zarr v{want_fmt} arrays)")
        if zarr_options.get("overwrite", True) and os.path.exists(output_zarr_url) and chunk_filter is None:
            shutil.rmtree(output_zarr_url)
        if ome_zarr:
            create_kw = ngff_utils.update_zarr_array_creation_kwargs_for_ngff_version(ngff_version, create_kw)
            zarr_io.create_group(output_zarr_url, **ngff_utils.zarr_group_creation_kwargs_for_ngff_version(ngff_version))
            if create_kw.get("zarr_format") == 3:
                create_kw.setdefault("dimension_names", list(nsdims) + list(sdims))
        store_url = os.path.join(output_zarr_url, "0") if ome_zarr else output_zarr_url
        if zarr_io.array_exists(store_url):
            zarr_out = zarr_io.ZarrArray.open(store_url)      # a farm worker joining an array another worker created
        else:
            zarr_out = zarr_io.ZarrArray.create(
                store_url, ns_shape + out_shape_sp, (1,) * len(ns_shape) + tuple(store_chunksize[d] for d in sdims), dtype,
                **create_kw)
    result = None if (on_device or zarr_out is not None) else np.zeros(ns_shape + out_shape_sp, dtype=dtype)
    # output_on_backend: one device array for all (c, t) fields (_core.py:1275-1306 loops the fields); every field is fused
    # into its own contiguous sub-array.  A single field keeps the spatial dims only, as before.
    n_fields = int(np.prod(ns_shape)) if ns_shape else 1
    dev_full = DeviceArray.empty((ns_shape if n_fields > 1 else ()) + out_shape_sp, dtype, device) if on_device else None

    # batch_options (_core.py:1068-1141, 2044-2156): with a Zarr output the reference hands batches of block ids to
    # batch_func(fuse_chunk, block_ids, **batch_func_kwargs); fuse_chunk(block_id) fuses one block and writes its region
    batch_options = dict(batch_options or {})
    unknown = set(batch_options) - {"batch_func", "n_batch", "batch_func_kwargs"}
    if unknown:
        raise TypeError(f"unknown batch_options keys {sorted(unknown)}")
    if batch_options and output_zarr_url is None:
        raise ValueError("batch_options drive the block-wise Zarr output of fuse(); pass output_zarr_url as well")
    if batch_options and chunk_filter is not None:
        raise ValueError("batch_options and chunk_filter both select blocks; use one of them")

    plan_cache = {}

    def plan_for(it):
        """(affines of time point ``it``, plan): ``plan["per_chunk_entries"]`` lists, in block order, the chunks with their
        boxes, the contributing views and the index window of each view (``mvs_fuse_plan``)."""
        key = it if any(np.asarray(p).ndim == 3 for p in params) else 0
        if key not in plan_cache:
            sparams = [param_utils.select_time(p, it) for p in params]
            by_block, info = _plan_chunks(sparams, views_bb, output_stack_properties, output_chunksize, overlap_in_pixels,
                                          interpolation_order, sdims)
            entries = []
            for cbb, cbb_ov, cbb_res, block_index in zip(chunk_bbs, chunk_bbs_ov, chunk_bbs_res, block_indices):
                planewise, chunk_views = by_block.get(tuple(int(b) for b in block_index), (False, []))
                entries.append({"views": chunk_views, "output_bb": cbb, "output_bb_overlap": cbb_ov, "output_bb_result": cbb_res,
                                "fuse_planewise": planewise, "block_index": tuple(block_index)})
            plan_cache[key] = (sparams, dict(info, per_chunk_entries=entries, sparams=sparams))
        return plan_cache[key]

    def chunk_call(ns_index, entry, dev):
        """fuse_np arguments of one (field, block) and its window in the result."""
        ns_sel = {d: int(i) for d, i in zip(nsdims, ns_index)}
        sparams, _ = plan_for(ns_sel.get("t", 0))
        bi = entry["block_index"]
        cbb_ov = entry["output_bb_overlap"]
        slabs = [sims_[iv].isel(dict(ns_sel, **{d: slice(a, a + m) for d, a, m in zip(sdims, lo, n)})) for iv, lo, n in entry["views"]]
        idxs = [iv for iv, _, _ in entry["views"]]
        if entry["fuse_planewise"]:
            slabs = [s.isel({"z": 0}) for s in slabs]
            tmp_params = [sparams[iv][1:, 1:] for iv in idxs]
            cbb_use = mv_graph.project_bb_along_dim(cbb_ov, "z")
            fvb = [mv_graph.project_bb_along_dim(views_bb[iv], "z") for iv in idxs]
        else:
            tmp_params = [sparams[iv] for iv in idxs]
            cbb_use = cbb_ov
            fvb = [views_bb[iv] for iv in idxs]
        sl = tuple(
            slice(int(block_offsets[i][bi[i]]), int(block_offsets[i][bi[i]]) + int(entry["output_bb_result"]["shape"][d]))
            for i, d in enumerate(sdims)
        )
        kwargs = dict(
            sims=slabs, params=tmp_params, output_properties=cbb_use, fusion_func=fusion_func,
            fusion_func_kwargs=fusion_func_kwargs, weights_func=weights_func,
            weights_func_kwargs=weights_func_kwargs,
            trim_overlap_in_pixels=(overlap_in_pixels if trim_overlap else 0),
            interpolation_order=interpolation_order, full_view_bbs=fvb, blending_widths=blending_widths,
            shrink_distance=shrink_distance, backend="hip", device=dev, _cb_check=False,
        )
        if fusion_func in _FUSION_CODES and (weights_func is None or weights_func is content_based):
            fo_ = frame_origin if frame_origin is not None else output_stack_properties["origin"]
            kwargs["frame_origin"] = {d: fo_[d] for d in cbb_use["origin"]}
            if record is not None and not entry["fuse_planewise"]:
                kwargs["_record"] = record
                record["calls"] = record.get("calls", 0) + 1
                record["view_index"] = idxs
        return kwargs, sl

    if batch_options:
        # ---- block-wise Zarr output driven by batch_func ----
        nblocks = tuple(ns_shape) + tuple(len(c) for c in norm_chunks)
        by_block_cache = {}

        def fuse_chunk(block_id, device=device):
            """Fuse block ``block_id`` (index into the chunk grid of the output array, non-spatial axes first) and write
            it into its region of the output store; ``device`` lets a batch function place blocks on several GPUs."""
            block_id = tuple(int(b) for b in block_id)
            ns_index, bi = block_id[: len(ns_shape)], block_id[len(ns_shape):]
            _, plan_t = plan_for({d: i for d, i in zip(nsdims, ns_index)}.get("t", 0))
            by_block = by_block_cache.setdefault(id(plan_t), {tuple(e["block_index"]): e for e in plan_t["per_chunk_entries"]})
            entry = by_block[bi]
            if not entry["views"]:
                return None      # the store's fill value (0) stands for blocks without contributing views
            kwargs, sl = chunk_call(ns_index, entry, device)
            chunk = np.asarray(fuse_np(**kwargs))
            if entry["fuse_planewise"]:
                chunk = chunk[np.newaxis]
            zarr_out.write(list(ns_index) + [s_.start for s_ in sl], chunk.reshape((1,) * len(ns_index) + chunk.shape))
            return None

        batch_func = batch_options.get("batch_func")
        n_batch = int(batch_options.get("n_batch", 1))
        batch_func_kwargs = dict(batch_options.get("batch_func_kwargs") or {})
        block_iter = iter(np.ndindex(*nblocks))
        while True:
            batch = [b for _, b in zip(range(n_batch), block_iter)]
            if not batch:
                break
            if batch_func is None:
                for block_id in batch:
                    fuse_chunk(block_id)
            else:
                batch_func(fuse_chunk, batch, **batch_func_kwargs)
    else:
      for ns_index in np.ndindex(*ns_shape) if ns_shape else [()]:
        ns_sel = {d: int(i) for d, i in zip(nsdims, ns_index)}
        _, plan = plan_for(ns_sel.get("t", 0))
        dev_out = None
        if on_device:
            dev_out = dev_full[tuple(int(i) for i in ns_index)] if n_fields > 1 else dev_full
            entries = plan["per_chunk_entries"]
            # one chunk that is actually fused over the whole array writes every voxel; in every other case (several chunks,
            # a chunk without views, a chunk the filter rejects) untouched voxels must read 0 like the host result
            single = (len(entries) == 1 and bool(entries[0]["views"])
                      and (chunk_filter is None or chunk_filter(entries[0]["block_index"])))
            if not single:
                dev_out.fill_zero()
        # tiles read from Zarr stores and / or a result that leaves the device block by block: read-ahead, asynchronous transfers
        # and write-behind around the launch blocks (streaming.BlockPipeline) -- the same fuse_np calls in the same order
        pipe = None
        if streamed and _STREAM_PIPELINE[0] and not on_device and fusion_func in _FUSION_CODES and (weights_func is None or weights_func is content_based) \
                and _lib.device_count() > 0:
            from .streaming import BlockPipeline

            pipe = BlockPipeline(fuse_np, device)
        try:
            for entry in plan["per_chunk_entries"]:
                bi = entry["block_index"]
                if chunk_filter is not None and not chunk_filter(bi):
                    continue
                if not entry["views"]:
                    continue
                kwargs, sl = chunk_call(ns_index, entry, device)
                if pipe is not None:
                    def sink(chunk, entry=entry, sl=sl, ns_index=ns_index):
                        if entry["fuse_planewise"]:
                            chunk = chunk[np.newaxis]
                        if zarr_out is not None:
                            from .streaming import write_region

                            write_region(zarr_out, list(ns_index) + [s_.start for s_ in sl], chunk.reshape((1,) * len(ns_index) + chunk.shape))
                        else:
                            from .streaming import parallel_copy

                            parallel_copy(result[tuple(ns_index) + sl], chunk, kind="write")
                    kwargs.pop("device", None)
                    tiling = None
                    if zarr_out is not None and not entry["fuse_planewise"] and _STREAM_TILES[0]:
                        tiling = (zarr_out, list(ns_index) + [s_.start for s_ in sl])
                    pipe.submit(dict(kwargs, device=device), sink, tiling)
                    continue
                if on_device and single:
                    # (a plane-wise entry is fused with 2D parameters: hand it the one plane of the 3D result)
                    fuse_np(out=dev_out[0] if entry["fuse_planewise"] else dev_out, **kwargs)
                elif on_device:
                    # chunked workflow with a device-resident mosaic: every chunk is fused on the device and copied into
                    # its window of the mosaic device-to-device (stream-ordered, no host round trip)
                    chunk = fuse_np(output_on_backend=True, **kwargs)
                    chunk.copy_into(dev_out, [s_.start for s_ in sl])
                else:
                    chunk = np.asarray(fuse_np(**kwargs))
                    if entry["fuse_planewise"]:
                        chunk = chunk[np.newaxis]
                    if zarr_out is not None:
                        zarr_out.write(list(ns_index) + [s_.start for s_ in sl], chunk.reshape((1,) * len(ns_index) + chunk.shape))
                    elif chunk.shape == result.shape:
                        result = chunk           # one launch block and one field: the fused array is the result
                    else:
                        result[tuple(ns_index) + sl] = chunk
        except BaseException:
            if pipe is not None:
                pipe.abort()      # (queued reads are dropped, the stage threads end; the error of the block that failed goes up)
            raise
        if pipe is not None:
            pipe.finish()
    if on_device:
        dev_full.mark_written()
        data = dev_full
        dims = (list(nsdims) if n_fields > 1 else []) + list(sdims)
        if record is not None and record.get("calls") == 1 and "views" in record and not record.get("no_replay") and not nsdims and n_fields == 1 \
                and tuple(dev_full.shape) == record["res_shape"]:
            # one launch block wrote the whole result: replayable.  Slab pointers are remembered as offsets into their tiles.
            base = np.array([images[iv].data.ptr for iv in record["view_index"]], dtype=np.uint64)      # (all device-resident: fuse_np recorded)
            entry = dict(record, byte_offsets=(record["ptrs"] - base).astype(np.uint64), dims=tuple(dims),
                         spacing=dict(output_stack_properties["spacing"]), origin=dict(output_stack_properties["origin"]),
                         dtype=np.dtype(images[0].dtype), ndim=len(sdims))
            with _REPLAY_LOCK:
                while len(_REPLAY_MEMO) >= _REPLAY_CAP:
                    _REPLAY_MEMO.pop(next(iter(_REPLAY_MEMO)), None)      # (oldest first: dicts keep insertion order)
                _REPLAY_MEMO[fast_key] = entry
    else:
        data = result if zarr_out is None else zarr_out[...]
        dims = list(nsdims) + list(sdims)
    res = si_utils.to_spatial_image(
        data, dims=dims, scale=output_stack_properties["spacing"], translation=output_stack_properties["origin"],
        c_coords=sims_[0].coords.get("c") if "c" in dims else None,
        t_coords=sims_[0].coords.get("t") if "t" in dims else None,
    )
    si_utils.set_sim_affine(res, param_utils.identity_transform(len(sdims)), transform_key)
    if zarr_out is not None and ome_zarr and chunk_filter is None:
        # pyramid levels + multiscales metadata around the level-0 array written above (_core.py:1160-1171)
        res = ngff_utils.write_sim_to_ome_zarr(res, output_zarr_url, overwrite=False, ngff_version=ngff_version,
                                               zarr_array_creation_kwargs=zarr_options.get("zarr_array_creation_kwargs"),
                                               device=device)
    return res


def fuse(*args, **kwargs):
    import inspect
    import traceback
    import warnings

    bound = inspect.signature(_fuse_once).bind(*args, **kwargs)      # wherever merge_chunks was passed, it can be overridden
    failure = None
    try:
        res = _fuse_once(*bound.args, **bound.kwargs)
        if bound.arguments.get("weights_func") is content_based and bound.arguments.get("output_on_backend"):
            # chunks left on the device are not waited for one by one; a mask list of the fast content-based path that overflowed
            # in any of them shows here, and the mosaic is fused again through the bit-faithful passes
            dev_ = bound.arguments.get("device", 0)
            if _cb_overflowed(dev_):
                with _cb_exact(dev_):
                    res = _fuse_once(*bound.args, **bound.kwargs)
        return res
    except _lib.DeviceMemoryError as exc:
        # A merged launch block is sized from an ESTIMATE of what has to be staged on the device (_launch_budget); if the
        # device still runs out of memory the requested chunk grid -- the unit the caller sized for -- is used instead.
        if not bound.arguments.get("merge_chunks", True):
            raise
        failure = str(exc)
        # the retry must not run inside this handler: the traceback keeps the failed attempt's frames -- and with them the
        # mosaic-sized device buffer, staged peer copies and keep lists -- alive on a device that has just run out of memory
        traceback.clear_frames(exc.__traceback__)
    import gc

    gc.collect()
    warnings.warn(f"fuse(): a merged launch block did not fit the device ({failure}); falling back to the requested "
                  "output_chunksize", RuntimeWarning, stacklevel=2)
    bound.arguments["merge_chunks"] = False
    return _fuse_once(*bound.args, **bound.kwargs)


fuse.__doc__ = _fuse_once.__doc__
fuse.__wrapped__ = _fuse_once


def fuse_to_host(images, transform_key=None, n_slabs=8, out=None, device=0, return_timeline=False, **fuse_kwargs):
    """``fuse()`` of views resident on (or on their way to) the device with the RESULT in host memory, the download overlapped with
    the kernels: the output stack is cut into ``n_slabs`` slabs along its first axis, every slab is fused in one launch block
    (``output_on_backend=True``) and copied into its window of a pinned host array on the device's copy stream while the next
    slab is fused (mvs_mark on the fuse lane -> mvs_copy_async, csrc/mvs_transfer.hip).  The reference streams fused chunks out of a
    dask graph into host memory / a Zarr store (_core.py:1068-1170, 2044-2156); SURVEY 8d(2)'s end-to-end figure includes this
    D2H.  Every slab is fused in the index frame of the WHOLE stack (``frame_origin``), so the result equals ``fuse()`` of the
    whole stack voxel for voxel.  ``out``: a pinned C-contiguous array of the result's shape (``device.pinned_empty``) to fill
    instead of a new one.  ``return_timeline``: also return [(fuse done, download done)] per slab in ms since the first slab's
    launch block was queued (timed tickets) -- the overlap test reads it.  Remaining keyword arguments: those of ``fuse()``
    (single field images: spatial dims only)."""
    from . import device as dev_mod

    for bad in ("output_zarr_url", "batch_options", "output_on_backend", "chunk_filter", "sims"):
        if fuse_kwargs.get(bad):
            raise TypeError(f"fuse_to_host does not take {bad}")
    images = list(images)
    sdims = si_utils.get_spatial_dims_from_sim(images[0])
    if list(images[0].dims) != list(sdims):
        raise ValueError("fuse_to_host fuses single fields (spatial dims only)")
    osp = _bb_dicts(process_output_stack_properties(
        images, fuse_kwargs.pop("output_spacing", None), fuse_kwargs.pop("output_origin", None), fuse_kwargs.pop("output_shape", None),
        fuse_kwargs.pop("output_stack_properties", None), fuse_kwargs.pop("output_stack_mode", "union"), transform_key), sdims)
    shape = tuple(int(osp["shape"][d]) for d in sdims)
    dtype = np.dtype(images[0].dtype)
    if out is None:
        out = dev_mod.pinned_empty(shape, dtype)
    elif tuple(out.shape) != shape or out.dtype != dtype or not out.flags.c_contiguous or not dev_mod.is_pinned(out):
        raise ValueError(f"out must be a pinned C-contiguous {dtype} array of shape {shape}")
    d0 = sdims[0]
    n0 = shape[0]
    n_slabs = max(1, min(int(n_slabs), n0))
    cuts = np.linspace(0, n0, n_slabs + 1).round().astype(int)
    fuse_kwargs.setdefault("frame_origin", dict(osp["origin"]))
    fuse_kwargs.setdefault("output_chunksize", {d: 1 << 30 for d in sdims})
    t_start = dev_mod.mark(device)
    h_start = time.perf_counter()
    pending, timeline, host_ms = [], [], []      # host_ms: (fuse() of the slab entered, returned) on the host clock, ms
    # The class kernels of a slab's launch stay on the lane's own stream (option serial_classes) while downloads are in flight: forked
    # onto the LOW-priority side streams they were served only when the HIGH-priority copy stream's queue had drained -- slab k + 1 was
    # fused when slab k's download was through, in most runs for some slabs, in one run of ten for all of them, once 377 ms late
    # (A/B on one box, profiles/round6_summary.md 4).  A slab is 1/8 of a mosaic; forking saves it ~0.1 ms.  (The other half of the
    # same symptom: parameter blocks uploaded by a copy engine queued behind the downloads -- csrc/mvs_context.hip: mvs_upload_small.)
    serial_slabs = True
    if serial_slabs:
        _lib.set_option("serial_classes", 1, device)
    try:
        return _fuse_to_host_slabs(images, transform_key, n_slabs, cuts, osp, d0, out, device, return_timeline, fuse_kwargs, sdims, t_start, h_start,
                                   pending, timeline, host_ms)
    finally:
        if serial_slabs:
            _lib.set_option("serial_classes", 0, device)


def _fuse_to_host_slabs(images, transform_key, n_slabs, cuts, osp, d0, out, device, return_timeline, fuse_kwargs, sdims, t_start, h_start, pending,
                        timeline, host_ms):
    from . import device as dev_mod

    for k in range(n_slabs):
        a, b = int(cuts[k]), int(cuts[k + 1])
        if b <= a:
            continue
        sub = {"origin": dict(osp["origin"], **{d0: osp["origin"][d0] + a * osp["spacing"][d0]}), "spacing": dict(osp["spacing"]),
               "shape": dict(osp["shape"], **{d0: b - a})}
        h0 = time.perf_counter()
        fused = fuse(images, transform_key=transform_key, output_stack_properties=sub, output_on_backend=True, device=device, **fuse_kwargs)
        host_ms.append(((h0 - h_start) * 1e3, (time.perf_counter() - h_start) * 1e3))
        t_fused = dev_mod.mark(device)
        t_down = fused.data.download_async(out[a:b], after=t_fused)
        pending.append((fused, t_fused, t_down))      # (the slab stays alive until its download has passed)
    for fused, t_fused, t_down in pending:
        dev_mod.ticket_sync(t_down)
        if return_timeline:
            timeline.append((dev_mod.ticket_elapsed_ms(t_start, t_fused), dev_mod.ticket_elapsed_ms(t_start, t_down)))
    res = si_utils.to_spatial_image(out, dims=list(sdims), scale=osp["spacing"], translation=osp["origin"])
    si_utils.set_sim_affine(res, param_utils.identity_transform(len(sdims)), transform_key)
    fuse_to_host.last_host_ms = host_ms          # (measurement: where the host was while the slabs were queued)
    return (res, timeline) if return_timeline else res
