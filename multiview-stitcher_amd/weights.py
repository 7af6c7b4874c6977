"""Blending / fusion weights on the HIP backend (mirror of the reference's weights.py).

Host side: the 5^ndim blending support (weights.py:430-470) and its pixel-space
affine; device side: resample + cosine ramp fused into ``mvs_fuse_chunk``
(csrc/mvs_fuse.hip), or as a standalone volume through ``mvs_blend_weights``.
"""

from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib
from .transformation import _as_zyx, embed3, get_pixel_affine, shape3

DEFAULT_BLENDING_WIDTHS = {"z": 3, "y": 10, "x": 10}


def content_based_required_overlap(sigma_2=11, **_):
    """``@requires_overlap(lambda kwargs: 2 * kwargs["sigma_2"])`` of weights.py:22."""
    return 2 * sigma_2


def _shrink_source_bb(origin, spacing, shape, shrink_distance, sdims):
    """weights._shrink_source_bb (weights.py:348-388)."""
    if isinstance(shrink_distance, dict):
        sd = np.array([float(shrink_distance.get(d, 0)) for d in sdims])
    else:
        sd = np.full(len(sdims), float(shrink_distance))
    return origin + sd, spacing, shape - 2 * sd / spacing


_TABLE_CACHE = {}


def blending_support(source_bb, blending_widths=None, shrink_distance=0):
    """The EDT support image of get_blending_weights (weights.py:430-470).

    Returns (table float32 (5,)*ndim, support origin, support spacing as read
    back from the SpatialImage coordinates).  The table is the closed form of
    ``distance_transform_edt(mask, sampling)`` for the fixed 5^n mask whose inner
    3^n block is foreground: the nearest background voxel of an inner voxel lies
    straight along one axis, so  edt[i] = min_d( min(i_d, 4 - i_d) * sampling_d ).
    """
    sdims = sorted(source_bb["origin"].keys())[::-1] if isinstance(source_bb["origin"], dict) else ["z", "y", "x"][-len(source_bb["origin"]):]
    ndim = len(sdims)
    if blending_widths is None:
        blending_widths = DEFAULT_BLENDING_WIDTHS
    bw = _as_zyx(blending_widths, sdims)
    origin = _as_zyx(source_bb["origin"], sdims)
    spacing = _as_zyx(source_bb["spacing"], sdims)
    shape = _as_zyx(source_bb["shape"], sdims)
    if (isinstance(shrink_distance, dict) and any(shrink_distance.values())) or (
        not isinstance(shrink_distance, dict) and shrink_distance
    ):
        origin, spacing, shape = _shrink_source_bb(origin, spacing, shape, shrink_distance, sdims)
    support_spacing = (shape - 1) / 4 * spacing
    edt_support_spacing = support_spacing * (shape - 1 + 2 * 1) / (shape - 1)
    edt_support_origin = origin - 1 * spacing
    sampling = edt_support_spacing / bw
    key = tuple(sampling.tolist())
    table = _TABLE_CACHE.get(key)
    if table is None:      # all tiles of a mosaic share one table
        tent = np.minimum(np.arange(5), 4 - np.arange(5)).astype(np.float64)
        per_axis = [tent * sampling[d] for d in range(ndim)]
        grids = np.meshgrid(*per_axis, indexing="ij")
        table = np.minimum.reduce(grids).astype(np.float32)
        if len(_TABLE_CACHE) > 64:
            _TABLE_CACHE.clear()
        _TABLE_CACHE[key] = table
    # origin/spacing as the reference reads them back from the coordinate arrays
    # (spatial_image_utils.py:316-317, 554-589): coords = o + s*arange(5)
    c0 = edt_support_origin + edt_support_spacing * 0.0
    c1 = edt_support_origin + edt_support_spacing * 1.0
    return table, c0, c1 - c0


def blending_supports(origins, spacings, shapes, sdims, blending_widths=None, shrink_distance=0):
    """``blending_support`` for a stack of views ((n, ndim) arrays in ``sdims`` order): (tables, support origins, support
    spacings); views with the same sampling share one table object."""
    ndim = len(sdims)
    if blending_widths is None:
        blending_widths = DEFAULT_BLENDING_WIDTHS
    bw = _as_zyx(blending_widths, sdims)
    origin, spacing, shape = (np.asarray(a, dtype=np.float64) for a in (origins, spacings, shapes))
    if (isinstance(shrink_distance, dict) and any(shrink_distance.values())) or (
        not isinstance(shrink_distance, dict) and shrink_distance
    ):
        origin, spacing, shape = _shrink_source_bb(origin, spacing, shape, shrink_distance, sdims)
    support_spacing = (shape - 1) / 4 * spacing
    edt_support_spacing = support_spacing * (shape - 1 + 2 * 1) / (shape - 1)
    edt_support_origin = origin - 1 * spacing
    sampling = edt_support_spacing / bw
    tables = []
    for row in sampling:
        key = tuple(row.tolist())
        table = _TABLE_CACHE.get(key)
        if table is None:
            tent = np.minimum(np.arange(5), 4 - np.arange(5)).astype(np.float64)
            grids = np.meshgrid(*[tent * row[d] for d in range(ndim)], indexing="ij")
            table = np.minimum.reduce(grids).astype(np.float32)
            if len(_TABLE_CACHE) > 64:
                _TABLE_CACHE.clear()
            _TABLE_CACHE[key] = table
        tables.append(table)
    c0 = edt_support_origin + edt_support_spacing * 0.0
    c1 = edt_support_origin + edt_support_spacing * 1.0
    return tables, c0, c1 - c0


def fill_view_weights(view, source_bb, affine, target_origin, target_spacing, blending_widths=None, shrink_distance=0):
    """Fill the blending half of an ``mvs_view_t`` (w_matrix, w_offset, edt)."""
    table, sup_origin, sup_spacing = blending_support(source_bb, blending_widths, shrink_distance)
    ndim = table.ndim
    wm, wo = get_pixel_affine(np.linalg.inv(np.asarray(affine, dtype=np.float64)), sup_origin, sup_spacing,
                              target_origin, target_spacing)
    m3, o3 = embed3(wm, wo)
    view.w_matrix[:] = m3.reshape(-1).tolist()
    view.w_offset[:] = o3.tolist()
    flat = np.zeros(125, dtype=np.float32)
    flat[: table.size] = table.reshape(-1)
    C.memmove(view.edt, flat.ctypes.data, 500)
    return ndim


def get_blending_weights(target_bb, source_bb, affine, blending_widths=None, shrink_distance=0, device=0):
    """weights.get_blending_weights (weights.py:391-511): float32 volume on the target grid."""
    lib = _lib.init(device)
    sdims = sorted(source_bb["origin"].keys())[::-1] if isinstance(source_bb["origin"], dict) else ["z", "y", "x"][-len(source_bb["origin"]):]
    view = _lib.mvs_view_t()
    ndim = fill_view_weights(view, source_bb, affine, _as_zyx(target_bb["origin"], sdims),
                             _as_zyx(target_bb["spacing"], sdims), blending_widths, shrink_distance)
    out_shape = [int(v) for v in _as_zyx(target_bb["shape"], sdims)]
    out = np.empty(tuple(out_shape), dtype=np.float32)
    rc = lib.mvs_blend_weights(device, C.byref(view), ndim, _lib.i64x3(shape3(out_shape)), out.ctypes.data,
                               _lib.MVS_MEM_HOST)
    _lib.check(rc, device, "mvs_blend_weights")
    return out
