"""CPU oracle: chunk-wise fusion (affine resample + blend weights + fuse).

TEST INFRASTRUCTURE (see oracle/__init__.py).  numpy/scipy restatement of

  transformation.transform_sim          src/multiview_stitcher/transformation.py:15-148
  weights.get_blending_weights          src/multiview_stitcher/weights.py:391-511
  weights.normalize_weights             src/multiview_stitcher/weights.py:325-345
  weights.nan_gaussian_filter           src/multiview_stitcher/weights.py:293-322
  weights.content_based                 src/multiview_stitcher/weights.py:22-74
  fusion.weighted_average_fusion        src/multiview_stitcher/fusion/_core.py:61-94
  fusion.max_fusion / simple_average    src/multiview_stitcher/fusion/_core.py:42-58, 97-131
  fusion.fuse_np                        src/multiview_stitcher/fusion/_core.py:1513-1733

Data model (no xarray here): a bounding box / stack-properties dict is
``{"origin": (ndim,) f64, "spacing": (ndim,) f64, "shape": (ndim,) int}`` in
(z,)y,x order; a view is ``{"data": ndarray, "origin": ..., "spacing": ...}``;
an affine is a homogeneous (ndim+1, ndim+1) float64 matrix (view -> world).
"""

from __future__ import annotations

import warnings

import numpy as np
from scipy.ndimage import (
    affine_transform,
    distance_transform_edt,
    gaussian_filter,
)


def bb(origin, spacing, shape):
    return {
        "origin": np.asarray(origin, dtype=np.float64),
        "spacing": np.asarray(spacing, dtype=np.float64),
        "shape": np.asarray(shape, dtype=np.int64),
    }


def coords_origin_spacing(origin, spacing, shape):
    """What the reference reads back from a SpatialImage's coordinate arrays.

    spatial_image_utils._get_axis_coords (spatial_image_utils.py:316-317) builds
    ``coords = translation + scale * arange(n)``; get_origin_from_sim
    (:554-561) returns coords[0] and get_spacing_from_sim (:574-589) returns
    ``coords[1] - coords[0]`` (1.0 for a singleton axis).  The spacing therefore
    carries one float rounding that we reproduce.
    """
    origin = np.asarray(origin, dtype=np.float64)
    spacing = np.asarray(spacing, dtype=np.float64)
    c0 = origin + spacing * 0.0
    c1 = origin + spacing * 1.0
    derived = np.where(np.asarray(shape) > 1, c1 - c0, 1.0)
    return c0, derived


def transform_params(p, in_origin, in_spacing, out_bb):
    """Pixel-space (matrix', offset') of transformation.py:37-83."""
    p = np.asarray(p, dtype=np.float64)
    ndim = p.shape[0] - 1
    matrix = p[:ndim, :ndim]
    offset = p[:ndim, ndim]
    Sx = np.diag(np.asarray(out_bb["spacing"], dtype=np.float64))
    Sy = np.diag(np.asarray(in_spacing, dtype=np.float64))
    Ox = np.asarray(out_bb["origin"], dtype=np.float64)
    Oy = np.asarray(in_origin, dtype=np.float64)

    matrix_prime = np.linalg.solve(Sy, np.dot(matrix, Sx))
    local_input_origin = Oy - Ox
    local_offset = offset + np.dot(matrix - np.eye(ndim), Ox)
    offset_prime = np.linalg.solve(Sy, local_offset - local_input_origin)

    matrix_prime = np.around(matrix_prime, decimals=10)
    offset_prime = np.around(offset_prime, decimals=10)
    nearest_integer = np.round(offset_prime)
    near_integer = np.isclose(offset_prime, nearest_integer, rtol=0, atol=1e-6)
    offset_prime[near_integer] = nearest_integer[near_integer]
    return matrix_prime, offset_prime


def transform_array(
    data, p, in_origin, in_spacing, out_bb, order=1, cval=0.0
):
    """transformation.transform_sim for numpy-backed data (transformation.py:85-139)."""
    ndim = data.ndim
    if p is None:
        p = np.eye(ndim + 1)
    matrix_prime, offset_prime = transform_params(p, in_origin, in_spacing, out_bb)
    out_shape = tuple(int(s) for s in out_bb["shape"])
    is_noop = (
        out_shape == tuple(data.shape)
        and np.allclose(matrix_prime, np.eye(ndim), rtol=0, atol=1e-10)
        and np.allclose(offset_prime, 0, rtol=0, atol=1e-10)
    )
    if is_noop:
        return data
    return affine_transform(
        data,
        matrix=matrix_prime,
        offset=offset_prime,
        output_shape=out_shape,
        mode="constant",
        cval=cval,
        order=order,
    )


DEFAULT_BLENDING_WIDTHS = {"z": 3.0, "y": 10.0, "x": 10.0}


def _widths_array(blending_widths, ndim):
    if blending_widths is None:
        blending_widths = DEFAULT_BLENDING_WIDTHS
    if isinstance(blending_widths, dict):
        return np.array(
            [blending_widths[d] for d in ["z", "y", "x"][-ndim:]], dtype=np.float64
        )
    return np.asarray(blending_widths, dtype=np.float64)


def shrink_source_bb(source_bb, shrink_distance):
    """weights._shrink_source_bb (weights.py:348-388)."""
    ndim = len(source_bb["origin"])
    sd = np.broadcast_to(np.asarray(shrink_distance, dtype=np.float64), (ndim,))
    return {
        "origin": source_bb["origin"] + sd,
        "spacing": source_bb["spacing"],
        "shape": source_bb["shape"] - 2 * sd / source_bb["spacing"],
    }


def edt_support(source_bb, blending_widths=None, shrink_distance=0):
    """The 5^ndim EDT support image of weights.py:430-470.

    Returns (table f64, support_origin, support_spacing)."""
    ndim = len(source_bb["origin"])
    bw = _widths_array(blending_widths, ndim)
    if np.any(np.asarray(shrink_distance) != 0):
        source_bb = shrink_source_bb(source_bb, shrink_distance)
    shape = np.asarray(source_bb["shape"], dtype=np.float64)
    spacing = np.asarray(source_bb["spacing"], dtype=np.float64)
    origin = np.asarray(source_bb["origin"], dtype=np.float64)

    mask = np.zeros([3 + 2] * ndim)
    mask[(slice(1, -1),) * ndim] = 1
    support_spacing = (shape - 1) / 4 * spacing
    edt_support_spacing = support_spacing * (shape - 1 + 2 * 1) / (shape - 1)
    edt_support_origin = origin - 1 * spacing
    table = distance_transform_edt(mask, sampling=list(edt_support_spacing / bw))
    return table, edt_support_origin, edt_support_spacing


def cosine_weights(x):
    """weights.py:502-507 (operates in the array's own dtype, float32)."""
    mask = x < 1
    x[mask] = (np.cos((1 - x[mask]) * np.pi) + 1) / 2
    x = np.clip(x, 0, 1)
    return x


def get_blending_weights(
    target_bb, source_bb, affine, blending_widths=None, shrink_distance=0
):
    """weights.get_blending_weights (weights.py:391-511)."""
    table, sup_origin, sup_spacing = edt_support(
        source_bb, blending_widths, shrink_distance
    )
    # the support is wrapped in a SpatialImage and its origin/spacing are read
    # back from the coordinate arrays (weights.py:465-470, transformation.py:45-51)
    o, s = coords_origin_spacing(sup_origin, sup_spacing, table.shape)
    target_weights = transform_array(
        table.astype(np.float32),
        np.linalg.inv(affine),
        o,
        s,
        target_bb,
        order=1,
        cval=0.0,
    )
    if target_weights.dtype != np.float32:  # pragma: no cover
        target_weights = target_weights.astype(np.float32)
    target_weights = np.array(target_weights, copy=True)
    return cosine_weights(target_weights)


def normalize_weights(weights):
    """weights.normalize_weights (weights.py:325-345)."""
    wsum = np.nansum(weights, axis=0)
    wsum[wsum == 0] = 1
    return weights / wsum


def nan_gaussian_filter(ar, *args, **kwargs):
    """weights.nan_gaussian_filter (weights.py:293-322)."""
    U = ar
    nan_mask = np.isnan(U)
    V = U.copy()
    V[nan_mask] = 0
    VV = gaussian_filter(V, *args, **kwargs)
    W = 0 * U.copy() + 1
    W[nan_mask] = 0
    WW = gaussian_filter(W, *args, **kwargs)
    WW[nan_mask] = 1
    Z = VV / WW
    Z[nan_mask] = np.nan
    return Z


def content_based(transformed_views, blending_weights, sigma_1=5, sigma_2=11):
    """weights.content_based (weights.py:22-74); required_overlap = 2*sigma_2."""
    transformed_views = transformed_views.astype(np.float32)
    transformed_views[blending_weights < 1e-7] = np.nan
    weights = [
        nan_gaussian_filter(
            (sim_t - nan_gaussian_filter(sim_t, sigma=sigma_1, mode="reflect")) ** 2,
            sigma=sigma_2,
            mode="reflect",
        )
        for sim_t in transformed_views
    ]
    weights = np.stack(weights, axis=0)
    return normalize_weights(weights)


def weighted_average_fusion(transformed_views, blending_weights, fusion_weights=None):
    """fusion.weighted_average_fusion (_core.py:61-94)."""
    if fusion_weights is None:
        additive_weights = blending_weights
    else:
        additive_weights = blending_weights * fusion_weights
        additive_weights = normalize_weights(additive_weights)
    product = transformed_views * additive_weights
    return np.nansum(product, axis=0).astype(transformed_views[0].dtype)


def max_fusion(transformed_views):
    """fusion.max_fusion (_core.py:42-58)."""
    return np.nanmax(transformed_views, axis=0)


def simple_average_fusion(transformed_views):
    """fusion.simple_average_fusion (_core.py:97-131)."""
    number_of_valid_views = np.zeros(transformed_views[0].shape, dtype=np.float32)
    for tv in transformed_views:
        number_of_valid_views = np.nansum(
            [number_of_valid_views, ~np.isnan(tv)], axis=0
        )
    number_of_valid_views[number_of_valid_views == 0] = np.nan
    return (np.nansum(transformed_views, axis=0) / number_of_valid_views).astype(
        transformed_views[0].dtype
    )


FUSION_FUNCS = {
    "weighted_average": weighted_average_fusion,
    "max": max_fusion,
    "simple_average": simple_average_fusion,
}


def fuse_np(
    views,
    params,
    output_properties,
    fusion="weighted_average",
    weights=None,
    weights_kwargs=None,
    trim_overlap_in_pixels=0,
    interpolation_order=1,
    full_view_bbs=None,
    blending_widths=None,
    shrink_distance=0,
    return_float=False,
    return_debug=False,
):
    """fusion.fuse_np (_core.py:1513-1733) for the built-in fusion/weight funcs.

    views[i] = {"data", "origin", "spacing"} is the in-memory slab handed to the
    chunk task; full_view_bbs[i] is the bounding box of the whole view (used for
    the blending weights and for the spacing, _core.py:1611-1619).
    ``return_float=True`` additionally returns the float32 array before the
    final ``nan_to_num().astype(input dtype)`` (_core.py:1713).
    """
    ndim = views[0]["data"].ndim
    input_dtype = views[0]["data"].dtype
    if full_view_bbs is None:
        full_view_bbs = [
            bb(v["origin"], v["spacing"], v["data"].shape) for v in views
        ]
    spacings = [fvb["spacing"] for fvb in full_view_bbs]

    field_ims_t = np.stack(
        [
            transform_array(
                v["data"].astype(np.float32),
                np.linalg.inv(param),
                v["origin"],
                spacing,
                output_properties,
                order=interpolation_order,
                cval=np.nan,
            )
            for v, param, spacing in zip(views, params, spacings)
        ]
    )

    needs_blending = fusion == "weighted_average" or weights == "content_based"
    if needs_blending:
        field_ws_t = np.stack(
            [
                get_blending_weights(
                    output_properties,
                    full_view_bbs[iv],
                    params[iv],
                    blending_widths=blending_widths,
                    shrink_distance=shrink_distance,
                )
                for iv in range(len(views))
            ]
        )
        field_ws_t = field_ws_t * ~np.isnan(field_ims_t)
        raw_ws_t = field_ws_t
        field_ws_t = normalize_weights(field_ws_t)
    else:
        field_ws_t = None
        raw_ws_t = None

    with warnings.catch_warnings():
        warnings.simplefilter("ignore", category=RuntimeWarning)
        if fusion == "weighted_average":
            fusion_weights = None
            if weights == "content_based":
                fusion_weights = content_based(
                    field_ims_t, field_ws_t, **(weights_kwargs or {})
                )
            fused = weighted_average_fusion(field_ims_t, field_ws_t, fusion_weights)
        else:
            fused = FUSION_FUNCS[fusion](field_ims_t)

    trim = np.broadcast_to(np.asarray(trim_overlap_in_pixels, dtype=np.int64), (ndim,))
    if np.any(trim > 0):
        fused = fused[
            tuple(slice(t, -t) if t > 0 else slice(None) for t in trim)
        ]
    fused_f = np.nan_to_num(fused)
    out = fused_f.astype(input_dtype)
    if return_debug:
        # float32 result plus what a test needs to bound the reference's OWN rounding noise:
        # the masked, un-normalised blending weights and the resampled views (untrimmed)
        return out, fused_f, {"raw_weights": raw_ws_t, "views": field_ims_t, "trim": trim}
    if return_float:
        return out, fused_f
    return out
