"""Multiscale spatial images for the register+fuse path (mirror of msi_utils call shapes).

Reference: src/multiview_stitcher/msi_utils.py (xr.DataTree of ``scaleN/image``
plus transforms as data variables, msi_utils.py:351-430, 596-617).  Here a
MultiscaleSpatialImage is a small container {"scaleN": SpatialImage} with one
transforms dict shared by all scales; pyramid levels beyond scale0 are built by
block-mean downsampling on request (msi_utils.py:49-79).
"""

from __future__ import annotations

import copy

import numpy as np

from . import param_utils
from . import spatial_image_utils as si_utils


class MultiscaleSpatialImage:
    def __init__(self, sims, transforms=None):
        self.scales = {f"scale{i}": s for i, s in enumerate(sims)}
        self.transforms = transforms if transforms is not None else {}

    def keys(self):
        return list(self.scales.keys())

    def __getitem__(self, key):
        if key.endswith("/image"):
            key = key[: -len("/image")]
        return self.scales[key]

    def __repr__(self):
        return f"<MultiscaleSpatialImage scales={self.keys()} transforms={list(self.transforms)}>"


def is_msim(obj):
    return isinstance(obj, MultiscaleSpatialImage)


def get_sorted_scale_keys(msim):
    return sorted(msim.keys(), key=lambda k: int(k.split("scale")[-1]))


def get_ndim(msim):
    return si_utils.get_ndim_from_sim(msim["scale0/image"])


def _downsample_sim(sim, factors):
    """Block-mean downsample along the spatial dims (msi_utils.py:49-79), cast back to dtype."""
    sdims = si_utils.get_spatial_dims_from_sim(sim)
    data = np.asarray(sim.data)
    sl = []
    for d in sim.dims:
        f = int(factors.get(d, 1)) if d in sdims else 1
        n = (data.shape[sim.dims.index(d)] // f) * f
        sl.append(slice(0, n))
    data = data[tuple(sl)]
    new_shape = []
    axes = []
    for ax, d in enumerate(sim.dims):
        f = int(factors.get(d, 1)) if d in sdims else 1
        new_shape += [data.shape[ax] // f, f]
        axes.append(2 * ax + 1)
    out = data.reshape(new_shape).mean(axis=tuple(axes)).astype(sim.dtype)
    sp, o = si_utils.get_spacing_from_sim(sim), si_utils.get_origin_from_sim(sim)
    scale = {d: sp[d] * factors.get(d, 1) for d in sdims}
    trans = {d: o[d] + (factors.get(d, 1) - 1) * sp[d] / 2 for d in sdims}
    res = si_utils.to_spatial_image(out, sim.dims, scale, trans, sim.coords.get("c"), sim.coords.get("t"))
    res.attrs["transforms"] = dict(sim.attrs.get("transforms", {}))
    return res


def calc_resolution_levels(spatial_shape, downscale_factors_per_spatial_dim=None, min_shape=100):
    """Shapes, relative and absolute factors of the pyramid levels, level 0 included (msi_utils.py:279-325): a dim is
    halved (or divided by its factor) as long as the result stays above ``min_shape``."""
    sdims = list(spatial_shape.keys())
    if downscale_factors_per_spatial_dim is None:
        downscale_factors_per_spatial_dim = {d: 2 for d in sdims}
    shapes, rel, ab = [dict(spatial_shape)], [{d: 1 for d in sdims}], [{d: 1 for d in sdims}]
    while True:
        f = {d: downscale_factors_per_spatial_dim[d] if shapes[-1][d] // downscale_factors_per_spatial_dim[d] > min_shape else 1
             for d in sdims}
        if not any(v > 1 for v in f.values()):
            return shapes, rel, ab
        ab.append({d: ab[-1][d] * f[d] for d in sdims})
        shapes.append({d: shapes[-1][d] // f[d] for d in sdims})
        rel.append(f)


def get_msim_from_sim(sim, scale_factors=None, chunks=None):
    """msi_utils.get_msim_from_sim (msi_utils.py:373-430); default: scale0 only."""
    sims = [sim]
    for sf in scale_factors or []:
        if not isinstance(sf, dict):
            sf = {d: sf for d in si_utils.get_spatial_dims_from_sim(sim)}
        sims.append(_downsample_sim(sims[-1], sf))
    return MultiscaleSpatialImage(sims, copy.deepcopy(sim.attrs.get("transforms", {})))


def get_sim_from_msim(msim, scale="scale0"):
    """msi_utils.get_sim_from_msim (msi_utils.py:351-370): the sim at ``scale`` with all transforms."""
    sim = msim[scale].copy()
    sim.attrs["transforms"] = dict(msim.transforms)
    return sim


def get_res_level_from_spacing(msim, spacing):
    """msi_utils.get_res_level_from_spacing (msi_utils.py:655-685): index of the coarsest level whose spacing is still <= the
    target spacing in every requested dimension (levels are scanned from scale0 and the scan stops at the first that is not)."""
    best = 0
    for i, key in enumerate(get_sorted_scale_keys(msim)):
        actual = si_utils.get_spacing_from_sim(msim[key])
        if all(actual[d] <= spacing[d] for d in spacing):
            best = i
        else:
            break
    return best


def get_msim_from_sims(sims):
    """msi_utils.get_msim_from_sims (msi_utils.py:433-480): a multiscale image from already computed levels -- ordered by
    decreasing spatial shape (which must be comparable), transforms taken from the finest one."""
    sims = list(sims)
    if not sims:
        raise ValueError("sims must contain at least one image.")
    dims = list(sims[0].dims)
    if any(list(s.dims) != dims for s in sims[1:]):
        raise ValueError("All sims must have the same dimensions.")
    sdims = si_utils.get_spatial_dims_from_sim(sims[0])
    sims = sorted(sims, key=lambda s: tuple(s.sizes[d] for d in sdims), reverse=True)
    for a, b in zip(sims, sims[1:]):
        if not all(b.sizes[d] <= a.sizes[d] for d in sdims):
            raise ValueError("sims cannot be ordered into resolution levels (shapes are not comparable).")
    # copies (shallow: the voxels are shared), so that the caller's images keep their attrs; the lower levels inherit exactly
    # the transform keys of scale0 (msi_utils.py:466-480)
    sims = [s.copy() for s in sims]
    t0 = sims[0].attrs.get("transforms", {})
    for s in sims[1:]:
        s.attrs["transforms"] = {k: np.array(v, dtype=np.float64, copy=True) for k, v in t0.items()}
    return MultiscaleSpatialImage(sims, copy.deepcopy(t0))


def get_res_level_from_binning_factors(msim, binning_factors):
    """msi_utils.get_res_level_from_binning_factors (msi_utils.py:688-773): the LOWEST resolution level whose integer
    downsampling factors (shape of scale0 / shape of the level, per spatial dim) do not exceed the requested binning and
    divide it; returns ``(scale_key, remaining_binning)`` -- scale0 with the full binning when no other level qualifies."""
    sim0 = msim["scale0"]
    sdims = si_utils.get_spatial_dims_from_sim(sim0)
    shape0 = {d: sim0.sizes[d] for d in sdims}
    best_scale, best_remaining = "scale0", dict(binning_factors)
    for scale_key in get_sorted_scale_keys(msim):
        sim = msim[scale_key]
        factors = {d: shape0[d] / sim.sizes[d] for d in sdims}
        valid = True
        for d in sdims:
            if d not in binning_factors:
                continue
            if not np.isclose(factors[d], round(factors[d]), rtol=1e-6):
                valid = False
                break
            f = int(round(factors[d]))
            if f > binning_factors[d] or binning_factors[d] % f != 0:
                valid = False
                break
        if valid:
            best_scale = scale_key
            best_remaining = {d: (binning_factors[d] // int(round(factors[d])) if d in binning_factors else 1) for d in sdims}
    return best_scale, best_remaining


def get_transform_from_msim(msim, transform_key=None):
    return msim.transforms[transform_key]


def get_transforms_from_dataset_as_dict(msim):
    return dict(msim.transforms)


def set_affine_transform(msim, xaffine=None, transform_key=None, base_transform_key=None):
    """msi_utils.set_affine_transform (msi_utils.py:596-617)."""
    if transform_key is None:
        raise ValueError("transform_key must be provided")
    if xaffine is None:
        xaffine = param_utils.identity_transform(get_ndim(msim))
    xaffine = np.asarray(xaffine, dtype=np.float64)
    if base_transform_key is not None:
        xaffine = param_utils.rebase_affine(xaffine, get_transform_from_msim(msim, base_transform_key))
    msim.transforms[transform_key] = xaffine
    for s in msim.scales.values():
        s.attrs.setdefault("transforms", {})[transform_key] = xaffine
