"""Numpy-backed SpatialImage and the helpers the register+fuse path uses.

Mirror of the reference's ``spatial_image_utils`` call shapes
(src/multiview_stitcher/spatial_image_utils.py) without xarray/dask/zarr, none
of which exist on the MI355X box.  A SpatialImage is a labelled array:

    sim.data    ndarray, dims ordered as a subset of (c, t, z, y, x)
    sim.dims    tuple of dim names
    sim.coords  {dim: 1-D coordinate array}; spatial coords = origin + spacing*i
                (spatial_image_utils.py:316-317)
    sim.attrs["transforms"][key] = affine ([t,] ndim+1, ndim+1) float64
                (spatial_image_utils.py:951-959, 1234-1245)
"""

from __future__ import annotations

import copy as _copy

import numpy as np

from . import param_utils

SPATIAL_DIMS = ["z", "y", "x"]
SPATIAL_IMAGE_DIMS = ["c", "t", "z", "y", "x"]
DEFAULT_TRANSFORM_KEY = "affine_metadata"
DEFAULT_SPATIAL_CHUNKSIZES_3D = {"z": 256, "y": 256, "x": 256}
DEFAULT_SPATIAL_CHUNKSIZES_2D = {"y": 1024, "x": 1024}


class SpatialImage:
    """Minimal labelled array (the part of xr.DataArray this path touches)."""

    def __init__(self, data, dims, coords=None, attrs=None, name=None):
        self.data = data
        self.dims = tuple(dims)
        assert len(self.dims) == data.ndim, (self.dims, data.shape)
        self.coords = {}
        coords = coords or {}
        for ax, dim in enumerate(self.dims):
            if dim in coords:
                c = np.asarray(coords[dim])
                assert len(c) == data.shape[ax], (dim, len(c), data.shape)
                self.coords[dim] = c
            else:
                self.coords[dim] = np.arange(data.shape[ax])
        self.attrs = attrs if attrs is not None else {}
        self.name = name

    @classmethod
    def _from_parts(cls, data, dims, coords, attrs, name):
        """Constructor for parts that are known to be consistent (slices / copies of a valid image): no coordinate checks."""
        out = object.__new__(cls)
        out.data, out.dims, out.coords, out.attrs, out.name = data, dims, coords, attrs, name
        return out

    shape = property(lambda self: tuple(self.data.shape))
    dtype = property(lambda self: self.data.dtype)
    ndim = property(lambda self: self.data.ndim)
    sizes = property(lambda self: dict(zip(self.dims, self.data.shape)))

    def copy(self, deep=False, data=None):
        d = self.data if data is None else data
        if deep and data is None:
            d = np.array(d, copy=True)
        attrs = _copy.deepcopy(self.attrs) if deep else dict(self.attrs, transforms=dict(self.attrs.get("transforms", {})))
        coords = {k: v.copy() for k, v in self.coords.items()}
        if data is None or tuple(d.shape) == tuple(self.data.shape):
            return SpatialImage._from_parts(d, self.dims, coords, attrs, self.name)
        return SpatialImage(d, self.dims, coords, attrs, self.name)

    def astype(self, dtype):
        return self.copy(data=self.data.astype(dtype))

    def isel(self, indexers):
        """Positional selection; ints drop the dim, slices keep it."""
        idx = []
        dims = []
        coords = {}
        for ax, dim in enumerate(self.dims):
            sel = indexers.get(dim, slice(None))
            idx.append(sel)
            if isinstance(sel, (int, np.integer)):
                continue
            dims.append(dim)
            coords[dim] = self.coords[dim][sel]
        # (slices of consistent parts are consistent: no coordinate checks)
        return SpatialImage._from_parts(self.data[tuple(idx)], tuple(dims), coords,
                                        dict(self.attrs, transforms=dict(self.attrs.get("transforms", {}))), self.name)

    def sel(self, indexers):
        """Label selection: scalar (exact coordinate) or slice(lo, hi) inclusive, like xarray."""
        pos = {}
        for dim, sel in indexers.items():
            c = self.coords[dim]
            if isinstance(sel, slice):
                lo = 0 if sel.start is None else int(np.searchsorted(c, sel.start, side="left"))
                hi = len(c) if sel.stop is None else int(np.searchsorted(c, sel.stop, side="right"))
                pos[dim] = slice(lo, hi)
            else:
                hit = np.nonzero(c == sel)[0]
                if not len(hit):
                    raise KeyError(f"{sel!r} not found in coordinate {dim!r}")
                pos[dim] = int(hit[0])
        return self.isel(pos)

    def squeeze(self, dim=None, drop=True):
        dims = [dim] if isinstance(dim, str) else (dim or [d for d, n in self.sizes.items() if n == 1])
        return self.isel({d: 0 for d in dims if self.sizes.get(d) == 1})

    def expand_dims(self, dim, axis=0):
        data = np.expand_dims(self.data, axis)
        dims = list(self.dims)
        dims.insert(axis, dim)
        out = SpatialImage(data, dims, dict(self.coords), dict(self.attrs), self.name)
        return out

    def transpose(self, *dims):
        perm = [self.dims.index(d) for d in dims]
        return SpatialImage(np.transpose(self.data, perm), dims, dict(self.coords), dict(self.attrs), self.name)

    def __repr__(self):
        return f"<SpatialImage {dict(self.sizes)} {self.dtype} transforms={list(self.attrs.get('transforms', {}))}>"


def _get_axis_coords(dim, size, scale, translation):
    """spatial_image_utils._get_axis_coords (spatial_image_utils.py:316-317)."""
    return translation + scale * np.arange(size, dtype=float)


def to_spatial_image(data, dims=None, scale=None, translation=None, c_coords=None, t_coords=None):
    """spatial_image_utils.to_spatial_image (spatial_image_utils.py:320-370)."""
    if scale is None or translation is None:
        raise ValueError("scale and translation must be provided")
    name = None
    if isinstance(data, SpatialImage):
        name = data.name
        data = data.data
    if dims is None:
        dims = SPATIAL_DIMS[-data.ndim:]
    dims = tuple(dims)
    coords = {}
    for axis, dim in enumerate(dims):
        size = data.shape[axis]
        if dim in SPATIAL_DIMS:
            coords[dim] = _get_axis_coords(dim, size, scale[dim], translation[dim])
        elif dim == "c":
            coords[dim] = np.asarray(c_coords) if c_coords is not None else np.arange(size)
        elif dim == "t":
            coords[dim] = np.asarray(t_coords) if t_coords is not None else np.arange(size)
    return SpatialImage(data, dims, coords, {}, name)


def get_default_spatial_chunksizes(ndim):
    assert ndim in [2, 3]
    return dict(DEFAULT_SPATIAL_CHUNKSIZES_2D if ndim == 2 else DEFAULT_SPATIAL_CHUNKSIZES_3D)


def get_sim_from_array(
    array,
    dims=None,
    scale=None,
    translation=None,
    affine=None,
    transform_key=DEFAULT_TRANSFORM_KEY,
    c_coords=None,
    t_coords=None,
):
    """spatial_image_utils.get_sim_from_array (spatial_image_utils.py:416-542).

    Missing ``c``/``t`` axes are added as singletons and dims are ordered
    (c, t, z, y, x) like the reference."""
    if isinstance(array, SpatialImage):
        if dims is None:
            dims = list(array.dims)
        if c_coords is None and "c" in array.coords and "c" in array.dims:
            c_coords = array.coords["c"]
        if t_coords is None and "t" in array.coords and "t" in array.dims:
            t_coords = array.coords["t"]
        array = array.data
    array = np.asarray(array)
    if dims is None:
        dims = ["t", "c", "z", "y", "x"][-array.ndim:]
    dims = list(dims)
    assert len(dims) == array.ndim
    for nsdim in ["c", "t"]:
        if nsdim not in dims:
            array = array[None]
            dims = [nsdim] + dims
    new_dims = [d for d in SPATIAL_IMAGE_DIMS if d in dims]
    if new_dims != dims:
        array = np.transpose(array, [dims.index(d) for d in new_dims])
        dims = new_dims
    spatial_dims = [d for d in dims if d in SPATIAL_DIMS]
    ndim = len(spatial_dims)
    if scale is None:
        scale = {d: 1 for d in spatial_dims}
    if translation is None:
        translation = {d: 0 for d in spatial_dims}
    sim = to_spatial_image(array, dims, scale, translation, c_coords, t_coords)
    affine_x = param_utils.identity_transform(ndim) if affine is None else param_utils.affine_to_xaffine(affine)
    set_sim_affine(sim, affine_x, transform_key=transform_key)
    return sim


def get_spatial_dims_from_sim(sim):
    return [dim for dim in ["z", "y", "x"] if dim in sim.dims]


def get_nonspatial_dims_from_sim(sim):
    sdims = get_spatial_dims_from_sim(sim)
    return [dim for dim in sim.dims if dim not in sdims]


def get_ndim_from_sim(sim):
    return len(get_spatial_dims_from_sim(sim))


def get_origin_from_sim(sim, asarray=False):
    """spatial_image_utils.py:554-561: first coordinate per spatial dim."""
    sdims = get_spatial_dims_from_sim(sim)
    origin = {dim: float(sim.coords[dim][0]) for dim in sdims}
    return np.array([origin[d] for d in sdims]) if asarray else origin


def get_shape_from_sim(sim, asarray=False):
    sdims = get_spatial_dims_from_sim(sim)
    shape = {dim: len(sim.coords[dim]) for dim in sdims}
    return np.array([shape[d] for d in sdims]) if asarray else shape


def get_spacing_from_sim(sim, asarray=False):
    """spatial_image_utils.py:574-589: coords[1]-coords[0], 1.0 for singleton axes."""
    sdims = get_spatial_dims_from_sim(sim)
    spacing = {
        dim: float(sim.coords[dim][1] - sim.coords[dim][0]) if len(sim.coords[dim]) > 1 else 1.0
        for dim in sdims
    }
    return np.array([spacing[d] for d in sdims]) if asarray else spacing


def get_stack_properties_from_sim(sim, transform_key=None, asarray=False):
    """spatial_image_utils.py:863-873."""
    props = {
        "shape": get_shape_from_sim(sim, asarray=asarray),
        "spacing": get_spacing_from_sim(sim, asarray=asarray),
        "origin": get_origin_from_sim(sim, asarray=asarray),
    }
    if transform_key is not None:
        props["transform"] = get_affine_from_sim(sim, transform_key)
    return props


def get_affine_from_sim(sim, transform_key):
    if transform_key not in sim.attrs.get("transforms", {}):
        raise Exception("Transform key %s not found in sim" % transform_key)
    return sim.attrs["transforms"][transform_key]


def get_tranform_keys_from_sim(sim):
    return list(sim.attrs.get("transforms", {}).keys())


def set_sim_affine(sim, xaffine, transform_key, base_transform_key=None):
    """spatial_image_utils.set_sim_affine (spatial_image_utils.py:1234-1245)."""
    if "transforms" not in sim.attrs:
        sim.attrs["transforms"] = {}
    if base_transform_key is not None:
        xaffine = param_utils.rebase_affine(xaffine, get_affine_from_sim(sim, base_transform_key))
    sim.attrs["transforms"][transform_key] = np.asarray(xaffine, dtype=np.float64)


def get_center_of_sim(sim, transform_key=None):
    """spatial_image_utils.py:1248-1275."""
    sdims = get_spatial_dims_from_sim(sim)
    ndim = len(sdims)
    sp, o, sh = get_spacing_from_sim(sim), get_origin_from_sim(sim), get_shape_from_sim(sim)
    center = np.array([o[d] + sp[d] * (sh[d] - 1) / 2 for d in sdims])
    if transform_key is not None:
        affine = param_utils.select_time(get_affine_from_sim(sim, transform_key), 0)
        center = np.matmul(affine, np.concatenate([center, np.ones(1)]))[:ndim]
    return center


def sim_sel_coords(sim, sel_dict):
    """spatial_image_utils.sim_sel_coords (spatial_image_utils.py:1278-1300)."""
    ssim = sim.sel(sel_dict)
    if "t" in sel_dict and not isinstance(sel_dict["t"], slice):
        it = int(np.nonzero(sim.coords["t"] == sel_dict["t"])[0][0])
        ssim.attrs["transforms"] = {
            k: param_utils.select_time(v, it) for k, v in sim.attrs.get("transforms", {}).items()
        }
    return ssim


def get_sim_field(sim, ns_coords=None):
    """spatial_image_utils.get_sim_field (spatial_image_utils.py:1303-1315)."""
    nsdims = get_nonspatial_dims_from_sim(sim)
    if not nsdims:
        return sim
    if ns_coords is None:
        ns_coords = {dim: sim.coords[dim][0] for dim in nsdims}
    return sim_sel_coords(sim, ns_coords)


def max_project_sim(sim, dim="z"):
    """spatial_image_utils.max_project_sim (spatial_image_utils.py:1553-1585), numpy only."""
    ax = sim.dims.index(dim)
    data = sim.data.max(axis=ax)
    dims = [d for d in sim.dims if d != dim]
    out = SpatialImage(data, dims, {d: sim.coords[d] for d in dims}, {}, sim.name)
    keep = [i for i, d in enumerate(get_spatial_dims_from_sim(sim) + ["1"]) if d != dim]
    out.attrs["transforms"] = {
        k: np.asarray(v)[..., keep, :][..., :, keep] for k, v in sim.attrs.get("transforms", {}).items()
    }
    return out
