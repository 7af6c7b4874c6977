"""OME-Zarr (NGFF 0.4 on Zarr v2, NGFF 0.5 on Zarr v3) reading and writing around the fuse path.

Mirror of the call shapes of src/multiview_stitcher/ngff_utils.py for the part the hot path touches:
``write_sim_to_ome_zarr`` (ngff_utils.py:1564-1760: resolution levels by block means, one Zarr array per level,
``multiscales`` metadata), ``calc_ngff_coordinate_transformations_and_axes`` (ngff_utils.py:1493-1561) and the
readers that hand zarr-backed views to ``fusion.fuse`` / ``registration.register``.  Storage goes through
``zarr_io`` (Zarr v2 / v3 restated, no zarr / ngff-zarr package here); the pyramid's block means run on the GPU
(``mvs_bin_mean``, the same kernel registration binning uses).  ``ngff_version="0.5"`` writes a Zarr v3 hierarchy with the
metadata under the ``ome`` key (ngff_utils.py:1185-1281, 1820-1905).
"""

from __future__ import annotations

import os

import numpy as np

from . import msi_utils, param_utils, zarr_io
from . import spatial_image_utils as si_utils

DEFAULT_NGFF_TIME_TRANSFORM = {"scale": 1.0, "translation": 0.0, "unit": None}


def calc_ngff_coordinate_transformations_and_axes(stack_properties_res0, res_abs_factors, nsdims=None, time_transform=None):
    """Per level ``[scale, translation]`` and the axes list (ngff_utils.py:1493-1561): level spacing =
    spacing * factor, level origin = origin + (factor - 1) * spacing / 2 (centre of the first block)."""
    spacing, origin = stack_properties_res0["spacing"], stack_properties_res0["origin"]
    sdims = list(spacing.keys())
    nsdims = list(nsdims or [])
    tt = {**DEFAULT_NGFF_TIME_TRANSFORM, **(time_transform or {})}
    ns_scales = [float(tt["scale"]) if d == "t" else 1.0 for d in nsdims]
    ns_trans = [float(tt["translation"]) if d == "t" else 0 for d in nsdims]
    coordtfs = [
        [
            {"type": "scale", "scale": ns_scales + [float(spacing[d] * f[d]) for d in sdims]},
            {"type": "translation", "translation": ns_trans + [float(origin[d] + (f[d] - 1) * spacing[d] / 2) for d in sdims]},
        ]
        for f in res_abs_factors
    ]
    axes = []
    for d in nsdims + sdims:
        ax = {"name": d, "type": "channel" if d == "c" else ("time" if d == "t" else "space")}
        if d in sdims:
            ax["unit"] = "micrometer"
        elif d == "t" and tt["unit"]:
            ax["unit"] = tt["unit"]
        axes.append(ax)
    return coordtfs, axes


def write_multiscales_metadata(group_path, axes, datasets, ngff_version="0.4", name="/"):
    """``multiscales`` attribute of the group (ngff_utils.py:1185-1230): only axes, datasets, name and version.  NGFF 0.4
    keeps ``multiscales`` (each with its ``version``) at the top of the group attributes, 0.5 nests ``multiscales`` and
    ``version`` inside an ``ome`` object."""
    v = str(ngff_version)
    if not (v.startswith("0.4") or v.startswith("0.5")):
        raise ValueError(f"ngff_version {ngff_version} not supported")
    multiscale = {
        "axes": [dict(a) for a in axes],
        "datasets": [{"path": d["path"], "coordinateTransformations": [dict(t) for t in d["coordinateTransformations"]]} for d in datasets],
        "name": name,
    }
    attrs = zarr_io.read_attrs(group_path)
    if v.startswith("0.4"):
        multiscale["version"] = v
        attrs["multiscales"] = [multiscale]
    else:
        ome = dict(attrs.get("ome") or {})
        ome["version"] = v
        ome["multiscales"] = [multiscale]
        attrs["ome"] = ome
    zarr_io.write_attrs(group_path, attrs)


def zarr_group_creation_kwargs_for_ngff_version(ngff_version):
    """NGFF 0.4 is a Zarr v2 hierarchy, 0.5 a Zarr v3 one (ngff_utils.py:1243-1255)."""
    v = str(ngff_version)
    if v.startswith("0.4"):
        return {"zarr_format": 2}
    if v.startswith("0.5"):
        return {"zarr_format": 3}
    raise ValueError(f"ngff_version {ngff_version} not supported")


def update_zarr_array_creation_kwargs_for_ngff_version(ngff_version, zarr_array_creation_kwargs=None):
    """NGFF 0.4 arrays are Zarr v2 with '/' as dimension separator, 0.5 arrays Zarr v3 (ngff_utils.py:1258-1281)."""
    kw = dict(zarr_array_creation_kwargs or {})
    if str(ngff_version) == "0.4":
        kw["dimension_separator"] = "/"
        kw["zarr_format"] = 2
    elif str(ngff_version) == "0.5":
        kw["zarr_format"] = 3
    else:
        raise ValueError(f"ngff_version {ngff_version} not supported")
    return kw


def _chunk_shape_from_sim(sim):
    """Chunk shape of a level: the stored chunks of zarr-backed data, else the reference's defaults
    (spatial_image_utils.py:28-29), 1 along c and t."""
    data = sim.data
    if zarr_io.is_zarr_backed(data) and getattr(data, "array", data).ndim == len(sim.dims):
        return list(getattr(data, "array", data).chunks)
    sdims = si_utils.get_spatial_dims_from_sim(sim)
    default = si_utils.DEFAULT_SPATIAL_CHUNKSIZES_3D if len(sdims) == 3 else si_utils.DEFAULT_SPATIAL_CHUNKSIZES_2D
    return [min(default[d], sim.sizes[d]) if d in sdims else 1 for d in sim.dims]


def _downsample_level(src, dst, dims, factors, device):
    """dst = block mean of src by ``factors`` over the spatial dims (``coarsen(mean).astype(dtype)`` with the excess
    trimmed, ngff_utils.py:1284-1330), streamed: one destination chunk column at a time through ``mvs_bin_mean``."""
    from . import _reg_ops

    sdims = [d for d in dims if d in si_utils.SPATIAL_DIMS]
    nns = len(dims) - len(sdims)
    bins = [int(factors[d]) for d in sdims]
    sp_shape, sp_chunks = dst.shape[nns:], dst.chunks[nns:]
    # work unit: one destination chunk (all of it comes from bins x chunk source voxels)
    for ns in np.ndindex(*dst.shape[:nns]) if nns else [()]:
        for cidx in np.ndindex(*[-(-s // c) for s, c in zip(sp_shape, sp_chunks)]):
            lo = [i * c for i, c in zip(cidx, sp_chunks)]
            hi = [min(l + c, s) for l, c, s in zip(lo, sp_chunks, sp_shape)]
            win = src.read(list(ns) + [l * b for l, b in zip(lo, bins)], [n + 1 for n in ns] + [h * b for h, b in zip(hi, bins)])
            win = win.reshape(win.shape[nns:])
            out = _reg_ops.bin_mean(win, bins, device) if max(bins) > 1 else win
            dst.write(list(ns) + lo, np.asarray(out).reshape((1,) * nns + tuple(out.shape)))


def write_sim_to_ome_zarr(sim, output_zarr_url, downscale_factors_per_spatial_dim=None, overwrite=False,
                          ngff_version="0.4", zarr_array_creation_kwargs=None, device=0):
    """Write ``sim`` as a multiscale NGFF 0.4 image and return a sim backed by level 0 of the new store
    (ngff_utils.py:1564-1760).  Existing levels are kept when ``overwrite`` is False (so a level 0 that
    ``fusion.fuse(output_zarr_url=...)`` streamed out chunk by chunk is only completed with its pyramid);
    metadata is rewritten in any case.  Transforms are not stored in the file; the returned sim carries them."""
    kw = update_zarr_array_creation_kwargs_for_ngff_version(ngff_version, zarr_array_creation_kwargs)
    dims = list(sim.dims)
    sdims = si_utils.get_spatial_dims_from_sim(sim)
    nsdims = si_utils.get_nonspatial_dims_from_sim(sim)
    spacing, origin = si_utils.get_spacing_from_sim(sim), si_utils.get_origin_from_sim(sim)
    spatial_shape = {d: int(sim.sizes[d]) for d in sdims}
    res_shapes, res_rel, res_abs = msi_utils.calc_resolution_levels(spatial_shape, downscale_factors_per_spatial_dim)
    coordtfs, axes = calc_ngff_coordinate_transformations_and_axes(
        {"spacing": spacing, "origin": origin, "shape": spatial_shape}, res_abs, nsdims=nsdims)
    chunks = _chunk_shape_from_sim(sim)
    if kw.get("chunks") is not None:      # zarr_array_creation_kwargs={"chunks": ...}: full-rank, or spatial dims only
        req = [int(v) for v in kw["chunks"]]
        if len(req) == len(sdims):
            req = [1] * len(nsdims) + req
        if len(req) != len(dims):
            raise ValueError(f"chunks {kw['chunks']} do not match dims {dims}")
        chunks = req
    kw = {k: v for k, v in kw.items() if k != "chunks"}
    ns_shape = [int(sim.sizes[d]) for d in nsdims]

    zarr_io.create_group(output_zarr_url, overwrite=overwrite, **zarr_group_creation_kwargs_for_ngff_version(ngff_version))
    if kw.get("zarr_format") == 3:
        kw.setdefault("dimension_names", dims)
    prev = None
    for level, shp in enumerate(res_shapes):
        url = os.path.join(output_zarr_url, str(level))
        if not overwrite and zarr_io.array_exists(url):
            prev = zarr_io.ZarrArray.open(url)
            continue
        arr = zarr_io.ZarrArray.create(url, ns_shape + [shp[d] for d in sdims], chunks, sim.dtype, overwrite=True, **kw)
        if level == 0:
            # level 0: chunk by chunk out of whatever backs the sim (host array or another zarr window)
            for cidx in np.ndindex(*arr.grid):
                lo = [i * c for i, c in zip(cidx, arr.chunks)]
                hi = [min(l + c, s) for l, c, s in zip(lo, arr.chunks, arr.shape)]
                arr.write(lo, np.asarray(sim.data[tuple(slice(l, h) for l, h in zip(lo, hi))]))
        else:
            _downsample_level(prev, arr, dims, res_rel[level], device)
        prev = arr
    write_multiscales_metadata(
        output_zarr_url, axes, [{"path": str(i), "coordinateTransformations": coordtfs[i]} for i in range(len(res_shapes))],
        ngff_version=ngff_version)
    out = read_sim_from_ome_zarr(output_zarr_url, 0)
    out.attrs["transforms"] = dict(sim.attrs.get("transforms", {}))
    return out


def _multiscales_of(path):
    attrs = zarr_io.read_attrs(path)
    ms = attrs.get("multiscales") or attrs.get("ome", {}).get("multiscales")
    if not ms:
        raise ValueError(f"{path} holds no NGFF multiscales metadata")
    return ms[0]


def read_sim_from_ome_zarr(path, resolution_level=0, transform_key=None):
    """Zarr-backed (lazy) sim of one resolution level; scale / translation from the dataset's coordinate
    transformations (ngff_utils.py:1101-1139), identity affine under ``transform_key`` if given."""
    ms = _multiscales_of(path)
    dims = [a["name"] for a in ms["axes"]]
    ds = ms["datasets"][resolution_level]
    scale, trans = [1.0] * len(dims), [0.0] * len(dims)
    for t in list(ms.get("coordinateTransformations") or []) + list(ds.get("coordinateTransformations") or []):
        if t.get("type") == "scale":
            scale = [a * float(b) for a, b in zip(scale, t["scale"])]
        elif t.get("type") == "translation":
            trans = [a + float(b) for a, b in zip(trans, t["translation"])]
    arr = zarr_io.ZarrArray.open(os.path.join(path, ds["path"]))
    sdims = [d for d in dims if d in si_utils.SPATIAL_DIMS]
    sim = si_utils.to_spatial_image(
        arr[...], dims=dims, scale={d: scale[dims.index(d)] for d in sdims},
        translation={d: trans[dims.index(d)] for d in sdims})
    if transform_key is not None:
        si_utils.set_sim_affine(sim, param_utils.identity_transform(len(sdims)), transform_key)
    return sim


def read_msim_from_ome_zarr(path, transform_key=None):
    """All levels as a multiscale image (ngff_utils.py:1142-1182)."""
    ms = _multiscales_of(path)
    sims = [read_sim_from_ome_zarr(path, i, transform_key) for i in range(len(ms["datasets"]))]
    return msi_utils.MultiscaleSpatialImage(sims, dict(sims[0].attrs.get("transforms", {})))
