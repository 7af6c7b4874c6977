"""Pairwise registration on the HIP backend (mirror of the reference's ``registration`` API for the hot path).

``phase_correlation_registration`` == registration.phase_correlation_registration
    (src/multiview_stitcher/registration.py:353-565): the host keeps the reference's control flow
    (candidate enumeration, quirks Q1-Q3, nanargmax) and every voxel-sized step runs in
    libmvs_hip.so: mvs_rescale_intensity, mvs_phasecorr, mvs_score_candidates.
"""

from __future__ import annotations

import logging
import os
import warnings

import threading

import numpy as np

from . import _lib, _reg_ops, param_utils
from .device import is_device_array

logger = logging.getLogger(__name__)


def _as_array(x):
    return x.data if hasattr(x, "dims") and hasattr(x, "coords") else x


def _enumerate_candidates(shift_candidates, shape, ndim):
    """registration.py:453-477: per shift estimate and per axis with a non-zero shift the four variants
    {s, -s, -(s - N), -s - N}; keep |t| < max over ALL dims of the image shape.  Same order as the reference's
    nested loops (itertools.product over the axes, last axis fastest); plain Python floats, no per-element numpy calls."""
    import itertools

    max_shift_per_dim = float(max(shape))
    t_candidates = []
    for shift_candidate in shift_candidates:
        per_axis = []
        for d in range(ndim):
            sc = shift_candidate[d]
            if sc == 0:
                per_axis.append((sc,))
            else:
                n = shape[d]
                per_axis.append((sc, -sc, -(sc - n), -sc - n))
        for t in itertools.product(*per_axis):
            if max(abs(v) for v in t) < max_shift_per_dim:
                t_candidates.append(list(t))
    return t_candidates


def phase_correlation_registration(fixed_data, moving_data, disambiguate_region_mode=None, device=0,
                                   return_debug=False, _constant_check=False, **skimage_phase_corr_kwargs):
    """Translation between two same-shape overlap crops (float32, NaN = outside the view).

    Returns ``{"affine_matrix": (ndim+1, ndim+1) translation mapping fixed px -> moving px,
    "quality": float}`` exactly like the reference, including its ``[zeros(ndim)]`` return when no
    candidate survives (registration.py:479-480)."""
    im0 = _as_array(fixed_data)
    im1 = _as_array(moving_data)
    if tuple(im0.shape) != tuple(im1.shape):
        raise ValueError("fixed and moving crops must have the same shape")
    shape = tuple(int(s) for s in im0.shape)
    ndim = len(shape)
    on_dev = is_device_array(im0)

    if not return_debug and set(skimage_phase_corr_kwargs) <= {"upsample_factor"}:
        # the whole function as one library call (same steps, same quirks); the Python flow below stays for the debug
        # outputs and for callers that pass other phase_cross_correlation keywords
        uf = skimage_phase_corr_kwargs.get("upsample_factor", 10 if ndim == 2 else 2)
        t, quality, status, _ = _reg_ops.register_crops(im0, im1, uf, disambiguate_region_mode, _constant_check, device)
        if status == 2:
            warnings.warn(
                "An overlap region between tiles/views is all zero or constant. Assuming identity transform.",
                UserWarning, stacklevel=3,
            )
            return {"affine_matrix": param_utils.identity_transform(ndim), "quality": np.nan}
        if status == 1:
            return [np.zeros(ndim)]
        if status == 3:
            raise ValueError("All-NaN slice encountered")     # what np.nanargmax raises in the reference (registration.py:558-559)
        return {"affine_matrix": param_utils.affine_from_translation([float(v) for v in t]), "quality": quality}

    # normalise (registration.py:381-389); the kernel also reports nanmin/nanmax/#valid of the INPUT
    im0, min0, max0, nvalid0 = _reg_ops.rescale_intensity(im0, device, out_on_device=on_dev)
    im1, min1, max1, nvalid1 = _reg_ops.rescale_intensity(im1, device, out_on_device=on_dev)
    if _constant_check and (min0 == max0 or min1 == max1):
        # dispatch_pairwise_reg_func's guard (registration.py:1500-1520), folded in here so that the
        # nanmin / nanmax pass over both crops is not run twice
        warnings.warn(
            "An overlap region between tiles/views is all zero or constant. Assuming identity transform.",
            UserWarning, stacklevel=3,
        )
        return {"affine_matrix": param_utils.identity_transform(ndim), "quality": np.nan}
    n = int(np.prod(shape))
    has_nan = (nvalid0 < n) or (nvalid1 < n)
    if disambiguate_region_mode is None:
        disambiguate_region_mode = "intersection" if has_nan else "union"
    if "upsample_factor" not in skimage_phase_corr_kwargs:
        skimage_phase_corr_kwargs["upsample_factor"] = 10 if ndim == 2 else 2
    upsample_factor = skimage_phase_corr_kwargs["upsample_factor"]

    # strategy of the reference: phase correlation with and without normalisation, the candidate with
    # the best structural similarity wins (registration.py:413-431). NaNs -> 0 happens in the kernel.
    shift_candidates, pcc_debug = [], []
    for s, dbg in _reg_ops.phase_cross_correlation_multi(im0, im1, upsample_factor, ("phase", None), device):
        shift_candidates.append(s)
        pcc_debug.append(dbg)
    if has_nan:
        # the masked variant is called with inverted masks on NaN-holding images (registration.py:433-443)
        # and yields a zero shift: it only adds the candidate t = 0
        shift_candidates.append(np.zeros(ndim, dtype=np.float32))

    # after rescaling, the values present are exactly min -> 0 and max -> 1 per image
    def _rescaled_range(mn, mx, nv):
        if nv == 0:
            return np.nan, np.nan
        return (0.0, 1.0) if mx != mn else (float(np.float32(mn)), float(np.float32(mn)))

    lo0, hi0 = _rescaled_range(min0, max0, nvalid0)
    lo1, hi1 = _rescaled_range(min1, max1, nvalid1)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore", category=RuntimeWarning)
        data_range = np.float32(np.nanmax([hi0, hi1])) - np.float32(np.nanmin([lo0, lo1]))
    im1_min = lo1

    t_candidates = _enumerate_candidates(shift_candidates, shape, ndim)
    if not len(t_candidates):
        return [np.zeros(ndim)]

    # only the Spearman value of the winning candidate is reported, so it is only evaluated for that one
    # (all of them when the debug lists are requested)
    ssim, spear, codes = _reg_ops.score_candidates(im0, im1, t_candidates, disambiguate_region_mode, data_range, im1_min,
                                                   device, quality_for_all=return_debug)
    # metric lists exactly as the reference builds them: code 2 (`continue`, registration.py:530-533) appends nothing
    disambiguate_metric_vals = [float(ssim[i]) for i in range(len(codes)) if codes[i] != 2]
    quality_metric_vals = [float(spear[i]) for i in range(len(codes)) if codes[i] != 2]
    with warnings.catch_warnings():
        warnings.simplefilter("ignore", category=RuntimeWarning)
        argmax_index = int(np.nanargmax(disambiguate_metric_vals))
    t = t_candidates[argmax_index]
    reg_result = {
        "affine_matrix": param_utils.affine_from_translation([float(v) for v in t]),
        "quality": quality_metric_vals[argmax_index],
    }
    if return_debug:
        reg_result["debug"] = {
            "shift_candidates": [np.asarray(s, dtype=np.float64) for s in shift_candidates], "pcc": pcc_debug,
            "t_candidates": np.asarray(t_candidates, dtype=np.float64), "codes": list(codes),
            "ssim": disambiguate_metric_vals, "spearman": quality_metric_vals, "argmax_index": argmax_index,
            "region_mode": disambiguate_region_mode, "data_range": float(data_range), "im1_min": float(im1_min),
        }
    return reg_result


def get_optimal_registration_binning(sim1, sim2, max_total_pixels_per_stack=400**3, overlap_tolerance=None):
    """registration.get_optimal_registration_binning (registration.py:114-191): +1 steps (not doublings) on
    the axis with the smallest current spacing (z alone, or x and y together) until the larger of the two
    stacks has fewer than 400^3 voxels."""
    from . import spatial_image_utils as si_utils

    if overlap_tolerance is not None:
        raise NotImplementedError("overlap_tolerance")
    sdims = si_utils.get_spatial_dims_from_sim(sim1)
    ndim = len(sdims)
    input_spacings = [si_utils.get_spacing_from_sim(s) for s in [sim1, sim2]]
    sh1, sh2 = si_utils.get_shape_from_sim(sim1), si_utils.get_shape_from_sim(sim2)
    # a pure function of shapes and spacings: the pairs of a regular mosaic all share one answer
    memo_key = (tuple(sdims), tuple(sh1[d] for d in sdims), tuple(sh2[d] for d in sdims),
                tuple(input_spacings[0][d] for d in sdims), tuple(input_spacings[1][d] for d in sdims), max_total_pixels_per_stack)
    if memo_key in _BINNING_MEMO:
        return dict(_BINNING_MEMO[memo_key])
    overlap = {d: max(sh1[d], sh2[d]) for d in sdims}
    binning = {d: 1 for d in sdims}
    spacings = input_spacings
    while np.prod([overlap[d] / binning[d] for d in sdims]) >= max_total_pixels_per_stack:
        dim_to_bin = np.argmin([min(spacings[i][d] for i in range(2)) for d in sdims])
        if ndim == 3 and dim_to_bin == 0:
            binning["z"] += 1
        else:
            for d in ["x", "y"]:
                binning[d] += 1
        spacings = [{d: input_spacings[i][d] * binning[d] for d in sdims} for i in range(2)]
    if len(_BINNING_MEMO) > 256:
        _BINNING_MEMO.clear()
    _BINNING_MEMO[memo_key] = dict(binning)
    return binning


_BINNING_MEMO = {}


# =====================================================================================================
# Pair preparation and the register() workflow (host glue around the kernels)
# =====================================================================================================
def _is_axis_aligned(affine, tol=1e-9):
    A = np.asarray(affine)[:-1, :-1]
    return bool(np.all(np.abs(A - np.diag(np.diag(A))) < tol))


def _get_overlap_bboxes(sim1, sim2, input_transform_key=None, output_transform_key=None, overlap_tolerance=None, closed_form=True):
    """registration._get_overlap_bboxes (registration.py:194-277): the two view boxes are intersected as halfspaces in the
    coordinate system of ``input_transform_key`` (mv_graph.py:301-338) and the vertices of the intersection polytope are
    projected into each view's intrinsic frame; lowers / uppers are their extrema.  Axis-aligned pairs (every tile grid)
    take the closed form -- the polytope is the intersection of the world AABBs, whose 2^ndim corners are the vertices;
    rotated / sheared / scaled views go through scipy's linprog + HalfspaceIntersection like the reference.
    Returns None when the views do not overlap."""
    from . import mv_graph
    from . import spatial_image_utils as si_utils
    from .transformation import transform_pts

    sims = [sim1, sim2]
    affines = [param_utils.select_time(si_utils.get_affine_from_sim(s, input_transform_key), 0) for s in sims]
    sps = [si_utils.get_stack_properties_from_sim(s) for s in sims]
    if overlap_tolerance is not None:
        sps = [si_utils_extend(sp, overlap_tolerance) for sp in sps]
    if closed_form and all(_is_axis_aligned(a) for a in affines):
        res = mv_graph.get_overlap_aabb(sps[0], affines[0], sps[1], affines[1])
        if res is None:
            return None
        lo, hi = res
        ndim = len(lo)
        corners = np.array(list(np.ndindex(*([2] * ndim)))) * (hi - lo) + lo
        vol = float(np.prod(hi - lo))
    else:
        # (closed_form=False: the reference's own sequence -- linprog feasible point, Qhull halfspace intersection -- also for
        # axis-aligned pairs, whose vertices then carry Qhull's round-off like the reference's: registration.py:229-239)
        vol, hs = mv_graph.get_overlap_between_pair_of_stack_props(dict(sps[0], transform=affines[0]), dict(sps[1], transform=affines[1]),
                                                                   closed_form=closed_form)
        if hs is None:
            return None
        corners = np.asarray(hs.intersections)
    if output_transform_key is None:
        cts = [transform_pts(corners, np.linalg.inv(a)) for a in affines]
    elif output_transform_key == input_transform_key:
        cts = [corners, corners]
    else:
        raise NotImplementedError
    return {"lowers": [np.min(c, axis=0) for c in cts], "uppers": [np.max(c, axis=0) for c in cts], "vol": float(vol)}


def si_utils_extend(stack_props, extend_by):
    """spatial_image_utils.extend_stack_props (spatial_image_utils.py:889-913) on a copy."""
    sp = {k: dict(v) for k, v in stack_props.items()}
    if not isinstance(extend_by, dict):
        extend_by = {d: extend_by for d in sp["spacing"]}
    for d, val in extend_by.items():
        sp["shape"][d] += int(np.ceil(2 * val / sp["spacing"][d]))
        sp["origin"][d] -= val
    return sp


def _bin_sim(sim, binning, device, wait=True, out=None):
    """sim.coarsen(binning, boundary="trim").mean().astype(dtype) incl. the coarsened coordinates
    (registration.py:1732-1741)."""
    from . import spatial_image_utils as si_utils

    sdims = si_utils.get_spatial_dims_from_sim(sim)
    bins = [int(binning.get(d, 1)) for d in sdims]
    if max(bins) == 1:
        return sim
    if is_device_array(sim.data):
        sim.data.wait_ready(device)      # (an upload still in flight: this lane's stream waits for it)
    data = _reg_ops.bin_mean(sim.data, bins, device, wait=wait) if out is None else _reg_ops.bin_mean(sim.data, bins, device, wait=wait, out=out)
    coords = {}
    for d, b, n in zip(sdims, bins, data.shape):
        c = sim.coords[d][: n * b].reshape(n, b).mean(axis=1)
        coords[d] = c
    out = si_utils.SpatialImage(data, sdims, coords, {"transforms": dict(sim.attrs.get("transforms", {}))})
    return out


def sims_to_intrinsic_coord_system(sim1, sim2, transform_key, overlap_bboxes, device=0):
    """registration.sims_to_intrinsic_coord_system (registration.py:280-350): both views resampled (float32,
    NaN outside) onto the fixed view's pixel grid over the overlap box."""
    from . import spatial_image_utils as si_utils
    from .transformation import transform_sim

    sdims = si_utils.get_spatial_dims_from_sim(sim1)
    lowers, uppers = overlap_bboxes
    spacing = np.max([si_utils.get_spacing_from_sim(s, asarray=True) for s in [sim1, sim2]], axis=0)
    affines = [param_utils.select_time(s.attrs["transforms"][transform_key], 0) for s in [sim1, sim2]]
    transf_affine = np.matmul(np.linalg.inv(affines[1]), affines[0])
    shape = np.floor(np.array(uppers[0] - lowers[0]) / spacing + 1).astype(np.uint64)
    osp = {
        "origin": {d: float(lowers[0][i]) for i, d in enumerate(sdims)},
        "spacing": {d: float(spacing[i]) for i, d in enumerate(sdims)},
        "shape": {d: int(shape[i]) for i, d in enumerate(sdims)},
    }
    out = []
    for isim, sim in enumerate([sim1, sim2]):
        # allow_noop=False: the reference's no-op shortcut would hand back the input (already float32 there,
        # transformation.py:102-119); resampling on an identical grid yields the same values as float32
        t = transform_sim(sim, [None, transf_affine][isim], output_stack_properties=osp, mode="constant", cval=np.nan,
                          device=device, allow_noop=False)
        si_utils.set_sim_affine(t, si_utils.get_affine_from_sim(sim1, transform_key), transform_key)
        out.append(t)
    return out[0], out[1]


def get_affine_from_intrinsic_affine(data_affine, sim_fixed, sim_moving, transform_key_fixed=None, transform_key_moving=None):
    """registration.get_affine_from_intrinsic_affine (registration.py:1382-1474), including its use of
    ``transform_key_moving`` for the fixed view (:1423-1425)."""
    from . import spatial_image_utils as si_utils

    n = data_affine.shape[0]
    p2w_fixed = np.eye(n) if transform_key_fixed is None else param_utils.select_time(sim_fixed.attrs["transforms"][transform_key_moving], 0)
    p2w_moving = np.eye(n) if transform_key_moving is None else param_utils.select_time(sim_moving.attrs["transforms"][transform_key_moving], 0)

    def d_to_p(sim):
        return np.matmul(
            param_utils.affine_from_translation(si_utils.get_origin_from_sim(sim, asarray=True)),
            np.diag(list(si_utils.get_spacing_from_sim(sim, asarray=True)) + [1]),
        )

    D_to_W_f = np.matmul(p2w_moving, d_to_p(sim_moving))
    D_to_W_c = np.matmul(p2w_fixed, d_to_p(sim_fixed))
    return np.matmul(D_to_W_f, np.matmul(data_affine, np.linalg.inv(D_to_W_c)))


def dispatch_pairwise_reg_func(pairwise_reg_func, fixed_data=None, moving_data=None, skip_constant_check=False, device=0,
                               **pairwise_reg_func_kwargs):
    """registration.dispatch_pairwise_reg_func (registration.py:1477-1544): constant-image guard + call."""
    if pairwise_reg_func is phase_correlation_registration and fixed_data is not None and moving_data is not None \
            and not skip_constant_check:
        return pairwise_reg_func(fixed_data=fixed_data, moving_data=moving_data, device=device, _constant_check=True,
                                 **pairwise_reg_func_kwargs)
    if fixed_data is not None and moving_data is not None and not skip_constant_check:
        for name, im in (("fixed", fixed_data), ("moving", moving_data)):
            _, mn, mx, _ = _reg_ops.rescale_intensity(_as_array(im), device, out_on_device=is_device_array(_as_array(im)))
            if mn == mx:
                warnings.warn(
                    "An overlap region between tiles/views is all zero or constant. Assuming identity transform.",
                    UserWarning, stacklevel=2,
                )
                nd = _as_array(fixed_data).ndim
                return {"affine_matrix": param_utils.identity_transform(nd), "quality": np.nan}
    if fixed_data is not None:
        pairwise_reg_func_kwargs["fixed_data"] = fixed_data
        pairwise_reg_func_kwargs["moving_data"] = moving_data
    try:
        return pairwise_reg_func(device=device, **pairwise_reg_func_kwargs)
    except TypeError as e:
        if "device" not in str(e):
            raise
        return pairwise_reg_func(**pairwise_reg_func_kwargs)


# ---- lean per-pair host path ----------------------------------------------------------------------------------------
# The generic functions above spend ~0.45 ms of interpreter time per pair in small numpy calls, and every worker thread
# needs the GIL for them: on the north-star mosaic that serial share (144 x 0.45 ms) is a third of the registration time
# (adding 0.3 ms of Python per pair adds 40 ms per step).  For the common case -- views whose transform is a pure
# translation, the built-in phase correlation -- the same steps are done here with plain Python floats, in the same
# order of IEEE operations as the numpy expressions they replace (equality with the generic path is tested on the CPU
# for the plan and on the GPU for the result).
class _TileGeom:
    """What the pair plans need of one (binned) view: its coordinate arrays as lists, origin / spacing / shape as the
    spatial_image_utils getters derive them, and the translation of its transform."""

    __slots__ = ("sim", "sdims", "coords", "origin", "spacing", "shape", "t", "affine", "_src")

    def __init__(self, sim, transform_key, like=None, coords=None):
        self.sim = sim
        dims = sim.dims
        self.sdims = sd = [d for d in ("z", "y", "x") if d in dims]
        # (float64 arrays, not lists: 128 geometries of a 64-tile mosaic were 150 000 float objects to build under the GIL and
        # 2.5 ms to tear down when register() returns; float64 scalars subtract exactly like Python floats)
        co = sim.coords
        # ``coords``: the geometry of a binned version of ``sim`` whose voxels do not exist (the batched pair path plans on it)
        self.coords = cs = coords if coords is not None else [np.ascontiguousarray(co[d], dtype=np.float64) for d in sd]
        self.origin = [float(c[0]) for c in cs]
        self.spacing = [float(c[1] - c[0]) if len(c) > 1 else 1.0 for c in cs]
        self.shape = [len(c) for c in cs]
        src = sim.attrs["transforms"][transform_key]
        self._src = src
        if like is not None and like._src is src:       # the binned view of ``like``'s image: the very same transform array
            self.t, self.affine = like.t, like.affine
            return
        a = param_utils.select_time(np.asarray(src, dtype=np.float64), 0)
        n = len(sd)
        pure = False
        if a.shape == (n + 1, n + 1):
            b = a.copy()
            b[:n, n] = 0.0
            pure = b.tobytes() == _eye_bytes(n)      # (a -0.0 entry counts as "not pure": the generic path handles it)
        self.t = [float(v) for v in a[:n, n]] if pure else None
        self.affine = a


_EYE_BYTES = {}


def _eye_bytes(n):
    if n not in _EYE_BYTES:
        _EYE_BYTES[n] = np.eye(n + 1).tobytes()
    return _EYE_BYTES[n]


def _lean_world_box(g, tol):
    """world_aabb of stack props extended by ``tol`` (si_utils_extend) under a pure translation: per axis
    (fl(o + t), fl(fl(fl((n - 1) s) + o) + t)) with o, n the extended origin / shape."""
    lo, hi = [], []
    for k in range(len(g.sdims)):
        n, o, s = g.shape[k], g.origin[k], g.spacing[k]
        if tol is not None:
            n = n + int(np.ceil(2 * tol[k] / s))
            o = o - tol[k]
        lo.append(o + g.t[k])
        hi.append(((n - 1) * 1.0 * s + o) + g.t[k])
    return lo, hi


def _lean_overlap(g1, g2, tol, to_intrinsic):
    """_get_overlap_bboxes for two translated views: (lowers, uppers) per view, in each view's own frame
    (``to_intrinsic``) or in world coordinates."""
    lo1, hi1 = _lean_world_box(g1, tol)
    lo2, hi2 = _lean_world_box(g2, tol)
    lo = [max(a, b) for a, b in zip(lo1, lo2)]
    hi = [min(a, b) for a, b in zip(hi1, hi2)]
    if any(h < l for l, h in zip(lo, hi)):
        return None
    up = [1.0 * (h - l) + l for l, h in zip(lo, hi)]            # the far corner as the generic path builds it: gv * (hi - lo) + lo
    if not to_intrinsic:
        return [lo, lo], [up, up]
    lowers = [[l + (-t) for l, t in zip(lo, g.t)] for g in (g1, g2)]
    uppers = [[u + (-t) for u, t in zip(up, g.t)] for g in (g1, g2)]
    return lowers, uppers


# ---- crop length on the knife edge (registration.py:229-239, 314-316; mv_graph.py:301-338) ----------------------------------------
# The reference sizes the overlap crop from the vertices Qhull returns for the intersection of the two view boxes:
# floor((upper - lower) / spacing + 1).  For views on one pixel grid that quotient is an integer up to Qhull's round-off, so the
# reference registers N or N - 1 samples along an axis depending on the SIGN of a 1e-13 term that no closed form predicts (it
# depends on linprog's interior point and Qhull's elimination order).  The closed form registers N.  The default mode therefore
# asks the reference's own sequence ONCE per pair geometry whether it lands on N - 1 somewhere (memo below: a register + fuse loop
# over one mosaic pays the ~3 ms per pair once), and only such pairs leave the fast path for the ``overlap_bbox="reference"`` one.
_HOST_UPLOAD_ASYNC = [os.environ.get("MVS_HOST_STREAM", "1") != "0"]      # register() of plain host tiles: pinned staging + asynchronous uploads
_KNIFE_MEMO = {}
_KNIFE_LOCK = threading.Lock()
_KNIFE_CHECK = [os.environ.get("MVS_KNIFE_CHECK", "1") != "0"]


def set_crop_length_check(enabled):
    """Switch the default crop rule's question to the reference's Qhull sequence on or off for this process (on by default; the
    environment variable MVS_KNIFE_CHECK=0 starts a process with it off).  Off: every pair of views on one pixel grid registers the
    closed form's N samples -- the reference registers N - 1 on some of them (BASELINE config C1's pair), so its translation is
    then not guaranteed -- and a mosaic whose geometry has not been seen before saves ~2 ms of host time per pair.  Returns the
    previous setting."""
    old = _KNIFE_CHECK[0]
    _KNIFE_CHECK[0] = bool(enabled)
    return old


def _reference_crop_shape(g1, g2, tol):
    """Crop shape of the pair (binned views ``g1``, ``g2``: _TileGeom) as the reference derives it, or None without overlap."""
    from . import mv_graph
    from .transformation import transform_pts

    sps = []
    for g in (g1, g2):
        sp = {"origin": dict(zip(g.sdims, g.origin)), "spacing": dict(zip(g.sdims, g.spacing)), "shape": dict(zip(g.sdims, g.shape))}
        if any(tol):
            sp = si_utils_extend(sp, dict(zip(g.sdims, tol)))
        sps.append(dict(sp, transform=g.affine))
    vol, hs = mv_graph.get_overlap_between_pair_of_stack_props(sps[0], sps[1], closed_form=False, need_volume=False)
    if hs is None:
        return None
    c = transform_pts(np.asarray(hs.intersections), np.linalg.inv(g1.affine))
    lo, up = np.min(c, axis=0), np.max(c, axis=0)
    out_spacing = np.maximum(np.asarray(g1.spacing), np.asarray(g2.spacing))
    return np.floor((up - lo) / out_spacing + 1).astype(np.int64)


def _knife_key(g1, g2, tol):
    return (tuple(g1.origin), tuple(g1.spacing), tuple(g1.shape), tuple(g1.t), tuple(g2.origin), tuple(g2.spacing), tuple(g2.shape),
            tuple(g2.t), tuple(tol))


def _reference_crop_differs(g1, g2, tol, closed_form_shape):
    """True when the reference's crop of this pair is not the closed form's (one sample shorter along some axis)."""
    if not _KNIFE_CHECK[0]:
        return False
    key = _knife_key(g1, g2, tol)
    with _KNIFE_LOCK:
        hit = _KNIFE_MEMO.get(key)
    if hit is None:
        ref = _reference_crop_shape(g1, g2, tol)
        hit = bool(ref is not None and not np.array_equal(ref, np.asarray(closed_form_shape, dtype=np.int64)))
        with _KNIFE_LOCK:
            if len(_KNIFE_MEMO) > 65536:
                _KNIFE_MEMO.clear()
            _KNIFE_MEMO[key] = hit
    return hit


def _around10(x):
    """np.around(x, 10) for a Python float: rint(x * 1e10) / 1e10 (round() is half-to-even like rint)."""
    return round(x * 1e10) / 1e10


def _lean_pair_plan(g1, g2, tol):
    """Crop windows, output grid and pixel affines of one pair (steps of register_pair_of_msims /
    sims_to_intrinsic_coord_system / get_pixel_affine for translated views).  None when the views do not overlap."""
    import bisect

    ov = _lean_overlap(g1, g2, tol, True)
    if ov is None:
        return None
    lowers, uppers = ov
    n = len(g1.sdims)
    windows, origins, spacings = [], [], []
    for i, g in enumerate((g1, g2)):
        win, o, sp = [], [], []
        for k in range(n):
            c = g.coords[k]
            start = lowers[i][k] - 1e-6 - g.spacing[k]
            stop = uppers[i][k] + 1e-6 + g.spacing[k]
            # (bisect on the array, not np.searchsorted: a numpy call that drops the GIL makes the 16 pair threads queue for it)
            a, b = bisect.bisect_left(c, start), bisect.bisect_right(c, stop)
            if b <= a:
                return None
            win.append((a, b))
            o.append(float(c[a]))
            sp.append(float(c[a + 1] - c[a]) if b - a > 1 else 1.0)
        windows.append(win)
        origins.append(o)
        spacings.append(sp)
    out_spacing = [max(a, b) for a, b in zip(spacings[0], spacings[1])]
    out_origin = list(lowers[0])
    out_shape = [int(np.floor((uppers[0][k] - lowers[0][k]) / out_spacing[k] + 1)) for k in range(n)]
    t_rel = [a + (-b) for a, b in zip(g1.t, g2.t)]              # inv(A2) @ A1 for translations
    mats, offs = [], []
    for i in range(2):
        tt = [0.0] * n if i == 0 else t_rel
        m, o = [], []
        for k in range(n):
            m.append(_around10((1.0 * out_spacing[k]) / spacings[i][k]))
            v = _around10(((tt[k] + 0.0) - (origins[i][k] - out_origin[k])) / spacings[i][k])
            r = float(round(v))
            o.append(r if abs(v - r) <= 1e-6 else v)
        mats.append(m)
        offs.append(o)
    return {"windows": windows, "out_origin": out_origin, "out_spacing": out_spacing, "out_shape": out_shape,
            "matrix_diag": mats, "offset": offs}


def _lean_register_pair(g1b, g2b, g1, g2, sdims, tol, upsample_factor, transform_key, device):
    """register_pair_of_msims for translated views and the built-in phase correlation, on plans made of floats."""
    from . import spatial_image_utils as si_utils
    from .transformation import resample_array

    plan = _lean_pair_plan(g1b, g2b, tol)
    if plan is None:
        raise ValueError("views do not overlap")
    n = len(sdims)
    slabs = [g.sim.data[tuple(slice(a, b) for a, b in plan["windows"][i])] for i, g in enumerate((g1b, g2b))]
    uf = (10 if n == 2 else 2) if upsample_factor is None else upsample_factor
    if all(is_device_array(d) and d.dtype in _lib.DTYPE_CODES for d in slabs):
        # resample both crops + register them: one library call, no crop allocation, no wait in between
        t, quality, status, _ = _reg_ops.register_views(slabs[0], plan["matrix_diag"][0], plan["offset"][0], slabs[1], plan["matrix_diag"][1],
                                                        plan["offset"][1], plan["out_shape"], uf, None, True, device)
    else:
        crops = [resample_array(d, np.diag(plan["matrix_diag"][i]), np.array(plan["offset"][i]), plan["out_shape"], 1, np.nan, device)
                 for i, d in enumerate(slabs)]
        t, quality, status, _ = _reg_ops.register_crops(crops[0], crops[1], uf, None, True, device)
    if status == 2:
        warnings.warn("An overlap region between tiles/views is all zero or constant. Assuming identity transform.", UserWarning, stacklevel=3)
        affine, quality = param_utils.identity_transform(n), np.nan
    elif status == 1:
        raise RuntimeError("phase correlation produced no admissible shift candidate (registration.py:479-480)")
    elif status == 3:
        raise ValueError("All-NaN slice encountered")
    else:
        affine = param_utils.affine_from_translation([float(v) for v in t])
    # physical affine (get_affine_from_intrinsic_affine with both transformed crops on the output grid and carrying view 1's
    # transform): kept in numpy, its matmul / inverse chain is part of the result's rounding
    # origin / spacing as the getters read them back from the crops' coordinate arrays (o + s * arange): c[0] and c[1] - c[0]
    eff_spacing = [((o + sp * 1.0) - (o + sp * 0.0)) if nn > 1 else 1.0 for o, sp, nn in zip(plan["out_origin"], plan["out_spacing"], plan["out_shape"])]
    eff_origin = [o + sp * 0.0 for o, sp in zip(plan["out_origin"], plan["out_spacing"])]
    d_to_p = np.matmul(param_utils.affine_from_translation(np.array(eff_origin)), np.diag(eff_spacing + [1]))
    D = np.matmul(g1b.affine, d_to_p)
    affine_phys = np.matmul(D, np.matmul(affine, np.linalg.inv(D)))
    lo, up = _lean_overlap(g1, g2, tol, False)
    return {"transform": affine_phys, "quality": float(quality) if quality is not None else np.nan,
            "bbox": np.array([lo[0], up[0]])}


_lean_enabled = [True]      # tests switch the lean path off to compare it with the generic one
_PREBIN_GROUP = [8]         # tiles per binning launch / ticket of register()'s pre-binning (the pairs of a group's tiles start after it)
_native_graph = [True]      # tests: register() through mv_graph's generic graph functions instead of mvs_view_graph_prune
_native_resolution = [True]     # tests: register() through param_resolution's generic functions instead of mvs_resolve_translations


_JOB_DTYPE = np.dtype(_lib.mvs_pair_job_t)

_pair_timeline = None         # tests / bench: a list that the batched pair path fills with ((i, j), done ticket) per registered pair
_raw_crops_enabled = [True]  # tests: the batched pair path through binned copies of the tiles instead of binning inside the crop kernel
_BATCH_LANES = [8]          # context lanes / native worker threads of the batched pair path when the caller does not say (n_parallel_pairwise_regs)
_batch_enabled = [True]     # tests: compute_pairwise_registrations through the per-pair worker threads instead of mvs_register_pairs


def _pair_results_from_plan(ts, qualities, statuses, out_origin, out_spacing, out_shape, fixed_affines, lo_world, up_world):
    """The result dicts of _lean_register_pair for all pairs at once: the pixel translation of every pair turned into the
    physical affine (get_affine_from_intrinsic_affine with both crops on the fixed view's overlap grid, registration.py:1382-1474)
    -- the same matmul / inverse chain per pair, on stacked (pairs, n + 1, n + 1) arrays -- and the overlap box in world
    coordinates.  Statuses as mvs_register_crops reports them (2: constant overlap -> identity + warning; 1 / 3 raise)."""
    ne, n = ts.shape
    for k in range(ne):      # errors in pair order, like the futures of the worker pool
        if statuses[k] == 1:
            raise RuntimeError("phase correlation produced no admissible shift candidate (registration.py:479-480)")
        if statuses[k] == 3:
            raise ValueError("All-NaN slice encountered")
    const = statuses == 2
    for _ in range(int(np.count_nonzero(const))):
        warnings.warn("An overlap region between tiles/views is all zero or constant. Assuming identity transform.", UserWarning, stacklevel=4)
    affine = np.zeros((ne, n + 1, n + 1))
    affine[:, np.arange(n + 1), np.arange(n + 1)] = 1.0
    affine[:, :n, n] = np.where(const[:, None], 0.0, ts)
    quality = np.where(const, np.nan, qualities)
    # origin / spacing as the getters read them back from the crops' coordinate arrays (o + s * arange): c[0] and c[1] - c[0]
    eff_spacing = np.where(out_shape > 1, (out_origin + out_spacing * 1.0) - (out_origin + out_spacing * 0.0), 1.0)
    eff_origin = out_origin + out_spacing * 0.0
    shift = np.zeros((ne, n + 1, n + 1))
    shift[:, np.arange(n + 1), np.arange(n + 1)] = 1.0
    shift[:, :n, n] = eff_origin
    scale = np.zeros((ne, n + 1, n + 1))
    scale[:, np.arange(n), np.arange(n)] = eff_spacing
    scale[:, n, n] = 1.0
    d_to_p = np.matmul(shift, scale)
    D = np.matmul(fixed_affines, d_to_p)
    affine_phys = np.matmul(D, np.matmul(affine, np.linalg.inv(D)))
    return [{"transform": affine_phys[k], "quality": float(quality[k]), "bbox": np.array([lo_world[k], up_world[k]])} for k in range(ne)]


def _register_pairs_batched(sims, edges, transform_key, registration_binning, overlap_tolerance, pairwise_reg_func_kwargs, device,
                            n_lanes, cache, knife_check=True):
    """compute_pairwise_registrations for the common case in two library calls: plain device-resident images of one dtype whose
    transforms are pure translations, one binning for all pairs, the built-in phase correlation.  ``mvs_plan_pairs`` derives the
    crop windows and pixel affines of all pairs (host code), ``mvs_register_pairs`` registers them on ``n_lanes`` context lanes
    driven by native threads; the interpreter only assembles the inputs and turns the pixel shifts into physical affines
    (stacked matmuls).  Returns the list of result dicts, or None when the case is not covered (the caller then takes the
    per-pair path).  Same plans, same kernels, same results as _lean_register_pair (tests/test_pair_batch.py, -m gpu tests)."""
    import ctypes as C

    from . import msi_utils
    from . import spatial_image_utils as si_utils

    if not edges or any(msi_utils.is_msim(m) for m in sims):
        return None
    if any(is_device_array(sims[v].data) and sims[v].data.ready_ticket for e in edges for v in e):
        # uploads in flight (in list order): the pairs in the order in which their later tile lands, results back in edge order
        order = sorted(range(len(edges)), key=lambda k: max(edges[k]))
        if order != list(range(len(edges))):
            part = _register_pairs_batched(sims, [edges[k] for k in order], transform_key, registration_binning, overlap_tolerance,
                                           pairwise_reg_func_kwargs, device, n_lanes, cache, knife_check=knife_check)
            if part is None:
                return None
            res = [None] * len(edges)
            for k, r in zip(order, part):
                res[k] = r
            return res
    used = sorted({v for e in edges for v in e})
    sdims = si_utils.get_spatial_dims_from_sim(sims[used[0]])
    n = len(sdims)
    first = sims[used[0]]
    for v in used:
        s_ = sims[v]
        if list(s_.dims) != list(sdims) or not is_device_array(s_.data) or s_.data.dtype != first.data.dtype or s_.data.dtype not in _lib.DTYPE_CODES:
            return None
    if registration_binning is None:
        shape0, sp0 = si_utils.get_shape_from_sim(first), si_utils.get_spacing_from_sim(first)
        if any(si_utils.get_shape_from_sim(sims[v]) != shape0 or si_utils.get_spacing_from_sim(sims[v]) != sp0 for v in used[1:]):
            return None                  # the optimal binning would differ between pairs
        binning = get_optimal_registration_binning(sims[edges[0][0]], sims[edges[0][1]])
    else:
        binning = dict(registration_binning)
    if overlap_tolerance is None:
        tol = [0.0] * n
    elif isinstance(overlap_tolerance, (int, float)):
        tol = [float(overlap_tolerance)] * n
    else:
        tol = [float(overlap_tolerance.get(d, 0.0)) for d in sdims]
    bkey = tuple(sorted(binning.items()))
    do_bin = max(binning.values()) > 1
    bins = [int(binning.get(d, 1)) for d in sdims]
    dev_ok = all((sims[v].data.device & 0xff) == (device & 0xff) and sims[v].data.strides[-1] == 1 for v in used)
    if not dev_ok:
        return None
    # ---- geometry: the views as they are and as the registration binning leaves them (coordinates only: means of groups of b
    # samples, the sum divided by b as numpy's mean does it -- _bin_sim's coordinates -- without touching a voxel) ----
    geoms, geoms_b = {}, {}
    for v in used:
        geoms[v] = _TileGeom(sims[v], transform_key)
        if geoms[v].t is None:
            return None
    if do_bin:
        same_len = all(geoms[v].shape == geoms[used[0]].shape for v in used)
        if same_len:
            stacked = []
            for k, b in enumerate(bins):
                m = geoms[used[0]].shape[k] // b
                st = np.stack([geoms[v].coords[k][: m * b] for v in used])
                stacked.append(np.ascontiguousarray(np.add.reduce(st.reshape(len(used), m, b), axis=2) / b))
            for i, v in enumerate(used):
                geoms_b[v] = _TileGeom(sims[v], transform_key, like=geoms[v], coords=[stacked[k][i] for k in range(n)])
        else:
            for v in used:
                cs = []
                for k, b in enumerate(bins):
                    m = geoms[v].shape[k] // b
                    cs.append(np.ascontiguousarray(np.add.reduce(geoms[v].coords[k][: m * b].reshape(m, b), axis=1) / b))
                geoms_b[v] = _TileGeom(sims[v], transform_key, like=geoms[v], coords=cs)
        if any(min(g.shape) < 1 for g in geoms_b.values()):
            return None
    else:
        geoms_b = geoms
    slot = {v: i for i, v in enumerate(used)}
    nv, ne = len(used), len(edges)
    # ---- plans of all pairs (mvs_plan_pairs: the arithmetic of _lean_pair_plan) ----
    cptr = (C.c_void_p * (nv * n))(*[geoms_b[v].coords[k].__array_interface__["data"][0] for v in used for k in range(n)])
    clen = np.array([len(geoms_b[v].coords[k]) for v in used for k in range(n)], dtype=np.int64)
    tr = np.array([geoms_b[v].t for v in used], dtype=np.float64).reshape(nv, n)
    tolv = np.array(tol, dtype=np.float64)
    pr = np.array([[slot[a], slot[b]] for a, b in edges], dtype=np.int32).reshape(ne, 2)
    windows = np.zeros((ne, 2, 3, 2), dtype=np.int64)
    out_origin, out_spacing = np.zeros((ne, 3)), np.zeros((ne, 3))
    out_shape = np.ones((ne, 3), dtype=np.int64)
    mdiag, offs = np.zeros((ne, 2, 3)), np.zeros((ne, 2, 3))
    pstat = np.zeros(ne, dtype=np.int32)
    ptr = lambda a, ty: a.ctypes.data_as(C.POINTER(ty))
    lib = _lib.init(device & 0xff)
    rc = lib.mvs_plan_pairs(n, nv, cptr, ptr(clen, C.c_int64), ptr(tr, C.c_double), ptr(tolv, C.c_double), ne, ptr(pr, C.c_int32),
                            ptr(windows, C.c_int64), ptr(out_origin, C.c_double), ptr(out_spacing, C.c_double), ptr(out_shape, C.c_int64),
                            ptr(mdiag, C.c_double), ptr(offs, C.c_double), ptr(pstat, C.c_int32))
    if rc != 0:
        return None
    if np.any(pstat != 0):
        raise ValueError("views do not overlap")
    knife_thread = None
    if knife_check and _KNIFE_CHECK[0]:
        # pairs whose reference crop is one sample shorter (see _reference_crop_differs) take the overlap_bbox="reference" path one by
        # one; the others stay here.  One memo entry per MOSAIC geometry: a list of pair indices, looked up with a few hundred bytes.
        mkey = ("mosaic", n, tr.tobytes(), clen.tobytes(), pr.tobytes(), tolv.tobytes(),
                np.array([[geoms_b[v].coords[k][0], geoms_b[v].spacing[k]] for v in used for k in range(n)]).tobytes())
        with _KNIFE_LOCK:
            odd = _KNIFE_MEMO.get(mkey)
        if odd is None:
            # a geometry seen for the first time: the question costs ~2 ms of scipy per pair on the host (GIL-bound), the pairs
            # themselves ~0.3 ms each on the GPU -- so it is asked on a thread of its own WHILE all pairs are registered with the
            # closed-form crops below (mvs_register_pairs releases the interpreter lock), and the pairs it names, if any, are
            # registered again through the reference's sequence afterwards: the result is the reference's crop for every pair either way.
            knife_box = {}

            def ask(box=knife_box, shapes=out_shape[:, :n].copy()):
                try:
                    box["odd"] = [e for e in range(ne) if _reference_crop_differs(geoms_b[edges[e][0]], geoms_b[edges[e][1]], tol, shapes[e])]
                except BaseException as exc:      # noqa: BLE001 - handed to the caller's thread below
                    box["error"] = exc

            knife_thread = threading.Thread(target=ask, name="mvs-crop-length", daemon=True)
            knife_thread.start()
        elif odd:
            cache.reference_pairs = len(odd)
            odd_set = set(odd)
            keep = [e for e in range(ne) if e not in odd_set]
            res = [None] * ne
            if keep:
                part = _register_pairs_batched(sims, [edges[e] for e in keep], transform_key, registration_binning, overlap_tolerance,
                                               pairwise_reg_func_kwargs, device, n_lanes, cache, knife_check=False)
                if part is None:
                    return None
                for e, r in zip(keep, part):
                    res[e] = r
            for e in odd:
                i, j = edges[e]
                res[e] = register_pair_of_msims(sims[i], sims[j], transform_key, registration_binning=registration_binning,
                                                overlap_tolerance=overlap_tolerance, pairwise_reg_func_kwargs=pairwise_reg_func_kwargs,
                                                device=device, _bin_cache=cache, overlap_bbox="reference")
            return res

    def knife_fixup(res):
        """(first call of a geometry) wait for the crop-length answers, remember them, redo the pairs they name the reference's way."""
        if knife_thread is None:
            return res
        knife_thread.join()
        if "error" in knife_box:
            raise knife_box["error"]
        odd_ = knife_box["odd"]
        with _KNIFE_LOCK:
            _KNIFE_MEMO[mkey] = odd_
        if res is None:
            return None
        if odd_:
            cache.reference_pairs = len(odd_)
            if raw_before is not None and cache.raw_crops > raw_before:
                cache.raw_crops = max(raw_before, cache.raw_crops - len(odd_))      # (their closed-form results are discarded)
            for e in odd_:
                i, j = edges[e]
                res[e] = register_pair_of_msims(sims[i], sims[j], transform_key, registration_binning=registration_binning,
                                                overlap_tolerance=overlap_tolerance, pairwise_reg_func_kwargs=pairwise_reg_func_kwargs,
                                                device=device, _bin_cache=cache, overlap_bbox="reference")
        return res

    def _register_planned_pairs():
        """Jobs of all planned pairs -> mvs_register_pairs -> result dicts (None: not covered, the caller takes the per-pair path)."""
        # ---- where the crops come from.  Binned integer tiles whose crops are whole-pixel translations (every regular mosaic): straight
        # from the RAW tiles, the binning applied inside the crop kernel (job.bin) -- no binned copy of a tile is ever made, a pair reads
        # only the slabs its overlap needs.  Otherwise from binned tiles (pre-binned in groups with stream tickets, see _prebin_views). ----
        raw_crops = (do_bin and _raw_crops_enabled[0] and first.data.dtype in (np.dtype(np.uint8), np.dtype(np.uint16))
                     and bool(np.all(mdiag[:, :, :n] == 1.0)) and bool(np.all(offs[:, :, :n] == np.floor(offs[:, :, :n]))))
        tickets = {v: 0 for v in used}
        prebin_lane = None
        if raw_crops or not do_bin:
            source = {v: sims[v].data for v in used}
            # tiles still on their way to the device (device.to_device_async): a pair's lane waits for the uploads of ITS two tiles
            tickets = {v: source[v].ready_ticket for v in used}
        else:
            if all(cache.ticket_of((id(sims[v].data), bkey)) is None and (id(sims[v].data), bkey) not in cache._items for v in used):
                # (queues the binning of all views on the last context lane; None: not a regular mosaic, the views are binned one by one below)
                prebin_lane = _prebin_views([sims[v] for v in used], binning, device, cache)
            source = {}
            for v in used:
                s_ = sims[v]
                key = (id(s_.data), bkey)
                b_ = cache.get_or_compute(key, lambda s_=s_: _bin_sim(s_, binning, device), keep=s_.data)
                tickets[v] = cache.ticket_of(key) or 0
                if (b_.data.device & 0xff) != (device & 0xff) or b_.data.strides[-1] != 1 or [len(b_.coords[d]) for d in sdims] != geoms_b[v].shape:
                    return None
                source[v] = b_.data
        cache.raw_crops = ne if raw_crops else 0
        # ---- jobs: the two crop windows of every pair as mvs_view_t (strided windows into the tiles) ----
        jobs = (_lib.mvs_pair_job_t * ne)()
        ja = np.frombuffer(jobs, dtype=_JOB_DTYPE)
        k0 = 3 - n
        base = np.array([source[v].ptr for v in used], dtype=np.uint64)
        strides = np.array([[int(x) for x in source[v].strides] for v in used], dtype=np.int64).reshape(nv, n)
        item = first.data.dtype.itemsize
        code = _lib.DTYPE_CODES[first.data.dtype]
        tk = np.array([tickets[v] for v in used], dtype=np.uint64)
        scale = np.array(bins if raw_crops else [1] * n, dtype=np.int64)      # binned index -> index of the source array
        for i, name in enumerate(("fixed", "moving")):
            f = ja[name]
            vi = pr[:, i]
            a, b = windows[:, i, :n, 0] * scale, windows[:, i, :n, 1] * scale
            st = strides[vi]
            f["data"] = base[vi] + (np.sum(a * st, axis=1) * item).astype(np.uint64)
            f["dtype"] = code
            f["mem"] = _lib.MVS_MEM_DEVICE
            f["shape"][:, :k0] = 1
            f["shape"][:, k0:] = b - a
            if n == 3:
                f["stride"][:] = st
            else:       # a 2D slab is one z plane
                f["stride"][:, 0] = st[:, 0] * (b - a)[:, 0]
                f["stride"][:, 1:] = st
            f["matrix"][:] = 0.0
            f["matrix"][:, [0, 4, 8]] = 1.0
            f["matrix"][:, [(k0 + k) * 4 for k in range(n)]] = mdiag[:, i, :n]
            f["offset"][:] = 0.0
            f["offset"][:, k0:] = offs[:, i, :n]
            ja["wait_ticket"][:, i] = tk[vi]
        ja["out_shape"][:, :k0] = 1
        ja["out_shape"][:, k0:] = out_shape[:, :n]
        if raw_crops:
            ja["bin"][:, :k0] = 1
            ja["bin"][:, k0:] = scale
        if _pair_timeline is not None:
            ja["flags"][:] = 1       # every pair leaves a timed ticket of its last kernel (read below)
        upsample_factor = (pairwise_reg_func_kwargs or {}).get("upsample_factor")
        uf = (10 if n == 2 else 2) if upsample_factor is None else upsample_factor
        t3, q = np.zeros((ne, 3)), np.zeros(ne)
        status, ncand, rcs = np.zeros(ne, dtype=np.int32), np.zeros(ne, dtype=np.int32), np.zeros(ne, dtype=np.int32)
        try:
            rc = lib.mvs_register_pairs(device & 0xff, ne, jobs, n, int(uf), -1, 1, int(n_lanes), ptr(t3, C.c_double), ptr(q, C.c_double),
                                        ptr(status, C.c_int32), ptr(ncand, C.c_int32), ptr(rcs, C.c_int32))
        finally:
            if prebin_lane is not None:
                _lib.synchronize(prebin_lane)      # (done long ago unless a pair failed: nothing stays queued on the caller's tiles)
        _lib.check(rc, device & 0xff, "mvs_register_pairs")
        if _pair_timeline is not None:
            _pair_timeline.extend((tuple(edges[e]), int(ja["wait_ticket"][e, 0])) for e in range(ne))
        # ---- overlap boxes in world coordinates (_lean_overlap on the UNBINNED views) and the physical affines ----
        o = np.array([geoms[v].origin for v in used], dtype=np.float64).reshape(nv, n)
        sp = np.array([geoms[v].spacing for v in used], dtype=np.float64).reshape(nv, n)
        shp = np.array([geoms[v].shape for v in used], dtype=np.int64).reshape(nv, n)
        tw = np.array([geoms[v].t for v in used], dtype=np.float64).reshape(nv, n)
        shp = shp + np.ceil(2 * tolv / sp).astype(np.int64)
        o = o - tolv
        lo_v = o + tw
        hi_v = ((shp - 1) * 1.0 * sp + o) + tw
        lo = np.maximum(lo_v[pr[:, 0]], lo_v[pr[:, 1]])
        hi = np.minimum(hi_v[pr[:, 0]], hi_v[pr[:, 1]])
        up = 1.0 * (hi - lo) + lo
        fixed_aff = np.array([geoms_b[v].affine for v in used], dtype=np.float64)[pr[:, 0]]
        return _pair_results_from_plan(t3[:, k0:], q, status, out_origin[:, :n], out_spacing[:, :n], out_shape[:, :n], fixed_aff, lo, up)

    raw_before = getattr(cache, "raw_crops", None)
    try:
        res_all = _register_planned_pairs()
    except BaseException:
        if knife_thread is not None:
            knife_thread.join()
        raise
    return knife_fixup(res_all)


def _geom_of(sim, transform_key, cache):
    """_TileGeom of a view, shared by the pairs of one compute_pairwise_registrations call."""
    if cache is None:
        return _TileGeom(sim, transform_key)
    return cache.get_or_compute(("geom", id(sim), transform_key), lambda: _TileGeom(sim, transform_key), keep=sim)


def _select_registration_level(msim1, msim2, registration_binning, reg_res_level):
    """The three branches of registration.py:1639-1717: which resolution level of the two images is registered and which
    binning is still applied to it.  ``reg_res_level`` alone: that level, no binning; with ``registration_binning``: that
    level, the binning divided by the level's downsampling factors (which must divide it); neither / binning alone: the
    (optimal) binning split into the lowest level that divides it and a remainder.  Plain SpatialImages count as
    images with scale0 only.  Returns (sim1, sim2, remaining_binning)."""
    from . import msi_utils
    from . import spatial_image_utils as si_utils

    ms = [m if msi_utils.is_msim(m) else None for m in (msim1, msim2)]

    def level(i, scale_key):
        m = (msim1, msim2)[i]
        return msi_utils.get_sim_from_msim(m, scale=scale_key) if ms[i] is not None else m

    def scale_keys(i):
        return msi_utils.get_sorted_scale_keys(ms[i]) if ms[i] is not None else ["scale0"]

    sim1_0, sim2_0 = level(0, "scale0"), level(1, "scale0")
    sdims = si_utils.get_spatial_dims_from_sim(sim1_0)
    if reg_res_level is not None:
        scale_key = f"scale{reg_res_level}"
        if scale_key not in scale_keys(0) or scale_key not in scale_keys(1):
            raise ValueError(f"Resolution level {reg_res_level} (scale{reg_res_level}) does not exist in the multiscale image")
        sim1, sim2 = level(0, scale_key), level(1, scale_key)
        if registration_binning is None:
            return sim1, sim2, {d: 1 for d in sdims}
        factors = {d: sim1_0.sizes[d] / sim1.sizes[d] for d in sdims}
        for d in sdims:
            if d in registration_binning and registration_binning[d] % int(round(factors[d])) != 0:
                raise ValueError(
                    f"Resolution level {reg_res_level} has downsampling factor {int(round(factors[d]))} for dimension {d}, which "
                    f"is not a divisor of registration_binning[{d}]={registration_binning[d]}")
        # (like the reference, every spatial dim must be present in registration_binning here: registration.py:1674-1677)
        return sim1, sim2, {d: registration_binning[d] // int(round(factors[d])) for d in sdims}
    if registration_binning is None:
        registration_binning = get_optimal_registration_binning(sim1_0, sim2_0)
    if ms[0] is None or len(scale_keys(0)) == 1:
        # (missing dims count as 1, like get_res_level_from_binning_factors: msi_utils.py:688-773)
        return sim1_0, sim2_0, {d: registration_binning.get(d, 1) for d in sdims}
    scale_key, remaining = msi_utils.get_res_level_from_binning_factors(ms[0], registration_binning)
    if scale_key not in scale_keys(1):
        raise ValueError(f"{scale_key} does not exist in the second multiscale image")
    return level(0, scale_key), level(1, scale_key), remaining


def register_pair_of_msims(msim1, msim2, transform_key, registration_binning=None, overlap_tolerance=None,
                           pairwise_reg_func=phase_correlation_registration, pairwise_reg_func_kwargs=None, device=0,
                           _bin_cache=None, reg_res_level=None, overlap_bbox="closed_form"):
    """registration.register_pair_of_msims (registration.py:1547-2058) for pixel-space registration functions
    (the form the reference's phase correlation has): returns {"transform", "quality", "bbox"}.  ``reg_res_level`` /
    ``registration_binning`` pick the pyramid level of multiscale inputs as the reference does (registration.py:1639-1717)."""
    from . import msi_utils
    from . import spatial_image_utils as si_utils

    pairwise_reg_func_kwargs = dict(pairwise_reg_func_kwargs or {})
    sim1, sim2, registration_binning = _select_registration_level(msim1, msim2, registration_binning, reg_res_level)
    for s_ in (sim1, sim2):
        if is_device_array(s_.data):
            s_.data.wait_ready(device)       # (tiles uploaded with device.to_device_async: this lane's stream waits for them)
    sdims = si_utils.get_spatial_dims_from_sim(sim1)
    ndim = len(sdims)
    if overlap_tolerance is None:
        overlap_tolerance = {d: 0.0 for d in sdims}
    elif isinstance(overlap_tolerance, (int, float)):
        overlap_tolerance = {d: float(overlap_tolerance) for d in sdims}
    else:
        overlap_tolerance = {d: float(overlap_tolerance.get(d, 0.0)) for d in sdims}

    def binned(sim):
        if max(registration_binning.values()) <= 1:
            return sim
        key = (id(sim.data), tuple(sorted(registration_binning.items())))
        if _bin_cache is None:
            return _bin_sim(sim, registration_binning, device)
        # (the cache slot keeps sim.data alive, so its id() cannot be recycled for another tile while the cache lives)
        out = _bin_cache.get_or_compute(key, lambda: _bin_sim(sim, registration_binning, device), keep=sim.data)
        ticket = _bin_cache.ticket_of(key)
        if ticket is not None:       # pre-binned on another lane, possibly still in flight: this lane's stream waits for it (no host wait)
            _lib.check(_lib.init(device).mvs_event_wait(device, ticket), device, "mvs_event_wait")
        return out

    if overlap_bbox not in ("closed_form", "reference"):
        raise ValueError("overlap_bbox must be 'closed_form' or 'reference'")
    closed_form = overlap_bbox == "closed_form"
    reg_sims_b = [binned(sim1), binned(sim2)]
    if closed_form and pairwise_reg_func is phase_correlation_registration and set(pairwise_reg_func_kwargs) <= {"upsample_factor"} and _lean_enabled[0] \
            and all(list(s_.dims) == list(sdims) for s_ in (sim1, sim2)):
        geoms = [_geom_of(s_, transform_key, _bin_cache) for s_ in (reg_sims_b[0], reg_sims_b[1], sim1, sim2)]
        if all(g.t is not None for g in geoms):
            tol_l = [overlap_tolerance[d] for d in sdims]
            plan = _lean_pair_plan(geoms[0], geoms[1], tol_l) if _KNIFE_CHECK[0] else None
            if plan is None or not _reference_crop_differs(geoms[0], geoms[1], tol_l, plan["out_shape"]):
                return _lean_register_pair(geoms[0], geoms[1], geoms[2], geoms[3], sdims, tol_l,
                                           pairwise_reg_func_kwargs.get("upsample_factor"), transform_key, device)
            closed_form = False      # the reference registers a crop one sample shorter: this pair takes its sequence
    ov = _get_overlap_bboxes(reg_sims_b[0], reg_sims_b[1], transform_key, None, overlap_tolerance, closed_form=closed_form)
    if ov is None:
        raise ValueError("views do not overlap")
    spacings = [si_utils.get_spacing_from_sim(s) for s in reg_sims_b]
    if closed_form and _KNIFE_CHECK[0]:
        # (the generic path -- user registration functions, scaled views: the same rule as the fast paths, see _reference_crop_differs)
        ov_ref = _get_overlap_bboxes(reg_sims_b[0], reg_sims_b[1], transform_key, None, overlap_tolerance, closed_form=False)
        if ov_ref is not None:
            osp = np.maximum(np.array([spacings[0][d] for d in sdims]), np.array([spacings[1][d] for d in sdims]))
            n_cf = np.floor((ov["uppers"][0] - ov["lowers"][0]) / osp + 1)
            n_ref = np.floor((ov_ref["uppers"][0] - ov_ref["lowers"][0]) / osp + 1)
            if not np.array_equal(n_cf, n_ref):
                ov, closed_form = ov_ref, False
    lowers, uppers = ov["lowers"], ov["uppers"]
    tol = 1e-6
    reg_sims_b = [
        si_utils.sim_sel_coords(
            sim, {d: slice(lowers[i][k] - tol - spacings[i][d], uppers[i][k] + tol + spacings[i][d]) for k, d in enumerate(sdims)}
        )
        for i, sim in enumerate(reg_sims_b)
    ]
    fixed, moving = sims_to_intrinsic_coord_system(reg_sims_b[0], reg_sims_b[1], transform_key, (lowers, uppers), device)
    res = dispatch_pairwise_reg_func(pairwise_reg_func, fixed_data=fixed, moving_data=moving, device=device,
                                     **pairwise_reg_func_kwargs)
    if isinstance(res, list):   # Q2: the reference's `[zeros(ndim)]` return is unusable downstream; surface it
        raise RuntimeError("phase correlation produced no admissible shift candidate (registration.py:479-480)")
    affine = np.asarray(res["affine_matrix"], dtype=np.float64)
    affine_phys = get_affine_from_intrinsic_affine(affine, fixed, moving, transform_key, transform_key)
    ovp = _get_overlap_bboxes(sim1, sim2, transform_key, transform_key, overlap_tolerance, closed_form=closed_form)
    return {"transform": affine_phys, "quality": float(res["quality"]) if res["quality"] is not None else np.nan,
            "bbox": np.array([ovp["lowers"][0], ovp["uppers"][0]])}


def _prebin_views(sims, registration_binning, device, cache):
    """Queues the binning of all views of a regular mosaic (one shape, one spacing: every pair asks for the same binning)
    on the last context lane and puts the results into ``cache``; returns that lane's device id (to be synchronised before
    the pairs start) or None when there is nothing to do.  register() calls this before it builds the overlap graph, so
    the GPU bins the tiles while the interpreter builds and prunes the graph.  (A helper thread doing the same through
    the blocking call starves on the GIL: measured.)"""
    import threading
    from . import spatial_image_utils as si_utils
    from .device import is_device_array

    if len(sims) < 2 or any("t" in s.dims or "c" in s.dims or not is_device_array(s.data) for s in sims):
        return None
    shape0, sp0 = si_utils.get_shape_from_sim(sims[0]), si_utils.get_spacing_from_sim(sims[0])
    for s in sims[1:]:
        if si_utils.get_shape_from_sim(s) != shape0 or si_utils.get_spacing_from_sim(s) != sp0:
            return None
    binning = registration_binning if registration_binning is not None else get_optimal_registration_binning(sims[0], sims[1])
    if max(binning.values()) <= 1:
        return None
    bkey = tuple(sorted(binning.items()))
    lane_device = (device & 0xff) | (15 << 8)        # the last context lane: the pair workers take lanes from 0 upwards

    import ctypes as C

    from .device import DeviceArray
    from .transformation import shape3

    sdims = si_utils.get_spatial_dims_from_sim(sims[0])
    bins = [int(binning.get(d, 1)) for d in sdims]
    dtype = np.dtype(sims[0].data.dtype)
    if dtype not in _lib.DTYPE_CODES or any(tuple(s.data.shape) != tuple(sims[0].data.shape) or s.data.dtype != dtype for s in sims):
        return None
    nd = len(sdims)
    shape = [int(v) for v in sims[0].data.shape]
    oshape = tuple(n // b for n, b in zip(shape, bins))
    datas = [s.data.on_device(lane_device).wait_ready(lane_device) for s in sims]      # (uploads in flight: the binning lane waits for them)
    st0 = tuple(datas[0].strides)
    if any(tuple(d.strides) != st0 for d in datas):
        return None
    pool = DeviceArray.empty((len(sims),) + oshape, dtype, lane_device)     # one allocation for all binned tiles
    lib = _lib.init(lane_device)
    st = [int(v) for v in st0]
    st3 = st if nd == 3 else [st[0] * shape[0], st[0], st[1]]
    s3, b3, st3 = _lib.i64x3(shape3(shape)), _lib.i64x3([1] * (3 - nd) + bins), _lib.i64x3(st3)
    code = _lib.DTYPE_CODES[dtype]
    # the coarsened coordinates of _bin_sim (mean of each group of b: the sum divided by b, as numpy's mean does it), all views
    # of an axis in one reduction (the same additions per element as view by view)
    all_coords = {}
    for d, b, n in zip(sdims, bins, oshape):
        stack = np.stack([np.asarray(s.coords[d])[: n * b] for s in sims])
        all_coords[d] = np.add.reduce(stack.reshape(len(sims), n, b), axis=2) / b
    # the tiles are binned in groups (one library call and one ticket each): a pair starts as soon as the groups of its two
    # tiles are done, while the later groups are still being binned (mvs_event_record / mvs_event_wait, no host wait)
    n_v = len(sims)
    item = int(np.prod(oshape)) * dtype.itemsize
    group = max(1, _PREBIN_GROUP[0])
    tickets = []
    for g0 in range(0, n_v, group):
        g1 = min(g0 + group, n_v)
        ins = (C.c_void_p * (g1 - g0))(*[datas[i].ptr for i in range(g0, g1)])
        outs = (C.c_void_p * (g1 - g0))(*[pool.ptr + i * item for i in range(g0, g1)])
        _lib.check(lib.mvs_bin_mean_batch_async(lane_device, g1 - g0, ins, code, s3, st3, b3, outs), lane_device, "mvs_bin_mean_batch_async")
        tk = C.c_uint64(0)
        _lib.check(lib.mvs_event_record(lane_device, C.byref(tk)), lane_device, "mvs_event_record")
        tickets.append(int(tk.value))
    tdims = tuple(sdims)
    ostr = tuple(int(np.prod(oshape[i + 1:])) for i in range(nd))
    for i, s in enumerate(sims):
        out = DeviceArray(pool._buf, pool.ptr + i * item, oshape, ostr, dtype, lane_device)
        coords = {d: all_coords[d][i] for d in sdims}
        # (coordinates of the right lengths by construction: no checks)
        binned = si_utils.SpatialImage._from_parts(out, tdims, coords, {"transforms": dict(s.attrs.get("transforms", {}))}, None)
        cache.put((id(s.data), bkey), binned, keep=(s.data, datas[i]), ticket=tickets[i // group])
    return lane_device


def compute_pairwise_registrations(msims, edges, transform_key, registration_binning=None, overlap_tolerance=0.0,
                                   pairwise_reg_func=phase_correlation_registration, pairwise_reg_func_kwargs=None,
                                   pairwise_executor=None, device=0, host_threads=None, _bin_cache=None, reg_res_level=None,
                                   overlap_bbox="closed_form"):
    """registration.compute_pairwise_registrations (registration.py:2622-2714): either hand all edges to a
    user ``pairwise_executor(msims, edges, register_kwargs)`` or loop over them on one device."""
    register_kwargs = dict(transform_key=transform_key, registration_binning=registration_binning,
                           overlap_tolerance=overlap_tolerance, pairwise_reg_func=pairwise_reg_func,
                           pairwise_reg_func_kwargs=pairwise_reg_func_kwargs)
    if reg_res_level is not None:       # (absent by default: executors written for the round-1 keyword set keep working)
        register_kwargs["reg_res_level"] = reg_res_level
    if overlap_bbox != "closed_form":
        register_kwargs["overlap_bbox"] = overlap_bbox
    if pairwise_executor is not None:
        results = pairwise_executor(msims, list(edges), register_kwargs)
        if len(results) != len(edges):
            raise ValueError("pairwise_executor must return one result per edge")
        return results
    cache = _bin_cache if _bin_cache is not None else _BinCache()
    edges = list(edges)
    _host_threads_default = host_threads is None
    if host_threads is None:
        host_threads = 16
    n_threads = max(1, min(int(host_threads), len(edges), 16))   # 16 = context lanes per GPU (MVS_MAX_LANES)
    if _batch_enabled[0] and _lean_enabled[0] and overlap_bbox == "closed_form" and pairwise_reg_func is phase_correlation_registration and reg_res_level is None \
            and set(pairwise_reg_func_kwargs or {}) <= {"upsample_factor"}:
        # all pairs in two library calls (plans + registrations on native worker threads): no interpreter in the pair loop
        # (native workers spin inside the HIP runtime between launches: 6-12 of them reach the GPU's pair throughput, 16 exhaust a
        # 16-core CPU quota and are throttled to half speed -- profiles/round5_lanes.txt; the default is 8)
        batched = _register_pairs_batched(msims, edges, transform_key, registration_binning, overlap_tolerance, pairwise_reg_func_kwargs,
                                          device, min(n_threads, _BATCH_LANES[0] if _host_threads_default else 12), cache)
        if batched is not None:
            return batched
    if n_threads == 1:
        return [register_pair_of_msims(msims[i], msims[j], device=device, _bin_cache=cache, **register_kwargs) for i, j in edges]
    # Pairs are independent.  Every call into libmvs_hip.so releases the GIL; each worker thread drives its own
    # context lane (stream + scratch + lock) of the GPU, so the kernels of one pair fill the host round trips and
    # the Python glue of another.  Results keep the order of `edges`.
    pool, lanes, lane_lock = _pair_pool(n_threads)

    def work(i, j):
        # each worker thread owns one context lane of the GPU (`device | lane << 8`: own stream, scratch and lock)
        tid = threading.get_ident()
        with lane_lock:
            lane = lanes.setdefault(tid, len(lanes))
        return register_pair_of_msims(msims[i], msims[j], device=(device & 0xff) | (lane << 8), _bin_cache=cache, **register_kwargs)

    futs = [pool.submit(work, i, j) for i, j in edges]
    return [f.result() for f in futs]


_PAIR_POOLS = {}
_PAIR_POOLS_LOCK = threading.Lock()


def _pair_pool(n_threads):
    """The worker threads of compute_pairwise_registrations live across calls (starting 16 threads costs ~10 ms of the
    first pairs' time in every call otherwise); one pool per thread count, with its thread -> lane table."""
    from concurrent.futures import ThreadPoolExecutor

    with _PAIR_POOLS_LOCK:
        entry = _PAIR_POOLS.get(n_threads)
        if entry is None:
            from .executors import pin_worker_thread

            # the workers keep to one compact block of CPUs (see executors.pin_process_to_compact_cpus: spread over both sockets
            # of a large host they lose ~10 % of the pairwise wall); the caller's own threads are not touched
            entry = _PAIR_POOLS[n_threads] = (ThreadPoolExecutor(max_workers=n_threads, thread_name_prefix="mvs-pair",
                                                                 initializer=pin_worker_thread), {}, threading.Lock())
        return entry


def _set_event():
    import threading

    e = threading.Event()
    e.set()
    return e


_FINISHED = _set_event()


class _BinCache:
    """Binned tiles shared by the pairs of one compute_pairwise_registrations call (each tile is binned once).
    Thread safe: the first thread asking for a key computes it, the others wait for that result."""

    def __init__(self):
        import threading

        self._lock = threading.Lock()
        self._items = {}
        self.hits = self.misses = 0      # (tests: the pre-binned tiles of register() must be found by every pair)
        self.raw_crops = 0               # pairs whose crops were taken from the raw tiles (no binned copies at all)
        self.reference_pairs = 0         # pairs the default crop rule sent through the reference's Qhull sequence (N - 1 samples)

    def put(self, key, value, keep=None, ticket=None):
        """Store a finished value (``keep``: the objects whose id() is part of ``key``; the slot holds them alive).  ``ticket``:
        the value's device memory is complete once this mvs_event_record ticket has passed -- a lane that reads it first makes
        its stream wait for the ticket (``ticket_of``)."""
        slot = {"event": _FINISHED, "value": value, "error": None, "keep": keep, "ticket": ticket}      # (one shared, already set event)
        with self._lock:
            self._items[key] = slot

    def ticket_of(self, key):
        slot = self._items.get(key)
        return None if slot is None else slot.get("ticket")

    def get_or_compute(self, key, fn, keep=None):
        """``keep``: the object whose id() is part of ``key``; the slot holds a reference to it."""
        import threading

        with self._lock:
            slot = self._items.get(key)
            owner = slot is None
            if key[0] != "geom":          # (the statistics count the binned tiles)
                if owner:
                    self.misses += 1
                else:
                    self.hits += 1
            if owner:
                slot = self._items[key] = {"event": threading.Event(), "value": None, "error": None, "keep": keep}
        if owner:
            try:
                slot["value"] = fn()
            except BaseException as e:   # noqa: BLE001 - re-raised in every waiter
                slot["error"] = e
            slot["event"].set()
        else:
            slot["event"].wait()
        if slot["error"] is not None:
            raise slot["error"]
        return slot["value"]


def resolve_translations(n_views, edges, pair_results, reference_view=0, weights=None):
    """Minimal groupwise resolution for translation-only pairwise results: least squares on
    tau_j - tau_i = -d_ij with tau_ref = 0 (each connected component gets its own reference).

    Not the reference's method (that is ``param_resolution.groupwise_resolution``); kept as
    ``groupwise_resolution_method="linear"``: the fixed point of the reference's bead sweeps on a translation graph
    when no edge is pruned, in one solve."""
    ndim = np.asarray(pair_results[0]["transform"]).shape[0] - 1 if pair_results else 0
    params = [np.eye(ndim + 1) for _ in range(n_views)]
    if not edges:
        return params
    adj = {v: set() for v in range(n_views)}
    for i, j in edges:
        adj[i].add(j)
        adj[j].add(i)
    seen = set()
    for start in range(n_views):
        if start in seen:
            continue
        comp, stack = [], [start]
        seen.add(start)
        while stack:
            v = stack.pop()
            comp.append(v)
            for w in adj[v]:
                if w not in seen:
                    seen.add(w)
                    stack.append(w)
        comp = sorted(comp)
        if len(comp) == 1:
            continue
        ref = reference_view if reference_view in comp else comp[0]
        idx = {v: k for k, v in enumerate(comp)}
        rows, rhs, wts = [], [], []
        for k, (i, j) in enumerate(edges):
            if i not in idx:
                continue
            r = np.zeros(len(comp))
            r[idx[j]] = 1.0
            r[idx[i]] = -1.0
            rows.append(r)
            rhs.append(-np.asarray(pair_results[k]["transform"])[:ndim, ndim])
            wts.append(1.0 if weights is None else float(weights[k]))
        A = np.array(rows) * np.sqrt(np.array(wts))[:, None]
        B = np.array(rhs) * np.sqrt(np.array(wts))[:, None]
        anchor = np.zeros((1, len(comp)))
        anchor[0, idx[ref]] = 1e3
        A = np.vstack([A, anchor])
        B = np.vstack([B, np.zeros((1, ndim))])
        tau, *_ = np.linalg.lstsq(A, B, rcond=None)
        tau -= tau[idx[ref]]
        for v in comp:
            params[v] = param_utils.affine_from_translation(tau[idx[v]])
    return params


def register(msims, transform_key=None, reg_channel_index=None, reg_channel=None, new_transform_key=None,
             registration_binning=None, overlap_tolerance=0.0, pairwise_reg_func=phase_correlation_registration,
             pairwise_reg_func_kwargs=None, groupwise_resolution_method="global_optimization",
             groupwise_resolution_kwargs=None, pre_registration_pruning_method="alternating_pattern",
             pre_reg_pruning_method_kwargs=None,
             post_registration_do_quality_filter=False, post_registration_quality_threshold=0.2, pairs=None,
             n_parallel_pairwise_regs=None, pairwise_executor=None, return_dict=False, device=0, reg_res_level=None,
             overlap_bbox="closed_form"):
    """Register views to a common coordinate system (registration.register, registration.py:2227-2620).

    ``reg_res_level`` / ``registration_binning`` select the pyramid level of multiscale inputs per pair exactly as the
    reference does (registration.py:1639-1717, 2236, 2525): a level alone, a level plus the remaining binning, or -- by
    default -- the lowest level that divides the (optimal) binning.  The overlap graph is always built on scale0.

    Flow as in the reference: (1) overlap graph, (2) pairwise registrations of the selected edges,
    (3) groupwise resolution, (4) write ``new_transform_key`` (rebased on ``transform_key``).
    Accepts MultiscaleSpatialImages or SpatialImages (numpy- or DeviceArray-backed).  Host-side
    logic as in the reference: the overlap graph holds the intersection volume of every pair of nearby views (closed
    form for axis-aligned pairs, scipy's halfspace intersection otherwise) and is pruned by
    ``pre_registration_pruning_method`` (None, "alternating_pattern" (default), "shortest_paths_overlap_weighted",
    "otsu_threshold_on_overlap", "keep_axis_aligned").  The groupwise resolution is
    ``overlap_bbox``: how the overlap crop of a pair is sized.  "closed_form" (default): the intersection of the two world boxes
    of axis-aligned views in closed form -- N samples along an axis the views share exactly -- EXCEPT for the pairs on which the
    reference's own sequence (linprog feasible point + Qhull halfspace intersection, registration.py:229-239, 314-316) lands on
    N - 1: its vertices carry Qhull's round-off of ~1e-13, so ``floor((upper - lower) / spacing + 1)`` comes out as N or N - 1,
    and which one no closed form predicts.  The sequence is therefore asked once per pair geometry (~3 ms per pair, memoised: a
    register + fuse loop over one mosaic pays it in its first call; ``MVS_KNIFE_CHECK=0`` switches the question off), and only
    the N - 1 pairs (0 of 144 on the north star, 0 of 64 on C3, C1's one pair) leave the batched path for the reference's
    sequence.  "reference": that sequence for every pair (per-pair host path, no batching).
    ``param_resolution.groupwise_resolution`` (``groupwise_resolution_method``: "global_optimization" (default),
    "shortest_paths", a callable, or "linear" for the plain least-squares solve of ``resolve_translations``;
    ``groupwise_resolution_kwargs`` e.g. ``{"transform": "rigid", "reference_view": 0}``)."""
    from . import msi_utils, mv_graph, param_resolution
    from . import spatial_image_utils as si_utils

    if transform_key is None:
        raise ValueError("transform_key must be provided")
    sims = [msi_utils.get_sim_from_msim(m) if msi_utils.is_msim(m) else m for m in msims]
    if "c" in sims[0].dims and sims[0].sizes["c"] > 1:
        if reg_channel is None and reg_channel_index is None:
            raise Exception("Please choose a registration channel.")
        ci = reg_channel_index if reg_channel is None else int(np.nonzero(sims[0].coords["c"] == reg_channel)[0][0])
    else:
        ci = 0

    def channel_of(s):
        return s.isel({"c": ci}) if "c" in s.dims else s

    sims_reg = [channel_of(s) for s in sims]
    if _HOST_UPLOAD_ASYNC[0] and pairwise_executor is None and not any(msi_utils.is_msim(m) and len(m.keys()) > 1 for m in msims) \
            and reg_res_level is None and pairwise_reg_func is phase_correlation_registration \
            and all((isinstance(s.data, np.ndarray) or type(s.data).__name__ in ("ZarrArray", "ZarrView"))
                    and list(s.dims) == list(si_utils.get_spatial_dims_from_sim(s)) for s in sims_reg) \
            and sum(int(np.prod(s.data.shape)) * np.dtype(s.data.dtype).itemsize for s in sims_reg) >= (256 << 20) and _lib.device_count() > 0 \
            and all(np.dtype(s.data.dtype) in _lib.DTYPE_CODES for s in sims_reg):
        # plain host numpy tiles (or windows of Zarr arrays) -- what a user of the reference hands over: staged through pinned buffers and uploaded on the copy stream
        # while the overlap graph is built and the first pairs are registered (north star: 1.7-2.0 s of one synchronous pageable upload
        # per tile before; streaming.upload_host_sims_async).  The caller's images are only read; results are written to them below.
        from .streaming import upload_host_sims_async

        sims_reg = upload_host_sims_async(sims_reg, device & 0xff)
    nt = sims_reg[0].sizes.get("t", 1) if "t" in sims_reg[0].dims else 1
    # pyramid levels take part only when an image has more than scale0 or a level is asked for (registration.py:1639-1717)
    multiscale = reg_res_level is not None or any(msi_utils.is_msim(m) and len(m.keys()) > 1 for m in msims)
    if multiscale:
        levels_reg = [[channel_of(msi_utils.get_sim_from_msim(m, scale=k)) for k in msi_utils.get_sorted_scale_keys(m)]
                      if msi_utils.is_msim(m) else [channel_of(m)] for m in msims]

    # tiles of a regular mosaic are binned while the graph is built (the GPU would idle otherwise)
    # (binned copies of the tiles are only made when the batched pair path cannot take its crops from the raw tiles: it then
    # queues the binning of all views itself, in groups with stream tickets -- see _register_pairs_batched / _prebin_views)
    bin_cache, prebin = None, None
    if pairwise_executor is None and nt == 1 and not multiscale:
        bin_cache = _BinCache()

    # (1) graph
    sps = [si_utils.get_stack_properties_from_sim(s) for s in sims_reg]
    affs = [param_utils.select_time(si_utils.get_affine_from_sim(s, transform_key), 0) for s in sims_reg]
    # overlap graph of the views (mv_graph.py:35-180) and its pruning to the pairs that get registered (registration.py:2467-2491)
    tol = overlap_tolerance
    if tol is not None and not isinstance(tol, dict):
        tol = {d: float(tol) for d in sps[0]["spacing"]}
    try:
        views = [dict(sp, transform=a) for sp, a in zip(sps, affs)]
        # axis-aligned views + the default pruning: graph, overlap volumes and pruning in one library call (host code)
        edges = mv_graph.registration_edges_native(views, tol, pairs, pre_registration_pruning_method, pre_reg_pruning_method_kwargs) \
            if _native_graph[0] else None
        if edges is None:
            g_views = mv_graph.build_view_adjacency_graph(views, overlap_tolerance=tol, pairs=pairs)
            g_views = mv_graph.prune_view_adjacency_graph(g_views, pre_registration_pruning_method, pre_reg_pruning_method_kwargs)
            edges = [tuple(sorted(e)) for e in g_views.edges()]
    except BaseException:
        if prebin is not None:
            _lib.synchronize(prebin)     # graph building raised: nothing stays queued on tiles the caller may free
        raise
    # (no wait here: every lane that reads a binned tile makes its stream wait for the tile's ticket, see register_pair_of_msims;
    # the pre-binning lane is the last context lane, a pair worker that reuses it queues behind the binning anyway)

    # (2) pairwise registrations per time point
    params_t, all_results, resolution_info = [], [], []
    for it in range(nt):
        # per-time-point fields are shallow copies with their OWN transforms dict: the caller's images are never modified
        # (t-stacked affines of an image without a t axis would otherwise collapse to one time point for good)
        def field_of(s):
            tr = s.attrs.get("transforms", {})
            if (pairwise_executor is None or getattr(pairwise_executor, "reads_only", False)) and "t" not in s.dims \
                    and all(np.ndim(v) == 2 for v in tr.values()):
                # nothing to select and the built-in pair path (also behind sharding.ShardedPairExecutor, which says so) only reads:
                # the image itself; any other user executor gets copies
                return s
            f = s.isel({"t": it}) if "t" in s.dims else s.copy()
            f.attrs["transforms"] = {k: param_utils.select_time(v, it) for k, v in s.attrs.get("transforms", {}).items()}
            return f

        if multiscale:
            fields = []
            for lv in levels_reg:
                fl = [field_of(s) for s in lv]
                fields.append(msi_utils.MultiscaleSpatialImage(fl, fl[0].attrs["transforms"]))
        else:
            fields = [field_of(s) for s in sims_reg]
        try:
            results = compute_pairwise_registrations(
                fields, edges, transform_key, registration_binning, overlap_tolerance, pairwise_reg_func,
                pairwise_reg_func_kwargs, pairwise_executor, device,
                host_threads=n_parallel_pairwise_regs,
                _bin_cache=bin_cache, reg_res_level=reg_res_level, overlap_bbox=overlap_bbox,
            )
        finally:
            if prebin is not None:
                # (done long ago unless a tile took part in no pair or a pair raised: nothing stays queued on the caller's tiles)
                _lib.synchronize(prebin)
                prebin = None
        keep = list(range(len(edges)))
        if post_registration_do_quality_filter:
            # mv_graph.filter_edges removes edges with quality < threshold only: a NaN quality (constant overlap ->
            # identity transform) stays in the graph, as in the reference
            keep = [k for k in keep if not (results[k]["quality"] < post_registration_quality_threshold)]
        # (3) groupwise resolution (registration.py:2560-2580 -> param_resolution.groupwise_resolution)
        if groupwise_resolution_method == "linear":
            params_t.append(resolve_translations(len(sims), [edges[k] for k in keep], [results[k] for k in keep]))
            resolution_info.append(None)
        else:
            fast = None
            gkw = dict(groupwise_resolution_kwargs or {})
            if groupwise_resolution_method == "global_optimization" and _native_resolution[0] and keep and \
                    set(gkw) <= {"reference_view", "transform", "max_iter", "rel_tol", "abs_tol"}:
                # a connected translation mosaic whose sweeps end below abs_tol: the whole resolution in one library call (host code)
                fast = param_resolution.resolve_translations_native(
                    len(sims), [edges[k] for k in keep], [results[k] for k in keep],
                    [[sp["spacing"][d] for d in sp["spacing"]] for sp in sps], **gkw)
            if fast is not None:
                params_t.append(fast[0])
                resolution_info.append(fast[1])
                all_results.append(results)
                continue
            g = param_resolution.RegGraph(range(len(sims)), {v: sps[v] for v in range(len(sims))})
            for k in keep:
                g.add_edge(edges[k][0], edges[k][1], results[k]["transform"], quality=results[k]["quality"], bbox=results[k]["bbox"])
            p_nodes, info = param_resolution.groupwise_resolution(g, groupwise_resolution_method, **dict(groupwise_resolution_kwargs or {}))
            params_t.append([p_nodes[v] for v in range(len(sims))])
            resolution_info.append(info)
        all_results.append(results)
    params = [np.stack([params_t[it][v] for it in range(nt)], axis=0) if "t" in sims_reg[0].dims else params_t[0][v]
              for v in range(len(sims))]

    # (4) write back
    if new_transform_key is not None:
        for m, p in zip(msims, params):
            if msi_utils.is_msim(m):
                msi_utils.set_affine_transform(m, p, transform_key=new_transform_key, base_transform_key=transform_key)
            else:
                si_utils.set_sim_affine(m, p, new_transform_key, base_transform_key=transform_key)
    if return_dict:
        return {"params": params, "bin_cache_stats": None if bin_cache is None else {"hits": bin_cache.hits, "misses": bin_cache.misses,
                                                                                        "pairs_with_raw_crops": bin_cache.raw_crops,
                                                                                        "pairs_on_reference_sequence": bin_cache.reference_pairs},
                "groupwise_resolution": {"info": resolution_info},
                "pairwise_registration": {"edges": edges, "results": all_results,
                                          "metrics": {"qualities": {e: r["quality"] for e, r in zip(edges, all_results[0])}}}}
    return params
