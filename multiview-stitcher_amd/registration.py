"""Pairwise registration on the HIP backend (mirror of the reference's ``registration`` API for the hot path).

``phase_correlation_registration`` == registration.phase_correlation_registration
    (src/multiview_stitcher/registration.py:353-565): the host keeps the reference's control flow
    (candidate enumeration, quirks Q1-Q3, nanargmax) and every voxel-sized step runs in
    libmvs_hip.so: mvs_rescale_intensity, mvs_phasecorr, mvs_score_candidates.
"""

from __future__ import annotations

import logging
import warnings

import numpy as np

from . import _reg_ops, param_utils
from .device import is_device_array

logger = logging.getLogger(__name__)


def _as_array(x):
    return x.data if hasattr(x, "dims") and hasattr(x, "coords") else x


def _enumerate_candidates(shift_candidates, shape, ndim):
    """registration.py:453-477: per shift estimate and per axis with a non-zero shift the four variants
    {s, -s, -(s - N), -s - N}; keep |t| < max over ALL dims of the image shape."""
    max_shift_per_dim = np.max([shape, shape])
    t_candidates = []
    for shift_candidate in shift_candidates:
        for s in np.ndindex(tuple([1 if shift_candidate[d] == 0 else 4 for d in range(ndim)])):
            t = []
            for d in range(ndim):
                if s[d] == 0:
                    t.append(shift_candidate[d])
                elif s[d] == 1:
                    t.append(-shift_candidate[d])
                elif s[d] == 2:
                    t.append(-(shift_candidate[d] - shape[d]))
                elif s[d] == 3:
                    t.append(-shift_candidate[d] - shape[d])
            if np.max(np.abs(t)) < max_shift_per_dim:
                t_candidates.append(t)
    return t_candidates


def phase_correlation_registration(fixed_data, moving_data, disambiguate_region_mode=None, device=0,
                                   return_debug=False, **skimage_phase_corr_kwargs):
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

    # normalise (registration.py:381-389); the kernel also reports nanmin/nanmax/#valid of the INPUT
    im0, min0, max0, nvalid0 = _reg_ops.rescale_intensity(im0, device, out_on_device=on_dev)
    im1, min1, max1, nvalid1 = _reg_ops.rescale_intensity(im1, device, out_on_device=on_dev)
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
    for normalization in ["phase", None]:
        s, dbg = _reg_ops.phase_cross_correlation(im0, im1, upsample_factor, normalization, device, return_debug=True)
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

    ssim, spear, codes = _reg_ops.score_candidates(im0, im1, t_candidates, disambiguate_region_mode, data_range, im1_min, device)
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
    return binning


def _smoke(device=0):
    """Used by __graft_entry__.smoke(): one small pairwise registration checked against the oracle."""
    from oracle import reg_oracle as ro
    from scipy import ndimage

    rng = np.random.default_rng(0)
    big = ndimage.gaussian_filter(rng.random((40, 84, 76)), 1.0).astype(np.float32)
    a = np.ascontiguousarray(big[8:32, 10:74, 10:66])
    b = np.ascontiguousarray(big[9:33, 8:72, 13:69])
    want = ro.phase_correlation_registration(a, b)
    got = phase_correlation_registration(a, b, device=device)
    assert np.array_equal(got["affine_matrix"], want["affine_matrix"]), (got, want)
    assert abs(got["quality"] - want["quality"]) < 1e-5
    print("smoke: pairwise registration shift", got["affine_matrix"][:-1, -1], "quality", got["quality"], "matches oracle")
