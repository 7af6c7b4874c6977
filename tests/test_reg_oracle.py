"""CPU tests pinning oracle/reg_oracle.py: (i) against scikit-image 0.18.3 vectors (same upsampled-DFT
phase correlation and SSIM formula), (ii) against restated tolerance tests of the reference
(T/test_registration.py:241-336: an artificial translation is recovered to atol 0.1; :87-111 atol 1.5)."""
import os

import numpy as np
import pytest
from scipy import ndimage

from oracle import reg_oracle as ro

GOLD = np.load(os.path.join(os.path.dirname(__file__), "golden", "skimage018_pcc.npz"))


@pytest.mark.parametrize("name", ["2d_a", "2d_b", "3d_a", "3d_b", "2d_u1"])
def test_pcc_matches_skimage018(name):
    a, b, up = GOLD[name + "_a"], GOLD[name + "_b"], int(GOLD[name + "_up"])
    s = ro.phase_cross_correlation(a, b, upsample_factor=up, normalization=None)
    np.testing.assert_allclose(np.asarray(s, dtype=np.float64), GOLD[name + "_shift"], atol=1e-6)


@pytest.mark.parametrize("name", ["ssim2d", "ssim3d", "ssim3d_w3"])
def test_ssim_matches_skimage018(name):
    x, y = GOLD[name + "_x"], GOLD[name + "_y"]
    v = ro.structural_similarity(x, y, data_range=float(GOLD[name + "_dr"]), win_size=int(GOLD[name + "_win"]))
    assert abs(v - float(GOLD[name + "_val"])) < 1e-12


def _pair(shape, shift, seed=0, nan_border=0):
    rng = np.random.default_rng(seed)
    pad = 12
    big = ndimage.gaussian_filter(rng.random(tuple(s + 2 * pad for s in shape)), 1.0).astype(np.float32)
    a = np.ascontiguousarray(big[tuple(slice(pad, pad + s) for s in shape)])
    b = np.ascontiguousarray(big[tuple(slice(pad + d, pad + d + s) for d, s in zip(shift, shape))])
    if nan_border:
        b = b.copy()
        b[..., :nan_border] = np.nan
    return a, b


@pytest.mark.parametrize("shape,shift", [((60, 104), (3, -5)), ((41, 97), (-6, 2)), ((24, 64, 56), (2, -3, 4))])
def test_known_translation_is_recovered(shape, shift):
    a, b = _pair(shape, shift)
    res = ro.phase_correlation_registration(a, b)
    t = res["affine_matrix"][:-1, -1]
    # the affine maps fixed px -> moving px: content displaced by +shift in `b` sits at x - shift
    np.testing.assert_allclose(t, -np.asarray(shift), atol=0.5)     # cf. T/test_registration.py:332-336, 639-643
    assert res["quality"] > 0.9


def test_nan_border_uses_intersection_and_zero_candidate():
    a, b = _pair((50, 90), (2, 3), nan_border=4)
    res = ro.phase_correlation_registration(a, b, return_debug=True)
    assert res["debug"]["region_mode"] == "intersection"
    assert len(res["debug"]["shift_candidates"]) == 3       # Q1: masked variant adds a zero-shift candidate
    np.testing.assert_allclose(res["affine_matrix"][:-1, -1], (-2, -3), atol=1.5)   # T/test_registration.py:87-111


def test_rescale_keeps_float32_and_range():
    a, _ = _pair((30, 40), (0, 0))
    r = ro.rescale_intensity_01(a * 1000 + 7)
    assert r.dtype == np.float32 and r.min() == 0 and r.max() == 1


def test_binning_heuristic_matches_survey_cases():
    # 256x512x512 isotropic -> {z:2,y:1,x:1}; 512^3 -> all 2
    assert ro.get_optimal_registration_binning((256, 512, 512), (256, 512, 512), (1, 1, 1), (1, 1, 1)) == {"z": 2, "y": 1, "x": 1}
    assert ro.get_optimal_registration_binning((512,) * 3, (512,) * 3, (1, 1, 1), (1, 1, 1)) == {"z": 2, "y": 2, "x": 2}


# ---- round 2: pins for quirks Q1 / Q4 / Q5 and the "phase" normalisation (tests/golden/skimage018_round2.npz) ----------
GOLD2 = np.load(os.path.join(os.path.dirname(__file__), "golden", "skimage018_round2.npz"))


@pytest.mark.parametrize("name", ["q1_2d", "q1_3d", "q1_2d_fixed"])
def test_q1_masked_variant_with_inverted_masks_is_the_zero_shift(name):
    """registration.py:433-443 hands NaN-holding images and INVERTED masks to the masked phase correlation.  Executed
    with scikit-image 0.18.3: the normalised correlation is identically zero, every position is a maximum, the mean
    position is the zero shift.  (0.18.3 itself correlates axes (0, 1) only, so for 3D input its third component is the
    centre of the untouched axis; its correlation routine run over all axes -- what the n-D releases do -- gives 0.)"""
    a = GOLD2[name + "_a"]
    assert float(GOLD2[name + "_xcorr_absmax"]) == 0.0
    np.testing.assert_array_equal(np.abs(GOLD2[name + "_shift_allaxes"]), np.zeros(a.ndim))
    if a.ndim == 2:
        np.testing.assert_array_equal(GOLD2[name + "_shift"], np.zeros(2))
    else:
        np.testing.assert_array_equal(GOLD2[name + "_shift"], [0.0, 0.0, (a.shape[2] - 1) / 2])
    # ... and that is the candidate the oracle (and the product) append
    res = ro.phase_correlation_registration(a, GOLD2[name + "_b"], return_debug=True)
    cands = res["debug"]["shift_candidates"]
    assert len(cands) == 3
    np.testing.assert_array_equal(cands[2], np.zeros(a.ndim))


@pytest.mark.parametrize("name", ["rs_f32", "rs_u16like"])
def test_q5_rescale_intensity_values_match_skimage018(name):
    """0.18.3 returns float64 that holds the float32 arithmetic's values ((im - min) / (max - min) with the range as
    Python floats); releases >= 0.19 keep float32.  The oracle's float32 result must be those values exactly."""
    want = GOLD2[name + "_out"]
    assert str(GOLD2[name + "_out_dtype"]) == "float64"
    got = ro.rescale_intensity_01(GOLD2[name + "_in"])
    assert got.dtype == np.float32
    np.testing.assert_array_equal(np.isnan(got), np.isnan(want))
    m = ~np.isnan(want)
    np.testing.assert_array_equal(got[m].astype(np.float64), want[m])


@pytest.mark.parametrize("name", ["ssimf32_2d", "ssimf32_3d", "ssimf32_3d_w5"])
def test_q4_float32_ssim_within_tolerance_of_the_float64_value(name):
    """0.18.3 converts to float64 before filtering; >= 0.19 -- and the oracle -- filter float32 images in float32.
    Stated tolerance of the float32 evaluation against the executed float64 value: 2e-5 absolute."""
    x, y = GOLD2[name + "_x"], GOLD2[name + "_y"]
    assert x.dtype == np.float32
    v32 = ro.structural_similarity(x, y, data_range=float(GOLD2[name + "_dr"]), win_size=int(GOLD2[name + "_win"]))
    v64 = ro.structural_similarity(x.astype(np.float64), y.astype(np.float64), data_range=float(GOLD2[name + "_dr"]),
                                   win_size=int(GOLD2[name + "_win"]))
    assert abs(v64 - float(GOLD2[name + "_val"])) < 1e-12
    assert abs(v32 - float(GOLD2[name + "_val"])) < 2e-5


@pytest.mark.parametrize("name", ["ph_2d", "ph_2d_odd", "ph_3d"])
def test_phase_normalisation_matches_skimage018_machinery(name):
    """normalization="phase" does not exist in 0.18.3; the fixture applies the one published line
    (image_product /= max(|image_product|, 100 eps)) and runs everything else -- transforms, peak search, wrap-around,
    _upsampled_dft refinement -- with 0.18.3's own code."""
    a, b, up = GOLD2[name + "_a"], GOLD2[name + "_b"], int(GOLD2[name + "_up"])
    s, dbg = ro.phase_cross_correlation(a, b, upsample_factor=up, normalization="phase", return_debug=True)
    np.testing.assert_array_equal(dbg["peak_index"], GOLD2[name + "_peak"])
    np.testing.assert_allclose(np.asarray(s, dtype=np.float64), GOLD2[name + "_shift"], atol=1e-6)
