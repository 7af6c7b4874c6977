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
