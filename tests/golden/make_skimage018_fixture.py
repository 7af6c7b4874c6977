"""Generates tests/golden/skimage018_pcc.npz -- run with /opt/conda/bin/python3.9 (scikit-image 0.18.3).

scikit-image 0.26 (what the reference pins) is not installable here; 0.18.3 carries the same
Guizar-Sicairos upsampled-DFT phase correlation (== normalization=None in 0.26) and the same SSIM
formula (computed in float64).  The vectors pin oracle/reg_oracle.py's restatement of both.
"""
import numpy as np
import skimage
from scipy import ndimage
from skimage.metrics import structural_similarity
from skimage.registration import phase_cross_correlation

assert skimage.__version__.startswith("0.18"), skimage.__version__
rng = np.random.default_rng(42)
out = {}
cases = [("2d_a", (64, 104), (3, -5), 10), ("2d_b", (53, 80), (-7, 2), 10), ("3d_a", (27, 40, 36), (2, -3, 4), 2),
         ("3d_b", (16, 33, 52), (-1, 5, -6), 2), ("2d_u1", (48, 48), (4, 4), 1)]
for name, shape, shift, up in cases:
    pad = 10
    big = ndimage.gaussian_filter(rng.random(tuple(s + 2 * pad for s in shape)), 1.5).astype(np.float32)
    a = big[tuple(slice(pad, pad + s) for s in shape)]
    b = big[tuple(slice(pad + d, pad + d + s) for d, s in zip(shift, shape))]
    a = np.ascontiguousarray(a)
    b = np.ascontiguousarray(b + 0.01 * rng.standard_normal(shape).astype(np.float32))
    s, err, ph = phase_cross_correlation(a, b, upsample_factor=up)
    out[name + "_a"], out[name + "_b"], out[name + "_up"] = a, b, np.array(up)
    out[name + "_shift"] = np.asarray(s, dtype=np.float64)
for name, shape, win in [("ssim2d", (40, 50), 7), ("ssim3d", (12, 20, 18), 5), ("ssim3d_w3", (9, 11, 10), 3)]:
    x = ndimage.gaussian_filter(rng.random(shape), 1.0)
    y = x + 0.05 * rng.standard_normal(shape)
    dr = float(max(x.max(), y.max()) - min(x.min(), y.min()))
    out[name + "_x"], out[name + "_y"], out[name + "_win"], out[name + "_dr"] = x, y, np.array(win), np.array(dr)
    out[name + "_val"] = np.array(structural_similarity(x, y, data_range=dr, win_size=win))
np.savez_compressed(__file__.replace("make_skimage018_fixture.py", "skimage018_pcc.npz"), **out)
print("wrote", len(out), "arrays")

# ---- round 2: pins for the pieces the first fixture left open -------------
# Written to a second file so that the first one stays byte-identical.
from skimage.exposure import rescale_intensity                                    # noqa: E402
from skimage.registration._phase_cross_correlation import _upsampled_dft          # noqa: E402

out2 = {}
# Q1: the masked variant exactly as registration.py:433-443 calls it -- images still holding NaN, masks INVERTED
# (True = NaN).  scikit-image zeroes the pixels where the mask is False (the valid ones), keeps the NaNs, every
# correlation term becomes NaN, `denom > tol` is False everywhere, the normalised correlation is all zeros, every
# position is a maximum and the mean of all positions is the zero shift.
for name, shape, nanspec in [("q1_2d", (50, 90), "moving_border"), ("q1_3d", (12, 30, 26), "both"), ("q1_2d_fixed", (40, 44), "fixed_block")]:
    a = ndimage.gaussian_filter(rng.random(shape), 1.2).astype(np.float32)
    b = ndimage.gaussian_filter(rng.random(shape), 1.2).astype(np.float32)
    if nanspec in ("moving_border", "both"):
        b[..., :4] = np.nan
    if nanspec == "both":
        a[:2] = np.nan
    if nanspec == "fixed_block":
        a[5:9, 7:20] = np.nan
    am, bm = np.isnan(a), np.isnan(b)
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        s = phase_cross_correlation(a, b, reference_mask=am, moving_mask=bm, upsample_factor=10 if a.ndim == 2 else 2)
    s = s[0] if isinstance(s, tuple) else s
    out2[name + "_a"], out2[name + "_b"] = a, b
    out2[name + "_shift"] = np.asarray(s, dtype=np.float64)
    # 0.18.3 correlates axes (0, 1) only (`axes=(0, 1)` is hard-coded in _masked_phase_cross_correlation; later releases pass
    # every axis), so for 3D input its last component is the centre of the uncorrelated axis.  The same 0.18.3 correlation
    # routine over ALL axes + the function's own tail gives what the n-D releases return:
    from skimage.registration._masked_phase_cross_correlation import cross_correlate_masked
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        xcorr = cross_correlate_masked(b, a, bm, am, axes=tuple(range(a.ndim)), mode="full", overlap_ratio=0.3)
    maxima = np.stack(np.nonzero(xcorr == xcorr.max()), axis=1)
    out2[name + "_shift_allaxes"] = -(np.mean(maxima, axis=0) - np.array(a.shape) + 1)
    out2[name + "_xcorr_absmax"] = np.array(float(np.abs(xcorr).max()))

# Q5: rescale_intensity as registration.py:381-389 calls it (in_range = nanmin / nanmax, out_range (0, 1)), float32 and
# float32-with-NaN inputs.  0.18.3 returns float64 holding the float32 arithmetic's values; >= 0.19 keeps float32.
for name, arr in [("rs_f32", (ndimage.gaussian_filter(rng.random((37, 41)), 1.0) * 3000 + 17).astype(np.float32)),
                  ("rs_u16like", rng.integers(0, 4096, (20, 30, 10)).astype(np.float32))]:
    if name == "rs_f32":
        arr[3:5, :7] = np.nan
    r = rescale_intensity(arr, in_range=(np.nanmin(arr), np.nanmax(arr)), out_range=(0, 1))
    out2[name + "_in"], out2[name + "_out"] = arr, np.asarray(r)
    out2[name + "_out_dtype"] = np.array(str(np.asarray(r).dtype))

# Q4: SSIM of FLOAT32 images.  0.18.3 converts to float64 first, >= 0.19 (and the oracle) filter in float32: the pinned
# value is the float64 one, the test states the float32 tolerance.
for name, shape, win in [("ssimf32_2d", (48, 60), 7), ("ssimf32_3d", (14, 22, 20), 7), ("ssimf32_3d_w5", (9, 12, 30), 5)]:
    x = ndimage.gaussian_filter(rng.random(shape), 1.0).astype(np.float32)
    y = (x + 0.05 * rng.standard_normal(shape)).astype(np.float32)
    dr = float(max(x.max(), y.max()) - min(x.min(), y.min()))
    out2[name + "_x"], out2[name + "_y"], out2[name + "_win"], out2[name + "_dr"] = x, y, np.array(win), np.array(dr)
    out2[name + "_val"] = np.array(structural_similarity(x, y, data_range=dr, win_size=win))

# "phase" normalisation (added to scikit-image in 0.19: image_product /= max(|image_product|, 100 eps)): 0.18.3 has no
# such keyword, so the normalisation line itself stays a restatement -- but everything around it is executed here with
# 0.18.3's own code: scipy.fft transforms, argmax, wrap-around and its private _upsampled_dft on the normalised product.
import scipy.fft                                                                   # noqa: E402
for name, shape, shift, up in [("ph_2d", (64, 104), (3, -5), 10), ("ph_2d_odd", (53, 97), (-7, 2), 10), ("ph_3d", (27, 40, 36), (2, -3, 4), 2)]:
    pad = 10
    big = ndimage.gaussian_filter(rng.random(tuple(s_ + 2 * pad for s_ in shape)), 1.5).astype(np.float32)
    a = np.ascontiguousarray(big[tuple(slice(pad, pad + s_) for s_ in shape)])
    b = np.ascontiguousarray(big[tuple(slice(pad + d, pad + d + s_) for d, s_ in zip(shift, shape))] + 0.01 * rng.standard_normal(shape).astype(np.float32))
    src_freq, target_freq = scipy.fft.fftn(a), scipy.fft.fftn(b)
    image_product = src_freq * target_freq.conj()
    eps = np.finfo(image_product.real.dtype).eps
    image_product /= np.maximum(np.abs(image_product), 100 * eps)
    cc = scipy.fft.ifftn(image_product)
    maxima = np.unravel_index(np.argmax(np.abs(cc)), cc.shape)
    midpoints = np.array([np.fix(axis_size / 2) for axis_size in shape])
    shifts = np.stack(maxima).astype(np.float64)
    shifts[shifts > midpoints] -= np.array(shape)[shifts > midpoints]
    shifts = np.round(shifts * up) / up
    region = np.ceil(up * 1.5)
    dftshift = np.fix(region / 2.0)
    cc2 = _upsampled_dft(image_product.conj(), region, up, dftshift - shifts * up).conj()
    m2 = np.stack(np.unravel_index(np.argmax(np.abs(cc2)), cc2.shape)).astype(np.float64) - dftshift
    shifts = shifts + m2 / up
    out2[name + "_a"], out2[name + "_b"], out2[name + "_up"] = a, b, np.array(up)
    out2[name + "_peak"] = np.array(maxima)
    out2[name + "_shift"] = shifts
np.savez_compressed(__file__.replace("make_skimage018_fixture.py", "skimage018_round2.npz"), **out2)
print("wrote", len(out2), "arrays (round 2)")
