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
