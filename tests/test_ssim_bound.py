"""The bound the pruned arg-max search of mvs_score_candidates rests on: the per-voxel structural similarity
skimage computes in float32 never exceeds 1 + slack, slack = 0.01 max(1, (M / R)^2) (M: largest absolute value, R: data
range).  Restated here with the oracle's arithmetic (oracle/reg_oracle.structural_similarity, per-voxel map instead of the
mean) on inputs chosen to stress the float32 variance cancellation."""
import numpy as np
import pytest
from scipy import ndimage


def _ssim_map(im1, im2, data_range, win=7):
    K1, K2 = 0.01, 0.03
    im1 = im1.astype(np.float32)
    im2 = im2.astype(np.float32)
    NP = win ** im1.ndim
    cov_norm = NP / (NP - 1)
    f = lambda a: ndimage.uniform_filter(a, size=win)  # noqa: E731
    ux, uy = f(im1), f(im2)
    uxx, uyy, uxy = f(im1 * im1), f(im2 * im2), f(im1 * im2)
    vx = cov_norm * (uxx - ux * ux)
    vy = cov_norm * (uyy - uy * uy)
    vxy = cov_norm * (uxy - ux * uy)
    R = data_range
    C1, C2 = (K1 * R) ** 2, (K2 * R) ** 2
    S = ((2 * ux * uy + C1) * (2 * vxy + C2)) / ((ux**2 + uy**2 + C1) * (vx + vy + C2))
    pad = (win - 1) // 2
    return S[tuple(slice(pad, s - pad) for s in S.shape)]


def _cases():
    rng = np.random.default_rng(0)
    shape = (24, 40, 44)
    base = ndimage.gaussian_filter(rng.random(shape), 2.0).astype(np.float32)
    base = (base - base.min()) / (base.max() - base.min())
    noise = rng.random(shape).astype(np.float32)
    near1 = np.float32(0.999) + 1e-4 * rng.random(shape).astype(np.float32)
    near1b = np.float32(0.999) + 1e-4 * rng.random(shape).astype(np.float32)
    slab = base.copy()
    slab[8:16] = 1.0
    binary = (rng.random(shape) > 0.5).astype(np.float32)
    yield "identical smooth", base, base, 1.0
    yield "identical noise", noise, noise, 1.0
    yield "nearly constant near the top of the range", near1, near1b, 1.0
    yield "constant", np.ones(shape, np.float32), np.ones(shape, np.float32), 1.0
    yield "binary", binary, binary, 1.0
    yield "saturated slab", slab, slab, 1.0
    yield "smooth against its shift", base, np.roll(base, 1, 2), 1.0
    yield "smooth against noise", base, noise, 1.0
    yield "values twice the range", base + 1.0, base + 1.0, 1.0          # M / R = 2: slack 0.04


@pytest.mark.parametrize("name,a,b,R", list(_cases()), ids=[c[0] for c in _cases()])
def test_per_voxel_ssim_stays_below_one_plus_slack(name, a, b, R):
    S = _ssim_map(a, b, R)
    M = max(float(np.abs(a).max()), float(np.abs(b).max()))
    slack = 1e-2 * max(1.0, (M / R) ** 2)
    assert np.isfinite(S).all()
    # the search allows 1 + slack per voxel it has not visited; what the float32 arithmetic actually reaches stays a factor
    # of ten or more below that allowance
    assert float(S.max()) <= 1.0 + 0.1 * slack, (name, float(S.max()))
    assert float(S.min()) >= -1.0 - 0.1 * slack
