"""GPU parity: registration kernels (rescale, FFT phase correlation) against oracle/reg_oracle.py."""
import numpy as np
import pytest
from scipy import ndimage

from oracle import reg_oracle as ro

pytestmark = pytest.mark.gpu


def _pair(shape, shift, seed=0, sigma=1.0, noise=0.0):
    rng = np.random.default_rng(seed)
    pad = 12
    big = ndimage.gaussian_filter(rng.random(tuple(s + 2 * pad for s in shape)), sigma).astype(np.float32)
    a = np.ascontiguousarray(big[tuple(slice(pad, pad + s) for s in shape)])
    b = np.ascontiguousarray(big[tuple(slice(pad + d, pad + d + s) for d, s in zip(shift, shape))])
    if noise:
        b = b + noise * rng.standard_normal(shape).astype(np.float32)
    return a, b


# power-of-two, odd, prime and mixed sizes: the transform length is the overlap shape itself
SHAPES = [((64, 128), (3, -5)), ((53, 104), (-4, 6)), ((97, 411), (7, -11)), ((16, 32, 64), (1, -2, 3)),
          ((27, 40, 52), (2, 3, -4)), ((9, 131, 17), (-1, 5, 2)), ((1, 60, 70), (0, 2, -3))]


@pytest.mark.parametrize("shape,shift", SHAPES)
@pytest.mark.parametrize("normalization", ["phase", None])
def test_phasecorr_peak_index_bit_exact_and_shift(hip_device, shape, shift, normalization):
    from multiview_stitcher_amd import _reg_ops

    a, b = _pair(shape, shift, noise=0.002)
    up = 10 if len(shape) == 2 else 2
    want, wdbg = ro.phase_cross_correlation(a, b, upsample_factor=up, normalization=normalization, return_debug=True)
    got, gdbg = _reg_ops.phase_cross_correlation(a, b, upsample_factor=up, normalization=normalization, return_debug=True)
    np.testing.assert_array_equal(gdbg["peak_index"], wdbg["peak_index"])          # integer argmax: bit exact
    np.testing.assert_array_equal(np.asarray(got, dtype=np.float64), np.asarray(want, dtype=np.float64))
    assert abs(gdbg["peak_abs"] - wdbg["peak_abs"]) <= 1e-4 * abs(wdbg["peak_abs"]) + 1e-7


def test_phasecorr_upsample_one(hip_device):
    from multiview_stitcher_amd import _reg_ops

    a, b = _pair((48, 80), (4, -7))
    want = ro.phase_cross_correlation(a, b, upsample_factor=1, normalization=None)
    got = _reg_ops.phase_cross_correlation(a, b, upsample_factor=1, normalization=None)
    np.testing.assert_array_equal(np.asarray(got, np.float64), np.asarray(want, np.float64))


def test_argmax_tie_break_lowest_index(hip_device):
    """Identical constant-free periodic inputs: cc has its peak at index 0; a delta image makes every
    |cc| equal for the phase-normalised case -> np.argmax picks flat index 0."""
    from multiview_stitcher_amd import _reg_ops

    a = np.zeros((8, 16), np.float32)
    a[0, 0] = 1.0
    want, wd = ro.phase_cross_correlation(a, a, upsample_factor=1, normalization=None, return_debug=True)
    got, gd = _reg_ops.phase_cross_correlation(a, a, upsample_factor=1, normalization=None, return_debug=True)
    np.testing.assert_array_equal(gd["peak_index"], wd["peak_index"])
    np.testing.assert_array_equal(gd["peak_index"], [0, 0])


@pytest.mark.parametrize("shape", [(40, 50), (7, 33, 21)])
def test_rescale_intensity(hip_device, shape):
    from multiview_stitcher_amd import _reg_ops

    rng = np.random.default_rng(3)
    im = (rng.random(shape).astype(np.float32) * 937 + 11).astype(np.float32)
    im[..., :3] = np.nan
    want = ro.rescale_intensity_01(im)
    got, mn, mx, nv = _reg_ops.rescale_intensity(im)
    assert mn == np.nanmin(im) and mx == np.nanmax(im) and nv == np.sum(~np.isnan(im))
    np.testing.assert_array_equal(np.isnan(got), np.isnan(want))
    np.testing.assert_allclose(got[~np.isnan(got)], want[~np.isnan(want)], rtol=0, atol=1.2e-7)


def _check_scores(a, b, t_cands, region_mode):
    """mvs_score_candidates against the oracle's candidate loop on rescaled inputs."""
    from multiview_stitcher_amd import _reg_ops

    im0, im1 = ro.rescale_intensity_01(a), ro.rescale_intensity_01(b)
    im0nm = np.isnan(im0)
    data_range = np.nanmax([im0, im1]) - np.nanmin([im0, im1])
    im1_min = np.nanmin(im1)
    valid1 = np.sum(~np.isnan(im1))
    im0_bb = ro.get_bb_from_nanmask(~im0nm)
    ssim, spear, codes = _reg_ops.score_candidates(im0, im1, t_cands, region_mode, data_range, im1_min)
    for i, t in enumerate(t_cands):
        code, s, q = ro.score_candidate(im0, im1, im0nm, t, valid1, region_mode, data_range, im1_min, im0_bb)
        assert codes[i] == code, (i, t, codes[i], code)
        if code == 0:
            assert abs(ssim[i] - s) <= 2e-5 * max(abs(s), 1e-3), (i, t, ssim[i], s)
            assert abs(spear[i] - q) <= 1e-5, (i, t, spear[i], q)
        elif code == 1:
            assert ssim[i] == -1 and spear[i] == -1
    # quality_for_all=False: same SSIM / codes; Spearman only for the best-SSIM candidate(s), NaN for the rest
    ssim2, spear2, codes2 = _reg_ops.score_candidates(im0, im1, t_cands, region_mode, data_range, im1_min, quality_for_all=False)
    np.testing.assert_array_equal(codes2, codes)
    np.testing.assert_array_equal(ssim2, ssim)
    listed = [i for i in range(len(codes)) if codes[i] != 2]
    if listed:
        best = np.nanmax([ssim[i] for i in listed])
        for i in listed:
            if ssim[i] == best or codes[i] == 1:
                assert spear2[i] == spear[i], (i, spear2[i], spear[i])
            else:
                assert np.isnan(spear2[i])


@pytest.mark.parametrize("shape,shift", [((60, 104), (3, -5)), ((24, 40, 36), (2, -3, 4)), ((1, 50, 64), (0, 4, 2))])
def test_score_candidates_no_nan(hip_device, shape, shift):
    a, b = _pair(shape, shift)
    nd = len(shape)
    cands = [list(np.zeros(nd)), list(-np.asarray(shift, float)), list(np.asarray(shift, float) + 0.5),
             [s * 0.9 for s in shape][:nd], [-(shift[d] - shape[d]) for d in range(nd)]]
    _check_scores(a, b, cands, "union")
    _check_scores(a, b, cands + cands[:2], "union")     # exact duplicates are scored once and scattered


def test_score_candidates_with_nan_borders(hip_device):
    a, b = _pair((40, 90), (2, 3))
    a = a.copy(); b = b.copy()
    a[:3] = np.nan
    b[:, -5:] = np.nan
    cands = [[0.0, 0.0], [-2.0, -3.0], [-2.5, -3.5], [35.0, 0.0], [1.0, -80.0]]
    _check_scores(a, b, cands, "intersection")
    _check_scores(a, b, cands, "union")


@pytest.mark.parametrize("shape,shift", [((60, 104), (3, -5)), ((53, 97), (-6, 2)), ((24, 64, 56), (2, -3, 4)),
                                         ((9, 70, 66), (1, 4, -5))])
def test_phase_correlation_registration_matches_oracle(hip_device, shape, shift):
    """End to end: same selected candidate (bit-exact translation) and the same quality."""
    from multiview_stitcher_amd import registration

    a, b = _pair(shape, shift)
    want = ro.phase_correlation_registration(a, b, return_debug=True)
    got = registration.phase_correlation_registration(a, b, return_debug=True)
    np.testing.assert_array_equal(got["debug"]["t_candidates"], want["debug"]["t_candidates"])
    assert got["debug"]["codes"] == want["debug"]["codes"]
    assert got["debug"]["argmax_index"] == want["debug"]["argmax_index"]
    np.testing.assert_array_equal(got["affine_matrix"], want["affine_matrix"])
    assert abs(got["quality"] - want["quality"]) <= 1e-5


def test_phase_correlation_registration_with_nans(hip_device):
    from multiview_stitcher_amd import registration

    a, b = _pair((50, 90), (2, 3))
    b = b.copy()
    b[:, :4] = np.nan
    want = ro.phase_correlation_registration(a, b, return_debug=True)
    got = registration.phase_correlation_registration(a, b, return_debug=True)
    assert got["debug"]["region_mode"] == want["debug"]["region_mode"] == "intersection"
    np.testing.assert_array_equal(got["debug"]["t_candidates"], want["debug"]["t_candidates"])
    np.testing.assert_array_equal(got["affine_matrix"], want["affine_matrix"])
    assert abs(got["quality"] - want["quality"]) <= 1e-5


def test_device_resident_inputs(hip_device):
    from multiview_stitcher_amd import registration
    from multiview_stitcher_amd.device import DeviceArray

    a, b = _pair((20, 48, 40), (1, -2, 3))
    host = registration.phase_correlation_registration(a, b)
    dev = registration.phase_correlation_registration(DeviceArray.from_host(a), DeviceArray.from_host(b))
    np.testing.assert_array_equal(host["affine_matrix"], dev["affine_matrix"])
    assert host["quality"] == dev["quality"]


def test_constant_overlap_gives_identity_with_warning(hip_device):
    """registration.dispatch_pairwise_reg_func (registration.py:1500-1520): a constant crop -> warning + identity, NaN quality."""
    from multiview_stitcher_amd import registration

    a = np.full((20, 30), 3.0, np.float32)
    b = np.random.default_rng(0).random((20, 30)).astype(np.float32)
    with pytest.warns(UserWarning, match="constant"):
        res = registration.dispatch_pairwise_reg_func(registration.phase_correlation_registration, fixed_data=a, moving_data=b)
    np.testing.assert_array_equal(res["affine_matrix"], np.eye(3))
    assert np.isnan(res["quality"])
    # and the multi-normalisation entry point agrees with the single calls
    from multiview_stitcher_amd import _reg_ops
    a2, b2 = _pair((24, 40, 36), (2, -3, 4))
    a2, b2 = np.nan_to_num(ro.rescale_intensity_01(a2)), np.nan_to_num(ro.rescale_intensity_01(b2))
    multi = _reg_ops.phase_cross_correlation_multi(a2, b2, 2, ("phase", None))
    for (s, dbg), norm in zip(multi, ("phase", None)):
        s1, dbg1 = _reg_ops.phase_cross_correlation(a2, b2, 2, norm, return_debug=True)
        np.testing.assert_array_equal(s, s1)
        np.testing.assert_array_equal(dbg["peak_index"], dbg1["peak_index"])
        # both correlations share one inverse transform (|Re| / |Im| channel, power-of-two scaled): same peak, height to ~1e-6
        assert dbg["peak_abs"] == pytest.approx(dbg1["peak_abs"], rel=1e-5)


def test_context_lanes_give_identical_results(hip_device):
    """`device | lane << 8` addresses an independent context on the same GPU: same numbers, and a buffer
    produced through one lane is readable through another."""
    from multiview_stitcher_amd import _reg_ops
    from multiview_stitcher_amd.device import DeviceArray

    a, b = _pair((24, 40, 36), (2, -3, 4))
    a, b = np.nan_to_num(ro.rescale_intensity_01(a)), np.nan_to_num(ro.rescale_intensity_01(b))
    want = _reg_ops.phase_cross_correlation(a, b, 2, "phase", device=0, return_debug=True)
    da, db = DeviceArray.from_host(a, 0), DeviceArray.from_host(b, 1 << 8)      # lane 0 and lane 1 allocations
    for dev in (1 << 8, 3 << 8):
        got = _reg_ops.phase_cross_correlation(da, db, 2, "phase", device=dev, return_debug=True)
        np.testing.assert_array_equal(got[0], want[0])
        assert got[1]["peak_abs"] == want[1]["peak_abs"]


@pytest.mark.parametrize("shape", [(8, 8), (64, 128), (6, 10), (60, 104), (16, 32, 64), (5, 12, 27), (51, 64, 30), (4, 51, 16), (3, 7, 51),
                                   (51, 3, 5), (33, 34, 35), (63, 130), (130, 63), (2, 3, 70), (65, 66, 3), (256, 19, 128), (5, 256, 64), (3, 128, 256), (64, 7)])
@pytest.mark.parametrize("inverse", [False, True])
def test_fft_c2c_matches_numpy(hip_device, shape, inverse):
    """mvs_fft_c2c against numpy.fft on general complex input (powers of two: register transforms for 64 / 128 / 256 samples, Stockham
    radix-4/2 otherwise; other lengths: Bluestein, in registers up to 128 samples -- along x, y and z, ragged last workgroup
    included)."""
    from multiview_stitcher_amd import _reg_ops

    rng = np.random.default_rng(1)
    a = (rng.standard_normal(shape) + 1j * rng.standard_normal(shape)).astype(np.complex64)
    want = np.fft.ifftn(a.astype(np.complex128)) * a.size if inverse else np.fft.fftn(a.astype(np.complex128))
    got = _reg_ops.fftn(a, inverse=inverse)
    scale = np.abs(want).max()
    assert np.abs(got - want).max() <= 3e-6 * scale * np.log2(a.size)


def _line_lengths():
    """The lengths the whole-line register DFT is built for: non-powers of two from 17 to 64 with prime factors <= 19."""
    out = []
    for n in range(17, 65):
        if n & (n - 1) == 0:
            continue
        m = n
        for p in (2, 3, 5, 7, 11, 13, 17, 19):
            while m % p == 0:
                m //= p
        if m == 1:
            out.append(n)
    return out


@pytest.mark.parametrize("inverse", [False, True])
def test_whole_line_dft_every_length_every_axis(hip_device, inverse):
    """csrc/mvs_dft_small.h on the device: every supported length (33 of them: prime-factor splits such as 51 = 3 x 17,
    Cooley-Tukey ones such as 49 = 7 x 7 and 27 = 3 x 9, dense prime leaves up to 19) as the x, the y and the z axis of a small
    volume -- contiguous lines staged through LDS and strided lines loaded directly, ragged last workgroup included -- against
    numpy, and bit for bit against itself when the same lines are transformed through the other access path's neighbours."""
    from multiview_stitcher_amd import _lib, _reg_ops

    rng = np.random.default_rng(3)
    lengths = _line_lengths()
    assert len(lengths) == 33 and 51 in lengths and 49 in lengths and 57 in lengths
    for n in lengths:
        for shape in ((5, 7, n), (3, n, 9), (n, 2, 37)):
            a = (rng.standard_normal(shape) + 1j * rng.standard_normal(shape)).astype(np.complex64)
            want = np.fft.ifftn(a.astype(np.complex128)) * a.size if inverse else np.fft.fftn(a.astype(np.complex128))
            got = _reg_ops.fftn(a, inverse=inverse)
            assert np.abs(got - want).max() <= 3e-6 * np.abs(want).max() * np.log2(a.size), (n, shape)
    # the Bluestein kernels the lengths ran on before (option fft_no_line) agree to float32 rounding
    a = (rng.standard_normal((6, 51, 40)) + 1j * rng.standard_normal((6, 51, 40))).astype(np.complex64)
    got = _reg_ops.fftn(a, inverse=inverse)
    _lib.set_option("fft_no_line", 1)
    try:
        ref = _reg_ops.fftn(a, inverse=inverse)
    finally:
        _lib.set_option("fft_no_line", 0)
    assert np.abs(got - ref).max() <= 2e-6 * np.abs(ref).max() * np.log2(a.size)


@pytest.mark.parametrize("shape", [(3, 8192), (2, 5000), (4, 3000), (4100, 5), (2, 8200, 3), (1, 16384), (1, 65536 + 2)])
@pytest.mark.parametrize("inverse", [False, True])
def test_fft_c2c_long_axes_match_numpy(hip_device, shape, inverse):
    """Axes beyond the LDS core -- powers of two above 4096, other lengths above 2048 (scipy.fft inside registration.py:422-431
    has no limit: an unbinned pair of an 8k camera frame) -- run as a four-step transform in device memory (Bluestein on top of
    it for lengths that are no power of two); contiguous and strided axes, both directions."""
    from multiview_stitcher_amd import _reg_ops

    rng = np.random.default_rng(2)
    a = (rng.standard_normal(shape) + 1j * rng.standard_normal(shape)).astype(np.complex64)
    want = np.fft.ifftn(a.astype(np.complex128)) * a.size if inverse else np.fft.fftn(a.astype(np.complex128))
    got = _reg_ops.fftn(a, inverse=inverse)
    scale = np.abs(want).max()
    assert np.abs(got - want).max() <= 3e-6 * scale * np.log2(a.size)


def test_phase_correlation_of_a_long_2d_pair(hip_device):
    """A 2D overlap crop with an 8192-long axis (powers of two above 4096) and a 2500-long one (Bluestein above 2048):
    integer peak index equal to the oracle's."""
    from multiview_stitcher_amd import _reg_ops

    rng = np.random.default_rng(3)
    from scipy import ndimage
    big = ndimage.gaussian_filter(rng.random((2600, 8300)).astype(np.float32), 2.0)
    a = np.ascontiguousarray(big[40:2540, 50:8242])
    b = np.ascontiguousarray(big[47:2547, 39:8231])
    a, b = ro.rescale_intensity_01(a).astype(np.float32), ro.rescale_intensity_01(b).astype(np.float32)
    want = ro.phase_cross_correlation(a, b, upsample_factor=1, normalization="phase")
    got = _reg_ops.phase_cross_correlation(a, b, 1, "phase", device=0)
    np.testing.assert_array_equal(np.asarray(got), np.asarray(want, dtype=np.float32))
    np.testing.assert_array_equal(np.abs(np.asarray(got)), [7.0, 11.0])


@pytest.mark.parametrize("case", ["2d", "3d", "3d_nan", "2d_nan_union", "upsample1"])
def test_register_crops_equals_the_stepwise_flow(hip_device, case):
    """mvs_register_crops (the whole phase_correlation_registration in one call) against the step-by-step Python flow
    that the oracle tests pin: same translation bit for bit, same quality."""
    from multiview_stitcher_amd import registration

    kw = {}
    if case == "2d":
        a, b = _pair((60, 104), (3, -5))
    elif case == "3d":
        a, b = _pair((24, 64, 56), (2, -3, 4))
    elif case == "3d_nan":
        a, b = _pair((20, 48, 52), (-2, 3, 1))
        a = a.copy(); b = b.copy()
        a[:2] = np.nan
        b[:, :, -4:] = np.nan
    elif case == "2d_nan_union":
        a, b = _pair((40, 90), (2, 3))
        a = a.copy()
        a[:3] = np.nan
        kw["disambiguate_region_mode"] = "union"
    else:
        a, b = _pair((53, 97), (-6, 2))
        kw["upsample_factor"] = 1
    want = registration.phase_correlation_registration(a, b, return_debug=True, **kw)     # Python flow
    got = registration.phase_correlation_registration(a, b, **kw)                          # one library call
    np.testing.assert_array_equal(got["affine_matrix"], want["affine_matrix"])
    assert got["quality"] == want["quality"] or (np.isnan(got["quality"]) and np.isnan(want["quality"]))


def test_on_the_fly_shifted_ssim_equals_materialised_copies(hip_device):
    """Finite-only crops: the candidates' shifted values are evaluated inside the SSIM z pass (no shifted copies, no mask
    reduction, analytic valid boxes).  Must give the same bits as the path that writes the copies (option
    "materialize_shifts"), for integer, half-integer and tenth-pixel shifts, in 2D and 3D, incl. rejected candidates."""
    from multiview_stitcher_amd import _lib, _reg_ops

    rng = np.random.default_rng(5)
    for shape, cands in [((20, 48, 40), [(0, 0, 0), (1.5, -2, 0.5), (-3, 4, 2), (19, 0, 0), (0.5, 0.5, 0.5), (-18.5, 47, 39), (25, 0, 0)]),
                         ((64, 72), [(0, 0), (2.3, -1.7), (-4, 5), (0.1, 63.9), (70, 1)])]:
        a = rng.random(shape).astype(np.float32)
        b = rng.random(shape).astype(np.float32)
        dr = float(max(a.max(), b.max()) - min(a.min(), b.min()))
        res = []
        for flag in (1, 0):
            _lib.set_option("materialize_shifts", flag)
            try:
                res.append(_reg_ops.score_candidates(a, b, np.array(cands, dtype=np.float64), "union", dr, float(b.min()), quality_for_all=False))
            finally:
                _lib.set_option("materialize_shifts", 0)
        for x, y in zip(res[0], res[1]):
            np.testing.assert_array_equal(x, y)
        assert (res[0][2] == 0).sum() >= 3 and (res[0][2] == 1).sum() >= 1


def test_packed_inverse_transform_finds_the_same_peaks(hip_device):
    """mvs_phasecorr_multi with ("phase", None): one inverse transform carries both correlations.  Against one transform per
    normalisation (option "materialize_shifts" = the plain paths): identical integer peaks and sub-pixel shifts on smooth
    images (broad plain-correlation peaks: the hard case for cross-channel rounding noise), 2D and 3D, odd sizes."""
    from multiview_stitcher_amd import _lib, _reg_ops

    for shape, shift in [((51, 128, 96), (2, -3, 5)), ((40, 45, 53), (-1, 4, 0)), ((1, 200, 104), (0, 7, -6)), ((64, 64, 27), (3, 0, -2))]:
        a, b = _pair(shape, shift)
        a, b = np.nan_to_num(ro.rescale_intensity_01(a)), np.nan_to_num(ro.rescale_intensity_01(b))
        a2, b2 = (a[0], b[0]) if shape[0] == 1 else (a, b)
        res = []
        for flag in (1, 0):
            _lib.set_option("materialize_shifts", flag)
            try:
                res.append(_reg_ops.phase_cross_correlation_multi(a2, b2, 2 if a2.ndim == 3 else 10, ("phase", None)))
            finally:
                _lib.set_option("materialize_shifts", 0)
        for (s0, d0), (s1, d1) in zip(*res):
            np.testing.assert_array_equal(s0, s1)
            np.testing.assert_array_equal(d0["peak_index"], d1["peak_index"])
            assert d0["peak_abs"] == pytest.approx(d1["peak_abs"], rel=1e-5)


def test_register_crops_shortcuts_for_finite_crops_change_nothing(hip_device):
    """Finite crops take every shortcut of mvs_register_crops (packed inverse transform, no image statistics pass, analytic
    valid boxes, on-the-fly integer shifts); "materialize_shifts" switches all of them off.  Same translation, same quality."""
    from multiview_stitcher_amd import _lib, _reg_ops

    cases = [((24, 40, 36), (2, -3, 4), 2, False), ((51, 64, 48), (-1, 2, 3), 2, False), ((96, 80), (5, -7), 10, False),
             # integer-valued crops (uint16 tiles on the fixed grid): 16-bit rank keys for the fixed image; the last one with NaNs
             ((24, 40, 36), (2, -3, 4), 2, True), ((40, 72, 30), (0, 1, -2), 2, True), ((20, 48, 40), (1, 2, 0), 2, "nan")]
    for shape, shift, up, integer in cases:
        a, b = _pair(shape, shift, noise=0.002)
        if integer:
            a, b = np.round(a * 3000).astype(np.float32), np.round(b * 3000).astype(np.float32)
            if integer == "nan":
                a[:, :3] = np.nan
                b[..., -2:] = np.nan
        res = []
        for flag in (1, 0):
            _lib.set_option("materialize_shifts", flag)
            try:
                res.append(_reg_ops.register_crops(a, b, up))
            finally:
                _lib.set_option("materialize_shifts", 0)
        (t0, q0, st0, nc0), (t1, q1, st1, nc1) = res
        assert st0 == st1 == 0 and nc0 == nc1
        np.testing.assert_array_equal(t0, t1)
        if integer is True:
            # finite integer-valued crops + a shift in multiples of 1/2: the default path ranks by key histograms, the plain
            # one by radix sorts -- the same rank vectors, sums taken in a different order
            assert q0 == pytest.approx(q1, rel=1e-10)
            want = ro.phase_correlation_registration(a, b)
            np.testing.assert_array_equal(want["affine_matrix"][:-1, -1], t1)
            assert abs(q1 - want["quality"]) < 1e-6
        else:
            assert q0 == q1


@pytest.mark.parametrize("shape,bins,jit", [((48, 80, 64), (2, 2, 2), (1, -3, 2)), ((40, 64, 96), (2, 2, 2), (3, 1, -1)), ((64, 96), (1, 1), (2, -5))])
def test_histogram_ranks_on_binned_integer_tiles_match_oracle(hip_device, shape, bins, jit):
    """uint16 tiles binned by 2 (registration.py:1732-1741: block mean cast back to uint16) and shifted by odd pixel counts:
    the true shift is a half-integer on the binned grid, the winning candidate's moving crop is interpolated with weights
    1/2, and the Spearman coefficient comes from key histograms.  Translation equal to the oracle's, quality within 1e-6."""
    from multiview_stitcher_amd import _reg_ops

    rng = np.random.default_rng(7)
    ndim = len(shape)
    pad = 8
    big = ndimage.gaussian_filter(rng.random(tuple(s * b + 2 * pad for s, b in zip(shape, bins))), 2.0)
    big = np.round((big - big.min()) / (big.max() - big.min()) * 4000).astype(np.uint16)

    def binned(off):
        sl = tuple(slice(pad + o, pad + o + s * b) for o, s, b in zip(off, shape, bins))
        v = big[sl].astype(np.float64)
        v = v.reshape([q for s, b in zip(shape, bins) for q in (s, b)]).mean(axis=tuple(range(1, 2 * ndim, 2)))
        return v.astype(np.uint16).astype(np.float32)

    a, b = binned((0,) * ndim), binned(jit)
    t, q, st, nc = _reg_ops.register_crops(a, b, 2 if ndim == 3 else 10)
    want = ro.phase_correlation_registration(a, b)
    assert st == 0
    np.testing.assert_array_equal(t, want["affine_matrix"][:-1, -1])
    if ndim == 3:
        assert np.any(np.abs(t * 2 % 2) == 1)          # a genuinely half-integer component
    assert abs(q - want["quality"]) < 1e-6


@pytest.mark.parametrize("shape,bins", [((16, 64, 128), (2, 2, 2)), ((9, 33, 64), (1, 2, 2)), ((12, 20, 72), (3, 1, 2)), ((8, 30, 50), (2, 2, 2)),
                                        ((6, 24, 40), (2, 2, 1)), ((64, 96), (2, 2)), ((10, 16, 24), (2, 3, 4))])
def test_bin_mean_matches_numpy(hip_device, shape, bins):
    """coarsen(bins, boundary="trim").mean().astype(dtype) (registration.py:1732-1741): the vectorised uint16 kernel (bin 2 along x,
    aligned rows) and the generic one against numpy, host and device (strided window) inputs."""
    from multiview_stitcher_amd import _reg_ops
    from multiview_stitcher_amd.device import DeviceArray

    rng = np.random.default_rng(4)
    for dtype in (np.uint16, np.uint8, np.float32):
        a = (rng.random(shape) * (60000 if dtype == np.uint16 else 250)).astype(dtype)
        sl = tuple(slice(0, (n // b) * b) for n, b in zip(shape, bins))
        shp = []
        for n, b in zip(shape, bins):
            shp += [n // b, b]
        want = a[sl].reshape(shp).mean(axis=tuple(range(1, 2 * len(shape), 2))).astype(dtype)
        got = _reg_ops.bin_mean(a, list(bins))
        if dtype == np.float32:
            np.testing.assert_allclose(got, want, rtol=1e-6)
        else:
            np.testing.assert_array_equal(got, want)
        d = DeviceArray.from_host(a, 0)
        np.testing.assert_array_equal(_reg_ops.bin_mean(d, list(bins)).get(), got)


def test_many_candidates_span_several_batches(hip_device):
    """More candidates than one batch holds (16): the fixed image's shared SSIM terms are computed once, the batched z / y-x
    launches run per batch, the winner of a later batch is re-shifted for the rank correlation.  Same bits as the plain path."""
    from multiview_stitcher_amd import _lib, _reg_ops

    rng = np.random.default_rng(9)
    shape = (18, 40, 44)
    a = rng.random(shape).astype(np.float32)
    b = np.roll(a, (1, -2, 3), axis=(0, 1, 2)) + 0.01 * rng.random(shape).astype(np.float32)
    cands = [(dz, dy, dx) for dz in (-1, 0, 1.5) for dy in (-2, 0, 2) for dx in (-3, 0.5, 3)] + [(-1, 2, -3), (1, -2, 3)]
    assert len(cands) > 16
    dr = float(max(a.max(), b.max()) - min(a.min(), b.min()))
    res = []
    for flag in (1, 0):
        _lib.set_option("materialize_shifts", flag)
        try:
            res.append(_reg_ops.score_candidates(a, b, np.array(cands, dtype=np.float64), "union", dr, float(b.min()), quality_for_all=False))
        finally:
            _lib.set_option("materialize_shifts", 0)
    for x, y in zip(res[0], res[1]):
        np.testing.assert_array_equal(x, y)
    ssim, spear, codes = res[1]
    best = int(np.nanargmax(np.where(codes == 0, ssim, -np.inf)))
    assert tuple(cands[best]) == (1, -2, 3) and np.isfinite(spear[best]) and np.isnan(spear).sum() == (codes == 0).sum() - 1


@pytest.mark.parametrize("shape", [(51, 256, 256), (40, 56, 64), (30, 50, 70), (64, 64, 128), (20, 40, 300), (70, 300, 33)])
def test_fused_transform_ends_equal_separate_launches(hip_device, shape):
    """The fusions at the ends of the two transforms of the phase correlation (first pass of the forward transform reads the real
    crops, first pass of the inverse forms the cross power, its last pass reduces to the peaks, one first refinement stage for both
    normalisations) against the separate launches they replace (option "reg_unfused": pack, cross-power kernel, stored
    correlation + peak search, one refinement stage per normalisation): integer peaks, peak heights and refined shifts bit for bit.
    The shapes mix register-length axes (64 / 128 / 256, Bluestein up to 128 samples), where the fusions are taken, with axes
    that run on the LDS kernel, where they are not."""
    from multiview_stitcher_amd import _lib, _reg_ops

    a, b = _pair(shape, (2, -3, 4), noise=0.01)
    a, b = np.nan_to_num(ro.rescale_intensity_01(a)), np.nan_to_num(ro.rescale_intensity_01(b))
    res = []
    _lib.set_option("fft_no_slab", 1)        # (crops of the slab kind take three passes of their own: compared in the test below)
    try:
        for flag in (1, 0):
            _lib.set_option("reg_unfused", flag)
            try:
                res.append(_reg_ops.phase_cross_correlation_multi(a, b, upsample_factor=2, normalizations=("phase", None)))
            finally:
                _lib.set_option("reg_unfused", 0)
    finally:
        _lib.set_option("fft_no_slab", 0)
    for (s0, d0), (s1, d1) in zip(*res):
        np.testing.assert_array_equal(s0, s1)
        np.testing.assert_array_equal(d0["peak_index"], d1["peak_index"])
        assert d0["peak_abs"] == d1["peak_abs"]
    np.testing.assert_allclose(res[1][0][0], [2, -3, 4], atol=0.51)


def test_crop_statistics_of_the_crop_kernel_equal_the_separate_reduction(hip_device):
    """register() on device-resident uint16 tiles: the integer crop kernel reduces min / max / #valid of what it writes (no pass of
    its own over the crops for the normalisation) -- against option "reg_unfused", where the reduction kernel runs: parameters and
    qualities bit for bit."""
    from multiview_stitcher_amd import _lib, device, registration, sample_data, spatial_image_utils as si

    sims, jit, _ = sample_data.generate_tiled_dataset(ndim=3, tile_shape=(48, 96, 96), tiles=(1, 2, 2), overlap=(0, 32, 32),
                                                      dtype=np.uint16, max_jitter=2, seed=5)
    sims = [device.to_device(s.isel({"c": 0, "t": 0}), 0) for s in sims]
    res = []
    for flag in (1, 0):
        for lane in range(16):
            _lib.set_option("reg_unfused", flag, device=lane << 8)
        try:
            res.append(registration.register(sims, transform_key=si.DEFAULT_TRANSFORM_KEY, new_transform_key=None,
                                             registration_binning={"z": 1, "y": 2, "x": 2}, return_dict=True))
        finally:
            for lane in range(16):
                _lib.set_option("reg_unfused", 0, device=lane << 8)
    for p0, p1 in zip(res[0]["params"], res[1]["params"]):
        np.testing.assert_array_equal(p0, p1)
    q0, q1 = (r["pairwise_registration"]["metrics"]["qualities"] for r in res)
    assert q0 == q1 and len(q0) >= 3


@pytest.mark.parametrize("shape", [(51, 128, 120), (96, 51, 70), (40, 90, 29), (9, 30, 64)])
def test_fused_ssim_walk_equals_separate_passes_and_oracle(hip_device, shape):
    """Finite 3D crops, several candidates: one launch walks z per (y, x) tile and keeps the z- and y-filtered planes in LDS
    (ssim_fused_batch_kernel) instead of the z launch + y/x launch of option "ssim_two_pass".  Same codes and region
    decisions, SSIM equal up to the summation order of the float64 running sums (1e-9), and within the oracle bar."""
    from multiview_stitcher_amd import _lib, _reg_ops

    rng = np.random.default_rng(11)
    a, b = _pair(shape, (2, -3, 4), noise=0.01)
    a, b = np.nan_to_num(ro.rescale_intensity_01(a)), np.nan_to_num(ro.rescale_intensity_01(b))
    cands = np.array([(0, 0, 0), (-2, 3, -4), (-2.5, 3, -4), (-1.5, 2.5, -3.5), (3, -5, 7), (-2, 3, -3), (0, 0, 1), (shape[0] - 2, 0, 0),
                      (-1, -1, -1), (2 * shape[0], 0, 0)], dtype=np.float64)
    dr = float(max(a.max(), b.max()) - min(a.min(), b.min()))
    res = []
    for flag in (1, 0):
        _lib.set_option("ssim_two_pass", flag)
        try:
            res.append(_reg_ops.score_candidates(a, b, cands, "union", dr, float(b.min()), quality_for_all=False))
        finally:
            _lib.set_option("ssim_two_pass", 0)
    (s0, q0, c0), (s1, q1, c1) = res
    np.testing.assert_array_equal(c0, c1)
    np.testing.assert_allclose(s1, s0, rtol=1e-9, atol=1e-12)
    np.testing.assert_array_equal(np.isnan(q0), np.isnan(q1))
    np.testing.assert_allclose(q1[~np.isnan(q1)], q0[~np.isnan(q0)], rtol=1e-12)
    assert (c1 == 0).sum() >= 6 and (c1 == 1).sum() >= 1
    im0nm = np.isnan(a)
    valid1 = int(np.sum(~np.isnan(b)))
    bb0 = ro.get_bb_from_nanmask(~im0nm)
    for i in (0, 1, 3, 7):
        code, s, _ = ro.score_candidate(a, b, im0nm, cands[i], valid1, "union", dr, float(b.min()), bb0)
        assert code == c1[i]
        if code == 0:
            assert abs(s1[i] - s) <= 2e-5 * max(abs(s), 1e-3)


@pytest.mark.parametrize("shape,shift,noise", [((40, 96, 120), (2, -3, 4), 0.0), ((40, 96, 120), (2, -3, 4), 0.02),
                                               ((40, 96, 120), (2, -3, 4), 0.08), ((40, 96, 120), (1, 0, -5), 0.3),
                                               ((96, 40, 130), (-3, 2, 6), 0.01), ((120, 100, 36), (5, -4, 1), 0.05)])
def test_pruned_argmax_search_equals_full_scoring(hip_device, shape, shift, noise):
    """mvs_register_crops needs the arg-max candidate only: candidates whose SSIM bound (partial sum + (1 + slack) per voxel
    not yet visited) falls below a completely scored one are not finished (option "ssim_prune", default on).  The selected
    translation, the quality and the status are those of the full scoring -- bit for bit, at every noise level (clean pairs
    prune after 1/16 of the volume, noisy ones later or never) -- and those of the oracle."""
    from multiview_stitcher_amd import _lib, _reg_ops

    a, b = _pair(shape, shift, seed=3, sigma=1.5, noise=noise)
    res = {}
    stats = {}
    for flag in (1, 0):
        _lib.set_option("ssim_prune", flag)
        try:
            for key in ("reg_pruned", "reg_cand_volumes", "reg_candidates"):
                _lib.get_counter(key, reset=True)
            res[flag] = _reg_ops.register_crops(a, b, 2)
            stats[flag] = {key: _lib.get_counter(key, reset=True) for key in ("reg_pruned", "reg_cand_volumes", "reg_candidates")}
        finally:
            _lib.set_option("ssim_prune", 1)
    np.testing.assert_array_equal(res[1][0], res[0][0])
    assert res[1][1] == res[0][1] and res[1][2:] == res[0][2:]
    want = ro.phase_correlation_registration(a, b)
    np.testing.assert_array_equal(res[1][0], want["affine_matrix"][:-1, -1])
    assert abs(res[1][1] - want["quality"]) <= 1e-5
    assert stats[0]["reg_pruned"] == 0 and stats[0]["reg_cand_volumes"] == stats[0]["reg_candidates"]
    assert stats[1]["reg_cand_volumes"] <= stats[1]["reg_candidates"] + 1e-9
    if noise == 0.0:      # the wrong-sign candidates of a clean pair leave after the first round
        assert stats[1]["reg_pruned"] >= stats[1]["reg_candidates"] - 2, stats
        assert stats[1]["reg_cand_volumes"] < 0.4 * stats[1]["reg_candidates"], stats


def _sparse_pair(shape, shift, seed, invert, blob_frac=0.25, noise=0.0):
    """Zero background with one textured blob; the moving crop holds the same blob displaced by ``shift`` (optionally with
    inverted contrast, which makes the aligned candidate score BELOW an all-background one)."""
    rng = np.random.default_rng(seed)
    a = np.zeros(shape, np.float32)
    b = np.zeros(shape, np.float32)
    ext = [max(8, int(n * blob_frac)) for n in shape]
    lo = [int(n * 0.1) + 6 for n in shape]
    blob = ndimage.gaussian_filter(rng.random(tuple(ext)), 1.0).astype(np.float32) + 0.5
    a[tuple(slice(l, l + e) for l, e in zip(lo, ext))] = blob
    blob_b = (blob.max() + 0.5 - blob) if invert else blob
    if noise:
        blob_b = blob_b + noise * rng.standard_normal(blob.shape).astype(np.float32)
    b[tuple(slice(l - d, l - d + e) for l, d, e in zip(lo, shift, ext))] = np.maximum(blob_b, 0.05)
    return a, b


@pytest.mark.parametrize("shape,shift,invert,noise", [((40, 96, 120), (2, -3, 4), True, 0.0), ((40, 96, 120), (3, 4, 5), True, 0.0),
                                                      ((40, 96, 120), (-1, -6, -5), True, 0.1), ((96, 40, 130), (3, 2, -5), True, 0.0),
                                                      ((64, 64, 64), (-5, -5, -5), True, 0.1), ((40, 96, 120), (1, 0, -5), False, 0.4)])
def test_pruned_search_ignores_background_only_candidates(hip_device, shape, shift, invert, noise):
    """Sparse crops on a zero background: the wrapped-around candidates move the blob out of the frame, the region that stays holds
    nothing above im1_min -- the reference's `continue` case (registration.py:530-533) -- and against the equally sparse fixed crop
    such a candidate has a HIGH SSIM sum.  The pruned search must not drop the valid candidates against it: translation, quality and
    status equal the full scoring's and the oracle's."""
    from multiview_stitcher_amd import _lib, _reg_ops

    a, b = _sparse_pair(shape, shift, seed=5, invert=invert, noise=noise)
    res = {}
    for flag in (1, 0):
        _lib.set_option("ssim_prune", flag)
        try:
            res[flag] = _reg_ops.register_crops(a, b, 2)
        finally:
            _lib.set_option("ssim_prune", 1)
    np.testing.assert_array_equal(res[1][0], res[0][0])
    assert (res[1][1] == res[0][1] or (np.isnan(res[1][1]) and np.isnan(res[0][1]))) and res[1][2:] == res[0][2:]
    want = ro.phase_correlation_registration(a, b, return_debug=True)
    if invert:
        assert 2 in want["debug"]["codes"], "the case must contain a background-only candidate"
    np.testing.assert_array_equal(res[1][0], want["affine_matrix"][:-1, -1])
    assert abs(res[1][1] - want["quality"]) <= 1e-5


@pytest.mark.parametrize("shape", [(24, 40, 256), (25, 41, 256), (1, 64, 128), (2, 3, 64), (51, 64, 256)])
def test_partner_line_pairs_of_the_inverse_x_pass_equal_flat_order(hip_device, shape):
    """The first pass of the inverse transform (along x, cross power formed on the fly) takes its lines as partner pairs (kz, ky),
    (-kz, -ky) so that the packed spectrum is fetched once (fft_reg2_kernel, option "fft_no_pair" = flat order): even and odd
    line-grid sizes, the self-partner rows and lines, a single plane.  Same arithmetic per line: peaks, heights, shifts bit for bit."""
    from multiview_stitcher_amd import _lib, _reg_ops

    a, b = _pair(shape, (0 if shape[0] < 8 else 2, -1 if shape[1] < 8 else -3, 4), noise=0.01, seed=2)
    a, b = np.nan_to_num(ro.rescale_intensity_01(a)), np.nan_to_num(ro.rescale_intensity_01(b))
    res = []
    for flag in (0, 1):
        _lib.set_option("fft_no_pair", flag)
        try:
            res.append(_reg_ops.phase_cross_correlation_multi(a, b, upsample_factor=2, normalizations=("phase", None)))
        finally:
            _lib.set_option("fft_no_pair", 0)
    for (s0, d0), (s1, d1) in zip(*res):
        np.testing.assert_array_equal(s0, s1)
        np.testing.assert_array_equal(d0["peak_index"], d1["peak_index"])
        assert d0["peak_abs"] == d1["peak_abs"]

@pytest.mark.parametrize("shape", [(51, 256, 256), (256, 51, 256), (256, 256, 51), (20, 64, 128), (128, 33, 64), (64, 128, 45),
                                   (63, 256, 128), (256, 64, 63), (17, 256, 64), (128, 256, 26)])
def test_three_pass_phase_correlation_equals_the_single_axis_passes(hip_device, shape):
    """Crops with one short axis (a whole-line DFT length) and two power-of-two axes run the phase correlation in three passes:
    short + one long axis inside LDS (slab_kernel), the remaining axis forward / cross power / inverse in one kernel over partner line
    pairs (long_xp_kernel), the inverse slabs reduced to their peaks (mvs_fft_slab.hip; option "fft_no_slab" = the six single-axis
    passes).  All three orientations of the short axis, slab lengths 64 / 128 / 256, even and odd short lengths: the same peaks and
    shifts; the peak heights agree to float32 rounding (the transforms add in a different order)."""
    from multiview_stitcher_amd import _lib, _reg_ops

    sh = tuple(int(np.clip(v, -(n // 4), n // 4)) for v, n in zip((3, -5, 4), shape))
    a, b = _pair(shape, sh, noise=0.01, seed=5)
    a, b = np.nan_to_num(ro.rescale_intensity_01(a)), np.nan_to_num(ro.rescale_intensity_01(b))
    res, taken = [], []
    _lib.set_option("fft_slab_axes", 7)          # (default: crops whose short axis is the contiguous one)
    try:
        for flag in (0, 1):
            _lib.set_option("fft_no_slab", flag)
            _lib.get_counter("reg_slab_pairs", reset=True)
            try:
                res.append(_reg_ops.phase_cross_correlation_multi(a, b, upsample_factor=2, normalizations=("phase", None)))
            finally:
                _lib.set_option("fft_no_slab", 0)
            taken.append(_lib.get_counter("reg_slab_pairs", reset=True))
    finally:
        _lib.set_option("fft_slab_axes", 4)
    assert taken == [1.0, 0.0]
    for (s0, d0), (s1, d1) in zip(*res):
        np.testing.assert_array_equal(s0, s1)
        np.testing.assert_array_equal(d0["peak_index"], d1["peak_index"])
        assert abs(d0["peak_abs"] - d1["peak_abs"]) <= 2e-5 * abs(d1["peak_abs"]) + 1e-7
    # and the expected translation is found at all (guards against both paths failing alike)
    got, want = np.asarray(res[0][0][0], dtype=float), np.array(sh, dtype=float)
    assert min(np.abs(got - want).max(), np.abs(got + want).max()) <= 0.5

