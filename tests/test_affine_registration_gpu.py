"""Row a2 for NON-axis-aligned views (BASELINE config C4: BigStitcher-style views with full affine input transforms):
overlap boxes from the halfspace intersection (registration.py:194-277, mv_graph.py:301-338), both views resampled onto
the fixed view's grid (registration.py:280-350), phase-correlation registration, and register() + fuse() end to end."""
import numpy as np
import pytest
from scipy import ndimage

from oracle import reg_oracle as ro

pytestmark = pytest.mark.gpu


def _rot(deg_z, deg_y=0.0, scale=(1.0, 1.0, 1.0)):
    a, b = np.deg2rad(deg_z), np.deg2rad(deg_y)
    rz = np.array([[1, 0, 0], [0, np.cos(a), -np.sin(a)], [0, np.sin(a), np.cos(a)]])      # rotation in the y-x plane (about z)
    ry = np.array([[np.cos(b), 0, -np.sin(b)], [0, 1, 0], [np.sin(b), 0, np.cos(b)]])      # tilt: z-x plane
    return rz @ ry @ np.diag(scale)


def _c4_pair(hidden=(0.0, 0.0, 0.0), seed=3):
    """Two views of one smooth volume: view 1 on the world grid, view 2 with anisotropic spacing (z = 2), rotated 7 deg
    about z, tilted 2 deg, scaled by 1 / 0.99 / 1.01 and shifted; its metadata transform is off by ``hidden`` (world units)."""
    from multiview_stitcher_amd import spatial_image_utils as si

    rng = np.random.default_rng(seed)
    G = ndimage.gaussian_filter(rng.random((72, 120, 120)), 2.0).astype(np.float32)
    G = (G - G.min()) / (G.max() - G.min())
    o1, s1, n1 = np.array([6.0, 10.0, 8.0]), np.ones(3), (56, 96, 72)
    v1 = G[tuple(slice(int(o), int(o) + n) for o, n in zip(o1, n1))].copy()
    A1 = np.eye(4)
    lin = _rot(7.0, 2.0, (1.0, 0.99, 1.01))
    c = np.array([36.0, 60.0, 80.0])                       # rotation centre (world)
    A2 = np.eye(4)
    A2[:3, :3] = lin
    A2[:3, 3] = c - lin @ c + np.array([1.5, -2.25, 3.0])
    o2, s2, n2 = np.array([8.0, 14.0, 40.0]), np.array([2.0, 1.0, 1.0]), (26, 90, 70)
    # view 2 voxel p holds G(A2 @ (o2 + p * s2)): sample G there
    M = lin @ np.diag(s2)
    off = lin @ o2 + A2[:3, 3]
    v2 = ndimage.affine_transform(G, M, offset=off, output_shape=n2, order=1, mode="nearest").astype(np.float32)
    A2_meta = A2.copy()
    A2_meta[:3, 3] += np.asarray(hidden)
    sims = []
    for data, o, sp, A in ((v1, o1, s1, A1), (v2, o2, s2, A2_meta)):
        s = si.to_spatial_image(data, dims=["z", "y", "x"], scale=dict(zip("zyx", sp)), translation=dict(zip("zyx", o)))
        si.set_sim_affine(s, A, "stage")
        sims.append(s)
    return sims, (A1, A2, A2_meta), G


def test_overlap_boxes_and_intrinsic_crops_match_oracle(hip_device):
    from multiview_stitcher_amd import registration
    from multiview_stitcher_amd import spatial_image_utils as si

    sims, (A1, A2, A2m), _ = _c4_pair()
    views = [{"data": np.asarray(s.data), "origin": si.get_origin_from_sim(s, asarray=True), "spacing": si.get_spacing_from_sim(s, asarray=True)}
             for s in sims]
    stacks = [{"origin": v["origin"], "spacing": v["spacing"], "shape": np.array(v["data"].shape)} for v in views]
    want = ro.get_overlap_bboxes(stacks[0], A1, stacks[1], A2m)
    got = registration._get_overlap_bboxes(sims[0], sims[1], "stage", None, None)
    assert want is not None and got is not None
    for k in range(2):
        np.testing.assert_allclose(got["lowers"][k], want[0][k], atol=1e-8)
        np.testing.assert_allclose(got["uppers"][k], want[1][k], atol=1e-8)
    assert got["vol"] == pytest.approx(want[2], rel=1e-9)
    # the polytope is not a box: its vertices are not the corners of the intrinsic bounding box
    assert got["vol"] < np.prod(np.asarray(got["uppers"][0]) - np.asarray(got["lowers"][0])) * 0.98

    f_w, m_w, org, sp = ro.sims_to_intrinsic_coord_system(views[0], views[1], A1, A2m, want[0], want[1])
    f_g, m_g = registration.sims_to_intrinsic_coord_system(sims[0], sims[1], "stage", (got["lowers"], got["uppers"]))
    for g_, w_ in ((np.asarray(f_g.data), f_w), (np.asarray(m_g.data), m_w)):
        assert g_.shape == w_.shape and g_.dtype == np.float32
        np.testing.assert_array_equal(np.isnan(g_), np.isnan(w_))
        ok = ~np.isnan(w_)
        np.testing.assert_allclose(g_[ok], w_[ok], rtol=1e-5, atol=1e-6)
    assert np.isnan(m_w).any() and (~np.isnan(m_w)).mean() > 0.3          # a rotated view leaves NaN wedges: the masked paths run

    want_reg = ro.phase_correlation_registration(f_w, m_w, return_debug=True)
    got_reg = registration.phase_correlation_registration(f_w, m_w)
    np.testing.assert_array_equal(got_reg["affine_matrix"], want_reg["affine_matrix"])
    assert abs(got_reg["quality"] - want_reg["quality"]) < 1e-5


def test_register_recovers_hidden_shift_of_a_rotated_view_and_fuses(hip_device):
    from multiview_stitcher_amd import fusion, param_utils, registration
    from multiview_stitcher_amd import spatial_image_utils as si

    hidden = np.array([2.0, -3.0, 2.0])
    sims, (A1, A2, A2m), G = _c4_pair(hidden=hidden)
    registration.register(sims, transform_key="stage", new_transform_key="reg", device=0,
                          groupwise_resolution_kwargs={"reference_view": 0})
    p2 = param_utils.select_time(si.get_affine_from_sim(sims[1], "reg"), 0)
    p1 = param_utils.select_time(si.get_affine_from_sim(sims[0], "reg"), 0)
    np.testing.assert_allclose(p1, A1, atol=1e-9)
    # the registered transform of view 2 must undo the metadata error: compare where the view's centre lands
    ctr = si.get_origin_from_sim(sims[1], asarray=True) + (np.array(sims[1].data.shape) - 1) * si.get_spacing_from_sim(sims[1], asarray=True) / 2
    err_before = np.linalg.norm((A2m[:3, :3] @ ctr + A2m[:3, 3]) - (A2[:3, :3] @ ctr + A2[:3, 3]))
    err_after = np.linalg.norm((p2[:3, :3] @ ctr + p2[:3, 3]) - (A2[:3, :3] @ ctr + A2[:3, 3]))
    assert err_before > 4.0 and err_after < 0.75, (err_before, err_after)

    fused = fusion.fuse(sims, transform_key="reg", output_chunksize={"z": 64, "y": 64, "x": 64})
    f = np.asarray(fused.data, dtype=np.float64).squeeze()
    o = si.get_origin_from_sim(fused, asarray=True)
    sp = si.get_spacing_from_sim(fused, asarray=True)
    # the fused volume shows the ground truth: sample G on the fused grid
    want = ndimage.affine_transform(G.astype(np.float64), np.diag(sp), offset=o, output_shape=f.shape, order=1, mode="constant", cval=np.nan)
    inner = tuple(slice(6, -6) for _ in range(3))
    m = ~np.isnan(want[inner]) & (f[inner] > 0)
    assert m.mean() > 0.5
    assert np.abs(f[inner][m] - want[inner][m]).mean() < 0.01          # data range is [0, 1]; residual = resampling blur
