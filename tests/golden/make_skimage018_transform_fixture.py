"""Generates tests/golden/skimage018_transforms.npz -- run with /opt/conda/bin/python3.9 (scikit-image 0.18.3).

The reference's global optimisation estimates per-view rigid / similarity transforms with skimage's
EuclideanTransform / SimilarityTransform.estimate (global_optimization.py:248-259, 349-356); scikit-image 0.26 is not
installable here, 0.18.3 carries the same Umeyama fit.  The vectors pin param_resolution._umeyama.
"""
import numpy as np
import skimage
from skimage.transform import EuclideanTransform, SimilarityTransform
from skimage.transform._geometric import _umeyama

assert skimage.__version__.startswith("0.18"), skimage.__version__
rng = np.random.default_rng(7)
out = {}
for k, (ndim, npts) in enumerate([(2, 4), (2, 12), (3, 8), (3, 24), (2, 8), (3, 16)]):
    src = rng.normal(0, 50, (npts, ndim))
    ang = rng.normal(0, 0.2)
    R = np.eye(ndim)
    R[:2, :2] = [[np.cos(ang), -np.sin(ang)], [np.sin(ang), np.cos(ang)]]
    if ndim == 3:
        b = rng.normal(0, 0.1)
        R = R @ np.array([[1, 0, 0], [0, np.cos(b), -np.sin(b)], [0, np.sin(b), np.cos(b)]])
    dst = (1.0 + rng.normal(0, 0.05)) * src @ R.T + rng.normal(0, 10, ndim) + rng.normal(0, 0.5, (npts, ndim))
    out[f"c{k}_src"], out[f"c{k}_dst"] = src, dst
    for name, cls in [("rigid", EuclideanTransform), ("similarity", SimilarityTransform)]:
        # 0.18.3's classes are 2D only; their estimate() is _umeyama(src, dst, estimate_scale), which is n-D (the later
        # n-D classes call the same function)
        params = _umeyama(src, dst, name == "similarity")
        if ndim == 2:
            t = cls()
            assert t.estimate(src, dst)
            assert np.array_equal(np.asarray(t.params), params)
        out[f"c{k}_{name}"] = np.asarray(params, dtype=np.float64)
# Otsu threshold of 1-D samples (mv_graph.py:858-881 calls skimage.filters.threshold_otsu on the edge overlaps)
from skimage.filters import threshold_otsu
for k, vals in enumerate([np.r_[np.full(12, 5120.0), np.full(8, 100.0)], rng.gamma(2.0, 50.0, 40), np.r_[rng.normal(10, 1, 30), rng.normal(50, 5, 9)],
                          np.array([3.0, 3.0, 3.0, 7.0]), np.array([1.0, 2.0])]):
    out[f"otsu{k}_vals"], out[f"otsu{k}_thr"] = vals, np.array(threshold_otsu(vals))
out["n_otsu"] = np.array(5)
out["n_cases"] = np.array(6)
np.savez_compressed(__file__.replace("make_skimage018_transform_fixture.py", "skimage018_transforms.npz"), **out)
print("wrote", len(out), "arrays")
