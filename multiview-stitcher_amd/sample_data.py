"""Synthetic tiled datasets with known-answer shifts (host side, numpy/scipy).

Plays the role of the reference's ``sample_data.generate_tiled_dataset``
(src/multiview_stitcher/sample_data.py:11-140) for tests and benchmarks: a
seeded ground-truth image (smoothed uniform noise) is cut into a regular grid
of overlapping tiles; each tile is displaced by an integer jitter that the
stage metadata (the ``affine_metadata`` translation) does NOT contain, so
registration has a known answer.
"""

from __future__ import annotations

import numpy as np
from scipy import ndimage

from . import spatial_image_utils as si_utils

METADATA_TRANSFORM_KEY = si_utils.DEFAULT_TRANSFORM_KEY


def make_ground_truth(shape, dtype=np.uint16, seed=0, sigma=2.0):
    rng = np.random.default_rng(seed)
    gt = ndimage.gaussian_filter(rng.random(tuple(shape), dtype=np.float32), sigma)
    gt -= gt.min()
    gt /= max(float(gt.max()), 1e-12)
    if np.issubdtype(np.dtype(dtype), np.integer):
        gt = (gt * 4095.0).astype(dtype)
    else:
        gt = gt.astype(dtype)
    return gt


def generate_tiled_dataset(
    ndim=2,
    tile_shape=64,
    tiles=(2, 2),
    overlap=12,
    dtype=np.uint16,
    spacing=None,
    max_jitter=3,
    seed=0,
    transform_key=METADATA_TRANSFORM_KEY,
    jitter_in_metadata=False,
):
    """Return (sims, jitters, ground_truth).

    ``tiles`` = grid counts in (z,)y,x; ``tile_shape`` int or per-dim tuple;
    ``overlap`` pixels shared by neighbours; ``jitters[i]`` = integer pixel
    displacement of tile i's content (its true origin is nominal + jitter)."""
    sdims = ["z", "y", "x"][-ndim:]
    tile_shape = np.broadcast_to(np.asarray(tile_shape), (ndim,)).astype(int)
    tiles = np.asarray(tiles).astype(int)
    overlap = np.broadcast_to(np.asarray(overlap), (ndim,)).astype(int)
    spacing = np.ones(ndim) if spacing is None else np.asarray(spacing, dtype=float)
    step = tile_shape - overlap
    pad = max_jitter + 1
    gt_shape = step * (tiles - 1) + tile_shape + 2 * pad
    gt = make_ground_truth(gt_shape, dtype, seed)
    rng = np.random.default_rng(seed + 1)
    sims, jitters = [], []
    for idx in np.ndindex(*tiles):
        idx = np.asarray(idx)
        jitter = rng.integers(-max_jitter, max_jitter + 1, size=ndim) if max_jitter > 0 else np.zeros(ndim, int)
        if not idx.any():
            jitter[:] = 0
        start = idx * step + pad + jitter
        sl = tuple(slice(int(s), int(s + n)) for s, n in zip(start, tile_shape))
        data = np.ascontiguousarray(gt[sl])
        nominal = (idx * step + (jitter if jitter_in_metadata else 0)) * spacing
        sim = si_utils.get_sim_from_array(
            data,
            dims=sdims,
            scale=dict(zip(sdims, spacing)),
            translation=dict(zip(sdims, nominal)),
            transform_key=transform_key,
        )
        sims.append(sim)
        jitters.append(jitter)
    return sims, np.array(jitters), gt
