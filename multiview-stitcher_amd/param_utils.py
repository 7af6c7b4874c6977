"""Affine parameter helpers (numpy-backed mirror of the reference's param_utils).

Reference: src/multiview_stitcher/param_utils.py.  The reference stores affines
as xarray.DataArray with dims ([t,] x_in, x_out); without xarray here an affine
is a float64 ndarray of shape ([T,] ndim+1, ndim+1), homogeneous, axis order
(z,)y,x,1.
"""

from __future__ import annotations

import numpy as np


def affine_from_translation(translation):
    """param_utils.affine_from_translation (param_utils.py:7-14)."""
    translation = np.asarray(translation, dtype=np.float64)
    ndim = len(translation)
    M = np.eye(ndim + 1)
    M[:ndim, ndim] = translation
    return M


def affine_from_linear_affine(linear_affine):
    """param_utils.py:17-28: [matrix.flatten(), translation] -> homogeneous."""
    linear_affine = np.asarray(linear_affine, dtype=np.float64)
    ndim = 3 if len(linear_affine) == 12 else 2
    M = np.eye(ndim + 1)
    M[:ndim, :ndim] = linear_affine[: ndim**2].reshape((ndim, ndim))
    M[:ndim, ndim] = linear_affine[-ndim:]
    return M


def linear_affine_from_affine(affine):
    affine = np.asarray(affine)
    ndim = affine.shape[-1] - 1
    out = np.zeros(ndim**2 + ndim)
    out[: ndim**2] = affine[:ndim, :ndim].flatten()
    out[-ndim:] = affine[:ndim, ndim]
    return out


def identity_transform(ndim, t_coords=None):
    """param_utils.identity_transform (param_utils.py:124-125)."""
    return affine_to_xaffine(np.eye(ndim + 1), t_coords=t_coords)


def affine_to_xaffine(affine, t_coords=None):
    """param_utils.affine_to_xaffine (param_utils.py:128-150): optionally t-stacked."""
    affine = np.asarray(affine, dtype=np.float64)
    if t_coords is None:
        return affine.copy()
    return np.stack([affine] * len(t_coords), axis=0)


def select_time(xaffine, it=0):
    """The ([t,] n, n) affine at time index ``it`` (broadcast if not t-stacked)."""
    xaffine = np.asarray(xaffine, dtype=np.float64)
    if xaffine.ndim == 3:
        return xaffine[min(it, xaffine.shape[0] - 1)]
    return xaffine


def matmul_xparams(a, b):
    """param_utils.matmul_xparams (param_utils.py:192-203)."""
    return np.matmul(np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64))


def invert_xparams(a):
    """param_utils.invert_xparams (param_utils.py:206-216)."""
    return np.linalg.inv(np.asarray(a, dtype=np.float64))


def rebase_affine(xaffine, base_affine):
    """param_utils.rebase_affine (param_utils.py:219-243): chain xaffine @ base."""
    return matmul_xparams(xaffine, base_affine)


def translation_from_affine(affine):
    affine = np.asarray(affine)
    ndim = affine.shape[-1] - 1
    return affine[..., :ndim, ndim]


def expand_affine_dims(affine, ndim_out=3):
    """Embed a 2D (y,x) affine into 3D leaving z untouched (param_utils.py:153-189)."""
    affine = np.asarray(affine, dtype=np.float64)
    nd = affine.shape[-1] - 1
    if nd == ndim_out:
        return affine
    out = np.eye(ndim_out + 1)
    k = ndim_out - nd
    out[k:, k:] = affine
    return out
