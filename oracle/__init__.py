"""CPU oracle for the register+fuse hot path  --  TEST INFRASTRUCTURE ONLY.

This package is a numpy/scipy restatement of the arithmetic of the reference's
hot path (multiview-stitcher: ``registration.phase_correlation_registration``
and ``fusion.fuse_np`` with everything below them).  It exists so that the HIP
path can be checked against the reference's algorithm on the same inputs.

Rules (see DESIGN.md "Oracle"):

* Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s
  ``cpu_baseline`` leg may import anything from here.  The product package
  (``multiview_stitcher_amd``) never imports it and fails loudly when the HIP
  library is missing.
* Every function cites the reference file:line it follows (paths relative to
  the reference checkout, ``src/multiview_stitcher/...``).
* Third-party arithmetic: the fusion half calls the *same* scipy 1.15.3
  functions the reference calls (``scipy.ndimage.affine_transform``,
  ``distance_transform_edt``, ``gaussian_filter``), so that half is pinned by
  the library itself plus the reference's own constant-tile known-answer
  tests (``tests/test_oracle_reference_kats.py``).
  The registration half depends on scikit-image 0.26 (not installed, not in
  the reference tree): its published algorithm (Guizar-Sicairos upsampled-DFT
  phase correlation, Wang SSIM) is restated in ``reg_oracle.py``;
  exact peak indices / sub-pixel shifts are **parity unpinned** by any golden
  vector of the reference (its tests are tolerance tests only) and are pinned
  here by those tolerance tests + cross-checks against skimage 0.18.3.
"""
