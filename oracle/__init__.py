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
  tests (``tests/test_fuse_oracle.py``) and by non-constant blending-weight
  vectors derived by hand (``tests/test_weights_oracle.py``).
  The registration half depends on scikit-image 0.26 (not installed, not in
  the reference tree): its published algorithm (Guizar-Sicairos upsampled-DFT
  phase correlation, Wang SSIM) is restated in ``reg_oracle.py``.  The
  reference's own tests hold no golden vectors for it (tolerance tests only),
  so it is pinned by vectors EXECUTED with scikit-image 0.18.3
  (``tests/golden/skimage018_pcc.npz``, ``skimage018_round2.npz``; generator
  ``tests/golden/make_skimage018_fixture.py`` under /opt/conda/bin/python3.9):
  unnormalised phase correlation and its upsampled-DFT refinement, the
  "phase"-normalised variant through 0.18.3's own machinery around the one
  published normalisation line, the masked variant with inverted masks on
  NaN-holding images (quirk Q1 -> zero shift), ``rescale_intensity`` values
  (Q5) and SSIM for float64 and float32 inputs (Q4), plus the reference's
  tolerance tests (``tests/test_reg_oracle.py``).
* ``plan_oracle.py``: literal restatement of the reference's chunk -> view-slab planner (fusion/_core.py:354-722); the
  product plans with the library (``mvs_fuse_plan``) and ``tests/test_plan_oracle.py`` compares window for window.
"""
