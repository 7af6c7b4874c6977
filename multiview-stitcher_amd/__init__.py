"""multiview_stitcher_amd -- MI355X-native register+fuse hot path for multiview-stitcher workflows.

Host-side mirror of the reference's plugin API for this path
(``registration.register`` / ``fusion.fuse`` / ``msi_utils`` /
``spatial_image_utils``) on top of ``libmvs_hip.so`` (hand-written HIP kernels
for gfx950 behind the C ABI of ``include/mvs_hip.h``).  There is no CPU
fallback: every compute entry point raises if the HIP library is missing."""

__version__ = "0.1.0"
