// mvs_gauss.hip -- content-based fusion weights (weights.py:22-74). Placeholder until the
// separable NaN-aware Gaussian kernels land; reports MVS_ERR_UNSUPPORTED loudly.
#include "mvs_internal.h"

int mvs_fuse_content_based(MvsContext* c, const mvs_view_t*, int32_t, const mvs_fuse_opts_t*, void*) {
    return mvs_fail(c, MVS_ERR_UNSUPPORTED, "content_based weights: HIP kernels not built yet");
}
