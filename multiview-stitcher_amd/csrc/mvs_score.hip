// mvs_score.hip -- candidate scoring of the registration (placeholder until the kernels land).
#include "mvs_internal.h"

extern "C" int mvs_score_candidates(int device, const float*, const float*, int32_t, int32_t, const int64_t*,
                                    const double*, int32_t, int32_t, double, double, double*, double*, int32_t*) {
    MvsContext* c = mvs_ctx(device);
    return mvs_fail(c, MVS_ERR_UNSUPPORTED, "mvs_score_candidates: not built yet");
}
