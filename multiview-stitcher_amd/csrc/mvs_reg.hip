// mvs_reg.hip -- registration entry points (placeholder; see mvs_hip.h).
#include "mvs_internal.h"

extern "C" int mvs_phasecorr(int device, const float*, const float*, int32_t, int32_t, const int64_t*, int32_t,
                             int32_t, double*, int64_t*, float*) {
    MvsContext* c = mvs_ctx(device);
    return mvs_fail(c, MVS_ERR_UNSUPPORTED, "mvs_phasecorr: not built yet");
}

extern "C" int mvs_score_candidates(int device, const float*, const float*, int32_t, int32_t, const int64_t*,
                                    const double*, int32_t, int32_t, double, double, double*, double*, int32_t*) {
    MvsContext* c = mvs_ctx(device);
    return mvs_fail(c, MVS_ERR_UNSUPPORTED, "mvs_score_candidates: not built yet");
}
