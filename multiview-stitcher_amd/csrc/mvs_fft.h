// mvs_fft.h -- internal: batched complex64 FFT used by the registration kernels.
#pragma once
#include "mvs_internal.h"

int mvs_fft3_c2c(MvsContext* c, float2* data, const int64_t shape[3], bool inverse);
