// mvs_fft.h -- internal: batched complex64 FFT used by the registration kernels.
#pragma once
#include "mvs_internal.h"

// Optional fusions at the two ends of a 3D transform (taken only when the pass in question runs on the register kernels; *used tells):
//   re_src / im_src: the first pass reads the real and imaginary parts from two float arrays of the transform's shape instead of
//                    `data` (the packed pair a + i b of the phase correlation: no pack kernel, no packed array read);
//   peak_val / peak_idx [2]: the last pass does not store its output but reduces it to argmax |Re| and argmax |Im| (lowest flat
//                    index among equals) per workgroup -- peak_val[ch][wg], peak_idx[ch][wg], n_peak = number of workgroups --
//                    the correlation volume of the phase correlation is only ever searched for its peak.
struct MvsFftFuse {
    const float* re_src = nullptr;
    const float* im_src = nullptr;
    float* peak_val[2] = {nullptr, nullptr};
    long long* peak_idx[2] = {nullptr, nullptr};
    int peak_cap = 0;        // entries available per peak array
    int n_peak = 0;          // out: workgroups of the last pass (0: not fused, the caller searches `data`)
    bool src_used = false;   // out
};
int mvs_fft3_c2c(MvsContext* c, float2* data, const int64_t shape[3], bool inverse, MvsFftFuse* fuse = nullptr);
bool mvs_fft_reg_length(int n);      // does a line of n samples run on the kernels that carry the fusions?
