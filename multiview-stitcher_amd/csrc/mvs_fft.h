// mvs_fft.h -- internal: batched complex64 FFT used by the registration kernels.
#pragma once
#include "mvs_internal.h"

// Optional fusions at the two ends of a 3D transform (taken only when the pass in question runs on the register kernels; *used tells):
//   re_src / im_src: the first pass reads the real and imaginary parts from two float arrays of the transform's shape instead of
//                    `data` (the packed pair a + i b of the phase correlation: no pack kernel, no packed array read);
//   peak_val / peak_idx [2]: the last pass does not store its output but reduces it to argmax |Re| and argmax |Im| (lowest flat
//                    index among equals) per workgroup -- peak_val[ch][wg], peak_idx[ch][wg], n_peak = number of workgroups --
//                    the correlation volume of the phase correlation is only ever searched for its peak.
//   xp_src / xp_p2: (inverse transform along x first) the first pass builds its input from the packed spectrum Z = fft(a + i b)
//                    itself -- the cross power p = A conj(B) (stored to xp_p2 for the sub-pixel refinement) and the combination
//                    of the two normalisations that xpower_packed_kernel would have written -- so that array is never stored.
struct MvsFftFuse {
    const float* re_src = nullptr;
    const float* im_src = nullptr;
    const float2* xp_src = nullptr;
    float2* xp_p2 = nullptr;
    int xp_sel_a = 0, xp_sel_b = 0;
    bool xp_used = false;    // out
    float* peak_val[2] = {nullptr, nullptr};
    long long* peak_idx[2] = {nullptr, nullptr};
    int peak_cap = 0;        // entries available per peak array
    int n_peak = 0;          // out: workgroups of the last pass (0: not fused, the caller searches `data`)
    bool src_used = false;   // out
};
int mvs_fft3_c2c(MvsContext* c, float2* data, const int64_t shape[3], bool inverse, MvsFftFuse* fuse = nullptr);
// The three-pass form for crops with one short axis (mvs_fft_slab.hip): which axis is the short one (-1: not of that kind), the
// number of per-workgroup peak entries its last pass writes, and the passes themselves.
int mvs_phasecorr_slab_axis(const MvsContext* c, const int64_t shape[3]);
int mvs_phasecorr_slab_peaks(const int64_t shape[3], int short_axis);
int mvs_phasecorr_slab(MvsContext* c, const float* a, const float* b, float2* Z, float2* CC, float2* P2, float2* dc, const int64_t shape[3],
                       int short_axis, int sel_a, int sel_b, float* const peak_val[2], long long* const peak_idx[2], float2* z0_out);
bool mvs_fft_reg_length(int n);      // does a line of n samples run on the kernels that carry the fusions?

// ---- the cross-power arithmetic of the phase correlation, shared by xpower_packed_kernel (mvs_reg.hip) and the fused first pass ----
// channel scales of the packed pair of correlations (exact powers of two, see xpower_packed_kernel)
__device__ __forceinline__ void mvs_xpower_scales(float2 z0, long long n, float* scale_phase, float* scale_plain) {
    const float dc = fabsf(z0.x * z0.y);           // sum(a) * sum(b): the DC term of the plain cross power
    *scale_plain = (dc > 0.f && dc < INFINITY) ? ldexpf(1.f, -ilogbf(dc)) : 1.f;
    *scale_phase = ldexpf(1.f, -ilogbf((float)n));
}
// z = Z(k), m = Z(-k): p = A conj(B), p1 = p / max(|p|, 100 eps); returns the input of the inverse transform
// (sel_b < 0: one correlation, sel_a picks p1 / p; else sa pa + i sb pb with the channel scales)
__device__ __forceinline__ float2 mvs_xpower_value(float2 z, float2 m, int sel_a, int sel_b, float scale_phase, float scale_plain,
                                                   float2* p_out, float2* p1_out) {
    const float floor_ = 100.f * 1.1920928955078125e-7f;      // 100 * FLT_EPSILON
    const float2 f = make_float2(0.5f * (z.x + m.x), 0.5f * (z.y - m.y));
    const float2 g = make_float2(0.5f * (z.y + m.y), -0.5f * (z.x - m.x));
    const float2 p = make_float2(f.x * g.x + f.y * g.y, f.y * g.x - f.x * g.y);   // f * conj(g)
    const float a = fmaxf(hypotf(p.x, p.y), floor_);
    const float2 p1 = make_float2(p.x / a, p.y / a);
    *p_out = p;
    *p1_out = p1;
    const float2 pa = sel_a ? p1 : p;
    if (sel_b < 0) return pa;
    const float2 pb = sel_b ? p1 : p;
    const float sa = sel_a ? scale_phase : scale_plain, sb = sel_b ? scale_phase : scale_plain;
    return make_float2(pa.x * sa - pb.y * sb, pa.y * sa + pb.x * sb);
}
