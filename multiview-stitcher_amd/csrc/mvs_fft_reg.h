// mvs_fft_reg.h -- internal: the register transforms (radix 4 / 8 / 16 butterflies) and twiddle helpers shared by the short-line
// kernels of mvs_fft.hip (fft_reg2_kernel, bluestein_reg_kernel) and the slab / fused long-axis kernels of mvs_fft_slab*.hip.
#pragma once
#include <hip/hip_runtime.h>

namespace {

__device__ __forceinline__ float2 cmul(float2 a, float2 b) {
    return make_float2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x);
}
__device__ __forceinline__ float2 tw_at(const float2* tw, int m, int half) {   // exp(-2 pi i m / M) for m < M; tw holds m < M / 2
    const float2 w = tw[m & (half - 1)];
    return (m & half) ? make_float2(-w.x, -w.y) : w;
}
__device__ __forceinline__ float2 cadd(float2 a, float2 b) { return make_float2(a.x + b.x, a.y + b.y); }
__device__ __forceinline__ float2 csub(float2 a, float2 b) { return make_float2(a.x - b.x, a.y - b.y); }
__device__ __forceinline__ float2 mul_mi(float2 a) { return make_float2(a.y, -a.x); }          // a * (-i)

__device__ __forceinline__ void dft4(float2& a0, float2& a1, float2& a2, float2& a3) {
    const float2 s02 = cadd(a0, a2), d02 = csub(a0, a2), s13 = cadd(a1, a3), d13 = mul_mi(csub(a1, a3));
    a0 = cadd(s02, s13); a1 = cadd(d02, d13); a2 = csub(s02, s13); a3 = csub(d02, d13);
}
template <int R> __device__ __forceinline__ void dft_reg(float2 (&v)[R]);
template <> __device__ __forceinline__ void dft_reg<4>(float2 (&v)[4]) { dft4(v[0], v[1], v[2], v[3]); }
template <> __device__ __forceinline__ void dft_reg<8>(float2 (&v)[8]) {
    // 8 = 4 x 2: DFT4 over q of v[2 q + r] (r = 0, 1), twiddle W8^(r p), DFT2 over r: X[p + 4 k] for k = 0, 1
    dft4(v[0], v[2], v[4], v[6]);
    dft4(v[1], v[3], v[5], v[7]);
    constexpr float h = 0.70710678118654752440f;
    const float2 t1 = make_float2((v[3].x + v[3].y) * h, (v[3].y - v[3].x) * h);       // * W8^1
    const float2 t2 = mul_mi(v[5]);                                                    // * W8^2
    const float2 t3 = make_float2((v[7].y - v[7].x) * h, -(v[7].x + v[7].y) * h);      // * W8^3
    const float2 e0 = v[0], e1 = v[2], e2 = v[4], e3 = v[6], o0 = v[1];
    v[0] = cadd(e0, o0); v[4] = csub(e0, o0);
    v[1] = cadd(e1, t1); v[5] = csub(e1, t1);
    v[2] = cadd(e2, t2); v[6] = csub(e2, t2);
    v[3] = cadd(e3, t3); v[7] = csub(e3, t3);
}
template <> __device__ __forceinline__ void dft_reg<16>(float2 (&v)[16]) {
    // 16 = 4 x 4: DFT4 over q of v[4 q + r], twiddle W16^(r p), DFT4 over r: X[p + 4 k]
    constexpr float c1 = 0.92387953251128675613f, s1 = 0.38268343236508977173f, h = 0.70710678118654752440f;
#pragma unroll
    for (int r = 0; r < 4; ++r) dft4(v[r], v[4 + r], v[8 + r], v[12 + r]);        // now v[4 p + r] holds y_r[p]
    auto cm = [](float2 a, float wr, float wi) { return make_float2(a.x * wr - a.y * wi, a.x * wi + a.y * wr); };
    // W16^k = (cos(pi k / 8), -sin(pi k / 8))
    v[4 + 1] = cm(v[4 + 1], c1, -s1);  v[4 + 2] = cm(v[4 + 2], h, -h);    v[4 + 3] = cm(v[4 + 3], s1, -c1);
    v[8 + 1] = cm(v[8 + 1], h, -h);    v[8 + 2] = mul_mi(v[8 + 2]);       v[8 + 3] = cm(v[8 + 3], -h, -h);
    v[12 + 1] = cm(v[12 + 1], s1, -c1); v[12 + 2] = cm(v[12 + 2], -h, -h); v[12 + 3] = cm(v[12 + 3], -c1, s1);
#pragma unroll
    for (int p = 0; p < 4; ++p) dft4(v[4 * p], v[4 * p + 1], v[4 * p + 2], v[4 * p + 3]);   // v[4 p + k] = X[p + 4 k]
    // natural order: X[p + 4 k] sits at 4 p + k -> transpose the 4 x 4 block
#pragma unroll
    for (int p = 0; p < 4; ++p)
#pragma unroll
        for (int k = p + 1; k < 4; ++k) { const float2 t = v[4 * p + k]; v[4 * p + k] = v[4 * k + p]; v[4 * k + p] = t; }
}


}  // namespace
