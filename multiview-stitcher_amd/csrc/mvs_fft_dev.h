// mvs_fft_dev.h -- internal: launch arguments and device helpers shared by the FFT kernels of mvs_fft.hip and the whole-line
// DFT kernels of mvs_fft_lines*.hip (the latter are built in units of their own: ~30 fully unrolled instantiations).
#pragma once
#include "mvs_fft.h"

struct FftArgs {
    float2* data;
    long long n_lines;        // number of lines of this pass
    int n;                    // transform length
    int M, log2M;             // Stockham size (== n for powers of two)
    long long stride;         // element stride along the transform axis
    long long inner;          // lines are enumerated l = outer * inner + i: base = outer * outer_stride + i
    long long outer_stride;
    int lpb;                  // lines per workgroup
    int inverse;              // 1: conjugate transform (unnormalised)
    const float2* tw;         // exp(-2 pi i m / M), m < M/2
    const float2* chirp;      // Bluestein: w_n = exp(-i pi n^2 / N), n < N
    const float2* bfft;       // Bluestein: FFT_M of the wrapped conj chirp, scaled by 1/M
    // fusions of the register kernels (nullptr: off), see MvsFftFuse
    const float* re_src = nullptr;
    const float* im_src = nullptr;
    float* peak_val[2] = {nullptr, nullptr};
    long long* peak_idx[2] = {nullptr, nullptr};
    const float2* xp_src = nullptr;     // x pass only (stride 1, line = kz * xp_ny + ky)
    float2* xp_p2 = nullptr;
    int xp_ny = 0, xp_nz = 0, xp_sel_a = 0, xp_sel_b = 0;
    int xp_pair = 0;                    // fft_reg2_kernel: a workgroup's lines come as partner pairs (kz, ky), (-kz, -ky): see there
};

// element offset of line l of this pass
__device__ __forceinline__ long long line_base(const FftArgs& A, int l) {
    const int inner = (int)A.inner;
    return (long long)(l / inner) * A.outer_stride + (l % inner);
}

__device__ __forceinline__ float2 fft_load(const FftArgs& A, long long i) {
    if (!A.re_src) return A.data[i];
    float re = A.re_src[i], im = A.im_src[i];      // the pair a + i b of two real volumes, NaN -> 0 (np.nan_to_num, registration.py:403-408)
    re = (re != re) ? 0.f : re;
    im = (im != im) ? 0.f : im;
    return make_float2(re, im);
}
// The first pass of the inverse transform of the phase correlation, along x: its input is made from the packed spectrum Z (xp_src)
// and its mirror Z(-k) on the fly, the cross power goes to xp_p2 (mvs_xpower_value, mvs_fft.h).
struct XpLine { long long mbase; float scale_phase, scale_plain; };
__device__ __forceinline__ XpLine xp_line(const FftArgs& A, long long l) {
    XpLine x;
    const int ky = (int)(l % A.xp_ny), kz = (int)(l / A.xp_ny);
    const int my = ky ? A.xp_ny - ky : 0, mz = kz ? A.xp_nz - kz : 0;
    x.mbase = ((long long)mz * A.xp_ny + my) * A.n;
    x.scale_phase = x.scale_plain = 1.f;
    if (A.xp_sel_b >= 0) mvs_xpower_scales(A.xp_src[0], (long long)A.xp_nz * A.xp_ny * A.n, &x.scale_phase, &x.scale_plain);
    return x;
}
__device__ __forceinline__ float2 xp_load(const FftArgs& A, const XpLine& xl, long long base, int kx) {
    const int mx = kx ? A.n - kx : 0;
    float2 p, p1;
    const float2 v = mvs_xpower_value(A.xp_src[base + kx], A.xp_src[xl.mbase + mx], A.xp_sel_a, A.xp_sel_b, xl.scale_phase, xl.scale_plain, &p, &p1);
    A.xp_p2[base + kx] = p;
    return v;
}
// running argmax |Re| / |Im| with the lowest flat index among equal values (np.argmax)
struct Peak2 { float v[2]; long long i[2]; };
__device__ __forceinline__ void peak_init(Peak2& p) { p.v[0] = p.v[1] = -1.f; p.i[0] = p.i[1] = 0x7fffffffffffffffLL; }
__device__ __forceinline__ void peak_add(Peak2& p, float2 o, long long idx) {
    const float a[2] = {fabsf(o.x), fabsf(o.y)};
#pragma unroll
    for (int k = 0; k < 2; ++k)
        if (a[k] > p.v[k] || (a[k] == p.v[k] && idx < p.i[k])) { p.v[k] = a[k]; p.i[k] = idx; }
}
// ---- two axes of a 3D transform in one pass: the slab kernels (mvs_fft_slab.inc) ---------------------------------------------------
// A slab = all samples of two axes at one index of the third: S (short axis: a whole-line DFT length, mvs_dft_small.h) x L (64, 128
// or 256) complex samples, transformed along both axes by one workgroup.  Two layouts in memory:
//   rows (short_contig = 0): S rows of L contiguous samples, row_stride elements apart (crops (S, ., L): rows = z at one y; crops
//                            (., S, L): rows = y at one z);
//   short_contig = 1:        L rows of S contiguous samples back to back (crops (., L, S) at one z: the slab is one contiguous block).
struct SlabArgs {
    float2* data = nullptr;             // in place; forward with re_src / im_src: output only
    const float* re_src = nullptr;      // forward: the pair a + i b of two real volumes (NaN -> 0), see fft_load
    const float* im_src = nullptr;
    int S = 0, L = 0;
    int short_contig = 0;
    int inverse = 0;
    long long slab_stride = 0;          // elements between slabs
    long long row_stride = 0;           // rows layout: elements between the slab's rows
    const float2* tw = nullptr;         // exp(-2 pi i m / L), m < L / 2
    float2* dc = nullptr;               // forward: element (0, 0) of every slab's 2D spectrum (their sum is the DC term of the 3D one)
    float* peak_val[2] = {nullptr, nullptr};      // inverse: argmax |Re| / |Im| per workgroup instead of a store (Peak2)
    long long* peak_idx[2] = {nullptr, nullptr};
};
bool mvs_launch_slab_lo(MvsContext* c, const SlabArgs& A, unsigned grid, size_t lds_bytes, hipError_t* err);     // 17 <= S <= 44
bool mvs_launch_slab_hi(MvsContext* c, const SlabArgs& A, unsigned grid, size_t lds_bytes, hipError_t* err);     // 45 <= S <= 64

// the remaining (long) axis of the phase correlation in ONE pass: forward transform of a line and of its partner line, cross
// power (mvs_xpower_value), inverse transform of both -- long_xp_kernel (mvs_fft_slab.hip)
struct LongArgs {
    const float2* Z = nullptr;          // spectrum along the two slab axes (in)
    float2* CC = nullptr;               // out: packed correlation input, inverse transformed along this axis
    float2* P2 = nullptr;               // out: plain cross power (for the upsampled refinement)
    long long stride = 0;               // element stride along the transform axis
    int PA = 0, PB = 0;                 // the lines' plane: PA rows of PB adjacent lines; line (a, b) starts at a * pa_stride + b
    long long pa_stride = 0;
    int sel_a = 0, sel_b = 0;
    const float2* tw = nullptr;
    const float2* dc = nullptr;         // per-slab DC terms (SlabArgs::dc), ndc <= 256 of them
    int ndc = 0;
    long long ntotal = 0;               // samples of the volume (scale of the phase channel)
    float2* z0_out = nullptr;           // workgroup 0 writes the DC term of the packed spectrum here (host-visible)
};
int mvs_fft_twiddles(MvsContext* c, int n, const float2** tw);

// lengths the whole-line kernel is built for: non-powers of two from 17 to 64 whose prime factors are <= 19 (mvs_dft_small.h)
constexpr bool mvs_dft_line_length(int n) {
    if (n < 17 || n > 64 || (n & (n - 1)) == 0) return false;
    for (int p = 2; p <= 19; ++p)
        while (n % p == 0) n /= p;
    return n == 1;
}
// launches dft_line_kernel<A.n> on c->stream (grid workgroups of 64 lines); false: no kernel for this length
bool mvs_launch_dft_line_lo(MvsContext* c, const FftArgs& A, unsigned grid);     // 17 <= n <= 44
bool mvs_launch_dft_line_hi(MvsContext* c, const FftArgs& A, unsigned grid);     // 45 <= n <= 64
