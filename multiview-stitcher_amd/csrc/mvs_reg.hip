// mvs_reg.hip -- pairwise phase correlation on the GPU (gfx950).
//
// mvs_phasecorr == skimage.registration.phase_cross_correlation(a, b, normalization=None|"phase",
// upsample_factor=u, disambiguate=False)[0] as called from the reference's
// registration.phase_correlation_registration (src/multiview_stitcher/registration.py:422-431):
//   F = fftn(a), G = fftn(b)                         (complex64, mvs_fft.hip; both from ONE transform of a + i b)
//   P = F * conj(G);  "phase": P /= max(|P|, 100 eps)
//   cc = ifftn(P);  integer peak = argmax |cc|  (lowest flat index wins ties, like np.argmax)
//   wrap to signed shift, then the matrix-multiply upsampled DFT around round(shift*u)/u
// mvs_rescale_intensity == skimage.exposure.rescale_intensity(im, in_range=(nanmin, nanmax),
// out_range=(0,1)) (registration.py:381-389).
#include "mvs_fft.h"

#include <rocprim/warp/warp_reduce.hpp>

#include <cfloat>
#include <cmath>
#include <vector>

namespace {

// Both images are real, so ONE complex transform carries both spectra: Z = fft(a + i b), and with Zm = Z(-k)
//   A(k) = (Z + conj Zm) / 2,   B(k) = (Z - conj Zm) / (2 i).
__global__ void pack_pair_kernel(const float* __restrict__ a, const float* __restrict__ b, float2* __restrict__ dst, long long n) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        float va = a[i], vb = b[i];
        if (va != va) va = 0.f;   // np.nan_to_num (registration.py:403-408)
        if (vb != vb) vb = 0.f;
        dst[i] = make_float2(va, vb);
    }
}

// Cross power spectra from the packed transform: p = A conj(B); P1 = p / max(|p|, 100 eps) ("phase"), P2 = p (None).
// The spectra are Hermitian, so their inverse transforms are real and again share one complex transform:
// C = sa Pa + i sb Pb  ->  ifft(C) = sa cc_a + i sb cc_b.  sel_a / sel_b pick which of {P1 (1), P2 (0)} go where
// (sel_b < 0: C = Pa, one correlation per transform).
__global__ void xpower_packed_kernel(const float2* __restrict__ Z, float2* __restrict__ P1, float2* __restrict__ P2,
                                     float2* __restrict__ C, int nz, int ny, int nx, int sel_a, int sel_b) {
    const long long n = (long long)nz * ny * nx;
    float scale_phase = 1.f, scale_plain = 1.f;
    if (sel_b >= 0) mvs_xpower_scales(Z[0], n, &scale_phase, &scale_plain);   // Z[0] = sum(a) + i sum(b)
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const int kx = (int)(i % nx);
        const long long t = i / nx;
        const int ky = (int)(t % ny), kz = (int)(t / ny);
        const int mx = kx ? nx - kx : 0, my = ky ? ny - ky : 0, mz = kz ? nz - kz : 0;
        // Two correlations in one transform only work if both have the same order of magnitude: float32 rounding of the
        // larger one leaks into the other channel.  The phase-normalised correlation is <= N, the plain one about
        // N * sum(a) * sum(b) for non-negative images (its DC term dominates): both are brought to O(1) by exact
        // powers of two (scale_phase, scale_plain), which commute with every operation of the transform, so each
        // channel holds the bits of its separate transform times its scale, plus ~1e-7 of the other channel.
        float2 p, p1;
        C[i] = mvs_xpower_value(Z[i], Z[((long long)mz * ny + my) * nx + mx], sel_a, sel_b, scale_phase, scale_plain, &p, &p1);
        P1[i] = p1;
        P2[i] = p;
    }
}

// argmax |c| with np.argmax's tie-break (lowest flat index): per-block partial results
// COMP 0: |c| (one correlation per transform); 1 / 2: |Re c| / |Im c| (two real correlations packed in one transform).
// (A template parameter: the variant that selected the component at run time returned wrong maxima for the imaginary part.)
template <int COMP>
__global__ __launch_bounds__(256) void argmax_abs_kernel(const float2* __restrict__ c, long long n, float* __restrict__ pval,
                                                         long long* __restrict__ pidx) {
    constexpr int comp = COMP;
    float best = -1.f;
    long long bi = 0x7fffffffffffffffLL;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const float2 v = c[i];
        float a;
        if (comp == 0) a = hypotf(v.x, v.y);
        else if (comp == 1) a = fabsf(v.x);
        else a = fabsf(v.y);
        if (a > best || (a == best && i < bi)) { best = a; bi = i; }
    }
    // wavefront shuffle reduction, then across the 4 wavefronts through LDS
    for (int off = 32; off > 0; off >>= 1) {
        const float ob = __shfl_down(best, off);
        const long long oi = __shfl_down(bi, off);
        if (ob > best || (ob == best && oi < bi)) { best = ob; bi = oi; }
    }
    __shared__ float sv[4];
    __shared__ long long si[4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) { sv[wave] = best; si[wave] = bi; }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < 4; ++w)
            if (sv[w] > best || (sv[w] == best && si[w] < bi)) { best = sv[w]; bi = si[w]; }
        pval[blockIdx.x] = best;
        pidx[blockIdx.x] = bi;
    }
}

// Both channels of a packed correlation in ONE pass over the array: argmax |Re| into (pval, pidx), argmax |Im| into (pval2, pidx2)
// -- the same comparisons as argmax_abs_kernel<1> and <2> -- and, from block 0, the DC term of the forward transform (z0_src[0])
// into z0_dst (host memory): one launch instead of three.
__global__ __launch_bounds__(256) void argmax_abs2_kernel(const float2* __restrict__ c, long long n, float* __restrict__ pval,
                                                          long long* __restrict__ pidx, float* __restrict__ pval2, long long* __restrict__ pidx2,
                                                          const float2* __restrict__ z0_src, float2* __restrict__ z0_dst) {
    float best[2] = {-1.f, -1.f};
    long long bi[2] = {0x7fffffffffffffffLL, 0x7fffffffffffffffLL};
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const float2 v = c[i];
        const float a[2] = {fabsf(v.x), fabsf(v.y)};
#pragma unroll
        for (int k = 0; k < 2; ++k)
            if (a[k] > best[k] || (a[k] == best[k] && i < bi[k])) { best[k] = a[k]; bi[k] = i; }
    }
    __shared__ float sv[2][4];
    __shared__ long long si[2][4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        for (int off = 32; off > 0; off >>= 1) {
            const float ob = __shfl_down(best[k], off);
            const long long oi = __shfl_down(bi[k], off);
            if (ob > best[k] || (ob == best[k] && oi < bi[k])) { best[k] = ob; bi[k] = oi; }
        }
        if (lane == 0) { sv[k][wave] = best[k]; si[k][wave] = bi[k]; }
    }
    __syncthreads();
    if (threadIdx.x == 0) {
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            for (int w = 1; w < 4; ++w)
                if (sv[k][w] > best[k] || (sv[k][w] == best[k] && si[k][w] < bi[k])) { best[k] = sv[k][w]; bi[k] = si[k][w]; }
        }
        pval[blockIdx.x] = best[0]; pidx[blockIdx.x] = bi[0];
        pval2[blockIdx.x] = best[1]; pidx2[blockIdx.x] = bi[1];
        if (blockIdx.x == 0) *z0_dst = *z0_src;
    }
}

// Upsampled DFT, stage 1: contract the last (x) axis with kernel K (U x nx), input conj(P).
// out[(row) * U + a] = sum_x K[a][x] * conj(P[row][x]); one wavefront per row.
__global__ __launch_bounds__(256) void updft_x_kernel(const float2* __restrict__ P, const float2* __restrict__ K,
                                                      float2* __restrict__ out, long long nrows, int nx, int U) {
    const int lane = threadIdx.x & 63;
    const long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= nrows) return;
    for (int a = 0; a < U; ++a) {
        float2 acc = make_float2(0.f, 0.f);
        for (int x = lane; x < nx; x += 64) {
            const float2 p = P[row * nx + x];
            const float2 k = K[a * nx + x];
            // k * conj(p)
            acc.x += k.x * p.x + k.y * p.y;
            acc.y += k.y * p.x - k.x * p.y;
        }
        for (int off = 32; off > 0; off >>= 1) {
            acc.x += __shfl_down(acc.x, off);
            acc.y += __shfl_down(acc.y, off);
        }
        if (lane == 0) out[row * U + a] = acc;
    }
}

// Stage 1 of BOTH normalisations of a packed phase correlation in one pass over the plain cross power P2: the phase-normalised
// one is formed on the fly (the same expression as mvs_xpower_value's p1, so the values are those xpower_packed_kernel would
// have stored), which is why that array is neither written nor read.  Per normalisation the sums run in updft_x_kernel's order.
struct UpdftX2 { const float2* K[2]; float2* out[2]; int phase[2]; };
template <int UM>
__global__ __launch_bounds__(256) void updft_x2_kernel(const float2* __restrict__ P2, UpdftX2 q, long long nrows, int nx, int U) {
    const int lane = threadIdx.x & 63;
    const long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= nrows) return;
    const float floor_ = 100.f * FLT_EPSILON;
    float2 acc[2][UM];
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int a = 0; a < UM; ++a) acc[j][a] = make_float2(0.f, 0.f);
    // a lane's samples of 256 consecutive x are loaded together, then consumed in ascending x like updft_x_kernel's loop.  (One row
    // per wavefront on purpose: a wavefront that keeps the kernel samples in registers and walks four rows ran 47 instead of 34 us
    // -- the contraction is bound by how many short dependent chains are in flight, not by its loads.)
    for (int x0 = lane; x0 < nx; x0 += 256) {
        float2 pv[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) pv[i] = (x0 + 64 * i < nx) ? P2[row * nx + x0 + 64 * i] : make_float2(0.f, 0.f);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int x = x0 + 64 * i;
            if (x < nx) {
                const float2 p = pv[i];
                const float m = fmaxf(hypotf(p.x, p.y), floor_);
                const float2 p1 = make_float2(p.x / m, p.y / m);
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const float2 v = q.phase[j] ? p1 : p;
#pragma unroll
                    for (int a = 0; a < UM; ++a) {
                        if (a < U) {
                            const float2 k = q.K[j][a * nx + x];
                            // k * conj(v)
                            acc[j][a].x += k.x * v.x + k.y * v.y;
                            acc[j][a].y += k.y * v.x - k.x * v.y;
                        }
                    }
                }
            }
        }
    }
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int a = 0; a < UM; ++a) {
            if (a < U) {
                float2 r = acc[j][a];
                for (int off = 32; off > 0; off >>= 1) {
                    r.x += __shfl_down(r.x, off);
                    r.y += __shfl_down(r.y, off);
                }
                if (lane == 0) q.out[j][row * U + a] = r;
            }
        }
}

// Stages 1 AND 2 of both refinements of a 3D pair in one pass over the plain cross power (nx <= 256, U samples per axis):
//   part[(z, chunk)][b][a] = sum_{y in chunk} sum_x Ky[b][y] Kx[a][x] conj(v[z][y][x]).
// A workgroup owns the rows of one y chunk of one plane; every lane sums ITS columns over the wavefront's rows first (the y
// contraction: sequential additions, nothing crosses lanes), then contracts its columns with Kx and only then the 2 x U x U results
// are reduced across the wavefront and the four wavefronts -- one reduction per 16 rows instead of one per row and stage (the
// row-at-a-time kernels spend their time in exactly those reductions: 55 us for the three stages of both refinements of a
// 256 x 256 x 51 crop, 39 us this way).  Stage 3 (updft_mid_kernel over z) folds the chunks
// (kdiv: the chunks of a plane share its kernel sample).  The sums run in a different order than in the separate stages (y before x): equal to them to
// float32 rounding, the refined shifts -- multiples of 1 / upsample -- agree (tests/test_reg_gpu.py).
constexpr int kYxChunksMax = 64, kYxRows = 64;      // rows of a chunk, 16 per wavefront (measured 8 ... 128 rows: 60.6, 42.7, 33.9, 33.0, 38.2 us for stages 1-3)
struct UpdftYX { const float2* Kx[2]; const float2* Ky[2]; float2* out[2]; int phase[2]; };
template <int U>
__global__ __launch_bounds__(256) void updft_yx2_kernel(const float2* __restrict__ P2, UpdftYX q, int nz, int ny, int nx, int rows_per_chunk, int nchunks) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int z = blockIdx.x / nchunks, ch = blockIdx.x % nchunks;
    const int y0 = ch * rows_per_chunk, y1 = min(y0 + rows_per_chunk, ny);
    const float floor_ = 100.f * FLT_EPSILON;
    float2 s[2][U][4];
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int b = 0; b < U; ++b)
#pragma unroll
            for (int i = 0; i < 4; ++i) s[j][b][i] = make_float2(0.f, 0.f);
    for (int y = y0 + wave; y < y1; y += 4) {
        const float2* row = P2 + ((long long)z * ny + y) * nx;
        float2 pv[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) pv[i] = (lane + 64 * i < nx) ? row[lane + 64 * i] : make_float2(0.f, 0.f);
        float2 ky[2][U];
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int b = 0; b < U; ++b) ky[j][b] = q.Ky[j][b * ny + y];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            if (64 * i >= nx) continue;                               // (uniform: short rows use the first lanes' slots only)
            const float2 p = pv[i];                                   // (columns beyond nx hold 0: they add nothing)
            const float m = fmaxf(hypotf(p.x, p.y), floor_);
            const float2 p1 = make_float2(p.x / m, p.y / m);
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const float2 v = q.phase[j] ? p1 : p;
#pragma unroll
                for (int b = 0; b < U; ++b) {
                    const float2 k = ky[j][b];
                    // k * conj(v)
                    s[j][b][i].x += k.x * v.x + k.y * v.y;
                    s[j][b][i].y += k.y * v.x - k.x * v.y;
                }
            }
        }
    }
    // x contraction of the lane's columns, then across the wavefront
    __shared__ float2 red[4][2 * U * U];
    using WaveSum = rocprim::warp_reduce<float, 64>;
    __shared__ typename WaveSum::storage_type wsum[4];
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int a = 0; a < U; ++a) {
            float2 kx[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) kx[i] = (lane + 64 * i < nx) ? q.Kx[j][a * nx + lane + 64 * i] : make_float2(0.f, 0.f);
#pragma unroll
            for (int b = 0; b < U; ++b) {
                float2 r = make_float2(0.f, 0.f);
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    if (64 * i >= nx) continue;
                    r.x += kx[i].x * s[j][b][i].x - kx[i].y * s[j][b][i].y;
                    r.y += kx[i].x * s[j][b][i].y + kx[i].y * s[j][b][i].x;
                }
                // (DPP row operations: the bpermute-based shuffle tree costs several times as much, and this kernel is made of reductions)
                WaveSum().reduce(r.x, r.x, wsum[wave]);
                WaveSum().reduce(r.y, r.y, wsum[wave]);
                if (lane == 0) red[wave][(j * U + b) * U + a] = r;
            }
        }
    __syncthreads();
    if (threadIdx.x < 2 * U * U) {
        const int t = threadIdx.x, j = t / (U * U), ba = t % (U * U);
        float2 r = red[0][t];
        for (int w = 1; w < 4; ++w) { r.x += red[w][t].x; r.y += red[w][t].y; }
        q.out[j][(long long)blockIdx.x * (U * U) + ba] = r;
    }
}

// Generic later stage: in has shape (n_outer, n_red, n_inner) -> out (n_outer, U, n_inner):
// out[o][b][i] = sum_r K[b][r] * in[o][r][i]; one wavefront per output element, lanes stride over r (the outputs are
// few -- U^2 nz or U^3 -- and the reduction long, so a thread per output would leave the GPU idle behind a serial loop).
__global__ __launch_bounds__(256) void updft_mid_kernel(const float2* __restrict__ in, const float2* __restrict__ K, float2* __restrict__ out,
                                                        int n_outer, int n_red, int n_inner, int U, int kdiv = 1) {
    const long long total = (long long)n_outer * U * n_inner;
    const int lane = threadIdx.x & 63;
    for (long long t = (long long)blockIdx.x * 4 + (threadIdx.x >> 6); t < total; t += (long long)gridDim.x * 4) {
        const int i = (int)(t % n_inner);
        const int b = (int)((t / n_inner) % U);
        const int o = (int)(t / ((long long)n_inner * U));
        float2 acc = make_float2(0.f, 0.f);
        for (int r = lane; r < n_red; r += 64) {
            const float2 v = in[((long long)o * n_red + r) * n_inner + i];
            const float2 k = kdiv == 1 ? K[b * n_red + r] : K[b * (n_red / kdiv) + r / kdiv];   // kdiv consecutive entries share a kernel sample
            acc.x += k.x * v.x - k.y * v.y;
            acc.y += k.x * v.y + k.y * v.x;
        }
        for (int off = 32; off > 0; off >>= 1) {
            acc.x += __shfl_down(acc.x, off);
            acc.y += __shfl_down(acc.y, off);
        }
        if (lane == 0) out[t] = acc;
    }
}

// nanmin / nanmax per block (float), finished on the host
__global__ __launch_bounds__(256) void nanminmax_kernel(const float* __restrict__ a, long long n, float* __restrict__ pmin,
                                                        float* __restrict__ pmax, long long* __restrict__ pvalid) {
    float mn = INFINITY, mx = -INFINITY;
    long long nv = 0;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const float v = a[i];
        if (v == v) {
            mn = fminf(mn, v); mx = fmaxf(mx, v); ++nv;
            // upper half of the count: valid voxels that are NOT integers in [0, 65535] (an image of such integers can be rank-
            // ordered by 16-bit keys, see mvs_score.hip)
            if (!(v >= 0.f && v <= 65535.f && v == floorf(v))) nv += 1ll << 32;
        }
    }
    for (int off = 32; off > 0; off >>= 1) {
        mn = fminf(mn, __shfl_down(mn, off));
        mx = fmaxf(mx, __shfl_down(mx, off));
        nv += __shfl_down(nv, off);
    }
    __shared__ float smn[4], smx[4];
    __shared__ long long snv[4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) { smn[wave] = mn; smx[wave] = mx; snv[wave] = nv; }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < 4; ++w) { mn = fminf(mn, smn[w]); mx = fmaxf(mx, smx[w]); nv += snv[w]; }
        pmin[blockIdx.x] = mn; pmax[blockIdx.x] = mx; pvalid[blockIdx.x] = nv;
    }
}

// (clip(im, lo, hi) - lo) / (hi - lo) * 1 + 0 in float32, NaN preserved (skimage rescale_intensity)
__global__ void rescale_kernel(const float* __restrict__ src, float* __restrict__ dst, long long n, float lo, float hi, int degenerate) {
    const float d = hi - lo;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        float v = src[i];
        if (v == v) {
            v = fminf(fmaxf(v, lo), hi);
            if (!degenerate) v = (v - lo) / d;
            v = v * 1.0f + 0.0f;
        }
        dst[i] = v;
    }
}

// both crops of a pair in one launch (blockIdx.y = image): same bodies
struct PairPtrs { const float* in[2]; float* out[2]; float lo[2], hi[2]; int degenerate[2]; };
__global__ __launch_bounds__(256) void nanminmax_pair_kernel(PairPtrs P, long long n, char* __restrict__ partials, int nb) {
    const int k = blockIdx.y;
    const float* __restrict__ a = P.in[k];
    char* part = partials + (size_t)k * nb * 16;
    float mn = INFINITY, mx = -INFINITY;
    long long nv = 0;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const float v = a[i];
        if (v == v) {
            mn = fminf(mn, v); mx = fmaxf(mx, v); ++nv;
            if (!(v >= 0.f && v <= 65535.f && v == floorf(v))) nv += 1ll << 32;      // (see nanminmax_kernel)
        }
    }
    for (int off = 32; off > 0; off >>= 1) {
        mn = fminf(mn, __shfl_down(mn, off));
        mx = fmaxf(mx, __shfl_down(mx, off));
        nv += __shfl_down(nv, off);
    }
    __shared__ float smn[4], smx[4];
    __shared__ long long snv[4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) { smn[wave] = mn; smx[wave] = mx; snv[wave] = nv; }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < 4; ++w) { mn = fminf(mn, smn[w]); mx = fmaxf(mx, smx[w]); nv += snv[w]; }
        ((float*)part)[blockIdx.x] = mn;
        ((float*)part)[nb + blockIdx.x] = mx;
        ((long long*)(part + (size_t)nb * 8))[blockIdx.x] = nv;
    }
}
__global__ void rescale_pair_kernel(PairPtrs P, long long n) {
    const int k = blockIdx.y;
    const float* __restrict__ src = P.in[k];
    float* __restrict__ dst = P.out[k];
    const float lo = P.lo[k], hi = P.hi[k];
    const int degenerate = P.degenerate[k];
    const float d = hi - lo;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        float v = src[i];
        if (v == v) {
            v = fminf(fmaxf(v, lo), hi);
            if (!degenerate) v = (v - lo) / d;
            v = v * 1.0f + 0.0f;
        }
        dst[i] = v;
    }
}

inline int grid_for(long long n) { return (int)std::min<long long>((n + 255) / 256, 256 * 8); }

// a few bytes of device memory into the host-resident mailbox
__global__ void peek_kernel(const unsigned int* __restrict__ src, unsigned int* __restrict__ dst, int nwords) {
    for (int i = threadIdx.x; i < nwords; i += blockDim.x) dst[i] = src[i];
}

// fftfreq(n, d)[x] as numpy defines it
inline double fftfreq(int n, double d, int x) {
    const int half = (n - 1) / 2 + 1;
    return (double)(x < half ? x : x - n) / ((double)n * d);
}

}  // namespace

int mvs_device_nanminmax(MvsContext* c, const float* d_in, long long n, float* mn, float* mx, long long* nvalid) {
    const int nb = grid_for(n);
    char* scratch = (char*)mvs_scratch(c, 3, (size_t)nb * 16);
    if (!scratch) return mvs_alloc_failed(c);
    float* pmin = (float*)scratch;
    float* pmax = pmin + nb;
    long long* pval = (long long*)(scratch + (size_t)nb * 8);
    hipLaunchKernelGGL(nanminmax_kernel, dim3(nb), dim3(256), 0, c->stream, d_in, n, pmin, pmax, pval);
    MVS_HIP_TRY(c, hipGetLastError());
    std::vector<char> h((size_t)nb * 16);
    MVS_HIP_TRY(c, hipMemcpyAsync(h.data(), scratch, h.size(), hipMemcpyDeviceToHost, c->stream));
    MVS_HIP_TRY(c, hipStreamSynchronize(c->stream));
    const float* hmin = (const float*)h.data();
    const float* hmax = hmin + nb;
    const long long* hval = (const long long*)(h.data() + (size_t)nb * 8);
    float a = INFINITY, b = -INFINITY;
    long long v = 0;
    for (int i = 0; i < nb; ++i) { a = std::min(a, hmin[i]); b = std::max(b, hmax[i]); v += hval[i]; }
    v &= 0xffffffffll;      // the upper half counts voxels that are not 16-bit integers (used by mvs_rescale_pair_device only)
    if (v == 0) { a = NAN; b = NAN; }
    *mn = a; *mx = b; *nvalid = v;
    return MVS_OK;
}

// Intensity normalisation of BOTH crops of a pair (device resident) with one host round trip: the two reductions are
// queued together, their partials come back in one copy, then the two rescale kernels follow.  Same kernels and
// arithmetic as two mvs_rescale_intensity calls.
int mvs_rescale_pair_device(MvsContext* c, const float* in0, const float* in1, long long n, float* out0, float* out1,
                            float mn[2], float mx[2], long long nvalid[2], long long n_not_u16[2], const MvsCropStats* parked) {
    const int nb = std::min(grid_for(n), 512);      // (each block writes its partials over the host link: fewer, longer blocks)
    void *mb_host = nullptr, *mb_dev = nullptr;       // the per-block partials land in host memory directly
    int rcm = mvs_mailbox(c, (size_t)nb * 32, &mb_host, &mb_dev);
    if (rcm) return rcm;
    char* scratch = (char*)mb_dev;
    PairPtrs PP;
    PP.in[0] = in0; PP.in[1] = in1; PP.out[0] = out0; PP.out[1] = out1;
    // (mvs_register_views: the crop kernels have left these partials already -- same layout, same block count)
    // ... and only in the very allocation they were written to: a mailbox that was reallocated in between has lost them
    if (!(parked && parked->done[0] && parked->done[1] && parked->nb == nb && parked->gen == c->mbox_gen && (void*)parked->base == mb_dev))
        hipLaunchKernelGGL(nanminmax_pair_kernel, dim3(nb, 2), dim3(256), 0, c->stream, PP, n, scratch, nb);
    MVS_HIP_TRY(c, hipGetLastError());
    MVS_HIP_TRY(c, hipStreamSynchronize(c->stream));
    for (int k = 0; k < 2; ++k) {
        const char* part = (const char*)mb_host + (size_t)k * nb * 16;
        const float* hmin = (const float*)part;
        const float* hmax = hmin + nb;
        const long long* hval = (const long long*)(part + (size_t)nb * 8);
        float a = INFINITY, b = -INFINITY;
        long long v = 0;
        for (int i = 0; i < nb; ++i) { a = std::min(a, hmin[i]); b = std::max(b, hmax[i]); v += hval[i]; }
        n_not_u16[k] = v >> 32;
        v &= 0xffffffffll;
        if (v == 0) { a = NAN; b = NAN; }
        mn[k] = a; mx[k] = b; nvalid[k] = v;
        PP.lo[k] = a; PP.hi[k] = b; PP.degenerate[k] = a == b ? 1 : 0;
    }
    MVS_DUP("rescale", hipLaunchKernelGGL(rescale_pair_kernel, dim3(nb, 2), dim3(256), 0, c->stream, PP, n));
    MVS_HIP_TRY(c, hipGetLastError());
    return MVS_OK;
}

// device-resident building blocks (used by mvs_phasecorr and mvs_score.hip)
int mvs_stage_float_volume(MvsContext* c, const float* src, int32_t mem, long long n, int slot, float** dptr) {
    if (mem == MVS_MEM_DEVICE) { *dptr = (float*)src; return MVS_OK; }
    float* d = (float*)mvs_scratch(c, slot, (size_t)n * 4);
    if (!d) return mvs_alloc_failed(c);
    MVS_HIP_TRY(c, hipMemcpyAsync(d, src, (size_t)n * 4, hipMemcpyHostToDevice, c->stream));
    *dptr = d;
    return MVS_OK;
}

extern "C" int mvs_rescale_intensity(int device, const float* in, int32_t mem, int64_t n, float* out, int32_t out_mem,
                                     float* min_out, float* max_out, int64_t* nvalid_out) {
    MvsContext* c;
    int rc = mvs_check_ready(device, &c);
    if (rc) return rc;
    std::lock_guard<std::recursive_mutex> lock(c->mu);
    if (!in || !out || n < 1) return mvs_fail(c, MVS_ERR_INVALID_ARG, "mvs_rescale_intensity: bad argument");
    MVS_HIP_TRY(c, hipSetDevice(mvs_hip_device(device)));
    float* din;
    rc = mvs_stage_float_volume(c, in, mem, n, 4, &din);
    if (rc) return rc;
    float mn, mx;
    long long nv;
    rc = mvs_device_nanminmax(c, din, n, &mn, &mx, &nv);
    if (rc) return rc;
    float* dout = out;
    if (out_mem == MVS_MEM_HOST) {
        dout = (float*)mvs_scratch(c, 5, (size_t)n * 4);
        if (!dout) return mvs_alloc_failed(c);
    }
    hipLaunchKernelGGL(rescale_kernel, dim3(grid_for(n)), dim3(256), 0, c->stream, din, dout, (long long)n, mn, mx, mn == mx ? 1 : 0);
    MVS_HIP_TRY(c, hipGetLastError());
    if (out_mem == MVS_MEM_HOST) MVS_HIP_TRY(c, hipMemcpyAsync(out, dout, (size_t)n * 4, hipMemcpyDeviceToHost, c->stream));
    MVS_HIP_TRY(c, hipStreamSynchronize(c->stream));
    if (min_out) *min_out = mn;
    if (max_out) *max_out = mx;
    if (nvalid_out) *nvalid_out = nv;
    return MVS_OK;
}

// In-place complex64 (interleaved re, im) n-D transform of one C-contiguous (z,y,x) array: numpy.fft.fftn / ifftn
// without the 1/N of the inverse.  The building block of mvs_phasecorr, exposed so that it can be checked on its own.
extern "C" int mvs_fft_c2c(int device, void* data, int32_t mem, int32_t ndim, const int64_t shape[3], int32_t inverse) {
    MvsContext* c;
    int rc = mvs_check_ready(device, &c);
    if (rc) return rc;
    std::lock_guard<std::recursive_mutex> lock(c->mu);
    if (!data || !shape || (ndim != 2 && ndim != 3)) return mvs_fail(c, MVS_ERR_INVALID_ARG, "mvs_fft_c2c: bad argument");
    for (int k = 0; k < 3; ++k)
        if (shape[k] < 1) return mvs_fail(c, MVS_ERR_INVALID_ARG, "mvs_fft_c2c: bad shape");
    if (ndim == 2 && shape[0] != 1) return mvs_fail(c, MVS_ERR_INVALID_ARG, "mvs_fft_c2c: 2D needs shape[0]==1");
    MVS_HIP_TRY(c, hipSetDevice(mvs_hip_device(device)));
    const size_t bytes = (size_t)shape[0] * shape[1] * shape[2] * sizeof(float2);
    float2* d = (float2*)data;
    if (mem == MVS_MEM_HOST) {
        d = (float2*)mvs_scratch(c, 6, bytes);
        if (!d) return mvs_alloc_failed(c);
        MVS_HIP_TRY(c, hipMemcpyAsync(d, data, bytes, hipMemcpyHostToDevice, c->stream));
    }
    rc = mvs_fft3_c2c(c, d, shape, inverse != 0);
    if (rc) return rc;
    if (mem == MVS_MEM_HOST) MVS_HIP_TRY(c, hipMemcpyAsync(data, d, bytes, hipMemcpyDeviceToHost, c->stream));
    MVS_HIP_TRY(c, hipStreamSynchronize(c->stream));
    return MVS_OK;
}

extern "C" int mvs_phasecorr(int device, const float* fixed, const float* moving, int32_t mem, int32_t ndim,
                             const int64_t shape[3], int32_t normalization, int32_t upsample_factor,
                             double shift_out[3], int64_t peak_index_out[3], float* peak_abs_out) {
    return mvs_phasecorr_multi(device, fixed, moving, mem, ndim, shape, &normalization, 1, upsample_factor, shift_out,
                               peak_index_out, peak_abs_out);
}

extern "C" int mvs_phasecorr_multi(int device, const float* fixed, const float* moving, int32_t mem, int32_t ndim,
                                   const int64_t shape[3], const int32_t* normalizations, int32_t n_norm,
                                   int32_t upsample_factor, double* shifts_out, int64_t* peak_indices_out,
                                   float* peak_abs_out_all) {
    MvsContext* c;
    int rc = mvs_check_ready(device, &c);
    if (rc) return rc;
    std::lock_guard<std::recursive_mutex> lock(c->mu);
    if (!fixed || !moving || !shape || !shifts_out || !normalizations || n_norm < 1)
        return mvs_fail(c, MVS_ERR_INVALID_ARG, "mvs_phasecorr: NULL argument");
    if (ndim != 2 && ndim != 3) return mvs_fail(c, MVS_ERR_INVALID_ARG, "mvs_phasecorr: ndim must be 2 or 3");
    if (upsample_factor < 1) return mvs_fail(c, MVS_ERR_INVALID_ARG, "mvs_phasecorr: upsample_factor < 1");
    for (int k = 0; k < 3; ++k)
        if (shape[k] < 1) return mvs_fail(c, MVS_ERR_INVALID_ARG, "mvs_phasecorr: bad shape");
    if (ndim == 2 && shape[0] != 1) return mvs_fail(c, MVS_ERR_INVALID_ARG, "mvs_phasecorr: 2D needs shape[0]==1");
    MVS_HIP_TRY(c, hipSetDevice(mvs_hip_device(device)));
    const long long n = (long long)shape[0] * shape[1] * shape[2];
    const int nz = (int)shape[0], ny = (int)shape[1], nx = (int)shape[2];

    float *da, *db;
    rc = mvs_stage_float_volume(c, fixed, mem, n, 4, &da);
    if (rc) return rc;
    rc = mvs_stage_float_volume(c, moving, mem, n, 5, &db);
    if (rc) return rc;
    // complex work volumes: Z (ONE forward transform of a + i b carries both spectra), P1 / P2 (cross power with and
    // without phase normalisation, kept for the upsampled DFT), CC (inverse transform; both correlations in one transform
    // when two normalisations are asked for)
    float2* Z = (float2*)mvs_scratch(c, 6, (size_t)n * 8 * 4);
    if (!Z) return mvs_alloc_failed(c);
    float2* P1 = Z + n;
    float2* P2 = P1 + n;
    float2* CC = P2 + n;

    MVS_HIP_TRY(c, hipEventRecord(c->ev_start, c->stream));
    const int gb = grid_for(n);
    // first / last transformed axis (the transform runs x, y, z; axes of length 1 are skipped)
    int first_axis = -1, last_axis = -1;
    for (int axis = 2; axis >= 0; --axis)
        if (shape[axis] > 1) { if (first_axis < 0) first_axis = axis; last_axis = axis; }
    // crops with one short axis and two power-of-two axes, two normalisations, small refinement: three passes in all
    // (mvs_fft_slab.hip) -- the forward transform happens there, together with the cross power and the inverse transform
    const int slab_axis = (ndim == 3 && n_norm == 2 && (normalizations[0] != 0) != (normalizations[1] != 0) && !c->materialize_shifts &&
                           (int)ceilf((float)upsample_factor * 1.5f) <= 4)
                              ? mvs_phasecorr_slab_axis(c, shape) : -1;
    if (slab_axis >= 0) {
        rc = MVS_OK;
    } else if (first_axis >= 0 && mvs_fft_reg_length((int)shape[first_axis]) && !c->reg_unfused) {
        MvsFftFuse ff;      // the first pass reads a and b themselves: no packed copy is written and read back
        ff.re_src = da;
        ff.im_src = db;
        rc = mvs_fft3_c2c(c, Z, shape, false, &ff);
    } else {
        hipLaunchKernelGGL(pack_pair_kernel, dim3(gb), dim3(256), 0, c->stream, da, db, Z, n);
        rc = mvs_fft3_c2c(c, Z, shape, false);
    }
    if (rc) return rc;
  // Two different normalisations: both correlations come out of ONE inverse transform (real and imaginary channel, see
  // xpower_packed_kernel) and one host round trip; otherwise one complex transform per normalisation.
  const bool packed = n_norm == 2 && (normalizations[0] != 0) != (normalizations[1] != 0) && !c->materialize_shifts;
  // mailbox (host memory written by the kernels): [peak partials gb * 32 | z0 | refinement results of every normalisation]
  const int up_U0 = (int)ceilf((float)upsample_factor * 1.5f);
  const size_t res_elems = upsample_factor > 1 ? (ndim == 3 ? (size_t)up_U0 * up_U0 * up_U0 : (size_t)up_U0 * up_U0) : 0;
  // blocks of the peak searches (one host-memory write each): 512 for the stand-alone kernel, or the workgroups of the inverse
  // transform's last pass when that pass does the search itself (16 or 32 lines each)
  int ga = std::min(gb, 512);
  {
      int la = -1;
      for (int axis = 2; axis >= 0; --axis) if (shape[axis] > 1) la = axis;
      if (la >= 0 && mvs_fft_reg_length((int)shape[la])) ga = std::max<long long>(ga, (n / shape[la] + 15) / 16);
      if (slab_axis >= 0) ga = std::max(ga, mvs_phasecorr_slab_peaks(shape, slab_axis));
  }
  const size_t mb_red = (size_t)ga * 32, mb_z0 = mb_red, mb_res = mb_red + 256, mb_res_stride = (res_elems * sizeof(float2) + 255) / 256 * 256;
  void *mb_host = nullptr, *mb_dev = nullptr;
  // ... | the kernel vectors of every refinement (pinned: their upload is a plain asynchronous copy, no staging through the runtime)]
  const size_t mb_k = mb_res + mb_res_stride * (size_t)n_norm;
  const size_t mb_kbytes = upsample_factor > 1 ? ((size_t)up_U0 * (size_t)(nz + ny + nx) * sizeof(float2) + 255) / 256 * 256 : 0;
  rc = mvs_mailbox(c, mb_k + mb_kbytes * (size_t)n_norm, &mb_host, &mb_dev);
  if (rc) return rc;
  char* red = (char*)mb_dev;
  const char* hred = (const char*)mb_host;
  float packed_scale[2] = {1.f, 1.f};
  int n_red_packed = ga;
  // ... and when its first pass (along x) runs on them and the refinement is small, the cross power is formed inside that pass
  // (MvsFftFuse::xp_src) and both refinements' first stage read it in one launch (updft_x2_kernel): the combined spectrum and the
  // phase-normalised cross power are never stored.
  const bool fuse_xp = slab_axis >= 0 || (packed && first_axis == 2 && mvs_fft_reg_length((int)nx) && up_U0 <= 4 && !c->reg_unfused);
  // ... and in 3D (refinement of 3 samples per axis: upsample factor 2) the first TWO stages of both refinements are one pass
  // over it (updft_yx2_kernel): a lane sums its columns over the rows of a chunk before anything is reduced across lanes
  const bool fuse_yx = fuse_xp && ndim == 3 && up_U0 == 3 && nx <= 256 && ny >= 4;
  const int yx_chunks = fuse_yx ? std::max(1, std::min(kYxChunksMax, ((int)ny + kYxRows - 1) / kYxRows)) : 1;
  if (packed && slab_axis >= 0) {
    float* pv[2] = {(float*)red, (float*)(red + (size_t)ga * 16)};
    long long* pi[2] = {(long long*)(red + (size_t)ga * 8), (long long*)(red + (size_t)ga * 24)};
    rc = mvs_phasecorr_slab(c, da, db, Z, CC, P2, P1 /* scratch: the slabs' DC terms */, shape, slab_axis, normalizations[0] ? 1 : 0,
                            normalizations[1] ? 1 : 0, pv, pi, (float2*)((char*)mb_dev + mb_z0));
    if (rc) return rc;
    n_red_packed = mvs_phasecorr_slab_peaks(shape, slab_axis);
    c->reg_slab_pairs += 1;
    MVS_HIP_TRY(c, hipStreamSynchronize(c->stream));
    const float2 z0 = *(const float2*)((const char*)mb_host + mb_z0);
    const float dc = std::fabs(z0.x * z0.y);
    const float s_plain = (dc > 0.f && dc < INFINITY) ? std::ldexp(1.f, -std::ilogb(dc)) : 1.f;
    const float s_phase = std::ldexp(1.f, -std::ilogb((float)n));
    for (int ch = 0; ch < 2; ++ch) packed_scale[ch] = normalizations[ch] ? s_phase : s_plain;
  } else if (packed) {
    if (!fuse_xp)
        hipLaunchKernelGGL(xpower_packed_kernel, dim3(gb), dim3(256), 0, c->stream, Z, P1, P2, CC, nz, ny, nx, normalizations[0] ? 1 : 0,
                           normalizations[1] ? 1 : 0);
    // the correlation volume is only ever searched for its two peaks: when the last pass of the inverse transform runs on the
    // register kernels it reduces its output to per-workgroup maxima instead of storing it (no 27 MB written and read back)
    MvsFftFuse fp;
    if (last_axis >= 0 && mvs_fft_reg_length((int)shape[last_axis]) && !c->reg_unfused) {
        fp.peak_val[0] = (float*)red; fp.peak_idx[0] = (long long*)(red + (size_t)ga * 8);
        fp.peak_val[1] = (float*)(red + (size_t)ga * 16); fp.peak_idx[1] = (long long*)(red + (size_t)ga * 24);
        fp.peak_cap = ga;
    }
    if (fuse_xp) { fp.xp_src = Z; fp.xp_p2 = P2; fp.xp_sel_a = normalizations[0] ? 1 : 0; fp.xp_sel_b = normalizations[1] ? 1 : 0; }
    rc = mvs_fft3_c2c(c, CC, shape, true, &fp);
    if (rc) return rc;
    if (fuse_xp && !fp.xp_used) return mvs_fail(c, MVS_ERR_UNSUPPORTED, "phase correlation: the fused cross-power pass was not taken");
    if (fp.n_peak > 0) {
        n_red_packed = fp.n_peak;
        hipLaunchKernelGGL(peek_kernel, dim3(1), dim3(64), 0, c->stream, (const unsigned int*)Z, (unsigned int*)((char*)mb_dev + mb_z0), 2);
    } else {
        hipLaunchKernelGGL(argmax_abs2_kernel, dim3(ga), dim3(256), 0, c->stream, CC, n, (float*)red, (long long*)(red + (size_t)ga * 8),
                           (float*)(red + (size_t)ga * 16), (long long*)(red + (size_t)ga * 24), Z, (float2*)((char*)mb_dev + mb_z0));
    }
    MVS_HIP_TRY(c, hipGetLastError());
    MVS_HIP_TRY(c, hipStreamSynchronize(c->stream));
    const float2 z0 = *(const float2*)((const char*)mb_host + mb_z0);
    // the channel scales of xpower_packed_kernel (exact powers of two), to report the peak height unscaled
    const float dc = std::fabs(z0.x * z0.y);
    const float s_plain = (dc > 0.f && dc < INFINITY) ? std::ldexp(1.f, -std::ilogb(dc)) : 1.f;
    const float s_phase = std::ldexp(1.f, -std::ilogb((float)n));
    for (int ch = 0; ch < 2; ++ch) packed_scale[ch] = normalizations[ch] ? s_phase : s_plain;
  }
  // Per normalisation: integer peak -> upsampled DFT around it.  The refinements of all normalisations are queued before
  // the host waits once for their results (phase 2 below).
  struct NormState {
      float shift[3]; size_t nout = 0;
      const float2* P = nullptr; float2 *dk = nullptr, *o1 = nullptr, *o2 = nullptr, *o3 = nullptr; size_t koff[3] = {0, 0, 0};
  };
  std::vector<NormState> state((size_t)n_norm);
  const int up_U = (int)ceilf((float)upsample_factor * 1.5f);
  const size_t up_kbytes = ((size_t)up_U * (size_t)(nz + ny + nx) * sizeof(float2) + 255) / 256 * 256;
  const size_t up_s1 = (size_t)nz * ny * up_U, up_s2 = (size_t)nz * up_U * up_U, up_s3 = (size_t)up_U * up_U * up_U;
  const size_t up_bytes = up_kbytes + (up_s1 + up_s2 + up_s3) * sizeof(float2) + 1024;
  char* up_base = nullptr;
  if (upsample_factor > 1) {
      up_base = (char*)mvs_scratch(c, 7, up_bytes * (size_t)n_norm);
      if (!up_base) return mvs_alloc_failed(c);
  }
  for (int inorm = 0; inorm < n_norm; ++inorm) {
    const int normalization = normalizations[inorm];
    int64_t* peak_index_out = peak_indices_out ? peak_indices_out + 3 * inorm : nullptr;
    float* peak_abs_out = peak_abs_out_all ? peak_abs_out_all + inorm : nullptr;
    const float2* P = normalization ? P1 : P2;
    const char* hpart = hred + (packed ? (size_t)inorm * ga * 16 : 0);
    const int n_part = packed ? n_red_packed : ga;
    if (!packed) {
        hipLaunchKernelGGL(xpower_packed_kernel, dim3(gb), dim3(256), 0, c->stream, Z, P1, P2, CC, nz, ny, nx, normalization ? 1 : 0, -1);
        rc = mvs_fft3_c2c(c, CC, shape, true);   // cc (unnormalised inverse: argmax is scale invariant)
        if (rc) return rc;
        hipLaunchKernelGGL(argmax_abs_kernel<0>, dim3(ga), dim3(256), 0, c->stream, CC, n, (float*)red, (long long*)(red + (size_t)ga * 8));
        MVS_HIP_TRY(c, hipGetLastError());
        MVS_HIP_TRY(c, hipStreamSynchronize(c->stream));
    }
    float best = -1.f;
    long long bi = 0;
    {
        const float* hv = (const float*)hpart;
        const long long* hi = (const long long*)(hpart + (size_t)ga * 8);
        for (int i = 0; i < n_part; ++i)
            if (hv[i] > best || (hv[i] == best && hi[i] < bi)) { best = hv[i]; bi = hi[i]; }
    }
    if (packed) best /= packed_scale[inorm];
    const long long pz = bi / ((long long)ny * nx), py = (bi / nx) % ny, px = bi % nx;
    const long long peak[3] = {pz, py, px};
    if (peak_index_out) { peak_index_out[0] = pz; peak_index_out[1] = py; peak_index_out[2] = px; }
    if (peak_abs_out) *peak_abs_out = best / (float)n;   // ifftn's 1/N

    // ---- signed shift, float32 arithmetic like skimage ----
    float (&shift)[3] = state[inorm].shift;
    for (int k = 0; k < 3; ++k) {
        shift[k] = (float)peak[k];
        const float mid = truncf((float)shape[k] / 2.f);   // np.fix(axis_size / 2)
        if (shift[k] > mid) shift[k] -= (float)shape[k];
    }
    if (upsample_factor > 1) {
        const float uf = (float)upsample_factor;
        const int U = (int)ceilf(uf * 1.5f);
        const float dftshift = truncf((float)U / 2.f);
        float offs[3];
        for (int k = 0; k < 3; ++k) {
            shift[k] = nearbyintf(shift[k] * uf) / uf;            // np.round: half to even
            offs[k] = dftshift - shift[k] * uf;
        }
        // kernels exp(-2 pi i (a - off) * fftfreq(n, uf)[x]) computed in double, cast to complex64
        const int k0 = (ndim == 3) ? 0 : 1;
        float2* hk = (float2*)((char*)mb_host + mb_k + mb_kbytes * (size_t)inorm);   // stays untouched until the wait below
        size_t koff[3] = {0, 0, 0}, nk = 0;
        for (int k = k0; k < 3; ++k) {
            koff[k] = nk;
            const int nn = (int)shape[k];
            for (int a = 0; a < U; ++a)
                for (int x = 0; x < nn; ++x) {
                    const double ph = -2.0 * M_PI * ((double)a - (double)offs[k]) * fftfreq(nn, (double)uf, x);
                    hk[nk++] = make_float2((float)cos(ph), (float)sin(ph));
                }
        }
        const size_t kbytes = nk * sizeof(float2);
        const long long nrows = (long long)nz * ny;
        const size_t s1 = (size_t)nrows * U, s2 = (size_t)nz * U * U, s3 = (size_t)U * U * U;
        // device layout: [kernel vectors of every normalisation | stage outputs of every normalisation]
        float2* dk = (float2*)(up_base + (size_t)inorm * up_kbytes);
        float2* o1 = (float2*)(up_base + (size_t)n_norm * up_kbytes + (size_t)inorm * (up_bytes - up_kbytes));
        float2* o2 = o1 + s1;
        float2* o3 = o2 + s2;
        float2* mres = (float2*)((char*)mb_dev + mb_res + mb_res_stride * (size_t)inorm);     // the last stage writes host memory
        if (ndim == 3) o3 = mres; else o2 = mres;
        // one upload for all normalisations when their launches follow the last one (same strides in the mailbox and on the device)
        // (copied by a kernel out of the mapped mailbox: a DMA upload would queue behind any tile upload in flight)
        const char* hk_dev = (const char*)mb_dev + mb_k + mb_kbytes * (size_t)inorm;
        int rcu = MVS_OK;
        if (!fuse_xp) rcu = mvs_upload_from_mapped(c, dk, hk_dev, (kbytes + 15) / 16 * 16);
        else if (inorm == n_norm - 1)
            rcu = mvs_upload_from_mapped(c, up_base, (const char*)mb_dev + mb_k, ((size_t)inorm * up_kbytes + kbytes + 15) / 16 * 16);
        if (rcu) return rcu;
        NormState& st = state[inorm];
        st.P = P; st.dk = dk; st.o1 = o1; st.o2 = o2; st.o3 = o3;
        for (int k = 0; k < 3; ++k) st.koff[k] = koff[k];
        st.nout = ndim == 3 ? s3 : s2;
        if (!fuse_xp) {
            // stage 1: x  -> (z, y, ux)
            hipLaunchKernelGGL(updft_x_kernel, dim3((unsigned)((nrows + 3) / 4)), dim3(256), 0, c->stream, P, dk + koff[2], o1, nrows, nx, U);
        } else if (fuse_yx) {
            if (inorm == n_norm - 1) {
                // stages 1 and 2 of both normalisations: partial sums per (z, y chunk), folded by stage 3; they live in o1 (chunks * U <= ny)
                UpdftYX q;
                for (int j = 0; j < 2; ++j) {
                    q.Kx[j] = state[j].dk + state[j].koff[2]; q.Ky[j] = state[j].dk + state[j].koff[1]; q.out[j] = state[j].o1;
                    q.phase[j] = normalizations[j] ? 1 : 0;
                }
                const int rows_per_chunk = ((int)ny + yx_chunks - 1) / yx_chunks;
                MVS_DUP("updft", hipLaunchKernelGGL(updft_yx2_kernel<3>, dim3((unsigned)(nz * yx_chunks)), dim3(256), 0, c->stream, P2, q, (int)nz, (int)ny, (int)nx,
                                   rows_per_chunk, yx_chunks));
                for (int j = 0; j < n_norm; ++j)
                    MVS_DUP("updft_mid", hipLaunchKernelGGL(updft_mid_kernel, dim3((unsigned)std::min<long long>(((long long)s3 + 3) / 4, 4096)), dim3(256), 0, c->stream,
                                       state[j].o1, state[j].dk + state[j].koff[0], state[j].o3, 1, (int)nz * yx_chunks, U * U, U, yx_chunks));
            }
            MVS_HIP_TRY(c, hipGetLastError());
            continue;
        } else if (inorm == n_norm - 1) {
            // stage 1 of both normalisations from the plain cross power alone
            UpdftX2 q;
            for (int j = 0; j < 2; ++j) { q.K[j] = state[j].dk + state[j].koff[2]; q.out[j] = state[j].o1; q.phase[j] = normalizations[j] ? 1 : 0; }
            hipLaunchKernelGGL(updft_x2_kernel<4>, dim3((unsigned)((nrows + 3) / 4)), dim3(256), 0, c->stream, P2, q, nrows, (int)nx, U);
        }
        for (int j = (fuse_xp ? (inorm == n_norm - 1 ? 0 : n_norm) : inorm); j <= inorm && j < n_norm; ++j) {
            const NormState& sj = state[j];
            // stage 2: y  -> (z, uy, ux)
            hipLaunchKernelGGL(updft_mid_kernel, dim3((unsigned)std::min<long long>(((long long)s2 + 3) / 4, 4096)), dim3(256), 0, c->stream, sj.o1, sj.dk + sj.koff[1], sj.o2, nz, ny, U, U);
            // stage 3: z -> (uz, uy, ux)
            if (ndim == 3)
                hipLaunchKernelGGL(updft_mid_kernel, dim3((unsigned)std::min<long long>(((long long)s3 + 3) / 4, 4096)), dim3(256), 0, c->stream, sj.o2, sj.dk + sj.koff[0], sj.o3, 1, nz, U * U, U);
        }
        MVS_HIP_TRY(c, hipGetLastError());
    }
  }
  MVS_HIP_TRY(c, hipEventRecord(c->ev_stop, c->stream));
  c->timing_valid = true;
  if (upsample_factor > 1) MVS_HIP_TRY(c, hipStreamSynchronize(c->stream));
  // ---- phase 2: sub-pixel maximum of every refinement ----
  for (int inorm = 0; inorm < n_norm; ++inorm) {
    float (&shift)[3] = state[inorm].shift;
    if (upsample_factor > 1) {
        const float uf = (float)upsample_factor;
        const int U = up_U;
        const float dftshift = truncf((float)U / 2.f);
        const int k0 = (ndim == 3) ? 0 : 1;
        const float2* hres = (const float2*)((const char*)mb_host + mb_res + mb_res_stride * (size_t)inorm);
        // argmax |.| (conj does not change the modulus), lowest flat index
        float bu = -1.f;
        size_t iu = 0;
        for (size_t i = 0; i < state[inorm].nout; ++i) {
            const float a = hypotf(hres[i].x, hres[i].y);
            if (a > bu) { bu = a; iu = i; }
        }
        int m[3] = {0, 0, 0};
        m[2] = (int)(iu % U);
        m[1] = (int)((iu / U) % U);
        if (ndim == 3) m[0] = (int)(iu / ((size_t)U * U));
        for (int k = k0; k < 3; ++k) shift[k] += ((float)m[k] - dftshift) / uf;
    }
    double* shift_out = shifts_out + 3 * inorm;
    for (int k = 0; k < 3; ++k) {
        if (shape[k] == 1) shift[k] = 0.f;
        shift_out[k] = (double)shift[k];
    }
  }
    return MVS_OK;
}

// ---- block-mean binning == sim.coarsen(bins, boundary="trim").mean().astype(dtype) (registration.py:1732-1741) ----
namespace {
template <typename T>
__global__ void bin_mean_kernel(const T* __restrict__ in, long long sz, long long sy, T* __restrict__ out, int oz, int oy, int ox,
                                int bz, int by, int bx) {
    const long long n = (long long)oz * oy * ox;
    const double inv = 1.0 / ((double)bz * by * bx);
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const int x = (int)(i % ox);
        const long long t = i / ox;
        const int y = (int)(t % oy), z = (int)(t / oy);
        double acc = 0.0;
        for (int dz = 0; dz < bz; ++dz)
            for (int dy = 0; dy < by; ++dy) {
                const T* row = in + (long long)(z * bz + dz) * sz + (long long)(y * by + dy) * sy + (long long)x * bx;
                for (int dx = 0; dx < bx; ++dx) acc += (double)row[dx];
            }
        const double m = acc * inv;
        out[i] = (T)m;   // astype: truncation for integer dtypes
    }
}
// uint16, bin 2 along x, rows 16-byte aligned: a thread produces 4 consecutive outputs of a row from one 16-byte load per
// input row (the generic kernel reads element by element: 1.3 TB/s on a 512^3 tile).  Integer sums are exact in double,
// so the order of the additions does not matter and the result equals the generic kernel's.
__global__ __launch_bounds__(256) void bin_mean_u16x2_kernel(const unsigned short* __restrict__ in, long long sz, long long sy,
                                                             unsigned short* __restrict__ out, int oz, int oy, int ox, int bz, int by) {
    const int gx = ox >> 2;                                   // groups of 4 outputs per row (ox % 4 == 0)
    const long long ngroups = (long long)oz * oy * gx;
    const double inv = 1.0 / ((double)bz * by * 2);
    typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
    for (long long g = (long long)blockIdx.x * blockDim.x + threadIdx.x; g < ngroups; g += (long long)gridDim.x * blockDim.x) {
        const int xg = (int)(g % gx);
        const long long t = g / gx;
        const int y = (int)(t % oy), z = (int)(t / oy);
        unsigned int acc[4] = {0u, 0u, 0u, 0u};               // at most 2 * by * bz * 65535: fits for by * bz <= 32767
        for (int dz = 0; dz < bz; ++dz)
            for (int dy = 0; dy < by; ++dy) {
                const u32x4 v = *reinterpret_cast<const u32x4*>(in + (long long)(z * bz + dz) * sz + (long long)(y * by + dy) * sy + (long long)xg * 8);
#pragma unroll
                for (int k = 0; k < 4; ++k) acc[k] += (v[k] & 0xffffu) + (v[k] >> 16);
            }
        unsigned short r[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) r[k] = (unsigned short)((double)acc[k] * inv);   // astype: truncation
        *reinterpret_cast<uint2*>(out + ((long long)z * oy + y) * ox + (long long)xg * 4) =
            make_uint2((unsigned int)r[0] | ((unsigned int)r[1] << 16), (unsigned int)r[2] | ((unsigned int)r[3] << 16));
    }
}
// the same for up to kBinBatch views of one shape in one launch: blockIdx.y = view (pointer table in the kernel arguments)
constexpr int kBinBatch = 32;
struct BinBatch {
    const unsigned short* in[kBinBatch];
    unsigned short* out[kBinBatch];
};
__global__ __launch_bounds__(256) void bin_mean_u16x2_batch_kernel(BinBatch B, long long sz, long long sy, int oz, int oy, int ox, int bz, int by) {
    const unsigned short* __restrict__ in = B.in[blockIdx.y];
    unsigned short* __restrict__ out = B.out[blockIdx.y];
    const int gx = ox >> 2;
    const long long ngroups = (long long)oz * oy * gx;
    const double inv = 1.0 / ((double)bz * by * 2);
    typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
    for (long long g = (long long)blockIdx.x * blockDim.x + threadIdx.x; g < ngroups; g += (long long)gridDim.x * blockDim.x) {
        const int xg = (int)(g % gx);
        const long long t = g / gx;
        const int y = (int)(t % oy), z = (int)(t / oy);
        unsigned int acc[4] = {0u, 0u, 0u, 0u};
        for (int dz = 0; dz < bz; ++dz)
            for (int dy = 0; dy < by; ++dy) {
                const u32x4 v = *reinterpret_cast<const u32x4*>(in + (long long)(z * bz + dz) * sz + (long long)(y * by + dy) * sy + (long long)xg * 8);
#pragma unroll
                for (int k = 0; k < 4; ++k) acc[k] += (v[k] & 0xffffu) + (v[k] >> 16);
            }
        unsigned short r[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) r[k] = (unsigned short)((double)acc[k] * inv);   // astype: truncation
        *reinterpret_cast<uint2*>(out + ((long long)z * oy + y) * ox + (long long)xg * 4) =
            make_uint2((unsigned int)r[0] | ((unsigned int)r[1] << 16), (unsigned int)r[2] | ((unsigned int)r[3] << 16));
    }
}
}  // namespace

static int bin_mean_impl(int device, const void* in, int32_t dtype, int32_t mem, const int64_t shape[3], const int64_t stride[3],
                         const int64_t bin[3], void* out, int32_t out_mem, bool wait);

// mvs_bin_mean_async for n views of one shape, stride and dtype in one call (and, for 16-bit tiles binned by 2 along x, one launch per
// kBinBatch views): registration.register bins all tiles of a mosaic before its pairs start.
extern "C" int mvs_bin_mean_batch_async(int device, int32_t n_views, const void* const* in, int32_t dtype, const int64_t shape[3],
                                        const int64_t stride[3], const int64_t bin[3], void* const* out) {
    MvsContext* c;
    int rc = mvs_check_ready(device, &c);
    if (rc) return rc;
    std::lock_guard<std::recursive_mutex> lock(c->mu);
    if (n_views < 0 || (n_views > 0 && (!in || !out)) || !shape || !stride || !bin) return mvs_fail(c, MVS_ERR_INVALID_ARG, "mvs_bin_mean_batch_async: NULL argument");
    bool vec = dtype == MVS_U16 && stride[2] == 1 && bin[2] == 2 && bin[0] >= 1 && bin[1] >= 1 && shape[0] >= bin[0] && shape[1] >= bin[1] &&
               shape[2] >= 2 && (shape[2] / 2) % 4 == 0 && stride[1] % 8 == 0 && stride[0] % 8 == 0 && bin[0] * bin[1] <= 16384;
    for (int v = 0; v < n_views && vec; ++v) vec = in[v] && out[v] && ((uintptr_t)in[v] % 16) == 0 && ((uintptr_t)out[v] % 8) == 0;
    if (!vec) {
        for (int v = 0; v < n_views; ++v) {
            rc = bin_mean_impl(device, in[v], dtype, MVS_MEM_DEVICE, shape, stride, bin, out[v], MVS_MEM_DEVICE, false);
            if (rc) return rc;
        }
        return MVS_OK;
    }
    MVS_HIP_TRY(c, hipSetDevice(mvs_hip_device(device)));
    const int oz = (int)(shape[0] / bin[0]), oy = (int)(shape[1] / bin[1]), ox = (int)(shape[2] / 2);
    const long long n = (long long)oz * oy * ox;
    for (int v0 = 0; v0 < n_views; v0 += kBinBatch) {
        const int nb = std::min(kBinBatch, n_views - v0);
        BinBatch B;
        for (int k = 0; k < kBinBatch; ++k) {
            B.in[k] = (const unsigned short*)in[v0 + std::min(k, nb - 1)];
            B.out[k] = (unsigned short*)out[v0 + std::min(k, nb - 1)];
        }
        hipLaunchKernelGGL(bin_mean_u16x2_batch_kernel, dim3(grid_for(n / 4), nb), dim3(256), 0, c->stream, B, (long long)stride[0], (long long)stride[1],
                           oz, oy, ox, (int)bin[0], (int)bin[1]);
    }
    MVS_HIP_TRY(c, hipGetLastError());
    return MVS_OK;
}

extern "C" int mvs_bin_mean(int device, const void* in, int32_t dtype, int32_t mem, const int64_t shape[3],
                            const int64_t stride[3], const int64_t bin[3], void* out, int32_t out_mem) {
    return bin_mean_impl(device, in, dtype, mem, shape, stride, bin, out, out_mem, true);
}

// The same launch without the final wait (device memory on both sides only): the caller orders later work with
// mvs_synchronize(device) -- used to bin all tiles of a mosaic while the host builds its overlap graph.
extern "C" int mvs_bin_mean_async(int device, const void* in, int32_t dtype, const int64_t shape[3], const int64_t stride[3],
                                  const int64_t bin[3], void* out) {
    return bin_mean_impl(device, in, dtype, MVS_MEM_DEVICE, shape, stride, bin, out, MVS_MEM_DEVICE, false);
}

static int bin_mean_impl(int device, const void* in, int32_t dtype, int32_t mem, const int64_t shape[3], const int64_t stride[3],
                         const int64_t bin[3], void* out, int32_t out_mem, bool wait) {
    MvsContext* c;
    int rc = mvs_check_ready(device, &c);
    if (rc) return rc;
    std::lock_guard<std::recursive_mutex> lock(c->mu);
    if (!in || !out || !shape || !stride || !bin) return mvs_fail(c, MVS_ERR_INVALID_ARG, "mvs_bin_mean: NULL argument");
    const size_t es = mvs_dtype_size(dtype);
    if (!es) return mvs_fail(c, MVS_ERR_INVALID_ARG, "mvs_bin_mean: bad dtype");
    if (stride[2] != 1) return mvs_fail(c, MVS_ERR_UNSUPPORTED, "mvs_bin_mean: x stride must be 1");
    int o[3];
    for (int k = 0; k < 3; ++k) {
        if (bin[k] < 1 || shape[k] < bin[k]) return mvs_fail(c, MVS_ERR_INVALID_ARG, "mvs_bin_mean: bad bin/shape on axis %d", k);
        o[k] = (int)(shape[k] / bin[k]);
    }
    MVS_HIP_TRY(c, hipSetDevice(mvs_hip_device(device)));
    const void* din = in;
    long long sz = stride[0], sy = stride[1];
    if (mem == MVS_MEM_HOST) {
        if (stride[1] != shape[2] || stride[0] != shape[1] * shape[2])
            return mvs_fail(c, MVS_ERR_UNSUPPORTED, "mvs_bin_mean: host input must be C-contiguous");
        const size_t nb = (size_t)shape[0] * shape[1] * shape[2] * es;
        void* s = mvs_scratch(c, 4, nb);
        if (!s) return mvs_alloc_failed(c);
        MVS_HIP_TRY(c, hipMemcpyAsync(s, in, nb, hipMemcpyHostToDevice, c->stream));
        din = s;
    }
    const long long n = (long long)o[0] * o[1] * o[2];
    void* dout = out;
    if (out_mem == MVS_MEM_HOST) {
        dout = mvs_scratch(c, 5, (size_t)n * es);
        if (!dout) return mvs_alloc_failed(c);
    }
    const int gb = grid_for(n);
#define MVS_BIN(T) hipLaunchKernelGGL(bin_mean_kernel<T>, dim3(gb), dim3(256), 0, c->stream, (const T*)din, sz, sy, (T*)dout, \
                                      o[0], o[1], o[2], (int)bin[0], (int)bin[1], (int)bin[2])
    const bool vec_u16 = dtype == MVS_U16 && bin[2] == 2 && o[2] % 4 == 0 && sy % 8 == 0 && sz % 8 == 0 && ((uintptr_t)din % 16) == 0 &&
                         ((uintptr_t)dout % 8) == 0 && bin[0] * bin[1] <= 16384;
    if (vec_u16) {
        hipLaunchKernelGGL(bin_mean_u16x2_kernel, dim3(grid_for(n / 4)), dim3(256), 0, c->stream, (const unsigned short*)din, sz, sy,
                           (unsigned short*)dout, o[0], o[1], o[2], (int)bin[0], (int)bin[1]);
    } else
    switch (dtype) {
        case MVS_U8: MVS_BIN(unsigned char); break;
        case MVS_U16: MVS_BIN(unsigned short); break;
        default: MVS_BIN(float); break;
    }
#undef MVS_BIN
    MVS_HIP_TRY(c, hipGetLastError());
    if (out_mem == MVS_MEM_HOST) MVS_HIP_TRY(c, hipMemcpyAsync(out, dout, (size_t)n * es, hipMemcpyDeviceToHost, c->stream));
    if (wait) MVS_HIP_TRY(c, hipStreamSynchronize(c->stream));
    return MVS_OK;
}
