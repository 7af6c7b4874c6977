// mvs_gauss.hip -- content-based fusion weights (Preibisch) on the GPU (gfx950).
//
// mvs_fuse_chunk with weights == MVS_WEIGHTS_CONTENT_BASED reproduces, per output chunk (incl. halo),
// the reference's fuse_np with weights_func=weights.content_based (src/multiview_stitcher/):
//   field_ims_t[v]  = affine_transform(view v, cval=NaN)                         fusion/_core.py:1621-1633
//   field_ws_t[v]   = blending weights * ~isnan, normalised over views           _core.py:1636-1649
//   content_based:  I[bw < 1e-7] = NaN;  F_v = NG_s2((I - NG_s1(I))^2);  F = normalise(F)   weights.py:22-74
//     NG_s(U) = gaussian(U with NaN->0) / gaussian(valid mask), NaN kept            weights.py:293-322
//     gaussian = scipy.ndimage.gaussian_filter(sigma, mode="reflect", truncate=4): separable correlate1d,
//     float64 kernel and accumulation, float32 output after every axis
//   weighted_average_fusion: A = bw * F, normalise, sum_v I_v * A_v               _core.py:85-94
//   trim halo, nan_to_num, astype(input dtype)                                     _core.py:1687-1713
// The halo (2*sigma_2, weights.py:22) and the chunk grid are the caller's (fusion.fuse mirrors the
// reference's), because the reflect boundary of the Gaussians makes results depend on the chunking.
//
// Round 3: every view is processed on its BOX -- the part of the halo chunk it can reach (exact in-bounds interval for
// translations, bounding box of the mapped slab otherwise) -- instead of on the whole chunk.  Outside its box a view is NaN,
// i.e. value 0 and mask 0 in both terms of the NaN-aware Gaussian, so a line filter over the box with ZEROS beyond the ends
// that lie inside the chunk and the chunk's own REFLECTION beyond the ends that coincide with the chunk's border yields,
// at every voxel of the box, the sums the reference forms over the whole chunk line (same taps, same order: the taps that
// are skipped are exact zeros).  A chunk of a tile grid sees one view nearly whole and up to seven by a corner or a face:
// 1.3 chunk volumes of filter work instead of 8 on the 2x2x2 probe.
#include "mvs_fuse_dev.h"
#include "mvs_fuse_tr.h"

#include <algorithm>
#include <cmath>
#include <cstring>
#include <type_traits>
#include <vector>

namespace {

inline int grid_for(long long n) { return (int)std::min<long long>((n + 255) / 256, 256 * 16); }

struct Shape3 { int nz, ny, nx; };

struct CbBox { int lo[3], n[3]; long long off; };      // box of a view inside the chunk; off: its first float in the I / BW / F pools
static_assert(sizeof(CbBox) == 32, "CbBox layout");

__device__ __forceinline__ long long box_index(const CbBox& B, int z, int y, int x) {
    const int bz = z - B.lo[0], by = y - B.lo[1], bx = x - B.lo[2];
    if ((unsigned)bz >= (unsigned)B.n[0] || (unsigned)by >= (unsigned)B.n[1] || (unsigned)bx >= (unsigned)B.n[2]) return -1;
    return B.off + ((long long)bz * B.n[1] + by) * B.n[2] + bx;
}

// bw[v] *= ~isnan(I[v]) ; then normalise over views: wsum = sum_v bw (float32, view order), 0 -> 1.  One thread per chunk
// voxel; a view takes part where its box holds the voxel (elsewhere it is NaN with weight 0: adds an exact 0).
__global__ void mask_normalize_kernel(float* __restrict__ bw, const float* __restrict__ im, const CbBox* __restrict__ boxes, int nviews, Shape3 S) {
    const long long n = (long long)S.nz * S.ny * S.nx;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const int x = (int)(i % S.nx);
        const long long t = i / S.nx;
        const int y = (int)(t % S.ny), z = (int)(t / S.ny);
        float wsum = 0.f;
        for (int v = 0; v < nviews; ++v) {
            const long long k = box_index(boxes[v], z, y, x);
            if (k < 0) continue;
            float w = bw[k];
            const float xv = im[k];
            if (xv != xv) w = 0.f;   // w * False
            bw[k] = w;
            wsum += w;               // np.nansum over axis 0 adds view by view in float32
        }
        if (wsum == 0.f) wsum = 1.f;
        for (int v = 0; v < nviews; ++v) {
            const long long k = box_index(boxes[v], z, y, x);
            if (k >= 0) bw[k] /= wsum;
        }
    }
}

// The same for chunks seen by at most 8 views (every tile grid): the view loop is unrolled, a view's pool index is worked out once
// per voxel in 32-bit arithmetic and kept in a register, and the masked weight is written once, already normalised (the general
// kernel stores it, re-reads it and divides in place: 20 instead of 12 bytes per view and voxel).
struct CbBox32 { int lo[3], n[3], off; };
struct CbBoxes8 { CbBox32 b[8]; };
__device__ __forceinline__ int box_index32(const CbBox32& B, int z, int y, int x) {
    const int bz = z - B.lo[0], by = y - B.lo[1], bx = x - B.lo[2];
    if ((unsigned)bz >= (unsigned)B.n[0] || (unsigned)by >= (unsigned)B.n[1] || (unsigned)bx >= (unsigned)B.n[2]) return -1;
    return B.off + (bz * B.n[1] + by) * B.n[2] + bx;
}
// nan_masked (fast path, mvs_gauss_fast.inc): the view itself becomes NaN where its normalised weight is < 1e-7 (weights.py:54-55), so
// that "valid" is "finite" for every later pass; the final sum skips such a view there either way (its F is NaN).
__global__ __launch_bounds__(256) void mask_normalize8_kernel(float* __restrict__ bw, float* __restrict__ im, CbBoxes8 BX, int nviews, Shape3 S, int nan_masked) {
    const long long n = (long long)S.nz * S.ny * S.nx;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const int x = (int)(i % S.nx);
        const long long t = i / S.nx;
        const int y = (int)(t % S.ny), z = (int)(t / S.ny);
        int kk[8];
        float w[8];
        float wsum = 0.f;
#pragma unroll
        for (int v = 0; v < 8; ++v) {
            kk[v] = v < nviews ? box_index32(BX.b[v], z, y, x) : -1;
            w[v] = 0.f;
            if (kk[v] >= 0) {
                w[v] = bw[kk[v]];
                const float xv = im[kk[v]];
                if (xv != xv) w[v] = 0.f;      // w * False
                wsum += w[v];                  // np.nansum over axis 0 adds view by view in float32
            }
        }
        if (wsum == 0.f) wsum = 1.f;
#pragma unroll
        for (int v = 0; v < 8; ++v)
            if (kk[v] >= 0) {
                const float wn = w[v] / wsum;
                bw[kk[v]] = wn;
                if (nan_masked && wn < 1e-7f) im[kk[v]] = NAN;
            }
    }
}

// A = I with NaN where the (normalised) blending weight < 1e-7 (weights.py:54-55); V0 = A with NaN -> 0; M = valid mask
__global__ void prep_kernel(const float* __restrict__ im, const float* __restrict__ bw, long long n, float* __restrict__ A,
                            float* __restrict__ V0, float* __restrict__ M) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        float x = im[i];
        if (bw[i] < 1e-7f) x = NAN;
        const bool ok = (x == x);
        A[i] = x;
        V0[i] = ok ? x : 0.f;
        M[i] = ok ? 1.f : 0.f;
    }
}

// Position `p` of a box line (box-relative, any integer) in the chunk's line of length `full`, reflected at the chunk's ends
// (scipy mode="reflect": d c b a | a b c d | d c b a), back in box coordinates: -1 when it falls outside the box (a zero).
__device__ __forceinline__ int box_reflect(int p, int b0, int len, int full) {
    int q = p + b0;
    if (full == 1) q = 0;
    else {
        const int period = 2 * full;
        q %= period; if (q < 0) q += period; if (q >= full) q = period - 1 - q;
    }
    q -= b0;
    return ((unsigned)q < (unsigned)len) ? q : -1;
}

// ---- the valid mask of a view as a BOX (round 5) ------------------------------------------------------------------------------
// The mask term of the NaN-aware Gaussian filters M = (view is finite) & !(normalised blending weight < 1e-7).  For an integer tile
// under a whole-pixel translation that set is a box -- the part of the tile inside the chunk minus the outermost layer, where the
// blend weight vanishes -- and a separable indicator mz(z) my(y) mx(x) stays separable under the line filters: after the z pass
// the array is A(z) my(y) mx(x), after the y pass B(z, y) mx(x), with A = float32(filter of mz) and B = float32(filter of A(z) my)
// -- the very sums the line kernels form (same order, same fused multiply-adds, same float32 roundings), on 1-D and 2-D tables
// instead of 3-D arrays.  So: (1) cb_mask_bbox_kernel counts the valid voxels of every view's box and takes their bounding box;
// count == volume of the bounding box <=> the mask IS that box (checked on the device, per view and chunk: nothing is assumed);
// (2) cb_mask_table_kernel builds A and B for both filters; (3) the mask workgroups of the z / y passes return at once, and the
// x pass -- which divides value by mask -- stages B(z, y) mx(x) instead of loading a filtered mask.  A view whose mask is not a
// box (rotated views, float tiles holding NaNs) keeps the filtered path, decided by the same record.  Per view and filter two
// of the three mask passes disappear: a third of all line-filter work and of its HBM traffic.
struct CbMaskRec { unsigned long long cnt; int lo[3]; int hi[3]; };      // lo / hi: box-local bounding box of the valid voxels
static_assert(sizeof(CbMaskRec) == 32, "CbMaskRec layout");
__device__ __forceinline__ bool cb_mask_is_box(const CbMaskRec& r) {
    if (r.cnt == 0ull) return false;
    const unsigned long long vol = (unsigned long long)(r.hi[0] - r.lo[0] + 1) * (unsigned long long)(r.hi[1] - r.lo[1] + 1) *
                                   (unsigned long long)(r.hi[2] - r.lo[2] + 1);
    return vol == r.cnt;
}

// blockIdx.y = view; the workgroups of a view stride over its box
__global__ __launch_bounds__(256) void cb_mask_bbox_kernel(const float* __restrict__ im, const float* __restrict__ bw, CbBoxes8 BX, CbMaskRec* __restrict__ recs) {
    const CbBox32 B = BX.b[blockIdx.y];
    const long long n = (long long)B.n[0] * B.n[1] * B.n[2];
    unsigned long long cnt = 0;
    int lo[3] = {0x7fffffff, 0x7fffffff, 0x7fffffff}, hi[3] = {-1, -1, -1};
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const float x = im[B.off + i], w = bw[B.off + i];
        if ((x == x) && !(w < 1e-7f)) {
            const int bx = (int)(i % B.n[2]);
            const long long t = i / B.n[2];
            const int by = (int)(t % B.n[1]), bz = (int)(t / B.n[1]);
            ++cnt;
            lo[0] = min(lo[0], bz); hi[0] = max(hi[0], bz);
            lo[1] = min(lo[1], by); hi[1] = max(hi[1], by);
            lo[2] = min(lo[2], bx); hi[2] = max(hi[2], bx);
        }
    }
    for (int off = 32; off > 0; off >>= 1) {
        cnt += __shfl_down(cnt, off);
        for (int k = 0; k < 3; ++k) { lo[k] = min(lo[k], __shfl_down(lo[k], off)); hi[k] = max(hi[k], __shfl_down(hi[k], off)); }
    }
    __shared__ unsigned long long s_cnt[4];
    __shared__ int s_lo[4][3], s_hi[4][3];
    const int w = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) {
        s_cnt[w] = cnt;
        for (int k = 0; k < 3; ++k) { s_lo[w][k] = lo[k]; s_hi[w][k] = hi[k]; }
    }
    __syncthreads();
    if (threadIdx.x == 0) {      // one set of atomics per workgroup (a few hundred per view)
        for (int q = 1; q < 4; ++q) {
            cnt += s_cnt[q];
            for (int k = 0; k < 3; ++k) { lo[k] = min(lo[k], s_lo[q][k]); hi[k] = max(hi[k], s_hi[q][k]); }
        }
        if (cnt) {
            CbMaskRec* r = recs + blockIdx.y;
            atomicAdd(&r->cnt, cnt);
            for (int k = 0; k < 3; ++k) { atomicMin(&r->lo[k], lo[k]); atomicMax(&r->hi[k], hi[k]); }
        }
    }
}

// tables of one view and filter: B(z, y) (box-local, n[0] x n[1] floats).  grid = (n[0] max over views, views, 2 filters)
__global__ __launch_bounds__(256) void cb_mask_table_kernel(const CbMaskRec* __restrict__ recs, CbBoxes8 BX, Shape3 S, int ndim, int r1,
                                                            const double* __restrict__ fw1, int r2, const double* __restrict__ fw2,
                                                            float* __restrict__ tables, const long long* __restrict__ table_off) {
    const int v = blockIdx.y, f = blockIdx.z, z = blockIdx.x;
    const CbBox32 B = BX.b[v];
    if (z >= B.n[0] || B.n[1] <= 0 || B.n[2] <= 0) return;
    const CbMaskRec R = recs[v];
    if (!cb_mask_is_box(R)) return;
    const int radius = f ? r2 : r1;
    const double* fw = f ? fw2 : fw1;
    // A(z): the z pass over the indicator of [lo0, hi0] (3D); the raw indicator itself when there is no z pass (2D: n[0] == 1)
    float Az;
    if (ndim == 3) {
        auto mz = [&](int p) -> double {
            int q = p;
            if ((unsigned)q >= (unsigned)B.n[0]) q = box_reflect(p, B.lo[0], B.n[0], S.nz);
            return (q >= 0 && q >= R.lo[0] && q <= R.hi[0]) ? 1.0 : 0.0;
        };
        double acc = mz(z) * fw[radius];
        for (int j = radius; j >= 1; --j) acc = fma(mz(z - j) + mz(z + j), fw[radius - j], acc);
        Az = (float)acc;
    } else {
        Az = (z >= R.lo[0] && z <= R.hi[0]) ? 1.f : 0.f;
    }
    float* out = tables + table_off[v * 2 + f] + (long long)z * B.n[1];
    const double a = (double)Az;
    for (int y = threadIdx.x; y < B.n[1]; y += blockDim.x) {
        auto my = [&](int p) -> double {
            int q = p;
            if ((unsigned)q >= (unsigned)B.n[1]) q = box_reflect(p, B.lo[1], B.n[1], S.ny);
            return (q >= 0 && q >= R.lo[1] && q <= R.hi[1]) ? a : 0.0;
        };
        double acc = my(y) * fw[radius];
        for (int j = radius; j >= 1; --j) acc = fma(my(y - j) + my(y + j), fw[radius - j], acc);
        out[y] = (float)acc;
    }
}

// scipy.ndimage.correlate1d with a symmetric kernel along one axis, mode="reflect": double accumulation in
// scipy's order (centre tap, then pairs from the farthest to the nearest), float32 output.  The array is a box of the chunk:
// b0 = its first position along the axis inside the chunk line, full = the chunk line's length.
__global__ __launch_bounds__(256) void gauss1d_kernel(const float* __restrict__ src, float* __restrict__ dst, Shape3 S, int axis,
                                                      int radius, const double* __restrict__ fw, int b0, int full) {
    const long long n = (long long)S.nz * S.ny * S.nx;
    const int dims[3] = {S.nz, S.ny, S.nx};
    const long long strides[3] = {(long long)S.ny * S.nx, S.nx, 1};
    const int len = dims[axis];
    const long long st = strides[axis];
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const int x = (int)(i % S.nx);
        const long long t = i / S.nx;
        const int y = (int)(t % S.ny), z = (int)(t / S.ny);
        const int pos = (axis == 0) ? z : (axis == 1) ? y : x;
        const long long base = i - (long long)pos * st;
        double acc = (double)src[i] * fw[radius];
        for (int j = radius; j >= 1; --j) {
            const int p0 = box_reflect(pos - j, b0, len, full), p1 = box_reflect(pos + j, b0, len, full);
            const double a0 = p0 >= 0 ? (double)src[base + (long long)p0 * st] : 0.0;
            const double a1 = p1 >= 0 ? (double)src[base + (long long)p1 * st] : 0.0;
            acc += (a0 + a1) * fw[radius - j];
        }
        dst[i] = (float)acc;
    }
}

// The same filter with the lines staged in LDS: a workgroup takes T adjacent lines, copies them (+ the reflected halo
// of `radius` samples on both sides) into LDS once and every output reads its 2 * radius + 1 taps from there -- no
// index arithmetic, no reflection and no global load per tap (the tap-by-tap kernel above spends its time there: the
// sigma = 11 filter has 89 taps).  Same accumulation order, same rounding.  LDS layout [pos + radius][line], line
// pitch T + 1 (odd) so that both access directions are bank-conflict free.
constexpr int kGaussK = 8;      // outputs per thread of the LDS line filters
struct GaussLines { long long n_lines; int len; long long stride; long long inner; long long outer_stride; int T; int b0, full; };

__global__ __launch_bounds__(256) void gauss1d_lds_kernel(const float* __restrict__ src, float* __restrict__ dst, GaussLines L, int radius,
                                                          const double* __restrict__ fw, int pos_fastest) {
    extern __shared__ float sl[];
    const int T = L.T, TP = T + 1, len = L.len;
    const long long l0 = (long long)blockIdx.x * T;
    const int nl = (int)min((long long)T, L.n_lines - l0);
    const int total = T * len;
    // ---- stage the lines; note whether every sample of the tile has the same bits as its first one ----
    const float first = src[(l0 / L.inner) * L.outer_stride + (l0 % L.inner)];
    int same = 1;
    for (int idx = threadIdx.x; idx < total; idx += blockDim.x) {
        int line, pos;
        if (pos_fastest) { line = idx / len; pos = idx - line * len; }
        else { pos = idx / T; line = idx - pos * T; }
        float v = 0.f;
        if (line < nl) {
            const long long l = l0 + line;
            v = src[(l / L.inner) * L.outer_stride + (l % L.inner) + (long long)pos * L.stride];
            same &= (__float_as_uint(v) == __float_as_uint(first)) ? 1 : 0;
        }
        sl[(pos + radius) * TP + line] = v;
    }
    // (only when the box line IS the chunk line -- zeros beyond an end would change the sums -- or the constant is 0)
    const bool whole = (L.b0 == 0 && L.len == L.full) || __float_as_uint(first) == 0u;
    if (__syncthreads_and(same) && whole) {
        // a constant tile (outside a view's footprint: value and mask 0; deep inside it: mask 1, and the constants the
        // earlier axes made of those): every output is the same sum, evaluated once in the order of the general path
        double acc = (double)first * fw[radius];
        for (int j = radius; j >= 1; --j) acc = fma((double)first + (double)first, fw[radius - j], acc);
        const float r = (float)acc;
        for (int idx = threadIdx.x; idx < total; idx += blockDim.x) {
            int line, pos;
            if (pos_fastest) { line = idx / len; pos = idx - line * len; }
            else { pos = idx / T; line = idx - pos * T; }
            if (line >= nl) continue;
            const long long l = l0 + line;
            dst[(l / L.inner) * L.outer_stride + (l % L.inner) + (long long)pos * L.stride] = r;
        }
        return;
    }
    // ---- halo: the chunk line reflected at the chunk's ends (d c b a | a b c d | d c b a), zeros where that falls outside
    // the box ----
    for (int idx = threadIdx.x; idx < 2 * radius * T; idx += blockDim.x) {
        const int h = idx / T, line = idx - h * T;
        const int p = (h < radius) ? (h - radius) : (len + h - radius);      // position outside [0, len)
        const int q = box_reflect(p, L.b0, len, L.full);
        sl[(p + radius) * TP + line] = (q >= 0) ? sl[(q + radius) * TP + line] : 0.f;
    }
    __syncthreads();
    // ---- filter: a thread produces K consecutive outputs of one line.  Pair j of output k needs the samples k - j and
    // k + j: over the K outputs these are two windows of K samples that slide by one (in opposite directions) when j
    // drops by one, so every step costs two LDS reads and two conversions for K (add, multiply, add) triples -- the
    // tap-by-tap form read and converted 2 (2 r + 1) samples per output and was bound by exactly that.  The windows
    // rotate through fixed registers (the j loop is unrolled K-fold); per output the order of operations is scipy's, with
    // the multiply-add fused (one rounding less in float64, 1e-16 relative: invisible after the float32 store). ----
    constexpr int K = 8;
    const int nblk = (len + K - 1) / K;
    const int qmax = len - 1 + 2 * radius;                        // last staged row of the LDS array
    for (int idx = threadIdx.x; idx < T * nblk; idx += blockDim.x) {
        const int blk = idx / T, line = idx - blk * T;
        if (line >= nl) continue;
        const int p0 = blk * K;
        const float* c = sl + line;                                // sample at position q: c[(q + radius) * TP]
        double PA[K], PB[K], acc[K];
#pragma unroll
        for (int k = 0; k < K; ++k) {
            PA[k] = (double)c[min(p0 + k, qmax) * TP];                                  // position p0 + k - radius
            PB[k] = (double)c[min(p0 + k + 2 * radius, qmax) * TP];                     // position p0 + k + radius
            acc[k] = (double)c[min(p0 + k + radius, qmax) * TP] * fw[radius];
        }
        for (int jb = radius; jb >= 1; jb -= K) {
#pragma unroll
            for (int s = 0; s < K; ++s) {
                const int j = jb - s;
                if (j < 1) break;
                const double w = fw[radius - j];
#pragma unroll
                for (int k = 0; k < K; ++k) acc[k] = fma(PA[(k + s) % K] + PB[(k - s + K) % K], w, acc[k]);
                // windows of pair j - 1: one new sample each (positions p0 + K - 1 - (j - 1) and p0 + (j - 1))
                PA[s % K] = (double)c[min(p0 + K - j + radius, qmax) * TP];
                PB[(K - 1 - s) % K] = (double)c[(p0 + j - 1 + radius) * TP];
            }
        }
        const long long l = l0 + line;
        const long long obase = (l / L.inner) * L.outer_stride + (l % L.inner) + (long long)p0 * L.stride;
        if (L.stride == 1 && p0 + K <= len && ((obase & 3) == 0)) {
            float4* o4 = reinterpret_cast<float4*>(dst + obase);
            o4[0] = make_float4((float)acc[0], (float)acc[1], (float)acc[2], (float)acc[3]);
            o4[1] = make_float4((float)acc[4], (float)acc[5], (float)acc[6], (float)acc[7]);
        } else {
#pragma unroll
            for (int k = 0; k < K; ++k)
                if (p0 + k < len) dst[obase + (long long)k * L.stride] = (float)acc[k];
        }
    }
}

// ---- the value line and the mask line of a NaN-aware Gaussian in ONE launch (round 4) -----------------------------------------
// NG_s(U) = gaussian(U, NaN -> 0) / gaussian(valid mask): both filters read the same voxels, so a workgroup stages the T lines of
// BOTH arrays, filters both and -- on the last axis -- divides them on the spot.  What no longer travels through HBM:
//   * prep_kernel's three arrays (A, V0, M): the first pass derives value and mask from the resampled view and its normalised
//     blending weight itself (SRC_PREP; SRC_VMASK for the second filter, whose value input is the squared deviation);
//   * the separate finish kernels: the last pass of the first filter stores (A - VV / WW)^2 with NaN -> 0 (DST_SQ), the last pass
//     of the second one F = VV / WW with NaN where A is NaN (DST_F), A recomputed from the view at the output voxel;
//   * half of the launches (6 instead of 12 line passes per view) and the second staging of every line set.
// Same arithmetic as gauss1d_lds_kernel per quantity (scipy's order, double accumulation, float32 result per axis, the
// constant-tile short cut per quantity).
enum { SRC_AB = 0, SRC_PREP = 1, SRC_VMASK = 2, DST_AB = 0, DST_SQ = 1, DST_F = 2 };
struct PairIO {
    const float* a; const float* b;        // SRC_AB: value / mask lines; SRC_VMASK: a = value lines
    const float* im; const float* bw;      // resampled view and normalised blending weight (SRC_PREP / SRC_VMASK / DST_SQ / DST_F)
    float* oa; float* ob;                  // DST_AB: both results; DST_SQ / DST_F: oa = the single result
    int src, dst;
    const CbMaskRec* rec;                  // the view's mask record (NULL: the mask is always filtered) ...
    const float* mtab;                     // ... and its table B(z, y) for this filter (cb_mask_table_kernel)
};

__device__ __forceinline__ float cb_valid_value(const PairIO& P, long long i) {      // A: the view with NaN where bw < 1e-7
    float x = P.im[i];
    if (P.bw[i] < 1e-7f) x = NAN;
    return x;
}

// SPLIT: the passes that are not the last of their filter need no coupling between the two quantities (DST_AB), so a workgroup
// takes ONE of them (blockIdx.y: 0 value, 1 mask) and stages twice as many lines in the same LDS (T = 32: lines along y / z then
// move in 128-byte pieces instead of 64-byte ones); still one launch per pass.
#ifndef MVS_CB_NB
#define MVS_CB_NB 8      // samples a thread requests back to back while staging (memory-level parallelism of a workgroup)
#endif
template <int SRC, int DST, bool SPLIT>
__global__ __launch_bounds__(256) void gauss1d_pair_kernel(PairIO P, GaussLines L, int radius, const double* __restrict__ fw, int pos_fastest) {
    static_assert(!SPLIT || DST == DST_AB, "split passes store both quantities as they are");
    extern __shared__ float sl[];
    // T is a power of two (host: pair_T / split_T), so (line, position) come out of shifts and masks; the lines' first elements
    // are worked out once per workgroup (one 64-bit division per LINE instead of two per staged and stored SAMPLE)
    constexpr int NQ = SPLIT ? 1 : 2;
    const int qs = SPLIT ? (int)blockIdx.y : 0;       // SPLIT: the quantity of this workgroup
    // the view's mask is a box (see cb_mask_bbox_kernel): its z / y passes are tables, the x pass stages B(z, y) mx(x)
    bool mbox = false;
    int mx0 = 0, mx1 = -1;
    if (P.rec) {
        const CbMaskRec R = *P.rec;
        mbox = cb_mask_is_box(R);
        mx0 = R.lo[2]; mx1 = R.hi[2];
    }
    if (SPLIT && qs == 1 && mbox) return;             // (uniform: the whole workgroup)
    const int T = L.T, TP = T + 1, len = L.len, lt = 31 - __clz(T);
    const int span = len + 2 * radius;
    float* sq[2] = {sl, sl + (SPLIT ? 0 : (size_t)(span + kGaussK) * TP)};      // (kGaussK spare rows behind each array: see the filter loop)
    __shared__ long long lbase[32];
    const long long l0 = (long long)blockIdx.x * T;
    const int nl = (int)min((long long)T, L.n_lines - l0);
    if ((int)threadIdx.x < T) {
        const long long l = min(l0 + threadIdx.x, L.n_lines - 1);
        lbase[threadIdx.x] = (l / L.inner) * L.outer_stride + (l % L.inner);
    }
    __syncthreads();
    // raw loads first, interpretation afterwards: a batch of samples is requested without any control flow in between.
    // SPLIT: two loads from X / Y -- the array itself twice where the quantity is stored as it is (value of SRC_AB / SRC_VMASK,
    // mask of SRC_AB), the view and its blending weight where it is derived from them.
    const bool direct = SPLIT && (SRC == SRC_AB || (SRC == SRC_VMASK && qs == 0));
    const float* X = !SPLIT ? nullptr : (SRC == SRC_AB ? (qs ? P.b : P.a) : (direct ? P.a : P.im));
    const float* Y = !SPLIT ? nullptr : (direct ? X : P.bw);
    auto load_raw = [&](long long i, float& r0, float& r1, float& r2, auto one_array) {
        if constexpr (SPLIT) { r0 = X[i]; r1 = decltype(one_array)::value ? r0 : Y[i]; r2 = 0.f; }
        else if constexpr (SRC == SRC_AB) { r0 = P.a[i]; r1 = mbox ? 0.f : P.b[i]; r2 = 0.f; }
        else if constexpr (SRC == SRC_PREP) { r0 = P.im[i]; r1 = P.bw[i]; r2 = 0.f; }
        else { r0 = P.im[i]; r1 = P.bw[i]; r2 = P.a[i]; }
    };
    auto interpret = [&](float r0, float r1, float r2, float& v, float& m) {      // SPLIT: the workgroup's quantity comes back in v
        if constexpr (SRC == SRC_AB) { v = r0; m = r1; }
        else {
            const bool ok = (r0 == r0) && !(r1 < 1e-7f);      // A = the view with NaN where bw < 1e-7
            m = ok ? 1.f : 0.f;
            v = (SRC == SRC_PREP) ? (ok ? r0 : 0.f) : r2;
            if constexpr (SPLIT) v = direct ? r0 : (qs ? m : v);
        }
    };
    // ---- stage the lines; note per quantity whether every sample has the bits of the first one ----
    float first[2];
    {
        float r0, r1, r2;
        load_raw(lbase[0], r0, r1, r2, std::false_type{});
        interpret(r0, r1, r2, first[0], first[1]);
        if constexpr (!SPLIT && SRC == SRC_AB) {
            if (mbox) first[1] = (0 >= mx0 && 0 <= mx1) ? P.mtab[l0] : 0.f;
        }
    }
    int same0 = 1, same1 = 1;
    // A workgroup is a short dependent chain (stage -> filter -> store) and only a few of them fit a CU, so the staging loop
    // must not pay one memory round trip per sample: the loads of NB samples are issued back to back before the first of them
    // is written to LDS.
    constexpr int NB = MVS_CB_NB;
    const int per_line = pos_fastest ? (int)blockDim.x : ((int)blockDim.x >> lt);      // positions a sweep of the workgroup covers per line
    const int my_line = pos_fastest ? 0 : (int)(threadIdx.x & (T - 1));
    const int my_pos0 = pos_fastest ? (int)threadIdx.x : (int)(threadIdx.x >> lt);
    const int sweeps = (len + per_line - 1) / per_line;
    const int n_my = pos_fastest ? sweeps * T : sweeps;             // samples of this thread: (sweep[, line]) pairs
    auto stage = [&](auto one_array) __attribute__((always_inline)) {
    for (int b0 = 0; b0 < n_my; b0 += NB) {
        float q0[NB], q1[NB], q2[NB];
        int bl[NB], bp[NB];
#pragma unroll
        for (int u = 0; u < NB; ++u) {
            const int e = b0 + u;
            // pos_fastest: e = line * sweeps + sweep (threads run along a line); else e = sweep (threads run across the lines)
            const int line = pos_fastest ? e / sweeps : my_line;
            const int sweep = pos_fastest ? e - line * sweeps : e;
            const int pos = my_pos0 + sweep * per_line;
            bl[u] = line; bp[u] = pos;
            // (clamped address: the load itself is unconditional, what it returns is discarded below when out of range)
            load_raw(lbase[min(line, nl - 1)] + (long long)min(pos, len - 1) * L.stride, q0[u], q1[u], q2[u], one_array);
            if constexpr (!SPLIT && SRC == SRC_AB) {
                if (mbox) q1[u] = P.mtab[l0 + min(line, nl - 1)];      // (requested with the batch: the table entry of line l = (z, y))
            }
        }
#pragma unroll
        for (int u = 0; u < NB; ++u) {
            const int e = b0 + u;
            if (e < n_my && bp[u] < len && bl[u] < T) {
                float v = 0.f, m = 0.f;
                if (bl[u] < nl) {
                    interpret(q0[u], q1[u], q2[u], v, m);
                    if constexpr (!SPLIT && SRC == SRC_AB) {
                        if (mbox) m = (bp[u] >= mx0 && bp[u] <= mx1) ? q1[u] : 0.f;      // B(z, y) mx(x)
                    }
                    same0 &= (__float_as_uint(v) == __float_as_uint(first[0])) ? 1 : 0;
                    if (!SPLIT) same1 &= (__float_as_uint(m) == __float_as_uint(first[1])) ? 1 : 0;
                }
                sq[0][(bp[u] + radius) * TP + bl[u]] = v;
                if (!SPLIT) sq[1][(bp[u] + radius) * TP + bl[u]] = m;
            }
        }
    }
    };
    // (a quantity stored as it is needs ONE array: the second load of the general form would only occupy a slot of the batch)
    if (direct) stage(std::true_type{});
    else stage(std::false_type{});
    const bool box_is_line = (L.b0 == 0 && L.len == L.full);
    // (__syncthreads_or reduces the TRUTH of its argument, not its bits: one reduction per quantity)
    const int varies0 = __syncthreads_or(same0 ? 0 : 1);
    const int varies1 = SPLIT ? 1 : __syncthreads_or(same1 ? 0 : 1);
    bool cst[2];
    cst[0] = !varies0 && (box_is_line || __float_as_uint(first[0]) == 0u);
    cst[1] = !SPLIT && !varies1 && (box_is_line || __float_as_uint(first[1]) == 0u);
    float cval[2] = {0.f, 0.f};
#pragma unroll
    for (int q = 0; q < NQ; ++q)
        if (cst[q]) {
            double acc = (double)first[q] * fw[radius];
            for (int j = radius; j >= 1; --j) acc = fma((double)first[q] + (double)first[q], fw[radius - j], acc);
            cval[q] = (float)acc;
        }
    // ---- halo of the quantities that are filtered: chunk line reflected at the chunk's ends, zeros where that leaves the box ----
    for (int idx = threadIdx.x; idx < 2 * radius * T; idx += blockDim.x) {
        const int h = idx >> lt, line = idx & (T - 1);
        const int p = (h < radius) ? (h - radius) : (len + h - radius);
        const int q = box_reflect(p, L.b0, len, L.full);
#pragma unroll
        for (int k = 0; k < NQ; ++k)
            if (!cst[k]) sq[k][(p + radius) * TP + line] = (q >= 0) ? sq[k][(q + radius) * TP + line] : 0.f;
    }
    __syncthreads();
    // ---- filter: K consecutive outputs of one line per thread (the scheme of gauss1d_lds_kernel).  What differs is how a step
    // is fed: the windows' new samples come from two LDS pointers that move by one row per pair (no index arithmetic, no clamp:
    // the arrays end in K spare rows, whatever they hold only reaches outputs beyond the line's end, which are not stored), the
    // weights of K pairs are fetched together, and the K-pair blocks of the loop are free of branches, so that the reads and
    // conversions of a pair overlap the float64 arithmetic of the previous one. ----
    constexpr int K = kGaussK;
    const int nblk = (len + K - 1) / K;
    for (int idx = threadIdx.x; idx < T * nblk; idx += blockDim.x) {
        const int blk = idx >> lt, line = idx & (T - 1);
        if (line >= nl) continue;
        const int p0 = blk * K;
        float res[2][K];
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            if (cst[q]) {
#pragma unroll
                for (int k = 0; k < K; ++k) res[q][k] = cval[q];
                continue;
            }
            const float* c = sq[q] + line;                  // sample at position x: c[(x + radius) * TP]
            double PA[K], PB[K], acc[K];
            const double wc = fw[radius];
#pragma unroll
            for (int k = 0; k < K; ++k) {
                PA[k] = (double)c[(p0 + k) * TP];
                PB[k] = (double)c[(p0 + k + 2 * radius) * TP];
                acc[k] = (double)c[(p0 + k + radius) * TP] * wc;
            }
            // pair j - 1 needs one new sample per window: rows p0 + K + (radius - j) and p0 + 2 radius - 1 - (radius - j)
            const float* qa = c + (p0 + K) * TP;
            const float* qb = c + (p0 + 2 * radius - 1) * TP;
            int jb = radius;
            for (; jb >= K; jb -= K) {
                double w[K];
#pragma unroll
                for (int s = 0; s < K; ++s) w[s] = fw[radius - jb + s];
#pragma unroll
                for (int s = 0; s < K; ++s) {
#pragma unroll
                    for (int k = 0; k < K; ++k) acc[k] = fma(PA[(k + s) % K] + PB[(k - s + K) % K], w[s], acc[k]);
                    PA[s] = (double)qa[s * TP];
                    PB[K - 1 - s] = (double)qb[-s * TP];
                }
                qa += K * TP;
                qb -= K * TP;
            }
#pragma unroll
            for (int s = 0; s < K - 1; ++s) {                // the remaining jb < K pairs
                if (s >= jb) break;
                const double w = fw[radius - jb + s];
#pragma unroll
                for (int k = 0; k < K; ++k) acc[k] = fma(PA[(k + s) % K] + PB[(k - s + K) % K], w, acc[k]);
                PA[s] = (double)qa[s * TP];
                PB[K - 1 - s] = (double)qb[-s * TP];
            }
#pragma unroll
            for (int k = 0; k < K; ++k) res[q][k] = (float)acc[k];
        }
        const long long obase = lbase[line] + (long long)p0 * L.stride;
        if constexpr (SPLIT) {
            float* o = qs ? P.ob : P.oa;
            if (L.stride == 1 && p0 + K <= len && ((obase & 3) == 0)) {
                float4* o4 = reinterpret_cast<float4*>(o + obase);
                o4[0] = make_float4(res[0][0], res[0][1], res[0][2], res[0][3]);
                o4[1] = make_float4(res[0][4], res[0][5], res[0][6], res[0][7]);
            } else {
#pragma unroll
                for (int k = 0; k < K; ++k)
                    if (p0 + k < len) o[obase + (long long)k * L.stride] = res[0][k];
            }
        } else if constexpr (DST == DST_AB) {
            if (L.stride == 1 && p0 + K <= len && ((obase & 3) == 0)) {
                float4* o4 = reinterpret_cast<float4*>(P.oa + obase);
                o4[0] = make_float4(res[0][0], res[0][1], res[0][2], res[0][3]);
                o4[1] = make_float4(res[0][4], res[0][5], res[0][6], res[0][7]);
                o4 = reinterpret_cast<float4*>(P.ob + obase);
                o4[0] = make_float4(res[1][0], res[1][1], res[1][2], res[1][3]);
                o4[1] = make_float4(res[1][4], res[1][5], res[1][6], res[1][7]);
            } else {
#pragma unroll
                for (int k = 0; k < K; ++k)
                    if (p0 + k < len) { P.oa[obase + (long long)k * L.stride] = res[0][k]; P.ob[obase + (long long)k * L.stride] = res[1][k]; }
            }
        } else {
            // Z = VV / WW where A is valid (weights.py:314-320); DST_SQ: (A - Z)^2 with NaN -> 0; DST_F: Z, NaN where A is NaN
#pragma unroll
            for (int k = 0; k < K; ++k) {
                if (p0 + k >= len) break;
                const long long i = obase + (long long)k * L.stride;
                const float a = cb_valid_value(P, i);
                float o;
                if constexpr (DST == DST_SQ) {
                    o = 0.f;
                    if (a == a) {
                        const float d = a - res[0][k] / res[1][k];
                        const float qd = d * d;
                        o = (qd == qd) ? qd : 0.f;
                    }
                } else {
                    o = (a == a) ? res[0][k] / res[1][k] : NAN;
                }
                P.oa[i] = o;
            }
        }
    }
}

// Z = VV / WW with WW[nan] = 1, Z[nan] = NaN (weights.py:314-320); then D = (A - Z)^2, V1 = D with NaN -> 0
__global__ void ng_finish_sq_kernel(const float* __restrict__ VV, const float* __restrict__ WW, const float* __restrict__ A,
                                    long long n, float* __restrict__ V1) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const float a = A[i];
        if (a != a) { V1[i] = 0.f; continue; }
        const float zv = VV[i] / WW[i];
        const float d = a - zv;
        const float q = d * d;
        V1[i] = (q == q) ? q : 0.f;   // a finite -> q finite unless WW == 0 (cannot happen where a is valid)
    }
}

// F = VV2 / WW2 with NaN where A is NaN
__global__ void ng_finish_kernel(const float* __restrict__ VV, const float* __restrict__ WW, const float* __restrict__ A,
                                 long long n, float* __restrict__ F) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const float a = A[i];
        F[i] = (a == a) ? VV[i] / WW[i] : NAN;
    }
}

template <typename TOut> __device__ __forceinline__ TOut cast_cb(float v);
template <> __device__ __forceinline__ float cast_cb<float>(float v) { return v; }
template <> __device__ __forceinline__ unsigned short cast_cb<unsigned short>(float v) { return (unsigned short)(int)v; }
template <> __device__ __forceinline__ unsigned char cast_cb<unsigned char>(float v) { return (unsigned char)(int)v; }

// normalise F over views (nansum, 0 -> 1), A = bw * Fn, normalise A, out = nansum(I * A); trimmed, nan_to_num, cast.
// Views whose box does not hold the voxel are NaN there (every term they would add is skipped by the reference's nansum).
template <typename TOut>
__global__ void cb_fuse_kernel(const float* __restrict__ im, const float* __restrict__ bw, const float* __restrict__ F,
                               const CbBox* __restrict__ boxes, int nviews, int tz, int ty, int tx, Shape3 O, TOut* __restrict__ out) {
    const long long no = (long long)O.nz * O.ny * O.nx;
    for (long long o = (long long)blockIdx.x * blockDim.x + threadIdx.x; o < no; o += (long long)gridDim.x * blockDim.x) {
        const int x = (int)(o % O.nx);
        const long long t = o / O.nx;
        const int y = (int)(t % O.ny), z = (int)(t / O.ny);
        float fsum = 0.f;
        for (int v = 0; v < nviews; ++v) {
            const long long k = box_index(boxes[v], z + tz, y + ty, x + tx);
            if (k < 0) continue;
            const float f = F[k];
            if (f == f) fsum += f;
        }
        if (fsum == 0.f) fsum = 1.f;
        float asum = 0.f;
        for (int v = 0; v < nviews; ++v) {
            const long long k = box_index(boxes[v], z + tz, y + ty, x + tx);
            if (k < 0) continue;
            const float a = bw[k] * (F[k] / fsum);
            if (a == a) asum += a;
        }
        if (asum == 0.f) asum = 1.f;
        float acc = 0.f;
        for (int v = 0; v < nviews; ++v) {
            const long long k = box_index(boxes[v], z + tz, y + ty, x + tx);
            if (k < 0) continue;
            const float a = bw[k] * (F[k] / fsum);
            const float p = im[k] * (a / asum);
            if (p == p) acc += p;
        }
        if (!(fabsf(acc) <= 3.4028234e38f)) acc = 0.f;
        out[o] = cast_cb<TOut>(acc);
    }
}

// cb_fuse_kernel for at most 8 views: indices, F and the weights stay in registers over the three sums (same additions in the
// same view order)
template <typename TOut>
__global__ __launch_bounds__(256) void cb_fuse8_kernel(const float* __restrict__ im, const float* __restrict__ bw, const float* __restrict__ F,
                                                       CbBoxes8 BX, int nviews, int tz, int ty, int tx, Shape3 O, TOut* __restrict__ out) {
    const long long no = (long long)O.nz * O.ny * O.nx;
    for (long long o = (long long)blockIdx.x * blockDim.x + threadIdx.x; o < no; o += (long long)gridDim.x * blockDim.x) {
        const int x = (int)(o % O.nx);
        const long long t = o / O.nx;
        const int y = (int)(t % O.ny), z = (int)(t / O.ny);
        int kk[8];
        float f[8], a[8];
        float fsum = 0.f;
#pragma unroll
        for (int v = 0; v < 8; ++v) {
            kk[v] = v < nviews ? box_index32(BX.b[v], z + tz, y + ty, x + tx) : -1;
            f[v] = 0.f;
            if (kk[v] >= 0) {
                f[v] = F[kk[v]];
                if (f[v] == f[v]) fsum += f[v];
            }
        }
        if (fsum == 0.f) fsum = 1.f;
        float asum = 0.f;
#pragma unroll
        for (int v = 0; v < 8; ++v) {
            a[v] = 0.f;
            if (kk[v] >= 0) {
                a[v] = bw[kk[v]] * (f[v] / fsum);
                if (a[v] == a[v]) asum += a[v];
            }
        }
        if (asum == 0.f) asum = 1.f;
        float acc = 0.f;
#pragma unroll
        for (int v = 0; v < 8; ++v)
            if (kk[v] >= 0) {
                const float p = im[kk[v]] * (a[v] / asum);
                if (p == p) acc += p;
            }
        if (!(fabsf(acc) <= 3.4028234e38f)) acc = 0.f;
        out[o] = cast_cb<TOut>(acc);
    }
}

// scipy.ndimage._filters._gaussian_kernel1d(sigma, 0, radius), radius = int(truncate * sigma + 0.5)
void gaussian_kernel(double sigma, int* radius_out, std::vector<double>* w) {
    const int radius = (int)(4.0 * sigma + 0.5);
    w->resize(2 * radius + 1);
    const double sigma2 = sigma * sigma;
    double sum = 0.0;
    for (int k = -radius; k <= radius; ++k) {
        const double v = exp(-0.5 / sigma2 * (double)k * (double)k);
        (*w)[k + radius] = v;
        sum += v;
    }
    for (auto& v : *w) v /= sum;
    *radius_out = radius;
}

#include "mvs_gauss_fast.inc"

}  // namespace

static int cb_fast_chunk(MvsContext* c, const mvs_view_t* views, int32_t n_views, const mvs_fuse_opts_t* opts, void* out, bool* taken);

int mvs_fuse_content_based(MvsContext* c, const mvs_view_t* views, int32_t n_views, const mvs_fuse_opts_t* opts, void* out) {
    if (opts->order != 0 && opts->order != 1) return mvs_fail(c, MVS_ERR_UNSUPPORTED, "content_based: order 0|1 only");
    if (!c->cb_exact) {      // the default: mask from a box + list, one quantity per pass, float32 taps (mvs_gauss_fast.inc)
        bool taken = false;
        const int rc = cb_fast_chunk(c, views, n_views, opts, out, &taken);
        if (rc || taken) return rc;
    }
    const int dtype = views[0].dtype;
    const size_t es = mvs_dtype_size(dtype);
    const int64_t* cs = opts->out_shape;
    const Shape3 S = {(int)cs[0], (int)cs[1], (int)cs[2]};
    const long long n = (long long)S.nz * S.ny * S.nx;
    int64_t os[3];
    for (int k = 0; k < 3; ++k) os[k] = cs[k] - 2 * opts->trim[k];
    const Shape3 O = {(int)os[0], (int)os[1], (int)os[2]};
    const long long no = (long long)O.nz * O.ny * O.nx;

    // ---- device views and their boxes inside the chunk ----
    std::vector<DevView> dvs((size_t)n_views);
    std::vector<CbBox> boxes((size_t)n_views);
    size_t host_bytes = 0;
    for (int i = 0; i < n_views; ++i)
        if (views[i].mem == MVS_MEM_HOST) {
            if (views[i].stride[1] != views[i].shape[2] || views[i].stride[0] != views[i].shape[1] * views[i].shape[2])
                return mvs_fail(c, MVS_ERR_UNSUPPORTED, "host slabs must be C-contiguous");
            host_bytes += ((size_t)views[i].shape[0] * views[i].shape[1] * views[i].shape[2] * es + 255) / 256 * 256;
        }
    char* slab_base = nullptr;
    if (host_bytes) {
        slab_base = (char*)mvs_scratch(c, 0, host_bytes);
        if (!slab_base) return mvs_alloc_failed(c);
    }
    MVS_HIP_TRY(c, hipEventRecord(c->ev_start, c->stream));
    size_t cursor = 0;
    long long pool = 0, max_box = 1;
    for (int i = 0; i < n_views; ++i) {
        const void* dptr = views[i].data;
        if (views[i].mem == MVS_MEM_HOST) {
            const size_t nb = (size_t)views[i].shape[0] * views[i].shape[1] * views[i].shape[2] * es;
            MVS_HIP_TRY(c, hipMemcpyAsync(slab_base + cursor, views[i].data, nb, hipMemcpyHostToDevice, c->stream));
            dptr = slab_base + cursor;
            cursor += (nb + 255) / 256 * 256;
        }
        int rc = mvs_fill_dev_view(c, views[i], opts->ndim, dptr, &dvs[i]);
        if (rc) return rc;
        mvs_view_to_chunk_frame(&dvs[i], opts->index_origin, views[i].index_offset);
        int lo[3], hi[3];
        mvs_view_chunk_box(dvs[i], cs, lo, hi);
        CbBox& B = boxes[i];
        long long bv = 1;
        for (int k = 0; k < 3; ++k) {
            B.lo[k] = lo[k];
            B.n[k] = std::max(hi[k] - lo[k] + 1, 0);
            bv *= B.n[k];
        }
        if (bv == 0) { B.n[0] = B.n[1] = B.n[2] = 0; }
        B.off = pool;
        pool += (bv + 63) / 64 * 64;
        max_box = std::max(max_box, bv);
    }

    // ---- scratch layout (slot 6): pools I, BW, F (one box per view), 6 temporaries of the largest box, filter kernels, boxes ----
    const size_t pool_b = (size_t)pool * 4, tmp_b = ((size_t)max_box * 4 + 255) / 256 * 256;
    // paired line passes (gauss1d_pair_kernel): can every line set of every view be staged twice in 60 KiB of LDS?
    const int ndim = opts->ndim;
    int r1, r2;
    std::vector<double> w1, w2;
    gaussian_kernel((double)opts->sigma_1, &r1, &w1);
    gaussian_kernel((double)opts->sigma_2, &r2, &w2);
    auto pair_T = [&](int len, int radius, int axis) {      // lines per workgroup of a paired pass, 0: does not fit
        const size_t span = (size_t)len + 2 * (size_t)radius;
        int T = (axis == 2) ? 8 : 16;
        while (T > 1 && span * (T + 1) * 8 > 60 * 1024) T >>= 1;
        if (span * (T + 1) * 8 > 60 * 1024 || (axis != 2 && T < 4)) return 0;
        return T;
    };
    auto split_T = [&](int len, int radius) {         // lines per workgroup of a split pass (one quantity in LDS), 0: does not fit
        const size_t span = (size_t)len + 2 * (size_t)radius;
        int T = 32;      // (8 / 16 / 32 measured: 30.9 / 28.5 / 28.1 ms per probe call)
        while (T > 1 && span * (T + 1) * 4 > 60 * 1024) T >>= 1;
        return (span * (T + 1) * 4 > 60 * 1024 || T < 8) ? 0 : T;
    };
    bool paired = !c->cb_unpaired;
    for (int i = 0; i < n_views && paired; ++i)
        for (int axis = 3 - ndim; axis < 3; ++axis)
            if (boxes[i].n[axis] > 0 && (!pair_T(boxes[i].n[axis], r1, axis) || !pair_T(boxes[i].n[axis], r2, axis))) paired = false;
    // temporaries of the largest box, shared by the views: 5 arrays on the paired path, 6 on the separate-pass path
    const size_t tmp_total = (paired ? 5 : 6) * tmp_b;
    // mask records + table offsets (uploaded with the block below) and the mask tables B(z, y) of every view and filter
    const bool mask_tables = paired && n_views <= 8 && pool < (1ll << 31) && c->cb_mask_closed_form;
    std::vector<long long> table_off((size_t)n_views * 2, 0);
    long long table_floats = 0;
    if (mask_tables)
        for (int i = 0; i < n_views; ++i)
            for (int f = 0; f < 2; ++f) {
                table_off[(size_t)i * 2 + f] = table_floats;
                table_floats += ((long long)boxes[i].n[0] * boxes[i].n[1] + 63) / 64 * 64;
            }
    const size_t need = 3 * pool_b + tmp_total + 64 * 1024 + (size_t)n_views * (sizeof(CbBox) + sizeof(DevView) + sizeof(CbMaskRec) + 16) + 4096 +
                        (size_t)table_floats * 4;
    char* base = (char*)mvs_scratch(c, 6, need);
    if (!base) return mvs_alloc_failed(c);
    float* I = (float*)base;
    float* BW = (float*)(base + pool_b);
    float* F = (float*)(base + 2 * pool_b);
    float* A = (float*)(base + 3 * pool_b);
    float* V0 = (float*)((char*)A + tmp_b);
    float* M = (float*)((char*)V0 + tmp_b);
    float* T0 = (float*)((char*)M + tmp_b);
    float* T1 = (float*)((char*)T0 + tmp_b);
    float* T2 = (float*)((char*)T1 + tmp_b);
    // filter kernels, boxes and view records: ONE device block in the layout of the host staging block below (one upload per chunk)
    char* dblock = (char*)(((uintptr_t)(base + 3 * pool_b + tmp_total) + 255) / 256 * 256);
    if ((w1.size() + w2.size()) * 8 > 32 * 1024) return mvs_fail(c, MVS_ERR_UNSUPPORTED, "content_based: sigma too large");
    const size_t wb = ((w1.size() + w2.size()) * 8 + 255) / 256 * 256, bb = ((size_t)n_views * sizeof(CbBox) + 255) / 256 * 256,
                 vb = ((size_t)n_views * sizeof(DevView) + 255) / 256 * 256, rb = ((size_t)n_views * sizeof(CbMaskRec) + 255) / 256 * 256,
                 ob = ((size_t)n_views * 16 + 255) / 256 * 256;
    double* dfw1 = (double*)dblock;
    double* dfw2 = dfw1 + w1.size();
    CbBox* dboxes = (CbBox*)(dblock + wb);
    DevView* dviews_dev = (DevView*)(dblock + wb + bb);      // the views' records for the batched box launches
    CbMaskRec* drecs = (CbMaskRec*)(dblock + wb + bb + vb);
    long long* dtable_off = (long long*)(dblock + wb + bb + vb + rb);
    float* dtables = (float*)(dblock + wb + bb + vb + rb + ob);
    // The block travels through the context's pinned staging slot (waited for before it is refilled: mvs_pinned_slot /
    // mvs_pinned_mark), so the call does not have to wait for its own work: with the result on the device it returns as soon as
    // everything is queued, and the host prepares the next chunk while this one is filtered (the probe spent ~0.4 ms per chunk idle)
    {
        char* hp = (char*)mvs_pinned_slot(c, 0, wb + bb + vb + rb + ob + 64);
        if (!hp) return mvs_alloc_failed(c);
        memcpy(hp, w1.data(), w1.size() * 8);
        memcpy(hp + w1.size() * 8, w2.data(), w2.size() * 8);
        memcpy(hp + wb, boxes.data(), (size_t)n_views * sizeof(CbBox));
        memcpy(hp + wb + bb, &dvs[0], (size_t)n_views * sizeof(DevView));
        for (int i = 0; i < n_views; ++i) {      // empty records: count 0, inverted bounding box
            CbMaskRec r;
            r.cnt = 0;
            for (int k = 0; k < 3; ++k) { r.lo[k] = 0x7fffffff; r.hi[k] = -1; }
            memcpy(hp + wb + bb + vb + (size_t)i * sizeof(CbMaskRec), &r, sizeof(r));
        }
        memcpy(hp + wb + bb + vb + rb, table_off.data(), (size_t)n_views * 16);
        { const int rcu = mvs_upload_small(c, dblock, hp, wb + bb + vb + rb + ob); if (rcu) return rcu; }
        mvs_pinned_mark(c, 0);
    }

    {   // resampled views and blend weights on the views' boxes: two launches for all views of the chunk (<= 8 views)
        std::vector<float*> res_out(n_views), blend_out(n_views);
        std::vector<int64_t> shp((size_t)n_views * 3);
        std::vector<int> b0((size_t)n_views * 3);
        for (int i = 0; i < n_views; ++i) {
            const CbBox& B = boxes[i];
            res_out[i] = I + B.off;
            blend_out[i] = BW + B.off;
            for (int k = 0; k < 3; ++k) { shp[(size_t)i * 3 + k] = B.n[k]; b0[(size_t)i * 3 + k] = B.lo[k]; }
        }
        mvs_launch_boxes_batch(c, &dvs[0], dviews_dev, n_views, dtype, opts->order, NAN, res_out.data(), blend_out.data(),
                               (const int64_t (*)[3])shp.data(), (const int (*)[3])b0.data());
    }
    const int gb = grid_for(n);
    // <= 8 views and a pool below 2^31 floats: boxes as a kernel argument, 32-bit indices kept in registers
    const bool small = n_views <= 8 && pool < (1ll << 31);
    CbBoxes8 bx8;
    memset(&bx8, 0, sizeof(bx8));
    if (small)
        for (int i = 0; i < n_views; ++i) {
            for (int k = 0; k < 3; ++k) { bx8.b[i].lo[k] = boxes[i].lo[k]; bx8.b[i].n[k] = boxes[i].n[k]; }
            bx8.b[i].off = (int)boxes[i].off;
        }
    if (small) hipLaunchKernelGGL(mask_normalize8_kernel, dim3(gb), dim3(256), 0, c->stream, BW, I, bx8, n_views, S, 0);
    else hipLaunchKernelGGL(mask_normalize_kernel, dim3(gb), dim3(256), 0, c->stream, BW, I, dboxes, n_views, S);

    if (mask_tables) {      // is every view's valid mask a box?  (device-side: record + tables, no host round trip)
        long long max_bv = 1;
        int max_nz = 1;
        for (int i = 0; i < n_views; ++i) {
            max_bv = std::max(max_bv, (long long)boxes[i].n[0] * boxes[i].n[1] * boxes[i].n[2]);
            max_nz = std::max(max_nz, boxes[i].n[0]);
        }
        hipLaunchKernelGGL(cb_mask_bbox_kernel, dim3((unsigned)std::min<long long>((max_bv + 2047) / 2048, 512), n_views), dim3(256), 0, c->stream,
                           I, BW, bx8, drecs);
        hipLaunchKernelGGL(cb_mask_table_kernel, dim3(max_nz, n_views, 2), dim3(256), 0, c->stream, drecs, bx8, S, ndim, r1, dfw1, r2, dfw2,
                           dtables, dtable_off);
        if (c->cb_mask_count) {      // test switch: read the records back and count the views whose mask was found to be a box
            std::vector<CbMaskRec> hrec((size_t)n_views);
            MVS_HIP_TRY(c, hipMemcpyAsync(hrec.data(), drecs, (size_t)n_views * sizeof(CbMaskRec), hipMemcpyDeviceToHost, c->stream));
            MVS_HIP_TRY(c, hipStreamSynchronize(c->stream));
            for (const CbMaskRec& r : hrec) {
                const unsigned long long vol = r.cnt ? (unsigned long long)(r.hi[0] - r.lo[0] + 1) * (unsigned long long)(r.hi[1] - r.lo[1] + 1) *
                                                       (unsigned long long)(r.hi[2] - r.lo[2] + 1) : 0ull;
                c->cb_mask_views += 1;
                c->cb_mask_boxes += (r.cnt && vol == r.cnt) ? 1 : 0;
                if (getenv("MVS_CB_DEBUG")) {
                    const CbBox& Bd = boxes[(size_t)(&r - hrec.data())];
                    fprintf(stderr, "[cb mask] view box lo %d %d %d n %d %d %d: valid %llu, bbox %d..%d %d..%d %d..%d (volume %llu)\n", Bd.lo[0], Bd.lo[1], Bd.lo[2],
                            Bd.n[0], Bd.n[1], Bd.n[2], r.cnt, r.lo[0], r.hi[0], r.lo[1], r.hi[1], r.lo[2], r.hi[2], vol);
                }
            }
        }
    }

    auto gauss = [&](const float* src, float* dst, const CbBox& B, int radius, const double* fw) {
        // scipy filters axis 0, 1, 2 in turn; a 2D chunk has no z axis
        const Shape3 Sb = {B.n[0], B.n[1], B.n[2]};
        const long long bn = (long long)B.n[0] * B.n[1] * B.n[2];
        const int gbb = grid_for(bn);
        const float* cur = src;
        float* tmp[2] = {T1, T2};
        int pass = 0;
        for (int axis = 3 - ndim; axis < 3; ++axis, ++pass) {
            float* d = (axis == 2) ? dst : tmp[pass & 1];
            // LDS-staged lines when a useful tile of them fits into 64 KiB; else the tap-by-tap kernel
            GaussLines L;
            const long long nz = Sb.nz, ny = Sb.ny, nx = Sb.nx;
            if (axis == 2) { L.len = Sb.nx; L.stride = 1; L.n_lines = nz * ny; L.inner = 1; L.outer_stride = nx; }
            else if (axis == 1) { L.len = Sb.ny; L.stride = nx; L.n_lines = nz * nx; L.inner = nx; L.outer_stride = ny * nx; }
            else { L.len = Sb.nz; L.stride = ny * nx; L.n_lines = ny * nx; L.inner = ny * nx; L.outer_stride = 0; }
            L.b0 = B.lo[axis];
            L.full = (&S.nz)[axis];
            const int span = L.len + 2 * radius;
            int T = (axis == 2) ? 8 : 32;
            while (T > 1 && (size_t)span * (T + 1) * 4 > 60 * 1024) T >>= 1;
            if ((size_t)span * (T + 1) * 4 <= 60 * 1024 && (axis == 2 || T >= 8)) {
                L.T = T;
                const size_t lds = (size_t)span * (T + 1) * 4;
                const long long nb = (L.n_lines + T - 1) / T;
                hipLaunchKernelGGL(gauss1d_lds_kernel, dim3((unsigned)nb), dim3(256), lds, c->stream, cur, d, L, radius, fw, axis == 2 ? 1 : 0);
            } else {
                hipLaunchKernelGGL(gauss1d_kernel, dim3(gbb), dim3(256), 0, c->stream, cur, d, Sb, axis, radius, fw, L.b0, L.full);
            }
            cur = d;
        }
    };
    // ---- paired path: per view 2 x ndim launches, one view after the other on the context's stream.  (Round 4 first ran the
    // views' chains side by side on the side streams -- 27.0 ms per probe call against 27.5 serial; once the call stopped waiting
    // for its own work (below) the serial form took 23.4 ms and the forked one 32-33: consecutive chunks overlap on ONE stream by
    // themselves, and the fork / join events between low-priority side streams only got in their way.) ----
    if (paired) {
        std::vector<int> order;
        for (int v = 0; v < n_views; ++v)
            if ((long long)boxes[v].n[0] * boxes[v].n[1] * boxes[v].n[2] > 0) order.push_back(v);
        float* TP5 = A;      // 5 temporaries of the largest box, shared by the views
        for (size_t oi = 0; oi < order.size(); ++oi) {
            const int v = order[oi];
            const CbBox& B = boxes[v];
            hipStream_t st = c->stream;
            float* t[5];
            for (int k = 0; k < 5; ++k) t[k] = (float*)((char*)TP5 + (size_t)k * tmp_b);
            const float* Iv = I + B.off;
            const float* Bv = BW + B.off;
            for (int f = 0; f < 2; ++f) {
                const int radius = f ? r2 : r1;
                const double* fw = f ? dfw2 : dfw1;
                const float *ina = nullptr, *inb = nullptr;
                int pass = 0;
                for (int axis = 3 - ndim; axis < 3; ++axis, ++pass) {
                    const bool firstp = axis == 3 - ndim, lastp = axis == 2;
                    PairIO P;
                    P.im = Iv; P.bw = Bv;
                    P.a = firstp ? (f ? t[4] : nullptr) : ina;
                    P.b = firstp ? nullptr : inb;
                    P.src = firstp ? (f ? SRC_VMASK : SRC_PREP) : SRC_AB;
                    P.dst = lastp ? (f ? DST_F : DST_SQ) : DST_AB;
                    P.oa = lastp ? (f ? F + B.off : t[4]) : t[2 * (pass & 1)];
                    P.ob = lastp ? nullptr : t[2 * (pass & 1) + 1];
                    P.rec = mask_tables ? drecs + v : nullptr;
                    P.mtab = mask_tables ? dtables + table_off[(size_t)v * 2 + f] : nullptr;
                    GaussLines L;
                    const long long nz = B.n[0], ny = B.n[1], nx = B.n[2];
                    if (axis == 2) { L.len = B.n[2]; L.stride = 1; L.n_lines = nz * ny; L.inner = 1; L.outer_stride = nx; }
                    else if (axis == 1) { L.len = B.n[1]; L.stride = nx; L.n_lines = nz * nx; L.inner = nx; L.outer_stride = ny * nx; }
                    else { L.len = B.n[0]; L.stride = ny * nx; L.n_lines = ny * nx; L.inner = ny * nx; L.outer_stride = 0; }
                    L.b0 = B.lo[axis];
                    L.full = (&S.nz)[axis];
                    // passes along y / z that only hand both quantities on: one quantity per workgroup, twice the lines (128-byte pieces)
                    const int Ts = (P.dst == DST_AB && axis != 2 && !c->cb_nosplit) ? split_T(L.len, radius) : 0;
                    const bool split = Ts > 0;
                    L.T = split ? Ts : pair_T(L.len, radius, axis);
                    const size_t lds = (size_t)(L.len + 2 * radius + kGaussK) * (L.T + 1) * (split ? 4 : 8);
                    const long long nb = (L.n_lines + L.T - 1) / L.T;
                    const dim3 g((unsigned)nb, split ? 2 : 1), b(256);
                    const int pf = axis == 2 ? 1 : 0;
#define MVS_PAIR(S_, D_, SP_) hipLaunchKernelGGL((gauss1d_pair_kernel<S_, D_, SP_>), g, b, lds, st, P, L, radius, fw, pf)
                    if (split) {
                        if (P.src == SRC_PREP) MVS_PAIR(SRC_PREP, DST_AB, true);
                        else if (P.src == SRC_VMASK) MVS_PAIR(SRC_VMASK, DST_AB, true);
                        else MVS_PAIR(SRC_AB, DST_AB, true);
                    }
                    else if (P.src == SRC_PREP && P.dst == DST_AB) MVS_PAIR(SRC_PREP, DST_AB, false);
                    else if (P.src == SRC_PREP && P.dst == DST_SQ) MVS_PAIR(SRC_PREP, DST_SQ, false);
                    else if (P.src == SRC_VMASK && P.dst == DST_AB) MVS_PAIR(SRC_VMASK, DST_AB, false);
                    else if (P.src == SRC_VMASK && P.dst == DST_F) MVS_PAIR(SRC_VMASK, DST_F, false);
                    else if (P.src == SRC_AB && P.dst == DST_AB) MVS_PAIR(SRC_AB, DST_AB, false);
                    else if (P.src == SRC_AB && P.dst == DST_SQ) MVS_PAIR(SRC_AB, DST_SQ, false);
                    else MVS_PAIR(SRC_AB, DST_F, false);
#undef MVS_PAIR
                    ina = P.oa; inb = P.ob;
                }
            }
        }
        MVS_HIP_TRY(c, hipGetLastError());
    }
    for (int v = 0; v < n_views && !paired; ++v) {
        const CbBox& B = boxes[v];
        const long long bn = (long long)B.n[0] * B.n[1] * B.n[2];
        if (bn == 0) continue;
        const int gbb = grid_for(bn);
        const float* Iv = I + B.off;
        const float* Bv = BW + B.off;
        float* Fv = F + B.off;
        hipLaunchKernelGGL(prep_kernel, dim3(gbb), dim3(256), 0, c->stream, Iv, Bv, bn, A, V0, M);
        gauss(V0, T0, B, r1, dfw1);           // VV  (T0)
        gauss(M, Fv, B, r1, dfw1);            // WW  (Fv used as temporary)
        hipLaunchKernelGGL(ng_finish_sq_kernel, dim3(gbb), dim3(256), 0, c->stream, T0, Fv, A, bn, V0);   // V0 <- (A - Z)^2, NaN -> 0
        gauss(V0, T0, B, r2, dfw2);           // VV2 (T0)
        gauss(M, V0, B, r2, dfw2);            // WW2 (V0)
        hipLaunchKernelGGL(ng_finish_kernel, dim3(gbb), dim3(256), 0, c->stream, T0, V0, A, bn, Fv);
    }
    MVS_HIP_TRY(c, hipGetLastError());

    const size_t out_bytes = (size_t)no * es;
    void* dout = out;
    if (opts->out_mem == MVS_MEM_HOST) {
        dout = mvs_scratch(c, 1, out_bytes);
        if (!dout) return mvs_alloc_failed(c);
    }
    const int gbo = grid_for(no);
    const int tz = (int)opts->trim[0], ty = (int)opts->trim[1], tx = (int)opts->trim[2];
    if (small) {
        switch (dtype) {
            case MVS_U8: hipLaunchKernelGGL(cb_fuse8_kernel<unsigned char>, dim3(gbo), dim3(256), 0, c->stream, I, BW, F, bx8, n_views, tz, ty, tx, O, (unsigned char*)dout); break;
            case MVS_U16: hipLaunchKernelGGL(cb_fuse8_kernel<unsigned short>, dim3(gbo), dim3(256), 0, c->stream, I, BW, F, bx8, n_views, tz, ty, tx, O, (unsigned short*)dout); break;
            default: hipLaunchKernelGGL(cb_fuse8_kernel<float>, dim3(gbo), dim3(256), 0, c->stream, I, BW, F, bx8, n_views, tz, ty, tx, O, (float*)dout); break;
        }
    } else
    switch (dtype) {
        case MVS_U8: hipLaunchKernelGGL(cb_fuse_kernel<unsigned char>, dim3(gbo), dim3(256), 0, c->stream, I, BW, F, dboxes, n_views, tz, ty, tx, O, (unsigned char*)dout); break;
        case MVS_U16: hipLaunchKernelGGL(cb_fuse_kernel<unsigned short>, dim3(gbo), dim3(256), 0, c->stream, I, BW, F, dboxes, n_views, tz, ty, tx, O, (unsigned short*)dout); break;
        default: hipLaunchKernelGGL(cb_fuse_kernel<float>, dim3(gbo), dim3(256), 0, c->stream, I, BW, F, dboxes, n_views, tz, ty, tx, O, (float*)dout); break;
    }
    MVS_HIP_TRY(c, hipGetLastError());
    MVS_HIP_TRY(c, hipEventRecord(c->ev_stop, c->stream));
    c->timing_valid = true;
    if (opts->out_mem == MVS_MEM_HOST) {
        MVS_HIP_TRY(c, hipMemcpyAsync(out, dout, out_bytes, hipMemcpyDeviceToHost, c->stream));
        MVS_HIP_TRY(c, hipStreamSynchronize(c->stream));
    } else if (host_bytes) {
        MVS_HIP_TRY(c, hipStreamSynchronize(c->stream));      // host slabs were staged through scratch that the next call may overwrite
    }
    return MVS_OK;
}

// ---- the fast path of one chunk (kernels: mvs_gauss_fast.inc).  *taken = false: this chunk is not its kind (more than 8 views, a
// rotated / scaled view, a chunk axis shorter than a filter radius, a line that does not fit LDS, or -- host results only, where the
// call waits anyway -- a mask list that overflowed) and the caller runs the bit-faithful passes. ----
static int cb_fast_chunk(MvsContext* c, const mvs_view_t* views, int32_t n_views, const mvs_fuse_opts_t* opts, void* out, bool* taken) {
    *taken = false;
    if (n_views > 8 || n_views < 1) return MVS_OK;
    const int dtype = views[0].dtype;
    const size_t es = mvs_dtype_size(dtype);
    const int64_t* cs = opts->out_shape;
    const Shape3 S = {(int)cs[0], (int)cs[1], (int)cs[2]};
    const long long n = (long long)S.nz * S.ny * S.nx;
    const int ndim = opts->ndim;
    int64_t os[3];
    for (int k = 0; k < 3; ++k) os[k] = cs[k] - 2 * opts->trim[k];
    const Shape3 O = {(int)os[0], (int)os[1], (int)os[2]};
    const long long no = (long long)O.nz * O.ny * O.nx;
    int r1, r2;
    std::vector<double> w1, w2;
    gaussian_kernel((double)opts->sigma_1, &r1, &w1);
    gaussian_kernel((double)opts->sigma_2, &r2, &w2);
    for (int axis = 3 - ndim; axis < 3; ++axis)
        if (cs[axis] < std::max(r1, r2)) return MVS_OK;      // (further images of a voxel under the reflection would be in reach)
    static const double I9[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
    for (int i = 0; i < n_views; ++i)
        for (int k = 0; k < 9; ++k)
            if (views[i].matrix[k] != I9[k]) return MVS_OK;
    if (std::max(r1, r2) > kCbMaxRadius) return MVS_OK;

    // ---- device views and their boxes inside the chunk ----
    std::vector<DevView> dvs((size_t)n_views);
    size_t host_bytes = 0;
    for (int i = 0; i < n_views; ++i)
        if (views[i].mem == MVS_MEM_HOST) {
            if (views[i].stride[1] != views[i].shape[2] || views[i].stride[0] != views[i].shape[1] * views[i].shape[2])
                return mvs_fail(c, MVS_ERR_UNSUPPORTED, "host slabs must be C-contiguous");
            host_bytes += ((size_t)views[i].shape[0] * views[i].shape[1] * views[i].shape[2] * es + 255) / 256 * 256;
        }
    CbFastViews VS;
    memset(&VS, 0, sizeof(VS));
    VS.nv = n_views;
    long long pool = 0, total_rows = 0, tab_doubles = 0;
    std::vector<TrView> trv((size_t)n_views);
    bool tr_all = !c->cb_blend_generic;
    {
        for (int i = 0; i < n_views; ++i) {
            // (slab pointers of host views are filled in below, once the scratch exists; the geometry does not depend on them)
            int rc = mvs_fill_dev_view(c, views[i], ndim, views[i].data, &dvs[i]);
            if (rc) return rc;
            // (also moves the view into the chunk's frame; a view without the closed-form blend weight: the generic blend launch)
            if (!mvs_prepare_tr_view(&dvs[i], opts->order, cs, es, opts->index_origin, views[i].index_offset, &trv[i])) tr_all = false;
            int lo[3], hi[3];
            mvs_view_chunk_box(dvs[i], cs, lo, hi);
            CbFastView& B = VS.v[i];
            long long bv = 1;
            for (int k = 0; k < 3; ++k) {
                B.lo[k] = lo[k];
                B.n[k] = std::max(hi[k] - lo[k] + 1, 0);
                bv *= B.n[k];
            }
            if (bv == 0) { B.n[0] = B.n[1] = B.n[2] = 0; }
            if (pool + bv >= (1ll << 31)) return MVS_OK;
            B.off = (int)pool;
            pool += (bv + 63) / 64 * 64;
            B.row0 = (int)total_rows;
            total_rows += (long long)B.n[0] * B.n[1];
            B.tab0 = (int)tab_doubles;
            tab_doubles += 2ll * (B.n[0] + B.n[1] + B.n[2]);
        }
    }
    // lines per workgroup of every (axis, filter, view): 0 = a line does not fit -> not this path
    int Tsel[3][2][8];
    int xt_lo = 8, xt_hi = 32, dbg = 0;
    if (const char* e = getenv("MVS_CBF_XT")) { xt_lo = xt_hi = atoi(e); }
    if (const char* e = getenv("MVS_CBF_DBG")) dbg = atoi(e);
    for (int axis = 3 - ndim; axis < 3; ++axis)
        for (int f = 0; f < 2; ++f)
            for (int i = 0; i < n_views; ++i) {
                const int len = VS.v[i].n[axis];
                Tsel[axis][f][i] = len > 0 ? cb_fast_T(len, f ? r2 : r1, axis == 2 ? xt_lo : 32, axis == 2 ? xt_hi : 64, axis == 2) : 8;
                if (!Tsel[axis][f][i]) return MVS_OK;
            }
    char* slab_base = nullptr;
    if (host_bytes) {
        slab_base = (char*)mvs_scratch(c, 0, host_bytes);
        if (!slab_base) return mvs_alloc_failed(c);
    }
    if (!c->cb_flag_host) {
        MVS_HIP_TRY(c, hipHostMalloc((void**)&c->cb_flag_host, 64, hipHostMallocMapped));
        MVS_HIP_TRY(c, hipHostGetDevicePointer((void**)&c->cb_flag_dev, c->cb_flag_host, 0));
        *c->cb_flag_host = 0;
    }
    MVS_HIP_TRY(c, hipEventRecord(c->ev_start, c->stream));
    {
        size_t cursor = 0;
        for (int i = 0; i < n_views; ++i)
            if (views[i].mem == MVS_MEM_HOST) {
                const size_t nb = (size_t)views[i].shape[0] * views[i].shape[1] * views[i].shape[2] * es;
                MVS_HIP_TRY(c, hipMemcpyAsync(slab_base + cursor, views[i].data, nb, hipMemcpyHostToDevice, c->stream));
                dvs[i].data = slab_base + cursor;
                cursor += (nb + 255) / 256 * 256;
            }
    }

    // ---- scratch (slot 6): pools I, BW, F, T0; row records; [uploaded block: weights f64 / f32, view records, mask records]; lists; tables ----
    const size_t pool_b = ((size_t)pool * 4 + 255) / 256 * 256;
    const size_t rows_b = ((size_t)total_rows * 16 + 255) / 256 * 256;
    const size_t nw = w1.size() + w2.size();
    const size_t wdb = (nw * 8 + 255) / 256 * 256, wfb = (nw * 4 + 255) / 256 * 256, vb = ((size_t)n_views * sizeof(DevView) + 255) / 256 * 256,
                 rb = ((size_t)n_views * sizeof(CbFastRec) + 255) / 256 * 256;
    const size_t up_b = wdb + wfb + vb + rb;
    const size_t miss_b = (size_t)n_views * kCbMissCap * 16, tab_b = ((size_t)tab_doubles * 8 + 255) / 256 * 256;
    long long max_rows = 1;
    for (int i = 0; i < n_views; ++i) max_rows = std::max(max_rows, (long long)VS.v[i].n[0] * VS.v[i].n[1]);
    const unsigned rows_grid = (unsigned)std::min<long long>((max_rows + 15) / 16, 1024);      // a workgroup: 4 wavefronts x 4 rows per sweep
    const size_t part_b = ((size_t)n_views * rows_grid * sizeof(CbPartial) + 255) / 256 * 256;
    const size_t need = 4 * pool_b + rows_b + up_b + miss_b + tab_b + part_b + 4096;
    char* base = (char*)mvs_scratch(c, 6, need);
    if (!base) return mvs_alloc_failed(c);
    float* I = (float*)base;
    float* BW = (float*)(base + pool_b);
    float* F = (float*)(base + 2 * pool_b);
    float* T0 = (float*)(base + 3 * pool_b);
    int4* rows = (int4*)(base + 4 * pool_b);
    char* dblock = base + 4 * pool_b + rows_b;
    double* dfw1 = (double*)dblock;
    double* dfw2 = dfw1 + w1.size();
    float* ffw1 = (float*)(dblock + wdb);
    float* ffw2 = ffw1 + w1.size();
    DevView* dviews_dev = (DevView*)(dblock + wdb + wfb);
    CbFastRec* drecs = (CbFastRec*)(dblock + wdb + wfb + vb);
    int4* dmiss = (int4*)(dblock + up_b);
    double* dtabs = (double*)(dblock + up_b + miss_b);
    CbPartial* dpart = (CbPartial*)(dblock + up_b + miss_b + tab_b);
    {
        char* hp = (char*)mvs_pinned_slot(c, 0, up_b + 64);
        if (!hp) return mvs_alloc_failed(c);
        memcpy(hp, w1.data(), w1.size() * 8);
        memcpy(hp + w1.size() * 8, w2.data(), w2.size() * 8);
        float* hf = (float*)(hp + wdb);
        for (size_t k = 0; k < w1.size(); ++k) hf[k] = (float)w1[k];
        for (size_t k = 0; k < w2.size(); ++k) hf[w1.size() + k] = (float)w2[k];
        memcpy(hp + wdb + wfb, &dvs[0], (size_t)n_views * sizeof(DevView));
        for (int i = 0; i < n_views; ++i) {
            CbFastRec r;
            memset(&r, 0, sizeof(r));
            for (int k = 0; k < 3; ++k) { r.lo[k] = 0x7fffffff; r.hi[k] = -1; }
            memcpy(hp + wdb + wfb + vb + (size_t)i * sizeof(CbFastRec), &r, sizeof(r));
        }
        { const int rcu = mvs_upload_small(c, dblock, hp, up_b); if (rcu) return rcu; }
        mvs_pinned_mark(c, 0);
    }

    {   // resampled views and blend weights on the views' boxes: two launches for all views of the chunk
        std::vector<float*> res_out(n_views), blend_out(n_views);
        std::vector<int64_t> shp((size_t)n_views * 3);
        std::vector<int> b0((size_t)n_views * 3);
        for (int i = 0; i < n_views; ++i) {
            const CbFastView& B = VS.v[i];
            res_out[i] = I + B.off;
            blend_out[i] = BW + B.off;
            for (int k = 0; k < 3; ++k) { shp[(size_t)i * 3 + k] = B.n[k]; b0[(size_t)i * 3 + k] = B.lo[k]; }
        }
        mvs_launch_boxes_batch(c, &dvs[0], dviews_dev, n_views, dtype, opts->order, NAN, res_out.data(), blend_out.data(),
                               (const int64_t (*)[3])shp.data(), (const int (*)[3])b0.data(), tr_all ? trv.data() : nullptr);
    }
    CbBoxes8 bx8;
    memset(&bx8, 0, sizeof(bx8));
    for (int i = 0; i < n_views; ++i) {
        for (int k = 0; k < 3; ++k) { bx8.b[i].lo[k] = VS.v[i].lo[k]; bx8.b[i].n[k] = VS.v[i].n[k]; }
        bx8.b[i].off = VS.v[i].off;
    }
    hipLaunchKernelGGL(cb_normalize8_runs_kernel, dim3(grid_for((n + kCbRun - 1) / kCbRun)), dim3(256), 0, c->stream, BW, I, bx8, n_views, S);
    // ---- the valid mask of every view: bounding box + listed voxels; tables of the box under both filters ----
    hipLaunchKernelGGL(cb_rows_kernel, dim3(rows_grid, n_views), dim3(256), 0, c->stream, I, VS, rows, dpart);
    hipLaunchKernelGGL(cb_rec_reduce_kernel, dim3(n_views), dim3(256), 0, c->stream, dpart, (int)rows_grid, drecs);
    hipLaunchKernelGGL(cb_missing_kernel, dim3((unsigned)std::min<long long>((max_rows + 255) / 256, 1024), n_views), dim3(256), 0, c->stream, I, VS, rows, drecs, dmiss,
                       c->cb_flag_dev);
    hipLaunchKernelGGL(cb_sort_missing_kernel, dim3(n_views), dim3(kCbMissCap), 0, c->stream, drecs, dmiss);
    hipLaunchKernelGGL(cb_tables_kernel, dim3(3, n_views, 2), dim3(256), 0, c->stream, VS, drecs, S, ndim, r1, dfw1, r2, dfw2, dtabs);
    MVS_HIP_TRY(c, hipGetLastError());
    if (c->cb_mask_count) {      // test / profiling switch: the views' records (how many voxels their masks lack inside the bounding box)
        std::vector<CbFastRec> hrec((size_t)n_views);
        MVS_HIP_TRY(c, hipMemcpyAsync(hrec.data(), drecs, (size_t)n_views * sizeof(CbFastRec), hipMemcpyDeviceToHost, c->stream));
        MVS_HIP_TRY(c, hipStreamSynchronize(c->stream));
        for (int i = 0; i < n_views; ++i) {
            c->cb_mask_views += 1;
            c->cb_mask_boxes += hrec[i].nmiss;      // (fast path: the sum of the list lengths)
            if (getenv("MVS_CB_DEBUG"))
                fprintf(stderr, "[cb fast] view box lo %d %d %d n %d %d %d: valid %llu, bbox %d..%d %d..%d %d..%d, listed %d\n", VS.v[i].lo[0], VS.v[i].lo[1],
                        VS.v[i].lo[2], VS.v[i].n[0], VS.v[i].n[1], VS.v[i].n[2], hrec[i].cnt, hrec[i].lo[0], hrec[i].hi[0], hrec[i].lo[1], hrec[i].hi[1],
                        hrec[i].lo[2], hrec[i].hi[2], hrec[i].nmiss);
        }
    }

    // ---- 2 * ndim line passes, each ONE launch over all views: I -> T0 -> F -> (squared deviation) T0 -> F -> T0 -> F (3D) ----
    int pass = 0;
    for (int f = 0; f < 2; ++f)
        for (int axis = 3 - ndim; axis < 3; ++axis, ++pass) {
            // accumulators (option cb_taps_f64): 1 (default) = float64 for both filters, 0 = float32 for both, 2 / 3 = float64 for the
            // first / second filter only.  C3 at size, one-count flips of the fused uint16 voxels against the oracle (all of them at
            // truncation boundaries, none beyond the 1e-4 bar): 0.04 % (1; the bit-faithful passes: 0.04 %), 0.10 % (3), 0.21 % (2),
            // 0.22 % (0) -- the noise of float32 taps enters through the second filter, whose result IS the weight; the probe takes
            // 11.8 ms with 0 and 12.4 ms with 1 (the passes are bound by memory and latency, not by the taps)
            const bool f64 = c->cb_taps == 1 || (c->cb_taps == 2 && f == 0) || (c->cb_taps == 3 && f == 1);
            const bool firstp = axis == 3 - ndim, lastp = axis == 2;
            CbLineArgs A;
            A.src = (pass == 0) ? I : ((pass & 1) ? T0 : F);
            A.dst = (pass & 1) ? F : T0;
            A.I = I;
            A.axis = axis; A.radius = f ? r2 : r1; A.ndim = ndim; A.filt = f;
            A.S = S;
            A.fwf = f ? ffw2 : ffw1; A.fwd = f ? dfw2 : dfw1;
            A.tabs = dtabs; A.recs = drecs; A.miss = dmiss; A.dbg = dbg;
            CbFastViews B = VS;
            int nb = 0;
            size_t lds = 0;
            for (int i = 0; i < n_views; ++i) {
                CbFastView& V = B.v[i];
                V.T = Tsel[axis][f][i];
                V.blk0 = nb;
                V.zr0 = 0; V.nzr = V.n[0]; V.yr0 = 0; V.nyr = V.n[1];
                long long bn = (long long)V.n[0] * V.n[1] * V.n[2];
                // the weight F of a view is read on the TRIMMED chunk only: a view that does not reach it (a sliver of a neighbour in the
                // halo) is not filtered at all, and the last pass works on the rows inside it
                int tl[3], th[3];
                bool reaches = bn > 0;
                for (int k = 0; k < 3; ++k) {
                    tl[k] = std::max(V.lo[k], (int)opts->trim[k]) - V.lo[k];
                    th[k] = std::min(V.lo[k] + V.n[k], (int)(cs[k] - opts->trim[k])) - V.lo[k];
                    if (th[k] <= tl[k]) reaches = false;
                }
                if (!reaches) { V.nzr = V.nyr = 0; continue; }
                if (lastp && f == 1) {
                    V.zr0 = tl[0]; V.nzr = th[0] - tl[0]; V.yr0 = tl[1]; V.nyr = th[1] - tl[1];
                    bn = (long long)V.nzr * V.nyr * V.n[2];
                }
                const long long n_lines = bn / V.n[axis];
                nb += (int)((n_lines + V.T - 1) / V.T);
                lds = std::max(lds, cb_fast_lds(V.n[axis], A.radius, V.T, axis == 2));
            }
            if (nb == 0) continue;
            const int src = (pass == 0) ? CBS_NAN0 : CBS_PLAIN;
            const int dst = lastp ? (f ? CBD_F : CBD_SQ) : CBD_PLAIN;
            (void)firstp;
#define MVS_CBL(S_, D_, PF_) do { if (f64) hipLaunchKernelGGL((cb_line_kernel<S_, D_, double, PF_>), dim3(nb), dim3(256), lds, c->stream, A, B); \
                                  else hipLaunchKernelGGL((cb_line_kernel<S_, D_, float, PF_>), dim3(nb), dim3(256), lds, c->stream, A, B); } while (0)
            if (!lastp) { if (src == CBS_NAN0) MVS_CBL(CBS_NAN0, CBD_PLAIN, false); else MVS_CBL(CBS_PLAIN, CBD_PLAIN, false); }
            else if (dst == CBD_SQ) MVS_CBL(CBS_PLAIN, CBD_SQ, true);
            else MVS_CBL(CBS_PLAIN, CBD_F, true);
#undef MVS_CBL
            c->cb_line_launches += 1;
        }
    MVS_HIP_TRY(c, hipGetLastError());

    const size_t out_bytes = (size_t)no * es;
    void* dout = out;
    if (opts->out_mem == MVS_MEM_HOST) {
        dout = mvs_scratch(c, 1, out_bytes);
        if (!dout) return mvs_alloc_failed(c);
    }
    const int gbo = grid_for((no + kCbRun - 1) / kCbRun);
    const int tz = (int)opts->trim[0], ty = (int)opts->trim[1], tx = (int)opts->trim[2];
    switch (dtype) {
        case MVS_U8: hipLaunchKernelGGL(cb_fuse8_runs_kernel<unsigned char>, dim3(gbo), dim3(256), 0, c->stream, I, BW, F, bx8, n_views, tz, ty, tx, O, (unsigned char*)dout); break;
        case MVS_U16: hipLaunchKernelGGL(cb_fuse8_runs_kernel<unsigned short>, dim3(gbo), dim3(256), 0, c->stream, I, BW, F, bx8, n_views, tz, ty, tx, O, (unsigned short*)dout); break;
        default: hipLaunchKernelGGL(cb_fuse8_runs_kernel<float>, dim3(gbo), dim3(256), 0, c->stream, I, BW, F, bx8, n_views, tz, ty, tx, O, (float*)dout); break;
    }
    MVS_HIP_TRY(c, hipGetLastError());
    MVS_HIP_TRY(c, hipEventRecord(c->ev_stop, c->stream));
    c->timing_valid = true;
    *taken = true;
    if (opts->out_mem == MVS_MEM_HOST) {
        MVS_HIP_TRY(c, hipMemcpyAsync(out, dout, out_bytes, hipMemcpyDeviceToHost, c->stream));
        MVS_HIP_TRY(c, hipStreamSynchronize(c->stream));
        if (*c->cb_flag_host) {      // a mask list overflowed: this chunk again, through the bit-faithful passes (the caller falls through)
            *c->cb_flag_host = 0;
            c->cb_overflows += 1;
            *taken = false;
        }
    } else if (host_bytes) {
        MVS_HIP_TRY(c, hipStreamSynchronize(c->stream));      // host slabs were staged through scratch that the next call may overwrite
    }
    return MVS_OK;
}
