// mvs_score.hip -- candidate scoring of the pairwise registration (gfx950).
//
// mvs_score_candidates == the loop of the reference's registration.phase_correlation_registration
// (src/multiview_stitcher/registration.py:493-556), for n translation candidates t:
//   im1t = scipy.ndimage.affine_transform(im1, translate(t), order=1, mode="constant", cval=NaN)
//   mask = ~isnan(im1t) & ~isnan(im0);  skip (-1,-1) if empty or < 10 % of im1's valid voxels
//   region = union / intersection of the valid bounding boxes of im0 and im1t
//   `continue` if nanmax(im1t[region]) <= im1_min                                   (Q3)
//   SSIM(nan_to_num(im0[region]), nan_to_num(im1t[region]), data_range, win)        (skimage, float32)
//   quality = spearmanr(im0[mask], im1t[mask] - 1)                                  (scipy.stats)
// Everything voxel-sized runs on the device; the host only sees counters, bounding boxes and sums.
#include "mvs_internal.h"

#include <cstring>

#include <rocprim/rocprim.hpp>

#include <cmath>
#include <cstdlib>
#include <vector>

int mvs_stage_float_volume(MvsContext* c, const float* src, int32_t mem, long long n, int slot, float** dptr);   // mvs_reg.hip
int mvs_device_nanminmax(MvsContext* c, const float* d_in, long long n, float* mn, float* mx, long long* nvalid);   // mvs_reg.hip

namespace {

constexpr int kStatBlocks = 1024;    // workgroups (= partial results per candidate) of the statistics kernels
constexpr int kMaxResident = 16;     // shifted copies of the moving image kept per batch of candidates
constexpr int kChunk = 8;            // outputs one thread produces along the filtered axis

inline int grid_for(long long n) { return (int)std::min<long long>((n + 255) / 256, 256 * 8); }

struct Shape3 { int nz, ny, nx; };

// #valid voxels + bounding box of the valid voxels (min z,y,x, max z,y,x)
struct VoxStats { unsigned long long cnt; int bb[6]; };
// nanmax / has-NaN of the moving image over the SSIM region, and the sum of the SSIM map over its cropped interior
struct RegionStats { float mx; int hasnan; double ssim_sum; };

__device__ __forceinline__ void block_reduce_voxstats(unsigned long long cnt, int mnz, int mny, int mnx, int mxz, int mxy, int mxx,
                                                      VoxStats* __restrict__ out) {
    for (int off = 32; off > 0; off >>= 1) {
        cnt += __shfl_down(cnt, off);
        mnz = min(mnz, __shfl_down(mnz, off)); mny = min(mny, __shfl_down(mny, off)); mnx = min(mnx, __shfl_down(mnx, off));
        mxz = max(mxz, __shfl_down(mxz, off)); mxy = max(mxy, __shfl_down(mxy, off)); mxx = max(mxx, __shfl_down(mxx, off));
    }
    __shared__ unsigned long long s_cnt[4];
    __shared__ int s_bb[4][6];
    const int wave = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) {
        s_cnt[wave] = cnt;
        s_bb[wave][0] = mnz; s_bb[wave][1] = mny; s_bb[wave][2] = mnx; s_bb[wave][3] = mxz; s_bb[wave][4] = mxy; s_bb[wave][5] = mxx;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < 4; ++w) {
            cnt += s_cnt[w];
            mnz = min(mnz, s_bb[w][0]); mny = min(mny, s_bb[w][1]); mnx = min(mnx, s_bb[w][2]);
            mxz = max(mxz, s_bb[w][3]); mxy = max(mxy, s_bb[w][4]); mxx = max(mxx, s_bb[w][5]);
        }
        out->cnt = cnt;
        out->bb[0] = mnz; out->bb[1] = mny; out->bb[2] = mnx; out->bb[3] = mxz; out->bb[4] = mxy; out->bb[5] = mxx;
    }
}

// second stage: one workgroup per statistics record folds the kStatBlocks partials (no atomics anywhere: the
// same-address atomics of a one-stage reduction cost more than the voxel work of these kernels)
__global__ __launch_bounds__(256) void finish_voxstats_kernel(const VoxStats* __restrict__ partial, VoxStats* __restrict__ out) {
    const VoxStats* p = partial + (size_t)blockIdx.x * kStatBlocks;
    unsigned long long cnt = 0;
    int mnz = 0x7fffffff, mny = 0x7fffffff, mnx = 0x7fffffff, mxz = -1, mxy = -1, mxx = -1;
    for (int i = threadIdx.x; i < kStatBlocks; i += 256) {
        cnt += p[i].cnt;
        mnz = min(mnz, p[i].bb[0]); mny = min(mny, p[i].bb[1]); mnx = min(mnx, p[i].bb[2]);
        mxz = max(mxz, p[i].bb[3]); mxy = max(mxy, p[i].bb[4]); mxx = max(mxx, p[i].bb[5]);
    }
    block_reduce_voxstats(cnt, mnz, mny, mnx, mxz, mxy, mxx, out + blockIdx.x);
}

// ---- shifted copy of im1 with scipy's affine_transform semantics (order 1, cval NaN) ------------
__device__ __forceinline__ int tap2(int i0, int n) {
    int i1 = i0 + 1;
    if (i1 >= n) i1 = (n > 1) ? n - 2 : 0;   // mirrored edge offset, weight 0 there
    return i1;
}

// One axis of scipy's order-1 coordinate: c = pos + t (the identity-matrix row, same rounding as scipy's loop), valid iff
// 0 <= c <= n - 1, taps floor(c) and its neighbour (mirrored at the edge, weight 0 there), weight w = c - floor(c).
struct AxisTap { int ok, i0, i1; double w; };
__device__ __forceinline__ AxisTap axis_tap(int pos, double t, int n) {
    AxisTap a;
    const double c = (double)pos + t;
    a.ok = !(c < 0.0 || c > (double)(n - 1));
    const double f = floor(c);
    a.i0 = (int)f;
    a.w = c - f;
    a.i1 = a.ok ? tap2(a.i0, n) : 0;
    return a;
}

// Value of the shifted moving image at a voxel whose three axis parts are given (NaN outside).  scipy accumulates
// coeff * wz * wy * wx over the taps in z-major order.  A tap with weight 0 only matters when it can be NaN or inf
// (0 * NaN = NaN): for an all-finite moving image (skip_zero_taps) the taps of axes with an integer shift are skipped --
// adding their +0.0 would not change the sum (1 / 2 / 4 taps instead of 8).
__device__ __forceinline__ float shifted_value(const float* __restrict__ im1, int sy, int sz, const AxisTap& Z, const AxisTap& Y,
                                               const AxisTap& X, int skip_zero_taps) {
    if (!(Z.ok && Y.ok && X.ok)) return NAN;
    const double wz = Z.w, wy = Y.w, wx = X.w;
    const int iz = Z.i0, iy = Y.i0, ix = X.i0, iz1 = Z.i1, iy1 = Y.i1, ix1 = X.i1;
    double acc = 0.0;
    if (skip_zero_taps) {
        const bool nz2 = wz != 0.0, ny2 = wy != 0.0, nx2 = wx != 0.0;
        if (!nz2 && !ny2 && !nx2) return im1[iz * sz + iy * sy + ix];      // one tap of weight 1 * 1 * 1: (float)((double)v * 1.0) == v
        for (int a = 0; a <= (nz2 ? 1 : 0); ++a)
            for (int b = 0; b <= (ny2 ? 1 : 0); ++b)
                for (int cidx = 0; cidx <= (nx2 ? 1 : 0); ++cidx)
                    acc += (double)im1[(a ? iz1 : iz) * sz + (b ? iy1 : iy) * sy + (cidx ? ix1 : ix)] * (a ? wz : 1.0 - wz) *
                           (b ? wy : 1.0 - wy) * (cidx ? wx : 1.0 - wx);
        return (float)acc;
    }
    acc += (double)im1[iz * sz + iy * sy + ix] * (1.0 - wz) * (1.0 - wy) * (1.0 - wx);
    acc += (double)im1[iz * sz + iy * sy + ix1] * (1.0 - wz) * (1.0 - wy) * wx;
    acc += (double)im1[iz * sz + iy1 * sy + ix] * (1.0 - wz) * wy * (1.0 - wx);
    acc += (double)im1[iz * sz + iy1 * sy + ix1] * (1.0 - wz) * wy * wx;
    acc += (double)im1[iz1 * sz + iy * sy + ix] * wz * (1.0 - wy) * (1.0 - wx);
    acc += (double)im1[iz1 * sz + iy * sy + ix1] * wz * (1.0 - wy) * wx;
    acc += (double)im1[iz1 * sz + iy1 * sy + ix] * wz * wy * (1.0 - wx);
    acc += (double)im1[iz1 * sz + iy1 * sy + ix1] * wz * wy * wx;
    return (float)acc;
}

// Stats gathered while shifting: #(valid im1t & valid im0) and the bbox of valid im1t.  Each thread takes 4
// consecutive voxels (one index decode, one 16-byte store); the z and y parts of the coordinate only change when the
// group wraps to the next row, so they are evaluated per row, the x part per voxel.
template <bool STATS>
__device__ __forceinline__ void shift_body(const float* __restrict__ im1, const float* __restrict__ im0,
                                                    float* __restrict__ out, Shape3 S, double tz, double ty, double tx,
                                                    int skip_zero_taps, VoxStats* __restrict__ partial) {
    const unsigned int n = (unsigned int)S.nz * S.ny * S.nx;
    const unsigned int ngroups = (n + 3) / 4;
    unsigned long long cnt = 0;
    int mnz = 0x7fffffff, mny = 0x7fffffff, mnx = 0x7fffffff, mxz = -1, mxy = -1, mxx = -1;
    const int sy = S.nx, sz = S.ny * S.nx;
    for (unsigned int g = blockIdx.x * blockDim.x + threadIdx.x; g < ngroups; g += gridDim.x * blockDim.x) {
        const unsigned int i0 = g * 4;
        int x = (int)(i0 % (unsigned int)S.nx);
        const unsigned int t = i0 / (unsigned int)S.nx;
        int y = (int)(t % (unsigned int)S.ny), z = (int)(t / (unsigned int)S.ny);
        float r4[4];
        int row_z = -1, row_y = -1;
        AxisTap Z = {0, 0, 0, 0.0}, Y = {0, 0, 0, 0.0};
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            float r = NAN;
            if (i0 + k < n) {
                if (z != row_z || y != row_y) {
                    row_z = z; row_y = y;
                    Z = axis_tap(z, tz, S.nz);
                    Y = axis_tap(y, ty, S.ny);
                }
                r = shifted_value(im1, sy, sz, Z, Y, axis_tap(x, tx, S.nx), skip_zero_taps);
                if (STATS && r == r) {
                    mnz = min(mnz, z); mny = min(mny, y); mnx = min(mnx, x);
                    mxz = max(mxz, z); mxy = max(mxy, y); mxx = max(mxx, x);
                    const float a = im0[i0 + k];
                    if (a == a) ++cnt;
                }
                if (++x == S.nx) { x = 0; if (++y == S.ny) { y = 0; ++z; } }
            }
            r4[k] = r;
        }
        if (i0 + 3 < n) *reinterpret_cast<float4*>(out + i0) = make_float4(r4[0], r4[1], r4[2], r4[3]);
        else
            for (int k = 0; k < 4; ++k) if (i0 + k < n) out[i0 + k] = r4[k];
    }
    if (STATS) block_reduce_voxstats(cnt, mnz, mny, mnx, mxz, mxy, mxx, partial + blockIdx.x);
}

__global__ __launch_bounds__(256) void shift_kernel(const float* __restrict__ im1, const float* __restrict__ im0,
                                                    float* __restrict__ out, Shape3 S, double tz, double ty, double tx,
                                                    int skip_zero_taps, VoxStats* __restrict__ partial) {
    shift_body<true>(im1, im0, out, S, tz, ty, tx, skip_zero_taps, partial);
}

// The same shifted copy for a finite moving image and a shift whose components are all multiples of 1/2 (3D phase
// correlation refines to 1/upsample_factor = 1/2 pixel): c = pos + t is exact, so floor(c) = pos + floor(t), the weight is 0
// or exactly 1/2 on every axis, every tap product is an exact scaling by a power of two and the validity test is an integer
// comparison -- the same taps in the same (z-major) order as shifted_value, without its per-voxel double-precision
// coordinate arithmetic.  fz / fy / fx = floor(t), hz / hy / hx = 1 where the component has the fraction 1/2.
struct HalfShift { int fz, fy, fx, hz, hy, hx; };
__device__ __forceinline__ void shift_half_body(const float* __restrict__ im1, float* __restrict__ out, Shape3 S, HalfShift T) {
    const unsigned int n = (unsigned int)S.nz * S.ny * S.nx;
    const unsigned int ngroups = (n + 3) / 4;
    const int sy = S.nx, sz = S.ny * S.nx;
    // valid iff 0 <= pos + t <= n - 1: pos + f >= 0 and pos + f + h/2 <= n - 1
    const double scale = 1.0 / (double)(1 << (T.hz + T.hy + T.hx));          // product of the tap weights: 1, 1/2, 1/4 or 1/8
    for (unsigned int g = blockIdx.x * blockDim.x + threadIdx.x; g < ngroups; g += gridDim.x * blockDim.x) {
        const unsigned int i0 = g * 4;
        int x = (int)(i0 % (unsigned int)S.nx);
        const unsigned int t = i0 / (unsigned int)S.nx;
        int y = (int)(t % (unsigned int)S.ny), z = (int)(t / (unsigned int)S.ny);
        float r4[4];
        {
            // the common case -- the 4 voxels lie in one row and all their taps inside the image: every tap row is one
            // 16-byte load (+ 1 element for a half-pixel x shift) instead of 4 x 2 scalar loads
            const int iz = z + T.fz, iy = y + T.fy, ix = x + T.fx;
            if (x + 3 < S.nx && iz >= 0 && iz + T.hz <= S.nz - 1 && iy >= 0 && iy + T.hy <= S.ny - 1 && ix >= 0 && ix + 3 + T.hx <= S.nx - 1 &&
                (T.hz | T.hy | T.hx)) {
                const float* p = im1 + iz * sz + iy * sy + ix;
                double acc[4] = {0.0, 0.0, 0.0, 0.0};
                for (int a = 0; a <= T.hz; ++a)
                    for (int b = 0; b <= T.hy; ++b) {
                        const float* q = p + a * sz + b * sy;
                        float v[5];
#pragma unroll
                        for (int j = 0; j < 4; ++j) v[j] = q[j];
                        v[4] = T.hx ? q[4] : 0.f;
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            acc[j] += (double)v[j] * scale;
                            if (T.hx) acc[j] += (double)v[j + 1] * scale;
                        }
                    }
                *reinterpret_cast<float4*>(out + i0) = make_float4((float)acc[0], (float)acc[1], (float)acc[2], (float)acc[3]);
                continue;
            }
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            float r = NAN;
            if (i0 + k < n) {
                const int iz = z + T.fz, iy = y + T.fy, ix = x + T.fx;
                const bool ok = iz >= 0 && iz + T.hz <= S.nz - 1 && iy >= 0 && iy + T.hy <= S.ny - 1 && ix >= 0 && ix + T.hx <= S.nx - 1;
                if (ok) {
                    const float* p = im1 + iz * sz + iy * sy + ix;
                    double acc = 0.0;
                    for (int a = 0; a <= T.hz; ++a)
                        for (int b = 0; b <= T.hy; ++b)
                            for (int c = 0; c <= T.hx; ++c) acc += (double)p[a * sz + b * sy + c] * scale;
                    r = (T.hz | T.hy | T.hx) ? (float)acc : p[0];
                }
                if (++x == S.nx) { x = 0; if (++y == S.ny) { y = 0; ++z; } }
            }
            r4[k] = r;
        }
        if (i0 + 3 < n) *reinterpret_cast<float4*>(out + i0) = make_float4(r4[0], r4[1], r4[2], r4[3]);
        else
            for (int k = 0; k < 4; ++k) if (i0 + k < n) out[i0 + k] = r4[k];
    }
}

// shifted copies of several candidates in one launch (blockIdx.y = candidate); no statistics (the caller knows the valid boxes)
struct ShiftCand { float* out; double tz, ty, tx; int half; HalfShift H; };
struct ShiftBatch { ShiftCand c[kMaxResident]; };
__global__ __launch_bounds__(256) void shift_batch_kernel(const float* __restrict__ im1, const float* __restrict__ im0, Shape3 S, ShiftBatch B,
                                                          int skip_zero_taps) {
    const ShiftCand& C = B.c[blockIdx.y];
    if (C.half && skip_zero_taps) shift_half_body(im1, C.out, S, C.H);
    else shift_body<false>(im1, im0, C.out, S, C.tz, C.ty, C.tx, skip_zero_taps, nullptr);
}

// #valid voxels and their bbox for one image (get_bb_from_nanmask, registration.py:482-489; valid_pixels1 :400)
__global__ __launch_bounds__(256) void image_stats_kernel(const float* __restrict__ im, Shape3 S, VoxStats* __restrict__ partial) {
    const unsigned int n = (unsigned int)S.nz * S.ny * S.nx;
    const unsigned int ngroups = (n + 3) / 4;
    unsigned long long cnt = 0;
    int mnz = 0x7fffffff, mny = 0x7fffffff, mnx = 0x7fffffff, mxz = -1, mxy = -1, mxx = -1;
    for (unsigned int g = blockIdx.x * blockDim.x + threadIdx.x; g < ngroups; g += gridDim.x * blockDim.x) {
        const unsigned int i0 = g * 4;
        int x = (int)(i0 % (unsigned int)S.nx);
        const unsigned int t = i0 / (unsigned int)S.nx;
        int y = (int)(t % (unsigned int)S.ny), z = (int)(t / (unsigned int)S.ny);
        for (int k = 0; k < 4 && i0 + k < n; ++k) {
            const float v = im[i0 + k];
            if (v == v) {
                ++cnt;
                if (fabsf(v) == INFINITY) cnt += 1ull << 32;   // #inf rides in the upper half (both counts stay < 2^31)
                mnz = min(mnz, z); mny = min(mny, y); mnx = min(mnx, x);
                mxz = max(mxz, z); mxy = max(mxy, y); mxx = max(mxx, x);
            }
            if (++x == S.nx) { x = 0; if (++y == S.ny) { y = 0; ++z; } }
        }
    }
    block_reduce_voxstats(cnt, mnz, mny, mnx, mxz, mxy, mxx, partial + blockIdx.x);
}

// ---- SSIM: skimage.metrics.structural_similarity on nan_to_num'd float32 crops ------------------------------
// uniform_filter (mode="reflect", size=win) runs axis by axis with a float32 result per pass and double
// accumulation; the five filtered quantities are x, y, x*x, y*y, x*y.  Pass 1 reads the region straight out of
// im0 / im1t (nan_to_num + products on the fly, and the region's nanmax / has-NaN flags needed by
// registration.py:530, 539), the last pass (x axis) feeds the SSIM formula and its interior sum directly, so
// the only arrays that touch memory are the 5 outputs of pass 1 and, in 3D, of pass 2.
__device__ __forceinline__ int reflect_index(int p, int len) {
    // scipy "reflect": d c b a | a b c d | d c b a  (period 2*len)
    if (len == 1) return 0;
    const int period = 2 * len;
    p %= period;
    if (p < 0) p += period;
    if (p >= len) p = period - 1 - p;
    return p;
}

// WIN-tap box means of NOUT consecutive positions of a line, inputs v[0 .. NOUT + WIN - 2]: the running sum of scipy's
// uniform_filter1d (ni_filters.c: tmp += line[l + size - 1] - line[l - 1]; out = tmp / size, double accumulator), restarted
// per thread.  2 double operations and one float->double conversion per output instead of WIN of each, and no double
// division (x * (1 / WIN) rounds to the same float32 as x / WIN except on exact ties of the final rounding): these
// kernels were bound by exactly that arithmetic, not by memory.
template <int WIN, int NOUT, typename ACC = double>
__device__ __forceinline__ void box_means(const float (&v)[NOUT + WIN - 1], float (&out)[NOUT]) {
    constexpr ACC inv = (ACC)(1.0 / (double)WIN);
    ACC d[NOUT + WIN - 1];
#pragma unroll
    for (int k = 0; k < NOUT + WIN - 1; ++k) d[k] = (ACC)v[k];
    ACC run = (ACC)0;
#pragma unroll
    for (int j = 0; j < WIN; ++j) run += d[j];
    out[0] = (float)(run * inv);
#pragma unroll
    for (int k = 1; k < NOUT; ++k) {
        run += d[k + WIN - 1] - d[k - 1];
        out[k] = (float)(run * inv);
    }
}

struct Five { const float* src[5]; float* dst[5]; };

// pass 1: filter along `axis` (0 = z, 1 = y) of the region [lo, lo + R) of im0 / im1t
// SHIFTED: `im1t` is the unshifted moving image and its shifted value is evaluated on the fly (same arithmetic as
// shift_kernel), so the candidates of an all-finite pair never materialise their shifted copy.
struct ShiftArg { double tz, ty, tx; int skip_zero_taps; };
// QSET: which of the five quantities are filtered and written -- 0: all (x, y, xx, yy, xy); 1: only x, xx (the fixed
// image's own terms: the same for every candidate whose region is the whole volume, computed once per pair); 2: only
// y, yy, xy (the candidate's terms when x, xx are shared).
template <int WIN, bool SHIFTED, int QSET = 0>
__device__ __forceinline__ void ssim_first_pass_body(const float* __restrict__ im0, const float* __restrict__ im1t, Shape3 S,
                                                              int lz, int ly, int lx, Shape3 R, int axis, Five P,
                                                              float* __restrict__ pmax, int* __restrict__ phasnan, ShiftArg T) {
    constexpr int H = WIN / 2, NL = kChunk + 2 * H;
    const int len = axis == 0 ? R.nz : R.ny;
    const int nchunks = (len + kChunk - 1) / kChunk;
    const unsigned int other = axis == 0 ? (unsigned int)R.ny : (unsigned int)R.nz;   // the non-filtered one of (z, y)
    const unsigned int items = (unsigned int)nchunks * other * (unsigned int)R.nx;
    const int src_stride = axis == 0 ? S.ny * S.nx : S.nx;
    const int dst_stride = axis == 0 ? R.ny * R.nx : R.nx;
    float mx = -INFINITY;
    int hn = 0;
    for (unsigned int w = blockIdx.x * blockDim.x + threadIdx.x; w < items; w += gridDim.x * blockDim.x) {
        const int x = (int)(w % (unsigned int)R.nx);
        const unsigned int t = w / (unsigned int)R.nx;
        int z, y, p0;
        if (axis == 0) { y = (int)(t % other); p0 = (int)(t / other) * kChunk; z = 0; }
        else { p0 = (int)(t % (unsigned int)nchunks) * kChunk; z = (int)(t / (unsigned int)nchunks); y = 0; }
        const int src_base = ((z + lz) * S.ny + (y + ly)) * S.nx + (x + lx);
        float va[NL], vb[NL];
        // SHIFTED: the axis parts that do not move along the filtered axis are fixed for this thread
        AxisTap TX = {0, 0, 0, 0.0}, TO = {0, 0, 0, 0.0};
        if (SHIFTED) {
            TX = axis_tap(x + lx, T.tx, S.nx);
            TO = axis == 0 ? axis_tap(y + ly, T.ty, S.ny) : axis_tap(z + lz, T.tz, S.nz);
        }
#pragma unroll
        for (int k = 0; k < NL; ++k) {
            const int p = reflect_index(p0 - H + k, len);
            float a = im0[src_base + p * src_stride], b;
            if (SHIFTED) {
                if (axis == 0) b = shifted_value(im1t, S.nx, S.ny * S.nx, axis_tap(lz + p, T.tz, S.nz), TO, TX, T.skip_zero_taps);
                else b = shifted_value(im1t, S.nx, S.ny * S.nx, TO, axis_tap(ly + p, T.ty, S.ny), TX, T.skip_zero_taps);
            } else {
                b = im1t[src_base + p * src_stride];
            }
            if (k >= H && k < H + kChunk && p0 + (k - H) < len) {   // this voxel is the centre of one output
                if (b == b) mx = fmaxf(mx, b); else hn = 1;
            }
            va[k] = (a != a) ? 0.f : a;
            vb[k] = (b != b) ? 0.f : b;
        }
        const int dst_base = (z * R.ny + y) * R.nx + x;
        float f[5][kChunk];
        constexpr bool kX = QSET != 2, kY = QSET != 1;
        if (kX) box_means<WIN, kChunk>(va, f[0]);
        if (kY) box_means<WIN, kChunk>(vb, f[1]);
        {
            float prod[NL];   // products in float32 like the reference's `im * im` on float32 arrays
            if (kX) {
#pragma unroll
                for (int k = 0; k < NL; ++k) prod[k] = va[k] * va[k];
                box_means<WIN, kChunk>(prod, f[2]);
            }
            if (kY) {
#pragma unroll
                for (int k = 0; k < NL; ++k) prod[k] = vb[k] * vb[k];
                box_means<WIN, kChunk>(prod, f[3]);
#pragma unroll
                for (int k = 0; k < NL; ++k) prod[k] = va[k] * vb[k];
                box_means<WIN, kChunk>(prod, f[4]);
            }
        }
#pragma unroll
        for (int k = 0; k < kChunk; ++k) {
            if (p0 + k >= len) break;
            const int o = dst_base + (p0 + k) * dst_stride;
            if (kX) { P.dst[0][o] = f[0][k]; P.dst[2][o] = f[2][k]; }
            if (kY) { P.dst[1][o] = f[1][k]; P.dst[3][o] = f[3][k]; P.dst[4][o] = f[4][k]; }
        }
    }
    for (int off = 32; off > 0; off >>= 1) {
        mx = fmaxf(mx, __shfl_down(mx, off));
        hn |= __shfl_down(hn, off);
    }
    __shared__ float s_mx[4];
    __shared__ int s_hn[4];
    if ((threadIdx.x & 63) == 0) { s_mx[threadIdx.x >> 6] = mx; s_hn[threadIdx.x >> 6] = hn; }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < 4; ++w) { mx = fmaxf(mx, s_mx[w]); hn |= s_hn[w]; }
        pmax[blockIdx.x] = mx;
        phasnan[blockIdx.x] = hn;
    }
}

template <int WIN, bool SHIFTED, int QSET = 0>
__global__ __launch_bounds__(256) void ssim_first_pass_kernel(const float* __restrict__ im0, const float* __restrict__ im1t, Shape3 S,
                                                              int lz, int ly, int lx, Shape3 R, int axis, Five P,
                                                              float* __restrict__ pmax, int* __restrict__ phasnan, ShiftArg T) {
    ssim_first_pass_body<WIN, SHIFTED, QSET>(im0, im1t, S, lz, ly, lx, R, axis, P, pmax, phasnan, T);
}

// All candidates of a batch in ONE launch (blockIdx.y = candidate): the z pass of the candidate's terms (QSET 2) over the
// whole volume.  A step of the north-star mosaic issues ~13 500 kernels; the GPU is busy but every launch has its cost.
struct FirstCand { const float* src; float* dst1; float* dst3; float* dst4; ShiftArg T; int shifted; };
struct FirstBatch { FirstCand c[kMaxResident]; };
template <int WIN>
__global__ __launch_bounds__(256) void ssim_first_pass_batch_kernel(const float* __restrict__ im0, Shape3 S, FirstBatch B,
                                                                    float* __restrict__ pmax, int* __restrict__ phasnan) {
    const FirstCand& C = B.c[blockIdx.y];
    if (!C.src) return;       // slot of a rejected candidate or of one scored by its own launches
    Five P;
    for (int a = 0; a < 5; ++a) { P.src[a] = nullptr; P.dst[a] = nullptr; }
    P.dst[1] = C.dst1; P.dst[3] = C.dst3; P.dst[4] = C.dst4;
    float* pm = pmax + (size_t)blockIdx.y * kStatBlocks;
    int* ph = phasnan + (size_t)blockIdx.y * kStatBlocks;
    if (C.shifted) ssim_first_pass_body<WIN, true, 2>(im0, C.src, S, 0, 0, 0, S, 0, P, pm, ph, C.T);
    else ssim_first_pass_body<WIN, false, 2>(im0, C.src, S, 0, 0, 0, S, 0, P, pm, ph, C.T);
}

// last pass (x axis) + SSIM map + sum over the cropped interior (float32 map, float64 mean like skimage)
template <int WIN>
__global__ __launch_bounds__(256) void ssim_last_pass_kernel(Five P, Shape3 R, int ndim, float cov_norm, float C1, float C2,
                                                             double* __restrict__ partial) {
    constexpr int H = WIN / 2, NL = kChunk + 2 * H, pad = (WIN - 1) / 2;
    const int len = R.nx;
    const int nchunks = (len + kChunk - 1) / kChunk;
    const unsigned int rows = (unsigned int)R.nz * (unsigned int)R.ny;
    const unsigned int items = (unsigned int)nchunks * rows;
    double acc = 0.0;
    for (unsigned int w = blockIdx.x * blockDim.x + threadIdx.x; w < items; w += gridDim.x * blockDim.x) {
        const int p0 = (int)(w % (unsigned int)nchunks) * kChunk;
        const unsigned int r = w / (unsigned int)nchunks;
        const int y = (int)(r % (unsigned int)R.ny), z = (int)(r / (unsigned int)R.ny);
        const bool row_in = (y >= pad && y < R.ny - pad) && (ndim == 2 || (z >= pad && z < R.nz - pad));
        if (!row_in) continue;
        const int base = (int)r * R.nx;
        int off[NL];
#pragma unroll
        for (int k = 0; k < NL; ++k) off[k] = base + reflect_index(p0 - H + k, len);
        float f[5][kChunk];
#pragma unroll
        for (int a = 0; a < 5; ++a) {
            float v[NL];
#pragma unroll
            for (int k = 0; k < NL; ++k) v[k] = P.src[a][off[k]];
            box_means<WIN, kChunk>(v, f[a]);
        }
#pragma unroll
        for (int k = 0; k < kChunk; ++k) {
            const int x = p0 + k;
            if (x < pad || x >= R.nx - pad) continue;
            const float a = f[0][k], b = f[1][k];
            const float vx = cov_norm * (f[2][k] - a * a);
            const float vy = cov_norm * (f[3][k] - b * b);
            const float vxy = cov_norm * (f[4][k] - a * b);
            const float A1 = 2.f * a * b + C1, A2 = 2.f * vxy + C2;
            const float B1 = a * a + b * b + C1, B2 = vx + vy + C2;
            acc += (double)((A1 * A2) / (B1 * B2));
        }
    }
    for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off);
    __shared__ double s[4];
    if ((threadIdx.x & 63) == 0) s[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) partial[blockIdx.x] = s[0] + s[1] + s[2] + s[3];
}

// 3D: the y pass and the x pass + SSIM of one candidate in ONE kernel.  A workgroup takes a 32 x 56 tile of the cropped
// interior of one plane: the y-filtered values of the tile and its x halo (5 quantities) go to LDS (float32, the
// rounding point of the separate passes), the x filter and the SSIM formula read them from there -- the y-filtered
// arrays never touch memory, which removes 45 % of the HBM traffic of the three-pass version.
// MODE 0: all five quantities -> SSIM.  MODE 1: only x, xx (P.src[0], P.src[2]): their fully filtered values are WRITTEN to
// P.dst[0], P.dst[2] (no SSIM) -- once per pair.  MODE 2: y, yy, xy are filtered here, x and xx are read, already filtered, from
// P.dst[0], P.dst[2] (written by MODE 1) -> SSIM.  Same arithmetic per quantity in every mode.
template <int WIN, int MODE = 0>
__device__ __forceinline__ void ssim_yx_fused_body(Five P, Shape3 R, float cov_norm, float C1, float C2, double* __restrict__ partial) {
    // tile: 32 rows x 56 voxels -> 62 (WIN = 7) halo columns x 4 row chunks = 248 y-pass items: one round of the 256 threads
    constexpr int H = WIN / 2, pad = (WIN - 1) / 2, TY = 32, TX = 56, LX = TX + 2 * H, NL = kChunk + 2 * H;
    constexpr int NF = MODE == 0 ? 5 : MODE == 1 ? 2 : 3;                       // quantities filtered by this instantiation
    constexpr int QM[5] = {MODE == 2 ? 1 : 0, MODE == 0 ? 1 : MODE == 1 ? 2 : 3, MODE == 0 ? 2 : 4, 3, 4};   // slot -> quantity
    __shared__ float s[NF][TY][LX + 1];
    const int cz = R.nz - 2 * pad, cy = R.ny - 2 * pad, cx = R.nx - 2 * pad;
    double acc = 0.0;
    if (cz > 0 && cy > 0 && cx > 0) {
        const int nty = (cy + TY - 1) / TY, ntx = (cx + TX - 1) / TX;
        const int ntiles = cz * nty * ntx;
        for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
            const int tx = tile % ntx, ty = (tile / ntx) % nty, z = pad + tile / (ntx * nty);
            const int y0 = pad + ty * TY, x0 = pad + tx * TX;
            // ---- y pass: column c of the tile (+ halo), 8 rows per item ----
            for (int item = threadIdx.x; item < LX * (TY / kChunk); item += 256) {
                const int c = item % LX, rc = item / LX;
                const int base = z * R.ny * R.nx + reflect_index(x0 - H + c, R.nx);
                const int r0 = y0 + rc * kChunk;
                int off[NL];
#pragma unroll
                for (int k = 0; k < NL; ++k) off[k] = base + reflect_index(r0 - H + k, R.ny) * R.nx;
#pragma unroll
                for (int a = 0; a < NF; ++a) {
                    float v[NL];
#pragma unroll
                    for (int k = 0; k < NL; ++k) v[k] = P.src[QM[a]][off[k]];
                    float f[kChunk];
                    box_means<WIN, kChunk>(v, f);
#pragma unroll
                    for (int k = 0; k < kChunk; ++k) s[a][rc * kChunk + k][c] = f[k];
                }
            }
            __syncthreads();
            // ---- x pass + SSIM: row `row` of the tile, 8 voxels per thread ----
            {
                const int row = threadIdx.x >> 3, ch = threadIdx.x & 7;
                const int y = y0 + row;
                if (y < R.ny - pad && ch * kChunk < TX) {
                    float f[5][kChunk];
#pragma unroll
                    for (int a = 0; a < NF; ++a) {
                        float v[NL];
#pragma unroll
                        for (int k = 0; k < NL; ++k) v[k] = s[a][row][ch * kChunk + k];
                        box_means<WIN, kChunk>(v, f[QM[a]]);
                    }
                    const int obase = (z * R.ny + y) * R.nx + x0 + ch * kChunk;
#pragma unroll
                    for (int k = 0; k < kChunk; ++k) {
                        if (x0 + ch * kChunk + k >= R.nx - pad || ch * kChunk + k >= TX) continue;
                        if (MODE == 1) { P.dst[0][obase + k] = f[0][k]; P.dst[2][obase + k] = f[2][k]; continue; }
                        if (MODE == 2) { f[0][k] = P.dst[0][obase + k]; f[2][k] = P.dst[2][obase + k]; }
                        const float a = f[0][k], b = f[1][k];
                        const float vx = cov_norm * (f[2][k] - a * a);
                        const float vy = cov_norm * (f[3][k] - b * b);
                        const float vxy = cov_norm * (f[4][k] - a * b);
                        const float A1 = 2.f * a * b + C1, A2 = 2.f * vxy + C2;
                        const float B1 = a * a + b * b + C1, B2 = vx + vy + C2;
                        acc += (double)((A1 * A2) / (B1 * B2));
                    }
                }
            }
            __syncthreads();
        }
    }
    for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off);
    __shared__ double sred[4];
    if ((threadIdx.x & 63) == 0) sred[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (MODE != 1 && threadIdx.x == 0) partial[blockIdx.x] = sred[0] + sred[1] + sred[2] + sred[3];
}

template <int WIN, int MODE = 0>
__global__ __launch_bounds__(256) void ssim_yx_fused_kernel(Five P, Shape3 R, float cov_norm, float C1, float C2, double* __restrict__ partial) {
    ssim_yx_fused_body<WIN, MODE>(P, R, cov_norm, C1, C2, partial);
}

// y / x pass + SSIM of all candidates of a batch in one launch (MODE 2: the fixed image's terms come from ux / uxx)
struct YxCand { const float* src1; const float* src3; const float* src4; };
struct YxBatch { YxCand c[kMaxResident]; };
template <int WIN>
__global__ __launch_bounds__(256) void ssim_yx_batch_kernel(YxBatch B, float* __restrict__ ux, float* __restrict__ uxx, Shape3 R,
                                                            float cov_norm, float C1, float C2, double* __restrict__ partial) {
    const YxCand& C = B.c[blockIdx.y];
    if (!C.src1) return;
    Five P;
    for (int a = 0; a < 5; ++a) { P.src[a] = nullptr; P.dst[a] = nullptr; }
    P.src[1] = C.src1; P.src[3] = C.src3; P.src[4] = C.src4;
    P.dst[0] = ux; P.dst[2] = uxx;
    ssim_yx_fused_body<WIN, 2>(P, R, cov_norm, C1, C2, partial + (size_t)blockIdx.y * kStatBlocks);
}

// ---- all three passes + SSIM of every candidate of a batch in ONE launch (3D, WIN-wide window, whole-volume region, finite
// crops; the fixed image's own window means come, fully filtered, from ux / uxx).  A workgroup owns a TY x TX tile of the
// cropped interior and WALKS z: every thread keeps the running double sums of scipy's uniform_filter1d (tmp += new - old)
// for its pixels of the tile + halo patch, so a plane of z-filtered values (float32, the rounding point of the separate
// passes) goes to LDS, is filtered along y into a second LDS array and along x straight into the SSIM formula.  Nothing
// but the two crops is read -- they are shared by all candidates and stay in L2 / MALL -- and nothing is written: the
// separate passes moved 50 bytes per voxel and candidate through HBM (14 read + 12 written by the z pass, 24 read by the
// y / x pass).  Interior outputs (crop = window / 2) never reach a reflected sample, so no boundary handling is needed;
// the region statistics (nanmax / has-NaN of the candidate image, idempotent) are gathered from every sample loaded.
// Samples come through raw buffer loads: the pixel's byte offset inside a plane is a per-thread constant, the plane a scalar
// offset, so a sample costs no address arithmetic (the kernel is bound by the float64 filter arithmetic: ~300 of its ~580
// vector instructions per plane are f64 adds / multiplies / conversions, all at 1/2 of the fp32 issue rate on gfx950).
// A candidate is its shifted copy (dz = dy = dx = 0) or, for an integer shift, the moving crop itself read in place.
// sel: the work items (tile x z segment, numbered x fastest) this launch walks for the candidate, as a set of residues of the
// item number modulo 32 -- 0xffffffff: all of them; the pruned argmax search (mvs_score_candidates) scores a candidate in rounds
// Residue r of group g (items g K .. g K + K - 1) is item g K + (r + kSelRot g) mod K: without the rotation a residue class of a crop with
// 16 tiles per z segment (x neighbours: 256 x 256 x 51) is ONE tile row -- class 0 the row along the crop's border -- and the first
// 1 / 32 of a candidate says little about the rest of it (a wrong leader is completed, the others are walked further than needed).
constexpr int kSelRot = 7;
struct FusedCand { const float* src; int dz, dy, dx; unsigned int sel; };
struct FusedBatch { FusedCand c[kMaxResident]; };
#ifndef MVS_SSIM_WPE_LO
#define MVS_SSIM_WPE_LO 3
#define MVS_SSIM_WPE_HI 4
#endif
// F32 (round 6): the running sums and the box means in float32.  The SSIM of a candidate never leaves the registration -- only its
// arg max does (registration.py:558-563) -- so the pruned search walks in float32 and re-walks, with this kernel's float64 form, the
// candidates that end within a margin of the leader (see the search).  The running sums restart with every z segment, so their
// drift is bounded by the segment length; the means of the fixed image (ux / uxx) stay the float64 ones.
template <int WIN, bool F32>
__device__ __forceinline__ void ssim_fused_batch_body(const float* __restrict__ im0, Shape3 S, const FusedBatch& B,
                                                               const float* __restrict__ ux, const float* __restrict__ uxx, int zseg,
                                                               float cov_norm, float C1, float C2, float* __restrict__ pmax,
                                                               int* __restrict__ phasnan, double* __restrict__ psum, int selk) {
    constexpr int H = WIN / 2, pad = (WIN - 1) / 2, TY = 16, TX = 56, LY = TY + 2 * H, LX = TX + 2 * H;
    constexpr int NO = 4, NI = NO + 2 * H;           // outputs / inputs of one y- or x-pass item
    constexpr int NR = (LY + 3) / 4;                 // patch rows per thread: row = (tid >> 6) + 4 k, column = tid & 63
    typedef typename std::conditional<F32, float, double>::type run_t;
    constexpr run_t inv = (run_t)(1.0 / (double)WIN);
    static_assert(LX <= 64 && TY % NO == 0 && TX % NO == 0 && (TY * TX) / NO <= 256 && TY / NO == 4, "tile layout");
    __shared__ float sz_[3][LY][LX + 1];             // z-filtered y, yy, xy of the current plane (tile + halo)
    __shared__ float sy_[3][TY][LX + 1];             // ... filtered along y as well
    const FusedCand C = B.c[blockIdx.y];
    const int nselres = __popc(C.sel);
    if (!C.src || nselres == 0) return;
    const int tid = threadIdx.x, col = tid & 63, wrow = tid >> 6;
    const int cz = S.nz - 2 * pad, cy = S.ny - 2 * pad, cx = S.nx - 2 * pad;
    float mx = -INFINITY;
    int hn = 0;
    unsigned long long nanmask = 0;                  // lanes that met a NaN sample of the candidate
    double acc = 0.0;
    const long long vol_bytes = (long long)S.nz * S.ny * S.nx * 4;      // < 2^31 (checked by the host)
    const __amdgpu_buffer_rsrc_t r0 = __builtin_amdgcn_make_buffer_rsrc((void*)im0, 0, (int)vol_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t r1 = __builtin_amdgcn_make_buffer_rsrc((void*)C.src, 0, (int)vol_bytes, 0x00020000);
    if (cz > 0 && cy > 0 && cx > 0) {
        const int nty = (cy + TY - 1) / TY, ntx = (cx + TX - 1) / TX, nzs = (cz + zseg - 1) / zseg;
        const int nitems = nty * ntx * nzs;
        const int sy = S.nx, sz = S.ny * S.nx;
        const int nsel = ((nitems + selk - 1) / selk) * nselres;             // selected items: residue k of group g is number g * nselres + k
        for (int si = blockIdx.x; si < nsel; si += gridDim.x) {
            unsigned int rest = C.sel;
            for (int k = si % nselres; k > 0; --k) rest &= rest - 1;  // (uniform: scalar work)
            const int grp = si / nselres;
            const int item = grp * selk + (__ffs(rest) - 1 + kSelRot * grp) % selk;      // (residue rotated per group: see kSelRot)
            if (item >= nitems) continue;
            const int tx = item % ntx, ty = (item / ntx) % nty, zs = item / (ntx * nty);
            const int z0 = pad + zs * zseg, z1 = min(z0 + zseg, S.nz - pad);
            const int y0 = pad + ty * TY, x0 = pad + tx * TX;
            const int gx = min(x0 - H + col, S.nx - 1);          // (clamped duplicates feed outputs that are never used)
            const bool xin = (unsigned)(gx + C.dx) < (unsigned)S.nx;
            // byte offsets of this thread's pixels inside a plane of im0 / the candidate; the plane itself is a scalar offset
            // of the buffer loads, so a sample costs no address arithmetic.  A pixel outside the candidate in y / x gets an
            // offset beyond the buffer (the load returns 0) and counts as a NaN sample.
            int v0[NR], v1[NR];
            bool in1[NR], out1 = false;
#pragma unroll
            for (int k = 0; k < NR; ++k) {
                const int prow = wrow + 4 * k;
                const int gy = min(y0 - H + min(prow, LY - 1), S.ny - 1);
                in1[k] = xin && (unsigned)(gy + C.dy) < (unsigned)S.ny;
                v0[k] = (gy * sy + gx) * 4;
                v1[k] = in1[k] ? ((gy + C.dy) * sy + gx + C.dx) * 4 : 0x7ffffff0;
                out1 = out1 || (!in1[k] && prow < LY && col < LX);
            }
            if (out1) hn = 1;
            run_t s1[NR], s3[NR], s4[NR];
#pragma unroll
            for (int k = 0; k < NR; ++k) { s1[k] = (run_t)0; s3[k] = (run_t)0; s4[k] = (run_t)0; }
            // plane p + 1's samples (new: plane p + 1, old: plane p + 1 - WIN) are requested right after plane p's have been
            // folded into the running sums, so that their latency hides behind the LDS phases of plane p
            float an[NR], bn[NR], ao[NR], bo[NR];
            auto load_plane = [&](int p) __attribute__((always_inline)) {
                const bool zin = (unsigned)(p + C.dz) < (unsigned)S.nz;
                const bool have_old = p - (z0 - H) >= WIN;
                const bool zin_o = have_old && (unsigned)(p - WIN + C.dz) < (unsigned)S.nz;
                if (!zin) hn = 1;                                   // the whole plane lies outside the candidate
                const int pn0 = p * sz * 4, pn1 = (p + C.dz) * sz * 4, po0 = (p - WIN) * sz * 4, po1 = (p - WIN + C.dz) * sz * 4;
#pragma unroll
                for (int k = 0; k < NR; ++k) {
                    an[k] = __int_as_float(__builtin_amdgcn_raw_buffer_load_b32(r0, v0[k], pn0, 0));
                    bn[k] = zin ? __int_as_float(__builtin_amdgcn_raw_buffer_load_b32(r1, v1[k], pn1, 0)) : 0.f;
                    ao[k] = have_old ? __int_as_float(__builtin_amdgcn_raw_buffer_load_b32(r0, v0[k], po0, 0)) : 0.f;
                    bo[k] = zin_o ? __int_as_float(__builtin_amdgcn_raw_buffer_load_b32(r1, v1[k], po1, 0)) : 0.f;
                }
            };
            if (col < LX) load_plane(z0 - H);
            const int xrow = tid / (TX / NO), xch = tid % (TX / NO);      // x pass: row of the tile, group of NO voxels
            const bool xact = tid < TY * (TX / NO) && y0 + xrow < S.ny - pad;
            for (int p = z0 - H; p < z1 + H; ++p) {
                // the fixed image's window means of this plane's outputs: requested now, used after the two LDS phases
                float fa[NO], faxx[NO];
                if (xact && p >= z0 + H) {
                    const int obase = ((p - H) * S.ny + y0 + xrow) * S.nx + x0 + xch * NO;
#pragma unroll
                    for (int k = 0; k < NO; ++k) {
                        const bool in = x0 + xch * NO + k < S.nx - pad;
                        fa[k] = in ? ux[obase + k] : 0.f;
                        faxx[k] = in ? uxx[obase + k] : 0.f;
                    }
                }
                if (col < LX) {
#pragma unroll
                    for (int k = 0; k < NR; ++k) {
                        const int row = wrow + 4 * k;
                        if (row >= LY) break;
                        const float a = an[k], a2 = ao[k];           // the fixed image is finite (precondition of this launch)
                        float b = bn[k], b2 = bo[k];
                        nanmask |= __ballot(b != b);
                        mx = fmaxf(mx, in1[k] ? b : -INFINITY);      // (NaN never wins a maximum)
                        b = (b != b) ? 0.f : b;
                        b2 = (b2 != b2) ? 0.f : b2;
                        // products in float32 like `im * im`; tmp += new - old like uniform_filter1d
                        s1[k] += (run_t)b - (run_t)b2;
                        s3[k] += (run_t)(b * b) - (run_t)(b2 * b2);
                        s4[k] += (run_t)(a * b) - (run_t)(a2 * b2);
                        if (p >= z0 + H) {
                            sz_[0][row][col] = (float)(s1[k] * inv);
                            sz_[1][row][col] = (float)(s3[k] * inv);
                            sz_[2][row][col] = (float)(s4[k] * inv);
                        }
                    }
                    if (p + 1 < z1 + H) load_plane(p + 1);
                }
                if (p < z0 + H) continue;

                __syncthreads();
                // ---- y pass: column c of the patch, NO rows per item ----
                if (col < LX) {                                   // one wavefront per group of NO rows: lanes = consecutive columns (no bank conflicts)
                    const int c = col, rc = wrow;
#pragma unroll
                    for (int a = 0; a < 3; ++a) {
                        float v[NI], f[NO];
#pragma unroll
                        for (int k = 0; k < NI; ++k) v[k] = sz_[a][rc * NO + k][c];
                        box_means<WIN, NO, run_t>(v, f);
#pragma unroll
                        for (int k = 0; k < NO; ++k) sy_[a][rc * NO + k][c] = f[k];
                    }
                }
                __syncthreads();
                // ---- x pass + SSIM: row `row` of the tile, NO voxels per thread ----
                {
                    const int row = xrow, ch = xch;
                    if (xact) {
                        float f[3][NO];
#pragma unroll
                        for (int a = 0; a < 3; ++a) {
                            float v[NI];
#pragma unroll
                            for (int k = 0; k < NI; ++k) v[k] = sy_[a][row][ch * NO + k];
                            box_means<WIN, NO, run_t>(v, f[a]);
                        }
#pragma unroll
                        for (int k = 0; k < NO; ++k) {
                            if (x0 + ch * NO + k >= S.nx - pad) continue;
                            const float a = fa[k], axx = faxx[k], b = f[0][k];
                            const float vx = cov_norm * (axx - a * a);
                            const float vy = cov_norm * (f[1][k] - b * b);
                            const float vxy = cov_norm * (f[2][k] - a * b);
                            const float A1 = 2.f * a * b + C1, A2 = 2.f * vxy + C2;
                            const float B1 = a * a + b * b + C1, B2 = vx + vy + C2;
                            acc += (double)((A1 * A2) / (B1 * B2));
                        }
                    }
                }
            }
            __syncthreads();
        }
    }
    if (nanmask) hn = 1;
    for (int off = 32; off > 0; off >>= 1) {
        mx = fmaxf(mx, __shfl_down(mx, off));
        hn |= __shfl_down(hn, off);
        acc += __shfl_down(acc, off);
    }
    __shared__ float r_mx[4];
    __shared__ int r_hn[4];
    __shared__ double r_acc[4];
    if ((tid & 63) == 0) { r_mx[tid >> 6] = mx; r_hn[tid >> 6] = hn; r_acc[tid >> 6] = acc; }
    __syncthreads();
    if (tid == 0) {
        for (int w = 1; w < 4; ++w) { mx = fmaxf(mx, r_mx[w]); hn |= r_hn[w]; acc += r_acc[w]; }
        const size_t o = (size_t)blockIdx.y * kStatBlocks + blockIdx.x;
        pmax[o] = mx; phasnan[o] = hn; psum[o] = acc;
    }
}

#ifndef MVS_SSIM_F32_WPE_LO
#define MVS_SSIM_F32_WPE_LO 3
#define MVS_SSIM_F32_WPE_HI 4
#endif
template <int WIN, bool F32 = false>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(MVS_SSIM_WPE_LO, MVS_SSIM_WPE_HI)))
void ssim_fused_batch_kernel(const float* __restrict__ im0, Shape3 S, FusedBatch B, const float* __restrict__ ux, const float* __restrict__ uxx, int zseg,
                             float cov_norm, float C1, float C2, float* __restrict__ pmax, int* __restrict__ phasnan, double* __restrict__ psum, int selk) {
    ssim_fused_batch_body<WIN, false>(im0, S, B, ux, uxx, zseg, cov_norm, C1, C2, pmax, phasnan, psum, selk);
}
template <int WIN>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(MVS_SSIM_F32_WPE_LO, MVS_SSIM_F32_WPE_HI)))
void ssim_fused_batch_f32_kernel(const float* __restrict__ im0, Shape3 S, FusedBatch B, const float* __restrict__ ux, const float* __restrict__ uxx, int zseg,
                                 float cov_norm, float C1, float C2, float* __restrict__ pmax, int* __restrict__ phasnan, double* __restrict__ psum, int selk) {
    ssim_fused_batch_body<WIN, true>(im0, S, B, ux, uxx, zseg, cov_norm, C1, C2, pmax, phasnan, psum, selk);
}

// ---- the fixed image's own window means (x, x * x) by the same walk: one launch instead of the z pass + the y / x pass, and the
// z-filtered volumes (2 x 13 MB written and read back) never exist.  Same tiles, same per-thread running sums and rounding points
// as ssim_fused_batch_kernel with two quantities instead of three; the results go to ux / uxx on the cropped interior (all the
// candidates' walks read).  Equal to the separate passes up to where the float64 running sums start (1e-16 relative before the
// float32 rounding of each pass).  The fixed image is finite (precondition of the shared terms).
template <int WIN>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4, 8)))
void ssim_fixed_walk_kernel(const float* __restrict__ im0, Shape3 S, float* __restrict__ ux, float* __restrict__ uxx, int zseg) {
    constexpr int H = WIN / 2, pad = (WIN - 1) / 2, TY = 16, TX = 56, LY = TY + 2 * H, LX = TX + 2 * H;
    constexpr int NO = 4, NI = NO + 2 * H;
    constexpr int NR = (LY + 3) / 4;
    constexpr double inv = 1.0 / (double)WIN;
    static_assert(LX <= 64 && TY % NO == 0 && TX % NO == 0 && (TY * TX) / NO <= 256 && TY / NO == 4, "tile layout");
    __shared__ float sz_[2][LY][LX + 1];
    __shared__ float sy_[2][TY][LX + 1];
    const int tid = threadIdx.x, col = tid & 63, wrow = tid >> 6;
    const int cz = S.nz - 2 * pad, cy = S.ny - 2 * pad, cx = S.nx - 2 * pad;
    if (cz <= 0 || cy <= 0 || cx <= 0) return;
    const long long vol_bytes = (long long)S.nz * S.ny * S.nx * 4;      // < 2^31 (checked by the host)
    const __amdgpu_buffer_rsrc_t r0 = __builtin_amdgcn_make_buffer_rsrc((void*)im0, 0, (int)vol_bytes, 0x00020000);
    const int nty = (cy + TY - 1) / TY, ntx = (cx + TX - 1) / TX, nzs = (cz + zseg - 1) / zseg;
    const int nitems = nty * ntx * nzs;
    const int sy = S.nx, sz = S.ny * S.nx;
    for (int item = blockIdx.x; item < nitems; item += gridDim.x) {
        const int tx = item % ntx, ty = (item / ntx) % nty, zs = item / (ntx * nty);
        const int z0 = pad + zs * zseg, z1 = min(z0 + zseg, S.nz - pad);
        const int y0 = pad + ty * TY, x0 = pad + tx * TX;
        const int gx = min(x0 - H + col, S.nx - 1);          // (clamped duplicates feed outputs that are never used)
        int v0[NR];
#pragma unroll
        for (int k = 0; k < NR; ++k) {
            const int gy = min(y0 - H + min(wrow + 4 * k, LY - 1), S.ny - 1);
            v0[k] = (gy * sy + gx) * 4;
        }
        double s1[NR], s2[NR];
#pragma unroll
        for (int k = 0; k < NR; ++k) { s1[k] = 0.0; s2[k] = 0.0; }
        float an[NR], ao[NR];
        auto load_plane = [&](int p) __attribute__((always_inline)) {
            const bool have_old = p - (z0 - H) >= WIN;
            const int pn0 = p * sz * 4, po0 = (p - WIN) * sz * 4;
#pragma unroll
            for (int k = 0; k < NR; ++k) {
                an[k] = __int_as_float(__builtin_amdgcn_raw_buffer_load_b32(r0, v0[k], pn0, 0));
                ao[k] = have_old ? __int_as_float(__builtin_amdgcn_raw_buffer_load_b32(r0, v0[k], po0, 0)) : 0.f;
            }
        };
        if (col < LX) load_plane(z0 - H);
        const int xrow = tid / (TX / NO), xch = tid % (TX / NO);
        const bool xact = tid < TY * (TX / NO) && y0 + xrow < S.ny - pad;
        for (int p = z0 - H; p < z1 + H; ++p) {
            if (col < LX) {
#pragma unroll
                for (int k = 0; k < NR; ++k) {
                    const int row = wrow + 4 * k;
                    if (row >= LY) break;
                    const float a = an[k], a2 = ao[k];
                    // products in float32 like `im * im`; tmp += new - old like uniform_filter1d
                    s1[k] += (double)a - (double)a2;
                    s2[k] += (double)(a * a) - (double)(a2 * a2);
                    if (p >= z0 + H) {
                        sz_[0][row][col] = (float)(s1[k] * inv);
                        sz_[1][row][col] = (float)(s2[k] * inv);
                    }
                }
                if (p + 1 < z1 + H) load_plane(p + 1);
            }
            if (p < z0 + H) continue;
            __syncthreads();
            if (col < LX) {                                   // y pass: one wavefront per group of NO rows
#pragma unroll
                for (int a = 0; a < 2; ++a) {
                    float v[NI], f[NO];
#pragma unroll
                    for (int k = 0; k < NI; ++k) v[k] = sz_[a][wrow * NO + k][col];
                    box_means<WIN, NO>(v, f);
#pragma unroll
                    for (int k = 0; k < NO; ++k) sy_[a][wrow * NO + k][col] = f[k];
                }
            }
            __syncthreads();
            if (xact) {                                       // x pass: NO voxels of a row per thread
                float f[2][NO];
#pragma unroll
                for (int a = 0; a < 2; ++a) {
                    float v[NI];
#pragma unroll
                    for (int k = 0; k < NI; ++k) v[k] = sy_[a][xrow][xch * NO + k];
                    box_means<WIN, NO>(v, f[a]);
                }
                const int obase = ((p - H) * S.ny + y0 + xrow) * S.nx + x0 + xch * NO;
#pragma unroll
                for (int k = 0; k < NO; ++k) {
                    if (x0 + xch * NO + k >= S.nx - pad) continue;
                    ux[obase + k] = f[0][k];
                    uxx[obase + k] = f[1][k];
                }
            }
        }
        __syncthreads();
    }
}

// folds the per-workgroup partials of one candidate's SSIM passes
__global__ __launch_bounds__(256) void finish_region_kernel(const float* __restrict__ pmax, const int* __restrict__ phasnan,
                                                            const double* __restrict__ psum, RegionStats* __restrict__ out,
                                                            int nblk = kStatBlocks) {
    const size_t o = (size_t)blockIdx.x * kStatBlocks;
    float mx = -INFINITY;
    int hn = 0;
    double sum = 0.0;
    for (int i = threadIdx.x; i < nblk; i += 256) { mx = fmaxf(mx, pmax[o + i]); hn |= phasnan[o + i]; sum += psum[o + i]; }
    for (int off = 32; off > 0; off >>= 1) {
        mx = fmaxf(mx, __shfl_down(mx, off)); hn |= __shfl_down(hn, off); sum += __shfl_down(sum, off);
    }
    __shared__ float s_mx[4];
    __shared__ int s_hn[4];
    __shared__ double s_sum[4];
    if ((threadIdx.x & 63) == 0) { s_mx[threadIdx.x >> 6] = mx; s_hn[threadIdx.x >> 6] = hn; s_sum[threadIdx.x >> 6] = sum; }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < 4; ++w) { mx = fmaxf(mx, s_mx[w]); hn |= s_hn[w]; sum += s_sum[w]; }
        out[blockIdx.x].mx = mx; out[blockIdx.x].hasnan = hn; out[blockIdx.x].ssim_sum = sum;
    }
}

// ---- Spearman: compaction of the jointly valid voxels, two chained radix sorts, rank correlation ------------
// kx = im0[mask], ky = im1t[mask] - 1 (float32, like the reference).  Each thread takes 8 consecutive voxels; a
// workgroup reserves its output range with ONE atomic (the order of the pairs is irrelevant to a rank correlation).
// raw0 != nullptr: the fixed image's sort key is its integer-valued original (< 65536) stored as uint32 -- same order and
// ties as the rescaled value (the rescaling is strictly increasing and one-to-one on integers), a 2-pass radix sort instead of 4.
__global__ __launch_bounds__(256) void compact_kernel(const float* __restrict__ im0, const float* __restrict__ im1t, long long n,
                                                      float* __restrict__ kx, float* __restrict__ ky,
                                                      unsigned int* __restrict__ counter, const float* __restrict__ raw0) {
    constexpr int K = 8;
    __shared__ unsigned int s_wave[4];
    __shared__ unsigned int s_base;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const long long tile = 256LL * K;
    for (long long t0 = (long long)blockIdx.x * tile; t0 < n; t0 += (long long)gridDim.x * tile) {
        const long long i0 = t0 + (long long)threadIdx.x * K;
        float a[K], b[K];
        unsigned int cnt = 0;
#pragma unroll
        for (int k = 0; k < K; ++k) {
            const long long i = i0 + k;
            a[k] = (i < n) ? im0[i] : NAN;
            b[k] = (i < n) ? im1t[i] : NAN;
            cnt += (a[k] == a[k] && b[k] == b[k]) ? 1u : 0u;
        }
        // exclusive scan of cnt over the workgroup
        unsigned int incl = cnt;
        for (int off = 1; off < 64; off <<= 1) {
            const unsigned int o = __shfl_up(incl, off);
            if (lane >= off) incl += o;
        }
        if (lane == 63) s_wave[wave] = incl;
        __syncthreads();
        unsigned int wbase = 0;
        for (int w = 0; w < wave; ++w) wbase += s_wave[w];
        if (threadIdx.x == 255) s_base = atomicAdd(counter, wbase + incl);
        __syncthreads();
        unsigned int p = s_base + wbase + incl - cnt;
#pragma unroll
        for (int k = 0; k < K; ++k)
            if (a[k] == a[k] && b[k] == b[k]) {
                if (raw0) reinterpret_cast<unsigned int*>(kx)[p] = (unsigned int)raw0[i0 + k];
                else kx[p] = a[k];
                ky[p] = b[k] - 1.0f;
                ++p;
            }
        __syncthreads();
    }
}

// average rank (scipy.stats.rankdata method="average") of sorted position i: (first + last + 1) / 2 where [first, last) is the
// run of equal keys around i.
// Average ranks of one chunk of kRankChunk consecutive sorted keys, 8 per thread (thread t owns positions
// base + 8 t .. + 7): run starts / ends are flagged against the neighbouring key, the start (end) of the run a position
// belongs to is a forward max-scan (backward min-scan) of the flagged positions -- in registers within a thread, by
// shuffles within a wavefront, through LDS across the four wavefronts.  Only the run that enters the chunk from the left
// and the one that leaves it on the right need a search outside the chunk (galloping, one thread each).  Replaces a
// galloping search per element (~20 dependent loads each on the long runs of quantised image data).
constexpr int kRankItems = 8, kRankChunk = 256 * kRankItems;

template <typename K>
__device__ __forceinline__ void chunk_average_ranks(const K* __restrict__ sorted, unsigned int n, unsigned int chunk0,
                                                    float (&rank)[kRankItems], bool (&valid)[kRankItems]) {
    __shared__ unsigned int s_first[4], s_last[4], s_edge[2];
    const unsigned int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const unsigned int base = chunk0 + tid * kRankItems;
    const unsigned int chunk_end = min(chunk0 + (unsigned int)kRankChunk, n);   // exclusive
    K k[kRankItems + 2];                                                      // k[0] = left neighbour, k[9] = right neighbour
#pragma unroll
    for (int j = 0; j < kRankItems + 2; ++j) {
        const long long idx = (long long)base + j - 1;
        k[j] = (idx >= 0 && idx < (long long)n) ? sorted[idx] : (K)0;
    }
    if (tid == 0) {   // run entering the chunk from the left: its first index (galloping backwards), else the chunk start
        unsigned int first = chunk0;
        if (chunk0 > 0 && chunk0 < n && sorted[chunk0 - 1] == sorted[chunk0]) {
            const K key = sorted[chunk0];
            unsigned int pos = chunk0, step = 1;
            while (pos >= step && sorted[pos - step] == key) { pos -= step; step <<= 1; }
            unsigned int lo = pos >= step ? pos - step + 1 : 0, hi = pos;
            while (lo < hi) { const unsigned int m = (lo + hi) >> 1; if (sorted[m] < key) lo = m + 1; else hi = m; }
            first = lo;
        }
        s_edge[0] = first;
    }
    if (tid == 64) {  // run leaving the chunk on the right: its exclusive end (galloping forwards)
        unsigned int last = chunk_end;
        if (chunk_end < n && chunk_end > 0 && sorted[chunk_end] == sorted[chunk_end - 1]) {
            const K key = sorted[chunk_end - 1];
            unsigned int pos = chunk_end - 1, step = 1;
            while (pos + step < n && sorted[pos + step] == key) { pos += step; step <<= 1; }
            unsigned int lo = pos + 1, hi = min(pos + step, n);
            while (lo < hi) { const unsigned int m = (lo + hi) >> 1; if (sorted[m] <= key) lo = m + 1; else hi = m; }
            last = lo;
        }
        s_edge[1] = last;
    }
    // ---- forward: index of the last run start at or before each position (0 = none seen yet; stored as index + 1) ----
    unsigned int first[kRankItems], cur = 0;
#pragma unroll
    for (int j = 0; j < kRankItems; ++j) {
        const unsigned int idx = base + j;
        valid[j] = idx < n;
        if (valid[j] && (idx == 0 || k[j] != k[j + 1])) cur = idx + 1;
        first[j] = cur;
    }
    unsigned int incl = cur;   // inclusive max-scan of the per-thread carries over the wavefront
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const unsigned int o = __shfl_up(incl, off);
        if (lane >= (unsigned int)off) incl = max(incl, o);
    }
    unsigned int carry_f = __shfl_up(incl, 1);
    if (lane == 0) carry_f = 0;
    if (lane == 63) s_first[wave] = incl;
    // ---- backward: exclusive end of the run each position belongs to (0xffffffff = not seen yet) ----
    unsigned int last[kRankItems], curl = 0xffffffffu;
#pragma unroll
    for (int j = kRankItems - 1; j >= 0; --j) {
        const unsigned int idx = base + j;
        if (idx < n && (idx == n - 1 || k[j + 1] != k[j + 2])) curl = idx + 1;
        last[j] = curl;
    }
    unsigned int incl_l = curl;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const unsigned int o = __shfl_down(incl_l, off);
        if (lane + (unsigned int)off < 64) incl_l = min(incl_l, o);
    }
    unsigned int carry_l = __shfl_down(incl_l, 1);
    if (lane == 63) carry_l = 0xffffffffu;
    if (lane == 0) s_last[wave] = incl_l;
    __syncthreads();
    for (unsigned int w = 0; w < wave; ++w) carry_f = max(carry_f, s_first[w]);
    for (unsigned int w = wave + 1; w < 4; ++w) carry_l = min(carry_l, s_last[w]);
    const unsigned int edge_first = s_edge[0], edge_last = s_edge[1];
#pragma unroll
    for (int j = 0; j < kRankItems; ++j) {
        unsigned int f = max(first[j], carry_f), l = min(last[j], carry_l);
        f = f ? f - 1 : edge_first;
        if (l == 0xffffffffu) l = edge_last;
        rank[j] = (float)(0.5 * ((double)f + (double)l + 1.0));
    }
    __syncthreads();   // the LDS cells are reused by the next chunk
}

// ranks of the x keys in x-sorted order (they then ride along the second sort as its payload, so nothing is
// ever scattered back to voxel order).  Stored as float like scipy's float64 ranks rounded: exact for n < 2^24.
template <typename K>
__global__ __launch_bounds__(256) void ranks_sorted_kernel(const K* __restrict__ sorted, unsigned int n, float* __restrict__ rank_out) {
    const unsigned int nchunks = (n + kRankChunk - 1) / kRankChunk;
    for (unsigned int c = blockIdx.x; c < nchunks; c += gridDim.x) {
        float rank[kRankItems];
        bool valid[kRankItems];
        chunk_average_ranks(sorted, n, c * kRankChunk, rank, valid);
        const unsigned int base = c * kRankChunk + threadIdx.x * kRankItems;
        if (base + kRankItems <= n) {
            *reinterpret_cast<float4*>(rank_out + base) = make_float4(rank[0], rank[1], rank[2], rank[3]);
            *reinterpret_cast<float4*>(rank_out + base + 4) = make_float4(rank[4], rank[5], rank[6], rank[7]);
        } else {
#pragma unroll
            for (int j = 0; j < kRankItems; ++j) if (valid[j]) rank_out[base + j] = rank[j];
        }
    }
}

// in y-sorted order: rank of y computed on the fly, rank of x from the payload; sums for the Pearson
// correlation of the two rank vectors (both have mean (n+1)/2)
__global__ __launch_bounds__(256) void rankcorr_kernel(const float* __restrict__ ysorted, const float* __restrict__ rx, unsigned int n,
                                                       double mean, double* __restrict__ partial) {
    double sxy = 0.0, sxx = 0.0, syy = 0.0;
    const unsigned int nchunks = (n + kRankChunk - 1) / kRankChunk;
    for (unsigned int c = blockIdx.x; c < nchunks; c += gridDim.x) {
        float rank[kRankItems];
        bool valid[kRankItems];
        chunk_average_ranks(ysorted, n, c * kRankChunk, rank, valid);
        const unsigned int base = c * kRankChunk + threadIdx.x * kRankItems;
#pragma unroll
        for (int j = 0; j < kRankItems; ++j)
            if (valid[j]) {
                const double a = (double)rx[base + j] - mean, b = (double)rank[j] - mean;
                sxy += a * b; sxx += a * a; syy += b * b;
            }
    }
    for (int off = 32; off > 0; off >>= 1) {
        sxy += __shfl_down(sxy, off); sxx += __shfl_down(sxx, off); syy += __shfl_down(syy, off);
    }
    __shared__ double s[3][4];
    if ((threadIdx.x & 63) == 0) { s[0][threadIdx.x >> 6] = sxy; s[1][threadIdx.x >> 6] = sxx; s[2][threadIdx.x >> 6] = syy; }
    __syncthreads();
    if (threadIdx.x == 0)
        for (int k = 0; k < 3; ++k) partial[blockIdx.x * 3 + k] = s[k][0] + s[k][1] + s[k][2] + s[k][3];
}

// ---- Spearman by histogram ranks (integer-valued crops, shifts that are multiples of 1/2) -------------------------------
// When both crops hold 16-bit integers (uint8 / uint16 tiles on the fixed grid, binned or not) the rescaled fixed image is
// a strictly increasing function of the raw integer, and the shifted moving image -- linear interpolation with weights 0,
// 1/2 or 1 per axis when every shift component is a multiple of 1/2 -- is a strictly increasing function of
// KEY = 8 x (sum of the taps with a non-zero weight) / 2^(#fractional axes), an integer below 2^19.  The average rank of a
// voxel is then a function of its key alone: rank(v) = #keys < v + (#keys == v + 1) / 2, read from the prefix sums of a
// histogram.  Two passes over the voxels (histogram; correlation of the centred ranks) and one scan of the bins replace the
// compaction, two multi-pass radix sorts and the run-length rank kernels.  Ties and order are those of the exact
// interpolated values; the float32 values scipy ranks are those values rounded once, distinct wherever the exact ones are
// (spacing >= 2^-3 / range against 2^-24 relative), so the rank vectors coincide.
constexpr int kHistParts = 256;           // workgroups whose private histograms are written out and folded (more: atomic flush)
constexpr int kHistBinsMax = 48 * 1024;      // x + y bins a workgroup's private histogram can hold (16-bit counters in 96 KiB of LDS)
__device__ __forceinline__ int shifted_keysum(const float* __restrict__ raw1, int sy, int sz, const AxisTap& Z, const AxisTap& Y,
                                              const AxisTap& X) {
    // sum of the taps with a non-zero weight (2^#fractional-axes of them: the same number for every voxel of a candidate,
    // so the sum orders the voxels like the interpolated value); -1 outside the moving image
    if (!(Z.ok && Y.ok && X.ok)) return -1;
    const bool fz = Z.w != 0.0, fy = Y.w != 0.0, fx = X.w != 0.0;
    float acc = 0.f;
    for (int a = 0; a <= (fz ? 1 : 0); ++a)
        for (int b = 0; b <= (fy ? 1 : 0); ++b)
            for (int q = 0; q <= (fx ? 1 : 0); ++q)
                acc += raw1[(a ? Z.i1 : Z.i0) * sz + (b ? Y.i1 : Y.i0) * sy + (q ? X.i1 : X.i0)];
    return (int)acc;
}
// CORR = false: key histograms (hx: nbx bins from key kx0, hy: nby bins from key ky0).  Every workgroup counts its voxels
// (fewer than 65536) in a private LDS histogram of 16-bit counters -- neighbouring voxels of a smooth image share their
// keys, global atomics on them serialise -- and adds its non-empty bins to the global tables.
// CORR = true: sum of rx[kx] * ry[ky] over the jointly valid voxels.
template <bool CORR>
__global__ __launch_bounds__(CORR ? 256 : 1024) void hist_rank_kernel(const float* __restrict__ raw0, const float* __restrict__ raw1, Shape3 S, double tz,
                                                        double ty, double tx, int kx0, int nbx, int ky0, int nby,
                                                        unsigned int* __restrict__ hx, unsigned int* __restrict__ hy,
                                                        const float* __restrict__ rx, const float* __restrict__ ry,
                                                        double* __restrict__ partial, unsigned int* __restrict__ parts) {
    extern __shared__ unsigned int s_hist[];          // (nbx + nby + 1) / 2 words of two 16-bit counters
    const unsigned int n = (unsigned int)S.nz * S.ny * S.nx;
    const int sy = S.nx, sz = S.ny * S.nx;
    const int nwords = (nbx + nby + 1) / 2;
    if (!CORR) {
        for (int i = threadIdx.x; i < nwords; i += blockDim.x) s_hist[i] = 0u;
        __syncthreads();
    }
    double sxy = 0.0;
    // contiguous range of 4-voxel groups per workgroup (at most 16383 groups = 65532 voxels: the counters cannot overflow)
    const unsigned int ngroups = (n + 3) / 4;
    const unsigned int per = (ngroups + gridDim.x - 1) / gridDim.x;
    const unsigned int g0 = blockIdx.x * per, g1 = min(g0 + per, ngroups);
    for (unsigned int g = g0 + threadIdx.x; g < g1; g += blockDim.x) {
        const unsigned int i0 = g * 4;
        int x = (int)(i0 % (unsigned int)S.nx);
        const unsigned int t = i0 / (unsigned int)S.nx;
        int y = (int)(t % (unsigned int)S.ny), z = (int)(t / (unsigned int)S.ny);
        int row_z = -1, row_y = -1;
        AxisTap Z = {0, 0, 0, 0.0}, Y = {0, 0, 0, 0.0};
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            if (i0 + k < n) {
                if (z != row_z || y != row_y) {
                    row_z = z; row_y = y;
                    Z = axis_tap(z, tz, S.nz);
                    Y = axis_tap(y, ty, S.ny);
                }
                const int ks = shifted_keysum(raw1, sy, sz, Z, Y, axis_tap(x, tx, S.nx));
                if (ks >= 0) {
                    const int kx = (int)raw0[i0 + k] - kx0, ky = ks - ky0;
                    if (CORR) sxy += (double)rx[kx] * (double)ry[ky];
                    else {
                        atomicAdd(&s_hist[kx >> 1], 1u << (16 * (kx & 1)));
                        const int q = nbx + ky;
                        atomicAdd(&s_hist[q >> 1], 1u << (16 * (q & 1)));
                    }
                }
                if (++x == S.nx) { x = 0; if (++y == S.ny) { y = 0; ++z; } }
            }
        }
    }
    if (CORR) {
        for (int off = 32; off > 0; off >>= 1) sxy += __shfl_down(sxy, off);
        __shared__ double s[4];
        if ((threadIdx.x & 63) == 0) s[threadIdx.x >> 6] = sxy;
        __syncthreads();
        if (threadIdx.x == 0) partial[blockIdx.x] = s[0] + s[1] + s[2] + s[3];
    } else if (parts) {
        // the private histogram goes out whole (coalesced, no atomics); hist_fold_kernel adds the workgroups' parts
        __syncthreads();
        unsigned int* mine = parts + (size_t)blockIdx.x * nwords;
        for (int i = threadIdx.x; i < nwords; i += blockDim.x) mine[i] = s_hist[i];
    } else {
        __syncthreads();
        for (int i = threadIdx.x; i < nwords; i += blockDim.x) {
            const unsigned int w = s_hist[i];
            if (w) {
                const int b0 = 2 * i, b1 = 2 * i + 1;
                const unsigned int c0 = w & 0xffffu, c1 = w >> 16;
                if (c0) atomicAdd((b0 < nbx) ? hx + b0 : hy + (b0 - nbx), c0);
                if (c1) atomicAdd((b1 < nbx) ? hx + b1 : hy + (b1 - nbx), c1);
            }
        }
    }
}
// One workgroup per table: centred average ranks of the bins, rc[v] = #(keys < v) + (h[v] + 1) / 2 - (m + 1) / 2, and
// out[0] = sum_v h[v] rc[v]^2, out[1] = m (the number of keys).  1024 threads, contiguous bin ranges, two passes.
// sums the packed 16-bit histograms of `nparts` workgroups into the 32-bit histograms hx (bins < nbx) and hy: a workgroup
// takes 64 words, its four wavefronts a quarter of the parts each
__global__ __launch_bounds__(1024) void hist_fold_kernel(const unsigned int* __restrict__ parts, int nparts, int nwords, int nbx, int nby,
                                                         unsigned int* __restrict__ hx, unsigned int* __restrict__ hy) {
    // sixteen wavefronts, a sixteenth of the parts each: a wavefront's loads are one dependent-latency chain (256 parts in four
    // chains of 64 took 12 us for a few MB)
    __shared__ unsigned int s0[16][64], s1[16][64];
    const int lane = threadIdx.x & 63, q = threadIdx.x >> 6;
    const int i = blockIdx.x * 64 + lane;
    unsigned int c0 = 0, c1 = 0;
    if (i < nwords)
        for (int w = q; w < nparts; w += 16) {
            const unsigned int v = parts[(size_t)w * nwords + i];
            c0 += v & 0xffffu;
            c1 += v >> 16;
        }
    s0[q][lane] = c0; s1[q][lane] = c1;
    __syncthreads();
    if (q == 0 && i < nwords) {
        c0 = 0; c1 = 0;
        for (int k = 0; k < 16; ++k) { c0 += s0[k][lane]; c1 += s1[k][lane]; }
        const int b0 = 2 * i, b1 = 2 * i + 1;
        if (b0 < nbx) hx[b0] = c0; else if (b0 - nbx < nby) hy[b0 - nbx] = c0;
        if (b1 < nbx) hx[b1] = c1; else if (b1 - nbx < nby) hy[b1 - nbx] = c1;
    }
}

__global__ __launch_bounds__(1024) void rank_table_kernel(const unsigned int* __restrict__ hx, int nx, float* __restrict__ rx,
                                                          const unsigned int* __restrict__ hy, int ny, float* __restrict__ ry,
                                                          double* __restrict__ out) {
    const unsigned int* h = blockIdx.x ? hy : hx;
    const int nb = blockIdx.x ? ny : nx;
    float* r = blockIdx.x ? ry : rx;
    __shared__ unsigned long long s_part[1024];
    __shared__ double s_var[1024];
    const int per = (nb + 1023) / 1024, b0 = threadIdx.x * per, b1 = min(b0 + per, nb);
    unsigned long long loc = 0;
    for (int b = b0; b < b1; ++b) loc += h[b];
    // exclusive prefix over the 1024 per-thread counts: shuffle scan inside every wavefront, then over the 16 wavefront totals
    // (integers: any order gives the same sums; thread 0 walking the 1024 entries took 8 of this kernel's 19 us)
    {
        const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
        unsigned long long inc = loc;
        for (int off = 1; off < 64; off <<= 1) {
            const unsigned long long up = __shfl_up(inc, off);
            if (lane >= off) inc += up;
        }
        if (lane == 63) s_part[wv] = inc;          // wavefront totals
        __syncthreads();
        unsigned long long base = 0, total = 0;
        for (int k = 0; k < 16; ++k) { const unsigned long long t = s_part[k]; if (k < wv) base += t; total += t; }
        __syncthreads();
        s_part[threadIdx.x] = base + inc - loc;
        if (threadIdx.x == 0) s_var[0] = (double)total;       // m
    }
    __syncthreads();
    const double m = s_var[0];
    __syncthreads();
    unsigned long long before = s_part[threadIdx.x];
    double var = 0.0;
    for (int b = b0; b < b1; ++b) {
        const unsigned int c = h[b];
        const double rc = (double)before + ((double)c + 1.0) * 0.5 - (m + 1.0) * 0.5;
        r[b] = (float)rc;
        var += (double)c * rc * rc;
        before += c;
    }
    s_var[threadIdx.x] = var;
    __syncthreads();
    for (int off = 512; off > 0; off >>= 1) {
        if ((int)threadIdx.x < off) s_var[threadIdx.x] += s_var[threadIdx.x + off];
        __syncthreads();
    }
    if (threadIdx.x == 0) { out[2 * blockIdx.x] = s_var[0]; out[2 * blockIdx.x + 1] = m; }
}

struct DeviceBump {   // bump allocator over one scratch slot
    char* base; size_t cap, used;
    template <typename T> T* take(size_t count) {
        const size_t bytes = (count * sizeof(T) + 255) / 256 * 256;
        if (used + bytes > cap) return nullptr;
        T* p = (T*)(base + used);
        used += bytes;
        return p;
    }
};

// The fixed image's own terms (mean of x and of x * x over the window) of a 3D pair whose SSIM region is the whole volume for
// every candidate: z pass + y/x pass once, results in setB[2], setB[3] (shared_x of launch_ssim_passes).
template <int WIN>
void launch_ssim_shared_x(hipStream_t stream, const float* im0, Shape3 S, Shape3 R, float* const setA[5], float* const setB[5],
                          float* pmax, int* phasnan, bool walk) {
    Five P1, P2;
    for (int a = 0; a < 5; ++a) { P1.src[a] = nullptr; P1.dst[a] = setA[a]; P2.src[a] = setA[a]; P2.dst[a] = nullptr; }
    P2.dst[0] = setB[2];
    P2.dst[2] = setB[3];
    if (walk && WIN == 7 && R.nz == S.nz && R.ny == S.ny && R.nx == S.nx) {
        // one launch: z walk per (y, x) tile, about one resident round of work items
        const int tiles = ((S.ny - 6 + 15) / 16) * ((S.nx - 6 + 55) / 56), cz = S.nz - 6;
        const int nzs = std::max(1, std::min(768 / std::max(tiles, 1), (cz + 7) / 8));
        MVS_DUP("ssim_fixed", hipLaunchKernelGGL(ssim_fixed_walk_kernel<7>, dim3(kStatBlocks), dim3(256), 0, stream, im0, S, setB[2], setB[3], (cz + nzs - 1) / nzs));
        return;
    }
    hipLaunchKernelGGL((ssim_first_pass_kernel<WIN, false, 1>), dim3(kStatBlocks), dim3(256), 0, stream, im0, im0, S, 0, 0, 0, R, 0, P1, pmax, phasnan,
                       ShiftArg{0.0, 0.0, 0.0, 0});
    hipLaunchKernelGGL((ssim_yx_fused_kernel<WIN, 1>), dim3(kStatBlocks), dim3(256), 0, stream, P2, R, 0.f, 0.f, 0.f, (double*)nullptr);
}

template <int WIN>
void launch_ssim_passes(hipStream_t stream, const float* im0, const float* im1t, Shape3 S, const int lo[3], Shape3 R, int ndim,
                        float* const setA[5], float* const setB[5], float cov_norm, float C1, float C2, float* pmax, int* phasnan,
                        double* psum, const ShiftArg* on_the_fly = nullptr, bool shared_x = false) {
    Five P1, P2, P3;
    for (int a = 0; a < 5; ++a) { P1.src[a] = nullptr; P1.dst[a] = setA[a]; }
    if (shared_x) {      // 3D, region = whole volume: only the candidate's terms are filtered, x / xx come from setB[2], setB[3]
        if (on_the_fly)
            hipLaunchKernelGGL((ssim_first_pass_kernel<WIN, true, 2>), dim3(kStatBlocks), dim3(256), 0, stream, im0, im1t, S, lo[0], lo[1], lo[2], R,
                               0, P1, pmax, phasnan, *on_the_fly);
        else
            hipLaunchKernelGGL((ssim_first_pass_kernel<WIN, false, 2>), dim3(kStatBlocks), dim3(256), 0, stream, im0, im1t, S, lo[0], lo[1], lo[2], R,
                               0, P1, pmax, phasnan, ShiftArg{0.0, 0.0, 0.0, 0});
        for (int a = 0; a < 5; ++a) { P2.src[a] = setA[a]; P2.dst[a] = nullptr; }
        P2.dst[0] = setB[2];
        P2.dst[2] = setB[3];
        hipLaunchKernelGGL((ssim_yx_fused_kernel<WIN, 2>), dim3(kStatBlocks), dim3(256), 0, stream, P2, R, cov_norm, C1, C2, psum);
        return;
    }
    if (on_the_fly)      // im1t is the UNSHIFTED moving image
        hipLaunchKernelGGL((ssim_first_pass_kernel<WIN, true>), dim3(kStatBlocks), dim3(256), 0, stream, im0, im1t, S, lo[0], lo[1], lo[2], R,
                           ndim == 3 ? 0 : 1, P1, pmax, phasnan, *on_the_fly);
    else
        hipLaunchKernelGGL((ssim_first_pass_kernel<WIN, false>), dim3(kStatBlocks), dim3(256), 0, stream, im0, im1t, S, lo[0], lo[1], lo[2], R,
                           ndim == 3 ? 0 : 1, P1, pmax, phasnan, ShiftArg{0.0, 0.0, 0.0, 0});
    float* const* last_src = setA;
    if (ndim == 3) {
        for (int a = 0; a < 5; ++a) { P2.src[a] = setA[a]; P2.dst[a] = nullptr; }
        hipLaunchKernelGGL(ssim_yx_fused_kernel<WIN>, dim3(kStatBlocks), dim3(256), 0, stream, P2, R, cov_norm, C1, C2, psum);
        return;
    }
    for (int a = 0; a < 5; ++a) { P3.src[a] = last_src[a]; P3.dst[a] = nullptr; }
    hipLaunchKernelGGL(ssim_last_pass_kernel<WIN>, dim3(kStatBlocks), dim3(256), 0, stream, P3, R, ndim, cov_norm, C1, C2, psum);
}

}  // namespace

extern "C" int mvs_score_candidates(int device, const float* fixed, const float* moving, int32_t mem, int32_t ndim,
                                    const int64_t shape[3], const double* t_candidates, int32_t n_candidates,
                                    int32_t region_mode, double data_range, double im1_min, int32_t quality_for_all,
                                    double* ssim_out, double* spearman_out, int32_t* code_out) {
    return mvs_score_candidates_impl(device, fixed, moving, mem, ndim, shape, t_candidates, n_candidates, region_mode, data_range, im1_min,
                                     quality_for_all, ssim_out, spearman_out, code_out, MvsScoreOpts{});
}

int mvs_score_candidates_impl(int device, const float* fixed, const float* moving, int32_t mem, int32_t ndim, const int64_t shape[3],
                              const double* t_candidates, int32_t n_candidates, int32_t region_mode, double data_range, double im1_min,
                              int32_t quality_for_all, double* ssim_out, double* spearman_out, int32_t* code_out, const MvsScoreOpts& so) {
    MvsContext* c;
    int rc = mvs_check_ready(device, &c);
    if (rc) return rc;
    std::lock_guard<std::recursive_mutex> lock(c->mu);
    if (!fixed || !moving || !shape || !t_candidates || !ssim_out || !spearman_out || !code_out || n_candidates < 0)
        return mvs_fail(c, MVS_ERR_INVALID_ARG, "mvs_score_candidates: bad argument");
    if (ndim != 2 && ndim != 3) return mvs_fail(c, MVS_ERR_INVALID_ARG, "mvs_score_candidates: ndim must be 2 or 3");
    if (ndim == 2 && shape[0] != 1) return mvs_fail(c, MVS_ERR_INVALID_ARG, "mvs_score_candidates: 2D needs shape[0]==1");
    MVS_HIP_TRY(c, hipSetDevice(mvs_hip_device(device)));
    const Shape3 S = {(int)shape[0], (int)shape[1], (int)shape[2]};
    const long long n = (long long)S.nz * S.ny * S.nx;
    if (n >= (1ll << 31) - 8) return mvs_fail(c, MVS_ERR_UNSUPPORTED, "mvs_score_candidates: volume too large");
    const int k0 = 3 - ndim;

    float *im0, *im1;
    rc = mvs_stage_float_volume(c, fixed, mem, n, 4, &im0);
    if (rc) return rc;
    rc = mvs_stage_float_volume(c, moving, mem, n, 5, &im1);
    if (rc) return rc;

    // Candidates that survive the analytic pre-test are evaluated in batches: all their shifted copies of the
    // moving image stay resident (one host round trip for the masks / boxes of the whole batch, a second one
    // for the SSIM sums), and the winner's copy is still there for the rank correlation.
    const int nres = (int)std::max<long long>(1, std::min<long long>(std::min(kMaxResident, std::max(n_candidates, 1)),
                                                                     (4ll << 30) / (n * 4)));
    size_t sort_temp_bytes = 0;
    MVS_HIP_TRY(c, rocprim::radix_sort_pairs(nullptr, sort_temp_bytes, (float*)nullptr, (float*)nullptr, (float*)nullptr,
                                            (float*)nullptr, (size_t)n, 0, 32, c->stream));
    {
        size_t tb16 = 0;
        MVS_HIP_TRY(c, rocprim::radix_sort_pairs(nullptr, tb16, (unsigned int*)nullptr, (unsigned int*)nullptr, (float*)nullptr,
                                                (float*)nullptr, (size_t)n, 0, 16, c->stream));
        sort_temp_bytes = std::max(sort_temp_bytes, tb16);
    }
    const int gb = grid_for(n);
    const size_t stat_bytes = (size_t)(kMaxResident + 2) * kStatBlocks * (sizeof(VoxStats) + 4 + 4 + 8) + (kMaxResident + 2) * 64;
    // batched launches (one z pass / one y-x pass for all candidates of a batch) keep three z-filtered arrays per candidate
    const bool may_batch = ndim == 3 && region_mode == 0 && !quality_for_all && !c->materialize_shifts && (long long)n * 12 * nres <= (3ll << 30);
    constexpr int kMaxCls = 4;      // fraction classes of half-pixel shifts that get ONE shifted copy shared by their candidates
    const size_t need = (size_t)n * 4 * (10 + nres + (may_batch ? 3 * nres + kMaxCls : 0)) + 256 * (12 + 4 * nres + kMaxCls) + sort_temp_bytes + (size_t)gb * 32 + stat_bytes + 64 * 1024 + (size_t)(kHistBinsMax + 64) * 8 + (size_t)kHistParts * (kHistBinsMax / 2 + 1) * 4 + 2048;
    char* base = (char*)mvs_scratch(c, 6, need);
    if (!base) return mvs_alloc_failed(c);
    DeviceBump B{base, need, 0};
    std::vector<float*> im1t_buf(nres);
    for (int i = 0; i < nres; ++i) im1t_buf[i] = B.take<float>(n);
    float* setA[5]; float* setB[5];
    for (int a = 0; a < 5; ++a) setA[a] = B.take<float>(n);
    for (int a = 0; a < 5; ++a) setB[a] = B.take<float>(n);
    std::vector<float*> cand3((size_t)(may_batch ? 3 * nres : 0));
    for (float*& q : cand3) q = B.take<float>(n);
    // In 3D the phase correlation refines to half pixels, so the candidates of a pair -- t, -t, -(t - N), -t - N per axis -- share the
    // fractional part of their shift per axis, and every candidate image is an INTEGER shift of one "fraction-only" shifted copy of the
    // moving image (same taps, weights and order: c = o + t is exact in double).  One copy per fraction class instead of one per
    // candidate; the fused SSIM walk reads it at o + floor(t) exactly as it reads the moving image itself for integer shifts.
    struct ClsBuf { int key; float* buf; } cls[kMaxCls];
    int n_cls = 0;
    for (int k = 0; k < kMaxCls; ++k) cls[k] = ClsBuf{-1, may_batch ? B.take<float>(n) : nullptr};
    void* sort_temp = B.take<char>(sort_temp_bytes);
    double* partial = B.take<double>((size_t)gb * 4);
    VoxStats* vox_partial = B.take<VoxStats>((size_t)(kMaxResident + 2) * kStatBlocks);
    VoxStats* vox_out = B.take<VoxStats>(kMaxResident + 2);
    float* pmax = B.take<float>((size_t)kMaxResident * kStatBlocks);
    int* phasnan = B.take<int>((size_t)kMaxResident * kStatBlocks);
    double* psum = B.take<double>((size_t)kMaxResident * kStatBlocks);
    RegionStats* reg_out = B.take<RegionStats>(kMaxResident);
    unsigned int* d_hist = B.take<unsigned int>((size_t)kHistBinsMax + 64);     // key histograms of the rank correlation
    float* d_rank = B.take<float>((size_t)kHistBinsMax + 64);
    unsigned int* d_parts = B.take<unsigned int>((size_t)kHistParts * (kHistBinsMax / 2 + 1));   // per-workgroup packed histograms
    unsigned int* d_counter = B.take<unsigned int>(64);
    if (!d_counter) return mvs_fail(c, MVS_ERR_HIP, "mvs_score_candidates: scratch layout");
    // results the host reads (reduction partials of the rank correlation, voxel and region statistics) are written by the kernels
    // straight into the context's mailbox (pinned host memory): no copy launches
    const size_t mb_partial = 0, mb_vox = ((size_t)gb * 4 * sizeof(double) + 255) / 256 * 256;
    const size_t mb_reg = mb_vox + ((size_t)(kMaxResident + 2) * sizeof(VoxStats) + 255) / 256 * 256;
    void *mb_host = nullptr, *mb_dev = nullptr;
    {
        const int rcm = mvs_mailbox(c, mb_reg + (size_t)kMaxResident * sizeof(RegionStats), &mb_host, &mb_dev);
        if (rcm) return rcm;
    }
    partial = (double*)((char*)mb_dev + mb_partial);
    vox_out = (VoxStats*)((char*)mb_dev + mb_vox);
    reg_out = (RegionStats*)((char*)mb_dev + mb_reg);
    const double* h_partial = (const double*)((const char*)mb_host + mb_partial);
    const VoxStats* h_vox = (const VoxStats*)((const char*)mb_host + mb_vox);
    const RegionStats* h_reg = (const RegionStats*)((const char*)mb_host + mb_reg);

    // valid voxels of im1 and the bboxes of both images (registration.py:400, 491)
    VoxStats h_im[2];
    if (so.both_crops_finite) {
        // the caller (mvs_register_crops) has just reduced both images and found neither NaN nor inf: every voxel is valid,
        // the boxes are the whole volume -- no reduction, no host round trip
        for (int k = 0; k < 2; ++k) {
            h_im[k].cnt = (unsigned long long)n;
            h_im[k].bb[0] = h_im[k].bb[1] = h_im[k].bb[2] = 0;
            h_im[k].bb[3] = S.nz - 1; h_im[k].bb[4] = S.ny - 1; h_im[k].bb[5] = S.nx - 1;
        }
    } else {
        hipLaunchKernelGGL(image_stats_kernel, dim3(kStatBlocks), dim3(256), 0, c->stream, im0, S, vox_partial);
        hipLaunchKernelGGL(image_stats_kernel, dim3(kStatBlocks), dim3(256), 0, c->stream, im1, S, vox_partial + kStatBlocks);
        hipLaunchKernelGGL(finish_voxstats_kernel, dim3(2), dim3(256), 0, c->stream, vox_partial, vox_out);
        MVS_HIP_TRY(c, hipStreamSynchronize(c->stream));
        h_im[0] = h_vox[0];
        h_im[1] = h_vox[1];
    }
    const int* bb0 = h_im[0].bb;
    const int* bbm = h_im[1].bb;
    const unsigned int valid1 = (unsigned int)(h_im[1].cnt & 0xffffffffull);
    // every voxel of the moving image finite?  (lower half: #non-NaN, upper half: #inf, see image_stats_kernel)
    const int im1_all_finite = ((long long)valid1 == n && (h_im[1].cnt >> 32) == 0) ? 1 : 0;
    const int im0_all_finite = ((long long)(h_im[0].cnt & 0xffffffffull) == n && (h_im[0].cnt >> 32) == 0) ? 1 : 0;
    // Both crops finite (tiles on a common grid): the valid box of a shifted copy and with it the mask count are known
    // without touching the volume -- x is valid iff 0 <= fl(x + t) <= n - 1 per axis -- so the reduction of phase A and its
    // host round trip are skipped, and candidates with integer shifts are never materialised: the SSIM z pass reads the
    // moving image at the shifted position (the winner's copy is written afterwards, for the rank correlation).
    const bool on_the_fly = !quality_for_all && im0_all_finite && im1_all_finite && !c->materialize_shifts;

    MVS_HIP_TRY(c, hipEventRecord(c->ev_start, c->stream));
    // analytic pre-test: upper bound of the mask count from the valid bounding boxes -- im1t can only be valid
    // where x + t lies in im1's valid box.  If even the bound fails the 10 % test the candidate is rejected
    // exactly as the reference rejects it (registration.py:503-505) without touching the volume.
    std::vector<int> todo;
    for (int ic = 0; ic < n_candidates; ++ic) {
        double t[3] = {0.0, 0.0, 0.0};
        for (int k = 0; k < ndim; ++k) t[k0 + k] = t_candidates[ic * ndim + k];
        ssim_out[ic] = -1.0;
        spearman_out[ic] = -1.0;
        code_out[ic] = 0;
        double bound = 1.0;
        const int dimv[3] = {S.nz, S.ny, S.nx};
        for (int k = 0; k < 3; ++k) {
            const double lo1 = std::ceil((double)bbm[k] - t[k] - 1.0), hi1 = std::floor((double)bbm[3 + k] - t[k] + 1.0);
            const double lo = std::max(std::max(lo1, (double)bb0[k]), 0.0);
            const double hi = std::min(std::min(hi1, (double)bb0[3 + k]), (double)(dimv[k] - 1));
            bound *= std::max(hi - lo + 1.0, 0.0);
        }
        if (valid1 == 0 || bound == 0.0 || bound / (double)valid1 < 0.1) code_out[ic] = 1;
        else todo.push_back(ic);
    }

    std::vector<unsigned long long> cnts((size_t)std::max(n_candidates, 1), 0ull);
    std::vector<int> resident((size_t)std::max(n_candidates, 1), -1);   // buffer holding the candidate's im1t, if still there
    const float Rf = (float)data_range;
    // (K1 * R) ** 2 with R a float32 scalar: numpy keeps this in float32
    const float C1 = (0.01f * Rf) * (0.01f * Rf);
    const float C2 = (0.03f * Rf) * (0.03f * Rf);

    // Spearman over the jointly valid voxels of candidate ic, whose shifted image is `im1t`: compaction, sort by x
    // carrying y, ranks of x in sorted order, sort by y carrying rank(x), correlation sums in y-sorted order
    // im1t == nullptr: the candidate has no shifted copy of its own; it is made (into im1t_buf[0]) only if the sorting path needs it
    auto spearman_from = [&](int ic, const float* im1t) -> int {
        {
            // histogram ranks: both crops hold 16-bit integers (the caller vouches: raw_u16_keys) and are finite, every
            // component of this candidate's shift is a multiple of 1/2
            double t[3] = {0.0, 0.0, 0.0};
            bool halves = true;
            for (int k = 0; k < ndim; ++k) {
                t[k0 + k] = t_candidates[ic * ndim + k];
                halves = halves && (std::floor(t[k0 + k] * 2.0) == t[k0 + k] * 2.0);
            }
            int nf = 0;
            for (int k = 0; k < 3; ++k) nf += (std::floor(t[k]) != t[k]) ? 1 : 0;
            // key ranges from the raw extrema of the crops (so.raw_range: min / max of the fixed and of the moving crop)
            const long long kx0 = (long long)so.raw_range[0], nbx = (long long)so.raw_range[1] - kx0 + 1;
            const long long ky0 = (long long)so.raw_range[2] * (1 << nf), nby = ((long long)so.raw_range[3] - (long long)so.raw_range[2]) * (1 << nf) + 1;
            // < 65536 voxels per workgroup; up to kHistParts workgroups write their histograms out whole (folded by a
            // second kernel), beyond that the non-zero counters are flushed with atomics
            const long long hneed = ((long long)n / 4 + 16382) / 16383;
            const bool fold = hneed <= kHistParts;
            const long long hgb = fold ? std::max<long long>(hneed, std::min<long long>(kHistParts, ((long long)n + 8191) / 8192)) : std::max<long long>(gb, hneed);
            if (halves && so.raw_u16_keys[0] && so.raw_u16_keys[1] && so.both_crops_finite && !c->materialize_shifts && nbx > 0 && nby > 0 &&
                nbx + nby <= kHistBinsMax && hgb <= 65535) {
                static bool lds_attr[MVS_MAX_DEVICES] = {false};
                if (!lds_attr[mvs_hip_device(device)]) {
                    MVS_HIP_TRY(c, hipFuncSetAttribute((const void*)hist_rank_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024));
                    lds_attr[mvs_hip_device(device)] = true;
                }
                const size_t lds_bytes = (size_t)((nbx + nby + 1) / 2) * 4;
                if (!fold) MVS_HIP_TRY(c, hipMemsetAsync(d_hist, 0, sizeof(unsigned int) * (size_t)(nbx + nby), c->stream));
                MVS_DUP("hist_count", hipLaunchKernelGGL(hist_rank_kernel<false>, dim3((unsigned)hgb), dim3(1024), lds_bytes, c->stream, so.raw_u16_keys[0], so.raw_u16_keys[1],
                                   S, t[0], t[1], t[2], (int)kx0, (int)nbx, (int)ky0, (int)nby, d_hist, d_hist + nbx, (const float*)nullptr,
                                   (const float*)nullptr, (double*)nullptr, fold ? d_parts : (unsigned int*)nullptr));
                if (fold) {
                    const int nwords = (int)((nbx + nby + 1) / 2);
                    MVS_DUP("hist_fold", hipLaunchKernelGGL(hist_fold_kernel, dim3((nwords + 63) / 64), dim3(1024), 0, c->stream, d_parts, (int)hgb, nwords, (int)nbx, (int)nby,
                                       d_hist, d_hist + nbx));
                }
                MVS_DUP("rank_table", hipLaunchKernelGGL(rank_table_kernel, dim3(2), dim3(1024), 0, c->stream, d_hist, (int)nbx, d_rank, d_hist + nbx, (int)nby,
                                   d_rank + nbx, partial));
                MVS_DUP("hist_corr", hipLaunchKernelGGL(hist_rank_kernel<true>, dim3(gb), dim3(256), 0, c->stream, so.raw_u16_keys[0], so.raw_u16_keys[1], S, t[0], t[1],
                                   t[2], (int)kx0, (int)nbx, (int)ky0, (int)nby, d_hist, d_hist + nbx, d_rank, d_rank + nbx, partial + 4, (unsigned int*)nullptr));
                MVS_HIP_TRY(c, hipGetLastError());
                MVS_HIP_TRY(c, hipStreamSynchronize(c->stream));
                const double* hp = h_partial;
                double sxy = 0;
                for (int i = 0; i < gb; ++i) sxy += hp[4 + i];
                spearman_out[ic] = sxy / std::sqrt(hp[0] * hp[2]);
                return MVS_OK;
            }
        }
        if (!im1t) {
            double t[3] = {0.0, 0.0, 0.0};
            for (int k = 0; k < ndim; ++k) t[k0 + k] = t_candidates[ic * ndim + k];
            hipLaunchKernelGGL(shift_kernel, dim3(kStatBlocks), dim3(256), 0, c->stream, im1, im0, im1t_buf[0], S, t[0], t[1], t[2], im1_all_finite, vox_partial);
            std::fill(resident.begin(), resident.end(), -1);
            resident[ic] = 0;
            im1t = im1t_buf[0];
        }
        MVS_HIP_TRY(c, hipMemsetAsync(d_counter, 0, 4, c->stream));
        const float* raw0 = so.raw_u16_keys[0];      // 16-bit integer keys for the fixed image when the caller vouches for them
        hipLaunchKernelGGL(compact_kernel, dim3(gb), dim3(256), 0, c->stream, im0, im1t, n, setA[0], setA[1], d_counter, raw0);
        const unsigned int m = (unsigned int)cnts[ic];
        const int mgb = (int)std::min<long long>(((long long)m + kRankChunk - 1) / kRankChunk, 2048);   // one chunk of sorted keys per workgroup turn
        if (raw0) {
            MVS_HIP_TRY(c, rocprim::radix_sort_pairs(sort_temp, sort_temp_bytes, (unsigned int*)setA[0], (unsigned int*)setA[2], setA[1], setA[3], (size_t)m,
                                                    0, 16, c->stream));
            hipLaunchKernelGGL(ranks_sorted_kernel<unsigned int>, dim3(mgb), dim3(256), 0, c->stream, (const unsigned int*)setA[2], m, setA[4]);
        } else {
            MVS_HIP_TRY(c, rocprim::radix_sort_pairs(sort_temp, sort_temp_bytes, setA[0], setA[2], setA[1], setA[3], (size_t)m, 0, 32, c->stream));
            hipLaunchKernelGGL(ranks_sorted_kernel<float>, dim3(mgb), dim3(256), 0, c->stream, setA[2], m, setA[4]);
        }
        MVS_HIP_TRY(c, rocprim::radix_sort_pairs(sort_temp, sort_temp_bytes, setA[3], setB[0], setA[4], setB[1], (size_t)m, 0, 32, c->stream));
        hipLaunchKernelGGL(rankcorr_kernel, dim3(mgb), dim3(256), 0, c->stream, setB[0], setB[1], m, 0.5 * ((double)m + 1.0), partial);
        MVS_HIP_TRY(c, hipGetLastError());
        MVS_HIP_TRY(c, hipStreamSynchronize(c->stream));
        const double* hp = h_partial;
        double sxy = 0, sxx = 0, syy = 0;
        for (int i = 0; i < mgb; ++i) { sxy += hp[i * 3]; sxx += hp[i * 3 + 1]; syy += hp[i * 3 + 2]; }
        spearman_out[ic] = sxy / std::sqrt(sxx * syy);
        return MVS_OK;
    };

    // 3D, fixed image finite, "union" regions: the SSIM region of every candidate is the whole volume, so the fixed image's own
    // window means (x, xx: 2 of the 5 filtered quantities) are computed once for the pair instead of once per candidate
    bool shared_x = false;
    if (ndim == 3 && region_mode == 0 && im0_all_finite && todo.size() >= 2 && std::min(S.nz, std::min(S.ny, S.nx)) >= 7 &&
        !c->materialize_shifts) {
        shared_x = true;
        launch_ssim_shared_x<7>(c->stream, im0, S, S, setA, setB, pmax, phasnan, !c->ssim_two_pass && (long long)S.nz * S.ny * S.nx * 4 < (1ll << 31));
    }
    // slot of the fraction class of a shift whose components are all multiples of 1/2 (bit k of the key: axis k has the fraction
    // 1/2), -1 when the candidate keeps a copy of its own (other fractions, integer shift, no slot left, mode off); *fresh: the
    // class copy still has to be written
    auto share_cls = [&](const double t[3], bool* fresh) -> int {
        *fresh = false;
        if (!may_batch || !im1_all_finite || c->materialize_shifts || c->ssim_two_pass) return -1;
        int key = 0;
        for (int k = 0; k < 3; ++k) {
            const double t2 = t[k] * 2.0;
            if (!(std::floor(t2) == t2 && std::fabs(t[k]) < 1e6)) return -1;
            key |= (t[k] != std::floor(t[k])) ? (1 << k) : 0;
        }
        if (key == 0) return -1;
        for (int q = 0; q < n_cls; ++q)
            if (cls[q].key == key) return q;
        if (n_cls == kMaxCls) return -1;
        cls[n_cls].key = key;
        *fresh = true;
        return n_cls++;
    };
    for (size_t b0 = 0; b0 < todo.size(); b0 += (size_t)nres) {
        const int nb = (int)std::min<size_t>((size_t)nres, todo.size() - b0);
        std::fill(resident.begin(), resident.end(), -1);
        // ---- phase A: shifted copies + mask counts / bboxes of the whole batch ----
        VoxStats h_vs[kMaxResident];
        ShiftArg shifts[kMaxResident];
        bool otf[kMaxResident] = {};
        int cls_of[kMaxResident];      // fraction class whose copy the candidate reads (-1: its own copy / the moving image)
        for (int j = 0; j < kMaxResident; ++j) cls_of[j] = -1;
        const bool batched = on_the_fly && shared_x && may_batch;
        ShiftBatch shift_batch;
        int n_shift_batch = 0;
        for (int j = 0; j < nb; ++j) {
            const int ic = todo[b0 + j];
            double t[3] = {0.0, 0.0, 0.0};
            for (int k = 0; k < ndim; ++k) t[k0 + k] = t_candidates[ic * ndim + k];
            shifts[j] = ShiftArg{t[0], t[1], t[2], im1_all_finite};
            if (on_the_fly) {
                // the predicate of axis_tap, evaluated on the host in the same double arithmetic: first and last valid index
                const int dims[3] = {S.nz, S.ny, S.nx};
                unsigned long long cnt = 1;
                for (int k = 0; k < 3; ++k) {
                    auto ok = [&](long long x) { const double cc = (double)x + t[k]; return !(cc < 0.0 || cc > (double)(dims[k] - 1)); };
                    long long lo = (long long)std::ceil(-t[k]), hi = (long long)std::floor((double)(dims[k] - 1) - t[k]);
                    lo = std::min<long long>(std::max<long long>(lo, 0), dims[k]);
                    hi = std::max<long long>(std::min<long long>(hi, dims[k] - 1), -1);
                    while (lo > 0 && ok(lo - 1)) --lo;
                    while (lo < dims[k] && !ok(lo)) ++lo;
                    while (hi < dims[k] - 1 && ok(hi + 1)) ++hi;
                    while (hi >= 0 && !ok(hi)) --hi;
                    h_vs[j].bb[k] = (int)lo;
                    h_vs[j].bb[3 + k] = (int)hi;
                    cnt *= (unsigned long long)std::max<long long>(hi - lo + 1, 0);
                }
                h_vs[j].cnt = cnt;
                // integer shifts: one tap of weight 1 per voxel -- the z pass reads the moving image directly.  Fractional
                // shifts keep their shifted copy (its 2-8 double-precision taps per voxel would be re-evaluated 1.75 times
                // by the windowed z pass), but nobody waits for its statistics.
                otf[j] = t[0] == std::floor(t[0]) && t[1] == std::floor(t[1]) && t[2] == std::floor(t[2]);
                if (otf[j]) continue;
                if (batched) {
                    bool fresh = false;
                    const int q = share_cls(t, &fresh);
                    if (q >= 0) {
                        cls_of[j] = q;
                        if (fresh) {      // the class copy: the fraction-only shift, through the same launch as the other copies
                            const int key = cls[q].key;
                            ShiftCand sc{cls[q].buf, 0.5 * (key & 1), 0.5 * ((key >> 1) & 1), 0.5 * ((key >> 2) & 1), 1,
                                         HalfShift{0, 0, 0, key & 1, (key >> 1) & 1, (key >> 2) & 1}};
                            shift_batch.c[n_shift_batch++] = sc;
                        }
                        continue;
                    }
                }
                if (batched) {      // all fractional shifts of the batch in one launch, after this loop
                    ShiftCand sc{im1t_buf[j], t[0], t[1], t[2], 0, HalfShift{0, 0, 0, 0, 0, 0}};
                    if (!c->materialize_shifts) {
                        bool half = true;
                        int f[3], h[3];
                        for (int k = 0; k < 3; ++k) {
                            const double t2 = t[k] * 2.0;
                            half = half && std::floor(t2) == t2 && std::fabs(t[k]) < 1e6;
                            f[k] = (int)std::floor(t[k]);
                            h[k] = (t[k] != std::floor(t[k])) ? 1 : 0;
                        }
                        if (half) { sc.half = 1; sc.H = HalfShift{f[0], f[1], f[2], h[0], h[1], h[2]}; }
                    }
                    shift_batch.c[n_shift_batch++] = sc;
                    resident[ic] = j;
                    continue;
                }
            }
            hipLaunchKernelGGL(shift_kernel, dim3(kStatBlocks), dim3(256), 0, c->stream, im1, im0, im1t_buf[j], S, t[0], t[1], t[2],
                               im1_all_finite, vox_partial + (size_t)j * kStatBlocks);
            resident[ic] = j;
        }
        if (n_shift_batch)
            MVS_DUP("shift", hipLaunchKernelGGL(shift_batch_kernel, dim3(kStatBlocks, n_shift_batch), dim3(256), 0, c->stream, im1, im0, S, shift_batch, im1_all_finite));
        if (!on_the_fly) {
            hipLaunchKernelGGL(finish_voxstats_kernel, dim3(nb), dim3(256), 0, c->stream, vox_partial, vox_out);
            MVS_HIP_TRY(c, hipStreamSynchronize(c->stream));
            for (int j = 0; j < nb; ++j) h_vs[j] = h_vox[j];
        }

        // ---- phase B: SSIM passes of every candidate that keeps enough jointly valid voxels ----
        Shape3 Rs[kMaxResident];
        int wins[kMaxResident];
        bool scored[kMaxResident];
        FirstBatch first_batch;
        YxBatch yx_batch;
        FusedBatch fused_batch;
        bool any_batched = false;
        float batch_cov_norm = 0.f;
        for (int j = 0; j < nb; ++j) {
            first_batch.c[j] = FirstCand{nullptr, nullptr, nullptr, nullptr, ShiftArg{0.0, 0.0, 0.0, 0}, 0};
            yx_batch.c[j] = YxCand{nullptr, nullptr, nullptr};
            fused_batch.c[j] = FusedCand{nullptr, 0, 0, 0, 0xffffffffu};
        }
        for (int j = 0; j < nb; ++j) {
            const int ic = todo[b0 + j];
            const unsigned long long cnt = h_vs[j].cnt;
            const int* bb1 = h_vs[j].bb;
            scored[j] = false;
            wins[j] = 0;
            Rs[j] = {0, 0, 0};
            cnts[ic] = cnt;
            if (cnt == 0 || (double)cnt / (double)valid1 < 0.1) {   // registration.py:503-505
                code_out[ic] = 1;
                continue;
            }
            // region slices (registration.py:509-528)
            int lo[3], hi[3];
            for (int k = 0; k < 3; ++k) {
                if (region_mode == 0) { lo[k] = std::min(bb0[k], bb1[k]); hi[k] = std::max(bb0[3 + k], bb1[3 + k]) + 1; }
                else { lo[k] = std::max(bb0[k], bb1[k]); hi[k] = std::min(bb0[3 + k], bb1[3 + k]) + 1; }
            }
            const Shape3 R = {std::max(hi[0] - lo[0], 0), std::max(hi[1] - lo[1], 0), std::max(hi[2] - lo[2], 0)};
            Rs[j] = R;
            const long long rn = (long long)R.nz * R.ny * R.nx;
            if (rn <= 0) continue;
            int min_shape = 0x7fffffff;
            for (int k = k0; k < 3; ++k) min_shape = std::min(min_shape, (&R.nz)[k]);
            int win = std::min(7, min_shape - ((min_shape - 1) % 2));
            if (win < 3) win = 3;   // SSIM is -1 then (decided below); the pass still yields the region statistics
            wins[j] = win;
            double NP = 1.0;
            for (int k = 0; k < ndim; ++k) NP *= (double)win;
            const float cov_norm = (float)(NP / (NP - 1.0));
            float* pm = pmax + (size_t)j * kStatBlocks;
            int* ph = phasnan + (size_t)j * kStatBlocks;
            double* ps = psum + (size_t)j * kStatBlocks;
            const float* second = otf[j] ? im1 : im1t_buf[j];
            const ShiftArg* sa = otf[j] ? &shifts[j] : nullptr;
            const bool full = R.nz == S.nz && R.ny == S.ny && R.nx == S.nx;
            if (cls_of[j] >= 0 && !(batched && win == 7 && full)) {
                // (cannot happen for a whole-volume region with a 7-wide window; keeps the single-candidate paths below whole)
                hipLaunchKernelGGL(shift_kernel, dim3(kStatBlocks), dim3(256), 0, c->stream, im1, im0, im1t_buf[j], S, shifts[j].tz, shifts[j].ty,
                                   shifts[j].tx, im1_all_finite, vox_partial + (size_t)j * kStatBlocks);
                resident[ic] = j;
                cls_of[j] = -1;
            }
            if (batched && win == 7 && full) {     // joins the two batched launches below
                float* d1 = cand3[(size_t)3 * j], *d3 = cand3[(size_t)3 * j + 1], *d4 = cand3[(size_t)3 * j + 2];
                first_batch.c[j] = FirstCand{second, d1, d3, d4, shifts[j], otf[j] ? 1 : 0};
                yx_batch.c[j] = YxCand{d1, d3, d4};
                fused_batch.c[j] = otf[j] ? FusedCand{im1, (int)shifts[j].tz, (int)shifts[j].ty, (int)shifts[j].tx, 0xffffffffu}
                                   : cls_of[j] >= 0 ? FusedCand{cls[cls_of[j]].buf, (int)std::floor(shifts[j].tz), (int)std::floor(shifts[j].ty), (int)std::floor(shifts[j].tx), 0xffffffffu}
                                                    : FusedCand{im1t_buf[j], 0, 0, 0, 0xffffffffu};
                any_batched = true;
                batch_cov_norm = cov_norm;
            }
            else if (win == 7 && shared_x && full) launch_ssim_passes<7>(c->stream, im0, second, S, lo, R, ndim, setA, setB, cov_norm, C1, C2, pm, ph, ps, sa, true);
            else if (win == 7) launch_ssim_passes<7>(c->stream, im0, second, S, lo, R, ndim, setA, setB, cov_norm, C1, C2, pm, ph, ps, sa);
            else if (win == 5) launch_ssim_passes<5>(c->stream, im0, second, S, lo, R, ndim, setA, setB, cov_norm, C1, C2, pm, ph, ps, sa);
            else launch_ssim_passes<3>(c->stream, im0, second, S, lo, R, ndim, setA, setB, cov_norm, C1, C2, pm, ph, ps, sa);
            scored[j] = true;
        }
        RegionStats h_rs[kMaxResident];
        bool have_rs = false;
        if (any_batched && !c->ssim_two_pass) {
            const int tiles = ((S.ny - 6 + 15) / 16) * ((S.nx - 6 + 55) / 56), cz = S.nz - 6;
            // ---- pruned argmax search (the caller needs the arg-max candidate only: mvs_register_crops) -------------------------------
            // The SSIM of a candidate is the mean of per-voxel values S <= 1 (S = l * cs with l <= 1 by the AM-GM inequality and
            // |cs| <= 1 by Cauchy-Schwarz; the float32 window means move a variance by at most a few 2^-23 M^2, M the largest value,
            // against C2 = (0.03 R)^2 in the denominator: S <= 1 + slack with the slack below).  So once ONE candidate is scored
            // completely (sum S*), a candidate with partial sum p over n of the N voxels can at best reach p + (N - n)(1 + slack); if
            // that is below S* it cannot be the arg max, whatever the rest of its volume holds -- the reference's nanargmax picks the
            // same candidate, and the Spearman coefficient is only ever evaluated for that one.  All candidates are walked on 1 / 32 of
            // the work items (spread over the volume), the leader is completed, the others continue in rounds only while their bound
            // still reaches the best complete sum (see the plan inside the loop).  On the bench mosaic the decorrelated candidates
            // (mean 0.01-0.15 against 0.90-0.975) leave after 3/32-8/32 of their volume, the sign flips of a half-pixel axis (0.6-0.94)
            // after 6/32-22/32: 2.6 instead of 9.2 candidate volumes per pair (profiles/round4_prune_ab.txt).
            bool prune = so.argmax_only && c->ssim_prune && todo.size() <= (size_t)nres;
            int n_in = 0;
            for (int j = 0; j < nb; ++j) {
                if (fused_batch.c[j].src) ++n_in;
                else if (scored[j]) prune = false;          // a candidate on the separate passes: everything is scored in full
            }
            const double vb = so.value_bound;
            const double slack = 1e-2 * std::max(1.0, (vb / data_range) * (vb / data_range));
            prune = prune && n_in >= 2 && data_range > 0.0 && std::isfinite(slack) && slack <= 0.05;
            if (!prune) {
                // ~1536 workgroups (two resident rounds of 3 per CU): 243 instead of 282 us per pair with 768 -- a workgroup spends its
                // time waiting (two barriers and a load round trip per plane), so a second round hides more than its 6 halo planes cost
                const int nzs = std::max(1, std::min(1536 / std::max(tiles * nb, 1), (cz + 7) / 8));
                MVS_DUP("ssim_fused", hipLaunchKernelGGL(ssim_fused_batch_kernel<7>, dim3(kStatBlocks, nb), dim3(256), 0, c->stream, im0, S, fused_batch, setB[2], setB[3],
                                   (cz + nzs - 1) / nzs, batch_cov_norm, C1, C2, pmax, phasnan, psum, 32));
                c->reg_cand_volumes += (double)n_in;
            } else {
                const int cy = S.ny - 6, cx = S.nx - 6, nty = (cy + 15) / 16, ntx = (cx + 55) / 56;
                static const int target_items = [] { const char* e = getenv("MVS_SSIM_PRUNE_ITEMS"); return (e && atoi(e) > 0) ? atoi(e) : 320; }();
                const int nzs0 = std::max(1, std::min(target_items / std::max(tiles, 1), (cz + 7) / 8));
                const int zseg = (cz + nzs0 - 1) / nzs0, nzs = (cz + zseg - 1) / zseg, nitems = nty * ntx * nzs;
                // residue classes of the work items (a candidate's volume is walked in K-ths): 32 (measured against 16: 2.60 instead of
                // 2.84 candidate volumes per pair on the bench mosaic); MVS_SSIM_PRUNE_CLASSES=16 for the A/B
                static const int K = [] { const char* e = getenv("MVS_SSIM_PRUNE_CLASSES"); return (e && atoi(e) == 16) ? 16 : 32; }();
                const unsigned int kAll = K == 32 ? 0xffffffffu : 0xffffu;
                double vol_res[32];                     // output voxels of the work items of every residue class (the kernel's own geometry)
                for (int r = 0; r < 32; ++r) vol_res[r] = 0.0;
                for (int item = 0; item < nitems; ++item) {
                    const int tx = item % ntx, ty = (item / ntx) % nty, zs = item / (ntx * nty);
                    const int z0 = 3 + zs * zseg, z1 = std::min(z0 + zseg, S.nz - 3);
                    const int res = (((item % K) - kSelRot * (item / K)) % K + K) % K;      // the residue class whose member this item is (kSelRot)
                    vol_res[res] += (double)(z1 - z0) * (double)std::min(16, cy - ty * 16) * (double)std::min(56, cx - tx * 56);
                }
                const double Ntot = (double)cz * (double)cy * (double)cx;
                auto vol_of = [&](unsigned int m) { double v = 0.0; for (int r = 0; r < 32; ++r) if ((m >> r) & 1u) v += vol_res[r]; return v; };
                double acc[kMaxResident];
                float amx[kMaxResident];
                int ahn[kMaxResident];
                unsigned int done[kMaxResident], masks[kMaxResident];
                for (int j = 0; j < kMaxResident; ++j) { acc[j] = 0.0; amx[j] = -INFINITY; ahn[j] = 0; done[j] = 0; masks[j] = 0; }
                // float32 walk (option ssim_f32, default on) + float64 re-walk of the candidates that end within `margin` (mean SSIM) of
                // the best: a float32 window variance is off by <= a few 1e-7 (values rescaled to [0, 1], sums restarted per segment)
                // against C2 = 9e-4 in the denominator -- up to ~1e-3 of a voxel's value in flat regions, far less in the mean over a
                // crop; 1e-3 of the MEAN is the margin.  Candidates are dropped only when their bound stays below the best sum by it.
                const bool walk_f32 = c->ssim_f32;
                const double margin = walk_f32 ? 1e-3 : 0.0;
                bool rewalk = false;
                auto run_round = [&]() -> int {
                    FusedBatch fb = fused_batch;
                    int maxsel = 0;
                    for (int j = 0; j < nb; ++j) {
                        fb.c[j].sel = masks[j];
                        if (!masks[j] || !fb.c[j].src) { fb.c[j].src = nullptr; masks[j] = 0; continue; }
                        maxsel = std::max(maxsel, ((nitems + K - 1) / K) * __builtin_popcount(masks[j]));
                    }
                    if (maxsel == 0) return MVS_OK;
                    const int gx = std::min(kStatBlocks, maxsel);
                    // (117 VGPRs at 3-4 waves per SIMD; forced to 5 waves it spills: pairwise 38.9 -> 43.2 ms, measured)
                    if (walk_f32 && !rewalk)
                        MVS_DUP("ssim_fused", hipLaunchKernelGGL(ssim_fused_batch_f32_kernel<7>, dim3(gx, nb), dim3(256), 0, c->stream, im0, S, fb, setB[2], setB[3], zseg,
                                           batch_cov_norm, C1, C2, pmax, phasnan, psum, K));
                    else
                        MVS_DUP("ssim_fused", hipLaunchKernelGGL(ssim_fused_batch_kernel<7>, dim3(gx, nb), dim3(256), 0, c->stream, im0, S, fb, setB[2], setB[3], zseg,
                                           batch_cov_norm, C1, C2, pmax, phasnan, psum, K));
                    MVS_DUP("finish", hipLaunchKernelGGL(finish_region_kernel, dim3(nb), dim3(256), 0, c->stream, pmax, phasnan, psum, reg_out, gx));
                    MVS_HIP_TRY(c, hipStreamSynchronize(c->stream));
                    for (int j = 0; j < nb; ++j) {
                        if (!masks[j]) continue;
                        acc[j] += h_reg[j].ssim_sum;
                        amx[j] = fmaxf(amx[j], h_reg[j].mx);
                        ahn[j] |= h_reg[j].hasnan;
                        done[j] |= masks[j];
                        masks[j] = 0;
                    }
                    return MVS_OK;
                };
                for (int j = 0; j < nb; ++j) masks[j] = fused_batch.c[j].src ? 0x0001u : 0u;
                rc = run_round();
                if (rc) return rc;
                // A candidate whose region maximum does not exceed im1_min is the reference's `continue` case (registration.py:530-533):
                // it takes no part in the arg max, so its sum must never be the one the others are dropped against (against a sparse
                // fixed image an all-background candidate can hold the highest sum).  Complete candidates know their maximum; the
                // leader is re-elected among the open candidates when it turns out to be such a candidate.
                auto excluded = [&](int j) { return done[j] == kAll && !((double)amx[j] > im1_min); };
                auto mean_of = [&](int j) { return acc[j] / std::max(vol_of(done[j]), 1.0); };
                auto elect = [&]() {
                    int l = -1;
                    for (int j = 0; j < nb; ++j)
                        if (fused_batch.c[j].src && !excluded(j) && !(done[j] == kAll && !std::isfinite(acc[j])) && (l < 0 || mean_of(j) > mean_of(l))) l = j;
                    return l;
                };
                int leader = elect();
                bool pruned[kMaxResident];
                double ub[kMaxResident];
                for (int j = 0; j < kMaxResident; ++j) { pruned[j] = false; ub[j] = 0.0; }
                bool have_best = false;
                double s_best = 0.0;
                static const bool dbg = getenv("MVS_PRUNE_DEBUG") != nullptr;
                for (int round = 0; round < 34; ++round) {
                    // Plan: the leader is completed; every other open candidate advances to the fraction at which its bound would
                    // fall below the reference sum if its mean stayed what it is so far (residues are taken in rising order; the
                    // reference is the best complete sum, before there is one the leader's extrapolated sum -- a guess that only
                    // sizes the round: candidates are dropped against complete sums alone).
                    if (!have_best && (leader < 0 || done[leader] == kAll)) leader = elect();      // the leader was a `continue` candidate / NaN
                    const double s_ref = have_best ? s_best : leader >= 0 ? mean_of(leader) * Ntot : Ntot * (1.0 + slack);
                    bool more = false;
                    for (int j = 0; j < nb; ++j) {
                        if (!fused_batch.c[j].src || done[j] == kAll || pruned[j]) continue;
                        const int k_done = __builtin_popcount(done[j]);
                        int k_to = K;
                        if (j != leader && (double)amx[j] > im1_min) {
                            const double mean_c = acc[j] / std::max(vol_of(done[j]), 1.0);
                            const double den = (1.0 + slack) - mean_c;
                            const double f = den > 0.0 ? ((1.0 + slack) - s_ref / Ntot) / den : 2.0;
                            if (f < 1.0) k_to = std::min(K, std::max(k_done + 1, (int)std::ceil((double)K * f * 1.15 + 0.25)));
                            if (4 * k_to >= 3 * K) k_to = K;
                        }
                        masks[j] = (unsigned int)((1ull << k_to) - 1ull) & ~(unsigned int)((1ull << k_done) - 1ull);
                        more = true;
                    }
                    if (!more) break;
                    rc = run_round();
                    if (rc) return rc;
                    for (int j = 0; j < nb; ++j)
                        if (fused_batch.c[j].src && done[j] == kAll && !excluded(j) && std::isfinite(acc[j]) && (!have_best || acc[j] > s_best)) {
                            s_best = acc[j];
                            have_best = true;
                        }
                    for (int j = 0; j < nb; ++j) {
                        if (!fused_batch.c[j].src || done[j] == kAll || pruned[j]) continue;
                        // (a candidate whose samples so far do not exceed im1_min may still be the reference's `continue` case: in full)
                        ub[j] = acc[j] + (Ntot - vol_of(done[j])) * (1.0 + slack);
                        if ((double)amx[j] > im1_min && ub[j] < s_best - (1e-9 + margin) * Ntot) pruned[j] = true;
                    }
                }
                if (walk_f32 && have_best) {
                    // the complete candidates within the margin of the best float32 sum: with two or more of them the arg max is decided
                    // by their float64 sums (whole volume, fresh accumulators)
                    int near = 0;
                    for (int j = 0; j < nb; ++j)
                        if (fused_batch.c[j].src && done[j] == kAll && !pruned[j] && !excluded(j) && std::isfinite(acc[j]) && acc[j] >= s_best - margin * Ntot) ++near;
                    if (near >= 2) {
                        rewalk = true;
                        for (int j = 0; j < nb; ++j) {
                            const bool sel = fused_batch.c[j].src && done[j] == kAll && !pruned[j] && !excluded(j) && std::isfinite(acc[j]) && acc[j] >= s_best - margin * Ntot;
                            masks[j] = sel ? kAll : 0u;
                            if (sel) { acc[j] = 0.0; c->reg_rewalks += 1; c->reg_cand_volumes += 1.0; }
                        }
                        rc = run_round();
                        if (rc) return rc;
                    }
                }
                if (dbg) {
                    fprintf(stderr, "prune: best %.4f |", s_best / Ntot);
                    for (int j = 0; j < nb; ++j)
                        if (fused_batch.c[j].src) fprintf(stderr, " %d/%d:%.3f%s", __builtin_popcount(done[j]), K, acc[j] / std::max(vol_of(done[j]), 1.0), pruned[j] ? "x" : "");
                    fprintf(stderr, "\n");
                }
                for (int j = 0; j < nb; ++j) {
                    h_rs[j].mx = amx[j];
                    h_rs[j].hasnan = ahn[j];
                    h_rs[j].ssim_sum = pruned[j] ? ub[j] : acc[j];        // pruned: the bound it could not exceed (< the best sum)
                    if (fused_batch.c[j].src) {
                        c->reg_cand_volumes += vol_of(done[j]) / Ntot;
                        c->reg_pruned += pruned[j] ? 1 : 0;
                    }
                }
                have_rs = true;
            }
        } else if (any_batched) {
            hipLaunchKernelGGL(ssim_first_pass_batch_kernel<7>, dim3(kStatBlocks, nb), dim3(256), 0, c->stream, im0, S, first_batch, pmax, phasnan);
            hipLaunchKernelGGL(ssim_yx_batch_kernel<7>, dim3(kStatBlocks, nb), dim3(256), 0, c->stream, yx_batch, setB[2], setB[3], S, batch_cov_norm, C1, C2, psum);
        }
        bool any = false;
        for (int j = 0; j < nb; ++j) any = any || scored[j];
        if (any && !have_rs) {
            hipLaunchKernelGGL(finish_region_kernel, dim3(nb), dim3(256), 0, c->stream, pmax, phasnan, psum, reg_out, kStatBlocks);
            MVS_HIP_TRY(c, hipStreamSynchronize(c->stream));
            for (int j = 0; j < nb; ++j) h_rs[j] = h_reg[j];
        }
        for (int j = 0; j < nb; ++j) {
            const int ic = todo[b0 + j];
            if (code_out[ic] != 0) continue;
            float region_nanmax = NAN;
            int region_hasnan = 0;
            if (scored[j]) {
                region_hasnan = h_rs[j].hasnan;
                region_nanmax = (h_rs[j].mx == -INFINITY) ? NAN : h_rs[j].mx;   // all-NaN region
            }
            // `if np.nanmax(im1t[mask_slices]) <= im1_min: continue` (Q3: nothing is appended)
            if (region_nanmax <= (float)im1_min) {
                code_out[ic] = 2;
                continue;
            }
            const Shape3 R = Rs[j];
            int min_shape = 0x7fffffff;
            for (int k = k0; k < 3; ++k) min_shape = std::min(min_shape, (&R.nz)[k]);
            const int win = std::min(7, min_shape - ((min_shape - 1) % 2));
            const float region_max = region_hasnan ? NAN : region_nanmax;   // np.max propagates NaN
            if (win < 3 || region_max <= (float)im1_min || !scored[j]) {
                ssim_out[ic] = -1.0;
            } else {
                const int pad = (win - 1) / 2;
                double cropn = 1.0;
                for (int k = k0; k < 3; ++k) cropn *= (double)((&R.nz)[k] - 2 * pad);
                ssim_out[ic] = h_rs[j].ssim_sum / cropn;
            }
        }
        if (quality_for_all)
            for (int j = 0; j < nb; ++j) {
                const int ic = todo[b0 + j];
                if (code_out[ic] != 0) continue;
                rc = spearman_from(ic, im1t_buf[j]);
                if (rc) return rc;
            }
    }
    if (!quality_for_all) {
        // The reference reports the Spearman coefficient of the SSIM-argmax candidate only (registration.py:
        // 543-556), so the rank correlation is evaluated for the candidates that hold the best SSIM (all of
        // them when several tie) and is NaN for the others.
        double best = -INFINITY;
        for (int ic = 0; ic < n_candidates; ++ic)
            if (code_out[ic] != 2 && ssim_out[ic] > best) best = ssim_out[ic];
        for (int ic = 0; ic < n_candidates; ++ic) {
            if (code_out[ic] != 0) continue;
            if (!(ssim_out[ic] == best)) { spearman_out[ic] = NAN; continue; }
            rc = spearman_from(ic, resident[ic] >= 0 ? im1t_buf[resident[ic]] : (const float*)nullptr);
            if (rc) return rc;
        }
    }
    MVS_HIP_TRY(c, hipEventRecord(c->ev_stop, c->stream));
    c->timing_valid = true;
    MVS_HIP_TRY(c, hipStreamSynchronize(c->stream));
    return MVS_OK;
}
