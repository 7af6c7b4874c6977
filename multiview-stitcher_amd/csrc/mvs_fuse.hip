// mvs_fuse.hip -- fused affine-resample + blend-weight + accumulate kernels (gfx950).
//
// Replaces, per output chunk, the body of the reference's fusion.fuse_np
// (src/multiview_stitcher/fusion/_core.py:1608-1713):
//   V x scipy.ndimage.affine_transform of the view slabs (transformation.py:136-139, cval=NaN)
//   V x get_blending_weights (weights.py:391-511: resampled 5^n EDT support + cosine ramp)
//   mask by ~isnan, normalize_weights (weights.py:325-345)
//   weighted_average_fusion / max_fusion / simple_average_fusion (_core.py:42-131)
//   trim halo, nan_to_num, astype(input dtype) (_core.py:1687-1713)
// with ONE pass over HBM: every needed input voxel is read once, every output
// voxel written once, the weights are evaluated analytically in registers.
//
// Work decomposition: a 256-thread workgroup owns a brick of 4(z) x 4(y) x 64(x)
// output voxels (16(y) x 64(x) for 2D); a wavefront owns 4 rows of 64 voxels, a
// lane owns 4 consecutive x voxels -> one 8 B (u16) / 16 B (f32) coalesced store.
// Wave 0 culls the chunk's views against the brick's back-projected bounding
// box (ballot compaction keeps view order, so the f32 sum order is fixed),
// then all waves loop over the surviving views only.
//
// Numerics: input coordinates are evaluated in double with scipy's operation
// order ((z*m0 + y*m1) + x*m2) + offset and NO fma contraction, so the
// in-bounds test (c < 0 || c > n-1 -> cval) classifies every voxel exactly as
// NI_GeometricTransform does; interpolation and accumulation are float32.
#include "mvs_internal.h"

#include <cmath>
#include <cstring>

namespace {

constexpr int kBrickX = 64;      // voxels along x per brick (16 lanes x 4 voxels)
constexpr int kVPT = 4;          // voxels per thread along x
constexpr int kLdsTables = 8;    // blend tables staged in LDS per pass
constexpr float kPiHalf = 1.57079632679489661923f;

struct DevView {
    const void* data;
    long long stride_z, stride_y;   // elements
    int nz, ny, nx;                 // slab shape
    int wnz;                        // z extent of the support table: 5 (3D) or 1 (2D)
    double m[9];
    double off[3];
    double wm[9];
    double woff[3];
    float edt[125];
    // ---- translation fast path (valid when tr_ok): matrix == I, diagonal support map ----
    int tr_ok;
    int io[3];        // input index = chunk index + io
    float fw[3];      // fractional interpolation weights (0 => single tap on that axis)
    int lo[3], hi[3]; // chunk-index range where the view is in bounds, exact per scipy's test
    float ws[3];      // tent scales of the closed-form support table edt = min_d(ws_d * tent(i_d))
    float pad[2];
};

struct FuseParams {
    const DevView* views;
    int nviews;
    void* out;
    int oz, oy, ox;        // shape of the (trimmed) result
    int tz, ty, tx;        // trim: chunk index = result index + trim
    int nbz, nby, nbx;     // brick grid
    int bz, by;            // brick extent along z and y (4,4) or (1,16)
};

template <typename T> __device__ __forceinline__ float load_as_float(const T* p, long long i);
template <> __device__ __forceinline__ float load_as_float<unsigned char>(const unsigned char* p, long long i) { return (float)p[i]; }
template <> __device__ __forceinline__ float load_as_float<unsigned short>(const unsigned short* p, long long i) { return (float)p[i]; }
template <> __device__ __forceinline__ float load_as_float<float>(const float* p, long long i) { return p[i]; }

// second tap of a linear interpolation at the upper border: scipy maps the
// out-of-range index by mirroring (ni_interpolation.c, edge offsets), its
// weight is 0 there.
__device__ __forceinline__ int second_tap(int i0, int n) {
    int i1 = i0 + 1;
    if (i1 >= n) i1 = (n > 1) ? n - 2 : 0;
    return i1;
}

// Sample one view at in-bounds double coordinates. ORDER 1: trilinear with all
// 8 taps always loaded (0 * NaN = NaN propagates like scipy); ORDER 0: nearest
// = floor(c + 0.5).
template <typename TIn, int ORDER>
__device__ __forceinline__ float sample_view(const DevView& V, double cz, double cy, double cx) {
    const TIn* p = (const TIn*)V.data;
    if (ORDER == 0) {
        int iz = (int)floor(cz + 0.5), iy = (int)floor(cy + 0.5), ix = (int)floor(cx + 0.5);
        return load_as_float<TIn>(p, iz * V.stride_z + iy * V.stride_y + ix);
    } else {
        double fz_ = floor(cz), fy_ = floor(cy), fx_ = floor(cx);
        int iz = (int)fz_, iy = (int)fy_, ix = (int)fx_;
        float wz = (float)(cz - fz_), wy = (float)(cy - fy_), wx = (float)(cx - fx_);
        int iz1 = second_tap(iz, V.nz), iy1 = second_tap(iy, V.ny), ix1 = second_tap(ix, V.nx);
        long long b00 = iz * V.stride_z + iy * V.stride_y;
        long long b01 = iz * V.stride_z + iy1 * V.stride_y;
        long long b10 = iz1 * V.stride_z + iy * V.stride_y;
        long long b11 = iz1 * V.stride_z + iy1 * V.stride_y;
        float v000 = load_as_float<TIn>(p, b00 + ix), v001 = load_as_float<TIn>(p, b00 + ix1);
        float v010 = load_as_float<TIn>(p, b01 + ix), v011 = load_as_float<TIn>(p, b01 + ix1);
        float v100 = load_as_float<TIn>(p, b10 + ix), v101 = load_as_float<TIn>(p, b10 + ix1);
        float v110 = load_as_float<TIn>(p, b11 + ix), v111 = load_as_float<TIn>(p, b11 + ix1);
        float ux = 1.f - wx, uy = 1.f - wy, uz = 1.f - wz;
        float a00 = fmaf(v001, wx, v000 * ux);
        float a01 = fmaf(v011, wx, v010 * ux);
        float a10 = fmaf(v101, wx, v100 * ux);
        float a11 = fmaf(v111, wx, v110 * ux);
        float b0 = fmaf(a01, wy, a00 * uy);
        float b1 = fmaf(a11, wy, a10 * uy);
        return fmaf(b1, wz, b0 * uz);
    }
}

// cosine ramp of weights.py:502-507: x<1 -> (cos((1-x)pi)+1)/2 == sin^2(pi x/2), clip to [0,1].
__device__ __forceinline__ float blend_ramp(float x) {
    if (!(x < 1.f)) return 1.f;
    if (x <= 0.f) return 0.f;
    float a = x * kPiHalf;
    float a2 = a * a;
    // sin(a), a in [0, pi/2], odd Taylor polynomial through a^13 (|err| < 7e-10)
    float s = fmaf(a2, 1.6059043836821613e-10f, -2.5052108385441720e-08f);
    s = fmaf(s, a2, 2.7557319223985893e-06f);
    s = fmaf(s, a2, -1.9841269841269841e-04f);
    s = fmaf(s, a2, 8.3333333333333333e-03f);
    s = fmaf(s, a2, -1.6666666666666666e-01f);
    s = fmaf(s * a2, a, a);
    // The reference rounds cos((1-x)pi) to float32 before the "+1, /2", which quantises the
    // ramp to multiples of 2^-25 near 0 (and makes weights below 2^-26 exactly 0 -- e.g. the
    // corner voxels of a tile).  Reproduce that rounding: c = fl(2w - 1), w' = (c + 1) / 2.
    const float c = fmaf(2.f, s * s, -1.f);
    return (c + 1.f) * 0.5f;
}

// Blend weight of one view at support-grid coordinates (weights.py:475-509):
// linear interpolation of the 5^n table (cval 0 outside [0,4]) then the ramp.
template <bool LDS>
__device__ __forceinline__ float blend_weight(const float* __restrict__ tab, int wnz, double cz, double cy, double cx) {
    const double zmax = (double)(wnz - 1);
    if (cz < 0.0 || cz > zmax || cy < 0.0 || cy > 4.0 || cx < 0.0 || cx > 4.0) return 0.f;
    double fz_ = floor(cz), fy_ = floor(cy), fx_ = floor(cx);
    int iz = (int)fz_, iy = (int)fy_, ix = (int)fx_;
    float wz = (float)(cz - fz_), wy = (float)(cy - fy_), wx = (float)(cx - fx_);
    int iz1 = second_tap(iz, wnz), iy1 = second_tap(iy, 5), ix1 = second_tap(ix, 5);
    int r00 = (iz * 5 + iy) * 5, r01 = (iz * 5 + iy1) * 5, r10 = (iz1 * 5 + iy) * 5, r11 = (iz1 * 5 + iy1) * 5;
    float ux = 1.f - wx, uy = 1.f - wy, uz = 1.f - wz;
    float a00 = fmaf(tab[r00 + ix1], wx, tab[r00 + ix] * ux);
    float a01 = fmaf(tab[r01 + ix1], wx, tab[r01 + ix] * ux);
    float a10 = fmaf(tab[r10 + ix1], wx, tab[r10 + ix] * ux);
    float a11 = fmaf(tab[r11 + ix1], wx, tab[r11 + ix] * ux);
    float b0 = fmaf(a01, wy, a00 * uy);
    float b1 = fmaf(a11, wy, a10 * uy);
    return blend_ramp(fmaf(b1, wz, b0 * uz));
}

// Weighted-average accumulator with an exact single-contributor state.
// The reference normalises first (w/sum(w), weights.py:325-345) and then sums v*w (_core.py:92-94):
// a voxel seen by ONE view gets weight w/w == 1 and comes out as v exactly (or 0 if w == 0).
// (acc, den) encodes: den == +0 -> empty; signbit(den) -> single view so far (acc = v, |den| = w);
// den > 0 -> two or more views (acc = sum w*v, den = sum w).
__device__ __forceinline__ void wa_update(float& acc, float& den, float w, float v) {
    const float dabs = fabsf(den);
    if (signbit(den)) {
        if (dabs == 0.f) { acc = v; den = -w; }
        else { acc = fmaf(w, v, acc * dabs); den = dabs + w; }
    } else if (den == 0.f) {
        acc = v;
        den = -w;
    } else {
        acc = fmaf(w, v, acc);
        den += w;
    }
}
__device__ __forceinline__ float wa_result(float acc, float den) {
    if (signbit(den)) return (den < 0.f) ? acc : 0.f;
    return (den > 0.f) ? acc / den : 0.f;
}

template <typename TOut> __device__ __forceinline__ TOut cast_out(float v);
template <> __device__ __forceinline__ float cast_out<float>(float v) { return v; }
// np.nan_to_num(...).astype(uint16/uint8): C truncation toward zero (_core.py:1713)
template <> __device__ __forceinline__ unsigned short cast_out<unsigned short>(float v) { return (unsigned short)(int)v; }
template <> __device__ __forceinline__ unsigned char cast_out<unsigned char>(float v) { return (unsigned char)(int)v; }

template <typename TOut> struct Vec4;
template <> struct Vec4<float> { typedef float4 type; };
template <> struct Vec4<unsigned short> { typedef ushort4 type; };
template <> struct Vec4<unsigned char> { typedef uchar4 type; };

// Conservative test: can any voxel of the brick (chunk index box lo..hi, inclusive)
// land inside view V?  Exact classification happens per voxel afterwards.
__device__ __forceinline__ bool brick_hits_view(const DevView& V, const int lo[3], const int hi[3]) {
    const int n[3] = {V.nz, V.ny, V.nx};
#pragma unroll
    for (int d = 0; d < 3; ++d) {
        double cmin = V.off[d], cmax = V.off[d];
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            double a = V.m[d * 3 + k] * (double)lo[k], b = V.m[d * 3 + k] * (double)hi[k];
            cmin += fmin(a, b);
            cmax += fmax(a, b);
        }
        if (cmax < -1e-3 || cmin > (double)(n[d] - 1) + 1e-3) return false;
    }
    return true;
}

template <typename TOut>
__device__ __forceinline__ void store_row4(TOut* out, long long row, int x0, int ox, const float r[4]);

template <typename TIn, typename TOut, int ORDER, int FUSION>
__global__ __launch_bounds__(256) void fuse_kernel(FuseParams P) {
    __shared__ int s_act[64];
    __shared__ int s_nact;
    __shared__ float s_edt[kLdsTables][128];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;

    // brick coordinates (x fastest)
    int b = blockIdx.x;
    const int bx = b % P.nbx;
    b /= P.nbx;
    const int byi = b % P.nby;
    const int bzi = b / P.nby;

    const int lx = lane & 15, ly = lane >> 4;
    int z, y;
    if (P.bz == 4) {
        z = bzi * 4 + wave;
        y = byi * 4 + ly;
    } else {
        z = bzi;
        y = byi * 16 + wave * 4 + ly;
    }
    const int x0 = bx * kBrickX + lx * kVPT;
    const bool row_ok = (z < P.oz) && (y < P.oy) && (x0 < P.ox);

    // chunk-index box of this brick (for culling)
    int lo[3] = {bzi * P.bz + P.tz, byi * P.by + P.ty, bx * kBrickX + P.tx};
    int hi[3] = {min(bzi * P.bz + P.bz, P.oz) - 1 + P.tz, min(byi * P.by + P.by, P.oy) - 1 + P.ty,
                 min(bx * kBrickX + kBrickX, P.ox) - 1 + P.tx};

    const double pz = (double)(z + P.tz), py = (double)(y + P.ty);
    const double px0 = (double)(x0 + P.tx);

    float acc[kVPT], den[kVPT];
#pragma unroll
    for (int j = 0; j < kVPT; ++j) {
        acc[j] = (FUSION == MVS_FUSE_MAX) ? -INFINITY : 0.f;
        den[j] = 0.f;
    }

    for (int base = 0; base < P.nviews; base += 64) {
        __syncthreads();
        if (wave == 0) {
            int v = base + lane;
            bool act = false;
            if (v < P.nviews) act = brick_hits_view(P.views[v], lo, hi);
            unsigned long long mask = __ballot(act);
            if (act) s_act[__popcll(mask & ((1ull << lane) - 1ull))] = v;
            if (lane == 0) s_nact = __popcll(mask);
        }
        __syncthreads();
        const int nact = s_nact;
        if (FUSION == MVS_FUSE_WEIGHTED_AVERAGE) {
            const int ncopy = min(nact, kLdsTables) * 125;
            for (int i = tid; i < ncopy; i += 256) {
                int a = i / 125, k = i - a * 125;
                s_edt[a][k] = P.views[s_act[a]].edt[k];
            }
            __syncthreads();
        }
        if (!row_ok) continue;

        for (int a = 0; a < nact; ++a) {
            const int vi = __builtin_amdgcn_readfirstlane(s_act[a]);
            const DevView& V = P.views[vi];
            // row part of scipy's coordinate sum: (0 + z*m0) + y*m1
            const double rz = pz * V.m[0] + py * V.m[1];
            const double ry = pz * V.m[3] + py * V.m[4];
            const double rx = pz * V.m[6] + py * V.m[7];
            const double zmax = (double)(V.nz - 1), ymax = (double)(V.ny - 1), xmax = (double)(V.nx - 1);
            // support-grid coordinates: affine in the index, evaluated incrementally
            double wz0 = 0, wy0 = 0, wx0 = 0;
            if (FUSION == MVS_FUSE_WEIGHTED_AVERAGE) {
                wz0 = ((pz * V.wm[0] + py * V.wm[1]) + px0 * V.wm[2]) + V.woff[0];
                wy0 = ((pz * V.wm[3] + py * V.wm[4]) + px0 * V.wm[5]) + V.woff[1];
                wx0 = ((pz * V.wm[6] + py * V.wm[7]) + px0 * V.wm[8]) + V.woff[2];
            }
#pragma unroll
            for (int j = 0; j < kVPT; ++j) {
                const double px = px0 + (double)j;
                const double cz = (rz + px * V.m[2]) + V.off[0];
                const double cy = (ry + px * V.m[5]) + V.off[1];
                const double cx = (rx + px * V.m[8]) + V.off[2];
                const bool inb = !(cz < 0.0 || cz > zmax || cy < 0.0 || cy > ymax || cx < 0.0 || cx > xmax);
                if (!inb) continue;
                const float val = sample_view<TIn, ORDER>(V, cz, cy, cx);
                if (val != val) continue;   // NaN voxels are invalid (weights masked by ~isnan)
                if (FUSION == MVS_FUSE_WEIGHTED_AVERAGE) {
                    const double dj = (double)j;
                    const double cwz = wz0 + dj * V.wm[2], cwy = wy0 + dj * V.wm[5], cwx = wx0 + dj * V.wm[8];
                    float w;
                    if (a < kLdsTables) w = blend_weight<true>(s_edt[a], V.wnz, cwz, cwy, cwx);
                    else w = blend_weight<false>(V.edt, V.wnz, cwz, cwy, cwx);
                    wa_update(acc[j], den[j], w, val);
                } else if (FUSION == MVS_FUSE_MAX) {
                    acc[j] = fmaxf(acc[j], val);
                    den[j] = 1.f;
                } else {
                    acc[j] += val;
                    den[j] += 1.f;
                }
            }
        }
    }
    if (!row_ok) return;

    float r[kVPT];
#pragma unroll
    for (int j = 0; j < kVPT; ++j) {
        float o;
        if (FUSION == MVS_FUSE_MAX) o = (den[j] > 0.f) ? acc[j] : 0.f;
        else if (FUSION == MVS_FUSE_WEIGHTED_AVERAGE) o = wa_result(acc[j], den[j]);
        else o = (den[j] > 0.f) ? acc[j] / den[j] : 0.f;
        if (o != o) o = 0.f;   // nan_to_num
        r[j] = o;
    }
    store_row4<TOut>((TOut*)P.out, ((long long)z * P.oy + y) * (long long)P.ox, x0, P.ox, r);
}

template <typename TOut>
__device__ __forceinline__ void store_row4(TOut* out, long long row, int x0, int ox, const float r[4]) {
    if (x0 + kVPT <= ox && ((row + x0) & 3) == 0) {
        typename Vec4<TOut>::type v4;
        v4.x = cast_out<TOut>(r[0]);
        v4.y = cast_out<TOut>(r[1]);
        v4.z = cast_out<TOut>(r[2]);
        v4.w = cast_out<TOut>(r[3]);
        *reinterpret_cast<typename Vec4<TOut>::type*>(out + row + x0) = v4;
    } else {
#pragma unroll
        for (int j = 0; j < kVPT; ++j)
            if (x0 + j < ox) out[row + x0 + j] = cast_out<TOut>(r[j]);
    }
}


// =============================================================================================
// Translation fast path.  When every view's pixel matrix is the identity (tile grids: stage
// translations, integer or fractional) the resample degenerates to a fixed 2x2x2 stencil with
// per-view constant weights, the in-bounds test to an integer box test (the box is derived on the
// host from scipy's double-precision test, see prepare_translation_view), and the blend weight to
// a 1-D piecewise-linear function of x per output row:
//     table T[i][j][k] = min(sz*t_i, sy*t_j, sx*t_k), t = {0,1,2,1,0}
//     W(z,y,x) = sum_r wx_r * G(ax_r),   G(a) = sum_pq wz_p wy_q min(min(az_p, ay_q), a)
// ax_r only takes the values {0, sx, 2sx}, so per (row, view) G1 = G(sx), G2 = G(2sx) are wave
// uniform and W(x) = lerp over the nodes {0, G1, G2, G1, 0} at the support coordinate of x.
// A wavefront owns one output row segment of 256 voxels (z, y uniform -> scalar registers),
// a lane 4 consecutive voxels: inputs arrive as (unaligned) 8/16-byte vector loads per stencil
// row, the result leaves as one aligned 8/16-byte store.
// =============================================================================================
constexpr int kTrBrickX = 256;

template <typename T> struct RowVec;
template <> struct RowVec<unsigned short> { typedef unsigned short v4 __attribute__((ext_vector_type(4), aligned(2))); };
template <> struct RowVec<unsigned char> { typedef unsigned char v4 __attribute__((ext_vector_type(4), aligned(1))); };
template <> struct RowVec<float> { typedef float v4 __attribute__((ext_vector_type(4), aligned(4))); };

// x-interpolated values of 4 consecutive output voxels from one input row.
// p points at the first tap of voxel 0.  full: all 4 voxels (and their second taps) are in bounds.
template <typename TIn>
__device__ __forceinline__ void row_taps(const TIn* p, bool fracx, float wx, bool full, int jlo, int jhi, float r[4]) {
    float e[5];
    if (full) {
        typename RowVec<TIn>::v4 v = *reinterpret_cast<const typename RowVec<TIn>::v4*>(p);
        e[0] = (float)v.x; e[1] = (float)v.y; e[2] = (float)v.z; e[3] = (float)v.w;
        e[4] = fracx ? (float)p[4] : 0.f;
    } else {
#pragma unroll
        for (int j = 0; j < 5; ++j) {
            const bool need = (j >= jlo && j <= jhi) || (fracx && j - 1 >= jlo && j - 1 <= jhi);
            e[j] = need ? (float)p[j] : 0.f;
        }
    }
    if (fracx) {
        const float ux = 1.f - wx;
#pragma unroll
        for (int j = 0; j < 4; ++j) r[j] = fmaf(e[j + 1], wx, e[j] * ux);
    } else {
#pragma unroll
        for (int j = 0; j < 4; ++j) r[j] = e[j];
    }
}

__device__ __forceinline__ float tent5(int i) { return (float)min(i, 4 - i); }

// Row-uniform part of the blend weight: nodes G1, G2 (see header comment). Returns false if the
// row lies outside the support along z or y (weight 0 everywhere, cval of the table resample).
__device__ __forceinline__ bool blend_row_nodes(const DevView& V, double pz, double py, float& G1, float& G2) {
    float az0 = INFINITY, az1 = INFINITY, fz = 0.f;
    if (V.wnz > 1) {
        const double cz = pz * V.wm[0] + V.woff[0];
        if (cz < 0.0 || cz > 4.0) return false;
        const double f = floor(cz);
        const int i = (int)f;
        fz = (float)(cz - f);
        az0 = V.ws[0] * tent5(i);
        az1 = V.ws[0] * tent5(min(i + 1, 4));
    }
    const double cy = py * V.wm[4] + V.woff[1];
    if (cy < 0.0 || cy > 4.0) return false;
    const double fy_ = floor(cy);
    const int iy = (int)fy_;
    const float fy = (float)(cy - fy_);
    const float ay0 = V.ws[1] * tent5(iy), ay1 = V.ws[1] * tent5(min(iy + 1, 4));
    const float uz = 1.f - fz, uy = 1.f - fy;
    const float m00 = fminf(az0, ay0), m01 = fminf(az0, ay1), m10 = fminf(az1, ay0), m11 = fminf(az1, ay1);
    const float a1 = V.ws[2], a2 = 2.f * V.ws[2];
    // same association as the table interpolation: lerp along y, then along z
    float g0 = fmaf(fminf(m01, a1), fy, fminf(m00, a1) * uy);
    float g1 = fmaf(fminf(m11, a1), fy, fminf(m10, a1) * uy);
    G1 = fmaf(g1, fz, g0 * uz);
    g0 = fmaf(fminf(m01, a2), fy, fminf(m00, a2) * uy);
    g1 = fmaf(fminf(m11, a2), fy, fminf(m10, a2) * uy);
    G2 = fmaf(g1, fz, g0 * uz);
    return true;
}

template <typename TIn, typename TOut, int FUSION>
__global__ __launch_bounds__(256) void fuse_tr_kernel(FuseParams P) {
    __shared__ int s_act[64];
    __shared__ int s_nact;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    int b = blockIdx.x;
    const int bx = b % P.nbx;
    b /= P.nbx;
    const int byi = b % P.nby;
    const int bzi = b / P.nby;
    int z, y;
    if (P.bz == 2) {
        z = bzi * 2 + (wave >> 1);
        y = byi * 2 + (wave & 1);
    } else {
        z = bzi;
        y = byi * 4 + wave;
    }
    const int x0 = bx * kTrBrickX + lane * kVPT;
    const bool row_ok = (z < P.oz) && (y < P.oy);
    const int zc = z + P.tz, yc = y + P.ty, xc0 = x0 + P.tx;   // chunk indices
    // brick box in chunk indices (for culling)
    const int bz0 = bzi * P.bz + P.tz, bz1 = min(bzi * P.bz + P.bz, P.oz) - 1 + P.tz;
    const int by0 = byi * P.by + P.ty, by1 = min(byi * P.by + P.by, P.oy) - 1 + P.ty;
    const int bx0 = bx * kTrBrickX + P.tx, bx1 = min(bx * kTrBrickX + kTrBrickX, P.ox) - 1 + P.tx;

    float acc[kVPT], den[kVPT];
#pragma unroll
    for (int j = 0; j < kVPT; ++j) {
        acc[j] = (FUSION == MVS_FUSE_MAX) ? -INFINITY : 0.f;
        den[j] = 0.f;
    }

    for (int base = 0; base < P.nviews; base += 64) {
        __syncthreads();
        if (wave == 0) {
            const int v = base + lane;
            bool act = false;
            if (v < P.nviews) {
                const DevView& V = P.views[v];
                act = V.lo[0] <= bz1 && V.hi[0] >= bz0 && V.lo[1] <= by1 && V.hi[1] >= by0 && V.lo[2] <= bx1 &&
                      V.hi[2] >= bx0;
            }
            const unsigned long long mask = __ballot(act);
            if (act) s_act[__popcll(mask & ((1ull << lane) - 1ull))] = v;
            if (lane == 0) s_nact = __popcll(mask);
        }
        __syncthreads();
        const int nact = s_nact;
        if (!row_ok) continue;

        for (int a = 0; a < nact; ++a) {
            const int vi = __builtin_amdgcn_readfirstlane(s_act[a]);
            const DevView& V = P.views[vi];
            if (zc < V.lo[0] || zc > V.hi[0] || yc < V.lo[1] || yc > V.hi[1]) continue;   // wave uniform
            const int xlo = V.lo[2], xhi = V.hi[2];
            const int jlo = max(xlo - xc0, 0), jhi = min(xhi - xc0, kVPT - 1);
            const bool any = jlo <= jhi;
            const bool full = (jlo == 0) && (jhi == kVPT - 1);
            float G1 = 0.f, G2 = 0.f;
            bool wrow = true;
            if (FUSION == MVS_FUSE_WEIGHTED_AVERAGE) wrow = blend_row_nodes(V, (double)zc, (double)yc, G1, G2);
            if (!any) continue;

            const float wz = V.fw[0], wy = V.fw[1], wx = V.fw[2];
            const bool fracz = wz > 0.f, fracy = wy > 0.f, fracx = wx > 0.f;
            const TIn* p00 = (const TIn*)V.data + (long long)(zc + V.io[0]) * V.stride_z +
                             (long long)(yc + V.io[1]) * V.stride_y + (xc0 + V.io[2]);
            float val[kVPT];
            row_taps<TIn>(p00, fracx, wx, full, jlo, jhi, val);
            if (fracy) {
                float t[kVPT];
                row_taps<TIn>(p00 + V.stride_y, fracx, wx, full, jlo, jhi, t);
                const float uy = 1.f - wy;
#pragma unroll
                for (int j = 0; j < kVPT; ++j) val[j] = fmaf(t[j], wy, val[j] * uy);
            }
            if (fracz) {
                float v1[kVPT];
                row_taps<TIn>(p00 + V.stride_z, fracx, wx, full, jlo, jhi, v1);
                if (fracy) {
                    float t[kVPT];
                    row_taps<TIn>(p00 + V.stride_z + V.stride_y, fracx, wx, full, jlo, jhi, t);
                    const float uy = 1.f - wy;
#pragma unroll
                    for (int j = 0; j < kVPT; ++j) v1[j] = fmaf(t[j], wy, v1[j] * uy);
                }
                const float uz = 1.f - wz;
#pragma unroll
                for (int j = 0; j < kVPT; ++j) val[j] = fmaf(v1[j], wz, val[j] * uz);
            }

            double cw0 = 0.0;
            const double dwx = V.wm[8];
            float dG = 0.f;
            if (FUSION == MVS_FUSE_WEIGHTED_AVERAGE) {
                cw0 = (double)xc0 * dwx + V.woff[2];
                dG = G2 - G1;
            }
#pragma unroll
            for (int j = 0; j < kVPT; ++j) {
                if (j < jlo || j > jhi) continue;
                const float v = val[j];
                if (v != v) continue;
                if (FUSION == MVS_FUSE_WEIGHTED_AVERAGE) {
                    float w = 0.f;
                    const double cw = cw0 + (double)j * dwx;
                    if (wrow && cw >= 0.0 && cw <= 4.0) {
                        const float u = (float)fmin(cw, 4.0 - cw);
                        const float W = (u <= 1.f) ? u * G1 : fmaf(u - 1.f, dG, G1);
                        w = blend_ramp(W);
                    }
                    wa_update(acc[j], den[j], w, v);
                } else if (FUSION == MVS_FUSE_MAX) {
                    acc[j] = fmaxf(acc[j], v);
                    den[j] = 1.f;
                } else {
                    acc[j] += v;
                    den[j] += 1.f;
                }
            }
        }
    }
    if (!row_ok || x0 >= P.ox) return;
    float r[kVPT];
#pragma unroll
    for (int j = 0; j < kVPT; ++j) {
        float o;
        if (FUSION == MVS_FUSE_MAX) o = (den[j] > 0.f) ? acc[j] : 0.f;
        else if (FUSION == MVS_FUSE_WEIGHTED_AVERAGE) o = wa_result(acc[j], den[j]);
        else o = (den[j] > 0.f) ? acc[j] / den[j] : 0.f;
        if (o != o) o = 0.f;
        r[j] = o;
    }
    store_row4<TOut>((TOut*)P.out, ((long long)z * P.oy + y) * (long long)P.ox, x0, P.ox, r);
}

// ---- host side of the fast path ---------------------------------------------------------------
// c(i) = fl(i + off) is what the generic kernel (and scipy) test against [0, n-1] for an identity
// matrix; it is monotone in i, so the in-bounds set is an integer interval found exactly here.
static void exact_valid_range(double off, int n, int n_out, int* lo_out, int* hi_out) {
    auto c = [off](long long i) { return (double)i + off; };
    long long lo = (long long)ceil(-off);
    while (lo > 0 && c(lo - 1) >= 0.0) --lo;
    while (c(lo) < 0.0) ++lo;
    long long hi = (long long)floor((double)(n - 1) - off);
    while (c(hi + 1) <= (double)(n - 1)) ++hi;
    while (c(hi) > (double)(n - 1)) --hi;
    if (lo < 0) lo = 0;
    if (hi > n_out - 1) hi = n_out - 1;
    *lo_out = (int)lo;
    *hi_out = (int)hi;   // lo > hi: the view never contributes
}

// Decide whether a view qualifies for the translation fast path and derive its constants.
static void prepare_translation_view(DevView* d, int order, int fusion, const int64_t chunk_shape[3]) {
    d->tr_ok = 0;
    static const double I9[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
    for (int k = 0; k < 9; ++k)
        if (d->m[k] != I9[k]) return;
    const int n[3] = {d->nz, d->ny, d->nx};
    for (int k = 0; k < 3; ++k) {
        if (!(fabs(d->off[k]) < 65536.0) || n[k] > 65536 || chunk_shape[k] > 65536) return;
    }
    if (fusion == MVS_FUSE_WEIGHTED_AVERAGE) {
        const int offd[6] = {1, 2, 3, 5, 6, 7};
        for (int k : offd)
            if (d->wm[k] != 0.0) return;
        // closed-form check of the support table: edt == min_d(ws_d * tent(i_d))
        const int nz = d->wnz;
        float sz = INFINITY, sy, sx;
        if (nz == 5) {
            sz = d->edt[(1 * 5 + 2) * 5 + 2];
            sy = d->edt[(2 * 5 + 1) * 5 + 2];
            sx = d->edt[(2 * 5 + 2) * 5 + 1];
        } else {
            sy = d->edt[1 * 5 + 2];
            sx = d->edt[2 * 5 + 1];
        }
        for (int i = 0; i < nz; ++i)
            for (int j = 0; j < 5; ++j)
                for (int k = 0; k < 5; ++k) {
                    float t = fminf(sy * (float)std::min(j, 4 - j), sx * (float)std::min(k, 4 - k));
                    if (nz == 5) t = fminf(t, sz * (float)std::min(i, 4 - i));
                    if (t != d->edt[(i * 5 + j) * 5 + k]) return;
                }
        d->ws[0] = (nz == 5) ? sz : 0.f;
        d->ws[1] = sy;
        d->ws[2] = sx;
    }
    for (int k = 0; k < 3; ++k) {
        const double off = d->off[k];
        if (order == 0) {
            d->io[k] = (int)floor(off + 0.5);
            d->fw[k] = 0.f;
        } else {
            const double f = floor(off);
            d->io[k] = (int)f;
            d->fw[k] = (float)(off - f);
        }
        exact_valid_range(off, n[k], (int)chunk_shape[k], &d->lo[k], &d->hi[k]);
    }
    d->tr_ok = 1;
}

// Single-view resample to float32 with an arbitrary cval (transformation.py:136-139).
template <typename TIn, int ORDER>
__global__ __launch_bounds__(256) void resample_kernel(DevView V, float* out, int oz, int oy, int ox, float cval) {
    const long long n = (long long)oz * oy * ox;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        int x = (int)(i % ox);
        long long t = i / ox;
        int y = (int)(t % oy);
        int z = (int)(t / oy);
        const double pz = (double)z, py = (double)y, px = (double)x;
        const double cz = ((pz * V.m[0] + py * V.m[1]) + px * V.m[2]) + V.off[0];
        const double cy = ((pz * V.m[3] + py * V.m[4]) + px * V.m[5]) + V.off[1];
        const double cx = ((pz * V.m[6] + py * V.m[7]) + px * V.m[8]) + V.off[2];
        const bool inb = !(cz < 0.0 || cz > (double)(V.nz - 1) || cy < 0.0 || cy > (double)(V.ny - 1) ||
                           cx < 0.0 || cx > (double)(V.nx - 1));
        out[i] = inb ? sample_view<TIn, ORDER>(V, cz, cy, cx) : cval;
    }
}

// Blend-weight volume of one view (weights.py:391-511), float32.
__global__ __launch_bounds__(256) void blend_kernel(DevView V, float* out, int oz, int oy, int ox) {
    const long long n = (long long)oz * oy * ox;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        int x = (int)(i % ox);
        long long t = i / ox;
        int y = (int)(t % oy);
        int z = (int)(t / oy);
        const double pz = (double)z, py = (double)y, px = (double)x;
        const double cz = ((pz * V.wm[0] + py * V.wm[1]) + px * V.wm[2]) + V.woff[0];
        const double cy = ((pz * V.wm[3] + py * V.wm[4]) + px * V.wm[5]) + V.woff[1];
        const double cx = ((pz * V.wm[6] + py * V.wm[7]) + px * V.wm[8]) + V.woff[2];
        out[i] = blend_weight<false>(V.edt, V.wnz, cz, cy, cx);
    }
}

int fill_dev_view(MvsContext* c, const mvs_view_t& v, int ndim, const void* dev_data, DevView* d) {
    if (v.stride[2] != 1) return mvs_fail(c, MVS_ERR_UNSUPPORTED, "view stride along x must be 1");
    for (int k = 0; k < 3; ++k)
        if (v.shape[k] < 1 || v.shape[k] > 0x7fffffffLL)
            return mvs_fail(c, MVS_ERR_INVALID_ARG, "view shape[%d]=%lld out of range", k, (long long)v.shape[k]);
    d->data = dev_data;
    d->stride_z = v.stride[0];
    d->stride_y = v.stride[1];
    d->nz = (int)v.shape[0];
    d->ny = (int)v.shape[1];
    d->nx = (int)v.shape[2];
    d->wnz = (ndim == 3) ? 5 : 1;
    memcpy(d->m, v.matrix, sizeof(d->m));
    memcpy(d->off, v.offset, sizeof(d->off));
    memcpy(d->wm, v.w_matrix, sizeof(d->wm));
    memcpy(d->woff, v.w_offset, sizeof(d->woff));
    memcpy(d->edt, v.edt, sizeof(d->edt));
    d->pad[0] = d->pad[1] = d->pad[2] = 0.f;
    return MVS_OK;
}

template <typename TIn, typename TOut>
void launch_fuse(const FuseParams& P, int order, int fusion, int nblocks, hipStream_t s) {
#define MVS_LAUNCH(O, F) hipLaunchKernelGGL((fuse_kernel<TIn, TOut, O, F>), dim3(nblocks), dim3(256), 0, s, P)
    if (order == 0) {
        if (fusion == MVS_FUSE_WEIGHTED_AVERAGE) MVS_LAUNCH(0, MVS_FUSE_WEIGHTED_AVERAGE);
        else if (fusion == MVS_FUSE_MAX) MVS_LAUNCH(0, MVS_FUSE_MAX);
        else MVS_LAUNCH(0, MVS_FUSE_SIMPLE_AVERAGE);
    } else {
        if (fusion == MVS_FUSE_WEIGHTED_AVERAGE) MVS_LAUNCH(1, MVS_FUSE_WEIGHTED_AVERAGE);
        else if (fusion == MVS_FUSE_MAX) MVS_LAUNCH(1, MVS_FUSE_MAX);
        else MVS_LAUNCH(1, MVS_FUSE_SIMPLE_AVERAGE);
    }
#undef MVS_LAUNCH
}

template <typename TIn, typename TOut>
void launch_fuse_tr(const FuseParams& P, int fusion, int nblocks, hipStream_t s) {
    if (fusion == MVS_FUSE_WEIGHTED_AVERAGE)
        hipLaunchKernelGGL((fuse_tr_kernel<TIn, TOut, MVS_FUSE_WEIGHTED_AVERAGE>), dim3(nblocks), dim3(256), 0, s, P);
    else if (fusion == MVS_FUSE_MAX)
        hipLaunchKernelGGL((fuse_tr_kernel<TIn, TOut, MVS_FUSE_MAX>), dim3(nblocks), dim3(256), 0, s, P);
    else
        hipLaunchKernelGGL((fuse_tr_kernel<TIn, TOut, MVS_FUSE_SIMPLE_AVERAGE>), dim3(nblocks), dim3(256), 0, s, P);
}

inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

}  // namespace

int mvs_fuse_content_based(MvsContext* c, const mvs_view_t* views, int32_t n_views,
                           const mvs_fuse_opts_t* opts, void* out);   // mvs_gauss.hip

extern "C" int mvs_fuse_chunk(int device, const mvs_view_t* views, int32_t n_views,
                              const mvs_fuse_opts_t* opts, void* out) {
    MvsContext* c;
    int rc = mvs_check_ready(device, &c);
    if (rc) return rc;
    std::lock_guard<std::mutex> lock(c->mu);
    if (!views || n_views < 1 || !opts || !out)
        return mvs_fail(c, MVS_ERR_INVALID_ARG, "mvs_fuse_chunk: NULL/empty argument");
    if (opts->ndim != 2 && opts->ndim != 3)
        return mvs_fail(c, MVS_ERR_INVALID_ARG, "mvs_fuse_chunk: ndim must be 2 or 3");
    if (opts->order != 0 && opts->order != 1)
        return mvs_fail(c, MVS_ERR_UNSUPPORTED, "mvs_fuse_chunk: interpolation order %d (only 0|1)", opts->order);
    if (opts->fusion < 0 || opts->fusion > 2)
        return mvs_fail(c, MVS_ERR_INVALID_ARG, "mvs_fuse_chunk: unknown fusion %d", opts->fusion);
    const int dtype = views[0].dtype;
    if (!mvs_dtype_size(dtype)) return mvs_fail(c, MVS_ERR_INVALID_ARG, "mvs_fuse_chunk: bad dtype %d", dtype);
    for (int i = 0; i < n_views; ++i)
        if (views[i].dtype != dtype)
            return mvs_fail(c, MVS_ERR_UNSUPPORTED, "mvs_fuse_chunk: views must share one dtype");
    if (opts->out_dtype != dtype)
        return mvs_fail(c, MVS_ERR_UNSUPPORTED, "mvs_fuse_chunk: out_dtype must equal the input dtype");
    int64_t os[3];
    for (int k = 0; k < 3; ++k) {
        os[k] = opts->out_shape[k] - 2 * opts->trim[k];
        if (opts->trim[k] < 0 || os[k] < 1 || opts->out_shape[k] > 0x7fffffffLL)
            return mvs_fail(c, MVS_ERR_INVALID_ARG, "mvs_fuse_chunk: bad out_shape/trim on axis %d", k);
    }
    if (opts->ndim == 2 && opts->out_shape[0] != 1)
        return mvs_fail(c, MVS_ERR_INVALID_ARG, "mvs_fuse_chunk: 2D chunks need out_shape[0] == 1");
    MVS_HIP_TRY(c, hipSetDevice(device));

    if (opts->weights == MVS_WEIGHTS_CONTENT_BASED) {
        if (opts->fusion != MVS_FUSE_WEIGHTED_AVERAGE)
            return mvs_fail(c, MVS_ERR_UNSUPPORTED, "content_based weights need weighted_average fusion");
        return mvs_fuse_content_based(c, views, n_views, opts, out);
    }
    if (opts->weights != MVS_WEIGHTS_NONE)
        return mvs_fail(c, MVS_ERR_INVALID_ARG, "mvs_fuse_chunk: unknown weights %d", opts->weights);

    // stage host slabs into device scratch (slot 0)
    const size_t es = mvs_dtype_size(dtype);
    size_t host_bytes = 0;
    for (int i = 0; i < n_views; ++i)
        if (views[i].mem == MVS_MEM_HOST) {
            if (views[i].stride[1] != views[i].shape[2] || views[i].stride[0] != views[i].shape[1] * views[i].shape[2])
                return mvs_fail(c, MVS_ERR_UNSUPPORTED, "host slabs must be C-contiguous");
            host_bytes += align_up((size_t)views[i].shape[0] * views[i].shape[1] * views[i].shape[2] * es, 256);
        }
    char* slab_base = nullptr;
    if (host_bytes) {
        slab_base = (char*)mvs_scratch(c, 0, host_bytes);
        if (!slab_base) return MVS_ERR_HIP;
    }
    const size_t params_bytes = sizeof(DevView) * (size_t)n_views;
    DevView* hviews = (DevView*)mvs_pinned(c, params_bytes);
    if (!hviews) return MVS_ERR_HIP;
    DevView* dviews = (DevView*)mvs_scratch(c, 2, params_bytes);
    if (!dviews) return MVS_ERR_HIP;

    size_t cursor = 0;
    for (int i = 0; i < n_views; ++i) {
        const void* dptr = views[i].data;
        if (views[i].mem == MVS_MEM_HOST) {
            size_t nb = (size_t)views[i].shape[0] * views[i].shape[1] * views[i].shape[2] * es;
            MVS_HIP_TRY(c, hipMemcpyAsync(slab_base + cursor, views[i].data, nb, hipMemcpyHostToDevice, c->stream));
            dptr = slab_base + cursor;
            cursor += align_up(nb, 256);
        }
        rc = fill_dev_view(c, views[i], opts->ndim, dptr, &hviews[i]);
        if (rc) return rc;
        prepare_translation_view(&hviews[i], opts->order, opts->fusion, opts->out_shape);
    }
    bool use_tr = !c->force_generic;
    for (int i = 0; i < n_views && use_tr; ++i) use_tr = hviews[i].tr_ok != 0;
    MVS_HIP_TRY(c, hipMemcpyAsync(dviews, hviews, params_bytes, hipMemcpyHostToDevice, c->stream));

    const size_t out_bytes = (size_t)os[0] * os[1] * os[2] * es;
    void* dout = out;
    if (opts->out_mem == MVS_MEM_HOST) {
        dout = mvs_scratch(c, 1, out_bytes);
        if (!dout) return MVS_ERR_HIP;
    }

    FuseParams P;
    P.views = dviews;
    P.nviews = n_views;
    P.out = dout;
    P.oz = (int)os[0]; P.oy = (int)os[1]; P.ox = (int)os[2];
    P.tz = (int)opts->trim[0]; P.ty = (int)opts->trim[1]; P.tx = (int)opts->trim[2];
    if (use_tr) {
        P.bz = (os[0] > 1) ? 2 : 1;
        P.by = (os[0] > 1) ? 2 : 4;
    } else {
        P.bz = (os[0] > 1) ? 4 : 1;
        P.by = (os[0] > 1) ? 4 : 16;
    }
    const int brick_x = use_tr ? kTrBrickX : kBrickX;
    P.nbz = (P.oz + P.bz - 1) / P.bz;
    P.nby = (P.oy + P.by - 1) / P.by;
    P.nbx = (P.ox + brick_x - 1) / brick_x;
    const long long nblocks = (long long)P.nbz * P.nby * P.nbx;
    if (nblocks > 0x7fffffffLL) return mvs_fail(c, MVS_ERR_UNSUPPORTED, "chunk too large for one launch");

    MVS_HIP_TRY(c, hipEventRecord(c->ev_start, c->stream));
    if (use_tr) {
        switch (dtype) {
            case MVS_U8: launch_fuse_tr<unsigned char, unsigned char>(P, opts->fusion, (int)nblocks, c->stream); break;
            case MVS_U16: launch_fuse_tr<unsigned short, unsigned short>(P, opts->fusion, (int)nblocks, c->stream); break;
            default: launch_fuse_tr<float, float>(P, opts->fusion, (int)nblocks, c->stream); break;
        }
    } else {
        switch (dtype) {
            case MVS_U8: launch_fuse<unsigned char, unsigned char>(P, opts->order, opts->fusion, (int)nblocks, c->stream); break;
            case MVS_U16: launch_fuse<unsigned short, unsigned short>(P, opts->order, opts->fusion, (int)nblocks, c->stream); break;
            default: launch_fuse<float, float>(P, opts->order, opts->fusion, (int)nblocks, c->stream); break;
        }
    }
    MVS_HIP_TRY(c, hipGetLastError());
    MVS_HIP_TRY(c, hipEventRecord(c->ev_stop, c->stream));
    c->timing_valid = true;

    if (opts->out_mem == MVS_MEM_HOST) {
        MVS_HIP_TRY(c, hipMemcpyAsync(out, dout, out_bytes, hipMemcpyDeviceToHost, c->stream));
        MVS_HIP_TRY(c, hipStreamSynchronize(c->stream));
    } else if (host_bytes) {
        // host slabs were staged through scratch that the next call may overwrite
        MVS_HIP_TRY(c, hipStreamSynchronize(c->stream));
    }
    return MVS_OK;
}

namespace {
// shared host-side helper of mvs_resample / mvs_blend_weights
int stage_single_view(MvsContext* c, const mvs_view_t* view, int ndim, bool need_data, DevView* d) {
    const void* dptr = view->data;
    if (need_data) {
        size_t es = mvs_dtype_size(view->dtype);
        if (!es || !view->data) return mvs_fail(c, MVS_ERR_INVALID_ARG, "bad view dtype/data");
        if (view->mem == MVS_MEM_HOST) {
            if (view->stride[1] != view->shape[2] || view->stride[0] != view->shape[1] * view->shape[2])
                return mvs_fail(c, MVS_ERR_UNSUPPORTED, "host slabs must be C-contiguous");
            size_t nb = (size_t)view->shape[0] * view->shape[1] * view->shape[2] * es;
            void* s = mvs_scratch(c, 0, nb);
            if (!s) return MVS_ERR_HIP;
            MVS_HIP_TRY(c, hipMemcpyAsync(s, view->data, nb, hipMemcpyHostToDevice, c->stream));
            dptr = s;
        }
    }
    mvs_view_t tmp = *view;
    if (!need_data) { tmp.shape[0] = tmp.shape[1] = tmp.shape[2] = 1; tmp.stride[0] = tmp.stride[1] = tmp.stride[2] = 1; }
    return fill_dev_view(c, tmp, ndim, dptr, d);
}
}  // namespace

extern "C" int mvs_resample(int device, const mvs_view_t* view, const int64_t out_shape[3],
                            int32_t order, float cval, float* out, int32_t out_mem) {
    MvsContext* c;
    int rc = mvs_check_ready(device, &c);
    if (rc) return rc;
    std::lock_guard<std::mutex> lock(c->mu);
    if (!view || !out_shape || !out) return mvs_fail(c, MVS_ERR_INVALID_ARG, "mvs_resample: NULL argument");
    if (order != 0 && order != 1) return mvs_fail(c, MVS_ERR_UNSUPPORTED, "mvs_resample: order %d (only 0|1)", order);
    for (int k = 0; k < 3; ++k)
        if (out_shape[k] < 1 || out_shape[k] > 0x7fffffffLL)
            return mvs_fail(c, MVS_ERR_INVALID_ARG, "mvs_resample: bad out_shape");
    MVS_HIP_TRY(c, hipSetDevice(device));
    DevView d;
    rc = stage_single_view(c, view, 3, true, &d);
    if (rc) return rc;
    const long long n = (long long)out_shape[0] * out_shape[1] * out_shape[2];
    float* dout = out;
    if (out_mem == MVS_MEM_HOST) {
        dout = (float*)mvs_scratch(c, 1, (size_t)n * 4);
        if (!dout) return MVS_ERR_HIP;
    }
    int nblocks = (int)std::min<long long>((n + 255) / 256, 256 * 16);
    MVS_HIP_TRY(c, hipEventRecord(c->ev_start, c->stream));
#define MVS_RS(T, O) hipLaunchKernelGGL((resample_kernel<T, O>), dim3(nblocks), dim3(256), 0, c->stream, d, dout, \
                                        (int)out_shape[0], (int)out_shape[1], (int)out_shape[2], cval)
    switch (view->dtype) {
        case MVS_U8: if (order) MVS_RS(unsigned char, 1); else MVS_RS(unsigned char, 0); break;
        case MVS_U16: if (order) MVS_RS(unsigned short, 1); else MVS_RS(unsigned short, 0); break;
        default: if (order) MVS_RS(float, 1); else MVS_RS(float, 0); break;
    }
#undef MVS_RS
    MVS_HIP_TRY(c, hipGetLastError());
    MVS_HIP_TRY(c, hipEventRecord(c->ev_stop, c->stream));
    c->timing_valid = true;
    if (out_mem == MVS_MEM_HOST)
        MVS_HIP_TRY(c, hipMemcpyAsync(out, dout, (size_t)n * 4, hipMemcpyDeviceToHost, c->stream));
    MVS_HIP_TRY(c, hipStreamSynchronize(c->stream));
    return MVS_OK;
}

extern "C" int mvs_blend_weights(int device, const mvs_view_t* view, int32_t ndim,
                                 const int64_t out_shape[3], float* out, int32_t out_mem) {
    MvsContext* c;
    int rc = mvs_check_ready(device, &c);
    if (rc) return rc;
    std::lock_guard<std::mutex> lock(c->mu);
    if (!view || !out_shape || !out) return mvs_fail(c, MVS_ERR_INVALID_ARG, "mvs_blend_weights: NULL argument");
    if (ndim != 2 && ndim != 3) return mvs_fail(c, MVS_ERR_INVALID_ARG, "mvs_blend_weights: ndim must be 2 or 3");
    MVS_HIP_TRY(c, hipSetDevice(device));
    DevView d;
    rc = stage_single_view(c, view, ndim, false, &d);
    if (rc) return rc;
    const long long n = (long long)out_shape[0] * out_shape[1] * out_shape[2];
    float* dout = out;
    if (out_mem == MVS_MEM_HOST) {
        dout = (float*)mvs_scratch(c, 1, (size_t)n * 4);
        if (!dout) return MVS_ERR_HIP;
    }
    int nblocks = (int)std::min<long long>((n + 255) / 256, 256 * 16);
    MVS_HIP_TRY(c, hipEventRecord(c->ev_start, c->stream));
    hipLaunchKernelGGL(blend_kernel, dim3(nblocks), dim3(256), 0, c->stream, d, dout, (int)out_shape[0],
                       (int)out_shape[1], (int)out_shape[2]);
    MVS_HIP_TRY(c, hipGetLastError());
    MVS_HIP_TRY(c, hipEventRecord(c->ev_stop, c->stream));
    c->timing_valid = true;
    if (out_mem == MVS_MEM_HOST)
        MVS_HIP_TRY(c, hipMemcpyAsync(out, dout, (size_t)n * 4, hipMemcpyDeviceToHost, c->stream));
    MVS_HIP_TRY(c, hipStreamSynchronize(c->stream));
    return MVS_OK;
}
