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
#include "mvs_fuse_dev.h"
#include "mvs_fuse_tr.h"

#include <cmath>
#include <cstring>
#include <type_traits>

namespace {

constexpr int kBrickX = 64;      // voxels along x per brick (16 lanes x 4 voxels)
constexpr int kVPT = 4;          // voxels per thread along x
constexpr int kLdsTables = 8;    // blend tables staged in LDS per pass

struct FuseParams {
    const DevView* views;
    int nviews;
    void* out;
    int oz, oy, ox;        // shape of the (trimmed) result
    int tz, ty, tx;        // trim: chunk index = result index + trim
    int nbz, nby, nbx;     // brick grid
    int bz, by;            // brick extent along z and y
    const int* cull;       // fast path: [6][nviews] valid chunk-index boxes (zlo,zhi,ylo,yhi,xlo,xhi)
    int ablate;            // profiling only: 1 skip weights, 2 skip loads, 4 skip epilogue math, 8 skip accumulate
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
constexpr int kTrBrickX = 256;   // voxels along x per wavefront (64 lanes x 4)
constexpr int kTrRows = 4;       // consecutive y rows per wavefront
constexpr int kTrPlanes = 4;     // z planes per workgroup (one per wavefront)

#define MVS_GLOBAL __attribute__((address_space(1)))

template <typename T> struct RowVec;
template <> struct RowVec<unsigned short> { typedef unsigned short v4 __attribute__((ext_vector_type(4), aligned(2))); };
template <> struct RowVec<unsigned char> { typedef unsigned char v4 __attribute__((ext_vector_type(4), aligned(1))); };
template <> struct RowVec<float> { typedef float v4 __attribute__((ext_vector_type(4), aligned(4))); };

// x-interpolated values of a lane's 4 consecutive output voxels from one input row; p points at
// the first tap of voxel 0.  `vec`: the 5-element window p[0..4] lies inside the slab allocation,
// so it is fetched with one (unaligned) vector load + one scalar load even if some of the lane's
// voxels are out of bounds (their values are discarded by the caller).  Otherwise (first/last
// elements of a slab) every element is guarded.
template <typename TIn>
__device__ __forceinline__ void row_taps(const MVS_GLOBAL TIn* p, bool fracx, float wx, bool vec, int jlo, int jhi,
                                         float r[4]) {
    float e[5];
    if (vec) {
        typedef typename RowVec<TIn>::v4 V4;
        const V4 v = *reinterpret_cast<const MVS_GLOBAL V4*>(p);
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

constexpr int kRowValid = 1, kRowInside = 2, kRowAllOne = 4, kRowXTab = 8;
constexpr int kXTabMargin = 4;   // table entries kept on either side of [lo_x, hi_x] (a lane reads 4 consecutive ones)
constexpr int kCand = 16;   // candidate views evaluated per round (16 views x 4 rows = 64 lanes)



struct TrParams {
    const TrView* views;
    const int* cull;       // SoA [6][nviews]: zlo, zhi, ylo, yhi, xlo, xhi
    const float* xtab;     // per-view blend weight along x for rows in the yz interior of the view
    int* overflow;         // set to 1 if a column meets more than 64 views (the launch is then redone generically)
    int nviews;
    void* out;
    int oz, oy, ox;
    int tz, ty, tx;
    int nbz, nby, nbx;
    int is3d;
    int ablate;
};

// Blend weight along x of one view for rows whose bracketing (z, y) support nodes are all >= sx (and sx >= 1):
// there G1 = sx, the ramp zone lies in the first cell where W(x) = u(x) * sx, and the weight is 1 beyond it --
// the same profile for all such rows, tabulated once per call: tab[off + (x - lo_x)], x in [lo_x - margin, hi_x + margin].
__global__ void xweight_table_kernel(const TrView* __restrict__ views, int nviews, float* __restrict__ tab) {
    const int v = blockIdx.y;
    if (v >= nviews) return;
    const TrView& V = views[v];
    const int len = V.hi[2] - V.lo[2] + 1;
    if (len <= 0) return;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x - kXTabMargin; i < len + kXTabMargin; i += gridDim.x * blockDim.x) {
        const int x = V.lo[2] + i;
        const float u = fold_u(x, V.sup_ilo[2], V.sup_flo[2], V.sup_ihi[2], V.sup_fhi[2], V.sup_k[2]);
        tab[V.xtab_off + i] = (u >= 0.f) ? blend_ramp_nb(u * V.ws[2]) : 0.f;
    }
}

union F4I4 {
    float4 f;
    int4 i;
};

typedef unsigned int u32x2_t __attribute__((ext_vector_type(2)));
typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));

// Raw stencil-row fetch through a buffer resource: 4 consecutive elements (+ the 5th for the
// x interpolation) per lane, any alignment, hardware bounds check (out-of-slab reads return 0).
// The raw dwords live in a plain register array w[5]; decode() turns them into 5 floats.
template <typename T, bool FIVE> struct RowIO;
template <bool FIVE> struct RowIO<unsigned short, FIVE> {
    static __device__ __forceinline__ void load(__amdgpu_buffer_rsrc_t r, int vo, unsigned int (&w)[5]) {
        const u32x2_t a = __builtin_amdgcn_raw_buffer_load_b64(r, vo, 0, 0);
        w[0] = a.x; w[1] = a.y;
        if (FIVE) w[2] = __builtin_amdgcn_raw_buffer_load_b16(r, vo + 8, 0, 0);
    }
    static __device__ __forceinline__ void decode(const unsigned int (&w)[5], float (&v)[5]) {
        v[0] = (float)(w[0] & 0xffffu); v[1] = (float)(w[0] >> 16); v[2] = (float)(w[1] & 0xffffu); v[3] = (float)(w[1] >> 16);
        v[4] = FIVE ? (float)(w[2] & 0xffffu) : 0.f;
    }
};
template <bool FIVE> struct RowIO<unsigned char, FIVE> {
    static __device__ __forceinline__ void load(__amdgpu_buffer_rsrc_t r, int vo, unsigned int (&w)[5]) {
        w[0] = __builtin_amdgcn_raw_buffer_load_b32(r, vo, 0, 0);
        if (FIVE) w[1] = __builtin_amdgcn_raw_buffer_load_b8(r, vo + 4, 0, 0);
    }
    static __device__ __forceinline__ void decode(const unsigned int (&w)[5], float (&v)[5]) {
        v[0] = (float)(w[0] & 0xffu); v[1] = (float)((w[0] >> 8) & 0xffu); v[2] = (float)((w[0] >> 16) & 0xffu); v[3] = (float)(w[0] >> 24);
        v[4] = FIVE ? (float)(w[1] & 0xffu) : 0.f;
    }
};
template <bool FIVE> struct RowIO<float, FIVE> {
    static __device__ __forceinline__ void load(__amdgpu_buffer_rsrc_t r, int vo, unsigned int (&w)[5]) {
        const u32x4_t a = __builtin_amdgcn_raw_buffer_load_b128(r, vo, 0, 0);
        w[0] = a.x; w[1] = a.y; w[2] = a.z; w[3] = a.w;
        if (FIVE) w[4] = __builtin_amdgcn_raw_buffer_load_b32(r, vo + 16, 0, 0);
    }
    static __device__ __forceinline__ void decode(const unsigned int (&w)[5], float (&v)[5]) {
        v[0] = __uint_as_float(w[0]); v[1] = __uint_as_float(w[1]); v[2] = __uint_as_float(w[2]); v[3] = __uint_as_float(w[3]);
        v[4] = FIVE ? __uint_as_float(w[4]) : 0.f;
    }
};

__device__ __forceinline__ int rl(int x, int l) { return __builtin_amdgcn_readlane(x, l); }
__device__ __forceinline__ float rlf(float x, int l) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(x), l)); }

// A vector buffer load whose byte range is not entirely inside the slab (it starts before the
// first byte, ends after the last one, or a dword straddles either) comes back as 0 from the
// bounds check although part of it is in range.  Only windows touching the very first / last
// elements of a slab are affected; they are re-fetched element by element here.
template <typename TIn>
__device__ __forceinline__ void row_refetch(__amdgpu_buffer_rsrc_t r, int o, unsigned int (&w)[5]) {
    float v[5];
#pragma unroll
    for (int j = 0; j < 5; ++j) {
        const int oj = o + j * (int)sizeof(TIn);
        if (sizeof(TIn) == 2) v[j] = (float)(unsigned short)__builtin_amdgcn_raw_buffer_load_b16(r, oj, 0, 0);
        else if (sizeof(TIn) == 1) v[j] = (float)(unsigned char)__builtin_amdgcn_raw_buffer_load_b8(r, oj, 0, 0);
        else {
            v[j] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(r, oj, 0, 0));
            // keep the dword loads separate: merged into one dwordx4 they would fail the range
            // check as a whole again
            asm volatile("" ::: "memory");
        }
    }
    if (sizeof(TIn) == 2) {
        w[0] = (unsigned)v[0] | ((unsigned)v[1] << 16);
        w[1] = (unsigned)v[2] | ((unsigned)v[3] << 16);
        w[2] = (unsigned)v[4];
    } else if (sizeof(TIn) == 1) {
        w[0] = (unsigned)v[0] | ((unsigned)v[1] << 8) | ((unsigned)v[2] << 16) | ((unsigned)v[3] << 24);
        w[1] = (unsigned)v[4];
    } else {
#pragma unroll
        for (int j = 0; j < 5; ++j) w[j] = __float_as_uint(v[j]);
    }
}

constexpr int kTrGroups = 8;   // row groups (of kTrRows rows) a wavefront walks with one culled view list

// Wavefront-independent fast-path kernel: no hand-offs between wavefronts, no barriers.
// Each wavefront owns a column of kTrGroups x kTrRows consecutive y rows x 256 voxels of one z plane.
//  1. lanes test 64 views at a time against the column's box (coalesced SoA loads + ballot);
//  2. up to 16 surviving views are spread over the lanes (lane>>2 = view slot): every lane gathers
//     its view's record once; per row group, lane (slot, r) evaluates the row-uniform part of the
//     blend weight of row r (nodes G1, dG, flags) from registers;
//  3. per group a uniform loop walks the candidates: per-view constants and row nodes come out of
//     the lanes with v_readlane (scalar registers), all stencil rows of the view are fetched
//     back-to-back with bounds-checked buffer loads (one memory round trip per view and group),
//     then interpolated and accumulated.
// Accumulators per voxel: num = sum w*v, den = sum w, and (last, wlast) = value/weight of the last
// view that contributed with a ramp weight.  A voxel whose only contributor has weight w < 1 must
// come out as v exactly (normalised weight w/w == 1 in the reference): that is den == wlast.
template <typename TIn, typename TOut, int FUSION>
__global__ __launch_bounds__(256) void fuse_tr_kernel(TrParams P) {
    __shared__ int s_list[4][64];
    constexpr bool ISF = std::is_floating_point<TIn>::value;
    constexpr int ES = (int)sizeof(TIn);
    constexpr bool WA = (FUSION == MVS_FUSE_WEIGHTED_AVERAGE);
    constexpr int kColRows = kTrGroups * kTrRows;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    int b = blockIdx.x;
    const int bx = b % P.nbx;
    b /= P.nbx;
    const int byi = b % P.nby;
    const int bzi = b / P.nby;
    const int z = P.is3d ? bzi * kTrPlanes + wave : bzi;
    const int ycol = P.is3d ? byi * kColRows : (byi * 4 + wave) * kColRows;   // first row of the column
    const int x0 = bx * kTrBrickX + lane * kVPT;
    if (z >= P.oz || ycol >= P.oy) return;   // wave uniform; no barriers below
    const int zc = z + P.tz, xc0 = x0 + P.tx;   // chunk indices
    const int ycA = ycol + P.ty, ycB = min(ycol + kColRows, P.oy) - 1 + P.ty;
    const int bx0 = bx * kTrBrickX + P.tx, bx1 = min(bx * kTrBrickX + kTrBrickX, P.ox) - 1 + P.tx;

    // ---- 1. cull: all views whose box meets this column (view order preserved) ----
    int ntot = 0;
    for (int base = 0; base < P.nviews; base += 64) {
        const int v = base + lane;
        bool act = false;
        if (v < P.nviews) {
            const int* cb = P.cull + v;
            const int nv = P.nviews;
            act = cb[0] <= zc && cb[nv] >= zc && cb[2 * nv] <= ycB && cb[3 * nv] >= ycA && cb[4 * nv] <= bx1 &&
                  cb[5 * nv] >= bx0;
        }
        const unsigned long long m = __ballot(act);
        const int pos = ntot + __popcll(m & ((1ull << lane) - 1ull));
        if (act && pos < 64) s_list[wave][pos] = v;
        ntot += (int)__popcll(m);
    }
    // more than 64 views on one column: beyond this kernel's list -- flag it, mvs_fuse_chunk redoes the chunk
    // with the generic kernel (only possible when the chunk has more than 64 views at all)
    if (ntot > 64) {
        if (lane == 0) atomicExch(P.overflow, 1);
        ntot = 64;
    }

    for (int g = 0; g < kTrGroups; ++g) {
        const int y0 = ycol + g * kTrRows;
        if (y0 >= P.oy) break;
        const int yc0 = y0 + P.ty;
        const int yc1 = min(y0 + kTrRows, P.oy) - 1 + P.ty;

        float num[kTrRows][kVPT], den[kTrRows][kVPT], last[kTrRows][kVPT], wlast[kTrRows][kVPT];
#pragma unroll
        for (int r = 0; r < kTrRows; ++r)
#pragma unroll
            for (int j = 0; j < kVPT; ++j) {
                num[r][j] = (FUSION == MVS_FUSE_MAX) ? -INFINITY : 0.f;
                den[r][j] = 0.f;
                last[r][j] = 0.f;
                wlast[r][j] = 0.f;
            }

        for (int c0 = 0; c0 < ntot; c0 += kCand) {
            const int ncand = min(ntot - c0, kCand);
            // ---- 2. lane (c, r) = (lane >> 2, lane & 3): record of candidate c, row-uniform part of row r ----
            const int c_ = lane >> 2, r_ = lane & 3;
            const bool has = c_ < ncand;
            const int myv = s_list[wave][c0 + (has ? c_ : 0)];
            const float4* rec = reinterpret_cast<const float4*>(P.views + myv);
            F4I4 q0, q1, q2, q3, q4, q5, q6, q7, q8, q9;
            q0.f = rec[0]; q1.f = rec[1]; q2.f = rec[2]; q3.f = rec[3]; q4.f = rec[4];
            q5.f = rec[5]; q6.f = rec[6]; q7.f = rec[7]; q8.f = rec[8]; q9.f = rec[9];
            // field map (dwords): 0-2 lo, 3-5 hi, 6-8 io, 9 wnz, 10-12 fw, 13 pad, 14-15 data, 16-17 span,
            // 18 stride_y, 19 stride_z, 20-22 sup_ilo, 23-25 sup_ihi, 26-28 sup_flo, 29-31 sup_fhi,
            // 32-34 sup_k, 35-37 ws
            const int lo_z = q0.i.x, lo_y = q0.i.y, lo_x = q0.i.z, hi_z = q0.i.w, hi_y = q1.i.x, hi_x = q1.i.y;
            const int io_z = q1.i.z, io_y = q1.i.w, io_x = q2.i.x, wnz = q2.i.y;
            const float fw_z = q2.f.z, fw_y = q2.f.w, fw_x = q3.f.x;
            const int xtab_off = q3.i.y;
            const int data_lo = q3.i.z, data_hi = q3.i.w, span_lo = q4.i.x;
            const int st_y = q4.i.z, st_z = q4.i.w;
            const int silo_z = q5.i.x, silo_y = q5.i.y, silo_x = q5.i.z, sihi_z = q5.i.w, sihi_y = q6.i.x, sihi_x = q6.i.y;
            const float sflo_z = q6.f.z, sflo_y = q6.f.w, sflo_x = q7.f.x, sfhi_z = q7.f.y, sfhi_y = q7.f.z, sfhi_x = q7.f.w;
            const float sk_z = q8.f.x, sk_y = q8.f.y, sk_x = q8.f.z, ws_z = q8.f.w, ws_y = q9.f.x, ws_x = q9.f.y;

            int flags = 0;
            float G1 = 0.f, dG = 0.f;
            {
                const int ry = yc0 + r_;
                const int xa = max(bx0, lo_x), xb = min(bx1, hi_x);
                if (has && zc >= lo_z && zc <= hi_z && ry >= lo_y && ry <= hi_y && ry <= yc1 && xa <= xb) {
                    flags = kRowValid;
                    if (WA) {
                        float az0 = INFINITY, az1 = INFINITY, fz = 0.f, uz = 0.f;
                        const bool has_z = wnz > 1;
                        if (has_z) {
                            uz = fold_u(zc, silo_z, sflo_z, sihi_z, sfhi_z, sk_z);
                            tent_cell(fmaxf(uz, 0.f), ws_z, az0, az1, fz);
                        }
                        const float uy = fold_u(ry, silo_y, sflo_y, sihi_y, sfhi_y, sk_y);
                        if (uz >= 0.f && uy >= 0.f) {
                            flags |= kRowInside;
                            float ay0, ay1, fy;
                            tent_cell(uy, ws_y, ay0, ay1, fy);
                            const float uz_ = 1.f - fz, uy_ = 1.f - fy;
                            const float m00 = fminf(az0, ay0), m01 = fminf(az0, ay1), m10 = fminf(az1, ay0), m11 = fminf(az1, ay1);
                            const float a1 = ws_x, a2 = 2.f * ws_x;
                            // same association as the table interpolation: lerp along y, then z
                            float g0 = fmaf(fminf(m01, a1), fy, fminf(m00, a1) * uy_);
                            float g1 = fmaf(fminf(m11, a1), fy, fminf(m10, a1) * uy_);
                            G1 = has_z ? fmaf(g1, fz, g0 * uz_) : g0;
                            g0 = fmaf(fminf(m01, a2), fy, fminf(m00, a2) * uy_);
                            g1 = fmaf(fminf(m11, a2), fy, fminf(m10, a2) * uy_);
                            const float G2 = has_z ? fmaf(g1, fz, g0 * uz_) : g0;
                            dG = G2 - G1;
                            // every bracketing (z,y) node is at least sx from the border, so G1 == sx exactly and G2 >= G1;
                            // with sx >= 1 the ramp zone (W < 1) lies inside the first cell where W = u * sx, and beyond
                            // it the weight is 1 whatever G2 is: the row follows the view's tabulated x profile
                            if (fminf(fminf(m00, m01), fminf(m10, m11)) >= a1 && a1 >= 1.f) {
                                G1 = a1;
                                flags |= kRowXTab;
                            }
                            // W(x) is concave piecewise linear along the row: >= 1 at both ends of the
                            // covered part of the segment => every voxel's weight is exactly 1.
                            const float ua = fold_u(xa, silo_x, sflo_x, sihi_x, sfhi_x, sk_x);
                            const float ub = fold_u(xb, silo_x, sflo_x, sihi_x, sfhi_x, sk_x);
                            if (ua >= 0.f && ub >= 0.f && row_profile(ua, G1, dG) >= 1.f && row_profile(ub, G1, dG) >= 1.f)
                                flags |= kRowAllOne;
                        }
                    }
                }
            }

            // ---- 3. uniform walk over the candidates ----
            if (P.ablate & 16) {   // profiling: stop after the row-uniform part
                num[0][0] += G1 + dG + (float)flags;
                continue;
            }
            for (int c = 0; c < ncand; ++c) {
                const int l0 = c * 4;
                int fl[kTrRows];
                int anyf = 0, needw = 0, needt = 0;
#pragma unroll
                for (int r = 0; r < kTrRows; ++r) {
                    fl[r] = rl(flags, l0 + r);
                    anyf |= fl[r];
                    needw |= (fl[r] & kRowValid) && !(fl[r] & (kRowAllOne | kRowXTab));
                    needt |= (fl[r] & kRowValid) && (fl[r] & kRowXTab) && !(fl[r] & kRowAllOne);
                }
                if (!(anyf & kRowValid)) continue;
                const int xlo = rl(lo_x, l0), xhi = rl(hi_x, l0);
                const bool fullseg = (xlo <= bx0) && (xhi >= bx1);   // every lane of the segment is inside the view
                const float wz = rlf(fw_z, l0), wy = rlf(fw_y, l0), wx = rlf(fw_x, l0);
                const bool anyfrac = (wz > 0.f) || (wy > 0.f) || (wx > 0.f);
                const int sy = rl(st_y, l0), sz = rl(st_z, l0);
                const unsigned long long dptr = ((unsigned long long)(unsigned)rl(data_hi, l0) << 32) | (unsigned)rl(data_lo, l0);
                const int nbytes = rl(span_lo, l0) * ES;
                const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)dptr, 0, nbytes, 0x00020000);
                // byte offset of voxel 0's first tap; lanes far outside the slab get a sentinel (reads return 0)
                const long long ebytes = ((long long)(zc + rl(io_z, l0)) * sz + (long long)(yc0 + rl(io_y, l0)) * sy +
                                          (xc0 + rl(io_x, l0))) * ES;
                const bool e_ok = (ebytes > -(1ll << 29)) && (ebytes < (1ll << 30) + (1ll << 29));
                const int vo = e_ok ? (int)ebytes : 0x7f000000;
                constexpr int WB = 5 * ES;   // bytes of one 5-element window

                if (P.ablate & 32) {   // profiling: stop after the per-view scalar setup
                    num[0][1] += (float)vo + wz + wy + wx + (float)xlo + (float)fullseg;
                    continue;
                }
                float S[kTrRows + 1][kVPT];
                if (anyfrac) {
                    unsigned int raw0[kTrRows + 1][5], raw1[kTrRows + 1][5];
#pragma unroll
                    for (int k = 0; k <= kTrRows; ++k) {
                        RowIO<TIn, true>::load(rsrc, vo + k * sy * ES, raw0[k]);
                        RowIO<TIn, true>::load(rsrc, vo + (k * sy + sz) * ES, raw1[k]);
                    }
#pragma unroll
                    for (int k = 0; k <= kTrRows; ++k) {
                        const int o0 = vo + k * sy * ES, o1 = vo + (k * sy + sz) * ES;
                        if (__any((o0 < 0 && o0 + WB > 0) || (o0 < nbytes && o0 + WB > nbytes))) row_refetch<TIn>(rsrc, o0, raw0[k]);
                        if (__any((o1 < 0 && o1 + WB > 0) || (o1 < nbytes && o1 + WB > nbytes))) row_refetch<TIn>(rsrc, o1, raw1[k]);
                    }
                    const float ux = 1.f - wx, uz = 1.f - wz;
#pragma unroll
                    for (int k = 0; k <= kTrRows; ++k) {
                        float e[5], f[5];
                        RowIO<TIn, true>::decode(raw0[k], e);
                        RowIO<TIn, true>::decode(raw1[k], f);
#pragma unroll
                        for (int j = 0; j < kVPT; ++j) {
                            const float t0 = fmaf(e[j + 1], wx, e[j] * ux);
                            const float t1 = fmaf(f[j + 1], wx, f[j] * ux);
                            S[k][j] = fmaf(t1, wz, t0 * uz);
                        }
                    }
                } else {
                    unsigned int raw0[kTrRows][5];
#pragma unroll
                    for (int k = 0; k < kTrRows; ++k) RowIO<TIn, false>::load(rsrc, vo + k * sy * ES, raw0[k]);
#pragma unroll
                    for (int k = 0; k < kTrRows; ++k) {
                        const int o0 = vo + k * sy * ES;
                        if (__any((o0 < 0 && o0 + WB > 0) || (o0 < nbytes && o0 + WB > nbytes))) row_refetch<TIn>(rsrc, o0, raw0[k]);
                    }
#pragma unroll
                    for (int k = 0; k < kTrRows; ++k) {
                        float e[5];
                        RowIO<TIn, false>::decode(raw0[k], e);
#pragma unroll
                        for (int j = 0; j < kVPT; ++j) S[k][j] = e[j];
                    }
#pragma unroll
                    for (int j = 0; j < kVPT; ++j) S[kTrRows][j] = 0.f;
                }
                if (P.ablate & 2) {
#pragma unroll
                    for (int k = 0; k <= kTrRows; ++k)
#pragma unroll
                        for (int j = 0; j < kVPT; ++j) S[k][j] = (float)(j + k);
                }

                bool okj[kVPT];
                {
                    const int jlo = xlo - xc0, jw = xhi - xlo;
#pragma unroll
                    for (int j = 0; j < kVPT; ++j) okj[j] = (unsigned)(j - jlo) <= (unsigned)jw;
                }
                float dlb = 0.f, dhb = 0.f, flo_ = 0.f, fhi_ = 0.f, kx = 0.f;     // (one rounding per voxel: see mvs_fuse_region.hip)
                if (WA && needw) {
                    kx = rlf(sk_x, l0);
                    dlb = (float)(xc0 - rl(silo_x, l0)); flo_ = rlf(sflo_x, l0);
                    dhb = (float)(rl(sihi_x, l0) - xc0); fhi_ = rlf(sfhi_x, l0);
                }
                const float uy = 1.f - wy;
                float wt[kVPT] = {1.f, 1.f, 1.f, 1.f};
                if (WA && needt) {
                    // the view's x profile, shared by all yz-interior rows of this group
                    const int len = xhi - xlo + 1;
                    int ti = xc0 - xlo;
                    ti = min(max(ti, -kXTabMargin), len + kXTabMargin - kVPT);   // lanes off the view read a clamped window (masked below)
                    const float4 t4 = *reinterpret_cast<const float4*>(P.xtab + rl(xtab_off, l0) + ti);
                    wt[0] = t4.x; wt[1] = t4.y; wt[2] = t4.z; wt[3] = t4.w;
                }

#pragma unroll
                for (int r = 0; r < kTrRows; ++r) {
                    if (!(fl[r] & kRowValid)) continue;   // wave uniform
                    float val[kVPT];
#pragma unroll
                    for (int j = 0; j < kVPT; ++j) val[j] = anyfrac ? fmaf(S[r + 1][j], wy, S[r][j] * uy) : S[r][j];

                    if (WA) {
                        if (P.ablate & 8) {
#pragma unroll
                            for (int j = 0; j < kVPT; ++j) num[r][j] += val[j];
                            continue;
                        }
                        const bool allone = (fl[r] & kRowAllOne) || (P.ablate & 1);
                        if (allone && fullseg && !ISF) {
#pragma unroll
                            for (int j = 0; j < kVPT; ++j) {
                                num[r][j] += val[j];
                                den[r][j] += 1.f;
                            }
                        } else if (allone) {
#pragma unroll
                            for (int j = 0; j < kVPT; ++j) {
                                const bool ok = ISF ? (okj[j] && val[j] == val[j]) : okj[j];
                                num[r][j] += ok ? val[j] : 0.f;
                                den[r][j] += ok ? 1.f : 0.f;
                            }
                        } else {
                            const bool xtab = fl[r] & kRowXTab;
                            const float G1r = rlf(G1, l0 + r), dGr = rlf(dG, l0 + r);
                            const bool inside = fl[r] & kRowInside;
                            float wr[kVPT];
                            if (xtab) {
#pragma unroll
                                for (int j = 0; j < kVPT; ++j) wr[j] = wt[j];
                            } else {
#pragma unroll
                                for (int j = 0; j < kVPT; ++j) {
                                    const float u = fminf((dlb + (float)j) - flo_, (dhb - (float)j) - fhi_) * kx;
                                    const float W = row_profile(u, G1r, dGr);
                                    wr[j] = (u >= 0.f && inside) ? blend_ramp_nb(W) : 0.f;
                                }
                            }
#pragma unroll
                            for (int j = 0; j < kVPT; ++j) {
                                float w = wr[j];
                                const bool ok = ISF ? (okj[j] && val[j] == val[j]) : okj[j];
                                w = ok ? w : 0.f;
                                const bool pos = w > 0.f;
                                const float ve = pos ? val[j] : 0.f;
                                num[r][j] = fmaf(w, ve, num[r][j]);
                                den[r][j] += w;
                                // bit-select instead of ?: -- hipcc turns the paired float selects into a select
                                // between two stack slots (scratch traffic)
                                const int pm = pos ? -1 : 0;
                                last[r][j] = __int_as_float((__float_as_int(val[j]) & pm) | (__float_as_int(last[r][j]) & ~pm));
                                wlast[r][j] = __int_as_float((__float_as_int(w) & pm) | (__float_as_int(wlast[r][j]) & ~pm));
                            }
                        }
                    } else {
#pragma unroll
                        for (int j = 0; j < kVPT; ++j) {
                            const bool ok = ISF ? (okj[j] && val[j] == val[j]) : okj[j];
                            if (FUSION == MVS_FUSE_MAX) num[r][j] = ok ? fmaxf(num[r][j], val[j]) : num[r][j];
                            else num[r][j] += ok ? val[j] : 0.f;
                            den[r][j] += ok ? 1.f : 0.f;
                        }
                    }
                }
            }
        }

        // ---- epilogue of this row group ----
        if (x0 < P.ox) {
#pragma unroll
            for (int r = 0; r < kTrRows; ++r) {
                const int y = y0 + r;
                if (y < P.oy) {
                    float o[kVPT];
#pragma unroll
                    for (int j = 0; j < kVPT; ++j) {
                        float q;
                        if (P.ablate & 4) q = num[r][j] + den[r][j] + last[r][j] + wlast[r][j];
                        else if (FUSION == MVS_FUSE_MAX) q = (den[r][j] > 0.f) ? num[r][j] : 0.f;
                        else if (WA) {
                            q = num[r][j] * __builtin_amdgcn_rcpf(den[r][j]);
                            // only contributor had a ramp weight: its normalised weight is w/w == 1 -> the value itself
                            q = (den[r][j] == wlast[r][j]) ? last[r][j] : q;
                        } else q = num[r][j] / den[r][j];
                        if (!(fabsf(q) <= 3.4028234e38f)) q = 0.f;   // 0/0, x/0, NaN -> nan_to_num; no view -> 0
                        o[j] = q;
                    }
                    store_row4<TOut>((TOut*)P.out, ((long long)z * P.oy + y) * (long long)P.ox, x0, P.ox, o);
                }
            }
        }
    }
}

// ---- host side of the fast path ---------------------------------------------------------------
// c(i) = fl(i + off) is what the generic kernel (and scipy) test against [0, n-1] for an identity
// matrix; it is monotone in i, so the in-bounds set is an integer interval found exactly here.
// Index frame (mvs_fuse_opts_t.index_origin): `off` refers to frame index i = chunk index + org and to the whole view's pixel
// grid, in which the slab starts at pixel `ioff`: in bounds iff ioff <= fl(i + off) <= ioff + n - 1.
static void exact_valid_range(double off, int n, int n_out, int* lo_out, int* hi_out, long long org = 0, long long ioff = 0) {
    auto c = [off](long long i) { return (double)i + off; };
    const double cmin = (double)ioff, cmax = (double)(ioff + n - 1);
    long long lo = (long long)ceil(cmin - off);
    while (c(lo - 1) >= cmin) --lo;
    while (c(lo) < cmin) ++lo;
    long long hi = (long long)floor(cmax - off);
    while (c(hi + 1) <= cmax) ++hi;
    while (c(hi) > cmax) --hi;
    lo -= org;
    hi -= org;
    if (lo < 0) lo = 0;
    if (hi > n_out - 1) hi = n_out - 1;
    *lo_out = (int)std::min<long long>(lo, 0x7fffffff);
    *hi_out = (int)std::max<long long>(hi, -1);   // lo > hi: the view never contributes
}

// Decide whether a view qualifies for the translation fast path and derive its constants.
static void prepare_translation_view(DevView* d, int order, int fusion, const int64_t chunk_shape[3], size_t elem_size,
                                     const int64_t* org = nullptr, const int64_t* ioff = nullptr) {
    static const int64_t zero3[3] = {0, 0, 0};
    if (!org) org = zero3;
    if (!ioff) ioff = zero3;
    d->tr_ok = 0;
    static const double I9[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
    for (int k = 0; k < 9; ++k)
        if (d->m[k] != I9[k]) return;
    const int n[3] = {d->nz, d->ny, d->nx};
    for (int k = 0; k < 3; ++k) {
        if (!(fabs(d->off[k]) < 1048576.0) || n[k] > 65536 || chunk_shape[k] > 65536) return;
        if (std::llabs(org[k]) > (1 << 24) || std::llabs(ioff[k]) > (1 << 24)) return;
        if (!(fabs(d->off[k] + (double)(org[k] - ioff[k])) < 65536.0)) return;
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
        for (int k = 0; k < 3; ++k) {
            d->sup_k[k] = 0.f;
            d->sup_ilo[k] = d->sup_ihi[k] = 0;
            d->sup_flo[k] = d->sup_fhi[k] = 0.f;
            if (k == 0 && nz == 1) continue;
            const double wm = d->wm[k * 4], wo = d->woff[k];
            if (!(wm > 0.0)) return;
            const double lo = -wo / wm;          // frame index where the support coordinate is 0
            const double hi = (4.0 - wo) / wm;   // ... and 4
            if (!(fabs(lo) < 1e6) || !(fabs(hi) < 1e6)) return;
            d->sup_k[k] = (float)wm;
            d->sup_ilo[k] = (int)((long long)floor(lo) - org[k]);     // chunk index = frame index - org (integers: exact)
            d->sup_flo[k] = (float)(lo - floor(lo));
            d->sup_ihi[k] = (int)((long long)ceil(hi) - org[k]);
            d->sup_fhi[k] = (float)(ceil(hi) - hi);
        }
    }
    d->span = (long long)(d->nz - 1) * d->stride_z + (long long)(d->ny - 1) * d->stride_y + d->nx;
    if (d->stride_z > 0x7fffffffLL || d->stride_y > 0x7fffffffLL || d->stride_z < 0 || d->stride_y < 0) return;
    // 32-bit signed byte offsets in the buffer loads; windows may start up to a few rows before / after the slab
    if ((long long)d->span * (long long)elem_size >= (1ll << 31) - (1ll << 27) || (long long)d->stride_z * (long long)elem_size >= (1ll << 26)) return;
    for (int k = 0; k < 3; ++k) {
        const double off = d->off[k];
        // slab pixel = chunk index + org + off - ioff: the integer parts are combined as integers, so the fraction (and with
        // it every interpolation weight) does not depend on which chunk / slab the voxel is seen through
        if (order == 0) {
            d->io[k] = (int)((long long)floor(off + 0.5) + org[k] - ioff[k]);
            d->fw[k] = 0.f;
        } else {
            const double f = floor(off);
            d->io[k] = (int)((long long)f + org[k] - ioff[k]);
            d->fw[k] = (float)(off - f);
        }
        exact_valid_range(off, n[k], (int)chunk_shape[k], &d->lo[k], &d->hi[k], org[k], ioff[k]);
    }
    d->pad[0] = (order == 0) ? 0.f : 1.f;   // carried into TrView.linear
    d->tr_ok = 1;
}

// Single-view resample to float32 with an arbitrary cval (transformation.py:136-139).
template <typename TIn, int ORDER>
__global__ __launch_bounds__(256) void resample_kernel(DevView V, float* out, int oz, int oy, int ox, float cval, int z0, int y0, int x0) {
    const long long n = (long long)oz * oy * ox;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        int x = (int)(i % ox);
        long long t = i / ox;
        int y = (int)(t % oy);
        int z = (int)(t / oy);
        // (z0, y0, x0): chunk index of the output box's first voxel -- the coordinates are those of the whole chunk
        const double pz = (double)(z + z0), py = (double)(y + y0), px = (double)(x + x0);
        const double cz = ((pz * V.m[0] + py * V.m[1]) + px * V.m[2]) + V.off[0];
        const double cy = ((pz * V.m[3] + py * V.m[4]) + px * V.m[5]) + V.off[1];
        const double cx = ((pz * V.m[6] + py * V.m[7]) + px * V.m[8]) + V.off[2];
        const bool inb = !(cz < 0.0 || cz > (double)(V.nz - 1) || cy < 0.0 || cy > (double)(V.ny - 1) ||
                           cx < 0.0 || cx > (double)(V.nx - 1));
        out[i] = inb ? sample_view<TIn, ORDER>(V, cz, cy, cx) : cval;
    }
}

// The same for an integer tile under a whole-pixel translation (identity matrix, integer offset: the registration crops of a
// tile grid): scipy's order-0 / order-1 value at an integer coordinate of finite data is the sample itself (second tap weight
// 0), so the resample is a crop + conversion -- 8 outputs per thread, one 16-byte (8-byte for uint8) load where the whole group
// lies inside the tile.
// `stats` (optional): per-block minimum, maximum and number of the values written from inside the tile -- [float min[nb]][float
// max[nb]][long long count[nb]], the partials mvs_rescale_pair_device otherwise gets from a pass of its own over the crop (the values
// are integers of the tile's type, so the "not a 16-bit integer" half of that count is 0).
template <typename TIn>
__global__ __launch_bounds__(256) void crop_int_kernel(const TIn* __restrict__ data, long long stride_z, long long stride_y, int nz, int ny,
                                                       int nx, int tz, int ty, int tx, float* __restrict__ out, int oz, int oy, int ox,
                                                       float cval, char* __restrict__ stats) {
    typedef TIn in8_t __attribute__((ext_vector_type(8), aligned(sizeof(TIn))));
    typedef float f4_t __attribute__((ext_vector_type(4), aligned(4)));
    const int gpr = (ox + 7) / 8;                                   // groups of 8 outputs per row
    const long long ngroups = (long long)oz * oy * gpr;
    float smn = INFINITY, smx = -INFINITY;
    long long snv = 0;
    for (long long g = (long long)blockIdx.x * blockDim.x + threadIdx.x; g < ngroups; g += (long long)gridDim.x * blockDim.x) {
        const int xg = (int)(g % gpr) * 8;
        const long long row = g / gpr;
        const int y = (int)(row % oy), z = (int)(row / oy);
        const int iz = z + tz, iy = y + ty, ix = xg + tx;
        float* o = out + row * ox + xg;
        const bool zy = iz >= 0 && iz < nz && iy >= 0 && iy < ny;
        const TIn* p = data + (long long)iz * stride_z + (long long)iy * stride_y + ix;
        if (zy && xg + 8 <= ox && ix >= 0 && ix + 8 <= nx) {
            const in8_t v = *reinterpret_cast<const in8_t*>(p);
            f4_t a, b;
            a.x = (float)v[0]; a.y = (float)v[1]; a.z = (float)v[2]; a.w = (float)v[3];
            b.x = (float)v[4]; b.y = (float)v[5]; b.z = (float)v[6]; b.w = (float)v[7];
            *reinterpret_cast<f4_t*>(o) = a;
            *reinterpret_cast<f4_t*>(o + 4) = b;
            if (stats) {
                smn = fminf(smn, fminf(fminf(fminf(a.x, a.y), fminf(a.z, a.w)), fminf(fminf(b.x, b.y), fminf(b.z, b.w))));
                smx = fmaxf(smx, fmaxf(fmaxf(fmaxf(a.x, a.y), fmaxf(a.z, a.w)), fmaxf(fmaxf(b.x, b.y), fmaxf(b.z, b.w))));
                snv += 8;
            }
        } else {
            for (int j = 0; j < 8 && xg + j < ox; ++j) {
                const bool in = zy && ix + j >= 0 && ix + j < nx;
                const float v = in ? (float)p[j] : cval;
                o[j] = v;
                if (stats && in) { smn = fminf(smn, v); smx = fmaxf(smx, v); ++snv; }
            }
        }
    }
    if (stats) {
        for (int off = 32; off > 0; off >>= 1) {
            smn = fminf(smn, __shfl_down(smn, off));
            smx = fmaxf(smx, __shfl_down(smx, off));
            snv += __shfl_down(snv, off);
        }
        __shared__ float s_mn[4], s_mx[4];
        __shared__ long long s_nv[4];
        const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
        if (lane == 0) { s_mn[wave] = smn; s_mx[wave] = smx; s_nv[wave] = snv; }
        __syncthreads();
        if (threadIdx.x == 0) {
            for (int w = 1; w < 4; ++w) { smn = fminf(smn, s_mn[w]); smx = fmaxf(smx, s_mx[w]); snv += s_nv[w]; }
            const int nb = gridDim.x;
            ((float*)stats)[blockIdx.x] = smn;
            ((float*)stats)[nb + blockIdx.x] = smx;
            ((long long*)(stats + (size_t)nb * 8))[blockIdx.x] = snv;
        }
    }
}

// The same crop taken from the RAW tile with the registration binning applied on the fly (round 5): output voxel (z, y, x) is the
// binned sample (z + tz, y + ty, x + tx) of the window -- the truncated block mean of its bz x by x bx raw voxels, exactly
// bin_mean_kernel's arithmetic (integer sum, times 1 / (bz by bx) in double, cast to the tile's type) -- or `cval` outside the
// window's nz x ny x nx binned samples.  `data` = first raw voxel of the window.  A pair then reads only the slab of each tile its
// overlap needs (15 GB per north-star mosaic instead of binning all 17 GB of tiles and reading the binned windows back), and no
// binned copy of a tile exists.  bx == 2: a thread makes 8 outputs of a row from two 16-byte loads per raw row.
template <typename TIn>
__global__ __launch_bounds__(1024) void crop_bin_kernel(const TIn* __restrict__ data, long long stride_z, long long stride_y, int nz, int ny,
                                                       int nx, int bz, int by, int bx, int tz, int ty, int tx, float* __restrict__ out, int oz,
                                                       int oy, int ox, float cval, char* __restrict__ stats) {
    typedef float f4_t __attribute__((ext_vector_type(4), aligned(4)));
    typedef unsigned int u32x4_t __attribute__((ext_vector_type(4), aligned(4)));
    const int gpr = (ox + 7) / 8;
    const long long ngroups = (long long)oz * oy * gpr;
    const double inv = 1.0 / ((double)bz * by * bx);
    // the 4-byte / 16-byte loads of the bx == 2 branches read PAIRS of 16-bit samples: every pair must start on a 4-byte boundary
    // (window base and both strides; a tile of odd width has odd strides) -- otherwise the generic branch
    const bool pairs_aligned = (((unsigned long long)data & 3ull) == 0ull) && ((stride_z & 1ll) == 0ll) && ((stride_y & 1ll) == 0ll);
    float smn = INFINITY, smx = -INFINITY;
    long long snv = 0;
    for (long long g = (long long)blockIdx.x * blockDim.x + threadIdx.x; g < ngroups; g += (long long)gridDim.x * blockDim.x) {
        // the last group of a row whose length is no multiple of 8 is moved back to end with the row (it recomputes `dup` outputs of
        // its neighbour, same values): every group of a narrow crop (51 samples for x neighbours) then takes the vector path -- no
        // partial group, no divergence inside the wavefront
        const int xg0 = (int)(g % gpr) * 8;
        const int xg = (ox >= 8 && xg0 + 8 > ox) ? ox - 8 : xg0;
        const int dup = xg0 - xg;
        const long long row = g / gpr;
        const int y = (int)(row % oy), z = (int)(row / oy);
        const int iz = z + tz, iy = y + ty, ix = xg + tx;
        float* o = out + row * ox + xg;
        const bool zy = iz >= 0 && iz < nz && iy >= 0 && iy < ny;
        const TIn* p = data + (long long)iz * bz * stride_z + (long long)iy * by * stride_y + (long long)ix * bx;
        float v[8];
        bool in[8];
        if (sizeof(TIn) == 2 && bx == 2 && pairs_aligned && zy && xg + 8 <= ox && ix >= 0 && ix + 8 <= nx) {
            unsigned int acc[8] = {0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u};       // at most 2 by bz 65535: fits for by bz <= 32767
            for (int dz = 0; dz < bz; ++dz)
                for (int dy = 0; dy < by; ++dy) {
                    const TIn* r = p + (long long)dz * stride_z + (long long)dy * stride_y;
                    const u32x4_t a = *reinterpret_cast<const u32x4_t*>(r);
                    const u32x4_t b = *reinterpret_cast<const u32x4_t*>(r + 8);
#pragma unroll
                    for (int k = 0; k < 4; ++k) { acc[k] += (a[k] & 0xffffu) + (a[k] >> 16); acc[4 + k] += (b[k] & 0xffffu) + (b[k] >> 16); }
                }
#pragma unroll
            for (int k = 0; k < 8; ++k) { v[k] = (float)(TIn)((double)acc[k] * inv); in[k] = true; }
        } else if (sizeof(TIn) == 2 && bx == 2 && pairs_aligned) {
            // a group that hangs over the crop or the window (the narrow crops of x neighbours end in one): one 4-byte load per
            // output and raw row instead of the generic double-precision loops
            for (int j = 0; j < 8; ++j) {
                in[j] = zy && xg + j < ox && ix + j >= 0 && ix + j < nx;
                v[j] = cval;
                if (!in[j]) continue;
                unsigned int acc = 0u;
                for (int dz = 0; dz < bz; ++dz)
                    for (int dy = 0; dy < by; ++dy) {
                        const unsigned int w = *reinterpret_cast<const unsigned int*>(p + (long long)dz * stride_z + (long long)dy * stride_y + (long long)j * 2);
                        acc += (w & 0xffffu) + (w >> 16);
                    }
                v[j] = (float)(TIn)((double)acc * inv);
            }
        } else {
            for (int j = 0; j < 8; ++j) {
                in[j] = zy && xg + j < ox && ix + j >= 0 && ix + j < nx;
                v[j] = cval;
                if (!in[j]) continue;
                double acc = 0.0;
                for (int dz = 0; dz < bz; ++dz)
                    for (int dy = 0; dy < by; ++dy) {
                        const TIn* r = p + (long long)dz * stride_z + (long long)dy * stride_y + (long long)j * bx;
                        for (int dx = 0; dx < bx; ++dx) acc += (double)r[dx];
                    }
                v[j] = (float)(TIn)(acc * inv);
            }
        }
        if (xg + 8 <= ox) {
            f4_t a, b;
            a.x = v[0]; a.y = v[1]; a.z = v[2]; a.w = v[3];
            b.x = v[4]; b.y = v[5]; b.z = v[6]; b.w = v[7];
            *reinterpret_cast<f4_t*>(o) = a;
            *reinterpret_cast<f4_t*>(o + 4) = b;
        } else {
            for (int j = 0; j < 8 && xg + j < ox; ++j) o[j] = v[j];
        }
        if (stats)
            for (int j = 0; j < 8; ++j)
                if (in[j] && xg + j < ox && j >= dup) { smn = fminf(smn, v[j]); smx = fmaxf(smx, v[j]); ++snv; }
    }
    if (stats) {
        for (int off = 32; off > 0; off >>= 1) {
            smn = fminf(smn, __shfl_down(smn, off));
            smx = fmaxf(smx, __shfl_down(smx, off));
            snv += __shfl_down(snv, off);
        }
        __shared__ float s_mn[16], s_mx[16];      // (up to 1024 threads per workgroup: the statistics' layout fixes the NUMBER of
        __shared__ long long s_nv[16];            // workgroups, so the kernel gets its memory-level parallelism from their size)
        const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
        if (lane == 0) { s_mn[wave] = smn; s_mx[wave] = smx; s_nv[wave] = snv; }
        __syncthreads();
        if (threadIdx.x == 0) {
            for (int w = 1; w < (int)(blockDim.x >> 6); ++w) { smn = fminf(smn, s_mn[w]); smx = fmaxf(smx, s_mx[w]); snv += s_nv[w]; }
            const int nb = gridDim.x;
            ((float*)stats)[blockIdx.x] = smn;
            ((float*)stats)[nb + blockIdx.x] = smx;
            ((long long*)(stats + (size_t)nb * 8))[blockIdx.x] = snv;
        }
    }
}

// true when `V` is a whole-pixel translation of an integer tile (see crop_int_kernel); t = the integer offsets
static bool is_integer_crop(const DevView& V, int dtype, int t[3]) {
    if (dtype != MVS_U8 && dtype != MVS_U16) return false;
    const double I[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
    for (int k = 0; k < 9; ++k)
        if (V.m[k] != I[k]) return false;
    for (int k = 0; k < 3; ++k) {
        if (!(V.off[k] == std::floor(V.off[k])) || std::fabs(V.off[k]) > 1e9) return false;
        t[k] = (int)V.off[k];
    }
    return true;
}

// Blend-weight volume of one view (weights.py:391-511), float32.
__global__ __launch_bounds__(256) void blend_kernel(DevView V, float* out, int oz, int oy, int ox, int z0, int y0, int x0) {
    const long long n = (long long)oz * oy * ox;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        int x = (int)(i % ox);
        long long t = i / ox;
        int y = (int)(t % oy);
        int z = (int)(t / oy);
        const double pz = (double)(z + z0), py = (double)(y + y0), px = (double)(x + x0);
        const double cz = ((pz * V.wm[0] + py * V.wm[1]) + px * V.wm[2]) + V.woff[0];
        const double cy = ((pz * V.wm[3] + py * V.wm[4]) + px * V.wm[5]) + V.woff[1];
        const double cx = ((pz * V.wm[6] + py * V.wm[7]) + px * V.wm[8]) + V.woff[2];
        out[i] = blend_weight<false>(V.edt, V.wnz, cz, cy, cx);
    }
}

// ---- the boxes of ALL views of a chunk in one launch each (content-based weights, mvs_gauss.hip): a chunk of a tile grid sees one
// view nearly whole and up to seven by a face, an edge or a corner, i.e. seven launches of a few microseconds of work each per
// quantity, which cost their launch latency one after the other on the chunk's stream.  View v owns the workgroups
// [blk0[v], blk0[v + 1]) of the launch (in proportion to its voxels); same arithmetic per voxel as crop_int_kernel / blend_kernel.
struct BoxItem { const void* data; long long stride_z, stride_y; int nz, ny, nx, tz, ty, tx; float* out; int oz, oy, ox, z0, y0, x0; };
struct BoxBatch { BoxItem v[8]; int blk0[9]; };
template <typename TIn>
__global__ __launch_bounds__(256) void crop_int_batch_kernel(BoxBatch B, float cval) {
    typedef TIn in8_t __attribute__((ext_vector_type(8), aligned(sizeof(TIn))));
    typedef float f4_t __attribute__((ext_vector_type(4), aligned(4)));
    int v = 0;
    while (v < 7 && (int)blockIdx.x >= B.blk0[v + 1]) ++v;
    const BoxItem& C = B.v[v];
    const int bx = (int)blockIdx.x - B.blk0[v], nbk = B.blk0[v + 1] - B.blk0[v];
    const TIn* data = (const TIn*)C.data;
    const int gpr = (C.ox + 7) / 8;
    const long long ngroups = (long long)C.oz * C.oy * gpr;
    for (long long g = (long long)bx * blockDim.x + threadIdx.x; g < ngroups; g += (long long)nbk * blockDim.x) {
        const int xg = (int)(g % gpr) * 8;
        const long long row = g / gpr;
        const int y = (int)(row % C.oy), z = (int)(row / C.oy);
        const int iz = z + C.tz, iy = y + C.ty, ix = xg + C.tx;
        float* o = C.out + row * C.ox + xg;
        const bool zy = iz >= 0 && iz < C.nz && iy >= 0 && iy < C.ny;
        const TIn* p = data + (long long)iz * C.stride_z + (long long)iy * C.stride_y + ix;
        if (zy && xg + 8 <= C.ox && ix >= 0 && ix + 8 <= C.nx) {
            const in8_t q = *reinterpret_cast<const in8_t*>(p);
            f4_t a, b;
            a.x = (float)q[0]; a.y = (float)q[1]; a.z = (float)q[2]; a.w = (float)q[3];
            b.x = (float)q[4]; b.y = (float)q[5]; b.z = (float)q[6]; b.w = (float)q[7];
            *reinterpret_cast<f4_t*>(o) = a;
            *reinterpret_cast<f4_t*>(o + 4) = b;
        } else {
            for (int j = 0; j < 8 && xg + j < C.ox; ++j) {
                const bool in = zy && ix + j >= 0 && ix + j < C.nx;
                o[j] = in ? (float)p[j] : cval;
            }
        }
    }
}
// the same from the closed form of the translation path (mvs_fuse_tr.h: float arithmetic, the weights the region kernels fuse with)
struct TrViews8 { TrView v[8]; };
__global__ __launch_bounds__(256) void blend_tr_batch_kernel(TrViews8 TV, BoxBatch B) {
    int v = 0;
    while (v < 7 && (int)blockIdx.x >= B.blk0[v + 1]) ++v;
    const BoxItem& I = B.v[v];
    const TrView& V = TV.v[v];
    const int bx = (int)blockIdx.x - B.blk0[v], nbk = B.blk0[v + 1] - B.blk0[v];
    const long long n = (long long)I.oz * I.oy * I.ox;
    for (long long i = (long long)bx * blockDim.x + threadIdx.x; i < n; i += (long long)nbk * blockDim.x) {
        const int x = (int)(i % I.ox);
        const long long t = i / I.ox;
        const int y = (int)(t % I.oy), z = (int)(t / I.oy);
        I.out[i] = blend_ramp_nb(tr_weight_profile(V, z + I.z0, y + I.y0, x + I.x0));
    }
}
__global__ __launch_bounds__(256) void blend_batch_kernel(const DevView* __restrict__ views, BoxBatch B) {
    int v = 0;
    while (v < 7 && (int)blockIdx.x >= B.blk0[v + 1]) ++v;
    const BoxItem& I = B.v[v];
    const DevView& V = views[v];
    const int bx = (int)blockIdx.x - B.blk0[v], nbk = B.blk0[v + 1] - B.blk0[v];
    const long long n = (long long)I.oz * I.oy * I.ox;
    for (long long i = (long long)bx * blockDim.x + threadIdx.x; i < n; i += (long long)nbk * blockDim.x) {
        int x = (int)(i % I.ox);
        long long t = i / I.ox;
        int y = (int)(t % I.oy);
        int z = (int)(t / I.oy);
        const double pz = (double)(z + I.z0), py = (double)(y + I.y0), px = (double)(x + I.x0);
        const double cz = ((pz * V.wm[0] + py * V.wm[1]) + px * V.wm[2]) + V.woff[0];
        const double cy = ((pz * V.wm[3] + py * V.wm[4]) + px * V.wm[5]) + V.woff[1];
        const double cx = ((pz * V.wm[6] + py * V.wm[7]) + px * V.wm[8]) + V.woff[2];
        I.out[i] = blend_weight<false>(V.edt, V.wnz, cz, cy, cx);
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

static void fill_tr_view(const DevView& d, TrView* t) {
    memset(t, 0, sizeof(*t));
    for (int k = 0; k < 3; ++k) {
        t->lo[k] = d.lo[k];
        t->hi[k] = d.hi[k];
        t->io[k] = d.io[k];
        t->fw[k] = d.fw[k];
        t->sup_ilo[k] = d.sup_ilo[k];
        t->sup_ihi[k] = d.sup_ihi[k];
        t->sup_flo[k] = d.sup_flo[k];
        t->sup_fhi[k] = d.sup_fhi[k];
        t->sup_k[k] = d.sup_k[k];
        t->ws[k] = d.ws[k];
    }
    t->wnz = d.wnz;
    t->n[0] = d.nz; t->n[1] = d.ny; t->n[2] = d.nx;
    t->linear = d.pad[0] != 0.f ? 1 : 0;
    t->data = (unsigned long long)d.data;
    t->span = d.span;
    t->stride_y = (int)d.stride_y;
    t->stride_z = (int)d.stride_z;
    t->xtab_off = 0;
}

template <typename TIn, typename TOut>
void launch_fuse_tr(const TrParams& P, int fusion, int nblocks, hipStream_t s) {
    if (fusion == MVS_FUSE_WEIGHTED_AVERAGE)
        hipLaunchKernelGGL((fuse_tr_kernel<TIn, TOut, MVS_FUSE_WEIGHTED_AVERAGE>), dim3(nblocks), dim3(256), 0, s, P);
    else if (fusion == MVS_FUSE_MAX)
        hipLaunchKernelGGL((fuse_tr_kernel<TIn, TOut, MVS_FUSE_MAX>), dim3(nblocks), dim3(256), 0, s, P);
    else
        hipLaunchKernelGGL((fuse_tr_kernel<TIn, TOut, MVS_FUSE_SIMPLE_AVERAGE>), dim3(nblocks), dim3(256), 0, s, P);
}

inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

}  // namespace

// Translation-path record of a view for callers outside this file (mvs_gauss.hip): prepare_translation_view + the chunk frame +
// fill_tr_view.  Returns false when the view does not qualify (rotated / scaled views, a support table that is not the closed form).
bool mvs_prepare_tr_view(DevView* d, int order, const int64_t chunk_shape[3], size_t elem_size, const int64_t org[3], const int64_t ioff[3],
                         TrView* out) {
    prepare_translation_view(d, order, MVS_FUSE_WEIGHTED_AVERAGE, chunk_shape, elem_size, org, ioff);
    mvs_view_to_chunk_frame(d, org, ioff);
    if (!d->tr_ok) return false;
    fill_tr_view(*d, out);
    return true;
}

int mvs_fuse_regions(MvsContext* c, const TrView* htr, const TrView* dtr, int n_views, int dtype, void* dout, const int64_t os[3],
                     const int64_t trim[3], bool* done);   // mvs_fuse_region.hip

int mvs_fuse_rows(MvsContext* c, const TrView* htr, const TrView* dtr, int n_views, int dtype, void* dout, const int64_t os[3],
                  const int64_t trim[3], bool* done);      // mvs_fuse_rows.hip

int mvs_fuse_content_based(MvsContext* c, const mvs_view_t* views, int32_t n_views,
                           const mvs_fuse_opts_t* opts, void* out);   // mvs_gauss.hip

extern "C" int mvs_fuse_chunk(int device, const mvs_view_t* views, int32_t n_views,
                              const mvs_fuse_opts_t* opts, void* out) {
    MvsContext* c;
    int rc = mvs_check_ready(device, &c);
    if (rc) return rc;
    std::lock_guard<std::recursive_mutex> lock(c->mu);
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
    MVS_HIP_TRY(c, hipSetDevice(mvs_hip_device(device)));

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
        if (!slab_base) return mvs_alloc_failed(c);
    }
    const size_t views_bytes = align_up(sizeof(DevView) * (size_t)n_views, 256);
    const size_t cull_bytes = align_up(sizeof(int) * 6 * (size_t)n_views, 256);
    const size_t params_bytes = views_bytes + cull_bytes + sizeof(TrView) * (size_t)n_views;
    DevView* hviews = (DevView*)mvs_pinned(c, params_bytes);
    if (!hviews) return mvs_alloc_failed(c);
    DevView* dviews = (DevView*)mvs_scratch(c, 2, params_bytes);
    if (!dviews) return mvs_alloc_failed(c);

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
        prepare_translation_view(&hviews[i], opts->order, opts->fusion, opts->out_shape, es, opts->index_origin, views[i].index_offset);
        mvs_view_to_chunk_frame(&hviews[i], opts->index_origin, views[i].index_offset);
    }
    bool use_tr = !c->force_generic;
    for (int i = 0; i < n_views && use_tr; ++i) use_tr = hviews[i].tr_ok != 0;
    // Float tiles with linear interpolation: scipy multiplies the second tap of every axis by its (possibly zero) weight,
    // so a NaN there poisons the sample.  Only the row kernels (weighted average) and the generic kernel read all taps;
    // the column / region kernels read one tap at integer offsets, so they are not used for such tiles.
    const bool nan_exact = (dtype == MVS_F32 && opts->order == 1);
    if (nan_exact && opts->fusion != MVS_FUSE_WEIGHTED_AVERAGE) use_tr = false;
    size_t xtab_total = 0;
    int* hcull = (int*)((char*)hviews + views_bytes);
    TrView* htr = (TrView*)((char*)hviews + views_bytes + cull_bytes);
    if (use_tr)
        for (int i = 0; i < n_views; ++i) {
            for (int k = 0; k < 3; ++k) {
                hcull[(2 * k) * n_views + i] = hviews[i].lo[k];
                hcull[(2 * k + 1) * n_views + i] = hviews[i].hi[k];
            }
            fill_tr_view(hviews[i], &htr[i]);
            htr[i].xtab_off = (int)xtab_total + kXTabMargin;
            xtab_total += (size_t)std::max(hviews[i].hi[2] - hviews[i].lo[2] + 1, 0) + 2 * kXTabMargin;
        }
    { const int rcu = mvs_upload_small(c, dviews, hviews, params_bytes); if (rcu) return rcu; }
    mvs_pinned_mark(c, 0);

    const size_t out_bytes = (size_t)os[0] * os[1] * os[2] * es;
    void* dout = out;
    if (opts->out_mem == MVS_MEM_HOST) {
        dout = mvs_scratch(c, 1, out_bytes);
        if (!dout) return mvs_alloc_failed(c);
    }

    FuseParams P;
    P.views = dviews;
    P.nviews = n_views;
    P.out = dout;
    P.oz = (int)os[0]; P.oy = (int)os[1]; P.ox = (int)os[2];
    P.tz = (int)opts->trim[0]; P.ty = (int)opts->trim[1]; P.tx = (int)opts->trim[2];
    P.cull = (const int*)((const char*)dviews + views_bytes);
    P.ablate = c->ablate;
    long long nblocks = 0;
    auto set_brick_grid = [&](bool tr) {
        if (tr) {
            P.bz = (os[0] > 1) ? kTrPlanes : 1;
            P.by = (os[0] > 1) ? kTrRows * kTrGroups : 4 * kTrRows * kTrGroups;
        } else {
            P.bz = (os[0] > 1) ? 4 : 1;
            P.by = (os[0] > 1) ? 4 : 16;
        }
        const int brick_x = tr ? kTrBrickX : kBrickX;
        P.nbz = (P.oz + P.bz - 1) / P.bz;
        P.nby = (P.oy + P.by - 1) / P.by;
        P.nbx = (P.ox + brick_x - 1) / brick_x;
        nblocks = (long long)P.nbz * P.nby * P.nbx;
    };
    set_brick_grid(use_tr);
    if (nblocks > 0x7fffffffLL) return mvs_fail(c, MVS_ERR_UNSUPPORTED, "chunk too large for one launch");

    MVS_HIP_TRY(c, hipEventRecord(c->ev_start, c->stream));
    int* tr_overflow_ptr = nullptr;
    bool regions_done = false;
    if (!regions_done && use_tr && opts->fusion == MVS_FUSE_WEIGHTED_AVERAGE && (c->rows_v1 || (dtype == MVS_F32 && opts->order == 1))) {
        // direct-load row-owning kernels.  Float tiles take them by default: they read both taps of every axis even at
        // integer offsets, so a NaN next to a tap poisons the sample exactly as scipy's zero-weight multiply does
        // (the region kernels read one tap there).  Other dtypes: opt-in ("rows_v1"), the region kernels are faster so far.
        rc = mvs_fuse_rows(c, htr, (const TrView*)((const char*)dviews + views_bytes + cull_bytes), n_views, dtype, dout, os,
                           opts->trim, &regions_done);
        if (rc) return rc;
    }
    if (!regions_done && nan_exact && use_tr) {   // float tiles the row kernels could not take: generic kernel
        use_tr = false;
        set_brick_grid(false);
        if (nblocks > 0x7fffffffLL) return mvs_fail(c, MVS_ERR_UNSUPPORTED, "chunk too large for one launch");
    }
    if (!regions_done && use_tr && opts->fusion == MVS_FUSE_WEIGHTED_AVERAGE && !c->no_regions) {
        rc = mvs_fuse_regions(c, htr, (const TrView*)((const char*)dviews + views_bytes + cull_bytes), n_views, dtype, dout, os,
                              opts->trim, &regions_done);
        if (rc) return rc;
    }
    if (regions_done) {
        // fused by the region kernel
    } else if (use_tr) {
        TrParams T;
        T.views = (const TrView*)((const char*)dviews + views_bytes + cull_bytes);
        T.cull = P.cull;
        T.nviews = n_views;
        T.out = dout;
        T.oz = P.oz; T.oy = P.oy; T.ox = P.ox;
        T.tz = P.tz; T.ty = P.ty; T.tx = P.tx;
        T.nbz = P.nbz; T.nby = P.nby; T.nbx = P.nbx;
        T.is3d = (os[0] > 1) ? 1 : 0;
        T.ablate = c->ablate;
        float* xtab = (float*)mvs_scratch(c, 3, (xtab_total + 64) * sizeof(float) + 256);
        if (!xtab) return mvs_alloc_failed(c);
        T.xtab = xtab;
        T.overflow = (int*)(xtab + xtab_total + 64);
        MVS_HIP_TRY(c, hipMemsetAsync(T.overflow, 0, sizeof(int), c->stream));
        tr_overflow_ptr = T.overflow;
        if (opts->fusion == MVS_FUSE_WEIGHTED_AVERAGE) {
            hipLaunchKernelGGL(xweight_table_kernel, dim3(16, n_views), dim3(256), 0, c->stream, T.views, n_views, xtab);
        }
        switch (dtype) {
            case MVS_U8: launch_fuse_tr<unsigned char, unsigned char>(T, opts->fusion, (int)nblocks, c->stream); break;
            case MVS_U16: launch_fuse_tr<unsigned short, unsigned short>(T, opts->fusion, (int)nblocks, c->stream); break;
            default: launch_fuse_tr<float, float>(T, opts->fusion, (int)nblocks, c->stream); break;
        }
    } else {
        switch (dtype) {
            case MVS_U8: launch_fuse<unsigned char, unsigned char>(P, opts->order, opts->fusion, (int)nblocks, c->stream); break;
            case MVS_U16: launch_fuse<unsigned short, unsigned short>(P, opts->order, opts->fusion, (int)nblocks, c->stream); break;
            default: launch_fuse<float, float>(P, opts->order, opts->fusion, (int)nblocks, c->stream); break;
        }
    }
    MVS_HIP_TRY(c, hipGetLastError());
    if (use_tr && !regions_done && n_views > 64) {
        int ovf = 0;
        MVS_HIP_TRY(c, hipMemcpyAsync(&ovf, tr_overflow_ptr, sizeof(int), hipMemcpyDeviceToHost, c->stream));
        MVS_HIP_TRY(c, hipStreamSynchronize(c->stream));
        if (ovf) {
            // a column met more than 64 views: redo the chunk with the generic kernel (bricks of 4x4x64)
            P.bz = (os[0] > 1) ? 4 : 1;
            P.by = (os[0] > 1) ? 4 : 16;
            P.nbz = (P.oz + P.bz - 1) / P.bz;
            P.nby = (P.oy + P.by - 1) / P.by;
            P.nbx = (P.ox + kBrickX - 1) / kBrickX;
            const long long nb2 = (long long)P.nbz * P.nby * P.nbx;
            if (nb2 > 0x7fffffffLL) return mvs_fail(c, MVS_ERR_UNSUPPORTED, "chunk too large for one launch");
            switch (dtype) {
                case MVS_U8: launch_fuse<unsigned char, unsigned char>(P, opts->order, opts->fusion, (int)nb2, c->stream); break;
                case MVS_U16: launch_fuse<unsigned short, unsigned short>(P, opts->order, opts->fusion, (int)nb2, c->stream); break;
                default: launch_fuse<float, float>(P, opts->order, opts->fusion, (int)nb2, c->stream); break;
            }
            MVS_HIP_TRY(c, hipGetLastError());
        }
    }
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
            if (!s) return mvs_alloc_failed(c);
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
    return mvs_resample_impl(device, view, out_shape, order, cval, out, out_mem, MvsResampleOpts{});
}

int mvs_resample_impl(int device, const mvs_view_t* view, const int64_t out_shape[3], int32_t order, float cval, float* out, int32_t out_mem,
                      const MvsResampleOpts& ro) {
    MvsContext* c;
    int rc = mvs_check_ready(device, &c);
    if (rc) return rc;
    std::lock_guard<std::recursive_mutex> lock(c->mu);
    if (!view || !out_shape || !out) return mvs_fail(c, MVS_ERR_INVALID_ARG, "mvs_resample: NULL argument");
    if (order != 0 && order != 1) return mvs_fail(c, MVS_ERR_UNSUPPORTED, "mvs_resample: order %d (only 0|1)", order);
    for (int k = 0; k < 3; ++k)
        if (out_shape[k] < 1 || out_shape[k] > 0x7fffffffLL)
            return mvs_fail(c, MVS_ERR_INVALID_ARG, "mvs_resample: bad out_shape");
    MVS_HIP_TRY(c, hipSetDevice(mvs_hip_device(device)));
    DevView d;
    rc = stage_single_view(c, view, 3, true, &d);
    if (rc) return rc;
    const long long n = (long long)out_shape[0] * out_shape[1] * out_shape[2];
    float* dout = out;
    if (out_mem == MVS_MEM_HOST) {
        dout = (float*)mvs_scratch(c, 1, (size_t)n * 4);
        if (!dout) return mvs_alloc_failed(c);
    }
    MVS_HIP_TRY(c, hipEventRecord(c->ev_start, c->stream));
    mvs_launch_resample(c, d, view->dtype, order, cval, dout, out_shape, nullptr, out_mem == MVS_MEM_DEVICE ? ro.stats : nullptr, ro.stats_k);
    MVS_HIP_TRY(c, hipGetLastError());
    MVS_HIP_TRY(c, hipEventRecord(c->ev_stop, c->stream));
    c->timing_valid = true;
    if (out_mem == MVS_MEM_HOST)
        MVS_HIP_TRY(c, hipMemcpyAsync(out, dout, (size_t)n * 4, hipMemcpyDeviceToHost, c->stream));
    if (!(ro.defer_sync && out_mem == MVS_MEM_DEVICE)) MVS_HIP_TRY(c, hipStreamSynchronize(c->stream));
    return MVS_OK;
}

extern "C" int mvs_blend_weights(int device, const mvs_view_t* view, int32_t ndim,
                                 const int64_t out_shape[3], float* out, int32_t out_mem) {
    MvsContext* c;
    int rc = mvs_check_ready(device, &c);
    if (rc) return rc;
    std::lock_guard<std::recursive_mutex> lock(c->mu);
    if (!view || !out_shape || !out) return mvs_fail(c, MVS_ERR_INVALID_ARG, "mvs_blend_weights: NULL argument");
    if (ndim != 2 && ndim != 3) return mvs_fail(c, MVS_ERR_INVALID_ARG, "mvs_blend_weights: ndim must be 2 or 3");
    MVS_HIP_TRY(c, hipSetDevice(mvs_hip_device(device)));
    DevView d;
    rc = stage_single_view(c, view, ndim, false, &d);
    if (rc) return rc;
    const long long n = (long long)out_shape[0] * out_shape[1] * out_shape[2];
    float* dout = out;
    if (out_mem == MVS_MEM_HOST) {
        dout = (float*)mvs_scratch(c, 1, (size_t)n * 4);
        if (!dout) return mvs_alloc_failed(c);
    }
    int nblocks = (int)std::min<long long>((n + 255) / 256, 256 * 16);
    MVS_HIP_TRY(c, hipEventRecord(c->ev_start, c->stream));
    hipLaunchKernelGGL(blend_kernel, dim3(nblocks), dim3(256), 0, c->stream, d, dout, (int)out_shape[0],
                       (int)out_shape[1], (int)out_shape[2], 0, 0, 0);
    MVS_HIP_TRY(c, hipGetLastError());
    MVS_HIP_TRY(c, hipEventRecord(c->ev_stop, c->stream));
    c->timing_valid = true;
    if (out_mem == MVS_MEM_HOST)
        MVS_HIP_TRY(c, hipMemcpyAsync(out, dout, (size_t)n * 4, hipMemcpyDeviceToHost, c->stream));
    MVS_HIP_TRY(c, hipStreamSynchronize(c->stream));
    return MVS_OK;
}


// One registration crop straight from a RAW integer tile: `view` describes the raw window (data = its first voxel, shape = its raw
// extent, a whole number of bins per axis; identity matrix, offset = the whole-pixel translation in BINNED pixels), `bin` the
// registration binning.  Equals mvs_bin_mean of the window followed by mvs_resample(order 1, NaN outside) bit for bit
// (crop_bin_kernel).  Device memory on both sides; never waits (the composite caller orders later work).
int mvs_crop_bin_impl(int device, const mvs_view_t* view, const int32_t bin[3], const int64_t out_shape[3], float* out, const MvsResampleOpts& ro) {
    MvsContext* c;
    int rc = mvs_check_ready(device, &c);
    if (rc) return rc;
    std::lock_guard<std::recursive_mutex> lock(c->mu);
    if (!view || !bin || !out_shape || !out) return mvs_fail(c, MVS_ERR_INVALID_ARG, "mvs_crop_bin: NULL argument");
    if ((view->dtype != MVS_U8 && view->dtype != MVS_U16) || view->mem != MVS_MEM_DEVICE || view->stride[2] != 1)
        return mvs_fail(c, MVS_ERR_UNSUPPORTED, "mvs_crop_bin: needs a device-resident uint8 / uint16 window with unit x stride");
    const double I[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
    int t[3], nb3[3];
    for (int k = 0; k < 9; ++k)
        if (view->matrix[k] != I[k]) return mvs_fail(c, MVS_ERR_UNSUPPORTED, "mvs_crop_bin: the view must be a whole-pixel translation");
    for (int k = 0; k < 3; ++k) {
        if (bin[k] < 1 || view->shape[k] < bin[k] || view->shape[k] % bin[k] || out_shape[k] < 1 || out_shape[k] > 0x7fffffffLL)
            return mvs_fail(c, MVS_ERR_INVALID_ARG, "mvs_crop_bin: bad bin / shape on axis %d", k);
        if (!(view->offset[k] == std::floor(view->offset[k])) || std::fabs(view->offset[k]) > 1e9)
            return mvs_fail(c, MVS_ERR_UNSUPPORTED, "mvs_crop_bin: the view must be a whole-pixel translation");
        t[k] = (int)view->offset[k];
        nb3[k] = (int)(view->shape[k] / bin[k]);
    }
    if ((long long)bin[0] * bin[1] > 16384) return mvs_fail(c, MVS_ERR_UNSUPPORTED, "mvs_crop_bin: bin too large");
    MVS_HIP_TRY(c, hipSetDevice(mvs_hip_device(device)));
    const long long ng = (long long)out_shape[0] * out_shape[1] * ((out_shape[2] + 7) / 8);
    const int nthr = 1024;
    int nblk = (int)std::min<long long>((ng + nthr - 1) / nthr, 256 * 8);
    char* stats = nullptr;
    if (ro.stats && ro.stats->base && ro.stats->nb > 0) {
        stats = ro.stats->base + (size_t)(ro.stats_k & 1) * (size_t)ro.stats->nb * 16;
        nblk = ro.stats->nb;
        ro.stats->done[ro.stats_k & 1] = true;
    }
    if (view->dtype == MVS_U8)
        MVS_DUP("crop", hipLaunchKernelGGL(crop_bin_kernel<unsigned char>, dim3(nblk), dim3(nthr), 0, c->stream, (const unsigned char*)view->data, (long long)view->stride[0],
                           (long long)view->stride[1], nb3[0], nb3[1], nb3[2], (int)bin[0], (int)bin[1], (int)bin[2], t[0], t[1], t[2], out,
                           (int)out_shape[0], (int)out_shape[1], (int)out_shape[2], NAN, stats));
    else
        MVS_DUP("crop", hipLaunchKernelGGL(crop_bin_kernel<unsigned short>, dim3(nblk), dim3(nthr), 0, c->stream, (const unsigned short*)view->data, (long long)view->stride[0],
                           (long long)view->stride[1], nb3[0], nb3[1], nb3[2], (int)bin[0], (int)bin[1], (int)bin[2], t[0], t[1], t[2], out,
                           (int)out_shape[0], (int)out_shape[1], (int)out_shape[2], NAN, stats));
    MVS_HIP_TRY(c, hipGetLastError());
    if (!ro.defer_sync) MVS_HIP_TRY(c, hipStreamSynchronize(c->stream));
    return MVS_OK;
}

// ---- helpers shared with mvs_gauss.hip (content-based weights) --------------------------------
int mvs_fill_dev_view(MvsContext* c, const mvs_view_t& v, int ndim, const void* dev_data, DevView* d) {
    return fill_dev_view(c, v, ndim, dev_data, d);
}

void mvs_launch_resample(MvsContext* c, const DevView& d, int dtype, int order, float cval, float* out, const int64_t shape[3], const int* box0,
                         MvsCropStats* crop_stats, int crop_stats_k) {
    const long long n = (long long)shape[0] * shape[1] * shape[2];
    int t[3];
    const int b0[3] = {box0 ? box0[0] : 0, box0 ? box0[1] : 0, box0 ? box0[2] : 0};
    if (!c->force_generic && is_integer_crop(d, dtype, t)) {
        for (int k = 0; k < 3; ++k) t[k] += b0[k];
        const long long ng = (long long)shape[0] * shape[1] * ((shape[2] + 7) / 8);
        int nb = (int)std::min<long long>((ng + 255) / 256, 256 * 16);
        char* stats = nullptr;
        if (crop_stats && crop_stats->base && crop_stats->nb > 0 && cval != cval && !box0) {      // (NaN outside: "valid" == inside the tile)
            stats = crop_stats->base + (size_t)(crop_stats_k & 1) * (size_t)crop_stats->nb * 16;
            nb = crop_stats->nb;
            crop_stats->done[crop_stats_k & 1] = true;
        }
        if (dtype == MVS_U8)
            hipLaunchKernelGGL(crop_int_kernel<unsigned char>, dim3(nb), dim3(256), 0, c->stream, (const unsigned char*)d.data, d.stride_z, d.stride_y,
                               d.nz, d.ny, d.nx, t[0], t[1], t[2], out, (int)shape[0], (int)shape[1], (int)shape[2], cval, stats);
        else
            hipLaunchKernelGGL(crop_int_kernel<unsigned short>, dim3(nb), dim3(256), 0, c->stream, (const unsigned short*)d.data, d.stride_z, d.stride_y,
                               d.nz, d.ny, d.nx, t[0], t[1], t[2], out, (int)shape[0], (int)shape[1], (int)shape[2], cval, stats);
        return;
    }
    const int nblocks = (int)std::min<long long>((n + 255) / 256, 256 * 16);
#define MVS_RS(T, O) hipLaunchKernelGGL((resample_kernel<T, O>), dim3(nblocks), dim3(256), 0, c->stream, d, out, \
                                        (int)shape[0], (int)shape[1], (int)shape[2], cval, b0[0], b0[1], b0[2])
    switch (dtype) {
        case MVS_U8: if (order) MVS_RS(unsigned char, 1); else MVS_RS(unsigned char, 0); break;
        case MVS_U16: if (order) MVS_RS(unsigned short, 1); else MVS_RS(unsigned short, 0); break;
        default: if (order) MVS_RS(float, 1); else MVS_RS(float, 0); break;
    }
#undef MVS_RS
}

void mvs_launch_blend(MvsContext* c, const DevView& d, float* out, const int64_t shape[3], const int* box0) {
    const long long n = (long long)shape[0] * shape[1] * shape[2];
    const int nblocks = (int)std::min<long long>((n + 255) / 256, 256 * 16);
    hipLaunchKernelGGL(blend_kernel, dim3(nblocks), dim3(256), 0, c->stream, d, out, (int)shape[0], (int)shape[1], (int)shape[2],
                       box0 ? box0[0] : 0, box0 ? box0[1] : 0, box0 ? box0[2] : 0);
}

// Resampled view and blend weights of the boxes of up to 8 views in two launches (see BoxBatch).  `dviews`: the views' records in
// device memory (same order as `hviews`); boxes with no voxels are skipped.  Falls back to one launch per view for what the batched
// kernels do not cover (a view that is not a whole-pixel translation of an integer tile goes through mvs_launch_resample).
void mvs_launch_boxes_batch(MvsContext* c, const DevView* hviews, const DevView* dviews, int n_views, int dtype, int order, float cval,
                            float* const* res_out, float* const* blend_out, const int64_t (*shapes)[3], const int (*box0)[3], const TrView* tr) {
    if (n_views > 8 || c->force_generic) {
        for (int i = 0; i < n_views; ++i) {
            if (shapes[i][0] * shapes[i][1] * shapes[i][2] == 0) continue;
            mvs_launch_resample(c, hviews[i], dtype, order, cval, res_out[i], shapes[i], box0[i]);
            mvs_launch_blend(c, hviews[i], blend_out[i], shapes[i], box0[i]);
        }
        return;
    }
    BoxBatch R, W;
    memset(&R, 0, sizeof(R));
    memset(&W, 0, sizeof(W));
    int nr = 0, nw = 0;
    for (int i = 0; i < 8; ++i) {
        R.blk0[i] = nr;
        W.blk0[i] = nw;
        if (i >= n_views) continue;
        const long long n = (long long)shapes[i][0] * shapes[i][1] * shapes[i][2];
        if (n == 0) continue;
        BoxItem it;
        memset(&it, 0, sizeof(it));
        it.oz = (int)shapes[i][0]; it.oy = (int)shapes[i][1]; it.ox = (int)shapes[i][2];
        it.z0 = box0[i][0]; it.y0 = box0[i][1]; it.x0 = box0[i][2];
        int t[3];
        if (is_integer_crop(hviews[i], dtype, t)) {
            it.data = hviews[i].data; it.stride_z = hviews[i].stride_z; it.stride_y = hviews[i].stride_y;
            it.nz = hviews[i].nz; it.ny = hviews[i].ny; it.nx = hviews[i].nx;
            it.tz = t[0] + box0[i][0]; it.ty = t[1] + box0[i][1]; it.tx = t[2] + box0[i][2];
            it.out = res_out[i];
            R.v[i] = it;
            const long long ng = (long long)shapes[i][0] * shapes[i][1] * ((shapes[i][2] + 7) / 8);
            nr += (int)std::min<long long>((ng + 255) / 256, 256 * 16);
        } else {
            mvs_launch_resample(c, hviews[i], dtype, order, cval, res_out[i], shapes[i], box0[i]);
        }
        it.out = blend_out[i];
        W.v[i] = it;
        nw += (int)std::min<long long>((n + 255) / 256, 256 * 16);
    }
    R.blk0[8] = nr;
    W.blk0[8] = nw;
    if (nr > 0) {
        if (dtype == MVS_U8) hipLaunchKernelGGL(crop_int_batch_kernel<unsigned char>, dim3(nr), dim3(256), 0, c->stream, R, cval);
        else hipLaunchKernelGGL(crop_int_batch_kernel<unsigned short>, dim3(nr), dim3(256), 0, c->stream, R, cval);
    }
    if (nw > 0 && tr) {
        TrViews8 TV;
        memset(&TV, 0, sizeof(TV));
        for (int i = 0; i < n_views; ++i) TV.v[i] = tr[i];
        hipLaunchKernelGGL(blend_tr_batch_kernel, dim3(nw), dim3(256), 0, c->stream, TV, W);
    } else if (nw > 0) hipLaunchKernelGGL(blend_batch_kernel, dim3(nw), dim3(256), 0, c->stream, dviews, W);
}

// Index frame -> chunk frame for the kernels that evaluate the full affine map per voxel (generic fuse kernel, resample,
// blend): c = M (p + org) + off - ioff = M p + (off + M org - ioff).  (The translation fast path keeps the integer parts apart,
// see prepare_translation_view; here the last bit of a coordinate may depend on the chunk, as it does in the reference.)
void mvs_view_to_chunk_frame(DevView* d, const int64_t org[3], const int64_t ioff[3]) {
    if (!(org[0] | org[1] | org[2] | ioff[0] | ioff[1] | ioff[2])) return;
    const double o[3] = {(double)org[0], (double)org[1], (double)org[2]};
    for (int k = 0; k < 3; ++k) {
        d->off[k] = (((o[0] * d->m[3 * k] + o[1] * d->m[3 * k + 1]) + o[2] * d->m[3 * k + 2]) + d->off[k]) - (double)ioff[k];
        d->woff[k] = ((o[0] * d->wm[3 * k] + o[1] * d->wm[3 * k + 1]) + o[2] * d->wm[3 * k + 2]) + d->woff[k];
    }
}

// Chunk-index box [lo, hi] (inclusive; lo > hi: empty) outside of which view `d` is certainly out of bounds: exact for
// translations (the interval scipy's in-bounds test yields, as in the fast path), the bounding box of the slab's corners mapped
// into the chunk (+- 2 voxels) otherwise.
void mvs_view_chunk_box(const DevView& d, const int64_t shape[3], int lo[3], int hi[3]) {
    static const double I9[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
    bool ident = true;
    for (int k = 0; k < 9; ++k) ident = ident && d.m[k] == I9[k];
    const int n[3] = {d.nz, d.ny, d.nx};
    if (ident) {
        for (int k = 0; k < 3; ++k) {
            if (!(fabs(d.off[k]) < 1e9)) { lo[k] = 0; hi[k] = (int)shape[k] - 1; continue; }
            exact_valid_range(d.off[k], n[k], (int)shape[k], &lo[k], &hi[k]);
        }
        return;
    }
    const double* m = d.m;
    const double det = m[0] * (m[4] * m[8] - m[5] * m[7]) - m[1] * (m[3] * m[8] - m[5] * m[6]) + m[2] * (m[3] * m[7] - m[4] * m[6]);
    for (int k = 0; k < 3; ++k) { lo[k] = 0; hi[k] = (int)shape[k] - 1; }
    if (!(fabs(det) > 1e-12)) return;
    const double inv[9] = {(m[4] * m[8] - m[5] * m[7]) / det, (m[2] * m[7] - m[1] * m[8]) / det, (m[1] * m[5] - m[2] * m[4]) / det,
                           (m[5] * m[6] - m[3] * m[8]) / det, (m[0] * m[8] - m[2] * m[6]) / det, (m[2] * m[3] - m[0] * m[5]) / det,
                           (m[3] * m[7] - m[4] * m[6]) / det, (m[1] * m[6] - m[0] * m[7]) / det, (m[0] * m[4] - m[1] * m[3]) / det};
    double bl[3] = {1e300, 1e300, 1e300}, bh[3] = {-1e300, -1e300, -1e300};
    for (int q = 0; q < 8; ++q) {
        const double in[3] = {(q & 4) ? (double)(n[0] - 1) - d.off[0] : -d.off[0], (q & 2) ? (double)(n[1] - 1) - d.off[1] : -d.off[1],
                              (q & 1) ? (double)(n[2] - 1) - d.off[2] : -d.off[2]};
        for (int k = 0; k < 3; ++k) {
            const double o = inv[3 * k] * in[0] + inv[3 * k + 1] * in[1] + inv[3 * k + 2] * in[2];
            bl[k] = std::min(bl[k], o);
            bh[k] = std::max(bh[k], o);
        }
    }
    for (int k = 0; k < 3; ++k) {
        lo[k] = (int)std::max(0.0, std::floor(bl[k]) - 2.0);
        hi[k] = (int)std::min((double)shape[k] - 1.0, std::ceil(bh[k]) + 2.0);
    }
}
