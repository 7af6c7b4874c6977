// mvs_fuse_region.hip -- region-decomposed translation fast path of mvs_fuse_chunk (gfx950).
//
// For translation-only views (tile grids) the output chunk decomposes along every axis at the view
// borders into boxes ("regions") inside which the set of contributing views is CONSTANT and every
// voxel of the box is in bounds for each of them.  The host enumerates the regions (O(#views^3) tiny
// work), classifies per (region, view) whether the blend weight is 1 everywhere in the box (the weight
// profile is concave, so its minimum sits at a corner) and cuts every region into bricks of
// 4 planes x 32 rows x 128 voxels.  A wavefront takes one brick: lane (r, c) = (lane >> 4, lane & 15)
// owns row r of each 4-row group and 8 consecutive voxels at x = x0 + 8 c (one 16-byte store for u16).
// Compared with the column kernel (mvs_fuse.hip) this removes per-wave view culling, per-voxel validity
// masks and most weight evaluations:
//   * one view, weight 1            -> resample and store (no accumulators, no division)
//   * n views, all weights 1        -> plain average
//   * otherwise                     -> weighted accumulate; ramp weights only for the views that need them,
//                                      row-uniform nodes (G1, dG) evaluated once per plane by 32 lanes
// The unrolled NV = 1, 2, 4 instantiations keep all per-view constants in scalar registers.
#include "mvs_fuse_tr.h"

#include <algorithm>
#include <chrono>
#include <cstdlib>
#include <cstring>
#include <type_traits>
#include <vector>

namespace {

constexpr int kRB = 4;       // planes per brick
constexpr int kRG = 8;       // 4-row groups per brick
constexpr int kRV = 8;       // voxels per lane
constexpr int kMaxRV = 8;    // views per region handled here

struct Region {
    int z0, z1, y0, y1, x0, x1;   // chunk-index box, end exclusive
    int nviews;
    int allone_mask;              // bits 0-7: view ids[v] has blend weight 1 everywhere in the box; bit 15: every view is in
                                  // bounds with a strictly positive weight everywhere (plain weighted sums suffice);
                                  // bits 16-31: view ids[v] covers the box only partially (per-voxel bounds test)
    int ids[kMaxRV];
};
static_assert(sizeof(Region) == 64, "Region layout");

struct Item { int region_bx, by_bz; };   // region | bx << 16 ; by | bz << 16

struct RegionParams {
    const TrView* views;
    const Region* regions;
    const Item* items;
    int nitems;
    void* out;
    int oz, oy, ox;
    int tz, ty, tx;
};

typedef unsigned int u32x2_t __attribute__((ext_vector_type(2)));
typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));

// 8 consecutive elements (+ the 9th when NINE) of one row through a bounds-checked buffer load
template <typename T, bool NINE> struct Row8;
template <bool NINE> struct Row8<unsigned short, NINE> {
    static constexpr int NW = 5;
    static __device__ __forceinline__ void load(__amdgpu_buffer_rsrc_t r, int vo, unsigned int (&w)[9]) {
        const u32x4_t a = __builtin_amdgcn_raw_buffer_load_b128(r, vo, 0, 0);
        w[0] = a.x; w[1] = a.y; w[2] = a.z; w[3] = a.w;
        if (NINE) w[4] = __builtin_amdgcn_raw_buffer_load_b16(r, vo + 16, 0, 0);
    }
    static __device__ __forceinline__ void load_nt(__amdgpu_buffer_rsrc_t r, int vo, unsigned int (&w)[9]) {   // streaming hint
        const u32x4_t a = __builtin_amdgcn_raw_buffer_load_b128(r, vo, 0, 2);
        w[0] = a.x; w[1] = a.y; w[2] = a.z; w[3] = a.w;
    }
    static __device__ __forceinline__ void decode(const unsigned int (&w)[9], float (&v)[9]) {
#pragma unroll
        for (int k = 0; k < 4; ++k) { v[2 * k] = (float)(w[k] & 0xffffu); v[2 * k + 1] = (float)(w[k] >> 16); }
        v[8] = NINE ? (float)(w[4] & 0xffffu) : 0.f;
    }
};
template <bool NINE> struct Row8<unsigned char, NINE> {
    static constexpr int NW = 3;
    static __device__ __forceinline__ void load(__amdgpu_buffer_rsrc_t r, int vo, unsigned int (&w)[9]) {
        const u32x2_t a = __builtin_amdgcn_raw_buffer_load_b64(r, vo, 0, 0);
        w[0] = a.x; w[1] = a.y;
        if (NINE) w[2] = __builtin_amdgcn_raw_buffer_load_b8(r, vo + 8, 0, 0);
    }
    static __device__ __forceinline__ void decode(const unsigned int (&w)[9], float (&v)[9]) {
#pragma unroll
        for (int k = 0; k < 8; ++k) v[k] = (float)((w[k >> 2] >> (8 * (k & 3))) & 0xffu);
        v[8] = NINE ? (float)(w[2] & 0xffu) : 0.f;
    }
};
template <bool NINE> struct Row8<float, NINE> {
    static constexpr int NW = 9;
    static __device__ __forceinline__ void load(__amdgpu_buffer_rsrc_t r, int vo, unsigned int (&w)[9]) {
        const u32x4_t a = __builtin_amdgcn_raw_buffer_load_b128(r, vo, 0, 0);
        const u32x4_t b = __builtin_amdgcn_raw_buffer_load_b128(r, vo + 16, 0, 0);
        w[0] = a.x; w[1] = a.y; w[2] = a.z; w[3] = a.w; w[4] = b.x; w[5] = b.y; w[6] = b.z; w[7] = b.w;
        if (NINE) w[8] = __builtin_amdgcn_raw_buffer_load_b32(r, vo + 32, 0, 0);
    }
    static __device__ __forceinline__ void decode(const unsigned int (&w)[9], float (&v)[9]) {
#pragma unroll
        for (int k = 0; k < 8; ++k) v[k] = __uint_as_float(w[k]);
        v[8] = NINE ? __uint_as_float(w[8]) : 0.f;
    }
};

// Element-wise re-fetch of a 9-element window that touches the first / last bytes of the slab (a vector buffer load
// that is not entirely in range comes back as 0, see row_refetch in mvs_fuse.hip).  Rare (first / last rows of a
// slab only), so it is kept out of line and hands the values over through a wavefront-private LDS strip.
template <typename TIn>
__device__ __noinline__ void row8_refetch_lds(__amdgpu_buffer_rsrc_t r, int o, float* strip) {
    for (int j = 0; j < 9; ++j) {
        const int oj = o + j * (int)sizeof(TIn);
        float v;
        if (sizeof(TIn) == 2) v = (float)(unsigned short)__builtin_amdgcn_raw_buffer_load_b16(r, oj, 0, 0);
        else if (sizeof(TIn) == 1) v = (float)(unsigned char)__builtin_amdgcn_raw_buffer_load_b8(r, oj, 0, 0);
        else v = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(r, oj, 0, 0));
        strip[j * 64] = v;
    }
}
template <typename TIn>
__device__ __forceinline__ void row8_refetch(__amdgpu_buffer_rsrc_t r, int o, float (&v)[9], float* strip) {
    row8_refetch_lds<TIn>(r, o, strip);
#pragma unroll
    for (int j = 0; j < 9; ++j) v[j] = strip[j * 64];
}

__device__ __forceinline__ int rli(int x, int l) { return __builtin_amdgcn_readlane(x, l); }

// Lane layout of a brick: 2^lxb lanes side by side along x (8 voxels each), the rest of the 64 lanes stacked along y.
//   lxb = 4: 4 rows x 128 voxels per group, 8 groups per brick   (regular regions)
//   lxb = 1: 32 rows x 16 voxels, one group                      (thin ramp zones next to a view border)
//   lxb = 3: 8 rows x 64 voxels, 4 groups (overlap zones along x); lxb = 6: 1 row x 512 voxels, 32 groups (copy class)
struct LaneMap {
    int r, c, RG, NG, BXW;
    __device__ __forceinline__ LaneMap(int lane, int lxb)
        : r(lane >> lxb), c(lane & ((1 << lxb) - 1)), RG(64 >> lxb), NG(32 / (64 >> lxb)), BXW(kRV << lxb) {}
};

template <typename TOut> __device__ __forceinline__ TOut cast_r(float v);
template <> __device__ __forceinline__ float cast_r<float>(float v) { return v; }
template <> __device__ __forceinline__ unsigned short cast_r<unsigned short>(float v) { return (unsigned short)(int)v; }
template <> __device__ __forceinline__ unsigned char cast_r<unsigned char>(float v) { return (unsigned char)(int)v; }

template <typename T> struct Out8;
template <> struct Out8<unsigned short> { typedef unsigned short v8 __attribute__((ext_vector_type(8), aligned(2))); };
template <> struct Out8<unsigned char> { typedef unsigned char v8 __attribute__((ext_vector_type(8), aligned(1))); };
template <> struct Out8<float> { typedef float v8 __attribute__((ext_vector_type(8), aligned(4))); };

// Integer tiles copied into an output of the same type (copy class, integer offsets): the loaded dwords go out as they are
// -- no decode to float, no cast back, no packing.  A tail of nvalid < 8 elements leaves as 4 + 2 + 1 elements.
typedef unsigned int u32x4_a2 __attribute__((ext_vector_type(4), aligned(2)));
typedef unsigned int u32x2_a2 __attribute__((ext_vector_type(2), aligned(2)));
typedef unsigned int u32_a2 __attribute__((aligned(2)));
typedef unsigned int u32x2_a1 __attribute__((ext_vector_type(2), aligned(1)));
__device__ __forceinline__ void store8_bits(unsigned short* p, const unsigned int (&w)[9], int nvalid) {
    const unsigned int w0 = w[0], w1 = w[1], w2 = w[2], w3 = w[3];
    if (nvalid >= kRV) {
        u32x4_a2 o;
        o.x = w0; o.y = w1; o.z = w2; o.w = w3;
        __builtin_nontemporal_store(o, reinterpret_cast<u32x4_a2*>(p));      // the mosaic is written once and not read again
        return;
    }
    const bool has4 = (nvalid & 4) != 0;
    if (has4) {
        u32x2_a2 o;
        o.x = w0; o.y = w1;
        *reinterpret_cast<u32x2_a2*>(p) = o;
    }
    if (nvalid & 2) *reinterpret_cast<u32_a2*>(p + (has4 ? 4 : 0)) = has4 ? w2 : w0;
    if (nvalid & 1) {                  // the last element has the even index nvalid - 1: low half of dword (nvalid - 1) / 2
        const int d = (nvalid - 1) >> 1;
        const unsigned int v = (d == 0) ? w0 : (d == 1) ? w1 : (d == 2) ? w2 : w3;
        p[nvalid - 1] = (unsigned short)(v & 0xffffu);
    }
}
__device__ __forceinline__ void store8_bits(unsigned char* p, const unsigned int (&w)[9], int nvalid) {
    if (nvalid >= kRV) {
        u32x2_a1 o;
        o.x = w[0]; o.y = w[1];
        __builtin_nontemporal_store(o, reinterpret_cast<u32x2_a1*>(p));
        return;
    }
#pragma unroll
    for (int j = 0; j < kRV; ++j)
        if (j < nvalid) p[j] = (unsigned char)((w[j >> 2] >> (8 * (j & 3))) & 0xffu);
}
__device__ __forceinline__ void store8_bits(float*, const unsigned int (&)[9], int) {}     // (never taken: float tiles need nan_to_num)

template <typename TOut>
__device__ __forceinline__ void store8(TOut* p, const float (&q)[kRV], int nvalid) {
    if (nvalid >= kRV) {
        typename Out8<TOut>::v8 v;
#pragma unroll
        for (int j = 0; j < kRV; ++j) v[j] = cast_r<TOut>(q[j]);
        __builtin_nontemporal_store(v, reinterpret_cast<typename Out8<TOut>::v8*>(p));
    } else {
#pragma unroll
        for (int j = 0; j < kRV; ++j)
            if (j < nvalid) p[j] = cast_r<TOut>(q[j]);
    }
}

// uint16 output: pack first, then the same 1 (full) / <= 3 (tail) stores as the raw copy
template <>
__device__ __forceinline__ void store8<unsigned short>(unsigned short* p, const float (&q)[kRV], int nvalid) {
    unsigned int w[9];
#pragma unroll
    for (int k = 0; k < 4; ++k) w[k] = (unsigned int)(int)q[2 * k] | ((unsigned int)(int)q[2 * k + 1] << 16);
    store8_bits(p, w, nvalid);
}

// The per-view record fields live in lanes: lane (v * 10 + q) of set A (views 0..5) / set B (views 6, 7) holds
// float4 q of view v's TrView; field f (dword index) = component f & 3 of float4 f >> 2.
struct RecRegs { float4 a, b; };
template <int F>
__device__ __forceinline__ int rec_field(const RecRegs& R, int v) {
    const int q = F >> 2;
    float4 src = R.a;
    int lane = v * 10 + q;
    if (v >= 6) { src = R.b; lane = (v - 6) * 10 + q; }
    const float c = ((F & 3) == 0) ? src.x : ((F & 3) == 1) ? src.y : ((F & 3) == 2) ? src.z : src.w;
    return __builtin_amdgcn_readlane(__float_as_int(c), lane);
}
template <int F> __device__ __forceinline__ float rec_fieldf(const RecRegs& R, int v) { return __int_as_float(rec_field<F>(R, v)); }
// TrView dword indices
enum { F_LO_Z = 0, F_LO_Y, F_LO_X, F_HI_Z, F_HI_Y, F_HI_X, F_IO_Z, F_IO_Y, F_IO_X, F_WNZ, F_FW_Z, F_FW_Y, F_FW_X, F_XTAB,
       F_DATA_LO, F_DATA_HI, F_SPAN_LO, F_SPAN_HI, F_ST_Y, F_ST_Z, F_SILO_Z, F_SILO_Y, F_SILO_X, F_SIHI_Z, F_SIHI_Y, F_SIHI_X,
       F_SFLO_Z, F_SFLO_Y, F_SFLO_X, F_SFHI_Z, F_SFHI_Y, F_SFHI_X, F_SK_Z, F_SK_Y, F_SK_X, F_WS_Z, F_WS_Y, F_WS_X };

// Row-uniform nodes of view v at plane zc for brick row `row` (evaluated by lane `row`): G1, dG and whether the
// row lies inside the support along z and y.
__device__ __forceinline__ void row_nodes(const RecRegs& R, int v, int zc, int yc, float& G1, float& dG, bool& inside) {
    const float wsz = rec_fieldf<F_WS_Z>(R, v), wsy = rec_fieldf<F_WS_Y>(R, v), wsx = rec_fieldf<F_WS_X>(R, v);
    float az0 = INFINITY, az1 = INFINITY, fz = 0.f, uz = 0.f;
    const bool has_z = rec_field<F_WNZ>(R, v) > 1;
    if (has_z) {
        uz = fold_u(zc, rec_field<F_SILO_Z>(R, v), rec_fieldf<F_SFLO_Z>(R, v), rec_field<F_SIHI_Z>(R, v), rec_fieldf<F_SFHI_Z>(R, v),
                    rec_fieldf<F_SK_Z>(R, v));
        tent_cell(fmaxf(uz, 0.f), wsz, az0, az1, fz);
    }
    const float uy = fold_u(yc, rec_field<F_SILO_Y>(R, v), rec_fieldf<F_SFLO_Y>(R, v), rec_field<F_SIHI_Y>(R, v),
                            rec_fieldf<F_SFHI_Y>(R, v), rec_fieldf<F_SK_Y>(R, v));
    inside = (uz >= 0.f) && (uy >= 0.f);
    float ay0, ay1, fy;
    tent_cell(fmaxf(uy, 0.f), wsy, ay0, ay1, fy);
    const float uz_ = 1.f - fz, uy_ = 1.f - fy;
    const float m00 = fminf(az0, ay0), m01 = fminf(az0, ay1), m10 = fminf(az1, ay0), m11 = fminf(az1, ay1);
    const float a1 = wsx, a2 = 2.f * wsx;
    float g0 = fmaf(fminf(m01, a1), fy, fminf(m00, a1) * uy_);
    float g1 = fmaf(fminf(m11, a1), fy, fminf(m10, a1) * uy_);
    G1 = has_z ? fmaf(g1, fz, g0 * uz_) : g0;
    g0 = fmaf(fminf(m01, a2), fy, fminf(m00, a2) * uy_);
    g1 = fmaf(fminf(m11, a2), fy, fminf(m10, a2) * uy_);
    const float G2 = has_z ? fmaf(g1, fz, g0 * uz_) : g0;
    dG = G2 - G1;
}

// Buffer resource of view v's slab and the byte offset of the 8-voxel window of row (zc, yl) starting at chunk x = xl.
template <typename TIn>
__device__ __forceinline__ __amdgpu_buffer_rsrc_t view_window(const RecRegs& R, int v, int zc, int yl, int xl, int& vo, int& nbytes) {
    constexpr int ES = (int)sizeof(TIn);
    const int sy = rec_field<F_ST_Y>(R, v), sz = rec_field<F_ST_Z>(R, v);
    const unsigned long long dptr = ((unsigned long long)(unsigned)rec_field<F_DATA_HI>(R, v) << 32) | (unsigned)rec_field<F_DATA_LO>(R, v);
    nbytes = rec_field<F_SPAN_LO>(R, v) * ES;
    vo = (((zc + rec_field<F_IO_Z>(R, v)) * sz + (yl + rec_field<F_IO_Y>(R, v)) * sy) + (xl + rec_field<F_IO_X>(R, v))) * ES;
    return __builtin_amdgcn_make_buffer_rsrc((void*)dptr, 0, nbytes, 0x00020000);
}

// Values of view v at the lane's 8 voxels of row (zc, yl) starting at chunk x = xl: all stencil rows of the view are
// fetched back-to-back with bounds-checked buffer loads, then interpolated (x, then z, then y).
template <typename TIn>
__device__ __forceinline__ void fetch_val(const RecRegs& R, int v, int zc, int yl, int xl, float* strip, float (&val)[kRV]) {
    constexpr int ES = (int)sizeof(TIn);
    const float wz = rec_fieldf<F_FW_Z>(R, v), wy = rec_fieldf<F_FW_Y>(R, v), wx = rec_fieldf<F_FW_X>(R, v);
    const bool anyfrac = (wz > 0.f) || (wy > 0.f) || (wx > 0.f);
    const int sy = rec_field<F_ST_Y>(R, v), sz = rec_field<F_ST_Z>(R, v);
    const unsigned long long dptr = ((unsigned long long)(unsigned)rec_field<F_DATA_HI>(R, v) << 32) | (unsigned)rec_field<F_DATA_LO>(R, v);
    const int nbytes = rec_field<F_SPAN_LO>(R, v) * ES;
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)dptr, 0, nbytes, 0x00020000);
    const int vo = (((zc + rec_field<F_IO_Z>(R, v)) * sz + (yl + rec_field<F_IO_Y>(R, v)) * sy) + (xl + rec_field<F_IO_X>(R, v))) * ES;
    constexpr int WB = 9 * ES;

    if (anyfrac) {
        unsigned int w00[9], w01[9], w10[9], w11[9];
        Row8<TIn, true>::load(rsrc, vo, w00);
        Row8<TIn, true>::load(rsrc, vo + sy * ES, w01);
        Row8<TIn, true>::load(rsrc, vo + sz * ES, w10);
        Row8<TIn, true>::load(rsrc, vo + (sz + sy) * ES, w11);
        float e00[9], e01[9], e10[9], e11[9];
        Row8<TIn, true>::decode(w00, e00);
        Row8<TIn, true>::decode(w01, e01);
        Row8<TIn, true>::decode(w10, e10);
        Row8<TIn, true>::decode(w11, e11);
        const int olast = vo + (sz + sy) * ES;
        if (__any(vo < 0 || olast + WB > nbytes)) {   // windows touching the first / last bytes of the slab
            const int o1 = vo + sy * ES, o2 = vo + sz * ES;
            if (__any((vo < 0 && vo + WB > 0) || (vo < nbytes && vo + WB > nbytes))) row8_refetch<TIn>(rsrc, vo, e00, strip);
            if (__any((o1 < 0 && o1 + WB > 0) || (o1 < nbytes && o1 + WB > nbytes))) row8_refetch<TIn>(rsrc, o1, e01, strip);
            if (__any((o2 < 0 && o2 + WB > 0) || (o2 < nbytes && o2 + WB > nbytes))) row8_refetch<TIn>(rsrc, o2, e10, strip);
            if (__any((olast < 0 && olast + WB > 0) || (olast < nbytes && olast + WB > nbytes))) row8_refetch<TIn>(rsrc, olast, e11, strip);
        }
        const float ux = 1.f - wx, uy = 1.f - wy, uz = 1.f - wz;
#pragma unroll
        for (int j = 0; j < kRV; ++j) {
            const float a00 = fmaf(e00[j + 1], wx, e00[j] * ux), a01 = fmaf(e01[j + 1], wx, e01[j] * ux);
            const float a10 = fmaf(e10[j + 1], wx, e10[j] * ux), a11 = fmaf(e11[j + 1], wx, e11[j] * ux);
            const float s0 = fmaf(a10, wz, a00 * uz), s1 = fmaf(a11, wz, a01 * uz);
            val[j] = fmaf(s1, wy, s0 * uy);
        }
    } else {
        unsigned int w00[9];
        Row8<TIn, false>::load(rsrc, vo, w00);
        float e00[9];
        Row8<TIn, false>::decode(w00, e00);
        if (__any((vo < 0 && vo + WB > 0) || (vo < nbytes && vo + WB > nbytes))) row8_refetch<TIn>(rsrc, vo, e00, strip);
#pragma unroll
        for (int j = 0; j < kRV; ++j) val[j] = e00[j];
    }

}

// Lean path of a brick on which every view is in bounds, has blend weight 1 and an integer offset (the bulk of the
// overlap zones of a tile grid): the result is the plain mean of the views' values.  All row loads of a batch of row
// groups are issued back-to-back for all views before the first one is consumed.
template <typename TIn, typename TOut, int NV>
__device__ __forceinline__ void region_brick_avg(const RegionParams& P, const RecRegs& R, int nv, int z0b, int z1, int y0b, int y1,
                                                 int x0b, int x1, int lane, int lxb, float* strip) {
    constexpr int ES = (int)sizeof(TIn);
    constexpr int G = NV <= 2 ? 4 : NV <= 4 ? 2 : 1;   // row groups per batch
    constexpr int WB = 9 * ES;
    const LaneMap L(lane, lxb);
    const int r = L.r, c = L.c;
    const int xq = x0b + kRV * c;
    const int nvalid_x = min(max(x1 - xq, 0), kRV);
    const int xl = (nvalid_x > 0) ? xq : x0b;
    TOut* out = (TOut*)P.out;
    __amdgpu_buffer_rsrc_t rsrc[NV];
    int base[NV], sy[NV], sz[NV], nbytes[NV];
#pragma unroll
    for (int v = 0; v < NV; ++v) {
        const int vv = v < nv ? v : 0;
        sy[v] = rec_field<F_ST_Y>(R, vv) * ES; sz[v] = rec_field<F_ST_Z>(R, vv) * ES;
        const unsigned long long dptr = ((unsigned long long)(unsigned)rec_field<F_DATA_HI>(R, vv) << 32) | (unsigned)rec_field<F_DATA_LO>(R, vv);
        nbytes[v] = rec_field<F_SPAN_LO>(R, vv) * ES;
        rsrc[v] = __builtin_amdgcn_make_buffer_rsrc((void*)dptr, 0, nbytes[v], 0x00020000);
        base[v] = rec_field<F_IO_Z>(R, vv) * sz[v] + rec_field<F_IO_Y>(R, vv) * sy[v] + (xl + rec_field<F_IO_X>(R, vv)) * ES;
    }
    const float rn = __builtin_amdgcn_rcpf((float)nv);
    for (int p = 0; p < kRB; ++p) {
        const int zc = z0b + p;
        if (zc >= z1) break;
        for (int gb = 0; gb < L.NG; gb += G) {
            if (y0b + L.RG * gb >= y1) break;
            unsigned int raw[NV][G][9];
            int vo[NV][G];
#pragma unroll
            for (int v = 0; v < NV; ++v)
                if (v < nv) {
#pragma unroll
                    for (int g = 0; g < G; ++g) {
                        const int yl = min(y0b + L.RG * (gb + g) + r, y1 - 1);
                        vo[v][g] = base[v] + zc * sz[v] + yl * sy[v];
                        Row8<TIn, false>::load(rsrc[v], vo[v][g], raw[v][g]);
                    }
                }
#pragma unroll
            for (int g = 0; g < G; ++g) {
                const int yc = y0b + L.RG * (gb + g) + r;
                float num[kRV];
#pragma unroll
                for (int j = 0; j < kRV; ++j) num[j] = 0.f;
#pragma unroll
                for (int v = 0; v < NV; ++v)
                    if (v < nv) {
                        float e[9];
                        Row8<TIn, false>::decode(raw[v][g], e);
                        if (__any((vo[v][g] < 0 && vo[v][g] + WB > 0) || (vo[v][g] < nbytes[v] && vo[v][g] + WB > nbytes[v])))
                            row8_refetch<TIn>(rsrc[v], vo[v][g], e, strip);
#pragma unroll
                        for (int j = 0; j < kRV; ++j) num[j] += e[j];
                    }
                float q[kRV];
#pragma unroll
                for (int j = 0; j < kRV; ++j) q[j] = num[j] * rn;
                if (gb + g < L.NG && yc < y1 && nvalid_x > 0)      // (thin bricks have fewer row groups than one batch)
                    store8<TOut>(out + ((long long)(zc - P.tz) * P.oy + (yc - P.ty)) * (long long)P.ox + (xq - P.tx), q, nvalid_x);
            }
        }
    }
}

// One brick of a region with at most NV views: the view loop is unrolled, so per-view constants end up in scalar
// registers and the row nodes of ramp-weighted views in per-view vector registers.
template <typename TIn, typename TOut, int NV>
__device__ __forceinline__ void region_brick(const RegionParams& P, const RecRegs& R, int nviews, int masks, int z0b, int z1,
                                             int y0b, int y1, int x0b, int x1, int lane, int lxb, float* strip, float* nodes) {
    constexpr bool ISF = std::is_floating_point<TIn>::value;
    const int nv = nviews;   // <= NV
    const LaneMap L(lane, lxb);
    const int r = L.r, c = L.c;
    const int xq = x0b + kRV * c;
    const int nvalid_x = min(max(x1 - xq, 0), kRV);
    const int xl = (nvalid_x > 0) ? xq : x0b;      // lanes beyond the region read a valid window, nothing is stored
    int allone_mask = masks & 0xff;
    int partial_mask = (masks >> 16) & 0xff;
    const bool allpos = (masks >> 15) & 1;   // host: every view covers the box with weight > 0 everywhere
    // "Partial" is a property of the BOX: after the clustering of nearby view borders (the tiles of a registered grid row differ by
    // a few pixels) a view misses a sliver at one end of the box.  A brick away from that sliver is covered completely: its
    // voxels need no bounds test, and it qualifies for the unit / lean paths below like a brick of a fully covered box.
    if (partial_mask) {
        const int ze = min(z0b + kRB, z1) - 1, ye = min(y0b + 32, y1) - 1, xe = min(x0b + L.BXW, x1) - 1;
#pragma unroll
        for (int v = 0; v < NV; ++v) {
            if (v < nv && ((partial_mask >> v) & 1)) {
                const bool covered = rec_field<F_LO_Z>(R, v) <= z0b && rec_field<F_HI_Z>(R, v) >= ze && rec_field<F_LO_Y>(R, v) <= y0b &&
                                     rec_field<F_HI_Y>(R, v) >= ye && rec_field<F_LO_X>(R, v) <= x0b && rec_field<F_HI_X>(R, v) >= xe;
                if (covered) partial_mask &= ~(1 << v);
            }
        }
    }
    // The host classified whole regions; the set {weight == 1} is curved (near an edge of a view the profile is a
    // product of the axis coordinates), so most of a region can be "unit" without the region being so.  Refine per
    // brick: the profile is concave along every axis line, hence its minimum over the brick sits at one of the 8
    // brick corners -- lane k & 7 evaluates corner k, and a view whose 8 corner values are all >= 1 needs no weights.
    if ((allone_mask & ((1 << nv) - 1)) != ((1 << nv) - 1)) {
        const int k = lane & 7;
        const int zk = (k & 4) ? min(z0b + kRB, z1) - 1 : z0b;
        const int yk = (k & 2) ? min(y0b + 32, y1) - 1 : y0b;
        const int xk = (k & 1) ? min(x0b + L.BXW, x1) - 1 : x0b;
#pragma unroll
        for (int v = 0; v < NV; ++v) {
            if (v < nv && !((allone_mask >> v) & 1) && !((partial_mask >> v) & 1)) {
                float G1, dG;
                bool inside;
                row_nodes(R, v, zk, yk, G1, dG, inside);
                const float u = fold_u(xk, rec_field<F_SILO_X>(R, v), rec_fieldf<F_SFLO_X>(R, v), rec_field<F_SIHI_X>(R, v),
                                       rec_fieldf<F_SFHI_X>(R, v), rec_fieldf<F_SK_X>(R, v));
                const float W = (u >= 0.f && inside) ? row_profile(u, G1, dG) : 0.f;
                // a box seen by ONE view only needs a weight that does not round to 0 (w / w == 1): see the host's test
                const float need = (nv == 1) ? 3e-4f : 1.f;
                if (!__any(!(W >= need))) allone_mask |= 1 << v;
            }
        }
    }
    const bool all_unit = (allone_mask & ((1 << nv) - 1)) == ((1 << nv) - 1);
    // "Anchored" brick: at least one view covers it completely with blend weight exactly 1.  Then the weight sum is >= 1 at every
    // voxel (no 0 / 0, nothing to check for finiteness with integer tiles), and a voxel that only this view reaches comes out as
    // v * 1 / 1 = v exactly -- the two cases the single-contributor bookkeeping (last / wlast) and the finite test exist for cannot
    // occur, whatever the other views do (ramp weights, weights that round to 0, partial coverage after border clustering): plain
    // weighted sums are exact enough everywhere in the brick.  This is every ramp brick of a face / edge overlap of a tile grid
    // (one neighbour is always deep inside its own support there); only bricks on the rim of the mosaic and in the curved
    // corner zones stay on the general path.
    const bool anchored = !ISF && (allone_mask & ~partial_mask & ((1 << nv) - 1)) != 0;
    const bool plain = !ISF && (allpos || anchored);
    TOut* out = (TOut*)P.out;
    bool allint = true;   // every view has an integer offset: one tap row per view
#pragma unroll
    for (int v = 0; v < NV; ++v)
        if (v < nv) allint = allint && !(rec_fieldf<F_FW_Z>(R, v) > 0.f || rec_fieldf<F_FW_Y>(R, v) > 0.f || rec_fieldf<F_FW_X>(R, v) > 0.f);
    if (all_unit && !partial_mask && !ISF && allint) {
        region_brick_avg<TIn, TOut, NV>(P, R, nv, z0b, z1, y0b, y1, x0b, x1, lane, lxb, strip);
        return;
    }

    // ---- row nodes (G1, dG, inside) of the views with ramp weights for the 32 rows x 4 planes of the brick: lane
    // (row, plane pair) evaluates them once into a wavefront-private LDS table [view][plane][row][3] ----
    if (!all_unit) {
        const int row = lane & 31, pp = (lane >> 5) * 2;
#pragma unroll
        for (int v = 0; v < NV; ++v) {
            if (v < nv && !((allone_mask >> v) & 1)) {
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    float G1, dG;
                    bool inside;
                    row_nodes(R, v, min(z0b + pp + q, z1 - 1), min(y0b + row, y1 - 1), G1, dG, inside);
                    float* nd = nodes + ((v * kRB + pp + q) * 32 + row) * 3;
                    nd[0] = G1; nd[1] = dG; nd[2] = inside ? 1.f : 0.f;
                }
            }
        }
        __builtin_amdgcn_wave_barrier();
    }

    // Row groups outside, planes inside: with fractional offsets consecutive planes share a tap plane, and this order
    // re-reads it right away (L1 / L2 hit) instead of one whole brick plane later.
    for (int g = 0; g < L.NG; ++g) {
        const int yg = y0b + L.RG * g;
        if (yg >= y1) break;
        const int yc = yg + r;
        const bool row_ok = yc < y1;
        const int yl = row_ok ? yc : y1 - 1;
        for (int p = 0; p < kRB; ++p) {
            const int zc = z0b + p;
            if (zc >= z1) break;

            float num[kRV], den[kRV], last[kRV], wlast[kRV];
#pragma unroll
            for (int j = 0; j < kRV; ++j) { num[j] = 0.f; den[j] = 0.f; last[j] = 0.f; wlast[j] = 0.f; }

            // integer offsets: the tap rows of ALL views are requested before the first one is consumed (one memory
            // round trip per row group instead of one per view -- these kernels run at 2-3 wavefronts per SIMD)
            // (measured: pays for NV = 4, costs 9 % for NV = 2, whose two loads already overlap well enough)
            constexpr bool kBatch = NV >= 4;
            unsigned int raw[NV][9];
            if (kBatch && allint) {
#pragma unroll
                for (int v = 0; v < NV; ++v)
                    if (v < nv) {
                        int vo, nb;
                        const __amdgpu_buffer_rsrc_t rs = view_window<TIn>(R, v, zc, yl, xl, vo, nb);
                        Row8<TIn, false>::load(rs, vo, raw[v]);
                    }
            }
#pragma unroll
            for (int v = 0; v < NV; ++v) {
                if (v >= nv) break;
                float val[kRV];
                if (kBatch && allint) {
                    constexpr int WB = 9 * (int)sizeof(TIn);
                    int vo, nb;
                    const __amdgpu_buffer_rsrc_t rs = view_window<TIn>(R, v, zc, yl, xl, vo, nb);
                    float e[9];
                    Row8<TIn, false>::decode(raw[v], e);
                    if (__any((vo < 0 && vo + WB > 0) || (vo < nb && vo + WB > nb))) row8_refetch<TIn>(rs, vo, e, strip);
#pragma unroll
                    for (int j = 0; j < kRV; ++j) val[j] = e[j];
                } else {
                    fetch_val<TIn>(R, v, zc, yl, xl, strip, val);
                }

                // views that cover the box only partly: per-voxel in-bounds test against the view's valid box
                const bool partial = (partial_mask >> v) & 1;
                bool inb[kRV];
#pragma unroll
                for (int j = 0; j < kRV; ++j) inb[j] = true;
                if (partial) {
                    const bool zy_ok = (zc >= rec_field<F_LO_Z>(R, v)) && (zc <= rec_field<F_HI_Z>(R, v)) &&
                                       (yc >= rec_field<F_LO_Y>(R, v)) && (yc <= rec_field<F_HI_Y>(R, v));
                    const int jlo = rec_field<F_LO_X>(R, v) - xq, jw = rec_field<F_HI_X>(R, v) - rec_field<F_LO_X>(R, v);
#pragma unroll
                    for (int j = 0; j < kRV; ++j) inb[j] = zy_ok && ((unsigned)(j - jlo) <= (unsigned)jw);
                }
                bool unit = (allone_mask >> v) & 1;
                float w[kRV];
                if (!unit) {
                    const float* nd = nodes + ((v * kRB + p) * 32 + L.RG * g + r) * 3;
                    const float G1 = nd[0], dG = nd[1];
                    const int inside = nd[2] != 0.f;
                    const float kx = rec_fieldf<F_SK_X>(R, v);
                    // distances to the two ends of the support: (integer part + j) - fraction, ONE rounding per voxel exactly as
                    // fold_u does it -- (base - fraction) + j would round twice and make the last bit of a weight depend on
                    // where the lane's 8-voxel group starts, i.e. on the box decomposition of the launch block
                    const float dlb = (float)(xl - rec_field<F_SILO_X>(R, v)), flo_ = rec_fieldf<F_SFLO_X>(R, v);
                    const float dhb = (float)(rec_field<F_SIHI_X>(R, v) - xl), fhi_ = rec_fieldf<F_SFHI_X>(R, v);
                    const float dl0 = dlb - flo_, dh0 = dhb - fhi_;
                    // The profile is concave along x, so over the lane's 8 voxels its minimum sits at voxel 0 or 7:
                    // two evaluations tell whether the whole segment has weight 1.
                    const float u0 = fminf(dl0, dh0) * kx, u7 = fminf((dlb + 7.f) - flo_, (dhb - 7.f) - fhi_) * kx;
                    const float W0 = (u0 >= 0.f && inside) ? row_profile(u0, G1, dG) : 0.f;
                    const float W7 = (u7 >= 0.f && inside) ? row_profile(u7, G1, dG) : 0.f;
                    const bool lane_unit = fminf(W0, W7) >= ((nv == 1) ? 3e-4f : 1.f);   // one view: any weight > 0 yields the value
                    // Beyond the first support cell (u >= 1) a row whose nodes do not grow any more (dG == 0: the row lies in
                    // the ramp of ANOTHER axis) has the same profile value G1 at all 8 voxels: one ramp evaluation per lane.
                    const bool lane_flat = (fminf(u0, u7) >= 1.f) && (dG == 0.f);
                    if (!__any(!lane_unit)) unit = true;
                    else if (!__any(!(lane_unit || lane_flat))) {
                        const float w0 = blend_ramp_nb(W0);
#pragma unroll
                        for (int j = 0; j < kRV; ++j) w[j] = w0;
                    } else {
#pragma unroll
                        for (int j = 0; j < kRV; ++j) {
                            const float u = fminf((dlb + (float)j) - flo_, (dhb - (float)j) - fhi_) * kx;
                            const float W = (u >= 0.f && inside) ? row_profile(u, G1, dG) : 0.f;
                            w[j] = blend_ramp_nb(W);
                        }
                    }
                }
                if (unit) {
#pragma unroll
                    for (int j = 0; j < kRV; ++j) {
                        const bool ok = (ISF ? (val[j] == val[j]) : true) && inb[j];
                        num[j] += ok ? val[j] : 0.f;
                        den[j] += ok ? 1.f : 0.f;
                    }
                } else if (plain) {
                    // every view of the region is in bounds with a strictly positive weight everywhere, or the brick is anchored
                    // by a full unit view: plain weighted sums (a partial view weighs 0 outside its valid box)
#pragma unroll
                    for (int j = 0; j < kRV; ++j) {
                        const float we = (partial && !inb[j]) ? 0.f : w[j];
                        num[j] = fmaf(we, val[j], num[j]);
                        den[j] += we;
                    }
                } else {
#pragma unroll
                    for (int j = 0; j < kRV; ++j) {
                        const bool ok = (ISF ? (val[j] == val[j]) : true) && inb[j];
                        const float we = ok ? w[j] : 0.f;
                        const bool pos = we > 0.f;
                        const float ve = pos ? val[j] : 0.f;
                        num[j] = fmaf(we, ve, num[j]);
                        den[j] += we;
                        const bool ramp = pos && (we < 1.f);
                        const int pm = ramp ? -1 : 0;   // bit-select (see mvs_fuse.hip)
                        last[j] = __int_as_float((__float_as_int(val[j]) & pm) | (__float_as_int(last[j]) & ~pm));
                        wlast[j] = __int_as_float((__float_as_int(we) & pm) | (__float_as_int(wlast[j]) & ~pm));
                    }
                }
            }

            // ---- epilogue of this row group ----
            float q[kRV];
#pragma unroll
            for (int j = 0; j < kRV; ++j) {
                float o;
                if (nv == 1 && all_unit && !partial_mask && !ISF) o = num[j];      // a single full view with weight 1
                else if (plain) o = num[j] * __builtin_amdgcn_rcpf(den[j]);        // den > 0, finite integer data: nothing to check
                else {
                    o = num[j] * __builtin_amdgcn_rcpf(den[j]);
                    o = (den[j] == wlast[j]) ? last[j] : o;                        // single ramp contributor: exact value
                    if (!(fabsf(o) <= 3.4028234e38f)) o = 0.f;
                }
                q[j] = o;
            }
            if (row_ok && nvalid_x > 0) {
                const long long oi = ((long long)(zc - P.tz) * P.oy + (yc - P.ty)) * (long long)P.ox + (xq - P.tx);
                store8<TOut>(out + oi, q, nvalid_x);
            }
        }
    }
}

// Regions with more than 4 views (3D corners, clustered borders): not unrolled, no per-view registers; the blend
// profile of non-unit views is evaluated per voxel from the record in memory.  Rare, so compact beats fast here.
template <typename TIn, typename TOut>
__device__ __forceinline__ void region_brick_generic(const RegionParams& P, const RecRegs& R, int rw, int nviews, int masks, int z0b, int z1,
                                                  int y0b, int y1, int x0b, int x1, int lane, int lxb, float* strip) {
    constexpr bool ISF = std::is_floating_point<TIn>::value;
    const LaneMap L(lane, lxb);
    const int r = L.r, c = L.c;
    const int xq = x0b + kRV * c;
    const int nvalid_x = min(max(x1 - xq, 0), kRV);
    const int xl = (nvalid_x > 0) ? xq : x0b;
    const int allone_mask = masks & 0xff, partial_mask = (masks >> 16) & 0xff;
    TOut* out = (TOut*)P.out;
    for (int p = 0; p < kRB; ++p) {
        const int zc = z0b + p;
        if (zc >= z1) break;
        for (int g = 0; g < L.NG; ++g) {
            const int yg = y0b + L.RG * g;
            if (yg >= y1) break;
            const int yc = yg + r;
            const bool row_ok = yc < y1;
            const int yl = row_ok ? yc : y1 - 1;
            float num[kRV], den[kRV], last[kRV], wlast[kRV];
#pragma unroll
            for (int j = 0; j < kRV; ++j) { num[j] = 0.f; den[j] = 0.f; last[j] = 0.f; wlast[j] = 0.f; }
#pragma nounroll
            for (int v = 0; v < nviews; ++v) {
                float val[kRV];
                fetch_val<TIn>(R, v, zc, yl, xl, strip, val);
                const TrView& V = P.views[rli(rw, 8 + v)];
                const bool partial = (partial_mask >> v) & 1, unit = (allone_mask >> v) & 1;
                const bool zy_ok = !partial || (zc >= V.lo[0] && zc <= V.hi[0] && yc >= V.lo[1] && yc <= V.hi[1]);
#pragma unroll
                for (int j = 0; j < kRV; ++j) {
                    const int x = xq + j;
                    bool ok = zy_ok && (!partial || (x >= V.lo[2] && x <= V.hi[2]));
                    if (ISF) ok = ok && (val[j] == val[j]);
                    float we = 1.f;
                    if (!unit) we = blend_ramp_nb(tr_weight_profile(V, zc, yl, xl + j));
                    we = ok ? we : 0.f;
                    const bool pos = we > 0.f;
                    num[j] = fmaf(we, pos ? val[j] : 0.f, num[j]);
                    den[j] += we;
                    const int pm = (pos && we < 1.f) ? -1 : 0;
                    last[j] = __int_as_float((__float_as_int(val[j]) & pm) | (__float_as_int(last[j]) & ~pm));
                    wlast[j] = __int_as_float((__float_as_int(we) & pm) | (__float_as_int(wlast[j]) & ~pm));
                }
            }
            float q[kRV];
#pragma unroll
            for (int j = 0; j < kRV; ++j) {
                float o = num[j] * __builtin_amdgcn_rcpf(den[j]);
                o = (den[j] == wlast[j]) ? last[j] : o;
                if (!(fabsf(o) <= 3.4028234e38f)) o = 0.f;
                q[j] = o;
            }
            if (row_ok && nvalid_x > 0)
                store8<TOut>(out + ((long long)(zc - P.tz) * P.oy + (yc - P.ty)) * (long long)P.ox + (xq - P.tx), q, nvalid_x);
        }
    }
}

// Class A bricks: ONE view that covers the whole box with blend weight 1 (the interior of every tile: more than half
// of a 20 %-overlap mosaic).  The result is the resampled value itself -- no accumulators, no weights, no division.
// All constants are hoisted into scalar registers, the 8 row-group loads of a plane are issued back-to-back before
// the first one is consumed (integer offsets), so a wavefront keeps 8 KiB in flight.
template <typename TIn, typename TOut>
__device__ __forceinline__ void copy_brick_item(const RegionParams& P, const Item it, const int lane, float* strip_w) {
    constexpr int ES = (int)sizeof(TIn);
    const int rid = it.region_bx & 0xffff, bx = (unsigned)it.region_bx >> 16, by = it.by_bz & 0xffff, bz = (unsigned)it.by_bz >> 16;
    const int rw = reinterpret_cast<const int*>(P.regions + rid)[lane & 15];
    const int z1 = rli(rw, 1), y1 = rli(rw, 3), x1 = rli(rw, 5);
    const int lxb = (rli(rw, 6) >> 8) & 7;
    const LaneMap L(lane, lxb);
    const int z0b = rli(rw, 0) + kRB * bz, y0b = rli(rw, 2) + 32 * by, x0b = rli(rw, 4) + L.BXW * bx;
    float* strip = strip_w + lane;
    // the view's record: lane q < 10 holds float4 q
    RecRegs R;
    R.a = make_float4(0.f, 0.f, 0.f, 0.f);
    R.b = R.a;
    {
        const int id = rli(rw, 8);
        if (lane < 10) R.a = reinterpret_cast<const float4*>(P.views + id)[lane];
    }
    const float wz = rec_fieldf<F_FW_Z>(R, 0), wy = rec_fieldf<F_FW_Y>(R, 0), wx = rec_fieldf<F_FW_X>(R, 0);
    const bool anyfrac = (wz > 0.f) || (wy > 0.f) || (wx > 0.f);
    const int sy = rec_field<F_ST_Y>(R, 0), sz = rec_field<F_ST_Z>(R, 0);
    const unsigned long long dptr = ((unsigned long long)(unsigned)rec_field<F_DATA_HI>(R, 0) << 32) | (unsigned)rec_field<F_DATA_LO>(R, 0);
    const int nbytes = rec_field<F_SPAN_LO>(R, 0) * ES;
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)dptr, 0, nbytes, 0x00020000);
    const int ioz = rec_field<F_IO_Z>(R, 0), ioy = rec_field<F_IO_Y>(R, 0), iox = rec_field<F_IO_X>(R, 0);

    const int r = L.r, c = L.c;
    const int xq = x0b + kRV * c;
    const int nvalid_x = min(max(x1 - xq, 0), kRV);
    const int xl = (nvalid_x > 0) ? xq : x0b;
    constexpr int WB = 9 * ES;
    TOut* out = (TOut*)P.out;

    if (anyfrac) {
        // Fractional offset: an output plane needs the tap planes z and z + 1, so the 4 planes of the brick share 3 of
        // their 5 tap planes.  Per row group the 5 x 2 tap rows (y, y + 1) are requested once, back-to-back, interpolated
        // along x once, and every plane pair is then combined in registers (2.5 instead of 4 row loads per output row).
        const float ux = 1.f - wx, uy = 1.f - wy, uz = 1.f - wz;
        const int np = min(z1 - z0b, kRB);                     // planes of this brick
        for (int g = 0; g < L.NG; ++g) {
            if (y0b + L.RG * g >= y1) break;
            const int yc = y0b + L.RG * g + r;
            const int yl = min(yc, y1 - 1);
            unsigned int raw[kRB + 1][2][9];
            int vo[kRB + 1];
#pragma unroll
            for (int q = 0; q <= kRB; ++q) {
                const int zq = z0b + min(q, np);               // planes past the brick repeat the last needed one
                vo[q] = ((zq + ioz) * sz + (yl + ioy) * sy + (xl + iox)) * ES;
                Row8<TIn, true>::load(rsrc, vo[q], raw[q][0]);
                Row8<TIn, true>::load(rsrc, vo[q] + sy * ES, raw[q][1]);
            }
            float lo0[kRV], lo1[kRV];
#pragma unroll
            for (int q = 0; q <= kRB; ++q) {
                if (q > np) break;
                float e0[9], e1[9];
                Row8<TIn, true>::decode(raw[q][0], e0);
                Row8<TIn, true>::decode(raw[q][1], e1);
                const int o0 = vo[q], o1 = vo[q] + sy * ES;
                if (__any(o0 < 0 || o1 + WB > nbytes)) {
                    if (__any((o0 < 0 && o0 + WB > 0) || (o0 < nbytes && o0 + WB > nbytes))) row8_refetch<TIn>(rsrc, o0, e0, strip);
                    if (__any((o1 < 0 && o1 + WB > 0) || (o1 < nbytes && o1 + WB > nbytes))) row8_refetch<TIn>(rsrc, o1, e1, strip);
                }
                float hi0[kRV], hi1[kRV];
#pragma unroll
                for (int j = 0; j < kRV; ++j) {
                    hi0[j] = fmaf(e0[j + 1], wx, e0[j] * ux);
                    hi1[j] = fmaf(e1[j + 1], wx, e1[j] * ux);
                }
                if (q > 0) {
                    const int zc = z0b + q - 1;
                    float o[kRV];
#pragma unroll
                    for (int j = 0; j < kRV; ++j) {
                        const float s0 = fmaf(hi0[j], wz, lo0[j] * uz), s1 = fmaf(hi1[j], wz, lo1[j] * uz);
                        const float vv = fmaf(s1, wy, s0 * uy);
                        o[j] = (vv == vv) ? vv : 0.f;
                    }
                    if (yc < y1 && nvalid_x > 0)
                        store8<TOut>(out + ((long long)(zc - P.tz) * P.oy + (yc - P.ty)) * (long long)P.ox + (xq - P.tx), o, nvalid_x);
                }
#pragma unroll
                for (int j = 0; j < kRV; ++j) { lo0[j] = hi0[j]; lo1[j] = hi1[j]; }
            }
        }
        return;
    }

    for (int p = 0; p < kRB; ++p) {
        const int zc = z0b + p;
        if (zc >= z1) break;
        const int vo_p = ((zc + ioz) * sz + ioy * sy + (xl + iox)) * ES;   // + row * sy * ES
        if (!anyfrac && lxb >= 4) {
          for (int gb = 0; gb < L.NG; gb += kRG) {
            if (y0b + L.RG * gb >= y1) break;
            unsigned int raw[kRG][9];
            int vo[kRG];
#pragma unroll
            for (int g = 0; g < kRG; ++g) {
                const int yl = min(y0b + L.RG * (gb + g) + r, y1 - 1);
                vo[g] = vo_p + yl * sy * ES;
                if (std::is_same<TIn, unsigned short>::value) Row8<unsigned short, false>::load_nt(rsrc, vo[g], raw[g]);
                else Row8<TIn, false>::load(rsrc, vo[g], raw[g]);
            }
            int ends = 0;
#pragma unroll
            for (int g = 0; g < kRG; ++g) ends |= ((int)(vo[g] < 0) & (int)(vo[g] + WB > 0)) | ((int)(vo[g] < nbytes) & (int)(vo[g] + WB > nbytes));
            if (std::is_same<TIn, TOut>::value && !std::is_floating_point<TIn>::value && !__any(ends != 0)) {
                // same integer type in and out: the loaded dwords are the result
#pragma unroll
                for (int g = 0; g < kRG; ++g) {
                    const int yc = y0b + L.RG * (gb + g) + r;
                    if (yc < y1 && nvalid_x > 0)
                        store8_bits(out + ((long long)(zc - P.tz) * P.oy + (yc - P.ty)) * (long long)P.ox + (xq - P.tx), raw[g], nvalid_x);
                }
            } else {
#pragma unroll
                for (int g = 0; g < kRG; ++g) {
                    const int yc = y0b + L.RG * (gb + g) + r;
                    float e[9];
                    Row8<TIn, false>::decode(raw[g], e);
                    if (__any((vo[g] < 0 && vo[g] + WB > 0) || (vo[g] < nbytes && vo[g] + WB > nbytes))) row8_refetch<TIn>(rsrc, vo[g], e, strip);
                    float q[kRV];
#pragma unroll
                    for (int j = 0; j < kRV; ++j) q[j] = (e[j] == e[j]) ? e[j] : 0.f;   // nan_to_num (float tiles)
                    if (yc < y1 && nvalid_x > 0)
                        store8<TOut>(out + ((long long)(zc - P.tz) * P.oy + (yc - P.ty)) * (long long)P.ox + (xq - P.tx), q, nvalid_x);
                }
            }
          }
        } else {
            const float ux = 1.f - wx, uy = 1.f - wy, uz = 1.f - wz;
#pragma unroll 2
            for (int g = 0; g < L.NG; ++g) {
                const int yc = y0b + L.RG * g + r;
                if (y0b + L.RG * g >= y1) break;
                const int yl = min(yc, y1 - 1);
                const int v0 = vo_p + yl * sy * ES;
                unsigned int w00[9], w01[9], w10[9], w11[9];
                Row8<TIn, true>::load(rsrc, v0, w00);
                Row8<TIn, true>::load(rsrc, v0 + sy * ES, w01);
                Row8<TIn, true>::load(rsrc, v0 + sz * ES, w10);
                Row8<TIn, true>::load(rsrc, v0 + (sz + sy) * ES, w11);
                float e00[9], e01[9], e10[9], e11[9];
                Row8<TIn, true>::decode(w00, e00);
                Row8<TIn, true>::decode(w01, e01);
                Row8<TIn, true>::decode(w10, e10);
                Row8<TIn, true>::decode(w11, e11);
                const int olast = v0 + (sz + sy) * ES;
                if (__any(v0 < 0 || olast + WB > nbytes)) {
                    const int o1 = v0 + sy * ES, o2 = v0 + sz * ES;
                    if (__any((v0 < 0 && v0 + WB > 0) || (v0 < nbytes && v0 + WB > nbytes))) row8_refetch<TIn>(rsrc, v0, e00, strip);
                    if (__any((o1 < 0 && o1 + WB > 0) || (o1 < nbytes && o1 + WB > nbytes))) row8_refetch<TIn>(rsrc, o1, e01, strip);
                    if (__any((o2 < 0 && o2 + WB > 0) || (o2 < nbytes && o2 + WB > nbytes))) row8_refetch<TIn>(rsrc, o2, e10, strip);
                    if (__any((olast < 0 && olast + WB > 0) || (olast < nbytes && olast + WB > nbytes))) row8_refetch<TIn>(rsrc, olast, e11, strip);
                }
                float q[kRV];
#pragma unroll
                for (int j = 0; j < kRV; ++j) {
                    const float a00 = fmaf(e00[j + 1], wx, e00[j] * ux), a01 = fmaf(e01[j + 1], wx, e01[j] * ux);
                    const float a10 = fmaf(e10[j + 1], wx, e10[j] * ux), a11 = fmaf(e11[j + 1], wx, e11[j] * ux);
                    const float s0 = fmaf(a10, wz, a00 * uz), s1 = fmaf(a11, wz, a01 * uz);
                    const float vv = fmaf(s1, wy, s0 * uy);
                    q[j] = (vv == vv) ? vv : 0.f;
                }
                if (yc < y1 && nvalid_x > 0)
                    store8<TOut>(out + ((long long)(zc - P.tz) * P.oy + (yc - P.ty)) * (long long)P.ox + (xq - P.tx), q, nvalid_x);
            }
        }
    }
}

template <typename TIn, typename TOut>
__global__ __launch_bounds__(256) void copy_region_kernel(RegionParams P, int item0, int nitems) {
    __shared__ float s_strip[4][9 * 64];
    const int lane = threadIdx.x & 63;
    // XCD-aware order: workgroups are dealt round-robin to the 8 XCDs (each with its own L2), so XCD k takes the k-th
    // contiguous eighth of the brick list -- neighbouring bricks, which share the cache lines at their edges and (with
    // fractional offsets) whole planes, then meet in ONE L2 instead of being fetched once per XCD.
    const int wg = (blockIdx.x & 7) * (gridDim.x >> 3) + (blockIdx.x >> 3);
    const int li = wg * 4 + (threadIdx.x >> 6);
    if (li >= nitems) return;
    copy_brick_item<TIn, TOut>(P, P.items[item0 + li], lane, s_strip[threadIdx.x >> 6]);
}

// One kernel per view-count class (NVC = 1, 2, 4: regions with <= NVC views, unrolled; NVC = 0: any count), so every
// class gets its own register allocation and a small instruction footprint.  Items of a class are contiguous.
template <typename TIn, typename TOut, int NVC>
__device__ __forceinline__ void fuse_brick_item(const RegionParams& P, const Item it, const int lane, float* strip_w, float* nodes_w) {
    const int rid = it.region_bx & 0xffff, bx = (unsigned)it.region_bx >> 16, by = it.by_bz & 0xffff, bz = (unsigned)it.by_bz >> 16;
    // region descriptor: lane l < 16 loads dword l, fields are pulled out with readlane
    const int rw = reinterpret_cast<const int*>(P.regions + rid)[lane & 15];
    const int z1 = rli(rw, 1), y1 = rli(rw, 3), x1 = rli(rw, 5);
    const int nviews = rli(rw, 6) & 0xff, lxb = (rli(rw, 6) >> 8) & 7, masks = rli(rw, 7);
    const LaneMap L(lane, lxb);
    const int z0b = rli(rw, 0) + kRB * bz, y0b = rli(rw, 2) + 32 * by, x0b = rli(rw, 4) + L.BXW * bx;
    float* strip = strip_w + lane;

    if (nviews == 0) {   // nothing contributes: zeros (np.nansum of nothing, nan_to_num)
        const int r = L.r, c = L.c, xq = x0b + kRV * c;
        const int nvx = min(max(x1 - xq, 0), kRV);
        float q[kRV] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        for (int p = 0; p < kRB && z0b + p < z1; ++p)
            for (int g = 0; g < L.NG; ++g) {
                const int yc = y0b + L.RG * g + r;
                if (yc < y1 && nvx > 0)
                    store8<TOut>((TOut*)P.out + ((long long)(z0b + p - P.tz) * P.oy + (yc - P.ty)) * (long long)P.ox + (xq - P.tx), q, nvx);
            }
        return;
    }
    // gather the records of the region's views into lanes: set A lanes (v*10 + q), v < 6; set B views 6, 7
    RecRegs R;
    {
        const int va = lane / 10, qa = lane - va * 10;
        const int ida = __shfl(rw, 8 + min(va, 7));
        const int idb = __shfl(rw, 8 + min(6 + va, 7));
        R.a = make_float4(0.f, 0.f, 0.f, 0.f);
        R.b = R.a;
        if (va < min(nviews, 6)) R.a = reinterpret_cast<const float4*>(P.views + ida)[qa];
        if ((NVC == 0 || NVC > 6) && nviews > 6 && va < nviews - 6) R.b = reinterpret_cast<const float4*>(P.views + idb)[qa];
    }
    if (NVC == 0) region_brick_generic<TIn, TOut>(P, R, rw, nviews, masks, z0b, z1, y0b, y1, x0b, x1, lane, lxb, strip);
    else region_brick<TIn, TOut, (NVC ? NVC : 1)>(P, R, nviews, masks, z0b, z1, y0b, y1, x0b, x1, lane, lxb, strip, nodes_w);
}

template <typename TIn, typename TOut, int NVC>
__global__ __launch_bounds__(256) void fuse_region_kernel(RegionParams P, int item0, int nitems) {
    __shared__ float s_strip[4][9 * 64];
    __shared__ float s_nodes[4][(NVC ? NVC : 1) * kRB * 32 * 3];
    const int lane = threadIdx.x & 63;
    // XCD-aware order: see copy_region_kernel
    const int wg = (blockIdx.x & 7) * (gridDim.x >> 3) + (blockIdx.x >> 3);
    const int li = wg * 4 + (threadIdx.x >> 6);
    if (li >= nitems) return;
    fuse_brick_item<TIn, TOut, NVC>(P, P.items[item0 + li], lane, s_strip[threadIdx.x >> 6], s_nodes[threadIdx.x >> 6]);
}

// The copy class and the classes with one and two views in ONE launch over a list that is ordered in space across the classes
// (see mvs_fuse_regions): an overlap brick and the interior brick next to it -- different classes, the same cache lines of the
// tile rows they share -- run back to back on one XCD instead of in two kernels at two times.  Every wavefront looks up the
// class of its item's region (bits 12-14 of Region::nviews) and takes that class's code; the three bodies need at most 126
// VGPRs each, so the launch keeps the occupancy the widest of them had on its own.  Items whose region id is 0xffff pad the
// eight per-XCD stretches of the list to one length.
template <typename TIn, typename TOut>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4, 4))) void fuse_region_mixed_kernel(RegionParams P, int item0, int nitems) {
    __shared__ float s_strip[4][9 * 64];
    __shared__ float s_nodes[4][2 * kRB * 32 * 3];
    const int lane = threadIdx.x & 63;
    const int wg = (blockIdx.x & 7) * (gridDim.x >> 3) + (blockIdx.x >> 3);
    const int li = wg * 4 + (threadIdx.x >> 6);
    if (li >= nitems) return;
    const Item it = P.items[item0 + li];
    const int rid = it.region_bx & 0xffff;
    if (rid == 0xffff) return;
    const int cls = __builtin_amdgcn_readfirstlane((P.regions[rid].nviews >> 12) & 7);
    if (cls == 4) copy_brick_item<TIn, TOut>(P, it, lane, s_strip[threadIdx.x >> 6]);
    else if (cls == 0) fuse_brick_item<TIn, TOut, 1>(P, it, lane, s_strip[threadIdx.x >> 6], s_nodes[threadIdx.x >> 6]);
    else fuse_brick_item<TIn, TOut, 2>(P, it, lane, s_strip[threadIdx.x >> 6], s_nodes[threadIdx.x >> 6]);
}

}  // namespace

// ---- host: region enumeration, brick list, launch ------------------------------------------------------------
namespace {
struct PlanCache {
    unsigned long long hash = 0;
    int nitems = 0;
    int class_count[5] = {0, 0, 0, 0, 0};   // bricks of regions with <=1, 2, <=4, >4 views, and copy-class bricks (contiguous, in this order)
    size_t rbytes = 0;
    bool valid = false;
    int mixed_count = 0;      // option "fuse_mixed": the padded, space-ordered list of the copy / one-view / two-view bricks (stored first)
    bool mixed = false;
    double class_in_vox[5] = {0, 0, 0, 0, 0};    // sum over the class's boxes of voxels x views (input voxel reads the class cannot avoid)
    double class_out_vox[5] = {0, 0, 0, 0, 0};   // voxels of the class's boxes
    // measurement (counters "fuse_class_ms_<k>"): with option serial_classes the class kernels of a launch run one after the other
    // and are bracketed by these timing events (created at the first such launch)
    hipEvent_t class_ev[6] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    bool class_timed[5] = {false, false, false, false, false};
};
PlanCache g_plan[MVS_MAX_DEVICES * MVS_MAX_LANES];
double g_region_plan_ms[MVS_MAX_DEVICES * MVS_MAX_LANES];

unsigned long long fnv1a(const void* p, size_t n, unsigned long long h) {
    const unsigned char* b = (const unsigned char*)p;
    for (size_t i = 0; i < n; ++i) { h ^= b[i]; h *= 1099511628211ull; }
    return h;
}

// Break points of one axis.  View borders that lie within `tol` of each other (tiles of one grid row/column after
// registration differ by a few pixels) are clustered: a cluster of lower borders contributes its minimum, a cluster
// of upper borders (hi + 1) its maximum, so the sliver between the clustered borders falls into the overlap cell,
// where the affected views are flagged "partial", and the single-view interior cells keep full coverage.
void axis_breakpoints(const TrView* htr, int n_views, int d, int t, int o, std::vector<int>* out) {
    // kinds: 0 lower border (cluster -> min), 1 upper border + 1 (-> max),
    //        2 end of the lower ramp zone (-> max), 3 start of the upper ramp zone (-> min)
    std::vector<std::pair<int, int>> ev;
    auto clampi = [&](int v) { return std::min(std::max(v, t), t + o); };
    for (int v = 0; v < n_views; ++v) {
        const int lo = htr[v].lo[d], hi = htr[v].hi[d];
        if (lo > hi) continue;
        ev.push_back({clampi(lo), 0});
        ev.push_back({clampi(hi + 1), 1});
        // A thin shell next to every border: inside it the blend weight of the view can round to 0 (the reference
        // outputs 0 there even for a single view, weights.py:502-507); outside it a voxel seen by ONE view is simply
        // the resampled value whatever the weight is, so single-view boxes off the shell need no weights at all.
        // Along x a 4-voxel sliver costs a whole cache line per row and view, so the shell is only cut where it matters:
        // next to a border that no other view covers (the rim of the mosaic).  Inside an overlap the box simply is not
        // flagged "positive" if a weight can vanish there (it cannot, away from the edges of the view).
        const int shell = 4;
        bool cut_lo = true, cut_hi = true;
        if (d == 2) {
            for (int w = 0; w < n_views; ++w) {
                if (w == v || htr[w].lo[2] > htr[w].hi[2]) continue;
                const bool touches = htr[w].lo[0] <= htr[v].hi[0] && htr[w].hi[0] >= htr[v].lo[0] && htr[w].lo[1] <= htr[v].hi[1] &&
                                     htr[w].hi[1] >= htr[v].lo[1];
                if (!touches) continue;
                if (htr[w].lo[2] <= lo - 8 && htr[w].hi[2] >= lo + shell + 8) cut_lo = false;
                if (htr[w].lo[2] <= hi - shell - 8 && htr[w].hi[2] >= hi + 8) cut_hi = false;
            }
        }
        if (2 * shell + 8 < hi - lo + 1) {
            if (cut_lo) ev.push_back({clampi(lo + shell), 2});
            if (cut_hi) ev.push_back({clampi(hi + 1 - shell), 3});
        }
    }
    const int tol = 16;
    out->clear();
    out->push_back(t);
    for (int kind_group = 0; kind_group < 2; ++kind_group) {
        // borders and shell ends are clustered separately so that a shell end never merges with a border
        std::vector<std::pair<int, int>> e2;
        for (auto& e : ev)
            if (e.second / 2 == kind_group) e2.push_back(e);
        std::sort(e2.begin(), e2.end());
        size_t i = 0;
        while (i < e2.size()) {
            size_t j = i;
            bool want_min = false, want_max = false;
            while (j < e2.size() && e2[j].first - e2[i].first <= tol) {
                // lower borders and the starts of upper shells cluster to their minimum, the rest to the maximum
                if (e2[j].second == 0 || e2[j].second == 3) want_min = true; else want_max = true;
                ++j;
            }
            if (want_min) out->push_back(e2[i].first);
            if (want_max) out->push_back(e2[j - 1].first);
            i = j;
        }
    }
    out->push_back(t + o);
    std::sort(out->begin(), out->end());
    out->erase(std::unique(out->begin(), out->end()), out->end());
}
}  // namespace

double mvs_regions_last_plan_ms(MvsContext* c) { return g_region_plan_ms[mvs_ctx_index(c->device)]; }

// what = 0: input voxel reads, 1: output voxels, 2: kernel ms of class `cls` in the last launch (-1: that launch was not a serial one)
double mvs_regions_class_stat(MvsContext* c, int what, int cls) {
    PlanCache& pc = g_plan[mvs_ctx_index(c->device)];
    if (cls < 0 || cls > 4) return -1.0;
    if (what == 0) return pc.class_in_vox[cls];
    if (what == 1) return pc.class_out_vox[cls];
    if (!pc.class_timed[cls] || !pc.class_ev[cls] || !pc.class_ev[cls + 1]) return -1.0;
    float ms = -1.f;
    if (hipEventSynchronize(pc.class_ev[cls + 1]) != hipSuccess) return -1.0;
    if (hipEventElapsedTime(&ms, pc.class_ev[cls], pc.class_ev[cls + 1]) != hipSuccess) return -1.0;
    return (double)ms;
}

// Returns MVS_OK and sets *done = true when the chunk was fused by the region kernel; *done = false means the
// caller must use the column kernel (more than kMaxRV views on one region, or too many regions/bricks).
int mvs_fuse_regions(MvsContext* c, const TrView* htr, const TrView* dtr, int n_views, int dtype, void* dout, const int64_t os[3],
                     const int64_t trim[3], bool* done) {
    *done = false;
    const int t[3] = {(int)trim[0], (int)trim[1], (int)trim[2]};
    const int o[3] = {(int)os[0], (int)os[1], (int)os[2]};
    unsigned long long h = fnv1a(htr, sizeof(TrView) * (size_t)n_views, 1469598103934665603ull);
    h = fnv1a(t, sizeof(t), h);
    h = fnv1a(o, sizeof(o), h);
    const int mixed_mode = c->fuse_mixed ? 1 : 0;
    h = fnv1a(&mixed_mode, sizeof(mixed_mode), h);
    PlanCache& pc = g_plan[mvs_ctx_index(c->device)];
    g_region_plan_ms[mvs_ctx_index(c->device)] = 0.0;
    const auto t_plan0 = std::chrono::steady_clock::now();
    char* dbuf = nullptr;
    int nitems = 0;
    size_t rbytes = 0;
    if (pc.valid && pc.hash == h && (c->dev[8].ptr || pc.nitems == 0)) {
        dbuf = (char*)c->dev[8].ptr;      // same geometry as the previous call: the plan is still on the device
        nitems = pc.nitems;
        rbytes = pc.rbytes;
    } else {
        std::vector<int> pts[3];
        for (int d = 0; d < 3; ++d) axis_breakpoints(htr, n_views, d, t[d], o[d], &pts[d]);
        const size_t ncell = (pts[0].size() - 1) * (pts[1].size() - 1) * (pts[2].size() - 1);
        if (ncell == 0 || ncell > 60000) return MVS_OK;
        std::vector<Region> regions;
        std::vector<Item> items_by_class[5];
        double in_vox[5] = {0, 0, 0, 0, 0}, out_vox[5] = {0, 0, 0, 0, 0};
        regions.reserve(ncell);
        std::vector<int> zviews, yviews;
        struct SlabRegion { int rid, nbx, bytes_per_item; };
        std::vector<SlabRegion> slab;              // the mixed-class regions of the current (z, y) slab, in x order
        std::vector<Item> mixed_items;
        std::vector<int> mixed_work;
        for (size_t iz = 0; iz + 1 < pts[0].size(); ++iz) {
            zviews.clear();
            for (int v = 0; v < n_views; ++v)
                if (htr[v].lo[0] < pts[0][iz + 1] && htr[v].hi[0] >= pts[0][iz] && htr[v].lo[1] <= htr[v].hi[1] && htr[v].lo[2] <= htr[v].hi[2]) zviews.push_back(v);
            for (size_t iy = 0; iy + 1 < pts[1].size(); ++iy) {
                slab.clear();
                yviews.clear();
                for (int v : zviews)
                    if (htr[v].lo[1] < pts[1][iy + 1] && htr[v].hi[1] >= pts[1][iy]) yviews.push_back(v);
                for (size_t ix = 0; ix + 1 < pts[2].size(); ++ix) {
                    Region R;
                    memset(&R, 0, sizeof(R));
                    R.z0 = pts[0][iz]; R.z1 = pts[0][iz + 1];
                    R.y0 = pts[1][iy]; R.y1 = pts[1][iy + 1];
                    R.x0 = pts[2][ix]; R.x1 = pts[2][ix + 1];
                    int nv = 0;
                    bool positive_full = false, all_positive = true;
                    for (int v : yviews) {
                        if (!(htr[v].lo[2] < R.x1 && htr[v].hi[2] >= R.x0)) continue;   // does not touch the box
                        if (nv == kMaxRV) return MVS_OK;                              // too many views: column kernel
                        const bool full = htr[v].lo[0] <= R.z0 && htr[v].hi[0] >= R.z1 - 1 && htr[v].lo[1] <= R.y0 &&
                                          htr[v].hi[1] >= R.y1 - 1 && htr[v].lo[2] <= R.x0 && htr[v].hi[2] >= R.x1 - 1;
                        // the profile is concave, so its minimum over the box sits at one of the 8 corners
                        float wmin = INFINITY;
                        for (int k = 0; k < 8; ++k) {
                            const int z = (k & 4) ? R.z1 - 1 : R.z0, y = (k & 2) ? R.y1 - 1 : R.y0, x = (k & 1) ? R.x1 - 1 : R.x0;
                            wmin = fminf(wmin, tr_weight_profile(htr[v], z, y, x));
                        }
                        const bool unit = full && wmin >= 1.f;          // weight exactly 1 everywhere
                        // weight > 0 everywhere: the float32 ramp (cos(pi (1 - W)) + 1) / 2 only vanishes when the cosine
                        // rounds to -1, i.e. W < 7.8e-5; at W = 3e-4 the cosine is 7 ulp away from -1
                        if (full && wmin >= 3e-4f) positive_full = true;
                        else all_positive = false;
                        if (unit) R.allone_mask |= 1 << nv;
                        if (!full) R.allone_mask |= 1 << (16 + nv);
                        R.ids[nv++] = v;
                    }
                    if (nv > 0 && all_positive) R.allone_mask |= 1 << 15;
                    // profiling only (WRONG results): every view counts as a full unit view, i.e. every brick takes the plain-average
                    // path -- the floor of what the weight evaluation can be brought down to
                    // (compiled only into profiling builds -- make CXXFLAGS+=-DMVS_PROFILING_ABLATIONS, tools/fuse_floor.sh: a stray
                    // environment variable must not be able to corrupt the shipped path's output)
#ifdef MVS_PROFILING_ABLATIONS
                    static const bool ablate_unit = getenv("MVS_FUSE_ALL_UNIT") != nullptr;
                    if (ablate_unit) R.allone_mask = ((1 << nv) - 1) | (1 << 15);
#endif
                    // brick width: 16 voxels for thin boxes, 512 (one full tile row per load instruction: the longest
                    // contiguous runs, 4.0 instead of 3.0 TB/s on the copy class) for wide copy-class boxes, else 128
                    int lxb = (R.x1 - R.x0 <= 32) ? 1 : 4;
                    if (nv == 1 && positive_full && R.x1 - R.x0 > 160) lxb = 6;
                    // overlap zones along x (about 100 voxels wide): 64-voxel bricks, so that each brick holds only ONE of the
                    // zone's two ramp ends and the other view classifies as "unit" (measured best of 16/32/64/128)
                    if (nv >= 2 && R.x1 - R.x0 > 32 && R.x1 - R.x0 <= 136) lxb = 3;
                    // (measured, round 4: 256- / 512-voxel bricks for the wide NV >= 2 boxes -- the copy class's layout -- lose:
                    // launch 10.06 -> 10.3 / 11.2 ms; every wavefront then spans a ramp end and takes the per-voxel weights)
                    const bool copy_class = (nv == 1) && positive_full;   // one full view with positive weight everywhere
                    const int cls = copy_class ? 4 : nv <= 1 ? 0 : nv == 2 ? 1 : nv <= 4 ? 2 : 3;
                    R.nviews = nv | (lxb << 8) | (cls << 12);
                    {
                        const double vox = (double)(R.z1 - R.z0) * (double)(R.y1 - R.y0) * (double)(R.x1 - R.x0);
                        in_vox[cls] += vox * nv;
                        out_vox[cls] += vox;
                    }
                    const int rid = (int)regions.size();
                    if (rid >= 65535) return MVS_OK;                     // (0xffff marks a padding item)
                    regions.push_back(R);
                    const int bxw = kRV << lxb;
                    const int nbz = (R.z1 - R.z0 + kRB - 1) / kRB, nby = (R.y1 - R.y0 + 31) / 32, nbx = (R.x1 - R.x0 + bxw - 1) / bxw;
                    if (nbz >= 65536 || nby >= 65536 || nbx >= 65536) return MVS_OK;
                    if (mixed_mode && (cls == 4 || cls <= 1)) {         // joins the space-ordered list of this slab (below)
                        slab.push_back(SlabRegion{rid, nbx, std::min(bxw, R.x1 - R.x0) * (std::max(nv, 1) + 1)});
                        continue;
                    }
                    std::vector<Item>& dst = items_by_class[cls];
                    // x fastest, then z, then y: bricks that are neighbours along x share the cache lines at their common
                    // edge, neighbours along z share a whole plane when the offsets are fractional; both reuses then happen
                    // within a few bricks, i.e. inside the L2 of the XCD that owns this stretch of the list
                    for (int by = 0; by < nby; ++by)
                        for (int bz = 0; bz < nbz; ++bz)
                            for (int bx = 0; bx < nbx; ++bx) dst.push_back({rid | (bx << 16), by | (bz << 16)});
                }
                if (!slab.empty()) {
                    // one (z, y) slab of the cell grid: its regions share the z / y extents and the brick grid; y block, then z
                    // block, then ALL regions along x -- so the bricks of a row of the mosaic are neighbours in the list whatever
                    // their class
                    const int z0s = pts[0][iz], z1s = pts[0][iz + 1], y0s = pts[1][iy], y1s = pts[1][iy + 1];
                    const int nbz = (z1s - z0s + kRB - 1) / kRB, nby = (y1s - y0s + 31) / 32;
                    for (int by = 0; by < nby; ++by)
                        for (int bz = 0; bz < nbz; ++bz)
                            for (const SlabRegion& sr : slab)
                                for (int bx = 0; bx < sr.nbx; ++bx) {
                                    mixed_items.push_back({sr.rid | (bx << 16), by | (bz << 16)});
                                    mixed_work.push_back(sr.bytes_per_item);
                                }
                }
            }
        }
        if (getenv("MVS_PLAN_STATS")) {
            double vox[6][2] = {{0}};
            for (const Region& R : regions) {
                const int nv = R.nviews & 0xff;
                const bool au = (R.allone_mask & ((1 << nv) - 1)) == ((1 << nv) - 1);
                const int cls = nv <= 1 ? nv : nv == 2 ? 2 : nv <= 4 ? 3 : 4;
                vox[cls][au ? 1 : 0] += (double)(R.z1 - R.z0) * (R.y1 - R.y0) * (R.x1 - R.x0);
            }
            {   // brick-level emulation of the device-side refinement
                double tot[3] = {0, 0, 0}, unitb[3] = {0, 0, 0};
                for (const Region& R : regions) {
                    const int nv = R.nviews & 0xff, lxb = (R.nviews >> 8) & 7;
                    if (nv < 2 || nv > 4) continue;
                    const int bxw = kRV << lxb, cls = nv == 2 ? 0 : 1;
                    for (int z = R.z0; z < R.z1; z += kRB)
                        for (int y = R.y0; y < R.y1; y += 32)
                            for (int x = R.x0; x < R.x1; x += bxw) {
                                bool all = true;
                                for (int v = 0; v < nv && all; ++v) {
                                    if ((R.allone_mask >> v) & 1) continue;
                                    if ((R.allone_mask >> (16 + v)) & 1) { all = false; break; }
                                    for (int k = 0; k < 8; ++k) {
                                        const int zz = (k & 4) ? std::min(z + kRB, R.z1) - 1 : z, yy = (k & 2) ? std::min(y + 32, R.y1) - 1 : y,
                                                  xx = (k & 1) ? std::min(x + bxw, R.x1) - 1 : x;
                                        if (!(tr_weight_profile(htr[R.ids[v]], zz, yy, xx) >= 1.f)) { all = false; break; }
                                    }
                                }
                                tot[cls] += 1; if (all) unitb[cls] += 1;
                            }
                }
                fprintf(stderr, "[mvs plan] unit bricks: nv2 %.0f / %.0f, nv3-4 %.0f / %.0f\n", unitb[0], tot[0], unitb[1], tot[1]);
            }
            fprintf(stderr, "[mvs plan] regions %zu; Mvox (ramp / all-unit): nv0 %.1f/%.1f nv1 %.1f/%.1f nv2 %.1f/%.1f nv3-4 %.1f/%.1f nv5+ %.1f/%.1f; bricks",
                    regions.size(), vox[0][0] / 1e6, vox[0][1] / 1e6, vox[1][0] / 1e6, vox[1][1] / 1e6, vox[2][0] / 1e6, vox[2][1] / 1e6,
                    vox[3][0] / 1e6, vox[3][1] / 1e6, vox[4][0] / 1e6, vox[4][1] / 1e6);
            for (int k = 0; k < 5; ++k) fprintf(stderr, " %zu", items_by_class[k].size());
            fprintf(stderr, "\n");
        }
        std::vector<Item> items;
        pc.mixed = mixed_mode != 0;
        pc.mixed_count = 0;
        if (!mixed_items.empty()) {
            // eight stretches of equal WORK (bytes moved), one per XCD (see the workgroup -> item mapping of the kernels), padded to
            // one length with no-op items
            long long total = 0;
            for (int w : mixed_work) total += w;
            size_t cut[9];
            cut[0] = 0;
            long long acc = 0;
            size_t i = 0;
            for (int k = 1; k <= 8; ++k) {
                const long long target = total * k / 8;
                while (i < mixed_items.size() && acc < target) acc += mixed_work[i++];
                cut[k] = k == 8 ? mixed_items.size() : i;
            }
            size_t longest = 0;
            for (int k = 0; k < 8; ++k) longest = std::max(longest, cut[k + 1] - cut[k]);
            const size_t L = (longest + 3) / 4 * 4;
            items.reserve(8 * L);
            for (int k = 0; k < 8; ++k) {
                items.insert(items.end(), mixed_items.begin() + (long)cut[k], mixed_items.begin() + (long)cut[k + 1]);
                items.resize((size_t)(k + 1) * L, Item{0xffff, 0});
            }
            pc.mixed_count = (int)items.size();
        }
        for (int k = 0; k < 5; ++k) {
            pc.class_count[k] = (int)items_by_class[k].size();
            pc.class_in_vox[k] = in_vox[k];
            pc.class_out_vox[k] = out_vox[k];
            items.insert(items.end(), items_by_class[k].begin(), items_by_class[k].end());
        }
        if (items.empty() || items.size() > (1u << 28)) return MVS_OK;
        rbytes = (regions.size() * sizeof(Region) + 255) / 256 * 256;
        const size_t ibytes = items.size() * sizeof(Item);
        char* hbuf = (char*)mvs_pinned_slot(c, 1, rbytes + ibytes + 256);   // slot 0 holds the view parameters still in flight
        if (!hbuf) return mvs_alloc_failed(c);
        pc.valid = false;
        dbuf = (char*)mvs_scratch(c, 8, rbytes + ibytes + 256);
        if (!dbuf) return mvs_alloc_failed(c);
        memcpy(hbuf, regions.data(), regions.size() * sizeof(Region));
        memcpy(hbuf + rbytes, items.data(), ibytes);
        { const int rcu = mvs_upload_small(c, dbuf, hbuf, rbytes + ibytes); if (rcu) return rcu; }
        mvs_pinned_mark(c, 1);
        nitems = (int)items.size();
        pc.hash = h;
        pc.nitems = nitems;
        pc.rbytes = rbytes;
        pc.valid = true;
        g_region_plan_ms[mvs_ctx_index(c->device)] = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_plan0).count();
    }
    RegionParams P;
    P.views = dtr;
    P.regions = (const Region*)dbuf;
    P.items = (const Item*)(dbuf + rbytes);
    P.nitems = nitems;
    P.out = dout;
    P.oz = o[0]; P.oy = o[1]; P.ox = o[2];
    P.tz = t[0]; P.ty = t[1]; P.tx = t[2];
    // The five class kernels write disjoint voxels and have different bottlenecks (the copy class is bound by memory, the
    // NV = 4 / 8 classes by their weight arithmetic): they run side by side -- NV = 2, the largest, on the main stream, the
    // others on side streams that start after everything queued so far (fork event) and are waited for at the end (join).
    const bool fork = !c->serial_classes && nitems >= 4096;   // small chunks: five event round trips cost more than the overlap gains
    if (fork) {      // side streams: created once per context (tens of ms: outside the timed section)
        const int rca = mvs_ensure_aux_streams(c);
        if (rca) return rca;
    }
    MVS_HIP_TRY(c, hipEventRecord(c->ev_start, c->stream));   // kernel time only (the plan is host work, cached per geometry)
    if (fork) MVS_HIP_TRY(c, hipEventRecord(c->ev_fork, c->stream));
    const int side_of_class[5] = {0, -1, 1, 2, 3};          // class -> side stream (-1: main)
    bool side_used[4] = {false, false, false, false};
    int item0 = 0;
    if (pc.mixed_count) {       // copy + one-view + two-view bricks: one launch over the space-ordered list, on the main stream
        const int cnt = pc.mixed_count;                     // = 8 L, L a multiple of 4: grid = 2 L workgroups, a multiple of 8
        const dim3 grid(cnt / 4), block(256);
        if (dtype == MVS_U8) hipLaunchKernelGGL((fuse_region_mixed_kernel<unsigned char, unsigned char>), grid, block, 0, c->stream, P, 0, cnt);
        else if (dtype == MVS_U16) hipLaunchKernelGGL((fuse_region_mixed_kernel<unsigned short, unsigned short>), grid, block, 0, c->stream, P, 0, cnt);
        else hipLaunchKernelGGL((fuse_region_mixed_kernel<float, float>), grid, block, 0, c->stream, P, 0, cnt);
        item0 = cnt;
    }
    const bool time_classes = c->serial_classes && !pc.mixed_count;
    if (time_classes)
        for (hipEvent_t& e : pc.class_ev)
            if (!e) MVS_HIP_TRY(c, hipEventCreate(&e));
    for (int k = 0; k < 5; ++k) {
        const int cnt = pc.class_count[k];
        pc.class_timed[k] = time_classes && cnt > 0;
        if (time_classes) MVS_HIP_TRY(c, hipEventRecord(pc.class_ev[k], c->stream));
        hipStream_t kstream = c->stream;
        if (fork && cnt && side_of_class[k] >= 0) {
            kstream = c->aux_stream[side_of_class[k]];
            side_used[side_of_class[k]] = true;
            MVS_HIP_TRY(c, hipStreamWaitEvent(kstream, c->ev_fork, 0));
        }
        if (cnt && k == 4) {
            const dim3 grid(((cnt + 3) / 4 + 7) / 8 * 8), block(256);   // multiple of 8: see the XCD mapping in the kernels
            if (dtype == MVS_U8) hipLaunchKernelGGL((copy_region_kernel<unsigned char, unsigned char>), grid, block, 0, kstream, P, item0, cnt);
            else if (dtype == MVS_U16) hipLaunchKernelGGL((copy_region_kernel<unsigned short, unsigned short>), grid, block, 0, kstream, P, item0, cnt);
            else hipLaunchKernelGGL((copy_region_kernel<float, float>), grid, block, 0, kstream, P, item0, cnt);
        } else if (cnt) {
            const dim3 grid(((cnt + 3) / 4 + 7) / 8 * 8), block(256);   // multiple of 8: see the XCD mapping in the kernels
#define MVS_RK(T, NVC) hipLaunchKernelGGL((fuse_region_kernel<T, T, NVC>), grid, block, 0, kstream, P, item0, cnt)
#define MVS_RKD(NVC) do { if (dtype == MVS_U8) MVS_RK(unsigned char, NVC); else if (dtype == MVS_U16) MVS_RK(unsigned short, NVC); else MVS_RK(float, NVC); } while (0)
            if (k == 0) MVS_RKD(1); else if (k == 1) MVS_RKD(2); else if (k == 2) MVS_RKD(4); else MVS_RKD(8);
#undef MVS_RKD
#undef MVS_RK
        }
        item0 += cnt;
    }
    if (time_classes) MVS_HIP_TRY(c, hipEventRecord(pc.class_ev[5], c->stream));
    MVS_HIP_TRY(c, hipGetLastError());
    for (int a = 0; a < 4; ++a)
        if (side_used[a]) {
            MVS_HIP_TRY(c, hipEventRecord(c->ev_join[a], c->aux_stream[a]));
            MVS_HIP_TRY(c, hipStreamWaitEvent(c->stream, c->ev_join[a], 0));
        }
    *done = true;
    return MVS_OK;
}
