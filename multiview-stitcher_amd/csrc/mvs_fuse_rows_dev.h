// mvs_fuse_rows_dev.h -- internal: device side of the row-owning translation fast path of mvs_fuse_chunk (gfx950).
//
// Work decomposition (host side: mvs_fuse_rows.hip).  The chunk is cut along z and y at the view borders into
// STRIPS (a z cell x a y cell); inside a strip the set of views that touch a row is constant, so the row is cut
// along x ONCE per strip into CELLS with a constant view list (interior of a tile | overlap of two tiles | ...).
// A workgroup owns 16 consecutive COMPLETE output rows of one plane (4 wavefronts x 4 rows) and each wavefront
// walks all x cells of its 4 rows left to right:
//   * every output cache line is written by one wavefront within a few hundred cycles (whole 3.5 KB rows,
//     56 KB contiguous per workgroup) instead of by two kernels at different times,
//   * every input tile row is consumed by one wavefront in one go: [interior | overlap] pieces of the same
//     1 KiB row are requested back to back, so each HBM line is fetched once (the region kernels fetched the
//     lines at region faces once per region: 1.47x read traffic).
// Per cell the wavefront runs the code of the cell's class (zero fill | one full view: copy | blend of <= NVMAX
// views) with the view constants in scalar registers (scalar loads from the TrView records).  Strips are binned
// by the largest view count of their cells (NVMAX = 2 | 4 | 8) and by whether any of their views needs more
// than one tap (FRAC), one kernel instantiation each, so the common case (integer offsets, <= 2 views per
// voxel: 68 % of a 20 %-overlap mosaic) keeps a small register footprint.
//
// Arithmetic is that of the region kernels (mvs_fuse_region.hip): same tap order, same weight profile, same
// accumulator rules; reference: fusion/_core.py:1608-1713, weights.py:325-345, 391-511, transformation.py:136-139.
#pragma once
#include "mvs_fuse_tr.h"

#include <type_traits>

namespace mvsrows {

constexpr int kRV = 8;       // voxels per lane
constexpr int kWR = 4;       // rows per wavefront
constexpr int kWG = 4;       // wavefronts per workgroup
constexpr int kMaxCV = 8;    // views per cell

struct Cell {                // 64 bytes
    int x0, x1;               // chunk-index range, end exclusive
    int nv_lxb_cls;           // nviews | lxb << 8 | cls << 16   (cls 0: no view, 1: copy, 2: blend)
    int masks;                // bits 0-7: view ids[v] has weight 1 everywhere in the 3D box of (strip, cell); bit 15: every
                              // view covers the box with a weight > 0 everywhere; bits 16-23: view covers the box only partly
    int ids[kMaxCV];
    int pad[4];
};
static_assert(sizeof(Cell) == 64, "Cell layout");

struct Strip { int z0, z1, y0, y1, cell0, ncells, pad0, pad1; };
static_assert(sizeof(Strip) == 32, "Strip layout");

struct RowItem { int strip, z, y, pad; };   // one workgroup: rows y .. y + 15 (clipped to the strip) of plane z

typedef unsigned int u32x2_t __attribute__((ext_vector_type(2)));
typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));

// 8 consecutive elements (+ the 9th when NINE) of one row through a bounds-checked buffer load
template <typename T, bool NINE> struct Row8;
template <bool NINE> struct Row8<unsigned short, NINE> {
    static constexpr int NW = NINE ? 5 : 4;
    static __device__ __forceinline__ void load(__amdgpu_buffer_rsrc_t r, int vo, unsigned int (&w)[NW]) {
        const u32x4_t a = __builtin_amdgcn_raw_buffer_load_b128(r, vo, 0, 0);
        w[0] = a.x; w[1] = a.y; w[2] = a.z; w[3] = a.w;
        if (NINE) w[NW - 1] = __builtin_amdgcn_raw_buffer_load_b16(r, vo + 16, 0, 0);
    }
    static __device__ __forceinline__ void decode(const unsigned int (&w)[NW], float (&v)[9]) {
#pragma unroll
        for (int k = 0; k < 4; ++k) { v[2 * k] = (float)(w[k] & 0xffffu); v[2 * k + 1] = (float)(w[k] >> 16); }
        v[8] = NINE ? (float)(w[NW - 1] & 0xffffu) : 0.f;
    }
};
template <bool NINE> struct Row8<unsigned char, NINE> {
    static constexpr int NW = NINE ? 3 : 2;
    static __device__ __forceinline__ void load(__amdgpu_buffer_rsrc_t r, int vo, unsigned int (&w)[NW]) {
        const u32x2_t a = __builtin_amdgcn_raw_buffer_load_b64(r, vo, 0, 0);
        w[0] = a.x; w[1] = a.y;
        if (NINE) w[NW - 1] = __builtin_amdgcn_raw_buffer_load_b8(r, vo + 8, 0, 0);
    }
    static __device__ __forceinline__ void decode(const unsigned int (&w)[NW], float (&v)[9]) {
#pragma unroll
        for (int k = 0; k < 8; ++k) v[k] = (float)((w[k >> 2] >> (8 * (k & 3))) & 0xffu);
        v[8] = NINE ? (float)(w[NW - 1] & 0xffu) : 0.f;
    }
};
template <bool NINE> struct Row8<float, NINE> {
    static constexpr int NW = NINE ? 9 : 8;
    static __device__ __forceinline__ void load(__amdgpu_buffer_rsrc_t r, int vo, unsigned int (&w)[NW]) {
        const u32x4_t a = __builtin_amdgcn_raw_buffer_load_b128(r, vo, 0, 0);
        const u32x4_t b = __builtin_amdgcn_raw_buffer_load_b128(r, vo + 16, 0, 0);
        w[0] = a.x; w[1] = a.y; w[2] = a.z; w[3] = a.w; w[4] = b.x; w[5] = b.y; w[6] = b.z; w[7] = b.w;
        if (NINE) w[NW - 1] = __builtin_amdgcn_raw_buffer_load_b32(r, vo + 32, 0, 0);
    }
    static __device__ __forceinline__ void decode(const unsigned int (&w)[NW], float (&v)[9]) {
#pragma unroll
        for (int k = 0; k < 8; ++k) v[k] = __uint_as_float(w[k]);
        v[8] = NINE ? __uint_as_float(w[NW - 1]) : 0.f;
    }
};

template <typename TIn> __device__ __forceinline__ float load_elem(__amdgpu_buffer_rsrc_t r, int o) {
    if (sizeof(TIn) == 2) return (float)(unsigned short)__builtin_amdgcn_raw_buffer_load_b16(r, o, 0, 0);
    if (sizeof(TIn) == 1) return (float)(unsigned char)__builtin_amdgcn_raw_buffer_load_b8(r, o, 0, 0);
    return __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(r, o, 0, 0));
}

// Element-wise re-fetch of a 9-element window that touches the first / last bytes of the slab (a vector buffer load
// that is not entirely in range comes back as 0).  Rare (first / last rows of a slab only): out of line, values handed
// over through a wavefront-private LDS strip.
template <typename TIn>
__device__ __noinline__ void row8_refetch_lds(__amdgpu_buffer_rsrc_t r, int o, float* strip) {
    for (int j = 0; j < 9; ++j) strip[j * 64] = load_elem<TIn>(r, o + j * (int)sizeof(TIn));
}
template <typename TIn>
__device__ __forceinline__ void row8_refetch(__amdgpu_buffer_rsrc_t r, int o, float (&v)[9], float* strip) {
    row8_refetch_lds<TIn>(r, o, strip);
#pragma unroll
    for (int j = 0; j < 9; ++j) v[j] = strip[j * 64];
}
// does the 9-element window at byte offset o straddle the start or the end of the slab?
template <typename TIn> __device__ __forceinline__ bool straddles(int o, int nbytes) {
    constexpr int WB = 9 * (int)sizeof(TIn);
    return (o < 0 && o + WB > 0) || (o < nbytes && o + WB > nbytes);
}

template <typename TOut> __device__ __forceinline__ TOut cast_r(float v);
template <> __device__ __forceinline__ float cast_r<float>(float v) { return v; }
template <> __device__ __forceinline__ unsigned short cast_r<unsigned short>(float v) { return (unsigned short)(int)v; }
template <> __device__ __forceinline__ unsigned char cast_r<unsigned char>(float v) { return (unsigned char)(int)v; }

template <typename T> struct Out8;
template <> struct Out8<unsigned short> { typedef unsigned short v8 __attribute__((ext_vector_type(8), aligned(2))); };
template <> struct Out8<unsigned char> { typedef unsigned char v8 __attribute__((ext_vector_type(8), aligned(1))); };
template <> struct Out8<float> { typedef float v8 __attribute__((ext_vector_type(8), aligned(4))); };

template <typename TOut>
__device__ __forceinline__ void store8(TOut* p, const float (&q)[kRV], int nvalid) {
    if (nvalid >= kRV) {
        typename Out8<TOut>::v8 v;
#pragma unroll
        for (int j = 0; j < kRV; ++j) v[j] = cast_r<TOut>(q[j]);
        *reinterpret_cast<typename Out8<TOut>::v8*>(p) = v;
    } else {
#pragma unroll
        for (int j = 0; j < kRV; ++j)
            if (j < nvalid) p[j] = cast_r<TOut>(q[j]);
    }
}

// Lane layout of one step of a wavefront's 4 rows: 2^lxb lanes side by side along x (8 voxels each), the rest stacked
// along y.  lxb = 6: 1 row x 512 voxels, 4 steps; lxb = 5: 2 rows x 256 voxels, 2 steps; lxb = 4: 4 rows x 128 voxels.
struct LaneMap {
    int r, c, RG, NG, BXW;
    __device__ __forceinline__ LaneMap(int lane, int lxb)
        : r(lane >> lxb), c(lane & ((1 << lxb) - 1)), RG(64 >> lxb), NG(kWR / (64 >> lxb)), BXW(kRV << lxb) {}
};

// load constants of one view at plane zc (wave-uniform: scalar registers)
struct VC {
    __amdgpu_buffer_rsrc_t rsrc;
    int nbytes, sbase, syb, szb;     // bytes; window offset = sbase + row * syb + x * ES
};
template <typename TIn>
__device__ __forceinline__ VC view_consts(const TrView& V, int zc) {
    constexpr int ES = (int)sizeof(TIn);
    VC c;
    c.nbytes = (int)V.span * ES;
    c.rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)V.data, 0, c.nbytes, 0x00020000);
    c.syb = V.stride_y * ES;
    c.szb = V.stride_z * ES;
    c.sbase = ((zc + V.io[0]) * V.stride_z + V.io[1] * V.stride_y + V.io[2]) * ES;
    return c;
}

// does view V need more than the single tap (z, y, x) + io?  Integer tiles: only with a fractional offset.  Float tiles
// with linear interpolation: always -- scipy multiplies the second tap by its zero weight, so a NaN there poisons the
// sample (transformation.py:136-139 -> ni_interpolation.c), which the fused value must reproduce.
template <typename TIn> __device__ __forceinline__ bool needs_taps(const TrView& V) {
    const bool frac = (V.fw[0] > 0.f) || (V.fw[1] > 0.f) || (V.fw[2] > 0.f);
    return frac || (std::is_floating_point<TIn>::value && V.linear != 0);
}

// Row-uniform nodes of view V at plane zc, row yc: G1, dG of the x profile and whether the row lies inside the support
// along z and y (same arithmetic as tr_weight_profile).
__device__ __forceinline__ void row_nodes(const TrView& V, int zc, int yc, float& G1, float& dG, bool& inside) {
    float az0 = INFINITY, az1 = INFINITY, fz = 0.f, uz = 0.f;
    const bool has_z = V.wnz > 1;
    if (has_z) {
        uz = fold_u(zc, V.sup_ilo[0], V.sup_flo[0], V.sup_ihi[0], V.sup_fhi[0], V.sup_k[0]);
        tent_cell(fmaxf(uz, 0.f), V.ws[0], az0, az1, fz);
    }
    const float uy = fold_u(yc, V.sup_ilo[1], V.sup_flo[1], V.sup_ihi[1], V.sup_fhi[1], V.sup_k[1]);
    inside = (uz >= 0.f) && (uy >= 0.f);
    float ay0, ay1, fy;
    tent_cell(fmaxf(uy, 0.f), V.ws[1], ay0, ay1, fy);
    const float uz_ = 1.f - fz, uy_ = 1.f - fy;
    const float m00 = fminf(az0, ay0), m01 = fminf(az0, ay1), m10 = fminf(az1, ay0), m11 = fminf(az1, ay1);
    const float a1 = V.ws[2], a2 = 2.f * V.ws[2];
    float g0 = fmaf(fminf(m01, a1), fy, fminf(m00, a1) * uy_);
    float g1 = fmaf(fminf(m11, a1), fy, fminf(m10, a1) * uy_);
    G1 = has_z ? fmaf(g1, fz, g0 * uz_) : g0;
    g0 = fmaf(fminf(m01, a2), fy, fminf(m00, a2) * uy_);
    g1 = fmaf(fminf(m11, a2), fy, fminf(m10, a2) * uy_);
    const float G2 = has_z ? fmaf(g1, fz, g0 * uz_) : g0;
    dG = G2 - G1;
}

// Values of view V at the lane's 8 voxels of row yl (chunk index) starting at chunk x = xl, all taps: the four tap
// rows are requested back to back with bounds-checked buffer loads, then interpolated (x, then z, then y) like
// sample_view (mvs_fuse.hip).  Float tiles: the second tap at the upper border of an axis is scipy's mirrored index
// n - 2 (weight 0 there; matters only for NaN propagation).
template <typename TIn>
__device__ __forceinline__ void fetch_taps(const TrView& V, const VC& c, int zc, int yl, int xl, float* strip, float (&val)[kRV]) {
    constexpr int ES = (int)sizeof(TIn);
    constexpr bool ISF = std::is_floating_point<TIn>::value;
    typedef Row8<TIn, true> R9;
    const float wz = V.fw[0], wy = V.fw[1], wx = V.fw[2];
    const int vo = c.sbase + yl * c.syb + xl * ES;
    int dy = c.syb, dz = c.szb;
    if (ISF) {
        if (yl + V.io[1] + 1 >= V.n[1]) dy = (V.n[1] > 1) ? -c.syb : 0;
        if (zc + V.io[0] + 1 >= V.n[0]) dz = (V.n[0] > 1) ? -c.szb : 0;
    }
    unsigned int w00[R9::NW], w01[R9::NW], w10[R9::NW], w11[R9::NW];
    R9::load(c.rsrc, vo, w00);
    R9::load(c.rsrc, vo + dy, w01);
    R9::load(c.rsrc, vo + dz, w10);
    R9::load(c.rsrc, vo + dz + dy, w11);
    float e00[9], e01[9], e10[9], e11[9];
    R9::decode(w00, e00);
    R9::decode(w01, e01);
    R9::decode(w10, e10);
    R9::decode(w11, e11);
    const int o1 = vo + dy, o2 = vo + dz, o3 = vo + dz + dy;
    if (__any(straddles<TIn>(vo, c.nbytes) || straddles<TIn>(o1, c.nbytes) || straddles<TIn>(o2, c.nbytes) || straddles<TIn>(o3, c.nbytes))) {
        if (__any(straddles<TIn>(vo, c.nbytes))) row8_refetch<TIn>(c.rsrc, vo, e00, strip);
        if (__any(straddles<TIn>(o1, c.nbytes))) row8_refetch<TIn>(c.rsrc, o1, e01, strip);
        if (__any(straddles<TIn>(o2, c.nbytes))) row8_refetch<TIn>(c.rsrc, o2, e10, strip);
        if (__any(straddles<TIn>(o3, c.nbytes))) row8_refetch<TIn>(c.rsrc, o3, e11, strip);
    }
    if (ISF) {
        // x: the tap right of the last column nx - 1 is column nx - 2
        const int je = (V.n[2] - 1) - (xl + V.io[2]);          // window index of the last column
        if (__any((unsigned)je < (unsigned)kRV)) {
            const int om = (V.n[2] > 1) ? (je - 1) * ES : je * ES;   // byte offset of the mirrored column inside the window
            const bool mine = (unsigned)je < (unsigned)kRV;
            const float m00 = mine ? load_elem<TIn>(c.rsrc, vo + om) : 0.f, m01 = mine ? load_elem<TIn>(c.rsrc, o1 + om) : 0.f;
            const float m10 = mine ? load_elem<TIn>(c.rsrc, o2 + om) : 0.f, m11 = mine ? load_elem<TIn>(c.rsrc, o3 + om) : 0.f;
#pragma unroll
            for (int j = 0; j < kRV; ++j)
                if (mine && j == je) { e00[j + 1] = m00; e01[j + 1] = m01; e10[j + 1] = m10; e11[j + 1] = m11; }
        }
    }
    const float ux = 1.f - wx, uy = 1.f - wy, uz = 1.f - wz;
#pragma unroll
    for (int j = 0; j < kRV; ++j) {
        const float a00 = fmaf(e00[j + 1], wx, e00[j] * ux), a01 = fmaf(e01[j + 1], wx, e01[j] * ux);
        const float a10 = fmaf(e10[j + 1], wx, e10[j] * ux), a11 = fmaf(e11[j + 1], wx, e11[j] * ux);
        const float s0 = fmaf(a10, wz, a00 * uz), s1 = fmaf(a11, wz, a01 * uz);
        val[j] = fmaf(s1, wy, s0 * uy);
    }
}

template <typename TOut>
__device__ __forceinline__ TOut* out_ptr(TOut* out, int zc, int yc, int xq, int oy, int ox, int tz, int ty, int tx) {
    return out + ((long long)(zc - tz) * oy + (yc - ty)) * (long long)ox + (xq - tx);
}

// ---- cell class 0: no view ------------------------------------------------------------------------------------
template <typename TOut>
__device__ __forceinline__ void zero_cell(int zc, int yw, int y1, int x0, int x1, int lxb, int lane, TOut* out, int oy, int ox, int tz,
                                          int ty, int tx) {
    const LaneMap L(lane, lxb);
    const float q[kRV] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    for (int xb = x0; xb < x1; xb += L.BXW) {
        const int xq = xb + kRV * L.c, nvx = min(max(x1 - xq, 0), kRV);
        for (int g = 0; g < L.NG; ++g) {
            const int yc = yw + L.RG * g + L.r;
            if (yc < y1 && nvx > 0) store8<TOut>(out_ptr(out, zc, yc, xq, oy, ox, tz, ty, tx), q, nvx);
        }
    }
}

// ---- cell class 1: ONE view that covers the cell with a weight > 0 everywhere: the result is the resampled value ----
template <typename TIn, typename TOut, bool FRAC>
__device__ __forceinline__ void copy_cell(const TrView& V, int zc, int yw, int y1, int x0, int x1, int lxb, int lane, float* strip,
                                          TOut* out, int oy, int ox, int tz, int ty, int tx) {
    constexpr int ES = (int)sizeof(TIn);
    constexpr bool ISF = std::is_floating_point<TIn>::value;
    typedef Row8<TIn, false> R8;
    const LaneMap L(lane, lxb);
    const VC c = view_consts<TIn>(V, zc);
    const bool taps = FRAC && needs_taps<TIn>(V);
    for (int xb = x0; xb < x1; xb += L.BXW) {
        const int xq = xb + kRV * L.c;
        const int nvx = min(max(x1 - xq, 0), kRV);
        const int xl = (nvx > 0) ? xq : xb;        // lanes beyond the cell read a valid window, nothing is stored
        if (!taps) {
            // all row loads of the wavefront's 4 rows are issued back to back before the first one is consumed
            unsigned int raw[kWR][R8::NW];
            int vo[kWR];
#pragma unroll
            for (int g = 0; g < kWR; ++g)
                if (g < L.NG) {
                    const int yl = min(yw + L.RG * g + L.r, y1 - 1);
                    vo[g] = c.sbase + yl * c.syb + xl * ES;
                    R8::load(c.rsrc, vo[g], raw[g]);
                }
#pragma unroll
            for (int g = 0; g < kWR; ++g)
                if (g < L.NG) {
                    const int yc = yw + L.RG * g + L.r;
                    float e[9];
                    R8::decode(raw[g], e);
                    if (__any(straddles<TIn>(vo[g], c.nbytes))) row8_refetch<TIn>(c.rsrc, vo[g], e, strip);
                    float q[kRV];
#pragma unroll
                    for (int j = 0; j < kRV; ++j) q[j] = (!ISF || e[j] == e[j]) ? e[j] : 0.f;   // nan_to_num (float tiles)
                    if (yc < y1 && nvx > 0) store8<TOut>(out_ptr(out, zc, yc, xq, oy, ox, tz, ty, tx), q, nvx);
                }
        } else {
            for (int g = 0; g < L.NG; ++g) {
                const int yc = yw + L.RG * g + L.r;
                if (yw + L.RG * g >= y1) break;
                const int yl = min(yc, y1 - 1);
                float val[kRV], q[kRV];
                fetch_taps<TIn>(V, c, zc, yl, xl, strip, val);
#pragma unroll
                for (int j = 0; j < kRV; ++j) q[j] = (val[j] == val[j]) ? val[j] : 0.f;
                if (yc < y1 && nvx > 0) store8<TOut>(out_ptr(out, zc, yc, xq, oy, ox, tz, ty, tx), q, nvx);
            }
        }
    }
}

// ---- cell class 2: weighted average of <= NVMAX views ---------------------------------------------------------------
template <typename TIn, typename TOut, int NVMAX, bool FRAC>
__device__ __forceinline__ void blend_cell(const TrView* __restrict__ views, const Cell& C, int nv, int masks, int zc, int yw, int y1,
                                           int x0, int x1, int lxb, int lane, float* strip, TOut* out, int oy, int ox, int tz,
                                           int ty, int tx) {
    constexpr int ES = (int)sizeof(TIn);
    constexpr bool ISF = std::is_floating_point<TIn>::value;
    typedef Row8<TIn, false> R8;
    const LaneMap L(lane, lxb);
    const int full = (1 << nv) - 1;
    int allone_mask = masks & 0xff;
    const int partial_mask = (masks >> 16) & 0xff;
    const bool allpos = (masks >> 15) & 1;
    const float need = (nv == 1) ? 3e-4f : 1.f;   // a voxel seen by ONE view only needs a weight that does not round to 0

    VC vc[NVMAX];
    bool alltap1 = true;       // every view is read with a single tap
#pragma unroll
    for (int v = 0; v < NVMAX; ++v)
        if (v < nv) {
            const TrView& V = views[C.ids[v]];
            vc[v] = view_consts<TIn>(V, zc);
            if (FRAC && needs_taps<TIn>(V)) alltap1 = false;
        }

    // The host classified the 3D box of (strip, cell); the set {weight == 1} is curved, so most rows of a box can be
    // "unit" without the box being so.  Refine for this visit (4 rows x the cell at plane zc): the profile is concave
    // along every axis line, hence its minimum over the rectangle sits at one of its 4 corners.
    if ((allone_mask & full) != full) {
        const int k = lane & 3;
        const int yk = (k & 2) ? y1 - 1 : yw, xk = (k & 1) ? x1 - 1 : x0;
#pragma unroll
        for (int v = 0; v < NVMAX; ++v)
            if (v < nv && !((allone_mask >> v) & 1) && !((partial_mask >> v) & 1)) {
                const float W = tr_weight_profile(views[C.ids[v]], zc, yk, xk);
                if (!__any(!(W >= need))) allone_mask |= 1 << v;
            }
    }
    const bool all_unit = (allone_mask & full) == full;

    for (int xb = x0; xb < x1; xb += L.BXW) {
        const int xq = xb + kRV * L.c;
        const int nvx = min(max(x1 - xq, 0), kRV);
        const int xl = (nvx > 0) ? xq : xb;

        if (all_unit && !partial_mask && !ISF && alltap1) {
            // every view in bounds with weight 1 and a single tap: plain mean; all loads of a row step back to back
            const float rn = __builtin_amdgcn_rcpf((float)nv);
            for (int g = 0; g < L.NG; ++g) {
                if (yw + L.RG * g >= y1) break;
                const int yc = yw + L.RG * g + L.r;
                const int yl = min(yc, y1 - 1);
                unsigned int raw[NVMAX][R8::NW];
                int vo[NVMAX];
#pragma unroll
                for (int v = 0; v < NVMAX; ++v)
                    if (v < nv) {
                        vo[v] = vc[v].sbase + yl * vc[v].syb + xl * ES;
                        R8::load(vc[v].rsrc, vo[v], raw[v]);
                    }
                float num[kRV];
#pragma unroll
                for (int j = 0; j < kRV; ++j) num[j] = 0.f;
#pragma unroll
                for (int v = 0; v < NVMAX; ++v)
                    if (v < nv) {
                        float e[9];
                        R8::decode(raw[v], e);
                        if (__any(straddles<TIn>(vo[v], vc[v].nbytes))) row8_refetch<TIn>(vc[v].rsrc, vo[v], e, strip);
#pragma unroll
                        for (int j = 0; j < kRV; ++j) num[j] += e[j];
                    }
                float q[kRV];
#pragma unroll
                for (int j = 0; j < kRV; ++j) q[j] = num[j] * rn;
                if (yc < y1 && nvx > 0) store8<TOut>(out_ptr(out, zc, yc, xq, oy, ox, tz, ty, tx), q, nvx);
            }
            continue;
        }

        for (int g = 0; g < L.NG; ++g) {
            if (yw + L.RG * g >= y1) break;
            const int yc = yw + L.RG * g + L.r;
            const bool row_ok = yc < y1;
            const int yl = row_ok ? yc : y1 - 1;

            float num[kRV], den[kRV], last[kRV], wlast[kRV];
#pragma unroll
            for (int j = 0; j < kRV; ++j) { num[j] = 0.f; den[j] = 0.f; last[j] = 0.f; wlast[j] = 0.f; }

            // single-tap views: the rows of ALL views are requested before the first one is consumed
            unsigned int raw[NVMAX][R8::NW];
            int vo[NVMAX];
            if (alltap1) {
#pragma unroll
                for (int v = 0; v < NVMAX; ++v)
                    if (v < nv) {
                        vo[v] = vc[v].sbase + yl * vc[v].syb + xl * ES;
                        R8::load(vc[v].rsrc, vo[v], raw[v]);
                    }
            }
#pragma unroll
            for (int v = 0; v < NVMAX; ++v) {
                if (v >= nv) break;
                const TrView& V = views[C.ids[v]];
                float val[kRV];
                if (alltap1) {
                    float e[9];
                    R8::decode(raw[v], e);
                    if (__any(straddles<TIn>(vo[v], vc[v].nbytes))) row8_refetch<TIn>(vc[v].rsrc, vo[v], e, strip);
#pragma unroll
                    for (int j = 0; j < kRV; ++j) val[j] = e[j];
                } else if (FRAC && needs_taps<TIn>(V)) {
                    fetch_taps<TIn>(V, vc[v], zc, yl, xl, strip, val);
                } else {
                    const int o = vc[v].sbase + yl * vc[v].syb + xl * ES;
                    unsigned int w[R8::NW];
                    R8::load(vc[v].rsrc, o, w);
                    float e[9];
                    R8::decode(w, e);
                    if (__any(straddles<TIn>(o, vc[v].nbytes))) row8_refetch<TIn>(vc[v].rsrc, o, e, strip);
#pragma unroll
                    for (int j = 0; j < kRV; ++j) val[j] = e[j];
                }

                // views that cover the box only partly: per-voxel in-bounds test against the view's valid box
                const bool partial = (partial_mask >> v) & 1;
                bool inb[kRV];
#pragma unroll
                for (int j = 0; j < kRV; ++j) inb[j] = true;
                if (partial) {
                    const bool zy_ok = (zc >= V.lo[0]) && (zc <= V.hi[0]) && (yc >= V.lo[1]) && (yc <= V.hi[1]);
                    const int jlo = V.lo[2] - xq, jw = V.hi[2] - V.lo[2];
#pragma unroll
                    for (int j = 0; j < kRV; ++j) inb[j] = zy_ok && ((unsigned)(j - jlo) <= (unsigned)jw);
                }
                bool unit = (allone_mask >> v) & 1;
                float w[kRV];
                if (!unit) {
                    float G1, dG;
                    bool inside;
                    row_nodes(V, zc, yl, G1, dG, inside);
                    const float kx = V.sup_k[2];
                    const float dlb = (float)(xl - V.sup_ilo[2]), dhb = (float)(V.sup_ihi[2] - xl);     // (one rounding per voxel: see mvs_fuse_region.hip)
                    const float dl0 = dlb - V.sup_flo[2], dh0 = dhb - V.sup_fhi[2];
                    // The profile is concave along x, so over the lane's 8 voxels its minimum sits at voxel 0 or 7:
                    // two evaluations tell whether the whole segment has weight 1.
                    const float u0 = fminf(dl0, dh0) * kx, u7 = fminf((dlb + 7.f) - V.sup_flo[2], (dhb - 7.f) - V.sup_fhi[2]) * kx;
                    const float W0 = (u0 >= 0.f && inside) ? row_profile(u0, G1, dG) : 0.f;
                    const float W7 = (u7 >= 0.f && inside) ? row_profile(u7, G1, dG) : 0.f;
                    const bool lane_unit = fminf(W0, W7) >= need;
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
                            const float u = fminf((dlb + (float)j) - V.sup_flo[2], (dhb - (float)j) - V.sup_fhi[2]) * kx;
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
                } else if (allpos && !ISF) {
                    // every view of the box is in bounds with a strictly positive weight everywhere: plain weighted sums
#pragma unroll
                    for (int j = 0; j < kRV; ++j) {
                        num[j] = fmaf(w[j], val[j], num[j]);
                        den[j] += w[j];
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
                        const int pm = ramp ? -1 : 0;   // bit-select: keeps the exact value of a single ramp contributor
                        last[j] = __int_as_float((__float_as_int(val[j]) & pm) | (__float_as_int(last[j]) & ~pm));
                        wlast[j] = __int_as_float((__float_as_int(we) & pm) | (__float_as_int(wlast[j]) & ~pm));
                    }
                }
            }

            float q[kRV];
#pragma unroll
            for (int j = 0; j < kRV; ++j) {
                float o;
                if (nv == 1 && all_unit && !partial_mask && !ISF) o = num[j];      // a single full view with weight 1
                else if (allpos && !ISF) o = num[j] * __builtin_amdgcn_rcpf(den[j]);
                else {
                    o = num[j] * __builtin_amdgcn_rcpf(den[j]);
                    o = (den[j] == wlast[j]) ? last[j] : o;                        // single ramp contributor: exact value
                }
                if (!(fabsf(o) <= 3.4028234e38f)) o = 0.f;
                q[j] = o;
            }
            if (row_ok && nvx > 0) store8<TOut>(out_ptr(out, zc, yc, xq, oy, ox, tz, ty, tx), q, nvx);
        }
    }
}

// One workgroup = 16 consecutive complete rows of one plane of one strip (4 wavefronts x 4 rows).
template <typename TIn, typename TOut, int NVMAX, bool FRAC>
__global__ __launch_bounds__(256) void fuse_rows_kernel(const TrView* __restrict__ views, const Strip* __restrict__ strips,
                                                        const Cell* __restrict__ cells, const RowItem* __restrict__ items, int nitems,
                                                        TOut* __restrict__ out, int oy, int ox, int tz, int ty, int tx) {
    __shared__ float s_strip[kWG][9 * 64];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    // XCD-aware order: workgroups are dealt round-robin to the 8 XCDs (each with its own L2), so XCD k takes the k-th
    // contiguous eighth of the item list: consecutive row groups of a plane -- which share tap rows when the offsets
    // are fractional -- meet in ONE L2.
    const int wg = (blockIdx.x & 7) * (gridDim.x >> 3) + (blockIdx.x >> 3);
    if (wg >= nitems) return;
    const RowItem it = items[wg];
    const Strip S = strips[it.strip];
    const int zc = it.z;
    const int yw = it.y + kWR * wave;
    if (yw >= S.y1) return;
    const int y1 = min(yw + kWR, S.y1);
    float* strip = &s_strip[wave][lane];
    for (int ci = 0; ci < S.ncells; ++ci) {
        const Cell& C = cells[S.cell0 + ci];
        const int x0 = C.x0, x1 = C.x1, nv = C.nv_lxb_cls & 0xff, lxb = (C.nv_lxb_cls >> 8) & 7, cls = C.nv_lxb_cls >> 16;
        if (cls == 0) zero_cell<TOut>(zc, yw, y1, x0, x1, lxb, lane, out, oy, ox, tz, ty, tx);
        else if (cls == 1) copy_cell<TIn, TOut, FRAC>(views[C.ids[0]], zc, yw, y1, x0, x1, lxb, lane, strip, out, oy, ox, tz, ty, tx);
        else blend_cell<TIn, TOut, NVMAX, FRAC>(views, C, nv, C.masks, zc, yw, y1, x0, x1, lxb, lane, strip, out, oy, ox, tz, ty, tx);
    }
}

template <typename T>
void launch_rows(int nvclass, bool frac, int nblocks, int wpg, hipStream_t s, const TrView* views, const Strip* strips, const Cell* cells,
                 const RowItem* items, int nitems, void* out, int oy, int ox, int tz, int ty, int tx) {
    const dim3 grid(nblocks), block(64 * wpg);
#define MVS_ROWS(NV, FR) hipLaunchKernelGGL((fuse_rows_kernel<T, T, NV, FR>), grid, block, 0, s, views, strips, cells, items, nitems, (T*)out, oy, ox, tz, ty, tx)
    if (nvclass == 0) { if (frac) MVS_ROWS(2, true); else MVS_ROWS(2, false); }
    else if (nvclass == 1) { if (frac) MVS_ROWS(4, true); else MVS_ROWS(4, false); }
    else { if (frac) MVS_ROWS(8, true); else MVS_ROWS(8, false); }
#undef MVS_ROWS
}

}  // namespace mvsrows

// per-dtype launchers (one translation unit each: mvs_fuse_rows_u8.hip, _u16.hip, _f32.hip)
void mvs_launch_rows_u8(int nvclass, bool frac, int nblocks, int wpg, hipStream_t s, const TrView* views, const mvsrows::Strip* strips,
                        const mvsrows::Cell* cells, const mvsrows::RowItem* items, int nitems, void* out, int oy, int ox, int tz, int ty, int tx);
void mvs_launch_rows_u16(int nvclass, bool frac, int nblocks, int wpg, hipStream_t s, const TrView* views, const mvsrows::Strip* strips,
                         const mvsrows::Cell* cells, const mvsrows::RowItem* items, int nitems, void* out, int oy, int ox, int tz, int ty, int tx);
void mvs_launch_rows_f32(int nvclass, bool frac, int nblocks, int wpg, hipStream_t s, const TrView* views, const mvsrows::Strip* strips,
                         const mvsrows::Cell* cells, const mvsrows::RowItem* items, int nitems, void* out, int oy, int ox, int tz, int ty, int tx);
