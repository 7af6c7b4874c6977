// EXPERIMENT RECORD (round 3) -- not built, not part of libmvs_hip.so.  The row-walking fuse kernel measured in
// profiles/round3_summary.md section 3(a): 10.4 / 11.1 ms (exact / jittered north-star mosaic) against the region kernels'
// 10.3 / 10.5-11.0; every step forced onto its all-unit path: 8.7 / 9.6 ms -- the ceiling of its access pattern.  Kept so that the
// numbers can be reproduced: add it to csrc/Makefile and call mvs_fuse_walk() before mvs_fuse_regions() in mvs_fuse_chunk.
// mvs_fuse_walk.hip -- row-walking translation fast path of mvs_fuse_chunk (uint16 / uint8 tiles, integer offsets,
// weighted-average fusion with blending weights; gfx950).   reference: fusion/_core.py:1608-1713 (fuse_np body),
// weights.py:391-511 (blending weights), weights.py:325-345 (normalisation), fusion/_core.py:61-94 (weighted average).
//
// Why another kernel.  Measured on the north-star mosaic (profiles/round2_summary.md): kernels that own BOXES of the
// decomposition (mvs_fuse_region.hip) are limited by their access pattern -- 820-byte row pieces, cache lines shared between
// boxes: 3.0 TB/s where whole rows stream at 5+ -- and kernels that own whole ROWS 512 voxels at a time are limited by their
// instruction stream, because every 512-voxel unit of a 20 %-overlap grid holds a view border, so the general weighted path
// (per-voxel profile, ramp polynomial, per-voxel normalisation) runs for every unit although only ~35 voxels next to each
// border need it.  This kernel keeps the row-contiguous memory structure and makes the expensive path rare:
//
//   * a wavefront owns 8 consecutive output rows of one plane and WALKS them along x in steps of 64 voxels: lane (r, s) =
//     (lane >> 3, lane & 7) holds 8 consecutive voxels of row r, so one load instruction fetches 128 contiguous bytes of
//     each of the 8 rows, one store writes them; consecutive steps continue every row where the last one stopped, and
//     consecutive wavefronts take the next 8 rows: every tile row and every output row is touched once, front to back;
//   * rows are grouped into STRIPS (cells of the z / y cuts at every view border -- no clustering, a strip may be one row
//     thin) inside which the set of views per x step is constant: a strip is a list of SEGMENTS (step range + view list),
//     walked by a scalar loop; all per-view quantities of a segment (row nodes G1 / dG of the blend profile, buffer
//     offsets, support distances) are computed once per segment and row, not per voxel;
//   * per step and view, two evaluations of the profile (it is concave along x, so its minimum over a lane's 8 voxels sits at
//     an end) classify the lane as zero / unit / flat (a constant weight < 1 set by the row: the row lies in the ramp of a
//     z / y border) or ramp; ballots make the decision uniform per wavefront:
//       all views unit          -> integer arithmetic on the packed voxels (1 view: the loaded dwords are stored as they
//                                  are; 2 / 4 / 8 views: exact floor of the mean, which is what the reference's float32
//                                  sum of exactly representable terms truncates to), 3 / 5 / 6 / 7 views: float mean
//       constant per lane       -> one weight per lane and view, one reciprocal per lane
//       some lane in a ramp     -> per-voxel profile and ramp polynomial for THAT view only (8 rows x a 10-40 voxel ramp fill
//                                  most of the wavefront), per-voxel normalisation
//   * exactness rule "one contributor yields its value" (w v / w == v in the reference): the quotient is within 1e-6 of
//     an integer then and is rounded to it; the same rounding makes a weighted mean of EQUAL values exact (the reference
//     lands one count low on about half of those, within its own float32 noise).
//
// Algorithmic bytes per launch: every input voxel that reaches into the chunk once + every output voxel once.
#include "mvs_fuse_tr.h"

#include <algorithm>
#include <chrono>
#include <cstdlib>
#include <cstring>
#include <vector>

int mvs_fuse_walk(MvsContext* c, const TrView* htr, const TrView* dtr, int n_views, int dtype, void* dout, const int64_t os[3],
                  const int64_t trim[3], bool* done);
double mvs_walk_last_plan_ms(MvsContext* c);

namespace {

constexpr int kRV = 8;            // voxels per lane
constexpr int kLPR = 8;           // lanes per row
constexpr int kRows = 8;          // rows per wavefront
constexpr int kStep = kLPR * kRV; // voxels per row and step
constexpr int kMaxV = 8;          // views per segment

struct WSeg { int s0, s1, nv, flags; int ids[kMaxV]; };          // steps [s0, s1) of a strip see exactly the views ids[0 .. nv); flags bit 0:
                                                                  // all weights are 1 there, bit 1: a window may straddle the ends of a slab
static_assert(sizeof(WSeg) == 48, "WSeg layout");
struct WStrip { int z0, z1, y0, y1, seg0, nseg, wave0, nyg; };  // wave0: first wavefront of the strip; nyg: row groups per plane
static_assert(sizeof(WStrip) == 32, "WStrip layout");

struct WalkParams {
    const TrView* views;
    const WStrip* strips;
    const WSeg* segs;
    int nstrips, nwaves;
    void* out;
    int oy, ox, tz, ty, tx;
};

typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2_t __attribute__((ext_vector_type(2)));

// The plan tables and the view records are read through the CONSTANT address space: the address is uniform, so the loads
// become scalar loads and the values live in scalar registers.  (Through a generic pointer the compiler must assume that the
// kernel's own stores may alias them and falls back to per-lane vector loads: 88 of them and 560 v_readfirstlane in the first
// build of this kernel.)
template <typename S>
__device__ __forceinline__ S load_uniform(const S* p) {
    static_assert(sizeof(S) % 4 == 0, "dword-sized records");
    union { S s; unsigned int w[sizeof(S) / 4]; } u;
    const __attribute__((address_space(4))) unsigned int* src = (const __attribute__((address_space(4))) unsigned int*)(unsigned long long)p;
#pragma unroll
    for (unsigned k = 0; k < sizeof(S) / 4; ++k) u.w[k] = src[k];
    return u.s;
}

// ---- element type traits: a lane's 8 voxels as raw dwords ---------------------------------------------------------------
template <typename T> struct Px;
template <> struct Px<unsigned short> {
    static constexpr int ES = 2;               // bytes per voxel
    static __device__ __forceinline__ void load(__amdgpu_buffer_rsrc_t r, int vo, int so, unsigned int (&w)[4]) {
        const u32x4_t a = __builtin_amdgcn_raw_buffer_load_b128(r, vo, so, 0);
        w[0] = a.x; w[1] = a.y; w[2] = a.z; w[3] = a.w;
    }
    static __device__ __forceinline__ unsigned int load1(__amdgpu_buffer_rsrc_t r, int o) {
        return (unsigned short)__builtin_amdgcn_raw_buffer_load_b16(r, o, 0, 0);
    }
    static __device__ __forceinline__ void pack(const unsigned int (&e)[kRV], unsigned int (&w)[4]) {
#pragma unroll
        for (int k = 0; k < 4; ++k) w[k] = e[2 * k] | (e[2 * k + 1] << 16);
    }
    static __device__ __forceinline__ void decode(const unsigned int (&w)[4], float (&v)[kRV]) {
#pragma unroll
        for (int k = 0; k < 4; ++k) { v[2 * k] = (float)(w[k] & 0xffffu); v[2 * k + 1] = (float)(w[k] >> 16); }
    }
    // sums of the even / odd halves: lo[k] += w & 0xffff, hi[k] += w >> 16
    static __device__ __forceinline__ void add_split(const unsigned int (&w)[4], unsigned int (&lo)[4], unsigned int (&hi)[4]) {
#pragma unroll
        for (int k = 0; k < 4; ++k) { lo[k] += w[k] & 0xffffu; hi[k] += w[k] >> 16; }
    }
    static __device__ __forceinline__ void join_shift(const unsigned int (&lo)[4], const unsigned int (&hi)[4], int sh, unsigned int (&w)[4]) {
#pragma unroll
        for (int k = 0; k < 4; ++k) w[k] = (lo[k] >> sh) | ((hi[k] >> sh) << 16);
    }
    static __device__ __forceinline__ void avg2(const unsigned int (&a)[4], const unsigned int (&b)[4], unsigned int (&w)[4]) {
#pragma unroll
        for (int k = 0; k < 4; ++k) w[k] = (a[k] & b[k]) + (((a[k] ^ b[k]) >> 1) & 0x7fff7fffu);   // floor((a + b) / 2) per half
    }
    static __device__ __forceinline__ void store(unsigned short* p, const unsigned int (&w)[4], int nvalid) {
        typedef unsigned int u32x4_a2 __attribute__((ext_vector_type(4), aligned(2)));
        typedef unsigned int u32x2_a2 __attribute__((ext_vector_type(2), aligned(2)));
        typedef unsigned int u32_a2 __attribute__((aligned(2)));
        const unsigned int w0 = w[0], w1 = w[1], w2 = w[2], w3 = w[3];     // (scalars: an indexed array would be spilled to scratch)
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
        const unsigned int m2 = has4 ? w2 : w0, m3 = has4 ? w3 : w1;             // the dwords after the first 0 / 4 voxels
        if (nvalid & 2) *reinterpret_cast<u32_a2*>(p + (has4 ? 4 : 0)) = m2;
        if (nvalid & 1) p[nvalid - 1] = (unsigned short)(((nvalid & 2) ? m3 : m2) & 0xffffu);
    }
};
template <> struct Px<unsigned char> {
    static constexpr int ES = 1;
    static __device__ __forceinline__ void load(__amdgpu_buffer_rsrc_t r, int vo, int so, unsigned int (&w)[4]) {
        const u32x2_t a = __builtin_amdgcn_raw_buffer_load_b64(r, vo, so, 0);
        w[0] = a.x; w[1] = a.y; w[2] = 0; w[3] = 0;
    }
    static __device__ __forceinline__ unsigned int load1(__amdgpu_buffer_rsrc_t r, int o) {
        return (unsigned char)__builtin_amdgcn_raw_buffer_load_b8(r, o, 0, 0);
    }
    static __device__ __forceinline__ void pack(const unsigned int (&e)[kRV], unsigned int (&w)[4]) {
        w[0] = e[0] | (e[1] << 8) | (e[2] << 16) | (e[3] << 24);
        w[1] = e[4] | (e[5] << 8) | (e[6] << 16) | (e[7] << 24);
        w[2] = 0; w[3] = 0;
    }
    static __device__ __forceinline__ void decode(const unsigned int (&w)[4], float (&v)[kRV]) {
#pragma unroll
        for (int k = 0; k < kRV; ++k) v[k] = (float)((w[k >> 2] >> (8 * (k & 3))) & 0xffu);
    }
    // even / odd bytes of each dword as 16-bit fields: lo holds bytes 0, 2; hi bytes 1, 3
    static __device__ __forceinline__ void add_split(const unsigned int (&w)[4], unsigned int (&lo)[4], unsigned int (&hi)[4]) {
#pragma unroll
        for (int k = 0; k < 2; ++k) { lo[k] += w[k] & 0x00ff00ffu; hi[k] += (w[k] >> 8) & 0x00ff00ffu; }
    }
    static __device__ __forceinline__ void join_shift(const unsigned int (&lo)[4], const unsigned int (&hi)[4], int sh, unsigned int (&w)[4]) {
#pragma unroll
        for (int k = 0; k < 2; ++k) w[k] = ((lo[k] >> sh) & 0x00ff00ffu) | (((hi[k] >> sh) & 0x00ff00ffu) << 8);
        w[2] = 0; w[3] = 0;
    }
    static __device__ __forceinline__ void avg2(const unsigned int (&a)[4], const unsigned int (&b)[4], unsigned int (&w)[4]) {
#pragma unroll
        for (int k = 0; k < 2; ++k) w[k] = (a[k] & b[k]) + (((a[k] ^ b[k]) >> 1) & 0x7f7f7f7fu);
        w[2] = 0; w[3] = 0;
    }
    static __device__ __forceinline__ void store(unsigned char* p, const unsigned int (&w)[4], int nvalid) {
        typedef unsigned int u32x2_a1 __attribute__((ext_vector_type(2), aligned(1)));
        const unsigned int w0 = w[0], w1 = w[1];
        if (nvalid >= kRV) {
            u32x2_a1 o;
            o.x = w0; o.y = w1;
            __builtin_nontemporal_store(o, reinterpret_cast<u32x2_a1*>(p));
            return;
        }
#pragma unroll
        for (int j = 0; j < kRV; ++j)
            if (j < nvalid) p[j] = (unsigned char)(((j < 4 ? w0 : w1) >> (8 * (j & 3))) & 0xffu);
    }
};

// blend_ramp_nb of mvs_fuse_tr.h value for value: (c + 1) / 2 as one fma (scaling by 2 is exact) and "x >= 1 -> 1" as
// max(w, 1 + (xc - 1) 2^25) (1 for xc == 1, <= -1 below)
__device__ __forceinline__ float walk_ramp(float x) {
    const float xc = __builtin_amdgcn_fmed3f(x, 0.f, 1.f);
    const float a = xc * kPiHalf;
    const float a2 = a * a;
    float s = fmaf(a2, 1.6059043836821613e-10f, -2.5052108385441720e-08f);
    s = fmaf(s, a2, 2.7557319223985893e-06f);
    s = fmaf(s, a2, -1.9841269841269841e-04f);
    s = fmaf(s, a2, 8.3333333333333333e-03f);
    s = fmaf(s, a2, -1.6666666666666666e-01f);
    s = fmaf(s * a2, a, a);
    const float c = fmaf(2.f, s * s, -1.f);
    const float w = fmaf(c, 0.5f, 0.5f);
    return fmaxf(w, fmaf(xc - 1.f, 0x1p25f, 1.f));
}

// profile value at support distance m (output pixels from the nearer end of the support): the two branches of row_profile
// (mvs_fuse_tr.h) -- the table is concave along x, so the smaller one is the valid one
__device__ __forceinline__ float walk_profile(float m, float k, float G1, float dG) {
    const float u = m * k;
    return fminf(u * G1, fmaf(u - 1.f, dG, G1));
}

// quotient -> output value: NaN (0 / 0: no contributor) and negatives -> 0; `single`: exactly one view contributes (w v / w ==
// v in the reference, weights.py:325-345) -- the quotient is within 1e-6 of that integer and is rounded to it
__device__ __forceinline__ unsigned int walk_quant(float o, bool single) {
    o = fmaxf(o, 0.f);
    o = single ? __builtin_rintf(o) : o;
    return (unsigned int)(int)o;
}

template <typename T>
__device__ __forceinline__ void walk_refetch(__amdgpu_buffer_rsrc_t rs, int o, int nbytes, unsigned int (&raw)[4]) {
    // a window that straddles the first / last bytes of a slab: a vector buffer load that is not entirely in range comes back
    // as 0, so it is fetched element by element (first / last row of a slab only)
    typedef Px<T> X;
    const bool str = (o < 0 && o + kRV * X::ES > 0) || (o < nbytes && o + kRV * X::ES > nbytes);
    if (__any(str)) {
        if (str) {
            unsigned int el[kRV];
#pragma unroll
            for (int j = 0; j < kRV; ++j) el[j] = X::load1(rs, o + j * X::ES);
            X::pack(el, raw);
        }
    }
}

// All-unit segment (host: every view has blend weight exactly 1 on every voxel of these steps, in every row of the strip):
// the result is the plain mean -- integer arithmetic on the packed voxels, no weights, no row nodes.  U steps are requested
// back to back before the first one is consumed, so a wavefront keeps U x NV x 1 KiB in flight.
template <typename T, int NV, int U>
__device__ __forceinline__ void walk_unit(const WalkParams& P, const WSeg& G, int nv, int z, int y, bool row_ok, int lane, T* __restrict__ orow) {
    typedef Px<T> X;
    constexpr int ES = X::ES;
    const int sg = lane & (kLPR - 1);
    const int xl0 = P.tx + G.s0 * kStep + kRV * sg;
    const bool edge = (G.flags & 2) != 0;
    __amdgpu_buffer_rsrc_t rs[NV];
    int vo[NV], nbytes[NV];
#pragma unroll
    for (int v = 0; v < NV; ++v) {
        const TrView V = load_uniform(P.views + G.ids[v < nv ? v : 0]);
        nbytes[v] = (int)V.span * ES;
        rs[v] = __builtin_amdgcn_make_buffer_rsrc((void*)V.data, 0, nbytes[v], 0x00020000);
        vo[v] = ((z + V.io[0]) * V.stride_z + (y + V.io[1]) * V.stride_y + (xl0 + V.io[2])) * ES;
    }
    T* op = orow + (size_t)(xl0 - P.tx);
    const int xend = P.tx + P.ox;
    const float rn = __builtin_amdgcn_rcpf((float)nv);
    for (int s = G.s0; s < G.s1; s += U) {
        unsigned int raw[U][NV][4];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int so = (min(s + u, G.s1 - 1) - G.s0) * (kStep * ES);
#pragma unroll
            for (int v = 0; v < NV; ++v)
                if (v < nv) X::load(rs[v], vo[v] + so, 0, raw[u][v]);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (s + u >= G.s1) continue;
            const int so = (s + u - G.s0) * (kStep * ES);
            const int xl = xl0 + (s + u - G.s0) * kStep;
            const int nvalid = row_ok ? min(max(xend - xl, 0), kRV) : 0;
            if (edge) {
#pragma unroll
                for (int v = 0; v < NV; ++v)
                    if (v < nv) walk_refetch<T>(rs[v], vo[v] + so, nbytes[v], raw[u][v]);
            }
            unsigned int q[4];
            if (NV == 1) {
#pragma unroll
                for (int k = 0; k < 4; ++k) q[k] = raw[u][0][k];
            } else if (NV == 2) {
                X::avg2(raw[u][0], raw[u][NV >= 2 ? 1 : 0], q);
            } else if (nv == 4) {
                unsigned int lo[4] = {0, 0, 0, 0}, hi[4] = {0, 0, 0, 0};
#pragma unroll
                for (int v = 0; v < NV; ++v) X::add_split(raw[u][v], lo, hi);
                X::join_shift(lo, hi, 2, q);
            } else {
                float num[kRV];
#pragma unroll
                for (int j = 0; j < kRV; ++j) num[j] = 0.f;
#pragma unroll
                for (int v = 0; v < NV; ++v)
                    if (v < nv) {
                        float e[kRV];
                        X::decode(raw[u][v], e);
#pragma unroll
                        for (int j = 0; j < kRV; ++j) num[j] += e[j];
                    }
                unsigned int o[kRV];
#pragma unroll
                for (int j = 0; j < kRV; ++j) o[j] = walk_quant(num[j] * rn, false);
                X::pack(o, q);
            }
            if (nvalid > 0) X::store(op + (size_t)(s + u - G.s0) * kStep, q, nvalid);
        }
    }
}

// General segment: steps [G.s0, G.s1) with the views G.ids[0 .. nv), nv <= NV; weights classified per step.
template <typename T, int NV>
__device__ __forceinline__ void walk_segment(const WalkParams& P, const WSeg& G, int nv, int z, int y, bool row_ok, int lane, T* __restrict__ orow) {
    typedef Px<T> X;
    constexpr int ES = X::ES;
    const int sg = lane & (kLPR - 1);
    const int xl0 = P.tx + G.s0 * kStep + kRV * sg;        // chunk index of the lane's first voxel at the segment's first step
    __amdgpu_buffer_rsrc_t rs[NV];
    int vo[NV], nbytes[NV];
    float G1[NV], dG[NV], kx[NV], flo[NV], fhi[NV], dlb[NV], span[NV], wflat[NV];   // (kx, flo, fhi, span: uniform -> scalar registers)
#pragma unroll
    for (int v = 0; v < NV; ++v) {
        const TrView V = load_uniform(P.views + G.ids[v < nv ? v : 0]);
        float g1 = 0.f, g2 = 0.f;
        const bool in = tr_row_nodes(V, z, y, g1, g2);
        G1[v] = in ? g1 : 0.f;
        dG[v] = in ? g2 - g1 : 0.f;
        kx[v] = V.sup_k[2];
        flo[v] = V.sup_flo[2];
        fhi[v] = V.sup_fhi[2];
        dlb[v] = (float)(xl0 - V.sup_ilo[2]);
        span[v] = (float)(V.sup_ihi[2] - V.sup_ilo[2]);        // dh = span - dl: small integers, exact in float
        wflat[v] = walk_ramp(G1[v]);
        nbytes[v] = (int)V.span * ES;
        rs[v] = __builtin_amdgcn_make_buffer_rsrc((void*)V.data, 0, nbytes[v], 0x00020000);
        vo[v] = ((z + V.io[0]) * V.stride_z + (y + V.io[1]) * V.stride_y + (xl0 + V.io[2])) * ES;
    }
    T* op = orow + (size_t)(xl0 - P.tx);
    const int xend = P.tx + P.ox;
    unsigned int rawn[NV][4];
#pragma unroll
    for (int v = 0; v < NV; ++v)
        if (v < nv) X::load(rs[v], vo[v], 0, rawn[v]);
    for (int s = G.s0; s < G.s1; ++s) {
        const int so = (s - G.s0) * (kStep * ES);
        const int xl = xl0 + (s - G.s0) * kStep;
        const int nvalid = row_ok ? min(max(xend - xl, 0), kRV) : 0;
        unsigned int raw[NV][4];
#pragma unroll
        for (int v = 0; v < NV; ++v) {
#pragma unroll
            for (int k = 0; k < 4; ++k) raw[v][k] = rawn[v][k];
        }
        if (s + 1 < G.s1) {           // the next step's voxels are in flight while this one is evaluated
#pragma unroll
            for (int v = 0; v < NV; ++v)
                if (v < nv) X::load(rs[v], vo[v] + so + kStep * ES, 0, rawn[v]);
        }
        // lane classes (two evaluations of the concave profile per view): zero / unit / flat (constant per lane) or ramp
        float wl[NV];
        bool all_unit = true, all_const = true;
        bool v_const[NV];
        const float st = (float)((s - G.s0) * kStep);
#pragma unroll
        for (int v = 0; v < NV; ++v) {
            if (v >= nv) { wl[v] = 0.f; v_const[v] = true; continue; }
            const float dli = dlb[v] + st, dhi = span[v] - dli;
            const float dl0 = dli - flo[v], dh0 = dhi - fhi[v];
            const float dl7 = (dli + 7.f) - flo[v], dh7 = (dhi - 7.f) - fhi[v];
            const float mlo = fminf(fminf(dl0, dh0), fminf(dl7, dh7));
            const float Wlo = walk_profile(mlo, kx[v], G1[v], dG[v]);
            const bool unit = Wlo >= 1.f;
            const bool zero = !(dl7 > 0.f) || !(dh0 > 0.f) || !(G1[v] > 0.f);   // left / right of the support, or the row carries no weight
            const bool flat = (mlo * kx[v] >= 1.f) && (dG[v] == 0.f);           // beyond the first support cell of a flat row
            wl[v] = unit ? 1.f : zero ? 0.f : wflat[v];
            v_const[v] = !__any(!(unit || zero || flat));
            all_unit = all_unit && !__any(!unit);
            all_const = all_const && v_const[v];
        }
#pragma unroll
        for (int v = 0; v < NV; ++v)
            if (v < nv) walk_refetch<T>(rs[v], vo[v] + so, nbytes[v], raw[v]);
        unsigned int q[4];
        if (all_unit) {
            if (nv == 1) {
#pragma unroll
                for (int k = 0; k < 4; ++k) q[k] = raw[0][k];
            } else if (NV >= 2 && nv == 2) {
                X::avg2(raw[0], raw[NV >= 2 ? 1 : 0], q);
            } else if (NV >= 4 && nv == 4) {
                unsigned int lo[4] = {0, 0, 0, 0}, hi[4] = {0, 0, 0, 0};
#pragma unroll
                for (int v = 0; v < NV; ++v) X::add_split(raw[v], lo, hi);
                X::join_shift(lo, hi, 2, q);
            } else {
                float num[kRV];
#pragma unroll
                for (int j = 0; j < kRV; ++j) num[j] = 0.f;
#pragma unroll
                for (int v = 0; v < NV; ++v)
                    if (v < nv) {
                        float e[kRV];
                        X::decode(raw[v], e);
#pragma unroll
                        for (int j = 0; j < kRV; ++j) num[j] += e[j];
                    }
                const float rn = __builtin_amdgcn_rcpf((float)nv);
                unsigned int o[kRV];
#pragma unroll
                for (int j = 0; j < kRV; ++j) o[j] = walk_quant(num[j] * rn, false);
                X::pack(o, q);
            }
        } else if (all_const) {
            // one weight per lane and view
            float num[kRV], den = 0.f, npos = 0.f;
#pragma unroll
            for (int j = 0; j < kRV; ++j) num[j] = 0.f;
#pragma unroll
            for (int v = 0; v < NV; ++v)
                if (v < nv) {
                    float e[kRV];
                    X::decode(raw[v], e);
                    den += wl[v];
                    npos += (wl[v] > 0.f) ? 1.f : 0.f;
#pragma unroll
                    for (int j = 0; j < kRV; ++j) num[j] = fmaf(wl[v], e[j], num[j]);
                }
            const float rd = __builtin_amdgcn_rcpf(den);
            const bool single = npos < 1.5f;
            unsigned int o[kRV];
#pragma unroll
            for (int j = 0; j < kRV; ++j) o[j] = walk_quant(num[j] * rd, single);
            X::pack(o, q);
        } else {
            float num[kRV], den[kRV], cnt[kRV];
#pragma unroll
            for (int j = 0; j < kRV; ++j) { num[j] = 0.f; den[j] = 0.f; cnt[j] = 0.f; }
#pragma unroll
            for (int v = 0; v < NV; ++v)
                if (v < nv) {
                    float e[kRV];
                    X::decode(raw[v], e);
                    if (v_const[v]) {
                        const float pos = (wl[v] > 0.f) ? 1.f : 0.f;
#pragma unroll
                        for (int j = 0; j < kRV; ++j) { num[j] = fmaf(wl[v], e[j], num[j]); den[j] += wl[v]; cnt[j] += pos; }
                    } else {
                        const float dl = dlb[v] + st, dh = span[v] - dl;
#pragma unroll
                        for (int j = 0; j < kRV; ++j) {
                            const float m = fminf((dl + (float)j) - flo[v], (dh - (float)j) - fhi[v]);
                            const float w = walk_ramp(walk_profile(m, kx[v], G1[v], dG[v]));
                            num[j] = fmaf(w, e[j], num[j]);
                            den[j] += w;
                            cnt[j] += __builtin_amdgcn_fmed3f(w * 0x1p100f, 0.f, 1.f);      // weights are 0 or >= 2^-25
                        }
                    }
                }
            unsigned int o[kRV];
#pragma unroll
            for (int j = 0; j < kRV; ++j) o[j] = walk_quant(num[j] * __builtin_amdgcn_rcpf(den[j]), cnt[j] < 1.5f);
            X::pack(o, q);
        }
        if (nvalid > 0) X::store(op + (size_t)(s - G.s0) * kStep, q, nvalid);
    }
}

// Segments with more than 4 views (corners of a 3D tile grid: well under 1 % of the voxels): one view after the other in a
// rolled loop, everything per voxel, the per-view constants re-derived in every step -- compact beats fast here, and the
// unrolled variants keep their register budget (the kernel's budget is that of its largest path).
template <typename T>
__device__ __noinline__ void walk_segment_many(const WalkParams& P, const WSeg& G, int nv, int z, int y, bool row_ok, int lane, T* __restrict__ orow) {
    typedef Px<T> X;
    constexpr int ES = X::ES;
    const int sg = lane & (kLPR - 1);
    const int xend = P.tx + P.ox;
    for (int s = G.s0; s < G.s1; ++s) {
        const int xl = P.tx + s * kStep + kRV * sg;
        const int nvalid = row_ok ? min(max(xend - xl, 0), kRV) : 0;
        float num[kRV], den[kRV], cnt[kRV];
#pragma unroll
        for (int j = 0; j < kRV; ++j) { num[j] = 0.f; den[j] = 0.f; cnt[j] = 0.f; }
#pragma nounroll
        for (int v = 0; v < nv; ++v) {
            const TrView V = load_uniform(P.views + G.ids[v]);
            float g1 = 0.f, g2 = 0.f;
            const bool in = tr_row_nodes(V, z, y, g1, g2);
            const float G1 = in ? g1 : 0.f, dG = in ? g2 - g1 : 0.f;
            const int nbytes = (int)V.span * ES;
            const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)V.data, 0, nbytes, 0x00020000);
            const int o = ((z + V.io[0]) * V.stride_z + (y + V.io[1]) * V.stride_y + (xl + V.io[2])) * ES;
            unsigned int raw[4];
            X::load(rs, o, 0, raw);
            walk_refetch<T>(rs, o, nbytes, raw);
            float e[kRV];
            X::decode(raw, e);
            const float dl = (float)(xl - V.sup_ilo[2]), dh = (float)(V.sup_ihi[2] - xl);
#pragma unroll
            for (int j = 0; j < kRV; ++j) {
                const float m = fminf((dl + (float)j) - V.sup_flo[2], (dh - (float)j) - V.sup_fhi[2]);
                const float w = walk_ramp(walk_profile(m, V.sup_k[2], G1, dG));
                num[j] = fmaf(w, e[j], num[j]);
                den[j] += w;
                cnt[j] += __builtin_amdgcn_fmed3f(w * 0x1p100f, 0.f, 1.f);
            }
        }
        unsigned int o8[kRV], q[4];
#pragma unroll
        for (int j = 0; j < kRV; ++j) o8[j] = walk_quant(num[j] * __builtin_amdgcn_rcpf(den[j]), cnt[j] < 1.5f);
        X::pack(o8, q);
        if (nvalid > 0) X::store(orow + (size_t)(xl - P.tx), q, nvalid);
    }
}

// MAXNV: the largest view count of the strips this launch walks (strips are sorted into three classes on the host, so
// that the register budget of the common rows -- at most two views per step -- is not set by the rare corner rows)
template <typename T, int MAXNV>
__global__ __launch_bounds__(256) void fuse_walk_kernel(WalkParams P) {
    const int lane = threadIdx.x & 63;
    // XCD-aware order: workgroups are dealt round-robin to the 8 XCDs, so XCD k takes the k-th contiguous eighth of the row
    // groups (neighbouring rows share the cache lines at their ends: the output pitch is no multiple of the line size)
    const int wg = (blockIdx.x & 7) * (gridDim.x >> 3) + (blockIdx.x >> 3);
    const int wi = __builtin_amdgcn_readfirstlane(wg * 4 + (int)(threadIdx.x >> 6));
    if (wi >= P.nwaves) return;
    // strip of this wavefront: the last one whose first wavefront is <= wi
    int a = 0, b = P.nstrips - 1;
    while (a < b) {
        const int m = (a + b + 1) >> 1;
        if (load_uniform(P.strips + m).wave0 <= wi) a = m; else b = m - 1;
    }
    const WStrip S = load_uniform(P.strips + a);
    const int k = wi - S.wave0;
    const int z = S.z0 + k / S.nyg, y0 = S.y0 + (k % S.nyg) * kRows;
    const int r = lane >> 3;
    const bool row_ok = y0 + r < S.y1;
    const int y = row_ok ? y0 + r : S.y1 - 1;
    T* orow = (T*)P.out + ((size_t)(z - P.tz) * P.oy + (size_t)(y - P.ty)) * (size_t)P.ox;
    for (int g = 0; g < S.nseg; ++g) {
        const WSeg G = load_uniform(P.segs + S.seg0 + g);
        const int nv = G.nv;
        const bool unit = (G.flags & 1) != 0;
        if (nv == 0) {
            const int sg = lane & (kLPR - 1);
            const unsigned int zq[4] = {0, 0, 0, 0};
            for (int s = G.s0; s < G.s1; ++s) {
                const int xl = s * kStep + kRV * sg;
                const int nvalid = row_ok ? min(max(P.ox - xl, 0), kRV) : 0;
                if (nvalid > 0) Px<T>::store(orow + xl, zq, nvalid);
            }
        }
        else if (nv == 1) { if (unit) walk_unit<T, 1, 4>(P, G, nv, z, y, row_ok, lane, orow); else walk_segment<T, 1>(P, G, nv, z, y, row_ok, lane, orow); }
        else if (nv == 2) { if (unit) walk_unit<T, 2, 4>(P, G, nv, z, y, row_ok, lane, orow); else walk_segment<T, 2>(P, G, nv, z, y, row_ok, lane, orow); }
        else if (MAXNV >= 4 && nv <= 4) { if (unit) walk_unit<T, 4, 2>(P, G, nv, z, y, row_ok, lane, orow); else walk_segment<T, 4>(P, G, nv, z, y, row_ok, lane, orow); }
        else if (MAXNV > 4) walk_segment_many<T>(P, G, nv, z, y, row_ok, lane, orow);
    }
}

struct WalkCache {
    unsigned long long hash = 0;
    bool valid = false, usable = false;
    int nstrips[3] = {0, 0, 0}, nwaves[3] = {0, 0, 0};      // per class: strips whose segments hold <= 2, <= 4, more views
    size_t off_strips[3] = {0, 0, 0}, off_segs = 0;
};
WalkCache g_walk[MVS_MAX_DEVICES * MVS_MAX_LANES];
double g_walk_plan_ms[MVS_MAX_DEVICES * MVS_MAX_LANES];

unsigned long long fnv1a(const void* p, size_t n, unsigned long long h) {
    const unsigned char* b = (const unsigned char*)p;
    for (size_t i = 0; i < n; ++i) { h ^= b[i]; h *= 1099511628211ull; }
    return h;
}

bool view_ok(const TrView& V, int es) {
    if (V.fw[0] > 0.f || V.fw[1] > 0.f || V.fw[2] > 0.f) return false;                 // one tap per voxel only
    if (V.span * es > 0x7fffffffLL || (long long)V.stride_z * es > 0x3fffffffLL) return false;
    if (std::llabs(((long long)V.io[0] * V.stride_z + (long long)V.io[1] * V.stride_y + V.io[2]) * es) > 0x2fffffffLL) return false;
    return true;
}

// x range (chunk indices, inclusive; empty: lo > hi) on which view V has blend weight exactly 1 in EVERY row of the strip
// [z0, z1) x [y0, y1): the profile is concave along every axis line, so over the strip's rows its minimum sits in one of the
// four corner rows; in a row it reaches 1 at support coordinate uA = 1 / G1 (G1 >= 1) or 1 + (1 - G1) / dG (first cell < 1,
// second cell rising), never if the row's nodes stay below 1.  One voxel of slack on either side covers the float rounding of
// the device's evaluation (the kernel re-checks nothing inside an all-unit segment).
void unit_range(const TrView& V, int z0, int z1, int y0, int y1, int* lo, int* hi) {
    float m_need = 0.f;
    for (int k = 0; k < 4; ++k) {
        float G1 = 0.f, G2 = 0.f;
        if (!tr_row_nodes(V, (k & 2) ? z1 - 1 : z0, (k & 1) ? y1 - 1 : y0, G1, G2)) { *lo = 1; *hi = 0; return; }
        const float dG = G2 - G1;
        float uA;
        if (G1 >= 1.f) uA = 1.f / G1;
        else if (dG > 0.f && G1 + dG >= 1.f) uA = 1.f + (1.f - G1) / dG;
        else { *lo = 1; *hi = 0; return; }
        m_need = fmaxf(m_need, uA / V.sup_k[2]);
    }
    const double a = (double)V.sup_ilo[2] + (double)V.sup_flo[2] + (double)m_need;
    const double b = (double)V.sup_ihi[2] - (double)V.sup_fhi[2] - (double)m_need;
    *lo = std::max((int)std::ceil(a) + 1, V.lo[2]);
    *hi = std::min((int)std::floor(b) - 1, V.hi[2]);
    // (verified against the device formula at the two ends)
    while (*lo <= *hi && !(tr_weight_profile(V, z0, y0, *lo) >= 1.f && tr_weight_profile(V, z1 - 1, y1 - 1, *lo) >= 1.f &&
                           tr_weight_profile(V, z0, y1 - 1, *lo) >= 1.f && tr_weight_profile(V, z1 - 1, y0, *lo) >= 1.f)) ++*lo;
    while (*lo <= *hi && !(tr_weight_profile(V, z0, y0, *hi) >= 1.f && tr_weight_profile(V, z1 - 1, y1 - 1, *hi) >= 1.f &&
                           tr_weight_profile(V, z0, y1 - 1, *hi) >= 1.f && tr_weight_profile(V, z1 - 1, y0, *hi) >= 1.f)) --*hi;
}

}  // namespace

double mvs_walk_last_plan_ms(MvsContext* c) { return g_walk_plan_ms[mvs_ctx_index(c->device)]; }

// Sets *done when the chunk was fused here; otherwise (a view with a fractional offset, more than 8 views over one voxel,
// float tiles, a chunk narrower than one step) the caller continues with the other fast paths.
int mvs_fuse_walk(MvsContext* c, const TrView* htr, const TrView* dtr, int n_views, int dtype, void* dout, const int64_t os[3],
                  const int64_t trim[3], bool* done) {
    *done = false;
    g_walk_plan_ms[mvs_ctx_index(c->device)] = 0.0;
    if (dtype != MVS_U16 && dtype != MVS_U8) return MVS_OK;
    const int es = dtype == MVS_U16 ? 2 : 1;
    const int t[3] = {(int)trim[0], (int)trim[1], (int)trim[2]};
    const int o[3] = {(int)os[0], (int)os[1], (int)os[2]};
    if (o[0] < 1 || o[1] < 1 || o[2] < kStep) return MVS_OK;
    if ((long long)o[0] * o[1] * o[2] * es > (1ll << 46)) return MVS_OK;
    unsigned long long h = fnv1a(htr, sizeof(TrView) * (size_t)n_views, 1469598103934665603ull);
    h = fnv1a(t, sizeof(t), h);
    h = fnv1a(o, sizeof(o), h);
    h ^= 0x77616c6bull + (unsigned long long)es;
    WalkCache& wc = g_walk[mvs_ctx_index(c->device)];
    char* dbuf = nullptr;
    if (wc.valid && wc.hash == h && (!wc.usable || c->dev[16].ptr)) {
        if (!wc.usable) return MVS_OK;
        dbuf = (char*)c->dev[16].ptr;
    } else {
        const auto t0 = std::chrono::steady_clock::now();
        wc.valid = false;
        wc.usable = false;
        wc.hash = h;
        std::vector<int> live;
        for (int v = 0; v < n_views; ++v) {
            const TrView& V = htr[v];
            if (V.lo[0] > V.hi[0] || V.lo[1] > V.hi[1] || V.lo[2] > V.hi[2]) continue;
            if (!view_ok(V, es)) { wc.valid = true; return MVS_OK; }
            live.push_back(v);
        }
        // exact cuts of z and y at every view border: inside a cell a view covers all rows or none
        std::vector<int> cut[2];
        for (int d = 0; d < 2; ++d) {
            cut[d].push_back(t[d]);
            cut[d].push_back(t[d] + o[d]);
            for (int v : live) {
                const TrView& V = htr[v];
                auto clampi = [&](int x) { return std::min(std::max(x, t[d]), t[d] + o[d]); };
                cut[d].push_back(clampi(V.lo[d]));
                cut[d].push_back(clampi(V.hi[d] + 1));
                // one more cut inside either border, where the profile along this axis alone has reached 2: rows beyond it
                // have row nodes >= 2 (unless another axis is near its border too), i.e. an x ramp no longer than one step,
                // so that everything between the x ramps of such rows becomes an all-unit segment
                if (getenv("MVS_WALK_CUTS") && (V.wnz > 1 || d == 1)) {
                    const float slope = V.ws[d] * V.sup_k[d];
                    if (slope > 0.f) {
                        const int D = (int)std::ceil(2.0 / (double)slope) + 1;
                        if (2 * D < V.hi[d] - V.lo[d]) { cut[d].push_back(clampi(V.lo[d] + D)); cut[d].push_back(clampi(V.hi[d] + 1 - D)); }
                    }
                }
            }
            std::sort(cut[d].begin(), cut[d].end());
            cut[d].erase(std::unique(cut[d].begin(), cut[d].end()), cut[d].end());
        }
        const int nsteps = (o[2] + kStep - 1) / kStep;
        std::vector<WStrip> strips[3];
        std::vector<WSeg> segs;
        long long nwaves[3] = {0, 0, 0};
        std::vector<int> zv, yv, mark((size_t)nsteps + 1);
        for (size_t iz = 0; iz + 1 < cut[0].size(); ++iz) {
            const int z0 = cut[0][iz], z1 = cut[0][iz + 1];
            zv.clear();
            for (int v : live)
                if (htr[v].lo[0] <= z0 && htr[v].hi[0] >= z1 - 1) zv.push_back(v);
            for (size_t iy = 0; iy + 1 < cut[1].size(); ++iy) {
                const int y0 = cut[1][iy], y1 = cut[1][iy + 1];
                yv.clear();
                for (int v : zv)
                    if (htr[v].lo[1] <= y0 && htr[v].hi[1] >= y1 - 1) yv.push_back(v);
                WStrip S{z0, z1, y0, y1, (int)segs.size(), 0, 0, (y1 - y0 + kRows - 1) / kRows};
                // per view: the steps that hold at least one voxel of its valid box, the steps on which its weight is 1 in
                // every row of the strip, and whether a lane's window can straddle the first / last bytes of its slab here
                std::fill(mark.begin(), mark.end(), 0);
                struct VR { int s0, s1, u0, u1; bool edge; };      // step ranges (inclusive): presence, all-unit
                std::vector<VR> vr(yv.size());
                for (size_t i = 0; i < yv.size(); ++i) {
                    const TrView& V = htr[yv[i]];
                    const int a = std::max(V.lo[2], t[2]), b = std::min(V.hi[2], t[2] + o[2] - 1);
                    if (a > b) { vr[i] = {1, 0, 1, 0, false}; continue; }
                    vr[i].s0 = (a - t[2]) / kStep;
                    vr[i].s1 = (b - t[2]) / kStep;
                    int ulo, uhi;
                    unit_range(V, z0, z1, y0, y1, &ulo, &uhi);
                    // a step is all-unit for the view when all of its 64 voxels lie inside [ulo, uhi]
                    vr[i].u0 = (std::max(ulo, t[2]) - t[2] + kStep - 1) / kStep;
                    vr[i].u1 = (std::min(uhi, t[2] + o[2] - 1) + 1 - t[2]) / kStep - 1;
                    if (uhi >= t[2] + o[2] - 1) vr[i].u1 = nsteps - 1;      // (the last, partial step)
                    if (ulo > uhi) { vr[i].u0 = 1; vr[i].u1 = 0; }
                    // slab rows 0 and n - 1 in z AND y: only there a 16-byte window can reach past the slab's first / last byte
                    const int zf = -V.io[0], zl = V.n[0] - 1 - V.io[0], yf = -V.io[1], yl = V.n[1] - 1 - V.io[1];
                    vr[i].edge = (z0 <= zf && zf < z1 && y0 <= yf && yf < y1) || (z0 <= zl && zl < z1 && y0 <= yl && yl < y1);
                    mark[vr[i].s0] = 1;
                    mark[vr[i].s1 + 1] = 1;
                    if (vr[i].u0 <= vr[i].u1) { mark[vr[i].u0] = 1; mark[vr[i].u1 + 1] = 1; }
                }
                mark[0] = 1;
                mark[nsteps] = 1;
                int s0 = 0, max_nv = 0;
                for (int s = 1; s <= nsteps; ++s) {
                    if (!mark[s]) continue;
                    WSeg G;
                    memset(&G, 0, sizeof(G));
                    G.s0 = s0;
                    G.s1 = s;
                    bool unit = true, edge = false;
                    for (size_t i = 0; i < yv.size(); ++i)
                        if (vr[i].s0 <= s0 && vr[i].s1 >= s - 1) {
                            if (G.nv == kMaxV) { wc.valid = true; return MVS_OK; }       // too many views over one voxel
                            G.ids[G.nv++] = yv[i];
                            unit = unit && vr[i].u0 <= s0 && vr[i].u1 >= s - 1;
                            edge = edge || vr[i].edge;
                        }
                    if (getenv("MVS_WALK_FORCE_UNIT")) unit = true;       // (profiling only: wrong weights)
                    G.flags = ((unit && G.nv > 0 && G.nv <= 4) ? 1 : 0) | (edge ? 2 : 0);
                    max_nv = std::max(max_nv, G.nv);
                    segs.push_back(G);
                    s0 = s;
                }
                S.nseg = (int)segs.size() - S.seg0;
                const int cls = max_nv <= 2 ? 0 : max_nv <= 4 ? 1 : 2;
                S.wave0 = (int)nwaves[cls];
                nwaves[cls] += (long long)(z1 - z0) * S.nyg;
                strips[cls].push_back(S);
            }
        }
        const long long total_waves = nwaves[0] + nwaves[1] + nwaves[2];
        if (total_waves == 0 || total_waves > 0x3fffffffLL || segs.size() > (1u << 26)) { wc.valid = true; return MVS_OK; }
        auto al = [](size_t v) { return (v + 255) / 256 * 256; };
        size_t b_strips[3], total = 0;
        for (int k = 0; k < 3; ++k) { wc.off_strips[k] = total; b_strips[k] = al(strips[k].size() * sizeof(WStrip)); total += b_strips[k]; }
        wc.off_segs = total;
        total += al(segs.size() * sizeof(WSeg));
        dbuf = (char*)mvs_scratch(c, 16, total);
        if (!dbuf) return MVS_ERR_HIP;
        char* hb = (char*)mvs_pinned_slot(c, 1, total);       // slot 0 holds the view parameters still in flight
        if (!hb) return MVS_ERR_HIP;
        for (int k = 0; k < 3; ++k) memcpy(hb + wc.off_strips[k], strips[k].data(), strips[k].size() * sizeof(WStrip));
        memcpy(hb + wc.off_segs, segs.data(), segs.size() * sizeof(WSeg));
        MVS_HIP_TRY(c, hipMemcpyAsync(dbuf, hb, total, hipMemcpyHostToDevice, c->stream));
        mvs_pinned_mark(c, 1);
        for (int k = 0; k < 3; ++k) { wc.nstrips[k] = (int)strips[k].size(); wc.nwaves[k] = (int)nwaves[k]; }
        wc.valid = true;
        wc.usable = true;
        g_walk_plan_ms[mvs_ctx_index(c->device)] = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
        if (getenv("MVS_PLAN_STATS")) {
            double unit_steps = 0, all_steps = 0;
            for (int k = 0; k < 3; ++k)
                for (const WStrip& S : strips[k])
                    for (int g = 0; g < S.nseg; ++g) {
                        const WSeg& G = segs[S.seg0 + g];
                        const double w = (double)(S.z1 - S.z0) * (S.y1 - S.y0) * (G.s1 - G.s0);
                        all_steps += w;
                        if (G.flags & 1) unit_steps += w;
                    }
            fprintf(stderr, "[mvs walk plan] strips %zu / %zu / %zu, segments %zu, wavefronts %lld / %lld / %lld, all-unit share %.3f\n",
                    strips[0].size(), strips[1].size(), strips[2].size(), segs.size(), nwaves[0], nwaves[1], nwaves[2],
                    all_steps > 0 ? unit_steps / all_steps : 0.0);
        }
    }
    WalkParams P;
    P.views = dtr;
    P.segs = (const WSeg*)(dbuf + wc.off_segs);
    P.out = dout;
    P.oy = o[1]; P.ox = o[2];
    P.tz = t[0]; P.ty = t[1]; P.tx = t[2];
    MVS_HIP_TRY(c, hipEventRecord(c->ev_start, c->stream));   // kernel time only (the plan is host work, cached per geometry)
    for (int k = 0; k < 3; ++k) {
        if (!wc.nwaves[k]) continue;
        P.strips = (const WStrip*)(dbuf + wc.off_strips[k]);
        P.nstrips = wc.nstrips[k];
        P.nwaves = wc.nwaves[k];
        const dim3 grid(((wc.nwaves[k] + 3) / 4 + 7) / 8 * 8), block(256);
#define MVS_WK(T) do { if (k == 0) hipLaunchKernelGGL((fuse_walk_kernel<T, 2>), grid, block, 0, c->stream, P); \
                       else if (k == 1) hipLaunchKernelGGL((fuse_walk_kernel<T, 4>), grid, block, 0, c->stream, P); \
                       else hipLaunchKernelGGL((fuse_walk_kernel<T, 8>), grid, block, 0, c->stream, P); } while (0)
        if (dtype == MVS_U16) MVS_WK(unsigned short); else MVS_WK(unsigned char);
#undef MVS_WK
    }
    MVS_HIP_TRY(c, hipGetLastError());
    *done = true;
    return MVS_OK;
}
