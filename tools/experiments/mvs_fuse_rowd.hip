// EXPERIMENT RECORD (round 3) -- not built, not part of libmvs_hip.so.  "Rows for the memory system, dense items for the ramps":
// the fourth fuse-launch design measured this round (profiles/round3_summary.md section 3(e)).  Bit-identical to the class kernels
// on every geometry tried (4x4x4 of 512^3 exact / jittered, 3x3x3 of 300x400x700, 2x3x3 of 301x403x611); 12.3 / 15.0 ms on the
// north-star mosaic (exact / jittered) where the class kernels took 12.05 / 13.1 on the same box, 3.45 instead of 4.9 ms on the
// 3x3x3 grid of wide tiles.  PMC: 5.1 G vector + 1.7 G scalar wavefront instructions per launch (2.4 G row pass, 2.7 G dense
// pass) against the class kernels' 3.7 G + 1.0 G -- the kernel is bound by its instruction stream, not by its access pattern.
// To reproduce: add the file to csrc/Makefile and call mvs_fuse_rowd() before mvs_fuse_regions() in mvs_fuse_chunk.
// mvs_fuse_rowd.hip -- translation fast path of mvs_fuse_chunk for integer tiles at integer offsets (uint16 / uint8,
// weighted-average fusion with blending weights; gfx950): whole ROWS for the memory system, DENSE items for the blend ramps.
//   reference: fusion/_core.py:1608-1713 (fuse_np body), weights.py:391-511 (blending weights), weights.py:325-345
//   (normalisation), fusion/_core.py:61-94 (weighted average).
//
// Three rounds of measurements behind this design (profiles/round2_summary.md 3, 7; profiles/round3_summary.md 3):
//   * kernels that own BOXES (mvs_fuse_region.hip) or 8 rows x 64 voxels per load instruction stream at ~3 TB/s; only load /
//     store instructions that cover a whole kilobyte of ONE row reach 5+ TB/s;
//   * with whole rows every 512-voxel unit of a 20 %-overlap grid holds a view border, and evaluating the general weighted
//     path (per-voxel profile, ramp polynomial, per-voxel normalisation) for all 64 lanes because 5 of them sit in a ramp is
//     what made the row kernels of round 2 instruction-bound.
// Here a wavefront owns 4 consecutive output rows and walks them in steps of 512 voxels (lane = 8 consecutive voxels; every
// load fetches 1 KB of one tile row, every store writes 1 KB of one output row; the next row's loads are in flight while a row
// is evaluated).  Per row, step and view two evaluations of the concave blend profile classify a lane as zero / unit / flat
// (a constant weight set by the row) or RAMP.  Lanes without a ramp finish the row themselves: integer arithmetic on the
// packed voxels when every weight is 1 (copy; exact floor of the mean of 2 / 4 views), one weight per lane and view otherwise.
// A lane with a ramp stores nothing: it drops its packed voxels of all views into an LDS queue as an ITEM (row, lane).  After
// the 4 rows the queue -- typically 20-40 items, the ramp zones of 4 rows -- is evaluated DENSELY: lane i takes item i and
// computes the per-voxel profile, ramp polynomial and normalisation for its 8 voxels and stores their 16 bytes; the expensive
// path runs once per step for a wavefront full of voxels that need it, not once per unit for a handful of lanes.
#include "mvs_fuse_tr.h"

#include <algorithm>
#include <chrono>
#include <cstdlib>
#include <cstring>
#include <vector>

int mvs_fuse_rowd(MvsContext* c, const TrView* htr, const TrView* dtr, int n_views, int dtype, void* dout, const int64_t os[3],
                  const int64_t trim[3], bool* done);
double mvs_rowd_last_plan_ms(MvsContext* c);

namespace {

constexpr int kRV = 8;            // voxels per lane
#ifndef MVS_ROWD_ROWS
#define MVS_ROWD_ROWS 4
#endif
constexpr int kRows = MVS_ROWD_ROWS;   // rows per wavefront (4 or 8)
constexpr int kStep = 64 * kRV;   // voxels per row and step
constexpr int kMaxV = 12;         // views per segment (more than 4: the rolled per-voxel path)
constexpr int kMaxSV = 64;        // views of a strip (their row nodes live in LDS)
constexpr int kItems = 64 * kRows; // queue of ramp items per wavefront and step (every lane of every row: it cannot overflow)

struct RSeg { int s0, s1, nv, nx; int ids[kMaxV]; int lv[kMaxV]; };   // steps [s0, s1) of a strip see the views ids[0 .. nv + nx) (lv: index in the strip's view
                                                                       // list): nv of them in the row pass, nx "extra" ones (few voxels) in the dense pass only
static_assert(sizeof(RSeg) == 112, "RSeg layout");
struct RStrip { int z0, z1, y0, y1, seg0, nseg, wave0, nyg, sv0, nsv, pad0, pad1; };   // sv0: first entry of the strip's view list
static_assert(sizeof(RStrip) == 48, "RStrip layout");

struct RowdParams {
    const TrView* views;
    const RStrip* strips;
    const RSeg* segs;
    const int* svlist;
    int nstrips, nwaves;
    void* out;
    int oy, ox, tz, ty, tx;
    int ablate;       // profiling only: bit 0 drops the dense pass, bit 1 forces the integer path (wrong results)
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
__device__ __forceinline__ float rd_ramp(float x) {
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
__device__ __forceinline__ float rd_profile(float m, float k, float G1, float dG) {
    const float u = m * k;
    return fminf(u * G1, fmaf(u - 1.f, dG, G1));
}

template <typename T>
__device__ __forceinline__ void rd_refetch(__amdgpu_buffer_rsrc_t rs, int o, int nbytes, unsigned int (&raw)[4]) {
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


__device__ __forceinline__ unsigned int rd_quant(float num, float den, float last, float wlast) {
    // same arithmetic as the class kernels (mvs_fuse_region.hip): quotient by the hardware reciprocal; a single ramp contributor
    // (the float32 weight sum IS its weight: w v / w == v in the reference, weights.py:325-345) yields its value; 0 / 0 -> 0
    float o = num * __builtin_amdgcn_rcpf(den);
    o = (den == wlast) ? last : o;
    if (!(fabsf(o) <= 3.4028234e38f)) o = 0.f;
    return (unsigned int)(int)o;
}

// ---- LDS of one wavefront ------------------------------------------------------------------------------------------------
struct alignas(16) RowdLds {
    float4 nodes[kMaxSV * kRows];          // view i of the strip, row r: (G1, dG, ramp(G1), 1 if the row carries no weight) at [i * kRows + r]
    unsigned int meta[kItems];             // the queue of ramp items of one step: row | lane << 3
};

template <int NV>
struct SegViews {      // per-view constants of a segment (uniform: scalar registers)
    float kx[NV], flo[NV], fhi[NV], span[NV];
    float dls[NV];     // (float)(first voxel of the segment's first step - support start): small integers, exact
    int xs0;            // chunk index of the first voxel of the segment's first step
    int xa[NV], xb[NV]; // first voxel of the segment's first step - first valid voxel, last valid voxel - that first voxel
    int lv[NV];
    __amdgpu_buffer_rsrc_t rs[NV];
    int vb[NV];         // byte offset of (row 0, first voxel of the segment's first step) in the view's slab
    int pitch[NV], nbytes[NV];
};

// One view and the 8 voxels of one lane, all rows: the smaller support distance of the two ends (the blend profile is concave
// along x, so its minimum over the 8 voxels sits at an end), "all 8 lie outside the support or the valid box", "cut by the border
// of the valid box", "beyond the first support cell".
struct LaneGeo { float mlo; bool zg, part, fl1; };
__device__ __forceinline__ LaneGeo lane_geo(float dl, float span, float flo, float fhi, float kx, int xa, int xb) {
    // dl: (float)(first voxel - support start); voxel j lies in the valid box when -xa <= j <= xb
    const float dh = span - dl;
    const float dl0 = dl - flo, dh0 = dh - fhi;
    const float dl7 = (dl + 7.f) - flo, dh7 = (dh - 7.f) - fhi;
    LaneGeo g;
    g.mlo = fminf(fminf(dl0, dh0), fminf(dl7, dh7));
    const bool vout = (xa < -(kRV - 1)) || (xb < 0);
    g.zg = !(dl7 > 0.f) || !(dh0 > 0.f) || vout;
    g.part = !vout && (xa < 0 || xb < kRV - 1);
    g.fl1 = g.mlo * kx >= 1.f;
    return g;
}
// ... and one row (n4: its nodes): zero / unit / flat -- one weight wl for the 8 voxels -- or ramp
__device__ __forceinline__ void row_class(const LaneGeo& g, float kx, const float4& n4, bool& unit, bool& zero, bool& ramp, float& wl) {
    const float Wlo = rd_profile(g.mlo, kx, n4.x, n4.y);
    unit = Wlo >= 1.f;
    zero = g.zg || (n4.w != 0.f);
    const bool flat = g.fl1 && (n4.y == 0.f);
    unit = unit && !zero;
    ramp = !zero && (g.part || !(unit || flat));
    wl = zero ? 0.f : unit ? 1.f : n4.z;
}

struct Acc { float num[kRV], den[kRV], last[kRV], wlast[kRV]; };
// per-voxel weights of one view (same arithmetic as the class kernels: one rounding per distance, weight -> ramp polynomial)
__device__ __forceinline__ void acc_voxels(Acc& A, const float (&e)[kRV], float dl, float span, float flo, float fhi, float kx, float G1, float dG, int xa,
                                           int xb) {
    const float dh = span - dl;
#pragma unroll
    for (int j = 0; j < kRV; ++j) {
        const float mm = fminf((dl + (float)j) - flo, (dh - (float)j) - fhi);
        float w = rd_ramp(rd_profile(mm, kx, G1, dG));
        w = (xa >= -j && xb >= j) ? w : 0.f;
        A.num[j] = fmaf(w, e[j], A.num[j]);
        A.den[j] += w;
        const bool rp = (w > 0.f) && (w < 1.f);
        A.last[j] = rp ? e[j] : A.last[j];
        A.wlast[j] = rp ? w : A.wlast[j];
    }
}
// one weight for the lane's 8 voxels
__device__ __forceinline__ void acc_const(Acc& A, const float (&e)[kRV], float w) {
    const bool rp = (w > 0.f) && (w < 1.f);
#pragma unroll
    for (int j = 0; j < kRV; ++j) {
        A.num[j] = fmaf(w, e[j], A.num[j]);
        A.den[j] += w;
        A.last[j] = rp ? e[j] : A.last[j];
        A.wlast[j] = rp ? w : A.wlast[j];
    }
}

// The queue of ramp items of one step, densely: lane i evaluates item i = (row, lane of the row pass) -- per-voxel profile, ramp
// polynomial and normalisation of its 8 voxels -- and stores their 16 bytes.  A view is evaluated per voxel only when some item of
// the pass has a ramp of THAT view.
template <typename T, int NV>
__device__ __forceinline__ void rowd_dense(const RowdParams& P, RowdLds& L, int nitems, int nv, const SegViews<NV>& C, int sti, int lane,
                                           T* __restrict__ orow0) {
    // sti: voxels from the segment's first step to this one
    typedef Px<T> X;
    constexpr int ES = X::ES;
    const int xend = P.tx + P.ox;
    const float st = (float)sti;
    const int xs = C.xs0 + sti;
    for (int b = 0; b < nitems; b += 64) {
        const int i = b + lane;
        const bool act = i < nitems;
        const unsigned int m = L.meta[act ? i : b];
        const int r = (int)(m & 7u), ln = (int)(m >> 3);
        const int xq = xs + kRV * ln;
        Acc A;
#pragma unroll
        for (int j = 0; j < kRV; ++j) { A.num[j] = 0.f; A.den[j] = 0.f; A.last[j] = 0.f; A.wlast[j] = 0.f; }
#pragma unroll
        for (int v = 0; v < NV; ++v) {
            if (v >= nv) continue;
            // the item's voxels again (the row pass had them a moment ago: L1 / L2 hits)
            unsigned int raw[4];
            const int o = C.vb[v] + (sti + kRV * ln) * ES + r * C.pitch[v];
            X::load(C.rs[v], o, 0, raw);
            rd_refetch<T>(C.rs[v], o, C.nbytes[v], raw);      // (a window cut by the slab's first / last byte is cut by the valid box too: always an item)
            float e[kRV];
            X::decode(raw, e);
            const float4 n4 = L.nodes[C.lv[v] * kRows + r];
            const float dl = (C.dls[v] + st) + (float)(kRV * ln);
            const int xa = C.xa[v] + sti + kRV * ln, xb = C.xb[v] - sti - kRV * ln;
            const LaneGeo g = lane_geo(dl, C.span[v], C.flo[v], C.fhi[v], C.kx[v], xa, xb);
            bool unit, zero, ramp;
            float wl;
            row_class(g, C.kx[v], n4, unit, zero, ramp, wl);
            if (__any(act && ramp)) acc_voxels(A, e, dl, C.span[v], C.flo[v], C.fhi[v], C.kx[v], n4.x, n4.y, xa, xb);
            else acc_const(A, e, wl);
        }
        unsigned int o8[kRV], q[4];
#pragma unroll
        for (int j = 0; j < kRV; ++j) o8[j] = rd_quant(A.num[j], A.den[j], A.last[j], A.wlast[j]);
        X::pack(o8, q);
        if (act) X::store(orow0 + (size_t)r * (size_t)P.ox + (size_t)(xq - P.tx), q, min(xend - xq, kRV));
    }
}

// The same for the items at the BACK of the queue (meta[kItems - 1 - i]) with a rolled loop over ALL views of the segment, in
// ascending view order (the accumulation order of the reference; gp->ids holds the nv row-pass views and the nx extra ones, each
// group ascending): lanes that see an extra view, and segments with more than 4 views of comparable weight.
template <typename T>
__device__ __forceinline__ void rowd_dense_rolled(const RowdParams& P, RowdLds& L, int nitems, const RSeg* gp, int nv, int nx, int xs, int z, int y0, int lane,
                                               T* __restrict__ orow0) {
    typedef Px<T> X;
    constexpr int ES = X::ES;
    const int xend = P.tx + P.ox;
    for (int b = 0; b < nitems; b += 64) {
        const int i = b + lane;
        const bool act = i < nitems;
        const unsigned int m = L.meta[kItems - 1 - (act ? i : b)];
        const int r = (int)(m & 7u), ln = (int)(m >> 3);
        const int xq = xs + kRV * ln;
        Acc A;
#pragma unroll
        for (int j = 0; j < kRV; ++j) { A.num[j] = 0.f; A.den[j] = 0.f; A.last[j] = 0.f; A.wlast[j] = 0.f; }
        int ia = 0, ib = nv;
        const int n = nv + nx;
        while (ia < nv || ib < n) {
            const int ida = ia < nv ? load_uniform(gp->ids + ia) : 0x7fffffff;
            const int idb = ib < n ? load_uniform(gp->ids + ib) : 0x7fffffff;
            const int x = (ida < idb) ? ia : ib;
            if (ida < idb) ++ia; else ++ib;
            const TrView V = load_uniform(P.views + min(ida, idb));
            const int xa = xq - V.lo[2], xb = V.hi[2] - xq;
            if (!__any(act && !((xa < -(kRV - 1)) || (xb < 0)))) continue;      // no item of this pass inside the view's valid box
            const float4 n4 = L.nodes[load_uniform(gp->lv + x) * kRows + r];
            const int nbytes = (int)V.span * ES;
            const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)V.data, 0, nbytes, 0x00020000);
            const int o = ((z + V.io[0]) * V.stride_z + (y0 + r + V.io[1]) * V.stride_y + (xq + V.io[2])) * ES;
            unsigned int raw[4];
            X::load(rs, o, 0, raw);
            rd_refetch<T>(rs, o, nbytes, raw);
            float e[kRV];
            X::decode(raw, e);
            acc_voxels(A, e, (float)(xq - V.sup_ilo[2]), (float)(V.sup_ihi[2] - V.sup_ilo[2]), V.sup_flo[2], V.sup_fhi[2], V.sup_k[2], n4.x, n4.y, xa, xb);
        }
        unsigned int o8[kRV], q[4];
#pragma unroll
        for (int j = 0; j < kRV; ++j) o8[j] = rd_quant(A.num[j], A.den[j], A.last[j], A.wlast[j]);
        X::pack(o8, q);
        if (act) X::store(orow0 + (size_t)r * (size_t)P.ox + (size_t)(xq - P.tx), q, min(xend - xq, kRV));
    }
}

// Steps [G.s0, G.s1) of the wavefront's rows with the unrolled views G.ids[0 .. nv), nv <= NV <= 4, and G.nx extra views.
template <typename T, int NV>
__device__ __forceinline__ void rowd_segment(const RowdParams& P, const RSeg& G, const RSeg* gp, int nv, int z, int y0, int nrows, int lane,
                                             T* __restrict__ orow0, RowdLds& L) {
    typedef Px<T> X;
    constexpr int ES = X::ES;
    constexpr int kOut = 0x7f000000;      // a byte offset beyond every slab: the lane's load returns 0 without touching memory
    const int xs0 = P.tx + G.s0 * kStep;
    const int xl0 = xs0 + kRV * lane;
    const int xend = P.tx + P.ox;
    int vo[NV];
    SegViews<NV> C;
    C.xs0 = xs0;
    float dlb[NV];
#pragma unroll
    for (int v = 0; v < NV; ++v) {
        const int vv = v < nv ? v : 0;
        const TrView V = load_uniform(P.views + G.ids[vv]);
        C.kx[v] = V.sup_k[2];
        C.flo[v] = V.sup_flo[2];
        C.fhi[v] = V.sup_fhi[2];
        C.span[v] = (float)(V.sup_ihi[2] - V.sup_ilo[2]);       // dh = span - dl: small integers, exact in float
        C.dls[v] = (float)(xs0 - V.sup_ilo[2]);
        C.lv[v] = G.lv[vv];
        C.xa[v] = xs0 - V.lo[2];
        C.xb[v] = V.hi[2] - xs0;
        dlb[v] = (float)(xl0 - V.sup_ilo[2]);
        C.nbytes[v] = (int)V.span * ES;
        C.rs[v] = __builtin_amdgcn_make_buffer_rsrc((void*)V.data, 0, C.nbytes[v], 0x00020000);
        C.vb[v] = ((z + V.io[0]) * V.stride_z + (y0 + V.io[1]) * V.stride_y + (xs0 + V.io[2])) * ES;
        vo[v] = C.vb[v] + kRV * lane * ES;
        C.pitch[v] = V.stride_y * ES;
    }
    // lanes that hold a voxel of an extra view's valid box become items whatever the unrolled views say: bit k = step G.s0 + k
    unsigned int forced = 0u;
#pragma nounroll
    for (int x = nv; x < nv + G.nx; ++x) {
        const int id = load_uniform(gp->ids + x);
        const int lo = load_uniform(&P.views[id].lo[2]), hi = load_uniform(&P.views[id].hi[2]);
        for (int k = 0; k < G.s1 - G.s0 && k < 32; ++k) {
            const int xl = xl0 + k * kStep;
            forced |= (xl + kRV - 1 >= lo && xl <= hi) ? (1u << k) : 0u;
        }
    }
    // lanes whose 8 voxels lie outside a view's valid box request nothing
    auto lane_off = [&](int v, int k) -> int {
        const int xa = C.xa[v] + k * kStep + kRV * lane, xb = C.xb[v] - k * kStep - kRV * lane;
        return ((xa < -(kRV - 1)) || (xb < 0)) ? kOut : vo[v] + k * (kStep * ES);
    };
    // a ring of row buffers: row r of the next step is requested as soon as row r of this one has been evaluated, so three rows
    // (kilobytes per view) are in flight behind the one being worked on
    unsigned int buf[kRows][NV][4];
    int von[NV];
#pragma unroll
    for (int v = 0; v < NV; ++v) von[v] = lane_off(v, 0);
#pragma unroll
    for (int r = 0; r < kRows; ++r) {
        const int rr = min(r, nrows - 1);
#pragma unroll
        for (int v = 0; v < NV; ++v)
            if (v < nv) X::load(C.rs[v], von[v] + rr * C.pitch[v], 0, buf[r][v]);
    }
    for (int s = G.s0; s < G.s1; ++s) {
        const int k = s - G.s0;
        const int xl = xl0 + k * kStep;
        const int nvalid = min(max(xend - xl, 0), kRV);
        const float st = (float)(k * kStep);
        LaneGeo geo[NV];
#pragma unroll
        for (int v = 0; v < NV; ++v) {
            geo[v] = lane_geo(dlb[v] + st, C.span[v], C.flo[v], C.fhi[v], C.kx[v], C.xa[v] + k * kStep + kRV * lane, C.xb[v] - k * kStep - kRV * lane);
            von[v] = lane_off(v, k + 1);
        }
        const bool force = ((forced >> (k & 31)) & 1u) != 0u;
        int nitems = 0, nforced = 0;
#pragma unroll
        for (int r = 0; r < kRows; ++r) {
            unsigned int raw[NV][4];
#pragma unroll
            for (int v = 0; v < NV; ++v) {
#pragma unroll
                for (int q = 0; q < 4; ++q) raw[v][q] = buf[r][v][q];
            }
            const int rr = min(r, nrows - 1);
            if (s + 1 < G.s1) {
#pragma unroll
                for (int v = 0; v < NV; ++v)
                    if (v < nv) X::load(C.rs[v], von[v] + rr * C.pitch[v], 0, buf[r][v]);
            }
            // lane classes: per view zero / unit / flat (one weight for the lane's 8 voxels) or ramp
            float wl[NV];
            unsigned int mu[NV];
            bool ramp = false, bin = true;
            int cnt = 0;
#pragma unroll
            for (int v = 0; v < NV; ++v) {
                if (v >= nv) { wl[v] = 0.f; mu[v] = 0u; continue; }
                const float4 n4 = L.nodes[C.lv[v] * kRows + r];      // (uniform address: a broadcast read)
                bool unit, zero, rp;
                row_class(geo[v], C.kx[v], n4, unit, zero, rp, wl[v]);
                mu[v] = unit ? 0xffffffffu : 0u;
                ramp = ramp || rp;
                bin = bin && (unit || zero);
                cnt += unit ? 1 : 0;
            }
            const bool live = (r < nrows) && (nvalid > 0);
            const bool frc = force && live;
            ramp = ramp && live && !frc;
            const unsigned long long rmask = __ballot(ramp), fmask = __ballot(frc);
            const bool store_it = live && !ramp && !frc;
            unsigned int q[4];
            if ((P.ablate & 2) || !__any(store_it && (!bin || cnt == 3))) {
                // every lane that stores has weights 0 / 1 only: integer arithmetic on the packed voxels (exact floor of the mean
                // of 1 / 2 / 4 values -- what the float32 sum of exactly representable terms truncates to)
                if (NV == 1) {
#pragma unroll
                    for (int d = 0; d < 4; ++d) q[d] = raw[0][d] & mu[0];
                } else if (NV == 2) {
                    unsigned int a2[4];
                    X::avg2(raw[0], raw[NV - 1], a2);
#pragma unroll
                    for (int d = 0; d < 4; ++d) q[d] = (cnt == 2) ? a2[d] : ((raw[0][d] & mu[0]) | (raw[NV - 1][d] & mu[NV - 1]));
                } else {
                    unsigned int lo[4] = {0, 0, 0, 0}, hi[4] = {0, 0, 0, 0};
#pragma unroll
                    for (int v = 0; v < NV; ++v) {
                        unsigned int mk[4];
#pragma unroll
                        for (int d = 0; d < 4; ++d) mk[d] = raw[v][d] & mu[v];
                        X::add_split(mk, lo, hi);
                    }
                    X::join_shift(lo, hi, cnt >> 1, q);      // cnt 1 / 2 / 4 -> shift 0 / 1 / 2
                }
            } else {
                // one weight per lane and view
                float num[kRV], den = 0.f, lw = 0.f;
                int vl = 0;
#pragma unroll
                for (int j = 0; j < kRV; ++j) num[j] = 0.f;
#pragma unroll
                for (int v = 0; v < NV; ++v)
                    if (v < nv) {
                        float e[kRV];
                        X::decode(raw[v], e);
                        den += wl[v];
                        const bool rp = (wl[v] > 0.f) && (wl[v] < 1.f);
                        lw = rp ? wl[v] : lw;
                        vl = rp ? v : vl;
#pragma unroll
                        for (int j = 0; j < kRV; ++j) num[j] = fmaf(wl[v], e[j], num[j]);
                    }
                const float rd = (den > 0.f) ? __builtin_amdgcn_rcpf(den) : 0.f;
                unsigned int o[kRV];
#pragma unroll
                for (int j = 0; j < kRV; ++j) o[j] = (unsigned int)(int)(num[j] * rd);
                X::pack(o, q);
                const bool one = (den == lw) && (lw > 0.f);      // a single flat contributor: its voxels as they are
                if (__any(one)) {
#pragma unroll
                    for (int v = 0; v < NV; ++v) {
#pragma unroll
                        for (int d = 0; d < 4; ++d) q[d] = (one && vl == v) ? raw[v][d] : q[d];
                    }
                }
            }
            if (store_it) X::store(orow0 + (size_t)r * (size_t)P.ox + (size_t)(xl - P.tx), q, nvalid);
            if (rmask) {
                if (ramp)
                    L.meta[nitems + (int)__builtin_amdgcn_mbcnt_hi((unsigned int)(rmask >> 32), __builtin_amdgcn_mbcnt_lo((unsigned int)rmask, 0u))] =
                        (unsigned int)r | ((unsigned int)lane << 3);
                nitems += __builtin_popcountll(rmask);
            }
            if (fmask) {      // lanes that see an extra view: the back of the queue
                if (frc)
                    L.meta[kItems - 1 - nforced - (int)__builtin_amdgcn_mbcnt_hi((unsigned int)(fmask >> 32), __builtin_amdgcn_mbcnt_lo((unsigned int)fmask, 0u))] =
                        (unsigned int)r | ((unsigned int)lane << 3);
                nforced += __builtin_popcountll(fmask);
            }
        }
        if (nitems > 0 && !(P.ablate & 1)) rowd_dense<T, NV>(P, L, nitems, nv, C, k * kStep, lane, orow0);
        if (nforced > 0) rowd_dense_rolled<T>(P, L, nforced, gp, nv, G.nx, xs0 + k * kStep, z, y0, lane, orow0);
    }
}

// Segments with more than 4 views of comparable weight (corners of a 3D tile grid: a few per cent of the voxels): the same two
// passes with a rolled loop over the views -- one weight per lane and view in the row pass, every view per voxel in the dense one.
template <typename T>
__device__ __forceinline__ void rowd_segment_many(const RowdParams& P, const RSeg& G, const RSeg* gp, int nv, int z, int y0, int nrows, int lane,
                                               T* __restrict__ orow0, RowdLds& L) {
    typedef Px<T> X;
    constexpr int ES = X::ES;
    const int xend = P.tx + P.ox;
    for (int s = G.s0; s < G.s1; ++s) {
        const int xl = P.tx + s * kStep + kRV * lane;
        const int nvalid = min(max(xend - xl, 0), kRV);
        int nitems = 0;
        for (int r = 0; r < nrows; ++r) {
            float num[kRV], den = 0.f, lw = 0.f;
            unsigned int lraw[4] = {0, 0, 0, 0};
            bool ramp = false;
#pragma unroll
            for (int j = 0; j < kRV; ++j) num[j] = 0.f;
#pragma nounroll
            for (int v = 0; v < nv; ++v) {
                const TrView V = load_uniform(P.views + load_uniform(gp->ids + v));
                const int xa = xl - V.lo[2], xb = V.hi[2] - xl;
                if (!__any(!((xa < -(kRV - 1)) || (xb < 0)))) continue;      // no lane inside the view's valid box
                const float4 n4 = L.nodes[load_uniform(gp->lv + v) * kRows + r];
                const int nbytes = (int)V.span * ES;
                const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)V.data, 0, nbytes, 0x00020000);
                const int o = ((z + V.io[0]) * V.stride_z + (y0 + r + V.io[1]) * V.stride_y + (xl + V.io[2])) * ES;
                unsigned int raw[4];
                X::load(rs, o, 0, raw);
                const LaneGeo g = lane_geo((float)(xl - V.sup_ilo[2]), (float)(V.sup_ihi[2] - V.sup_ilo[2]), V.sup_flo[2], V.sup_fhi[2], V.sup_k[2], xa, xb);
                bool unit, zero, rp;
                float wl;
                row_class(g, V.sup_k[2], n4, unit, zero, rp, wl);
                ramp = ramp || rp;
                float e[kRV];
                X::decode(raw, e);
                den += wl;
                const bool fl = (wl > 0.f) && (wl < 1.f);
                lw = fl ? wl : lw;
#pragma unroll
                for (int d = 0; d < 4; ++d) lraw[d] = fl ? raw[d] : lraw[d];
#pragma unroll
                for (int j = 0; j < kRV; ++j) num[j] = fmaf(wl, e[j], num[j]);
            }
            const float rd = (den > 0.f) ? __builtin_amdgcn_rcpf(den) : 0.f;
            unsigned int o8[kRV], q[4];
#pragma unroll
            for (int j = 0; j < kRV; ++j) o8[j] = (unsigned int)(int)(num[j] * rd);
            X::pack(o8, q);
            const bool one = (den == lw) && (lw > 0.f);
#pragma unroll
            for (int d = 0; d < 4; ++d) q[d] = one ? lraw[d] : q[d];
            ramp = ramp && (nvalid > 0);
            const unsigned long long rmask = __ballot(ramp);
            if (!ramp && nvalid > 0) X::store(orow0 + (size_t)r * (size_t)P.ox + (size_t)(xl - P.tx), q, nvalid);
            if (rmask) {
                if (ramp)
                    L.meta[kItems - 1 - nitems - (int)__builtin_amdgcn_mbcnt_hi((unsigned int)(rmask >> 32), __builtin_amdgcn_mbcnt_lo((unsigned int)rmask, 0u))] =
                        (unsigned int)r | ((unsigned int)lane << 3);
                nitems += __builtin_popcountll(rmask);
            }
        }
        if (nitems > 0) rowd_dense_rolled<T>(P, L, nitems, gp, nv, 0, P.tx + s * kStep, z, y0, lane, orow0);
    }
}

// MAXNV: the largest number of unrolled views of the strips this launch walks (strips are sorted into three classes on the host, so
// that the register budget of the common rows -- at most two views per step -- is not set by the rare corner rows)
template <typename T, int MAXNV>
__global__ __launch_bounds__(256) void fuse_rowd_kernel(RowdParams P) {
    __shared__ RowdLds lds[4];
    const int lane = threadIdx.x & 63;
    // XCD-aware order: workgroups are dealt round-robin to the 8 XCDs, so XCD k takes the k-th contiguous eighth of the row
    // groups (neighbouring rows share the cache lines at their ends: the output pitch is no multiple of the line size)
    const int wg = (blockIdx.x & 7) * (gridDim.x >> 3) + (blockIdx.x >> 3);
    const int wi = __builtin_amdgcn_readfirstlane(wg * 4 + (int)(threadIdx.x >> 6));
    if (wi >= P.nwaves) return;
    RowdLds& L = lds[threadIdx.x >> 6];
    int a = 0, b = P.nstrips - 1;      // strip of this wavefront: the last one whose first wavefront is <= wi
    while (a < b) {
        const int m = (a + b + 1) >> 1;
        if (load_uniform(P.strips + m).wave0 <= wi) a = m; else b = m - 1;
    }
    const RStrip S = load_uniform(P.strips + a);
    const int k = wi - S.wave0;
    const int z = S.z0 + k / S.nyg, y0 = S.y0 + (k % S.nyg) * kRows;
    const int nrows = min(kRows, S.y1 - y0);
    // row nodes of the blend profile: every (view of the strip, row) once
    for (int i = lane; i < S.nsv * kRows; i += 64) {
        const TrView& V = P.views[P.svlist[S.sv0 + i / kRows]];
        float g1 = 0.f, g2 = 0.f;
        const bool in = tr_row_nodes(V, z, min(y0 + i % kRows, S.y1 - 1), g1, g2);
        const float G1 = in ? g1 : 0.f, dG = in ? g2 - g1 : 0.f;
        L.nodes[i] = make_float4(G1, dG, rd_ramp(G1), (G1 > 0.f) ? 0.f : 1.f);
    }
    T* orow0 = (T*)P.out + ((size_t)(z - P.tz) * P.oy + (size_t)(y0 - P.ty)) * (size_t)P.ox;
    for (int g = 0; g < S.nseg; ++g) {
        const RSeg* gp = P.segs + S.seg0 + g;
        const RSeg G = load_uniform(gp);
        const int nv = G.nv;
        if (nv == 0) {
            const unsigned int zq[4] = {0, 0, 0, 0};
            for (int r = 0; r < nrows; ++r)
                for (int s = G.s0; s < G.s1; ++s) {
                    const int xl = s * kStep + kRV * lane;
                    const int nvalid = min(max(P.ox - xl, 0), kRV);
                    if (nvalid > 0) Px<T>::store(orow0 + (size_t)r * (size_t)P.ox + xl, zq, nvalid);
                }
        }
        else if (nv == 1) rowd_segment<T, 1>(P, G, gp, nv, z, y0, nrows, lane, orow0, L);
        else if (nv == 2) rowd_segment<T, 2>(P, G, gp, nv, z, y0, nrows, lane, orow0, L);
        else if (MAXNV >= 4 && nv <= 4) rowd_segment<T, 4>(P, G, gp, nv, z, y0, nrows, lane, orow0, L);
        else if (MAXNV > 4) rowd_segment_many<T>(P, G, gp, nv, z, y0, nrows, lane, orow0, L);
    }
}

struct RowdCache {
    unsigned long long hash = 0;
    bool valid = false, usable = false;
    int nstrips[3] = {0, 0, 0}, nwaves[3] = {0, 0, 0};      // per class: strips whose segments hold <= 2, <= 4, more views
    size_t off_strips[3] = {0, 0, 0}, off_segs = 0, off_sv = 0;
};
RowdCache g_rowd[MVS_MAX_DEVICES * MVS_MAX_LANES];
double g_rowd_plan_ms[MVS_MAX_DEVICES * MVS_MAX_LANES];

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

}  // namespace

double mvs_rowd_last_plan_ms(MvsContext* c) { return g_rowd_plan_ms[mvs_ctx_index(c->device)]; }

// Sets *done when the chunk was fused here; otherwise (a view with a fractional offset, more than 8 views over one voxel or 64
// over one row, float tiles, a chunk narrower than one step) the caller continues with the other fast paths.
int mvs_fuse_rowd(MvsContext* c, const TrView* htr, const TrView* dtr, int n_views, int dtype, void* dout, const int64_t os[3],
                  const int64_t trim[3], bool* done) {
    *done = false;
    g_rowd_plan_ms[mvs_ctx_index(c->device)] = 0.0;
    if (dtype != MVS_U16 && dtype != MVS_U8) return MVS_OK;
    const int es = dtype == MVS_U16 ? 2 : 1;
    const int t[3] = {(int)trim[0], (int)trim[1], (int)trim[2]};
    const int o[3] = {(int)os[0], (int)os[1], (int)os[2]};
    if (o[0] < 1 || o[1] < 1 || o[2] < kStep) return MVS_OK;
    if ((long long)o[0] * o[1] * o[2] * es > (1ll << 46)) return MVS_OK;
    unsigned long long h = fnv1a(htr, sizeof(TrView) * (size_t)n_views, 1469598103934665603ull);
    h = fnv1a(t, sizeof(t), h);
    h = fnv1a(o, sizeof(o), h);
    h ^= 0x726f7764ull + (unsigned long long)es;
    RowdCache& wc = g_rowd[mvs_ctx_index(c->device)];
    char* dbuf = nullptr;
    if (wc.valid && wc.hash == h && (!wc.usable || c->dev[14].ptr)) {
        if (!wc.usable) return MVS_OK;
        dbuf = (char*)c->dev[14].ptr;
    } else {
        const auto t0 = std::chrono::steady_clock::now();
        wc.valid = false;
        wc.usable = false;
        wc.hash = h;
        std::vector<int> live;
        for (int v = 0; v < n_views; ++v) {
            const TrView& V = htr[v];
            if (V.lo[0] > V.hi[0] || V.lo[1] > V.hi[1] || V.lo[2] > V.hi[2]) continue;
            if (!view_ok(V, es)) { wc.valid = true; if (getenv("MVS_PLAN_STATS")) fprintf(stderr, "[mvs rowd plan] view %d not eligible\n", v); return MVS_OK; }
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
            }
            std::sort(cut[d].begin(), cut[d].end());
            cut[d].erase(std::unique(cut[d].begin(), cut[d].end()), cut[d].end());
        }
        const int nsteps = (o[2] + kStep - 1) / kStep;
        std::vector<RStrip> strips[3];
        std::vector<RSeg> segs;
        std::vector<int> svlist;
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
                    if (htr[v].lo[1] <= y0 && htr[v].hi[1] >= y1 - 1 && std::max(htr[v].lo[2], t[2]) <= std::min(htr[v].hi[2], t[2] + o[2] - 1))
                        yv.push_back(v);
                if ((int)yv.size() > kMaxSV) { wc.valid = true; if (getenv("MVS_PLAN_STATS")) fprintf(stderr, "[mvs rowd plan] %zu views over one row\n", yv.size()); return MVS_OK; }      // too many views over one row
                RStrip S;
                memset(&S, 0, sizeof(S));
                S.z0 = z0; S.z1 = z1; S.y0 = y0; S.y1 = y1;
                S.seg0 = (int)segs.size();
                S.nyg = (y1 - y0 + kRows - 1) / kRows;
                S.sv0 = (int)svlist.size();
                S.nsv = (int)yv.size();
                for (int v : yv) svlist.push_back(v);
                // per view: the steps that hold at least one voxel of its valid box, and whether a lane's window can straddle the
                // first / last bytes of its slab here
                std::fill(mark.begin(), mark.end(), 0);
                struct VR { int s0, s1, a, b; };      // steps that hold a voxel of the valid box [a, b]
                std::vector<VR> vr(yv.size());
                for (size_t i = 0; i < yv.size(); ++i) {
                    const TrView& V = htr[yv[i]];
                    vr[i].a = std::max(V.lo[2], t[2]);
                    vr[i].b = std::min(V.hi[2], t[2] + o[2] - 1);
                    vr[i].s0 = (vr[i].a - t[2]) / kStep;
                    vr[i].s1 = (vr[i].b - t[2]) / kStep;
                    mark[vr[i].s0] = 1;
                    mark[vr[i].s1 + 1] = 1;
                }
                mark[0] = 1;
                mark[nsteps] = 1;
                int s0 = 0, max_nv = 0;
                for (int s = 1; s <= nsteps; ++s) {
                    if (!mark[s]) continue;
                    RSeg G;
                    memset(&G, 0, sizeof(G));
                    G.s0 = s0;
                    G.s1 = s;
                    // the views of these steps (ascending: the accumulation order of the reference and of the other kernels) and
                    // how many voxels of the steps each of them covers
                    const int xa = t[2] + s0 * kStep, xb = std::min(t[2] + s * kStep, t[2] + o[2]) - 1;
                    int who[kMaxV], cov[kMaxV], n = 0;
                    for (size_t i = 0; i < yv.size(); ++i)
                        if (vr[i].s0 <= s0 && vr[i].s1 >= s - 1) {
                            if (n == kMaxV) {
                                wc.valid = true;
                                if (getenv("MVS_PLAN_STATS")) fprintf(stderr, "[mvs rowd plan] more than %d views over one step\n", kMaxV);
                                return MVS_OK;
                            }
                            who[n] = (int)i;
                            cov[n] = std::min(vr[i].b, xb) - std::max(vr[i].a, xa) + 1;
                            ++n;
                        }
                    // a view that reaches less than a third of the way into the steps (the third tile along x of a 512-voxel window)
                    // is "extra": its lanes become items of the dense pass, the row pass does not see it.  More than 4 others: all
                    // views go through the rolled path.
                    bool extra[kMaxV];
                    int nmain = 0;
                    for (int q = 0; q < n; ++q) { extra[q] = n > 2 && 3 * cov[q] < xb - xa + 1; nmain += extra[q] ? 0 : 1; }
                    if (nmain > 4 || nmain == 0)
                        for (int q = 0; q < n; ++q) extra[q] = false;
                    for (int pass = 0; pass < 2; ++pass)
                        for (int q = 0; q < n; ++q)
                            if (extra[q] == (pass == 1)) {
                                const int k = G.nv + G.nx;
                                G.ids[k] = yv[who[q]];
                                G.lv[k] = who[q];
                                if (pass == 0) ++G.nv; else ++G.nx;
                            }
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
        if (total_waves == 0 || total_waves > 0x3fffffffLL || segs.size() > (1u << 24)) { wc.valid = true; return MVS_OK; }
        auto al = [](size_t v) { return (v + 255) / 256 * 256; };
        size_t total = 0;
        for (int k = 0; k < 3; ++k) { wc.off_strips[k] = total; total += al(strips[k].size() * sizeof(RStrip)); }
        wc.off_segs = total;
        total += al(segs.size() * sizeof(RSeg));
        wc.off_sv = total;
        total += al(svlist.size() * sizeof(int));
        dbuf = (char*)mvs_scratch(c, 14, total);
        if (!dbuf) return MVS_ERR_HIP;
        char* hb = (char*)mvs_pinned_slot(c, 1, total);       // slot 0 holds the view parameters still in flight
        if (!hb) return MVS_ERR_HIP;
        for (int k = 0; k < 3; ++k) memcpy(hb + wc.off_strips[k], strips[k].data(), strips[k].size() * sizeof(RStrip));
        memcpy(hb + wc.off_segs, segs.data(), segs.size() * sizeof(RSeg));
        memcpy(hb + wc.off_sv, svlist.data(), svlist.size() * sizeof(int));
        MVS_HIP_TRY(c, hipMemcpyAsync(dbuf, hb, total, hipMemcpyHostToDevice, c->stream));
        mvs_pinned_mark(c, 1);
        for (int k = 0; k < 3; ++k) { wc.nstrips[k] = (int)strips[k].size(); wc.nwaves[k] = (int)nwaves[k]; }
        wc.valid = true;
        wc.usable = true;
        g_rowd_plan_ms[mvs_ctx_index(c->device)] = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
        if (getenv("MVS_PLAN_STATS"))
            fprintf(stderr, "[mvs rowd plan] strips %zu / %zu / %zu, segments %zu, wavefronts %lld / %lld / %lld\n", strips[0].size(), strips[1].size(),
                    strips[2].size(), segs.size(), nwaves[0], nwaves[1], nwaves[2]);
    }
    RowdParams P;
    P.views = dtr;
    P.segs = (const RSeg*)(dbuf + wc.off_segs);
    P.svlist = (const int*)(dbuf + wc.off_sv);
    P.out = dout;
    P.oy = o[1]; P.ox = o[2];
    P.tz = t[0]; P.ty = t[1]; P.tx = t[2];
    P.ablate = c->ablate;
    MVS_HIP_TRY(c, hipEventRecord(c->ev_start, c->stream));   // kernel time only (the plan is host work, cached per geometry)
    for (int k = 0; k < 3; ++k) {
        if (!wc.nwaves[k]) continue;
        P.strips = (const RStrip*)(dbuf + wc.off_strips[k]);
        P.nstrips = wc.nstrips[k];
        P.nwaves = wc.nwaves[k];
        const dim3 grid(((wc.nwaves[k] + 3) / 4 + 7) / 8 * 8), block(256);
#define MVS_RD(T) do { if (k == 0) hipLaunchKernelGGL((fuse_rowd_kernel<T, 2>), grid, block, 0, c->stream, P); \
                       else if (k == 1) hipLaunchKernelGGL((fuse_rowd_kernel<T, 4>), grid, block, 0, c->stream, P); \
                       else hipLaunchKernelGGL((fuse_rowd_kernel<T, 8>), grid, block, 0, c->stream, P); } while (0)
        if (dtype == MVS_U16) MVS_RD(unsigned short); else MVS_RD(unsigned char);
#undef MVS_RD
    }
    MVS_HIP_TRY(c, hipGetLastError());
    *done = true;
    return MVS_OK;
}
