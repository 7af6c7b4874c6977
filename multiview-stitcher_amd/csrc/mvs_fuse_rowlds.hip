// mvs_fuse_rowlds.hip -- row-owning translation fast path of mvs_fuse_chunk with LDS-staged inputs (gfx950):
// uint16 tiles, one tap per view (integer offsets or interpolation order 0) -- the state of a tile grid after
// registration.  Reference: fusion/_core.py:1608-1713, weights.py:325-345, 391-511.
//
// Decomposition (host, cached per geometry): the chunk is cut along z and y at the view borders into STRIPS; inside a
// strip the views touching a row are constant, so a row is cut along x once per strip into CELLS with a constant view
// list.  A workgroup owns R consecutive COMPLETE output rows of one plane (R = 8 where two views overlap at most),
// so every output line is written by one compute unit within microseconds and every input tile row is consumed in
// one go (the region kernels, mvs_fuse_region.hip, fetched the lines at region faces once per region: 1.47x reads).
//
// Memory-level parallelism is decoupled from the arithmetic: per (cell, view) the R x nseg 16-byte segments the cell
// needs are flattened over lanes and gathered by LDS-DMA (`buffer_load_dwordx4 ... lds`: per-lane source address,
// lane-linear destination, no VGPR), 64 segments = 1 KiB per instruction.  A wavefront issues the DMA of ALL its
// units first (8-16 KiB in flight per wavefront at no register cost, ~20 wavefronts per CU), waits once
// (s_waitcnt vmcnt(0) -- it only reads what it loaded itself, so no workgroup barrier), then runs the arithmetic of
// its units out of LDS with aligned ds_read_b128.  Arithmetic = that of the region kernels (same weight profile,
// accumulator rules and exactness shortcuts).
#include "mvs_fuse_plan.h"

#include <chrono>
#include <cstdlib>
#include <cstring>
#include <vector>

using namespace mvsplan;

namespace {

constexpr int kRV = 8;        // voxels per lane (one 16-byte segment of uint16)
constexpr int kMaxCV = 8;     // views per cell

struct LCell {                // 64 bytes
    int x0, x1;               // chunk-index range, end exclusive
    int nv_cls;               // nviews | cls << 8   (cls 0: no view, 1: copy, 2: blend)
    int masks;                // bits 0-7: view ids[v] has weight 1 everywhere in the 3D box of (strip, cell); bit 15: every
                              // view covers the box with a weight > 0 everywhere; bits 16-23: view covers the box only partly
    int ids[kMaxCV];
    int nseg;                 // 16-byte segments per row: ceil((x1 - x0) / 8)
    int lds_off;              // first view block of the cell in LDS, in segments; view v at lds_off + v * nunits * 64
    int unit0, nunits;        // units (64 flattened (row, segment) pairs) of this cell: index of the first, count
};
static_assert(sizeof(LCell) == 64, "LCell layout");

// one workgroup: rows y .. y1 - 1 (at most R) of plane z of one strip; the strip's cells and (copies of) the records of
// its views are contiguous in the cell / strip-view tables
struct LItem { int z, y, y1, R, cell0, ncells, view0, nsv; };
static_assert(sizeof(LItem) == 32, "LItem layout");

constexpr int kMaxStripCells = 32;   // cells per strip (two lane tables of 16)
constexpr int kMaxStripViews = 16;   // distinct views per strip (one lane table)

typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void* lds_ptr_t;

typedef unsigned short us8_t __attribute__((ext_vector_type(8), aligned(2)));

__device__ __forceinline__ void store8(unsigned short* p, const float (&q)[kRV], int nvalid) {
    if (nvalid >= kRV) {
        us8_t v;
#pragma unroll
        for (int j = 0; j < kRV; ++j) v[j] = (unsigned short)(int)q[j];
        *reinterpret_cast<us8_t*>(p) = v;
    } else {
#pragma unroll
        for (int j = 0; j < kRV; ++j)
            if (j < nvalid) p[j] = (unsigned short)(int)q[j];
    }
}
__device__ __forceinline__ void store8_raw(unsigned short* p, u32x4_t w, int nvalid) {
    if (nvalid >= kRV) {
        us8_t v;
        v[0] = (unsigned short)(w.x & 0xffffu); v[1] = (unsigned short)(w.x >> 16);
        v[2] = (unsigned short)(w.y & 0xffffu); v[3] = (unsigned short)(w.y >> 16);
        v[4] = (unsigned short)(w.z & 0xffffu); v[5] = (unsigned short)(w.z >> 16);
        v[6] = (unsigned short)(w.w & 0xffffu); v[7] = (unsigned short)(w.w >> 16);
        *reinterpret_cast<us8_t*>(p) = v;
    } else {
        const unsigned int ww[4] = {w.x, w.y, w.z, w.w};
#pragma unroll
        for (int j = 0; j < kRV; ++j)
            if (j < nvalid) p[j] = (unsigned short)((ww[j >> 1] >> (16 * (j & 1))) & 0xffffu);
    }
}
__device__ __forceinline__ void decode8(u32x4_t w, float (&v)[kRV]) {
    v[0] = (float)(w.x & 0xffffu); v[1] = (float)(w.x >> 16);
    v[2] = (float)(w.y & 0xffffu); v[3] = (float)(w.y >> 16);
    v[4] = (float)(w.z & 0xffffu); v[5] = (float)(w.z >> 16);
    v[6] = (float)(w.w & 0xffffu); v[7] = (float)(w.w >> 16);
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

// flat index of a unit's lane -> (row, segment); nseg <= 8192, f < 2^20: exact in float
__device__ __forceinline__ void unflatten(int f, int nseg, float rnseg, int& row, int& seg) {
    row = (int)(((float)f + 0.5f) * rnseg);
    seg = f - row * nseg;
    if (seg < 0) { --row; seg += nseg; }
    if (seg >= nseg) { ++row; seg -= nseg; }
}

// Lane tables: the cells of the strip (16 dwords each) and the addressing fields of its views' records are loaded ONCE
// per wavefront into lane-distributed registers -- register j, lane 4 i + c holds dword 4 j + c of entry i -- and every
// wave-uniform field is pulled out with v_readlane, so the loops below contain no scalar memory operation (dependent
// s_load chains item -> cell -> view id -> view record cost ~0.5 us each, dozens per wavefront).
struct CellTab { unsigned int a[4], b[4]; };       // cells 0-15, 16-31
struct ViewTab { unsigned int r[4]; };             // dwords 4-19 of the TrView records: hi[1..2], io, ..., data, span, strides
__device__ __forceinline__ int rl(unsigned int x, int l) { return __builtin_amdgcn_readlane((int)x, l); }
template <int F> __device__ __forceinline__ int cell_field(const CellTab& T, int ci) {
    return (ci < 16) ? rl(T.a[F >> 2], 4 * ci + (F & 3)) : rl(T.b[F >> 2], 4 * (ci - 16) + (F & 3));
}
// LCell dword indices
enum { CF_X0 = 0, CF_X1, CF_NVCLS, CF_MASKS, CF_ID0, CF_NSEG = 12, CF_LDSOFF, CF_UNIT0, CF_NUNITS };
__device__ __forceinline__ int cell_id(const CellTab& T, int ci, int v) {     // ids[v], v runtime
    const int l = 4 * (ci & 15) + (v & 3);
    const unsigned int lo = (ci < 16) ? T.a[1] : T.b[1], hi = (ci < 16) ? T.a[2] : T.b[2];
    return (v < 4) ? rl(lo, l) : rl(hi, l);
}
template <int F> __device__ __forceinline__ int view_field(const ViewTab& T, int s) { return rl(T.r[(F >> 2) - 1], 4 * s + (F & 3)); }

// Per item the workgroup computes two small LDS tables while its DMA is in flight (the blend weight of view s at
// (z, y, x) is a piecewise-linear profile along x whose nodes depend on (s, z, y) only):
//   xpar[s]      the x parameters of strip view s (support ends, nodes per pixel, valid x range)
//   node[s][row] G1, dG of the x profile of view s in row y0 + row, flags: bit 0 inside the support along z and y,
//                bit 1 row inside the view's valid box (z, y)
struct XPar { int ilo, ihi; float flo, fhi; float kx; int lo2, hi2, pad; };
struct Node { float G1, dG; int flags, pad; };
static_assert(sizeof(XPar) == 32 && sizeof(Node) == 16, "table layouts");
constexpr int kTabSegs = (kMaxStripViews * 32 + kMaxStripViews * 8 * 16) / 16;   // 160 segments = 2560 bytes ahead of the data blocks

__global__ __launch_bounds__(256) void fuse_rowlds_kernel(const TrView* __restrict__ sviews, const LCell* __restrict__ cells,
                                                          const LItem* __restrict__ items, int nitems,
                                                          unsigned short* __restrict__ out, int oy, int ox, int tz, int ty, int tx, int ablate) {
    extern __shared__ u32x4_t lds[];
    constexpr int ES = 2;
    XPar* xpar = reinterpret_cast<XPar*>(lds);
    Node* nodes = reinterpret_cast<Node*>(lds + kMaxStripViews * 2);        // [s][8]
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int nwm = (int)(blockDim.x >> 6) - 1;      // wavefronts per workgroup - 1 (power of two - 1)
    // XCD-aware order: workgroups are dealt round-robin to the 8 XCDs (each with its own L2), so XCD k takes the k-th
    // contiguous eighth of the item list (consecutive row groups of a plane).
    const int wg = (blockIdx.x & 7) * (gridDim.x >> 3) + (blockIdx.x >> 3);
    if (wg >= nitems) return;
    const LItem it = items[wg];
    const int zc = it.z, y0 = it.y, y1 = it.y1, R = it.R, ncells = it.ncells;

    CellTab CT;
    ViewTab VT;
    {
        const int e = lane >> 2, comp = lane & 3;
        const unsigned int* cw = reinterpret_cast<const unsigned int*>(cells + it.cell0);
        const unsigned int* vw = reinterpret_cast<const unsigned int*>(sviews + it.view0);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            CT.a[j] = (e < ncells) ? cw[e * 16 + 4 * j + comp] : 0u;
            CT.b[j] = (e + 16 < ncells) ? cw[(e + 16) * 16 + 4 * j + comp] : 0u;
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) VT.r[j] = (e < it.nsv) ? vw[e * 44 + 4 * (j + 1) + comp] : 0u;
    }

    // ---- phase 1: gather every (cell, view) block of this wavefront's units into LDS ----
    for (int ci = 0; ci < ncells; ++ci) {
        const int nv = cell_field<CF_NVCLS>(CT, ci) & 0xff;
        if (nv == 0) continue;
        const int nseg = cell_field<CF_NSEG>(CT, ci), total = R * nseg;
        const int nunits = cell_field<CF_NUNITS>(CT, ci), unit0 = cell_field<CF_UNIT0>(CT, ci);
        const int x0 = cell_field<CF_X0>(CT, ci), lds_off = cell_field<CF_LDSOFF>(CT, ci);
        const float rnseg = 1.f / (float)nseg;
        for (int k = 0; k < nunits; ++k) {
            if (((unit0 + k) & nwm) != wave) continue;
            const int f = 64 * k + lane;
            const bool active = f < total;
            int row, seg;
            unflatten(min(f, total - 1), nseg, rnseg, row, seg);
            const int yl = min(y0 + row, y1 - 1), xl = x0 + kRV * seg;
            for (int v = 0; v < nv; ++v) {
                const int s = cell_id(CT, ci, v);
                const unsigned long long data = ((unsigned long long)(unsigned)view_field<15>(VT, s) << 32) | (unsigned)view_field<14>(VT, s);
                const int nbytes = view_field<16>(VT, s) * ES;
                const int sy = view_field<18>(VT, s), sz = view_field<19>(VT, s);
                const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)data, 0, nbytes, 0x00020000);
                const int vo = ((zc + view_field<6>(VT, s)) * sz + (yl + view_field<7>(VT, s)) * sy + (xl + view_field<8>(VT, s))) * ES;
                u32x4_t* dst = lds + (lds_off + (v * nunits + k) * 64);
                // a vector load that is not entirely inside the slab comes back as 0: windows touching its first / last bytes
                // are fetched element by element (first / last rows of a slab only)
                const bool str = active && ((vo < 0 && vo + 16 > 0) || (vo < nbytes && vo + 16 > nbytes));
                if (active && !str && !(ablate & 1)) __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lds_ptr_t)dst, 16, vo, 0, 0, 0);
                if (__any(str)) {
                    if (str) {
                        unsigned int e[kRV];
#pragma unroll
                        for (int j = 0; j < kRV; ++j) e[j] = (unsigned short)__builtin_amdgcn_raw_buffer_load_b16(rsrc, vo + j * ES, 0, 0);
                        u32x4_t w;
                        w.x = e[0] | (e[1] << 16); w.y = e[2] | (e[3] << 16); w.z = e[4] | (e[5] << 16); w.w = e[6] | (e[7] << 16);
                        dst[lane] = w;
                    }
                }
            }
        }
    }
    // ---- weight tables of the item (while the DMA is in flight): thread (s, row) ----
    {
        const int rsh = (R >= 8) ? 3 : (R >= 4) ? 2 : (R >= 2) ? 1 : 0;
        for (int idx = threadIdx.x; idx < (it.nsv << rsh); idx += blockDim.x) {
            const int s = idx >> rsh, row = idx & (R - 1);
            const TrView& V = sviews[it.view0 + s];
            const int yr = min(y0 + row, y1 - 1);
            float G1, dG;
            bool inside;
            row_nodes(V, zc, yr, G1, dG, inside);
            const bool zy_ok = (zc >= V.lo[0]) && (zc <= V.hi[0]) && (yr >= V.lo[1]) && (yr <= V.hi[1]);
            Node nd;
            nd.G1 = G1; nd.dG = dG; nd.flags = (inside ? 1 : 0) | (zy_ok ? 2 : 0); nd.pad = 0;
            nodes[s * 8 + row] = nd;
            if (row == 0) {
                XPar xp;
                xp.ilo = V.sup_ilo[2]; xp.ihi = V.sup_ihi[2]; xp.flo = V.sup_flo[2]; xp.fhi = V.sup_fhi[2];
                xp.kx = V.sup_k[2]; xp.lo2 = V.lo[2]; xp.hi2 = V.hi[2]; xp.pad = 0;
                xpar[s] = xp;
            }
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (ablate & 2) return;

    // ---- phase 2: arithmetic of this wavefront's units out of LDS ----
    for (int ci = 0; ci < ncells; ++ci) {
        const int nvcls = cell_field<CF_NVCLS>(CT, ci);
        const int nv = nvcls & 0xff, cls = nvcls >> 8;
        const int x0 = cell_field<CF_X0>(CT, ci), x1 = cell_field<CF_X1>(CT, ci);
        const int nseg = cell_field<CF_NSEG>(CT, ci), total = R * nseg;
        const int nunits = cell_field<CF_NUNITS>(CT, ci), unit0 = cell_field<CF_UNIT0>(CT, ci);
        const int lds_off = cell_field<CF_LDSOFF>(CT, ci), masks = cell_field<CF_MASKS>(CT, ci);
        const float rnseg = 1.f / (float)nseg;
        const int vstride = nunits * 64;
        const int full = (1 << nv) - 1;
        const int allone_mask = masks & 0xff, partial_mask = (masks >> 16) & 0xff;
        const bool allpos = (masks >> 15) & 1;
        const bool lean = ((allone_mask & full) == full) && !partial_mask;   // every view in bounds with weight 1: plain mean
        const float need = (nv == 1) ? 3e-4f : 1.f;   // a voxel seen by ONE view only needs a weight that does not round to 0
        for (int k = 0; k < nunits; ++k) {
            if (((unit0 + k) & nwm) != wave) continue;
            const int f = 64 * k + lane;
            const bool active = f < total;
            int row, seg;
            unflatten(min(f, total - 1), nseg, rnseg, row, seg);
            const int yc = y0 + row;
            const bool row_ok = active && yc < y1 && !(ablate & 4);
            const int yl = min(yc, y1 - 1);
            const int xq = x0 + kRV * seg;
            const int nvx = min(x1 - xq, kRV);
            unsigned short* op = out + ((long long)(zc - tz) * oy + (yl - ty)) * (long long)ox + (xq - tx);
            const u32x4_t* blk = lds + (lds_off + k * 64 + lane);

            if (cls == 0) {
                const float q[kRV] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
                if (row_ok) store8(op, q, nvx);
                continue;
            }
            if (cls == 1 || (ablate & 8)) {      // one full view with a weight > 0 everywhere: the result is the value
                const u32x4_t w = blk[0];
                if (row_ok) store8_raw(op, w, nvx);
                continue;
            }
            if (lean) {
                float num[kRV];
#pragma unroll
                for (int j = 0; j < kRV; ++j) num[j] = 0.f;
#pragma unroll 1
                for (int v = 0; v < nv; ++v) {
                    float e[kRV];
                    decode8(blk[v * vstride], e);
#pragma unroll
                    for (int j = 0; j < kRV; ++j) num[j] += e[j];
                }
                const float rn = __builtin_amdgcn_rcpf((float)nv);
                float q[kRV];
#pragma unroll
                for (int j = 0; j < kRV; ++j) q[j] = num[j] * rn;
                if (row_ok) store8(op, q, nvx);
                continue;
            }

            float num[kRV], den[kRV], last[kRV], wlast[kRV];
#pragma unroll
            for (int j = 0; j < kRV; ++j) { num[j] = 0.f; den[j] = 0.f; last[j] = 0.f; wlast[j] = 0.f; }
            bool all_unit = true;      // every view came out with weight 1 on this unit
#pragma unroll 1
            for (int v = 0; v < nv; ++v) {
                float val[kRV];
                decode8(blk[v * vstride], val);
                const bool partial = (partial_mask >> v) & 1;
                bool unit = (allone_mask >> v) & 1;
                if (unit && !partial) {
#pragma unroll
                    for (int j = 0; j < kRV; ++j) { num[j] += val[j]; den[j] += 1.f; }
                    continue;
                }
                const int s = cell_id(CT, ci, v);
                const Node nd = nodes[s * 8 + row];
                const XPar xp = xpar[s];
                // views that cover the box only partly: per-voxel in-bounds test against the view's valid box
                bool inb[kRV];
#pragma unroll
                for (int j = 0; j < kRV; ++j) inb[j] = true;
                if (partial) {
                    const bool zy_ok = (nd.flags & 2) != 0;
                    const int jlo = xp.lo2 - xq, jw = xp.hi2 - xp.lo2;
#pragma unroll
                    for (int j = 0; j < kRV; ++j) inb[j] = zy_ok && ((unsigned)(j - jlo) <= (unsigned)jw);
                }
                float w[kRV];
                if (!unit) {
                    const float G1 = nd.G1, dG = nd.dG;
                    const bool inside = (nd.flags & 1) != 0;
                    const float kx = xp.kx;
                    const float dl0 = (float)(xq - xp.ilo) - xp.flo;
                    const float dh0 = (float)(xp.ihi - xq) - xp.fhi;
                    // The profile is concave along x, so over the lane's 8 voxels its minimum sits at voxel 0 or 7:
                    // two evaluations tell whether the whole segment has weight 1.
                    const float u0 = fminf(dl0, dh0) * kx, u7 = fminf(dl0 + 7.f, dh0 - 7.f) * kx;
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
                            const float u = fminf(dl0 + (float)j, dh0 - (float)j) * kx;
                            const float W = (u >= 0.f && inside) ? row_profile(u, G1, dG) : 0.f;
                            w[j] = blend_ramp_nb(W);
                        }
                    }
                }
                if (unit) {
#pragma unroll
                    for (int j = 0; j < kRV; ++j) {
                        num[j] += inb[j] ? val[j] : 0.f;
                        den[j] += inb[j] ? 1.f : 0.f;
                    }
                } else if (allpos) {
                    // every view of the box is in bounds with a strictly positive weight everywhere: plain weighted sums
                    all_unit = false;
#pragma unroll
                    for (int j = 0; j < kRV; ++j) {
                        num[j] = fmaf(w[j], val[j], num[j]);
                        den[j] += w[j];
                    }
                } else {
                    all_unit = false;
#pragma unroll
                    for (int j = 0; j < kRV; ++j) {
                        const float we = inb[j] ? w[j] : 0.f;
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
                if (nv == 1 && all_unit && !partial_mask) o = num[j];      // a single full view with weight 1
                else if (allpos) o = num[j] * __builtin_amdgcn_rcpf(den[j]);
                else {
                    o = num[j] * __builtin_amdgcn_rcpf(den[j]);
                    o = (den[j] == wlast[j]) ? last[j] : o;                // single ramp contributor: exact value
                }
                if (!(fabsf(o) <= 3.4028234e38f)) o = 0.f;
                q[j] = o;
            }
            if (row_ok) store8(op, q, nvx);
        }
    }
}

struct LdsPlan {
    unsigned long long hash = 0;
    bool valid = false;
    int class_count[3] = {0, 0, 0};       // work items per LDS size class
    int class_lds[3] = {0, 0, 0};         // bytes of dynamic LDS per class
    int wpg = 4;
    size_t off_cells = 0, off_items = 0;
    double build_ms = 0.0;
};
LdsPlan g_lds_plan[MVS_MAX_DEVICES * MVS_MAX_LANES];
double g_lds_last_plan_ms[MVS_MAX_DEVICES * MVS_MAX_LANES];

struct HStrip { int z0, z1, y0, y1, cell0, ncells, R, view0, nsv, cls; };

}  // namespace

double mvs_rowlds_last_plan_ms(MvsContext* c) { return g_lds_last_plan_ms[mvs_ctx_index(c->device)]; }

// Returns MVS_OK and sets *done = true when the chunk was fused by the LDS-staged row kernel; *done = false means the
// caller must use another path (not uint16, a view with more than one tap, more than 8 views on one cell, more than 16
// views or 32 cells on one strip, a strip whose single row does not fit into LDS).
int mvs_fuse_rowlds(MvsContext* c, const TrView* htr, const TrView* dtr, int n_views, int dtype, void* dout, const int64_t os[3],
                    const int64_t trim[3], bool* done) {
    (void)dtr;
    *done = false;
    if (dtype != MVS_U16) return MVS_OK;
    for (int v = 0; v < n_views; ++v)
        if (view_needs_taps(htr[v], dtype)) return MVS_OK;
    const int t[3] = {(int)trim[0], (int)trim[1], (int)trim[2]};
    const int o[3] = {(int)os[0], (int)os[1], (int)os[2]};
    unsigned long long h = fnv1a(htr, sizeof(TrView) * (size_t)n_views, 1469598103934665603ull);
    h = fnv1a(t, sizeof(t), h);
    h = fnv1a(o, sizeof(o), h);
    LdsPlan& pc = g_lds_plan[mvs_ctx_index(c->device)];
    g_lds_last_plan_ms[mvs_ctx_index(c->device)] = 0.0;
    char* dbuf = nullptr;
    if (pc.valid && pc.hash == h && c->dev[14].ptr) {
        dbuf = (char*)c->dev[14].ptr;      // same geometry as the previous call: the plan is still on the device
    } else {
        const auto t_begin = std::chrono::steady_clock::now();
        std::vector<int> all(n_views);
        for (int v = 0; v < n_views; ++v) all[v] = v;
        std::vector<int> pz, py, px;
        axis_breakpoints(htr, all, 0, t[0], o[0], &pz);
        axis_breakpoints(htr, all, 1, t[1], o[1], &py);
        if ((pz.size() - 1) * (py.size() - 1) > 20000) return MVS_OK;
        const long long rows_total = (long long)o[0] * o[1];
        const int wpg = rows_total >= 32768 ? 4 : rows_total >= 8192 ? 2 : 1;
        const int lds_class_limit[3] = {40 * 1024, 80 * 1024, 160 * 1024};

        std::vector<HStrip> strips;
        std::vector<LCell> cells;
        std::vector<TrView> sviews;          // per strip: copies of the records of its views (strip-local slots)
        std::vector<int> zviews, svs, slot_of(n_views);
        int class_lds[3] = {0, 0, 0};
        for (size_t iz = 0; iz + 1 < pz.size(); ++iz) {
            zviews.clear();
            for (int v = 0; v < n_views; ++v)
                if (htr[v].lo[0] < pz[iz + 1] && htr[v].hi[0] >= pz[iz] && htr[v].lo[1] <= htr[v].hi[1] && htr[v].lo[2] <= htr[v].hi[2]) zviews.push_back(v);
            for (size_t iy = 0; iy + 1 < py.size(); ++iy) {
                svs.clear();
                for (int v : zviews)
                    if (htr[v].lo[1] < py[iy + 1] && htr[v].hi[1] >= py[iy]) svs.push_back(v);
                if ((int)svs.size() > kMaxStripViews) return MVS_OK;
                HStrip S;
                memset(&S, 0, sizeof(S));
                S.z0 = pz[iz]; S.z1 = pz[iz + 1]; S.y0 = py[iy]; S.y1 = py[iy + 1];
                S.cell0 = (int)cells.size();
                S.view0 = (int)sviews.size();
                S.nsv = (int)svs.size();
                for (size_t k = 0; k < svs.size(); ++k) { slot_of[svs[k]] = (int)k; sviews.push_back(htr[svs[k]]); }
                axis_breakpoints(htr, svs, 2, t[2], o[2], &px);
                if ((int)px.size() - 1 > kMaxStripCells) return MVS_OK;
                for (size_t ix = 0; ix + 1 < px.size(); ++ix) {
                    LCell C;
                    memset(&C, 0, sizeof(C));
                    C.x0 = px[ix]; C.x1 = px[ix + 1];
                    int nv = 0;
                    bool positive_full = false, all_positive = true;
                    for (int v : svs) {
                        if (!(htr[v].lo[2] < C.x1 && htr[v].hi[2] >= C.x0)) continue;   // does not touch the box
                        if (nv == kMaxCV) return MVS_OK;                                  // too many views: another path
                        const bool full = htr[v].lo[0] <= S.z0 && htr[v].hi[0] >= S.z1 - 1 && htr[v].lo[1] <= S.y0 &&
                                          htr[v].hi[1] >= S.y1 - 1 && htr[v].lo[2] <= C.x0 && htr[v].hi[2] >= C.x1 - 1;
                        // the profile is concave, so its minimum over the box sits at one of the 8 corners
                        float wmin = INFINITY;
                        for (int k = 0; k < 8; ++k) {
                            const int z = (k & 4) ? S.z1 - 1 : S.z0, y = (k & 2) ? S.y1 - 1 : S.y0, x = (k & 1) ? C.x1 - 1 : C.x0;
                            wmin = fminf(wmin, tr_weight_profile(htr[v], z, y, x));
                        }
                        const bool unit = full && wmin >= 1.f;          // weight exactly 1 everywhere
                        // weight > 0 everywhere: the float32 ramp (cos(pi (1 - W)) + 1) / 2 only vanishes when the cosine
                        // rounds to -1, i.e. W < 7.8e-5; at W = 3e-4 the cosine is 7 ulp away from -1
                        if (full && wmin >= 3e-4f) positive_full = true;
                        else all_positive = false;
                        if (unit) C.masks |= 1 << nv;
                        if (!full) C.masks |= 1 << (16 + nv);
                        C.ids[nv++] = slot_of[v];
                    }
                    if (nv > 0 && all_positive) C.masks |= 1 << 15;
                    const int cls = nv == 0 ? 0 : (nv == 1 && positive_full) ? 1 : 2;
                    C.nv_cls = nv | (cls << 8);
                    C.nseg = (C.x1 - C.x0 + kRV - 1) / kRV;
                    cells.push_back(C);
                }
                S.ncells = (int)cells.size() - S.cell0;
                // rows per workgroup: as many as fit the smallest LDS class, at least 4 (narrow overlap cells then still
                // fill most of a 64-lane unit), fewer only when even the largest class cannot hold them
                int R = 8, cls = -1, lds_bytes = 0;
                for (;;) {
                    int units = 0, off = kTabSegs;
                    for (int ci = S.cell0; ci < S.cell0 + S.ncells; ++ci) {
                        LCell& C = cells[ci];
                        C.nunits = (R * C.nseg + 63) / 64;
                        C.unit0 = units;
                        C.lds_off = off;
                        units += C.nunits;
                        off += (C.nv_cls & 0xff) * C.nunits * 64;
                    }
                    lds_bytes = off * 16;
                    cls = -1;
                    for (int k = 0; k < 3; ++k)
                        if (lds_bytes <= lds_class_limit[k]) { cls = k; break; }
                    if ((cls == 0) || (cls >= 0 && R <= 4) || R == 1) break;
                    R /= 2;
                }
                if (cls < 0) return MVS_OK;
                S.R = R;
                S.cls = cls;
                class_lds[cls] = std::max(class_lds[cls], lds_bytes);
                strips.push_back(S);
            }
        }
        if (cells.size() > (1u << 22)) return MVS_OK;
        // work items, z-major per class: consecutive workgroups write consecutive row groups of a plane
        std::vector<LItem> items_by_class[3];
        const size_t nys = py.size() - 1;
        for (size_t iz = 0; iz + 1 < pz.size(); ++iz)
            for (int z = pz[iz]; z < pz[iz + 1]; ++z)
                for (size_t iy = 0; iy < nys; ++iy) {
                    const HStrip& S = strips[iz * nys + iy];
                    std::vector<LItem>& dst = items_by_class[S.cls];
                    for (int y = S.y0; y < S.y1; y += S.R)
                        dst.push_back({z, y, std::min(y + S.R, S.y1), S.R, S.cell0, S.ncells, S.view0, S.nsv});
                }
        size_t nitems = 0;
        for (int k = 0; k < 3; ++k) nitems += items_by_class[k].size();
        if (nitems == 0 || nitems > (1u << 26)) return MVS_OK;
        const size_t vbytes = (sviews.size() * sizeof(TrView) + 255) / 256 * 256;
        const size_t cbytes = (cells.size() * sizeof(LCell) + 255) / 256 * 256;
        const size_t ibytes = nitems * sizeof(LItem);
        const size_t total = vbytes + cbytes + ibytes;
        char* hbuf = (char*)mvs_pinned_slot(c, 1, total + 256);   // slot 0 holds the view parameters still in flight
        if (!hbuf) return MVS_ERR_HIP;
        pc.valid = false;
        dbuf = (char*)mvs_scratch(c, 14, total + 256);
        if (!dbuf) return MVS_ERR_HIP;
        memcpy(hbuf, sviews.data(), sviews.size() * sizeof(TrView));
        memcpy(hbuf + vbytes, cells.data(), cells.size() * sizeof(LCell));
        size_t cur = vbytes + cbytes;
        for (int k = 0; k < 3; ++k) {
            pc.class_count[k] = (int)items_by_class[k].size();
            pc.class_lds[k] = class_lds[k];
            memcpy(hbuf + cur, items_by_class[k].data(), items_by_class[k].size() * sizeof(LItem));
            cur += items_by_class[k].size() * sizeof(LItem);
        }
        MVS_HIP_TRY(c, hipMemcpyAsync(dbuf, hbuf, total, hipMemcpyHostToDevice, c->stream));
        mvs_pinned_mark(c, 1);
        pc.hash = h;
        pc.wpg = wpg;
        pc.off_cells = vbytes;
        pc.off_items = vbytes + cbytes;
        pc.valid = true;
        pc.build_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_begin).count();
        g_lds_last_plan_ms[mvs_ctx_index(c->device)] = pc.build_ms;
        if (getenv("MVS_PLAN_STATS")) {
            fprintf(stderr, "[mvs rowlds plan] strips %zu cells %zu strip-views %zu; items / LDS bytes per class:", strips.size(), cells.size(), sviews.size());
            for (int k = 0; k < 3; ++k) fprintf(stderr, " %d / %d", pc.class_count[k], pc.class_lds[k]);
            fprintf(stderr, "; wpg %d, %.2f ms\n", wpg, pc.build_ms);
        }
    }
    static bool attr_set[MVS_MAX_DEVICES] = {false};
    if (!attr_set[mvs_hip_device(c->device)]) {
        MVS_HIP_TRY(c, hipFuncSetAttribute((const void*)fuse_rowlds_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        attr_set[mvs_hip_device(c->device)] = true;
    }
    const TrView* dsviews = (const TrView*)dbuf;
    const LCell* dcells = (const LCell*)(dbuf + pc.off_cells);
    const LItem* ditems = (const LItem*)(dbuf + pc.off_items);
    MVS_HIP_TRY(c, hipEventRecord(c->ev_start, c->stream));   // kernel time only (the plan is host work, cached per geometry)
    int item0 = 0;
    for (int k = 0; k < 3; ++k) {
        const int cnt = pc.class_count[k];
        if (!cnt) continue;
        const int nblocks = (cnt + 7) / 8 * 8;   // multiple of 8: see the XCD mapping in the kernel
        hipLaunchKernelGGL(fuse_rowlds_kernel, dim3(nblocks), dim3(64 * pc.wpg), (size_t)pc.class_lds[k], c->stream, dsviews, dcells,
                           ditems + item0, cnt, (unsigned short*)dout, o[1], o[2], t[0], t[1], t[2], c->ablate);
        item0 += cnt;
    }
    MVS_HIP_TRY(c, hipGetLastError());
    *done = true;
    return MVS_OK;
}
