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

struct LCell {                // 128 bytes
    int x0, x1;               // chunk-index range, end exclusive
    int nv_cls;               // nviews | cls << 8   (cls 0: no view, 1: copy, 2: blend)
    int masks;                // bits 0-7: view ids[v] has weight 1 everywhere in the 3D box of (strip, cell); bit 15: every
                              // view covers the box with a weight > 0 everywhere; bits 16-23: view covers the box only partly
    int ids_packed;           // strip-local view slots (< 16), 4 bits each: view v = (ids_packed >> 4 v) & 15
    int pad0[7];
    int nseg;                 // 16-byte segments per row: ceil((x1 - x0) / 8)
    int nunits;               // units (64 flattened (row, segment) pairs) per view block: ceil(R * nseg / 64)
    int sides;                // 2 bits per view: 1 / 2 = the whole cell lies on the first support interval next to the view's
                              // lower / upper x border and the view covers the box (blend weight = ramp(distance * kx * G1)),
                              // 0 = general profile
    int lsh;                  // lane layout of a unit: 2^lsh segment slots per row (>= nseg), 64 >> lsh rows: row = lane >> lsh,
                              // segment = lane & (2^lsh - 1) -- no division, the per-lane part of every address is constant
    int base[kMaxCV];         // byte offset of voxel (z = 0, y = 0, x = x0) (chunk indices) in view ids[v]'s slab
    int pad[8];
};
static_assert(sizeof(LCell) == 128, "LCell layout");

// addressing record of a strip view (what the DMA needs): 32 bytes
struct SvDma { unsigned int data_lo, data_hi; int nbytes, sy2, sz2, pad[3]; };
static_assert(sizeof(SvDma) == 32, "SvDma layout");

// one WAVEFRONT: rows y .. y1 - 1 (at most R) of plane z of one strip; the strip's cells and (copies of) the records of
// its views are contiguous in the cell / strip-view tables
struct LItem { int z, y, y1, R, cell0, ncells, view0, nsv; };
static_assert(sizeof(LItem) == 32, "LItem layout");

constexpr int kMaxStripCells = 32;   // cells per strip (two lane tables of 16)
constexpr int kMaxStripViews = 16;   // distinct views per strip (one lane table)

typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void* lds_ptr_t;

typedef unsigned short us8_t __attribute__((ext_vector_type(8), aligned(2)));

typedef unsigned int u32x4_a2 __attribute__((ext_vector_type(4), aligned(2)));
typedef unsigned int u32x2_a2 __attribute__((ext_vector_type(2), aligned(2)));
typedef unsigned int u32_a2 __attribute__((aligned(2)));

// 8 packed uint16 voxels (nvalid >= 8) or the first nvalid of them: a tail goes out as 4 + 2 + 1 elements (three
// predicated stores at most instead of one per element -- store instructions, not bytes, are what the partial segment at
// the end of every row of a cell costs).  Everything stays in packed dwords (no per-element arrays: a runtime-indexed
// array would live in scratch).
__device__ __forceinline__ void store8_packed(unsigned short* p, u32x4_t w, int nvalid) {
    if (nvalid >= kRV) {
        u32x4_a2 o;
        o.x = w.x; o.y = w.y; o.z = w.z; o.w = w.w;
        *reinterpret_cast<u32x4_a2*>(p) = o;
        return;
    }
    const bool has4 = (nvalid & 4) != 0;
    if (has4) {
        u32x2_a2 o;
        o.x = w.x; o.y = w.y;
        *reinterpret_cast<u32x2_a2*>(p) = o;
    }
    if (nvalid & 2) *reinterpret_cast<u32_a2*>(p + (has4 ? 4 : 0)) = has4 ? w.z : w.x;
    if (nvalid & 1) {                  // the last element has an even index nvalid - 1: the low half of dword (nvalid - 1) / 2
        const int d = (nvalid - 1) >> 1;
        const unsigned int v = (d == 0) ? w.x : (d == 1) ? w.y : (d == 2) ? w.z : w.w;
        p[nvalid - 1] = (unsigned short)(v & 0xffffu);
    }
}
__device__ __forceinline__ void store8(unsigned short* p, const float (&q)[kRV], int nvalid) {
    u32x4_t w;
    w.x = (unsigned int)(int)q[0] | ((unsigned int)(int)q[1] << 16);
    w.y = (unsigned int)(int)q[2] | ((unsigned int)(int)q[3] << 16);
    w.z = (unsigned int)(int)q[4] | ((unsigned int)(int)q[5] << 16);
    w.w = (unsigned int)(int)q[6] | ((unsigned int)(int)q[7] << 16);
    store8_packed(p, w, nvalid);
}
__device__ __forceinline__ void store8_raw(unsigned short* p, u32x4_t w, int nvalid) { store8_packed(p, w, nvalid); }
__device__ __forceinline__ u32x4_t pack8(const float (&q)[kRV]) {
    u32x4_t w;
    w.x = (unsigned int)(int)q[0] | ((unsigned int)(int)q[1] << 16);
    w.y = (unsigned int)(int)q[2] | ((unsigned int)(int)q[3] << 16);
    w.z = (unsigned int)(int)q[4] | ((unsigned int)(int)q[5] << 16);
    w.w = (unsigned int)(int)q[6] | ((unsigned int)(int)q[7] << 16);
    return w;
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

// Lane tables: the cells of the strip (32 dwords each) and the addressing records of its views (8 dwords each) are loaded
// ONCE per wavefront into lane-distributed registers -- register j, lane 4 i + c holds dword 4 j + c of entry i -- and every
// wave-uniform field is pulled out with v_readlane, so the loops below contain no scalar memory operation (dependent
// s_load chains item -> cell -> view id -> view record cost ~0.5 us each, dozens per wavefront).
struct CellTab { unsigned int r[16]; };            // register j, lane 2 i + c: dword 2 j + c of cell i (32 cells x 32 dwords)
struct ViewTab { unsigned int r[2]; };             // SvDma records of strip views 0-15
__device__ __forceinline__ int rl(unsigned int x, int l) { return __builtin_amdgcn_readlane((int)x, l); }
template <int F> __device__ __forceinline__ int cell_field(const CellTab& T, int ci) { return rl(T.r[F >> 1], 2 * ci + (F & 1)); }
// LCell dword indices
enum { CF_X0 = 0, CF_X1, CF_NVCLS, CF_MASKS, CF_IDS, CF_NSEG = 12, CF_NUNITS, CF_SIDES, CF_LSH, CF_BASE0 };
__device__ __forceinline__ int cell_base(const CellTab& T, int ci, int v) {       // base[v], v runtime < 8
    const int l = 2 * ci + (v & 1);
    const int b0 = rl(T.r[8], l), b1 = rl(T.r[9], l), b2 = rl(T.r[10], l), b3 = rl(T.r[11], l);
    const int h = v >> 1;
    return (h == 0) ? b0 : (h == 1) ? b1 : (h == 2) ? b2 : b3;
}
template <int F> __device__ __forceinline__ int view_field(const ViewTab& T, int s) { return rl(T.r[F >> 2], 4 * s + (F & 3)); }
enum { VF_DLO = 0, VF_DHI, VF_NBYTES, VF_SY2, VF_SZ2 };

// Per item the wavefront computes two small LDS tables while the DMA of its first cell is in flight (the blend weight of
// view s at (z, y, x) is a piecewise-linear profile along x whose nodes depend on (s, z, y) only):
//   xpar[s]      the x parameters of strip view s (support ends, nodes per pixel, valid x range)
//   node[s][row] G1, dG of the x profile of view s in row y0 + row, flags: bit 0 inside the support along z and y,
//                bit 1 row inside the view's valid box (z, y)
struct XPar { int ilo, ihi; float flo, fhi; float kx; int lo2, hi2, pad; };
struct Node { float G1, dG; int flags, pad; };
static_assert(sizeof(XPar) == 32 && sizeof(Node) == 16, "table layouts");

// One wavefront = R consecutive complete rows of one plane of one strip.  LDS per wavefront: tables | cell buffer 0 |
// cell buffer 1.  While cell c is computed out of one buffer the DMA of cell c + 1 fills the other.
__global__ __launch_bounds__(256) void fuse_rowlds_kernel(const TrView* __restrict__ sviews, const SvDma* __restrict__ svdma,
                                                          const LCell* __restrict__ cells,
                                                          const LItem* __restrict__ items, int nitems, int wave_segs, int tab_segs,
                                                          int buf_segs, unsigned short* __restrict__ out, int oy, int ox, int tz,
                                                          int ty, int tx, int ablate) {
    extern __shared__ u32x4_t lds[];
    constexpr int ES = 2;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int nw = (int)(blockDim.x >> 6);
    // XCD-aware order: workgroups are dealt round-robin to the 8 XCDs (each with its own L2), so XCD k takes the k-th
    // contiguous eighth of the item list (consecutive row groups of a plane).
    const int wg = (blockIdx.x & 7) * (gridDim.x >> 3) + (blockIdx.x >> 3);
    const int wi = wg * nw + wave;
    if (wi >= nitems) return;
    const LItem it = items[wi];
    const int zc = it.z, y0 = it.y, y1 = it.y1, R = it.R, ncells = it.ncells, nrows = y1 - y0;
    u32x4_t* mylds = lds + wave * wave_segs;
    XPar* xpar = reinterpret_cast<XPar*>(mylds);
    Node* nodes = reinterpret_cast<Node*>(mylds + it.nsv * 2);        // [s][R]
    u32x4_t* bufs = mylds + tab_segs;

    CellTab CT;
    ViewTab VT;
    {
        const unsigned int* cw = reinterpret_cast<const unsigned int*>(cells + it.cell0);
        const unsigned int* vw = reinterpret_cast<const unsigned int*>(svdma + it.view0);
        {
            const int e = lane >> 1, comp = lane & 1;
#pragma unroll
            for (int j = 0; j < 16; ++j) CT.r[j] = (e < ncells) ? cw[e * 32 + 2 * j + comp] : 0u;
        }
        {
            const int e = lane >> 2, comp = lane & 3;
#pragma unroll
            for (int j = 0; j < 2; ++j) VT.r[j] = (e < it.nsv) ? vw[e * 8 + 4 * j + comp] : 0u;
        }
    }
    const int rsh = (R >= 8) ? 3 : (R >= 4) ? 2 : (R >= 2) ? 1 : 0;

    // ---- DMA of one cell: every (view, unit) block, 64 flattened (row, segment) pairs = 1 KiB per instruction ----
    auto dma_cell = [&](int ci, u32x4_t* buf) __attribute__((always_inline)) {
        const int nv = cell_field<CF_NVCLS>(CT, ci) & 0xff;
        if (nv == 0) return;
        const int nseg = cell_field<CF_NSEG>(CT, ci), nunits = cell_field<CF_NUNITS>(CT, ci), lsh = cell_field<CF_LSH>(CT, ci);
        const int rpu = 64 >> lsh;                                   // rows per unit
        const int ids = cell_field<CF_IDS>(CT, ci);
        const int r = lane >> lsh, sg = lane & ((1 << lsh) - 1);
        const bool seg_ok = sg < nseg;
        for (int v = 0; v < nv; ++v) {
            const int s = (ids >> (4 * v)) & 15;
            const unsigned long long data = ((unsigned long long)(unsigned)view_field<VF_DHI>(VT, s) << 32) | (unsigned)view_field<VF_DLO>(VT, s);
            const int nbytes = view_field<VF_NBYTES>(VT, s), sy2 = view_field<VF_SY2>(VT, s);
            const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)data, 0, nbytes, 0x00020000);
            const int base = cell_base(CT, ci, v) + zc * view_field<VF_SZ2>(VT, s) + y0 * sy2;     // voxel (zc, y0, x0)
            // a vector load that is not entirely inside the slab comes back as 0: windows touching its first / last bytes are
            // fetched element by element (first / last rows of a slab only) -- checked per lane only when the cell's block
            // comes near the ends of the slab at all
            const bool near_ends = (base < 0) || (base + (nrows - 1) * sy2 + nseg * 16 > nbytes);
            u32x4_t* dst = buf + v * nunits * 64;
            for (int k = 0; k < nunits; ++k) {
                const int row = k * rpu + r;
                const int vo = base + min(row, nrows - 1) * sy2 + sg * 16;
                const bool active = seg_ok && row < R;
                bool str = false;
                if (near_ends) str = active && ((vo < 0 && vo + 16 > 0) || (vo < nbytes && vo + 16 > nbytes));
                if (active && !str && !(ablate & 1)) __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lds_ptr_t)(dst + k * 64), 16, vo, 0, 0, 0);
                if (near_ends && __any(str)) {
                    if (str) {
                        unsigned int e[kRV];
#pragma unroll
                        for (int j = 0; j < kRV; ++j) e[j] = (unsigned short)__builtin_amdgcn_raw_buffer_load_b16(rsrc, vo + j * ES, 0, 0);
                        u32x4_t w;
                        w.x = e[0] | (e[1] << 16); w.y = e[2] | (e[3] << 16); w.z = e[4] | (e[5] << 16); w.w = e[6] | (e[7] << 16);
                        dst[k * 64 + lane] = w;
                    }
                }
            }
        }
    };

    dma_cell(0, bufs);
    if (ncells > 1) dma_cell(1, bufs + buf_segs);
    // ---- weight tables of the item (while the first DMAs are in flight): lane (s, row) ----
    for (int idx = lane; idx < (it.nsv << rsh); idx += 64) {
        const int s = idx >> rsh, row = idx & (R - 1);
        const TrView& V = sviews[it.view0 + s];
        const int yr = y0 + min(row, nrows - 1);
        float G1, dG;
        bool inside;
        row_nodes(V, zc, yr, G1, dG, inside);
        const bool zy_ok = (zc >= V.lo[0]) && (zc <= V.hi[0]) && (yr >= V.lo[1]) && (yr <= V.hi[1]);
        Node nd;
        nd.G1 = G1; nd.dG = dG; nd.flags = (inside ? 1 : 0) | (zy_ok ? 2 : 0); nd.pad = 0;
        nodes[(s << rsh) + row] = nd;
        if (row == 0) {
            XPar xp;
            xp.ilo = V.sup_ilo[2]; xp.ihi = V.sup_ihi[2]; xp.flo = V.sup_flo[2]; xp.fhi = V.sup_fhi[2];
            xp.kx = V.sup_k[2]; xp.lo2 = V.lo[2]; xp.hi2 = V.hi[2]; xp.pad = 0;
            xpar[s] = xp;
        }
    }

    // Per cell: arithmetic of all its units into registers -> wait for everything issued one compute phase ago (the DMA of
    // cell ci + 1 and the stores of cell ci - 1) -> DMA of cell ci + 2 into the buffer just consumed -> stores of cell ci.
    // (Waiting right after issuing the stores would expose a full store round trip per cell.)
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");          // cells 0 and 1 have arrived, the tables are written
    constexpr int kMaxU = 4;                                             // units per cell (8 rows, >= 2 rows per unit)
    for (int ci = 0; ci < ncells; ++ci) {
        u32x4_t* buf = bufs + (ci & 1) * buf_segs;
        u32x4_t res[kMaxU];
        bool res_ok[kMaxU];
#pragma unroll
        for (int k = 0; k < kMaxU; ++k) { res[k] = u32x4_t{0u, 0u, 0u, 0u}; res_ok[k] = false; }

        // ---- arithmetic of cell ci out of LDS ----
        const int nvcls = cell_field<CF_NVCLS>(CT, ci);
        const int nv = nvcls & 0xff, cls = nvcls >> 8;
        const int x0 = cell_field<CF_X0>(CT, ci), x1 = cell_field<CF_X1>(CT, ci);
        const int nseg = cell_field<CF_NSEG>(CT, ci), nunits = cell_field<CF_NUNITS>(CT, ci);
        const int masks = cell_field<CF_MASKS>(CT, ci), sides = cell_field<CF_SIDES>(CT, ci);
        const int lsh = cell_field<CF_LSH>(CT, ci), ids = cell_field<CF_IDS>(CT, ci);
        const int rpu = 64 >> lsh;
        const int vstride = nunits * 64;
        const int full = (1 << nv) - 1;
        const int allone_mask = masks & 0xff, partial_mask = (masks >> 16) & 0xff;
        const bool allpos = (masks >> 15) & 1;
        const bool lean = ((allone_mask & full) == full) && !partial_mask;   // every view in bounds with weight 1: plain mean
        const float need = (nv == 1) ? 3e-4f : 1.f;   // a voxel seen by ONE view only needs a weight that does not round to 0
        const int r = lane >> lsh, sg = lane & ((1 << lsh) - 1);
        const int xq = x0 + kRV * sg;
        const int nvx = min(x1 - xq, kRV);
        const bool seg_ok = sg < nseg;
        unsigned short* orow = out + ((long long)(zc - tz) * oy + (y0 - ty)) * (long long)ox + (xq - tx);      // row y0 of the item
#pragma unroll
        for (int k = 0; k < kMaxU; ++k) {
            if (k >= nunits || (ablate & 2)) break;
            const int row = k * rpu + r;
            const bool row_ok = seg_ok && row < nrows && !(ablate & 4);
            res_ok[k] = row_ok;
            const int rowc = min(row, nrows - 1);
            const u32x4_t* blk = buf + k * 64 + lane;

            if (cls == 0) {
                continue;                      // res[k] is already 0
            }
            if (cls == 1 || (ablate & 8)) {      // one full view with a weight > 0 everywhere: the result is the value
                res[k] = blk[0];
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
                res[k] = pack8(q);
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
                const int s = (ids >> (4 * v)) & 15;
                const Node nd = nodes[(s << rsh) + rowc];
                const XPar xp = xpar[s];
                const int side = (sides >> (2 * v)) & 3;
                float w[kRV];
                if (side != 0) {
                    // the cell lies on the first support interval next to ONE x border of this (full) view: the profile is
                    // W = distance * kx * G1, monotonic along the row
                    const float slope = xp.kx * nd.G1;
                    const float d0 = (side == 1) ? (float)(xq - xp.ilo) - xp.flo : (float)(xp.ihi - xq) - xp.fhi;
                    const float st = (side == 1) ? 1.f : -1.f;
                    const float Wmin = ((side == 1) ? d0 : d0 - 7.f) * slope;
                    if (!__any(!(Wmin >= need))) unit = true;
                    else {
#pragma unroll
                        for (int j = 0; j < kRV; ++j) w[j] = blend_ramp_nb(fmaf(st, (float)j, d0) * slope);
                    }
                    if (unit) {
#pragma unroll
                        for (int j = 0; j < kRV; ++j) { num[j] += val[j]; den[j] += 1.f; }
                    } else {     // (side flags are only set on cells whose views all have a strictly positive weight)
                        all_unit = false;
#pragma unroll
                        for (int j = 0; j < kRV; ++j) { num[j] = fmaf(w[j], val[j], num[j]); den[j] += w[j]; }
                    }
                    continue;
                }
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
            res[k] = pack8(q);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (ci + 2 < ncells) dma_cell(ci + 2, buf);
#pragma unroll
        for (int k = 0; k < kMaxU; ++k) {
            if (k >= nunits) break;
            const int rowc = min(k * rpu + r, nrows - 1);
            if (res_ok[k]) store8_packed(orow + (long long)rowc * ox, res[k], nvx);
        }
    }
}

struct LdsPlan {
    unsigned long long hash = 0;
    bool valid = false;
    int class_count[3] = {0, 0, 0};       // work items (wavefronts) per LDS size class
    int class_tab[3] = {0, 0, 0}, class_buf[3] = {0, 0, 0}, class_nw[3] = {4, 4, 4};   // table / cell-buffer segments, wavefronts per workgroup
    size_t off_dma = 0, off_cells = 0, off_items = 0;
    double build_ms = 0.0;
};
LdsPlan g_lds_plan[MVS_MAX_DEVICES * MVS_MAX_LANES];
double g_lds_last_plan_ms[MVS_MAX_DEVICES * MVS_MAX_LANES];

struct HStrip { int z0, z1, y0, y1, cell0, ncells, R, view0, nsv, cls; };

}  // namespace

double mvs_rowlds_last_plan_ms(MvsContext* c) { return g_lds_last_plan_ms[mvs_ctx_index(c->device)]; }

// Returns MVS_OK and sets *done = true when the chunk was fused by the LDS-staged row kernel; *done = false means the
// caller must use another path (not uint16, a view with more than one tap, more than 8 views on one cell, more than 16
// views or 32 cells on one strip, a cell whose single row does not fit into the largest cell buffer).
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
        const int blk_limit[3] = {4, 8, 16};         // cell buffer of a class, in units of 64 segments (KiB)

        std::vector<HStrip> strips;
        std::vector<LCell> cells;
        std::vector<TrView> sviews;          // per strip: copies of the records of its views (strip-local slots)
        std::vector<SvDma> svdma;            // ... and their addressing records
        std::vector<int> zviews, svs, slot_of(n_views);
        int class_tab[3] = {0, 0, 0}, class_buf[3] = {0, 0, 0};
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
                for (size_t k = 0; k < svs.size(); ++k) {
                    const TrView& V = htr[svs[k]];
                    slot_of[svs[k]] = (int)k;
                    sviews.push_back(V);
                    SvDma d;
                    memset(&d, 0, sizeof(d));
                    d.data_lo = (unsigned int)(V.data & 0xffffffffull);
                    d.data_hi = (unsigned int)(V.data >> 32);
                    d.nbytes = (int)V.span * 2;
                    d.sy2 = V.stride_y * 2;
                    d.sz2 = V.stride_z * 2;
                    svdma.push_back(d);
                }
                axis_breakpoints(htr, svs, 2, t[2], o[2], &px);
                {   // Cells are cut on 8-voxel boundaries into chunks whose segment count fits a power-of-two lane layout:
                    // at most 32 segments where one view contributes (copy chunks: 2 rows per unit), at most 8 segments where
                    // several do (8 rows per unit; the two ends of an overlap zone -- where different views ramp -- then fall
                    // into different chunks, so the host's "weight 1 everywhere" flags spare most weight evaluations).
                    std::vector<int> cut;
                    for (size_t ix = 0; ix + 1 < px.size(); ++ix) {
                        // several views of which one has an x border within a quarter tile of the cell: x ramps -> 8 segments
                        int nvc = 0;
                        bool xramp = false;
                        for (int v : svs)
                            if (htr[v].lo[2] < px[ix + 1] && htr[v].hi[2] >= px[ix]) {
                                ++nvc;
                                const int quarter = (htr[v].hi[2] - htr[v].lo[2] + 1) / 4;
                                if (px[ix] - htr[v].lo[2] < quarter || htr[v].hi[2] + 1 - px[ix + 1] < quarter) xramp = true;
                            }
                        const int chunk = ((nvc >= 2 && xramp) ? 8 : 32) * kRV;
                        for (int x = px[ix]; x < px[ix + 1]; x += chunk) cut.push_back(x);
                    }
                    cut.push_back(px.back());
                    px.swap(cut);
                }
                if ((int)px.size() - 1 > kMaxStripCells) return MVS_OK;
                for (size_t ix = 0; ix + 1 < px.size(); ++ix) {
                    LCell C;
                    memset(&C, 0, sizeof(C));
                    C.x0 = px[ix]; C.x1 = px[ix + 1];
                    int nv = 0;
                    bool positive_full = false, all_positive = true;
                    int side_of[kMaxCV] = {0, 0, 0, 0, 0, 0, 0, 0};
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
                        if (full && !unit) {
                            // is the whole cell (plus the 7 voxels a lane may run past its end) on the first support interval
                            // next to ONE x border of the view?  Then W = distance * kx * G1 along the row.
                            const double lo_node = (double)htr[v].sup_ilo[2] + (double)htr[v].sup_flo[2];
                            const double hi_node = (double)htr[v].sup_ihi[2] - (double)htr[v].sup_fhi[2];
                            const double k = (double)htr[v].sup_k[2];
                            const double xa = C.x0, xb = C.x1 - 1 + 7;
                            if ((xb - lo_node) * k <= 1.0 - 1e-6 && (xb - lo_node) <= (hi_node - xb)) side_of[nv] = 1;
                            else if ((hi_node - xa) * k <= 1.0 - 1e-6 && (hi_node - xa) <= (xa - lo_node)) side_of[nv] = 2;
                        }
                        C.base[nv] = (htr[v].io[0] * htr[v].stride_z + htr[v].io[1] * htr[v].stride_y + C.x0 + htr[v].io[2]) * 2;
                        C.ids_packed |= slot_of[v] << (4 * nv);
                        ++nv;
                    }
                    if (nv > 0 && all_positive) {
                        C.masks |= 1 << 15;
                        for (int q = 0; q < nv; ++q) C.sides |= side_of[q] << (2 * q);
                    }
                    const int cls = nv == 0 ? 0 : (nv == 1 && positive_full) ? 1 : 2;
                    C.nv_cls = nv | (cls << 8);
                    C.nseg = (C.x1 - C.x0 + kRV - 1) / kRV;
                    C.lsh = C.nseg <= 8 ? 3 : C.nseg <= 16 ? 4 : C.nseg <= 32 ? 5 : 6;
                    cells.push_back(C);
                }
                S.ncells = (int)cells.size() - S.cell0;
                // rows per wavefront: as many (8, 4, 2, 1) as keep the largest (views x units) block of a cell within the
                // smallest cell buffer; larger buffers (fewer wavefronts per CU) only when 4 rows do not fit otherwise
                int R = 8, cls = -1, blk = 0;
                for (int k = 0; k < 3 && cls < 0; ++k)
                    for (R = 8; R >= 1; R >>= 1) {
                        blk = 0;
                        for (int ci = S.cell0; ci < S.cell0 + S.ncells; ++ci) {
                            const int rpu = 64 >> cells[ci].lsh;
                            blk = std::max(blk, (cells[ci].nv_cls & 0xff) * ((R + rpu - 1) / rpu));
                        }
                        if (blk <= blk_limit[k] && (R >= 4 || k == 2)) { cls = k; break; }
                    }
                if (cls < 0) return MVS_OK;
                for (int ci = S.cell0; ci < S.cell0 + S.ncells; ++ci) {
                    const int rpu = 64 >> cells[ci].lsh;
                    cells[ci].nunits = (R + rpu - 1) / rpu;
                    if (cells[ci].nunits > 4) return MVS_OK;      // the kernel keeps at most 4 unit results in registers
                }
                S.R = R;
                S.cls = cls;
                class_buf[cls] = std::max(class_buf[cls], std::max(blk, 1) * 64);
                class_tab[cls] = std::max(class_tab[cls], S.nsv * 2 + S.nsv * R);
                strips.push_back(S);
            }
        }
        if (cells.size() > (1u << 22)) return MVS_OK;
        // work items (one per wavefront), z-major per class: consecutive wavefronts write consecutive row groups of a plane
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
        const size_t dbytes = (svdma.size() * sizeof(SvDma) + 255) / 256 * 256;
        const size_t cbytes = (cells.size() * sizeof(LCell) + 255) / 256 * 256;
        const size_t ibytes = nitems * sizeof(LItem);
        const size_t total = vbytes + dbytes + cbytes + ibytes;
        char* hbuf = (char*)mvs_pinned_slot(c, 1, total + 256);   // slot 0 holds the view parameters still in flight
        if (!hbuf) return MVS_ERR_HIP;
        pc.valid = false;
        dbuf = (char*)mvs_scratch(c, 14, total + 256);
        if (!dbuf) return MVS_ERR_HIP;
        memcpy(hbuf, sviews.data(), sviews.size() * sizeof(TrView));
        memcpy(hbuf + vbytes, svdma.data(), svdma.size() * sizeof(SvDma));
        memcpy(hbuf + vbytes + dbytes, cells.data(), cells.size() * sizeof(LCell));
        size_t cur = vbytes + dbytes + cbytes;
        for (int k = 0; k < 3; ++k) {
            pc.class_count[k] = (int)items_by_class[k].size();
            pc.class_tab[k] = class_tab[k];
            pc.class_buf[k] = class_buf[k];
            // wavefronts per workgroup: at most 4, as few as keep a workgroup's LDS within a third of the CU's 160 KiB
            const int wave_bytes = (class_tab[k] + 2 * class_buf[k]) * 16;
            pc.class_nw[k] = std::max(1, std::min(4, wave_bytes ? (53 * 1024) / wave_bytes : 4));
            if (pc.class_nw[k] == 3) pc.class_nw[k] = 2;
            memcpy(hbuf + cur, items_by_class[k].data(), items_by_class[k].size() * sizeof(LItem));
            cur += items_by_class[k].size() * sizeof(LItem);
        }
        MVS_HIP_TRY(c, hipMemcpyAsync(dbuf, hbuf, total, hipMemcpyHostToDevice, c->stream));
        mvs_pinned_mark(c, 1);
        pc.hash = h;
        pc.off_dma = vbytes;
        pc.off_cells = vbytes + dbytes;
        pc.off_items = vbytes + dbytes + cbytes;
        pc.valid = true;
        pc.build_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_begin).count();
        g_lds_last_plan_ms[mvs_ctx_index(c->device)] = pc.build_ms;
        if (getenv("MVS_PLAN_STATS")) {
            fprintf(stderr, "[mvs rowlds plan] strips %zu cells %zu strip-views %zu; per class items / table segs / buffer segs / waves per group:", strips.size(), cells.size(), sviews.size());
            for (int k = 0; k < 3; ++k) fprintf(stderr, " %d / %d / %d / %d;", pc.class_count[k], pc.class_tab[k], pc.class_buf[k], pc.class_nw[k]);
            fprintf(stderr, " %.2f ms\n", pc.build_ms);
        }
    }
    static bool attr_set[MVS_MAX_DEVICES] = {false};
    if (!attr_set[mvs_hip_device(c->device)]) {
        MVS_HIP_TRY(c, hipFuncSetAttribute((const void*)fuse_rowlds_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        attr_set[mvs_hip_device(c->device)] = true;
    }
    const TrView* dsviews = (const TrView*)dbuf;
    const SvDma* dsvdma = (const SvDma*)(dbuf + pc.off_dma);
    const LCell* dcells = (const LCell*)(dbuf + pc.off_cells);
    const LItem* ditems = (const LItem*)(dbuf + pc.off_items);
    MVS_HIP_TRY(c, hipEventRecord(c->ev_start, c->stream));   // kernel time only (the plan is host work, cached per geometry)
    int item0 = 0;
    for (int k = 0; k < 3; ++k) {
        const int cnt = pc.class_count[k];
        if (!cnt) continue;
        const int nw = pc.class_nw[k];
        const int wave_segs = pc.class_tab[k] + 2 * pc.class_buf[k];
        const int ngroups = (cnt + nw - 1) / nw;
        const int nblocks = (ngroups + 7) / 8 * 8;   // multiple of 8: see the XCD mapping in the kernel
        hipLaunchKernelGGL(fuse_rowlds_kernel, dim3(nblocks), dim3(64 * nw), (size_t)wave_segs * 16 * nw, c->stream, dsviews, dsvdma, dcells,
                           ditems + item0, cnt, wave_segs, pc.class_tab[k], pc.class_buf[k], (unsigned short*)dout, o[1], o[2],
                           t[0], t[1], t[2], c->ablate);
        item0 += cnt;
    }
    MVS_HIP_TRY(c, hipGetLastError());
    *done = true;
    return MVS_OK;
}
