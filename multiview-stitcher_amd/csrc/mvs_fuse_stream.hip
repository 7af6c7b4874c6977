// mvs_fuse_stream.hip -- streaming row kernel of the translation fast path (uint16 tiles, integer offsets, weighted-average
// fusion with blending weights; gfx950).
//
// The region kernels (mvs_fuse_region.hip) cut the chunk into boxes with a constant view list and run one kernel per view-count
// class; their time is set by the boxes next to view borders (profiles/round2_summary.md section 3).  This kernel has no boxes
// and no classes.  A wavefront owns kRows complete output rows of one plane of a STRIP (a z cell x y cell of the region
// planner's grid in which every view that reaches into it covers all rows) and walks them in UNITS of 512 voxels (64 lanes x 8).
// Per unit it loops over the unit's ENTRIES (one per view reaching into it) and evaluates, for every voxel and view, the same
// arithmetic as the general path of the region kernels (reference: weights.py:391-511, 325-345, fusion/_core.py:61-94):
//     u    folded x coordinate of the voxel in the view's support grid   -> per-view table over the chunk's x range (host)
//     G1, dG  row nodes of the view at (z, y)                             -> per-view table over its z / y box (prologue kernel)
//     W = u <= 1 ? u G1 : G1 + (u - 1) dG,   w = blend_ramp(W),   num += w v,  den += w
// and, for the exactness rule "a single contributor yields its value", one packed accumulator cv += (w > 0) ? v + 65536 : 0
// (count in the high part, the value in the low part when the count is 1).  Everything is uniform: no branches on geometry, all
// view constants come from flat arrays through scalar loads, whole rows are read once and written once with 16-byte accesses.
#include "mvs_fuse_stream.h"

#include <algorithm>
#include <cstdlib>
#include <cstring>

namespace {

constexpr int kRV = 8;        // voxels per lane
constexpr int kRows = 2;      // output rows per wavefront
constexpr int kUnit = 64 * kRV;

struct SEntry {               // 64 bytes: one view reaching into one unit of one strip (one scalar load)
    unsigned int data_lo, data_hi;
    int nbytes;               // slab bytes (buffer range)
    int sy2, sz2;             // row / plane pitch in bytes
    int base;                 // byte offset of chunk voxel (0, 0, unit x0) in the slab
    int node_off;             // index of the view's first row node (float2)
    int zlo, ylo, ny;         // the node table covers z >= zlo, y in [ylo, ylo + ny)
    int x0;                   // chunk index of the unit's first voxel
    int flags;                // bit 0: first entry of its unit, bit 1: last entry of its unit
    int tab_off;              // index (float2) of the view's x table entry for the unit's first voxel
    int pad[3];
};
static_assert(sizeof(SEntry) == 64, "SEntry layout");
struct SStripD { int ent0, nent, pad0, pad1; };
struct SItem { int strip, z, y0, nrows; };

typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4_a2 __attribute__((ext_vector_type(4), aligned(2)));
typedef float f4_t __attribute__((ext_vector_type(4)));

// row nodes of every view over its valid z / y box: (G1, G2 - G1), or (0, 0) where the row lies outside the support
__global__ __launch_bounds__(256) void stream_nodes_kernel(const TrView* __restrict__ views, const int* __restrict__ node_off, int n_views,
                                                           float2* __restrict__ nodes) {
    const int v = blockIdx.y;
    if (v >= n_views) return;
    const TrView V = views[v];
    const int nz = V.hi[0] - V.lo[0] + 1, ny = V.hi[1] - V.lo[1] + 1;
    if (nz <= 0 || ny <= 0 || node_off[v] < 0) return;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < nz * ny; i += gridDim.x * blockDim.x) {
        float G1 = 0.f, G2 = 0.f;
        const bool in = tr_row_nodes(V, V.lo[0] + i / ny, V.lo[1] + i % ny, G1, G2);
        nodes[(size_t)node_off[v] + i] = in ? make_float2(G1, G2 - G1) : make_float2(0.f, 0.f);
    }
}

// blend_ramp_nb of mvs_fuse_tr.h, value for value, with a shorter tail: (c + 1) / 2 as one fma (scaling by 2 is exact) and the
// branch "x >= 1 -> 1" as max(w, 1 + (xc - 1) * 2^25) (1 for xc == 1, <= -1 below)
__device__ __forceinline__ float stream_ramp(float x) {
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

// the two rows of a wavefront in one view: 16 bytes per lane and row
template <int ABL>
__device__ __forceinline__ void stream_fetch(const SEntry& E, const SItem& it, int lane, u32x4_t (&raw)[kRows]) {
    const unsigned long long dptr = ((unsigned long long)E.data_hi << 32) | E.data_lo;
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)dptr, 0, E.nbytes, 0x00020000);
    const int vo0 = E.base + it.z * E.sz2 + it.y0 * E.sy2 + lane * (kRV * 2);
#pragma unroll
    for (int r = 0; r < kRows; ++r) {
        const int vo = vo0 + min(r, it.nrows - 1) * E.sy2;
        if (ABL == 2) { raw[r] = u32x4_t{(unsigned)vo, 1u, 2u, 3u}; continue; }
        raw[r] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, vo, 0, 0);
        // a vector load that is not entirely inside the slab comes back as 0: windows touching its first / last bytes
        // are fetched element by element (first / last row of a slab only)
        const bool str = (vo < 0 && vo + 16 > 0) || (vo < E.nbytes && vo + 16 > E.nbytes);
        if (__any(str)) {
            if (str) {
                unsigned int el[kRV];
#pragma unroll
                for (int j = 0; j < kRV; ++j) el[j] = (unsigned short)__builtin_amdgcn_raw_buffer_load_b16(rsrc, vo + 2 * j, 0, 0);
                raw[r].x = el[0] | (el[1] << 16); raw[r].y = el[2] | (el[3] << 16);
                raw[r].z = el[4] | (el[5] << 16); raw[r].w = el[6] | (el[7] << 16);
            }
        }
    }
}
__device__ __forceinline__ void stream_nodes(const SEntry& E, const SItem& it, const float2* __restrict__ nodes, float2 (&nd)[kRows]) {
#pragma unroll
    for (int r = 0; r < kRows; ++r)
        nd[r] = nodes[(size_t)E.node_off + (size_t)(it.z - E.zlo) * E.ny + (it.y0 + min(r, it.nrows - 1) - E.ylo)];
}

template <int ABL>
__global__ __launch_bounds__(256) void fuse_stream_kernel(const SItem* __restrict__ items, int nitems, const SStripD* __restrict__ strips,
                                                          const SEntry* __restrict__ entries, const float2* __restrict__ nodes,
                                                          const float2* __restrict__ xtab, unsigned short* __restrict__ out, int oy, int ox, int tz, int ty, int tx) {
    __shared__ float s_w[4][64];
    const int lane = threadIdx.x & 63;
    // XCD-aware order (as in the region kernels): XCD k walks the k-th contiguous eighth of the item list
    const int wg = (blockIdx.x & 7) * (gridDim.x >> 3) + (blockIdx.x >> 3);
    const int wi = __builtin_amdgcn_readfirstlane(wg * 4 + (int)(threadIdx.x >> 6));
    if (wi >= nitems) return;
    const SItem it = items[wi];
    const SStripD S = strips[it.strip];
    const SEntry* ent = entries + S.ent0;
    // software pipeline: the record of entry e + 2 and the voxels of entry e + 1 are in flight while entry e is evaluated
    SEntry E = ent[0];
    u32x4_t rawN[kRows];
    float2 ndN[kRows];
    stream_fetch<ABL>(E, it, lane, rawN);
    stream_nodes(E, it, nodes, ndN);
    SEntry En = ent[min(1, S.nent - 1)];
    float num[kRows][kRV], den[kRows][kRV], cnt[kRows][kRV];
    for (int e = 0; e < S.nent; ++e) {
        u32x4_t raw[kRows];
        float2 nd[kRows];
#pragma unroll
        for (int r = 0; r < kRows; ++r) { raw[r] = rawN[r]; nd[r] = ndN[r]; }
        const int x0 = E.x0, flags = E.flags;
        // x part of the profile (shared by the rows): t1 = min(u, 1), t2 = max(u - 1, 0) of the folded coordinate u, both 0
        // outside the view's box -- a table per view over the chunk's x range (host), so that W = t1 * G1 + t2 * dG
        const f4_t* tp = reinterpret_cast<const f4_t*>(xtab + (size_t)E.tab_off) + 4 * lane;
        const f4_t ta = tp[0], tb = tp[1], tc = tp[2], td = tp[3];
        E = En;
        if (e + 1 < S.nent) {
            stream_fetch<ABL>(E, it, lane, rawN);
            stream_nodes(E, it, nodes, ndN);
        }
        En = ent[min(e + 2, S.nent - 1)];
        if (flags & 1) {
#pragma unroll
            for (int r = 0; r < kRows; ++r)
#pragma unroll
                for (int j = 0; j < kRV; ++j) { num[r][j] = 0.f; den[r][j] = 0.f; cnt[r][j] = 0.f; }
        }
        const int x = x0 + kRV * lane;                         // chunk index of the lane's first voxel
        const float t1[kRV] = {ta.x, ta.z, tb.x, tb.z, tc.x, tc.z, td.x, td.z};
        const float t2[kRV] = {ta.y, ta.w, tb.y, tb.w, tc.y, tc.w, td.y, td.w};
        // profile W of the 2 x 8 voxels.  Along x it rises from 0 at the view's border to >= 1 within a few voxels (the blending
        // width) and stays there, so in most rows only the lanes at a border hold a value strictly between 0 and 1 and need the
        // ramp polynomial; everywhere else the weight is clamp(W) = 0 or 1.  W is unimodal along a row (folded coordinate,
        // non-decreasing profile), so a lane's first and last voxel decide.  The <= 64 values of the needy lanes of both rows are
        // spread over the wavefront through LDS: one polynomial per lane instead of 16.
        float w[kRows][kRV];                                   // clamp(W), then the weight
        bool needy[kRows];
#pragma unroll
        for (int r = 0; r < kRows; ++r) {
            const float G1 = nd[r].x, dG = nd[r].y;
#pragma unroll
            for (int j = 0; j < kRV; ++j) {
                w[r][j] = (ABL == 4) ? 1.f : __builtin_amdgcn_fmed3f(fmaf(t2[j], dG, t1[j] * G1), 0.f, 1.f);
            }
            needy[r] = fminf(w[r][0], w[r][kRV - 1]) < 1.f && fmaxf(w[r][0], w[r][kRV - 1]) > 0.f;
        }
        if (ABL != 1 && ABL != 4) {
            const unsigned long long m0 = __ballot(needy[0]), m1 = __ballot(needy[1]);
            const int n0 = __popcll(m0), n1 = __popcll(m1);
            if (n0 + n1 > 8) {
#pragma unroll
                for (int r = 0; r < kRows; ++r)
#pragma unroll
                    for (int j = 0; j < kRV; ++j) w[r][j] = stream_ramp(w[r][j]);
            } else if (n0 + n1 > 0) {
                float* L = s_w[threadIdx.x >> 6];
                const int i0 = __builtin_amdgcn_mbcnt_hi((unsigned int)(m0 >> 32), __builtin_amdgcn_mbcnt_lo((unsigned int)m0, 0u));
                const int i1 = n0 + __builtin_amdgcn_mbcnt_hi((unsigned int)(m1 >> 32), __builtin_amdgcn_mbcnt_lo((unsigned int)m1, 0u));
#pragma unroll
                for (int r = 0; r < kRows; ++r)
                    if (needy[r]) {
                        f4_t* q = reinterpret_cast<f4_t*>(L + (r ? i1 : i0) * kRV);
                        q[0] = f4_t{w[r][0], w[r][1], w[r][2], w[r][3]};
                        q[1] = f4_t{w[r][4], w[r][5], w[r][6], w[r][7]};
                    }
                __builtin_amdgcn_wave_barrier();
                L[lane] = stream_ramp(L[lane]);
                __builtin_amdgcn_wave_barrier();
#pragma unroll
                for (int r = 0; r < kRows; ++r)
                    if (needy[r]) {
                        const f4_t* q = reinterpret_cast<const f4_t*>(L + (r ? i1 : i0) * kRV);
                        const f4_t qa = q[0], qb = q[1];
                        w[r][0] = qa.x; w[r][1] = qa.y; w[r][2] = qa.z; w[r][3] = qa.w;
                        w[r][4] = qb.x; w[r][5] = qb.y; w[r][6] = qb.z; w[r][7] = qb.w;
                    }
                __builtin_amdgcn_wave_barrier();
            }
        }
#pragma unroll
        for (int r = 0; r < kRows; ++r) {
            const unsigned int w4[4] = {raw[r].x, raw[r].y, raw[r].z, raw[r].w};
#pragma unroll
            for (int j = 0; j < kRV; ++j) {
                const float v = (float)((j & 1) ? (w4[j >> 1] >> 16) : (w4[j >> 1] & 0xffffu));
                num[r][j] = fmaf(w[r][j], v, num[r][j]);
                den[r][j] += w[r][j];
                cnt[r][j] += __builtin_amdgcn_fmed3f(w[r][j] * 0x1p100f, 0.f, 1.f);     // views with a positive weight (w >= 2^-25 or 0)
            }
        }
        const int nvalid = min(max(tx + ox - x, 0), kRV);      // the row ends at chunk index tx + ox
        if ((flags & 2) && nvalid > 0 && (ABL != 3 || den[0][0] == -7.f)) {
#pragma unroll
            for (int r = 0; r < kRows; ++r) {
                if (r >= it.nrows) break;
                unsigned int q[kRV];
#pragma unroll
                for (int j = 0; j < kRV; ++j) {
                    // no contributor: 0 (0 * inf = NaN -> 0); one: its value, w v / w to the nearest integer (w / w == 1 in the
                    // reference; the quotient is off by < 0.02); several: the weighted mean
                    float o = fmaxf(num[r][j] * __builtin_amdgcn_rcpf(den[r][j]), 0.f);
                    o = (cnt[r][j] < 1.5f) ? __builtin_rintf(o) : o;
                    q[j] = (unsigned int)(int)o;
                }
                unsigned short* p = out + ((long long)(it.z - tz) * oy + (it.y0 + r - ty)) * (long long)ox + (x - tx);
                if (nvalid >= kRV) {
                    u32x4_a2 o4;
                    o4.x = q[0] | (q[1] << 16); o4.y = q[2] | (q[3] << 16); o4.z = q[4] | (q[5] << 16); o4.w = q[6] | (q[7] << 16);
                    __builtin_nontemporal_store(o4, reinterpret_cast<u32x4_a2*>(p));
                } else {
#pragma unroll
                    for (int j = 0; j < kRV; ++j)
                        if (j < nvalid) p[j] = (unsigned short)q[j];
                }
            }
        }
    }
}

struct StreamCache {
    unsigned long long hash = 0;
    bool valid = false;
    int nitems = 0;
    size_t off_strips = 0, off_entries = 0, off_tab = 0, off_noff = 0, off_nodes = 0;
};
StreamCache g_stream[MVS_MAX_DEVICES * MVS_MAX_LANES];

}  // namespace

bool mvs_stream_view_ok(const TrView& V) {
    if (V.fw[0] > 0.f || V.fw[1] > 0.f || V.fw[2] > 0.f) return false;       // one tap per voxel only
    if (V.lo[0] > V.hi[0] || V.lo[1] > V.hi[1] || V.lo[2] + 16 > V.hi[2]) return false;   // (>= 2 lanes wide: see the needy-lane test)
    if (V.span * 2 > 0x7fffffffLL || (long long)V.stride_z * 2 > 0x3fffffffLL ||
        std::llabs(((long long)V.io[0] * V.stride_z + (long long)V.io[1] * V.stride_y + V.io[2]) * 2) > 0x2fffffffLL) return false;
    return true;
}

int mvs_fuse_stream(MvsContext* c, const TrView* htr, const TrView* dtr, int n_views, const std::vector<StreamStrip>* strips_in,
                    unsigned long long hash, void* dout, const int o[3], const int t[3], hipStream_t stream) {
    StreamCache& sc = g_stream[mvs_ctx_index(c->device)];
    char* dbuf = nullptr;
    if (!strips_in) {                                   // same geometry as the previous call: the tables are on the device
        if (!(sc.valid && sc.hash == hash && (sc.nitems == 0 || c->dev[15].ptr))) return mvs_fail(c, MVS_ERR_INVALID_ARG, "stream plan missing");
        dbuf = (char*)c->dev[15].ptr;
    } else {
        sc.valid = false;
        const std::vector<StreamStrip>& strips = *strips_in;
        std::vector<SStripD> hs;
        std::vector<SEntry> he;
        std::vector<SItem> hi;
        const int x_begin = t[2], x_end = t[2] + o[2];
        // views used by some strip get a node table
        std::vector<int> node_off((size_t)n_views, -1), used((size_t)n_views, 0);
        for (auto& S : strips)
            for (int v : S.views) used[v] = 1;
        size_t n_nodes = 0;
        const int utw = (o[2] + kUnit + 63) / 64 * 64;          // x table pitch: the chunk's x range + one unit of slack
        std::vector<int> tab_row((size_t)n_views, -1);
        int n_tab = 0;
        for (int v = 0; v < n_views; ++v)
            if (used[v]) {
                node_off[v] = (int)n_nodes;
                n_nodes += (size_t)(htr[v].hi[0] - htr[v].lo[0] + 1) * (htr[v].hi[1] - htr[v].lo[1] + 1);
                tab_row[v] = n_tab++;
                if (n_nodes > (1u << 30)) return mvs_fail(c, MVS_ERR_UNSUPPORTED, "stream plan: node table too large");
            }
        std::vector<float> htab((size_t)std::max(n_tab, 1) * utw * 2, 0.f);
        for (int v = 0; v < n_views; ++v)
            if (used[v]) {
                const TrView& V = htr[v];
                for (int x = std::max(V.lo[2], x_begin); x <= std::min(V.hi[2], x_end - 1); ++x) {
                    const float u = fold_u(x, V.sup_ilo[2], V.sup_flo[2], V.sup_ihi[2], V.sup_fhi[2], V.sup_k[2]);
                    float* q = &htab[((size_t)tab_row[v] * utw + (x - x_begin)) * 2];
                    q[0] = (u < 0.f) ? 0.f : fminf(u, 1.f);
                    q[1] = fmaxf(u - 1.f, 0.f);
                }
            }
        for (size_t si = 0; si < strips.size(); ++si) {
            const StreamStrip& S = strips[si];
            SStripD D{(int)he.size(), 0, 0, 0};
            for (int x0 = x_begin; x0 < x_end; x0 += kUnit) {
                const size_t first = he.size();
                for (int v : S.views) {                      // ascending view index
                    const TrView& V = htr[v];
                    if (!(V.lo[2] < std::min(x0 + kUnit, x_end) && V.hi[2] >= x0)) continue;
                    SEntry E;
                    memset(&E, 0, sizeof(E));
                    E.data_lo = (unsigned int)(V.data & 0xffffffffull);
                    E.data_hi = (unsigned int)(V.data >> 32);
                    E.nbytes = (int)(V.span * 2);
                    E.sy2 = V.stride_y * 2;
                    E.sz2 = V.stride_z * 2;
                    E.base = (int)(((long long)V.io[0] * V.stride_z + (long long)V.io[1] * V.stride_y + (long long)(x0 + V.io[2])) * 2);
                    E.node_off = node_off[v];
                    E.zlo = V.lo[0]; E.ylo = V.lo[1]; E.ny = V.hi[1] - V.lo[1] + 1;
                    E.x0 = x0;
                    E.tab_off = tab_row[v] * utw + (x0 - x_begin);
                    he.push_back(E);
                }
                if (he.size() == first) {                    // no view reaches into the unit: zeros
                    SEntry E;
                    memset(&E, 0, sizeof(E));
                    E.x0 = x0;                               // (nbytes 0: every load is out of range and returns 0)
                    he.push_back(E);
                }
                he[first].flags |= 1;
                he.back().flags |= 2;
            }
            D.nent = (int)he.size() - D.ent0;
            hs.push_back(D);
            for (int z = S.z0; z < S.z1; ++z)
                for (int y = S.y0; y < S.y1; y += kRows) hi.push_back(SItem{(int)si, z, y, std::min(kRows, S.y1 - y)});
        }
        sc.hash = hash;
        sc.nitems = (int)hi.size();
        if (hi.empty()) { sc.valid = true; return MVS_OK; }
        if (hi.size() > (1u << 28)) return mvs_fail(c, MVS_ERR_UNSUPPORTED, "stream plan too large");
        auto al = [](size_t v) { return (v + 255) / 256 * 256; };
        const size_t b_items = al(hi.size() * sizeof(SItem)), b_strips = al(hs.size() * sizeof(SStripD)), b_ent = al(he.size() * sizeof(SEntry) + 64),
                     b_tab = al(htab.size() * 4 + 64), b_noff = al((size_t)n_views * 4 + 64), b_nodes = al(n_nodes * 8 + 64);
        const size_t upload = b_items + b_strips + b_ent + b_tab + b_noff;
        dbuf = (char*)mvs_scratch(c, 15, upload + b_nodes);
        if (!dbuf) return MVS_ERR_HIP;
        char* hb = (char*)mvs_pinned_slot(c, 1, upload);
        if (!hb) return MVS_ERR_HIP;
        memcpy(hb, hi.data(), hi.size() * sizeof(SItem));
        memcpy(hb + b_items, hs.data(), hs.size() * sizeof(SStripD));
        memcpy(hb + b_items + b_strips, he.data(), he.size() * sizeof(SEntry));
        memcpy(hb + b_items + b_strips + b_ent, htab.data(), htab.size() * 4);
        memcpy(hb + b_items + b_strips + b_ent + b_tab, node_off.data(), (size_t)n_views * 4);
        MVS_HIP_TRY(c, hipMemcpyAsync(dbuf, hb, upload, hipMemcpyHostToDevice, c->stream));
        mvs_pinned_mark(c, 1);
        sc.off_strips = b_items; sc.off_entries = b_items + b_strips; sc.off_tab = sc.off_entries + b_ent; sc.off_noff = sc.off_tab + b_tab;
        sc.off_nodes = upload;
        // row nodes of the views (device arithmetic, once per geometry)
        hipLaunchKernelGGL(stream_nodes_kernel, dim3(256, n_views), dim3(256), 0, c->stream, dtr, (const int*)(dbuf + sc.off_noff), n_views,
                           (float2*)(dbuf + sc.off_nodes));
        sc.valid = true;
    }
    if (sc.nitems == 0 || !stream) return MVS_OK;       // (stream == nullptr: build and upload only)
    const dim3 grid(((sc.nitems + 3) / 4 + 7) / 8 * 8), block(256);
#define MVS_SK(A) hipLaunchKernelGGL(fuse_stream_kernel<A>, grid, block, 0, stream, (const SItem*)dbuf, sc.nitems, (const SStripD*)(dbuf + sc.off_strips), \
                       (const SEntry*)(dbuf + sc.off_entries), (const float2*)(dbuf + sc.off_nodes), (const float2*)(dbuf + sc.off_tab), (unsigned short*)dout, o[1], o[2], t[0], t[1], t[2])
    switch (c->ablate) {       // profiling only
        case 1: MVS_SK(1); break;
        case 2: MVS_SK(2); break;
        case 3: MVS_SK(3); break;
        case 4: MVS_SK(4); break;
        default: MVS_SK(0);
    }
#undef MVS_SK
    return MVS_OK;
}
