// mvs_fuse_rows.hip -- host side of the row-owning translation fast path of mvs_fuse_chunk: strip / cell
// decomposition, classification, work list, launch.  Device side and the design: mvs_fuse_rows_dev.h.
#include "mvs_fuse_rows_dev.h"
#include "mvs_fuse_plan.h"

#include <algorithm>
#include <chrono>
#include <cstdlib>
#include <cstring>
#include <vector>

using namespace mvsrows;
using namespace mvsplan;

namespace {

struct RowsPlan {
    unsigned long long hash = 0;
    bool valid = false;
    int nstrips = 0, ncells = 0;
    int class_count[6] = {0, 0, 0, 0, 0, 0};   // work items per kernel class: (NV <= 2, <= 4, <= 8) x (single tap, taps)
    int wpg = kWG;                              // wavefronts per workgroup the items were built for
    size_t off_cells = 0, off_items = 0, bytes = 0;
    double build_ms = 0.0;
};
RowsPlan g_rows_plan[MVS_MAX_DEVICES * MVS_MAX_LANES];
double g_last_plan_ms[MVS_MAX_DEVICES * MVS_MAX_LANES];

}  // namespace

double mvs_rows_last_plan_ms(MvsContext* c) { return g_last_plan_ms[mvs_ctx_index(c->device)]; }

// Returns MVS_OK and sets *done = true when the chunk was fused by the row kernels; *done = false means the caller
// must use another path (more than kMaxCV views on one cell, or too many cells / items).
int mvs_fuse_rows(MvsContext* c, const TrView* htr, const TrView* dtr, int n_views, int dtype, void* dout, const int64_t os[3],
                  const int64_t trim[3], bool* done) {
    *done = false;
    const int t[3] = {(int)trim[0], (int)trim[1], (int)trim[2]};
    const int o[3] = {(int)os[0], (int)os[1], (int)os[2]};
    unsigned long long h = fnv1a(htr, sizeof(TrView) * (size_t)n_views, 1469598103934665603ull);
    h = fnv1a(t, sizeof(t), h);
    h = fnv1a(o, sizeof(o), h);
    h = fnv1a(&dtype, sizeof(dtype), h);
    RowsPlan& pc = g_rows_plan[mvs_ctx_index(c->device)];
    g_last_plan_ms[mvs_ctx_index(c->device)] = 0.0;
    char* dbuf = nullptr;
    if (pc.valid && pc.hash == h && c->dev[13].ptr) {
        dbuf = (char*)c->dev[13].ptr;      // same geometry as the previous call: the plan is still on the device
    } else {
        const auto t_begin = std::chrono::steady_clock::now();
        std::vector<int> all(n_views);
        for (int v = 0; v < n_views; ++v) all[v] = v;
        std::vector<int> pz, py, px;
        axis_breakpoints(htr, all, 0, t[0], o[0], &pz);
        axis_breakpoints(htr, all, 1, t[1], o[1], &py);
        if ((pz.size() - 1) * (py.size() - 1) > 20000) return MVS_OK;
        // small chunks: fewer rows per workgroup so that the launch still fills the chip
        const long long rows_total = (long long)o[0] * o[1];
        const int wpg = rows_total >= 16384 * 4 ? 4 : rows_total >= 8192 * 2 ? 2 : 1;

        std::vector<Strip> strips;
        std::vector<Cell> cells;
        std::vector<int> strip_class;
        std::vector<int> zviews, sviews;
        for (size_t iz = 0; iz + 1 < pz.size(); ++iz) {
            zviews.clear();
            for (int v = 0; v < n_views; ++v)
                if (htr[v].lo[0] < pz[iz + 1] && htr[v].hi[0] >= pz[iz] && htr[v].lo[1] <= htr[v].hi[1] && htr[v].lo[2] <= htr[v].hi[2]) zviews.push_back(v);
            for (size_t iy = 0; iy + 1 < py.size(); ++iy) {
                sviews.clear();
                for (int v : zviews)
                    if (htr[v].lo[1] < py[iy + 1] && htr[v].hi[1] >= py[iy]) sviews.push_back(v);
                Strip S;
                memset(&S, 0, sizeof(S));
                S.z0 = pz[iz]; S.z1 = pz[iz + 1]; S.y0 = py[iy]; S.y1 = py[iy + 1];
                S.cell0 = (int)cells.size();
                axis_breakpoints(htr, sviews, 2, t[2], o[2], &px);
                int nvmax = 0;
                bool taps = false;
                for (size_t ix = 0; ix + 1 < px.size(); ++ix) {
                    Cell C;
                    memset(&C, 0, sizeof(C));
                    C.x0 = px[ix]; C.x1 = px[ix + 1];
                    int nv = 0;
                    bool positive_full = false, all_positive = true;
                    for (int v : sviews) {
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
                        if (view_needs_taps(htr[v], dtype)) taps = true;
                        C.ids[nv++] = v;
                    }
                    if (nv > 0 && all_positive) C.masks |= 1 << 15;
                    const int w = C.x1 - C.x0;
                    const int lxb = w > 256 ? 6 : w > 128 ? 5 : 4;
                    const int cls = nv == 0 ? 0 : (nv == 1 && positive_full) ? 1 : 2;
                    C.nv_lxb_cls = nv | (lxb << 8) | (cls << 16);
                    nvmax = std::max(nvmax, cls == 2 ? nv : 0);
                    cells.push_back(C);
                }
                S.ncells = (int)cells.size() - S.cell0;
                strips.push_back(S);
                strip_class.push_back((nvmax <= 2 ? 0 : nvmax <= 4 ? 1 : 2) * 2 + (taps ? 1 : 0));
            }
        }
        if (cells.size() > (1u << 22)) return MVS_OK;
        // work items, z-major per class: consecutive workgroups write consecutive row groups of a plane
        std::vector<RowItem> items_by_class[6];
        const int rows_per_wg = kWR * wpg;
        const size_t nys = py.size() - 1;
        for (size_t iz = 0; iz + 1 < pz.size(); ++iz)
            for (int z = pz[iz]; z < pz[iz + 1]; ++z)
                for (size_t iy = 0; iy < nys; ++iy) {
                    const int sid = (int)(iz * nys + iy);
                    const Strip& S = strips[sid];
                    std::vector<RowItem>& dst = items_by_class[strip_class[sid]];
                    for (int y = S.y0; y < S.y1; y += rows_per_wg) dst.push_back({sid, z, y, 0});
                }
        size_t nitems = 0;
        for (int k = 0; k < 6; ++k) nitems += items_by_class[k].size();
        if (nitems == 0 || nitems > (1u << 26)) return MVS_OK;
        const size_t sbytes = (strips.size() * sizeof(Strip) + 255) / 256 * 256;
        const size_t cbytes = (cells.size() * sizeof(Cell) + 255) / 256 * 256;
        const size_t ibytes = nitems * sizeof(RowItem);
        const size_t total = sbytes + cbytes + ibytes;
        char* hbuf = (char*)mvs_pinned_slot(c, 1, total + 256);   // slot 0 holds the view parameters still in flight
        if (!hbuf) return mvs_alloc_failed(c);
        pc.valid = false;
        dbuf = (char*)mvs_scratch(c, 13, total + 256);
        if (!dbuf) return mvs_alloc_failed(c);
        memcpy(hbuf, strips.data(), strips.size() * sizeof(Strip));
        memcpy(hbuf + sbytes, cells.data(), cells.size() * sizeof(Cell));
        size_t cur = sbytes + cbytes;
        for (int k = 0; k < 6; ++k) {
            pc.class_count[k] = (int)items_by_class[k].size();
            memcpy(hbuf + cur, items_by_class[k].data(), items_by_class[k].size() * sizeof(RowItem));
            cur += items_by_class[k].size() * sizeof(RowItem);
        }
        { const int rcu = mvs_upload_small(c, dbuf, hbuf, total); if (rcu) return rcu; }
        mvs_pinned_mark(c, 1);
        pc.hash = h;
        pc.nstrips = (int)strips.size();
        pc.ncells = (int)cells.size();
        pc.wpg = wpg;
        pc.off_cells = sbytes;
        pc.off_items = sbytes + cbytes;
        pc.bytes = total;
        pc.valid = true;
        pc.build_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_begin).count();
        g_last_plan_ms[mvs_ctx_index(c->device)] = pc.build_ms;
        if (getenv("MVS_PLAN_STATS")) {
            fprintf(stderr, "[mvs rows plan] strips %zu cells %zu items", strips.size(), cells.size());
            for (int k = 0; k < 6; ++k) fprintf(stderr, " %d", pc.class_count[k]);
            fprintf(stderr, " wpg %d, %.2f ms\n", wpg, pc.build_ms);
        }
    }
    const Strip* dstrips = (const Strip*)dbuf;
    const Cell* dcells = (const Cell*)(dbuf + pc.off_cells);
    const RowItem* ditems = (const RowItem*)(dbuf + pc.off_items);
    MVS_HIP_TRY(c, hipEventRecord(c->ev_start, c->stream));   // kernel time only (the plan is host work, cached per geometry)
    int item0 = 0;
    for (int k = 0; k < 6; ++k) {
        const int cnt = pc.class_count[k];
        if (!cnt) continue;
        const int nblocks = (cnt + 7) / 8 * 8;   // multiple of 8: see the XCD mapping in the kernel
        const int nvclass = k >> 1;
        const bool frac = k & 1;
        if (dtype == MVS_U8) mvs_launch_rows_u8(nvclass, frac, nblocks, pc.wpg, c->stream, dtr, dstrips, dcells, ditems + item0, cnt, dout, o[1], o[2], t[0], t[1], t[2]);
        else if (dtype == MVS_U16) mvs_launch_rows_u16(nvclass, frac, nblocks, pc.wpg, c->stream, dtr, dstrips, dcells, ditems + item0, cnt, dout, o[1], o[2], t[0], t[1], t[2]);
        else mvs_launch_rows_f32(nvclass, frac, nblocks, pc.wpg, c->stream, dtr, dstrips, dcells, ditems + item0, cnt, dout, o[1], o[2], t[0], t[1], t[2]);
        item0 += cnt;
    }
    MVS_HIP_TRY(c, hipGetLastError());
    *done = true;
    return MVS_OK;
}
