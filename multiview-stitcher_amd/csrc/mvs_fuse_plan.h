// mvs_fuse_plan.h -- internal: host-side helpers shared by the row-owning fuse paths (mvs_fuse_rows.hip,
// formerly mvs_fuse_rowlds.hip): geometry hash and the clustered break points of one axis.
#pragma once
#include "mvs_fuse_tr.h"

#include <algorithm>
#include <vector>

namespace mvsplan {

inline unsigned long long fnv1a(const void* p, size_t n, unsigned long long h) {
    const unsigned char* b = (const unsigned char*)p;
    for (size_t i = 0; i < n; ++i) { h ^= b[i]; h *= 1099511628211ull; }
    return h;
}

// Break points of axis d for the views in `sub`, inside the chunk range [t, t + o).  View borders that lie within `tol`
// of each other (tiles of one grid row / column after registration differ by a few pixels) are clustered: a cluster of
// lower borders contributes its minimum, a cluster of upper borders (hi + 1) its maximum, so the sliver between the
// clustered borders falls into the overlap cell, where the affected views are flagged "partial", and the single-view
// interior cells keep full coverage.
inline void axis_breakpoints(const TrView* htr, const std::vector<int>& sub, int d, int t, int o, std::vector<int>* out) {
    // kinds: 0 lower border (cluster -> min), 1 upper border + 1 (-> max),
    //        2 end of the lower shell (-> max), 3 start of the upper shell (-> min)
    std::vector<std::pair<int, int>> ev;
    auto clampi = [&](int v) { return std::min(std::max(v, t), t + o); };
    for (int v : sub) {
        const int lo = htr[v].lo[d], hi = htr[v].hi[d];
        if (lo > hi) continue;
        ev.push_back({clampi(lo), 0});
        ev.push_back({clampi(hi + 1), 1});
        // A thin shell next to every border: inside it the blend weight of the view can round to 0 (the reference
        // outputs 0 there even for a single view, weights.py:502-507); outside it a voxel seen by ONE view is simply
        // the resampled value whatever the weight is.  Along x the shell is only cut next to a border that no other
        // view of the strip covers (the rim of the mosaic): inside an overlap the cell is a blend cell anyway.
        const int shell = 4;
        bool cut_lo = true, cut_hi = true;
        if (d == 2) {
            for (int w : sub) {
                if (w == v || htr[w].lo[2] > htr[w].hi[2]) continue;
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


inline bool view_needs_taps(const TrView& V, int dtype) {
    return V.fw[0] > 0.f || V.fw[1] > 0.f || V.fw[2] > 0.f || (dtype == MVS_F32 && V.linear != 0);
}

}  // namespace mvsplan
