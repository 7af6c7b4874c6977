// mvs_plan.hip -- host-side planner of fusion.fuse: which views, and which index window of each, feed which output chunk.
// No device work (the file is .hip only because the library is built by one rule).  Contract in include/mvs_hip.h
// (mvs_fuse_plan); the reference's form is fusion/_core.py:354-722 (+ mv_graph.py:934-1117 and the label selection of
// _core.py:1371-1386), a per-chunk x per-view Python loop over dicts.  Here the plan is arrays: one pass decides which axes
// are pure translations / lie on the views' sampling grid, one pass per view finds the range of chunks its padded world box
// can touch, and one pass per (chunk, candidate view) derives the integer window -- the arithmetic whose rounding decides a
// window edge (floor / ceil of pixel coordinates with the reference's tolerances) is kept expression by expression.
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <vector>

#include "mvs_hip.h"

namespace {

constexpr double kTol = 1e-6;

// np.isclose(a, b, atol=atol) with its default rtol = 1e-5 (False for NaN)
inline bool is_close(double a, double b, double atol) {
    if (a == b) return true;
    return std::fabs(a - b) <= atol + 1e-5 * std::fabs(b);
}

inline bool on_grid(double offset, double spacing) {
    if (spacing == 0.0) return false;
    const double po = offset / spacing;
    if (!std::isfinite(po)) return false;
    return is_close(po, std::nearbyint(po), kTol);
}

// floor / ceil to int64 with the cast kept in range (NaN -> a value no block index or window can reach)
inline int64_t to_i64(double x) {
    if (!(x == x)) return INT64_MIN / 4;
    return (int64_t)std::min(std::max(x, -4e18), 4e18);
}

struct Geometry {
    int nd, nv;
    const double* vo; const double* vs; const int64_t* vn;      // views: origin, spacing, shape   [view][dim]
    const double* P; const double* Pinv;                         // (nd + 1)^2 row-major per view
    const double* oo; const double* os; const int64_t* on;      // output stack
    const int64_t* cs; const int64_t* halo;
    double p(int v, int r, int c) const { return P[((size_t)v * (nd + 1) + r) * (nd + 1) + c]; }
    double pinv(int v, int r, int c) const { return Pinv[((size_t)v * (nd + 1) + r) * (nd + 1) + c]; }
};

// coordinate of sample i of a view axis, as the images carry it (translation + scale * i)
inline double coord(const Geometry& G, int v, int d, int64_t i) { return G.vo[v * G.nd + d] + G.vs[v * G.nd + d] * (double)i; }

// first index whose coordinate is >= x (np.searchsorted(coords, x, "left")) / > x ("right"), coords ascending
inline int64_t first_at_least(const Geometry& G, int v, int d, double x, bool strictly_greater) {
    const int64_t n = G.vn[v * G.nd + d];
    int64_t lo = 0, hi = n;
    while (lo < hi) {
        const int64_t mid = lo + (hi - lo) / 2;
        const double c = coord(G, v, d, mid);
        if (strictly_greater ? (c <= x) : (c < x)) lo = mid + 1; else hi = mid;
    }
    return lo;
}

}  // namespace

extern "C" int mvs_fuse_plan(int32_t ndim, int32_t n_views, const double* view_origin, const double* view_spacing,
                             const int64_t* view_shape, const double* params, const double* inv_params, const double* out_origin,
                             const double* out_spacing, const int64_t* out_shape, const int64_t* chunk_size, const int64_t* halo,
                             int32_t interpolation_order, mvs_plan_entry_t* entries, int64_t capacity, int64_t* n_entries_out,
                             int32_t* dim_masks_out) {
    if ((ndim != 2 && ndim != 3) || n_views < 0 || !n_entries_out || (n_views > 0 && (!view_origin || !view_spacing || !view_shape || !params)) ||
        !out_origin || !out_spacing || !out_shape || !chunk_size || !halo)
        return MVS_ERR_INVALID_ARG;
    const Geometry G{ndim, n_views, view_origin, view_spacing, view_shape, params, inv_params, out_origin, out_spacing, out_shape, chunk_size, halo};
    const int nd = ndim;
    for (int d = 0; d < nd; ++d)
        if (chunk_size[d] < 1 || out_shape[d] < 0) return MVS_ERR_INVALID_ARG;

    // ---- axes on which every view is a pure translation; of those, the ones where the output samples fall on every view's
    // sampling grid (same spacing, offset a whole number of pixels): no interpolation there, no extra taps ----
    int axis_mask = 0, grid_mask = 0;
    for (int d = 0; d < nd; ++d) {
        bool ok = true;
        for (int v = 0; v < n_views && ok; ++v) {
            if (!is_close(G.p(v, d, d), 1.0, kTol)) ok = false;
            for (int o = 0; o < nd && ok; ++o)
                if (o != d && (!is_close(G.p(v, d, o), 0.0, kTol) || !is_close(G.p(v, o, d), 0.0, kTol))) ok = false;
        }
        if (ok) axis_mask |= 1 << d;
    }
    for (int d = 0; d < nd; ++d) {
        if (!((axis_mask >> d) & 1)) continue;
        bool ok = true;
        for (int v = 0; v < n_views && ok; ++v)
            if (!is_close(out_spacing[d], view_spacing[v * nd + d], kTol)) ok = false;
        for (int v = 0; v < n_views && ok; ++v)
            if (!on_grid(out_origin[d] - G.p(v, d, nd) - view_origin[v * nd + d], view_spacing[v * nd + d])) ok = false;
        if (ok) grid_mask |= 1 << d;
    }
    if (dim_masks_out) { dim_masks_out[0] = axis_mask; dim_masks_out[1] = grid_mask; }
    const bool all_axis = axis_mask == (1 << nd) - 1;
    if (!all_axis && n_views > 0 && !inv_params) return MVS_ERR_INVALID_ARG;

    // ---- chunk grid: uniform chunks, a shorter last one ----
    int64_t nblk[3] = {1, 1, 1};
    for (int d = 0; d < nd; ++d) nblk[d] = out_shape[d] == 0 ? 1 : (out_shape[d] + chunk_size[d] - 1) / chunk_size[d];

    // ---- per view: the range of blocks its world bounding box, grown by the interpolation taps and the halo, reaches ----
    std::vector<int64_t> first((size_t)n_views * nd), last((size_t)n_views * nd);
    std::vector<char> nowhere((size_t)std::max(n_views, 1), 0);
    for (int v = 0; v < n_views; ++v) {
        double bmin[3] = {INFINITY, INFINITY, INFINITY}, bmax[3] = {-INFINITY, -INFINITY, -INFINITY};
        for (int corner = 0; corner < (1 << nd); ++corner) {
            double pt[3];
            for (int d = 0; d < nd; ++d) {
                const int bit = (corner >> (nd - 1 - d)) & 1;      // corners of the unit cube, last axis fastest
                pt[d] = (double)bit * (double)(view_shape[v * nd + d] - 1) * view_spacing[v * nd + d] + view_origin[v * nd + d];
            }
            for (int r = 0; r < nd; ++r) {
                double w = 0.0;
                for (int k = 0; k < nd; ++k) w += pt[k] * G.p(v, r, k);
                w += G.p(v, r, nd);
                bmin[r] = std::min(bmin[r], w);
                bmax[r] = std::max(bmax[r], w);
            }
        }
        for (int d = 0; d < nd; ++d) {
            const double interp_pad = ((grid_mask >> d) & 1) ? 0.0 : (double)interpolation_order * view_spacing[v * nd + d];
            const double pad = interp_pad + (double)halo[d] * out_spacing[d];
            // (a stack shorter than one chunk has ONE block of its own length: normalize_chunks)
            const double cs_phys = (double)std::max<int64_t>(std::min<int64_t>(chunk_size[d], out_shape[d]), 1) * out_spacing[d];
            const int64_t f = std::max<int64_t>(0, to_i64(std::floor((bmin[d] - pad - out_origin[d]) / cs_phys)));
            const int64_t l = std::min<int64_t>(nblk[d] - 1, to_i64(std::floor((bmax[d] + pad - out_origin[d]) / cs_phys)));
            first[(size_t)v * nd + d] = f;
            last[(size_t)v * nd + d] = l;
            if (f > l) nowhere[v] = 1;
        }
    }

    // ---- per chunk (block order: first axis slowest) and candidate view (ascending): the index window ----
    int64_t count = 0;
    int64_t blk[3] = {0, 0, 0};
    const int64_t total_blocks = nblk[0] * nblk[1] * nblk[2];
    for (int64_t b = 0; b < total_blocks; ++b) {
        {   // decode the block index
            int64_t rest = b;
            for (int d = nd - 1; d >= 0; --d) { blk[d] = rest % nblk[d]; rest /= nblk[d]; }
        }
        double t_origin[3];
        int64_t t_shape[3];
        for (int d = 0; d < nd; ++d) {
            const int64_t off = blk[d] * chunk_size[d];
            const int64_t n = std::min<int64_t>(chunk_size[d], out_shape[d] - off);
            const double origin = out_origin[d] + out_spacing[d] * (double)off;        // chunk origin ...
            t_origin[d] = origin - (double)halo[d] * out_spacing[d];                    // ... grown by the halo
            t_shape[d] = n + 2 * halo[d];
        }
        bool planewise = false;
        if (nd == 3 && (grid_mask & 1) && t_shape[0] == 1) planewise = true;
        for (int v = 0; v < n_views; ++v) {
            if (nowhere[v]) continue;
            bool cand = true;
            for (int d = 0; d < nd && cand; ++d) cand = blk[d] >= first[(size_t)v * nd + d] && blk[d] <= last[(size_t)v * nd + d];
            if (!cand) continue;
            int64_t lo[3] = {0, 0, 0}, n[3] = {0, 0, 0};
            bool hit = true;
            if (all_axis) {
                // pure translations: the chunk's first and last sample, moved into the view's frame, as pixel coordinates
                for (int d = 0; d < nd && hit; ++d) {
                    const double qs = view_spacing[v * nd + d], ts = out_spacing[d], tr = G.p(v, d, nd);
                    double qmin = t_origin[d] - tr;
                    double qmax = t_origin[d] + (double)(t_shape[d] - 1) * ts - tr;
                    if (qmin > qmax) std::swap(qmin, qmax);
                    const int64_t ext = ((grid_mask >> d) & 1) ? 0 : interpolation_order;
                    const double extra = (double)ext * qs;
                    const double start_f = (qmin - extra - view_origin[v * nd + d]) / qs;
                    const double stop_f = (qmax + extra - view_origin[v * nd + d]) / qs;
                    const int64_t start = to_i64(std::floor(start_f + kTol));
                    const int64_t stop = to_i64(std::ceil(stop_f - kTol)) + 1;
                    const int64_t a = std::max<int64_t>(start, 0), e = std::min<int64_t>(stop, view_shape[v * nd + d]);
                    if (a >= e) { hit = false; break; }
                    lo[d] = a;
                    n[d] = e - a;
                }
            } else {
                // general affine: bounding box of the chunk's corners in the view's frame -> box in physical units -> the samples
                // a label selection [origin - tol, last + tol] picks
                double cmin[3] = {INFINITY, INFINITY, INFINITY}, cmax[3] = {-INFINITY, -INFINITY, -INFINITY};
                for (int corner = 0; corner < (1 << nd); ++corner) {
                    double pt[3];
                    for (int d = 0; d < nd; ++d) {
                        const int bit = (corner >> (nd - 1 - d)) & 1;
                        pt[d] = (double)bit * (double)(t_shape[d] - 1) * out_spacing[d] + t_origin[d];
                    }
                    for (int r = 0; r < nd; ++r) {
                        double w = 0.0;
                        for (int k = 0; k < nd; ++k) w += pt[k] * G.pinv(v, r, k);
                        w += G.pinv(v, r, nd);
                        cmin[r] = std::min(cmin[r], w);
                        cmax[r] = std::max(cmax[r], w);
                    }
                }
                double box_o[3], sel_lo[3], sel_hi[3];
                int64_t box_n[3];
                for (int d = 0; d < nd && hit; ++d) {
                    const double qs = view_spacing[v * nd + d], qo = view_origin[v * nd + d];
                    const int64_t ext = ((grid_mask >> d) & 1) ? 0 : interpolation_order;
                    box_o[d] = cmin[d] - (double)ext * qs;
                    box_n[d] = to_i64(std::ceil((cmax[d] - cmin[d]) / qs)) + 1 + 2 * ext;
                    const double q_last = qo + (double)(view_shape[v * nd + d] - 1) * qs;
                    const double box_last = box_o[d] + (double)(box_n[d] - 1) * qs;
                    if (box_o[d] - kTol > q_last || box_last < qo - kTol) { hit = false; break; }
                    const double o = std::max(box_o[d], qo);
                    const int64_t m = to_i64(std::ceil((std::min(box_last, q_last) - o) / qs)) + 1;
                    if (m < 1) { hit = false; break; }
                    sel_lo[d] = o - kTol;
                    sel_hi[d] = o + (double)(m - 1) * qs + kTol;
                }
                for (int d = 0; d < nd && hit; ++d) {
                    const int64_t a = first_at_least(G, v, d, sel_lo[d], false), e = first_at_least(G, v, d, sel_hi[d], true);
                    if (a >= e) { hit = false; break; }
                    lo[d] = a;
                    n[d] = e - a;
                }
            }
            if (!hit) continue;
            if (entries && count < capacity) {
                mvs_plan_entry_t& E = entries[count];
                for (int d = 0; d < 3; ++d) { E.block[d] = d < nd ? blk[d] : 0; E.lo[d] = d < nd ? lo[d] : 0; E.n[d] = d < nd ? n[d] : 0; }
                E.view = v;
                E.planewise = planewise ? 1 : 0;
            }
            ++count;
        }
    }
    *n_entries_out = count;
    return MVS_OK;
}
