// Host-side inner loop of the groupwise resolution (no device work): the node sweeps of the virtual-bead optimisation for the
// translation model.  See include/mvs_hip.h (mvs_beads_translation_sweeps) for the contract; the edge-removal outer loop and
// the models with a linear part live in multiview_stitcher_amd/param_resolution.py.
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <vector>

#if defined(__x86_64__)
#include <immintrin.h>
#define MVS_RESOLVE_AVX2 1
#endif

#include "mvs_hip.h"

#ifdef MVS_RESOLVE_AVX2
// The residual pass of one sweep for the common case (3D, 8 beads per edge) on 4-wide vectors: lane = bead & 3, i.e. exactly the
// four accumulators of the scalar loops below, the same operations in the same order (separate multiplies and adds, no fused
// ones), so the results are bit-identical to the scalar form (tests/test_param_resolution.py runs both).  d0 is stored per edge as
// [x of beads 0..7 | y of beads 0..7 | z of beads 0..7].  Returns the sum over edges of (edge mean) and updates the three maxima.
__attribute__((target("avx2"))) static double residual_pass_avx2(int n_edges, const int32_t* edge_nodes, const double* d0soa,
                                                                 const double* translations, double* edge_residuals, double* prev,
                                                                 bool with_prev, double* mx_out, double* dmax_out) {
    __m256d m4 = _mm256_setzero_pd(), c4 = _mm256_setzero_pd();
    const __m256d absmask = _mm256_castsi256_pd(_mm256_set1_epi64x(0x7fffffffffffffffLL));
    double mean_acc = 0.0;
    for (int e = 0; e < n_edges; ++e) {
        const int a = edge_nodes[2 * e], b2 = edge_nodes[2 * e + 1];
        const double* dp = d0soa + (size_t)e * 24;
        double* rp = edge_residuals + (size_t)e * 8;
        double* pp = prev + (size_t)e * 8;
        __m256d es = _mm256_setzero_pd();
        __m256d dt[3];
        for (int d = 0; d < 3; ++d) dt[d] = _mm256_set1_pd(translations[(size_t)a * 3 + d] - translations[(size_t)b2 * 3 + d]);
        for (int h = 0; h < 2; ++h) {
            __m256d q = _mm256_setzero_pd();
            for (int d = 0; d < 3; ++d) {
                const __m256d v = _mm256_add_pd(_mm256_loadu_pd(dp + d * 8 + h * 4), dt[d]);
                q = _mm256_add_pd(q, _mm256_mul_pd(v, v));
            }
            const __m256d r = _mm256_sqrt_pd(q);
            es = _mm256_add_pd(es, r);
            m4 = _mm256_max_pd(r, m4);                              // (r, m4): a NaN residual leaves the maximum alone, like fmax
            if (with_prev) c4 = _mm256_max_pd(_mm256_and_pd(_mm256_sub_pd(r, _mm256_loadu_pd(pp + h * 4)), absmask), c4);
            _mm256_storeu_pd(rp + h * 4, r);
            _mm256_storeu_pd(pp + h * 4, r);
        }
        double esv[4];
        _mm256_storeu_pd(esv, es);
        mean_acc += ((esv[0] + esv[1]) + (esv[2] + esv[3])) / 8.0;
    }
    double mv[4], cv[4];
    _mm256_storeu_pd(mv, m4);
    _mm256_storeu_pd(cv, c4);
    *mx_out = std::fmax(std::fmax(mv[0], mv[1]), std::fmax(mv[2], mv[3]));
    *dmax_out = std::fmax(std::fmax(cv[0], cv[1]), std::fmax(cv[2], cv[3]));
    return mean_acc;
}
#endif      // MVS_RESOLVE_AVX2 (other hosts: the scalar loops below, same results bit for bit)

extern "C" int mvs_beads_translation_sweeps(int32_t ndim, int32_t n_nodes, int32_t n_edges, const int32_t* edge_nodes,
                                            const double* beads_a, const double* beads_b, int32_t n_beads, const int32_t* order,
                                            int32_t ref_node, int32_t max_iter, double rel_tol, double* translations,
                                            double* edge_residuals, double* mean_hist, double* max_hist, int32_t* n_iter_out) {
    if (ndim < 1 || ndim > 3 || n_nodes < 1 || n_edges < 0 || n_beads < 1 || max_iter < 0 || !edge_nodes || !order ||
        !translations || !edge_residuals || !mean_hist || !max_hist || !n_iter_out || (n_edges > 0 && (!beads_a || !beads_b)))
        return MVS_ERR_INVALID_ARG;
    for (int e = 0; e < n_edges; ++e)
        for (int k = 0; k < 2; ++k)
            if (edge_nodes[2 * e + k] < 0 || edge_nodes[2 * e + k] >= n_nodes) return MVS_ERR_INVALID_ARG;
    // The translation model makes every sum over beads affine in the translations: with d0 = bead_a - bead_b (constant) the bead
    // difference of edge e is v = d0 + (T_a - T_b), so the mean a node's update needs is sum_e +-(D0_e + n_beads (T_a - T_b)) / count
    // with D0_e = sum_b d0 -- O(degree) per node instead of O(degree x beads x dims) -- and only the residual norms need the beads
    // themselves.  Same quantities as the bead-by-bead form up to the rounding of a differently ordered sum (1e-16 relative; the
    // numpy form and this one are compared to 1e-10 in tests/test_param_resolution.py).  The sweeps are bound by the latency of
    // their addition chains, so the residual sums run in four independent accumulators.
    const size_t per_edge = (size_t)n_beads * (size_t)ndim;
    std::vector<double> d0((size_t)n_edges * per_edge), D0((size_t)n_edges * 3, 0.0);
    for (int e = 0; e < n_edges; ++e)
        for (int b = 0; b < n_beads; ++b)
            for (int d = 0; d < ndim; ++d) {
                const double v = beads_a[(size_t)e * per_edge + b * ndim + d] - beads_b[(size_t)e * per_edge + b * ndim + d];
                d0[(size_t)e * per_edge + b * ndim + d] = v;
                D0[(size_t)e * 3 + d] += v;
            }
    // incident edges per node (flattened): edge index and the sign of v in the node's update (+1 when the node is the b side)
    std::vector<int> inc_off((size_t)n_nodes + 1, 0), inc_edge((size_t)n_edges * 2), inc_other((size_t)n_edges * 2);
    std::vector<double> inc_sign((size_t)n_edges * 2);
    for (int e = 0; e < n_edges; ++e) { ++inc_off[(size_t)edge_nodes[2 * e] + 1]; ++inc_off[(size_t)edge_nodes[2 * e + 1] + 1]; }
    for (int n = 0; n < n_nodes; ++n) inc_off[(size_t)n + 1] += inc_off[(size_t)n];
    {
        std::vector<int> fill(inc_off.begin(), inc_off.end() - 1);
        for (int e = 0; e < n_edges; ++e) {
            const int a = edge_nodes[2 * e], b2 = edge_nodes[2 * e + 1];
            inc_edge[(size_t)fill[a]] = e; inc_other[(size_t)fill[a]] = b2; inc_sign[(size_t)fill[a]++] = -1.0;
            inc_edge[(size_t)fill[b2]] = e; inc_other[(size_t)fill[b2]] = a; inc_sign[(size_t)fill[b2]++] = 1.0;
        }
    }
    const double nb = (double)n_beads;
    const size_t nres = (size_t)n_edges * (size_t)n_beads;
    std::vector<double> prev(nres, 0.0);
    // vector form of the residual pass (3D mosaics: 8 beads per edge), unless switched off for the A/B test
    static const bool no_simd = getenv("MVS_RESOLVE_SCALAR") != nullptr;
#ifdef MVS_RESOLVE_AVX2
    const bool simd = ndim == 3 && n_beads == 8 && !no_simd && __builtin_cpu_supports("avx2");
#else
    const bool simd = false;
    (void)no_simd;
#endif
    std::vector<double> d0soa;
    if (simd) {
        d0soa.resize((size_t)n_edges * 24);
        for (int e = 0; e < n_edges; ++e)
            for (int b = 0; b < 8; ++b)
                for (int d = 0; d < 3; ++d) d0soa[(size_t)e * 24 + d * 8 + b] = d0[(size_t)e * 24 + b * 3 + d];
    }
    int it = 0;
    for (; it < max_iter; ++it) {
        for (int s = 0; s < n_nodes; ++s) {
            const int c = order[s];
            if (c < 0 || c >= n_nodes) return MVS_ERR_INVALID_ARG;
            const int i0 = inc_off[(size_t)c], i1 = inc_off[(size_t)c + 1];
            if (i0 == i1 || c == ref_node) continue;
            // TranslationTransform.estimate: mean over all bead pairs of (adjacent bead - own bead), both in world coordinates
            double sum[3] = {0.0, 0.0, 0.0};
            for (int i = i0; i < i1; ++i) {
                const int e = inc_edge[(size_t)i], o = inc_other[(size_t)i];
                const double sg = inc_sign[(size_t)i];
                // sign * (D0 + nb (T_a - T_b)): for the b side (sign +1) a = other, for the a side (sign -1) a = c
                for (int d = 0; d < ndim; ++d) {
                    const double ta = sg > 0.0 ? translations[(size_t)o * ndim + d] : translations[(size_t)c * ndim + d];
                    const double tb = sg > 0.0 ? translations[(size_t)c * ndim + d] : translations[(size_t)o * ndim + d];
                    sum[d] += sg * (D0[(size_t)e * 3 + d] + nb * (ta - tb));
                }
            }
            const double cnt = (double)((size_t)(i1 - i0) * (size_t)n_beads);
            for (int d = 0; d < ndim; ++d) translations[(size_t)c * ndim + d] += sum[d] / cnt;
        }
        // bead residuals of every edge, their mean of means and overall maximum
        double mean_acc = 0.0, mx = 0.0;
#ifdef MVS_RESOLVE_AVX2
        if (simd) {
            double dmax = 0.0;
            mean_acc = residual_pass_avx2(n_edges, edge_nodes, d0soa.data(), translations, edge_residuals, prev.data(), it > 5, &mx, &dmax);
            mean_hist[it] = n_edges ? mean_acc / (double)n_edges : 0.0;
            max_hist[it] = mx;
            if (it > 5 && (mx > 0.0 ? dmax / mx : 0.0) < rel_tol) { ++it; break; }
            continue;
        }
#endif
        for (int e = 0; e < n_edges; ++e) {
            const int a = edge_nodes[2 * e], b2 = edge_nodes[2 * e + 1];
            double dt[3] = {0.0, 0.0, 0.0};
            for (int d = 0; d < ndim; ++d) dt[d] = translations[(size_t)a * ndim + d] - translations[(size_t)b2 * ndim + d];
            double es[4] = {0.0, 0.0, 0.0, 0.0};
            const double* dp = d0.data() + (size_t)e * per_edge;
            double* rp = edge_residuals + (size_t)e * n_beads;
            for (int b = 0; b < n_beads; ++b) {
                double q = 0.0;
                for (int d = 0; d < ndim; ++d) {
                    const double v = dp[b * ndim + d] + dt[d];
                    q += v * v;
                }
                const double r = std::sqrt(q);
                rp[b] = r;
                es[b & 3] += r;
            }
            const double esum = (es[0] + es[1]) + (es[2] + es[3]);
            mean_acc += esum / nb;
        }
        double m4[4] = {0.0, 0.0, 0.0, 0.0};
        for (size_t i = 0; i < nres; ++i) m4[i & 3] = std::fmax(m4[i & 3], edge_residuals[i]);
        mx = std::fmax(std::fmax(m4[0], m4[1]), std::fmax(m4[2], m4[3]));
        mean_hist[it] = n_edges ? mean_acc / (double)n_edges : 0.0;
        max_hist[it] = mx;
        bool converged = false;
        if (it > 5) {   // global_optimization.py:399-417: largest relative change of any bead residual
            // max_i |d_i / mx| == (max_i |d_i|) / mx: the division by mx > 0 is monotonic, so it is applied to the largest |d_i| only
            double c4[4] = {0.0, 0.0, 0.0, 0.0};
            for (size_t i = 0; i < nres; ++i) c4[i & 3] = std::fmax(c4[i & 3], std::fabs(edge_residuals[i] - prev[i]));
            const double dmax = std::fmax(std::fmax(c4[0], c4[1]), std::fmax(c4[2], c4[3]));
            const double rel = mx > 0.0 ? dmax / mx : 0.0;
            converged = rel < rel_tol;
        }
        for (size_t i = 0; i < nres; ++i) prev[i] = edge_residuals[i];
        if (converged) { ++it; break; }
    }
    *n_iter_out = it;
    return MVS_OK;
}

// Brandes' edge betweenness of an unweighted, undirected graph, normalised by n (n - 1): the values
// networkx.edge_betweenness_centrality(g) returns with its defaults (the reference calls it in
// prune_graph_to_alternating_colors, mv_graph.py:664-741, and compares the derived edge values with <=, so the order of the
// floating-point accumulation is part of the contract).  Nodes are 0 .. n_nodes - 1 in the graph's node order; node v's
// neighbours are adj_nodes[adj_offsets[v] .. adj_offsets[v + 1]) in adjacency (insertion) order, adj_edge gives the index
// of the edge each adjacency entry belongs to.  Same traversal (BFS in adjacency order), same operations in the same
// order as the Python form in mv_graph.edge_betweenness_centrality, which tests pin against networkx itself.
extern "C" int mvs_edge_betweenness(int32_t n_nodes, int32_t n_edges, const int32_t* adj_offsets, const int32_t* adj_nodes,
                                    const int32_t* adj_edge, double* bet_out) {
    if (n_nodes < 0 || n_edges < 0 || (n_nodes > 0 && (!adj_offsets || !bet_out))) return MVS_ERR_INVALID_ARG;
    for (int e = 0; e < n_edges; ++e) bet_out[e] = 0.0;
    std::vector<int> S, queue, dist((size_t)n_nodes), pred_off((size_t)n_nodes + 1), pred_cnt((size_t)n_nodes);
    std::vector<double> sigma((size_t)n_nodes), delta((size_t)n_nodes);
    // predecessor lists: node w has at most deg(w) predecessors -> slots [adj_offsets[w], adj_offsets[w + 1])
    const int nadj = n_nodes ? adj_offsets[n_nodes] : 0;
    std::vector<int> pred((size_t)std::max(nadj, 1)), pred_e((size_t)std::max(nadj, 1));
    S.reserve((size_t)n_nodes);
    queue.reserve((size_t)n_nodes);
    for (int s = 0; s < n_nodes; ++s) {
        S.clear();
        queue.clear();
        for (int v = 0; v < n_nodes; ++v) { dist[v] = -1; sigma[v] = 0.0; pred_cnt[v] = 0; delta[v] = 0.0; }
        sigma[s] = 1.0;
        dist[s] = 0;
        queue.push_back(s);
        for (size_t head = 0; head < queue.size(); ++head) {
            const int v = queue[head];
            S.push_back(v);
            for (int a = adj_offsets[v]; a < adj_offsets[v + 1]; ++a) {
                const int w = adj_nodes[a];
                if (dist[w] < 0) {
                    queue.push_back(w);
                    dist[w] = dist[v] + 1;
                }
                if (dist[w] == dist[v] + 1) {
                    sigma[w] += sigma[v];
                    pred[(size_t)adj_offsets[w] + pred_cnt[w]] = v;
                    pred_e[(size_t)adj_offsets[w] + pred_cnt[w]] = adj_edge[a];
                    ++pred_cnt[w];
                }
            }
        }
        while (!S.empty()) {
            const int w = S.back();
            S.pop_back();
            const double coeff = (1.0 + delta[w]) / sigma[w];
            for (int k = 0; k < pred_cnt[w]; ++k) {
                const int v = pred[(size_t)adj_offsets[w] + k];
                const double c = sigma[v] * coeff;
                bet_out[pred_e[(size_t)adj_offsets[w] + k]] += c;
                delta[v] += c;
            }
        }
    }
    if (n_nodes > 1) {
        const double scale = 1.0 / ((double)n_nodes * (double)(n_nodes - 1));
        for (int e = 0; e < n_edges; ++e) bet_out[e] *= scale;
    }
    return MVS_OK;
}

// ---- overlap graph of axis-aligned views + "alternating_pattern" pruning (registration.register steps 1a / 1b) ---------------
// One call for what mv_graph.build_view_adjacency_graph (mv_graph.py:35-180) and prune_graph_to_alternating_colors
// (mv_graph.py:664-741) do between them when every view's world frame is an axis-aligned box: the overlap volume of every
// candidate pair (closed form of the box intersection, the value Qhull returns for it), the graph with networkx's orders (nodes
// 0 .. n - 1, a node's neighbours in the order their edges were added, edges() node by node), Brandes' edge betweenness, the edge
// values `overlap + bonus`, and the level-by-level removal of the weakest edges until a greedy largest-first colouring needs
// no more than n_colors colours.  Every comparison and floating-point operation is the one of the Python form in
// multiview_stitcher_amd/mv_graph.py, in the same order (tests/test_graph_native.py compares the two on regular and irregular
// mosaics; the Python form is pinned against networkx).  See include/mvs_hip.h for the contract.
namespace {
struct AdjEntry { int nb; int eid; };

// networkx.coloring.greedy_color(strategy="largest_first") on the alive edges; false as soon as a node needs colour >= limit
bool greedy_color_within(const std::vector<std::vector<AdjEntry>>& adj, const std::vector<char>& alive, const std::vector<int>& deg, int limit,
                         std::vector<int>* colors_out) {
    const int n = (int)adj.size();
    std::vector<int> order((size_t)n);
    for (int v = 0; v < n; ++v) order[(size_t)v] = v;
    std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return deg[(size_t)a] > deg[(size_t)b]; });
    std::vector<int> color((size_t)n, -1);
    std::vector<char> used;
    for (int u : order) {
        used.assign((size_t)deg[(size_t)u] + 2, 0);
        for (const AdjEntry& a : adj[(size_t)u]) {
            if (!alive[(size_t)a.eid]) continue;
            const int c = color[(size_t)a.nb];
            if (c >= 0 && c < (int)used.size()) used[(size_t)c] = 1;
        }
        int c = 0;
        while (used[(size_t)c]) ++c;
        if (limit >= 0 && c >= limit) return false;
        color[(size_t)u] = c;
    }
    if (colors_out) *colors_out = color;
    return true;
}
}   // namespace

extern "C" int mvs_view_graph_prune(int32_t ndim, int32_t n_views, const double* box_lo, const double* box_hi, int64_t n_pairs,
                                    const int32_t* pairs, int32_t method, int32_t n_colors, int32_t* edges_out, double* overlap_out,
                                    int32_t* n_edges_out, int32_t* n_graph_edges_out) {
    if (ndim < 1 || ndim > 3 || n_views < 1 || n_pairs < 0 || !box_lo || !box_hi || (n_pairs > 0 && !pairs) || !edges_out || !overlap_out ||
        !n_edges_out || (method != 0 && method != 1))
        return MVS_ERR_INVALID_ARG;
    const int n = n_views;
    // ---- the overlap graph: an edge per unordered pair with a positive intersection volume, added at the pair's FIRST appearance ----
    std::vector<std::vector<AdjEntry>> adj((size_t)n);
    std::vector<int> ea, eb;
    std::vector<double> eov;
    {
        std::vector<uint64_t> seen;      // keys already decided (sorted lookups would need a set; the pair list is short: hash by open addressing)
        size_t cap = 16;
        while (cap < (size_t)n_pairs * 2 + 16) cap <<= 1;
        seen.assign(cap, ~0ull);
        for (int64_t p = 0; p < n_pairs; ++p) {
            const int i = pairs[2 * p], j = pairs[2 * p + 1];
            if (i < 0 || j < 0 || i >= n || j >= n || i == j) return MVS_ERR_INVALID_ARG;
            const int a = std::min(i, j), b = std::max(i, j);
            const uint64_t key = ((uint64_t)(uint32_t)a << 32) | (uint32_t)b;
            size_t h = (size_t)((key * 0x9E3779B97F4A7C15ull) >> 20) & (cap - 1);
            bool known = false;
            while (seen[h] != ~0ull) {
                if (seen[h] == key) { known = true; break; }
                h = (h + 1) & (cap - 1);
            }
            if (known) continue;
            seen[h] = key;
            // ext = minimum(hi_a, hi_b) - maximum(lo_a, lo_b); volume = prod(ext) if all(ext > 0) else -1
            double vol = 0.0;
            bool pos = true;
            for (int d = 0; d < ndim; ++d) {
                const double hi = std::fmin(box_hi[(size_t)a * ndim + d], box_hi[(size_t)b * ndim + d]);
                const double lo = std::fmax(box_lo[(size_t)a * ndim + d], box_lo[(size_t)b * ndim + d]);
                const double ext = hi - lo;
                pos = pos && (ext > 0.0);
                vol = d == 0 ? ext : vol * ext;
            }
            if (!pos || !(vol > 0.0)) continue;
            const int eid = (int)ea.size();
            ea.push_back(i);      // (as given: adjacency rows are filled i first, then j)
            eb.push_back(j);
            eov.push_back(vol);
            adj[(size_t)i].push_back(AdjEntry{j, eid});
            adj[(size_t)j].push_back(AdjEntry{i, eid});
        }
    }
    const int ne = (int)ea.size();
    if (n_graph_edges_out) *n_graph_edges_out = ne;
    std::vector<char> alive((size_t)ne, 1);
    std::vector<int> deg((size_t)n);
    for (int v = 0; v < n; ++v) deg[(size_t)v] = (int)adj[(size_t)v].size();
    auto emit = [&]() {
        int k = 0;
        for (int v = 0; v < n; ++v)
            for (const AdjEntry& a : adj[(size_t)v])
                if (alive[(size_t)a.eid] && a.nb >= v) {
                    edges_out[2 * k] = v;
                    edges_out[2 * k + 1] = a.nb;
                    overlap_out[k] = eov[(size_t)a.eid];
                    ++k;
                }
        *n_edges_out = k;
    };
    if (method == 0 || ne == 0) { emit(); return MVS_OK; }

    // ---- edges() order of the graph: the index space of betweenness, values and the rising list ----
    std::vector<int> k_of((size_t)ne, -1), eid_of((size_t)ne);
    std::vector<int32_t> off((size_t)n + 1, 0), an, ae;
    {
        int k = 0;
        for (int v = 0; v < n; ++v)
            for (const AdjEntry& a : adj[(size_t)v])
                if (a.nb >= v) { k_of[(size_t)a.eid] = k; eid_of[(size_t)k] = a.eid; ++k; }
        for (int v = 0; v < n; ++v) {
            for (const AdjEntry& a : adj[(size_t)v]) { an.push_back(a.nb); ae.push_back(k_of[(size_t)a.eid]); }
            off[(size_t)v + 1] = (int32_t)an.size();
        }
    }
    std::vector<double> cent((size_t)ne);
    int rc = mvs_edge_betweenness(n, ne, off.data(), an.data(), ae.data(), cent.data());
    if (rc) return rc;
    double cmax = cent[0], cmin = cent[0], min_overlap = eov[(size_t)eid_of[0]];
    for (int k = 0; k < ne; ++k) {
        cmax = std::max(cmax, cent[(size_t)k]);
        cmin = std::min(cmin, cent[(size_t)k]);
        min_overlap = std::min(min_overlap, eov[(size_t)eid_of[(size_t)k]]);
    }
    if (cmax > cmin)
        for (int k = 0; k < ne; ++k) cent[(size_t)k] = (cent[(size_t)k] - cmin) / (cmax - cmin) * 0.5 * min_overlap;
    std::vector<double> vals((size_t)ne);
    for (int k = 0; k < ne; ++k) {
        vals[(size_t)k] = cent[(size_t)k] + eov[(size_t)eid_of[(size_t)k]];
        if (vals[(size_t)k] != vals[(size_t)k]) return MVS_ERR_UNSUPPORTED;       // NaN geometry: the generic path decides
    }
    std::vector<double> levels(vals);
    std::sort(levels.begin(), levels.end());
    levels.erase(std::unique(levels.begin(), levels.end()), levels.end());
    std::vector<int> rising((size_t)ne);
    for (int k = 0; k < ne; ++k) rising[(size_t)k] = k;
    std::stable_sort(rising.begin(), rising.end(), [&](int a, int b) { return vals[(size_t)a] < vals[(size_t)b]; });
    size_t nxt = 0, lev_i = 0;
    std::vector<int> kept, drop, rest;
    bool changed = true;
    for (;;) {
        if (changed && greedy_color_within(adj, alive, deg, n_colors, nullptr)) break;
        if (lev_i >= levels.size()) return MVS_ERR_UNSUPPORTED;      // (the Python form runs out of levels here: let it raise)
        const double lev = levels[lev_i];
        while (nxt < rising.size() && vals[(size_t)rising[nxt]] <= lev) kept.push_back(rising[nxt++]);
        drop.clear();
        rest.clear();
        for (int k : kept) {      // degrees as they are BEFORE this level removes anything
            const int e = eid_of[(size_t)k];
            const int a = std::min(ea[(size_t)e], eb[(size_t)e]), b = std::max(ea[(size_t)e], eb[(size_t)e]);
            if (deg[(size_t)a] > 1 && deg[(size_t)b] > 1) drop.push_back(k); else rest.push_back(k);
        }
        if (!drop.empty()) {
            kept.swap(rest);
            for (int k : drop) {
                const int e = eid_of[(size_t)k];
                alive[(size_t)e] = 0;
                --deg[(size_t)ea[(size_t)e]];
                --deg[(size_t)eb[(size_t)e]];
            }
        }
        changed = !drop.empty();
        ++lev_i;
    }
    emit();
    return MVS_OK;
}

// ---- groupwise resolution of a connected translation mosaic in one call (registration.register step 3) --------------------------
// param_resolution.groupwise_resolution(method="global_optimization", transform="translation") for the case every regular
// mosaic is: ONE connected component holding all views 0 .. n - 1, pairwise results that are pure translations, and sweeps that
// end below abs_tol, so that the edge-removal loop of global_optimization.py:419-505 never starts.  Same steps, orders and
// floating-point operations as the Python form in multiview_stitcher_amd/param_resolution.py (reference view = first maximum of
// the per-node quality sums in numpy's summation order, networkx's edge iteration order of the bead graph, node sweeps by
// descending degree, mvs_beads_translation_sweeps, RMS bead residual per edge with numpy's pairwise mean); anything else --
// several components, an edge whose residual stays at or above abs_tol -- is reported as MVS_ERR_UNSUPPORTED and resolved by
// the Python form.  See include/mvs_hip.h for the contract; tests/test_resolve_native.py compares the two.
namespace {
// np.sum of a float64 vector (numpy's pairwise summation, n <= 128: eight running sums combined as a tree, the tail added in order)
bool numpy_sum(const std::vector<double>& a, double* out) {
    const size_t n = a.size();
    if (n > 128) return false;
    if (n < 8) {
        double r = -0.0;
        for (double v : a) r += v;
        *out = r;
        return true;
    }
    double r[8];
    for (int j = 0; j < 8; ++j) r[j] = a[(size_t)j];
    size_t i = 8;
    for (; i < n - (n % 8); i += 8)
        for (int j = 0; j < 8; ++j) r[j] += a[i + (size_t)j];
    double res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
    for (; i < n; ++i) res += a[i];
    *out = res;
    return true;
}
}   // namespace

extern "C" int mvs_resolve_translations(int32_t ndim, int32_t n_views, int32_t n_edges, const int32_t* edges, const double* pair_t,
                                        const double* quality, const double* bbox_lo, const double* bbox_hi, const double* spacing,
                                        int32_t reference_view, int32_t max_iter, double rel_tol, double abs_tol, double* translations_out,
                                        double* edge_rms_out, double* mean_hist, double* max_hist, int32_t* n_iter_out, int32_t* ref_out) {
    if (ndim < 1 || ndim > 3 || n_views < 1 || n_edges < 1 || max_iter < 1 || !edges || !pair_t || !quality || !bbox_lo || !bbox_hi || !spacing ||
        !translations_out || !edge_rms_out || !mean_hist || !max_hist || !n_iter_out)
        return MVS_ERR_INVALID_ARG;
    const int n = n_views, ne = n_edges, nb = 1 << ndim;
    // ---- the registration graph: edges (i < j) in insertion order, every unordered pair once, one component with all views ----
    std::vector<std::vector<int>> gadj((size_t)n), gadj_e((size_t)n);
    for (int e = 0; e < ne; ++e) {
        const int a = edges[2 * e], b = edges[2 * e + 1];
        if (a < 0 || b >= n || a >= b) return MVS_ERR_UNSUPPORTED;
        for (int m : gadj[(size_t)a])
            if (m == b) return MVS_ERR_UNSUPPORTED;          // (a duplicate would overwrite the first result in the Python form)
        gadj[(size_t)a].push_back(b); gadj_e[(size_t)a].push_back(e);
        gadj[(size_t)b].push_back(a); gadj_e[(size_t)b].push_back(e);
        for (int d = 0; d < ndim; ++d)
            if (!std::isfinite(pair_t[(size_t)e * ndim + d]) || !std::isfinite(bbox_lo[(size_t)e * ndim + d]) || !std::isfinite(bbox_hi[(size_t)e * ndim + d]))
                return MVS_ERR_UNSUPPORTED;
    }
    {
        std::vector<char> seen((size_t)n, 0);
        std::vector<int> stack{0};
        seen[0] = 1;
        int cnt = 1;
        while (!stack.empty()) {
            const int v = stack.back();
            stack.pop_back();
            for (int w : gadj[(size_t)v])
                if (!seen[(size_t)w]) { seen[(size_t)w] = 1; ++cnt; stack.push_back(w); }
        }
        if (cnt != n) return MVS_ERR_UNSUPPORTED;
    }
    // ---- abs_tol: the voxel diagonal, maximum over views (global_optimization.py:104-121) ----
    if (!(abs_tol >= 0.0)) {
        double best = 0.0;
        for (int v = 0; v < n; ++v) {
            double s = -0.0;
            for (int d = 0; d < ndim; ++d) s += std::pow(spacing[(size_t)v * ndim + d], 2.0);
            const double diag = std::pow(s, 0.5);
            if (v == 0 || diag > best) best = diag;
        }
        abs_tol = best;
    }
    // ---- reference view: the given one, or the first maximum of the per-node sums of edge qualities (mv_graph.py:341-352) ----
    int ref_node = reference_view;
    if (ref_node < 0 || ref_node >= n) {
        if (reference_view >= n) return MVS_ERR_UNSUPPORTED;       // (a label outside the graph: the Python form's quirk decides)
        double best = 0.0;
        std::vector<double> w;
        for (int v = 0; v < n; ++v) {
            w.clear();
            for (int e : gadj_e[(size_t)v]) w.push_back(quality[(size_t)e]);
            double tot;
            if (!numpy_sum(w, &tot)) return MVS_ERR_UNSUPPORTED;
            if (v == 0 || tot > best) { best = tot; ref_node = v; }
        }
    }
    if (ref_out) *ref_out = ref_node;
    // ---- the bead graph with networkx's orders: edges added in the registration graph's edge ITERATION order ----
    std::vector<std::vector<AdjEntry>> badj((size_t)n);
    for (int v = 0; v < n; ++v)
        for (size_t k = 0; k < gadj[(size_t)v].size(); ++k) {
            const int m = gadj[(size_t)v][k], e = gadj_e[(size_t)v][k];
            if (m < v) continue;                                 // (seen: every earlier node is done)
            badj[(size_t)v].push_back(AdjEntry{m, e});
            badj[(size_t)m].push_back(AdjEntry{v, e});
        }
    std::vector<int32_t> en;           // the bead graph's edges() order: (a, b), a < b
    std::vector<int> eorig;
    for (int v = 0; v < n; ++v)
        for (const AdjEntry& a : badj[(size_t)v])
            if (a.nb >= v) { en.push_back(v); en.push_back(a.nb); eorig.push_back(a.eid); }
    // ---- virtual beads (param_resolution/utils.py:42-78): corners of the overlap box, and their images under the pair translation ----
    const size_t per_edge = (size_t)nb * (size_t)ndim;
    std::vector<double> ba((size_t)ne * per_edge), bb((size_t)ne * per_edge);
    for (int k = 0; k < ne; ++k) {
        const int e = eorig[(size_t)k];
        for (int b = 0; b < nb; ++b)
            for (int d = 0; d < ndim; ++d) {
                const double lo = bbox_lo[(size_t)e * ndim + d], hi = bbox_hi[(size_t)e * ndim + d];
                const double gv = (double)((b >> (ndim - 1 - d)) & 1);
                const double vert = gv * (hi - lo) + lo;
                ba[(size_t)k * per_edge + (size_t)b * ndim + d] = vert;
                bb[(size_t)k * per_edge + (size_t)b * ndim + d] = vert + pair_t[(size_t)e * ndim + d];
            }
    }
    // ---- node sweeps: most connected first (stable over node order) ----
    std::vector<int32_t> order((size_t)n);
    for (int v = 0; v < n; ++v) order[(size_t)v] = v;
    std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return badj[(size_t)a].size() > badj[(size_t)b].size(); });
    std::vector<double> t((size_t)n * ndim, 0.0), res((size_t)ne * nb);
    int32_t n_it = 0;
    int rc = mvs_beads_translation_sweeps(ndim, n, ne, en.data(), ba.data(), bb.data(), nb, order.data(), ref_node, max_iter, rel_tol, t.data(),
                                          res.data(), mean_hist, max_hist, &n_it);
    if (rc) return rc;
    *n_iter_out = n_it;
    if (n_it < 1) return MVS_ERR_UNSUPPORTED;
    if (ne >= 2 && !(max_hist[n_it - 1] < abs_tol)) return MVS_ERR_UNSUPPORTED;        // the edge-removal loop would start
    for (size_t i = 0; i < t.size(); ++i) translations_out[i] = t[i];
    // ---- RMS bead residual per edge of the registration graph (param_resolution/utils.py:81-101), in its edge order ----
    for (int k = 0; k < ne; ++k) {
        const int e = eorig[(size_t)k], a = en[(size_t)2 * k], b = en[(size_t)2 * k + 1];
        std::vector<double> sq((size_t)nb);
        for (int bd = 0; bd < nb; ++bd) {
            double s = -0.0;
            for (int d = 0; d < ndim; ++d) {
                const double dv = (ba[(size_t)k * per_edge + (size_t)bd * ndim + d] + t[(size_t)a * ndim + d]) -
                                  (bb[(size_t)k * per_edge + (size_t)bd * ndim + d] + t[(size_t)b * ndim + d]);
                s += dv * dv;
            }
            sq[(size_t)bd] = s;
        }
        double tot;
        numpy_sum(sq, &tot);
        edge_rms_out[(size_t)e] = std::sqrt(tot / (double)nb);
    }
    return MVS_OK;
}
