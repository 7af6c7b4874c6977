// mvs_pair.hip -- one call for the reference's phase_correlation_registration (gfx950 host side).
//
// mvs_register_crops == registration.phase_correlation_registration
// (src/multiview_stitcher/registration.py:353-565) for two same-shape float32 overlap crops (NaN = outside the view):
// intensity normalisation (:381-389), phase correlation with and without phase normalisation (:413-431), the zero-shift
// candidate of the masked variant when a crop holds NaNs (:433-443, quirk Q1), candidate enumeration (:453-477), candidate
// scoring (:493-556) and the selection by nanargmax with the reference's list bookkeeping (quirk Q3: the `continue` of
// :530-533 appends nothing, the arg-max index then addresses the UNFILTERED candidate list).  Every voxel-sized step is one
// of the library's own entry points; what this file adds is the control flow between them, which otherwise costs ~0.5 ms of
// interpreter time per image pair (with the GIL held) in the Python mirror -- multiview_stitcher_amd.registration keeps
// that mirror for the debug outputs and for custom keyword arguments.
#include "mvs_internal.h"

int mvs_rescale_pair_device(MvsContext* c, const float* in0, const float* in1, long long n, float* out0, float* out1,
                            float mn[2], float mx[2], long long nvalid[2], long long n_not_u16[2], const MvsCropStats* parked);   // mvs_reg.hip

#include <algorithm>
#include <cmath>
#include <vector>

static int register_crops_impl(int device, const float* fixed, const float* moving, int32_t mem, int32_t ndim, const int64_t shape[3],
                               int32_t upsample_factor, int32_t region_mode, int32_t constant_check, double t_out[3], double* quality_out,
                               int32_t* status_out, int32_t* n_candidates_out, const MvsCropStats* parked);

extern "C" int mvs_register_crops(int device, const float* fixed, const float* moving, int32_t mem, int32_t ndim, const int64_t shape[3],
                                  int32_t upsample_factor, int32_t region_mode, int32_t constant_check, double t_out[3], double* quality_out,
                                  int32_t* status_out, int32_t* n_candidates_out) {
    return register_crops_impl(device, fixed, moving, mem, ndim, shape, upsample_factor, region_mode, constant_check, t_out, quality_out, status_out,
                               n_candidates_out, nullptr);
}

// (parked: statistics the crop kernels of mvs_register_views left in the mailbox -- an argument of this call, not context state)
static int register_crops_impl(int device, const float* fixed, const float* moving, int32_t mem, int32_t ndim, const int64_t shape[3],
                               int32_t upsample_factor, int32_t region_mode, int32_t constant_check, double t_out[3], double* quality_out,
                               int32_t* status_out, int32_t* n_candidates_out, const MvsCropStats* parked) {
    MvsContext* c;
    int rc = mvs_check_ready(device, &c);
    if (rc) return rc;
    std::lock_guard<std::recursive_mutex> lock(c->mu);
    if (!fixed || !moving || !shape || !t_out || !quality_out || !status_out)
        return mvs_fail(c, MVS_ERR_INVALID_ARG, "mvs_register_crops: NULL argument");
    if (ndim != 2 && ndim != 3) return mvs_fail(c, MVS_ERR_INVALID_ARG, "mvs_register_crops: ndim must be 2 or 3");
    if (ndim == 2 && shape[0] != 1) return mvs_fail(c, MVS_ERR_INVALID_ARG, "mvs_register_crops: 2D needs shape[0]==1");
    if (region_mode < -1 || region_mode > 1) return mvs_fail(c, MVS_ERR_INVALID_ARG, "mvs_register_crops: region_mode must be -1, 0 or 1");
    const int64_t n = shape[0] * shape[1] * shape[2];
    if (n < 1) return mvs_fail(c, MVS_ERR_INVALID_ARG, "mvs_register_crops: empty shape");
    const int k0 = 3 - ndim;
    t_out[0] = t_out[1] = t_out[2] = 0.0;
    *quality_out = NAN;
    *status_out = 0;
    if (n_candidates_out) *n_candidates_out = 0;

    // ---- normalise (registration.py:381-389); also nanmin / nanmax / #valid of the inputs ----
    float* r0 = (float*)mvs_scratch(c, 9, (size_t)n * 4);
    float* r1 = (float*)mvs_scratch(c, 10, (size_t)n * 4);
    if (!r0 || !r1) return mvs_alloc_failed(c);
    float min0, max0, min1, max1;
    int64_t nv0, nv1;
    const float* raw_keys0 = nullptr;
    const float* raw_keys1 = nullptr;
    if (mem == MVS_MEM_DEVICE) {      // one host round trip for both crops
        MVS_HIP_TRY(c, hipSetDevice(mvs_hip_device(device)));
        float mn[2], mx[2];
        long long nv[2], not_u16[2];
        rc = mvs_rescale_pair_device(c, fixed, moving, n, r0, r1, mn, mx, nv, not_u16, parked);
        if (rc) return rc;
        // crops of integer tiles on the fixed grid hold 16-bit integers: rescaling is strictly increasing and one-to-one on them,
        // so the rank order (and every tie) of the rescaled image equals that of the raw integers -> 16-bit sort keys
        raw_keys0 = (not_u16[0] == 0) ? fixed : nullptr;
        raw_keys1 = (not_u16[1] == 0) ? moving : nullptr;
        min0 = mn[0]; max0 = mx[0]; nv0 = nv[0];
        min1 = mn[1]; max1 = mx[1]; nv1 = nv[1];
    } else {
        rc = mvs_rescale_intensity(device, fixed, mem, n, r0, MVS_MEM_DEVICE, &min0, &max0, &nv0);
        if (rc) return rc;
        rc = mvs_rescale_intensity(device, moving, mem, n, r1, MVS_MEM_DEVICE, &min1, &max1, &nv1);
        if (rc) return rc;
    }
    if (constant_check && (min0 == max0 || min1 == max1)) {   // dispatch_pairwise_reg_func's guard (registration.py:1500-1520)
        *status_out = 2;
        return MVS_OK;
    }
    const bool has_nan = (nv0 < n) || (nv1 < n);
    if (region_mode < 0) region_mode = has_nan ? 1 : 0;       // "intersection" with NaNs, else "union"

    // ---- the two phase correlations (registration.py:413-431); shifts come back as float32 values ----
    const int32_t norms[2] = {1, 0};
    double shifts[6];
    rc = mvs_phasecorr_multi(device, r0, r1, MVS_MEM_DEVICE, ndim, shape, norms, 2, upsample_factor, shifts, nullptr, nullptr);
    if (rc) return rc;
    std::vector<std::vector<float>> estimates;
    for (int e = 0; e < 2; ++e) {
        std::vector<float> s(ndim);
        for (int k = 0; k < ndim; ++k) s[k] = (float)shifts[3 * e + k0 + k];
        estimates.push_back(s);
    }
    if (has_nan) estimates.push_back(std::vector<float>(ndim, 0.f));   // the masked variant's zero shift (Q1)

    // ---- data_range / im1_min after rescaling: the values present are min -> 0 and max -> 1 per image ----
    auto rescaled_range = [](float mn, float mx, int64_t nv, float& lo, float& hi) {
        if (nv == 0) { lo = hi = NAN; return; }
        if (mx != mn) { lo = 0.f; hi = 1.f; } else { lo = hi = mn; }
    };
    float lo0, hi0, lo1, hi1;
    rescaled_range(min0, max0, nv0, lo0, hi0);
    rescaled_range(min1, max1, nv1, lo1, hi1);
    auto nanmaxf = [](float a, float b) { return (a != a) ? b : (b != b) ? a : std::max(a, b); };
    auto nanminf = [](float a, float b) { return (a != a) ? b : (b != b) ? a : std::min(a, b); };
    const float data_range = nanmaxf(hi0, hi1) - nanminf(lo0, lo1);    // float32 arithmetic like numpy
    const double im1_min = (double)lo1;

    // ---- candidate enumeration (registration.py:453-477): float32 arithmetic, itertools.product order ----
    float max_shift = 0.f;
    for (int k = 0; k < ndim; ++k) max_shift = std::max(max_shift, (float)shape[k0 + k]);
    std::vector<double> cand;    // n_cand x ndim
    for (const std::vector<float>& sc : estimates) {
        int nvar[3], idx[3] = {0, 0, 0};
        float var[3][4];
        for (int d = 0; d < ndim; ++d) {
            const float nn = (float)shape[k0 + d];
            if (sc[d] == 0.f) { nvar[d] = 1; var[d][0] = sc[d]; }
            else { nvar[d] = 4; var[d][0] = sc[d]; var[d][1] = -sc[d]; var[d][2] = -(sc[d] - nn); var[d][3] = -sc[d] - nn; }
        }
        for (;;) {
            float amax = 0.f;
            for (int d = 0; d < ndim; ++d) amax = std::max(amax, std::fabs(var[d][idx[d]]));
            if (amax < max_shift)
                for (int d = 0; d < ndim; ++d) cand.push_back((double)var[d][idx[d]]);
            int d = ndim - 1;                                    // last axis fastest
            while (d >= 0 && ++idx[d] == nvar[d]) { idx[d] = 0; --d; }
            if (d < 0) break;
        }
    }
    const int n_all = (int)(cand.size() / (size_t)ndim);
    if (n_candidates_out) *n_candidates_out = n_all;
    if (n_all == 0) {            // Q2: the reference returns `[zeros(ndim)]`
        *status_out = 1;
        return MVS_OK;
    }

    // ---- score every distinct candidate once (the two estimates usually agree), scatter back ----
    std::vector<int> uniq_of(n_all, -1);
    std::vector<double> uniq;
    for (int i = 0; i < n_all; ++i) {
        const int nu = (int)(uniq.size() / (size_t)ndim);
        for (int u = 0; u < nu && uniq_of[i] < 0; ++u) {
            bool same = true;
            for (int d = 0; d < ndim; ++d) same = same && (uniq[(size_t)u * ndim + d] == cand[(size_t)i * ndim + d]);
            if (same) uniq_of[i] = u;
        }
        if (uniq_of[i] < 0) {
            uniq_of[i] = nu;
            for (int d = 0; d < ndim; ++d) uniq.push_back(cand[(size_t)i * ndim + d]);
        }
    }
    const int n_uniq = (int)(uniq.size() / (size_t)ndim);
    std::vector<double> ssim_u(n_uniq), spear_u(n_uniq);
    std::vector<int32_t> code_u(n_uniq);
    // rescaled crops are finite iff the inputs were: no NaN (all voxels counted) and finite extrema (no inf)
    MvsScoreOpts so;
    so.both_crops_finite = !has_nan && std::isfinite(min0) && std::isfinite(max0) && std::isfinite(min1) && std::isfinite(max1) &&
                           !c->materialize_shifts;
    so.raw_u16_keys[0] = c->materialize_shifts ? nullptr : raw_keys0;
    so.raw_u16_keys[1] = c->materialize_shifts ? nullptr : raw_keys1;
    so.raw_range[0] = min0; so.raw_range[1] = max0; so.raw_range[2] = min1; so.raw_range[3] = max1;
    // only the arg-max candidate (and its rank correlation) leaves this function: the scoring may stop a candidate as soon as
    // it provably cannot win (see the pruned search in mvs_score_candidates_impl); the rescaled values lie in [lo, hi]
    so.argmax_only = true;
    so.value_bound = (double)std::max(std::max(std::fabs(lo0), std::fabs(hi0)), std::max(std::fabs(lo1), std::fabs(hi1)));
    const double cand_vol0 = c->reg_cand_volumes;
    rc = mvs_score_candidates_impl(device, r0, r1, MVS_MEM_DEVICE, ndim, shape, uniq.data(), n_uniq, region_mode, (double)data_range, im1_min, 0,
                                   ssim_u.data(), spear_u.data(), code_u.data(), so);
    if (rc) return rc;
    int n_scored = 0;      // candidates that went through the shift + SSIM kernels (the others were rejected from their boxes)
    for (int u = 0; u < n_uniq; ++u) n_scored += (code_u[u] == 0) ? 1 : 0;
    // candidate volumes the SSIM walk went through: a candidate the pruned search stopped counts the fraction it was scored on
    double cand_vol = c->reg_cand_volumes - cand_vol0;
    if (!(cand_vol > 0.0)) { cand_vol = (double)n_scored; c->reg_cand_volumes = cand_vol0 + cand_vol; }   // (paths that do not count themselves)
    c->reg_alg_bytes += (double)n * (2.0 * 28.0 + 20.0 * cand_vol + 64.0);
    c->reg_alg_bytes_full += (double)n * (2.0 * 28.0 + 20.0 * (double)n_scored + 64.0);
    c->reg_pairs += 1;
    c->reg_candidates += n_scored;

    // ---- metric lists as the reference builds them (code 2 appends nothing), nanargmax, Q3 indexing ----
    int best_pos = -1, pos = 0;
    double best = 0.0, best_quality = NAN;
    for (int i = 0; i < n_all; ++i) {
        const int u = uniq_of[i];
        if (code_u[u] == 2) continue;
        const double sv = ssim_u[u];
        if (sv == sv && (best_pos < 0 || sv > best)) { best = sv; best_pos = pos; best_quality = spear_u[u]; }
        ++pos;
    }
    if (best_pos < 0) {          // np.nanargmax of an empty / all-NaN list raises in the reference
        *status_out = 3;
        return MVS_OK;
    }
    for (int d = 0; d < ndim; ++d) t_out[k0 + d] = cand[(size_t)best_pos * ndim + d];   // index into the UNFILTERED list (Q3)
    *quality_out = best_quality;
    return MVS_OK;
}


// One call per image pair: both overlap crops are resampled onto the fixed view's grid (sims_to_intrinsic_coord_system,
// registration.py:280-350: order 1, NaN outside) into library scratch and registered (mvs_register_crops), without a host
// round trip in between.  Replaces three calls, two crop allocations and two waits of the Python flow.
static int register_views_impl(int device, const mvs_view_t* fixed_view, const mvs_view_t* moving_view, int32_t ndim, const int64_t out_shape[3],
                               int32_t upsample_factor, int32_t region_mode, int32_t constant_check, const int32_t* bin, double t_out[3],
                               double* quality_out, int32_t* status_out, int32_t* n_candidates_out);

extern "C" int mvs_register_views(int device, const mvs_view_t* fixed_view, const mvs_view_t* moving_view, int32_t ndim,
                                  const int64_t out_shape[3], int32_t upsample_factor, int32_t region_mode, int32_t constant_check,
                                  double t_out[3], double* quality_out, int32_t* status_out, int32_t* n_candidates_out) {
    return register_views_impl(device, fixed_view, moving_view, ndim, out_shape, upsample_factor, region_mode, constant_check, nullptr, t_out,
                               quality_out, status_out, n_candidates_out);
}

// bin != NULL: the views are windows of RAW integer tiles and the crops are taken with the registration binning applied on the fly
// (mvs_crop_bin_impl: block mean + whole-pixel translation in one pass, no binned copy of the tiles)
static int register_views_impl(int device, const mvs_view_t* fixed_view, const mvs_view_t* moving_view, int32_t ndim, const int64_t out_shape[3],
                               int32_t upsample_factor, int32_t region_mode, int32_t constant_check, const int32_t* bin, double t_out[3],
                               double* quality_out, int32_t* status_out, int32_t* n_candidates_out) {
    MvsContext* c;
    int rc = mvs_check_ready(device, &c);
    if (rc) return rc;
    std::lock_guard<std::recursive_mutex> lock(c->mu);
    if (!fixed_view || !moving_view || !out_shape) return mvs_fail(c, MVS_ERR_INVALID_ARG, "mvs_register_views: NULL argument");
    for (int k = 0; k < 3; ++k)
        if (out_shape[k] < 1) return mvs_fail(c, MVS_ERR_INVALID_ARG, "mvs_register_views: bad out_shape");
    const int64_t n = out_shape[0] * out_shape[1] * out_shape[2];
    float* crop0 = (float*)mvs_scratch(c, 11, (size_t)n * 4);
    float* crop1 = (float*)mvs_scratch(c, 12, (size_t)n * 4);
    if (!crop0 || !crop1) return mvs_alloc_failed(c);
    // crops of integer tiles under whole-pixel translations: the crop kernel reduces min / max / #valid of what it writes, so the
    // normalisation needs no pass of its own over the crops (same partial layout and block count as mvs_rescale_pair_device)
    const int nb_stats = (int)std::min<int64_t>(std::min<int64_t>((n + 255) / 256, 256 * 8), 512);
    void *mb_host = nullptr, *mb_dev = nullptr;
    rc = mvs_mailbox(c, (size_t)nb_stats * 32, &mb_host, &mb_dev);
    if (rc) return rc;
    MvsCropStats stats;                 // lives for this call only: nothing to clear on any return path
    stats.base = c->reg_unfused ? nullptr : (char*)mb_dev;
    stats.nb = nb_stats;
    stats.gen = c->mbox_gen;
    MvsResampleOpts ro;
    ro.defer_sync = true;
    ro.stats = &stats;
    ro.stats_k = 0;
    rc = bin ? mvs_crop_bin_impl(device, fixed_view, bin, out_shape, crop0, ro) : mvs_resample_impl(device, fixed_view, out_shape, 1, NAN, crop0, MVS_MEM_DEVICE, ro);
    if (rc) return rc;
    ro.stats_k = 1;
    rc = bin ? mvs_crop_bin_impl(device, moving_view, bin, out_shape, crop1, ro) : mvs_resample_impl(device, moving_view, out_shape, 1, NAN, crop1, MVS_MEM_DEVICE, ro);
    if (rc) return rc;
    return register_crops_impl(device, crop0, crop1, MVS_MEM_DEVICE, ndim, out_shape, upsample_factor, region_mode, constant_check, t_out,
                               quality_out, status_out, n_candidates_out, &stats);
}


// ---- all pairs of a mosaic in one call --------------------------------------------------------------------------------------------
// registration.compute_pairwise_registrations (registration.py:2622-2714) hands every pair to its own dask task; the Python
// mirror ran them from a pool of 16 interpreter threads, one context lane each -- and spent a third of the pairwise wall time
// in the interpreter (plans, slab views, ctypes marshalling: ~0.2 ms per pair with the GIL held, 144 pairs per mosaic).  Here the
// same loop lives in the library: mvs_plan_pairs derives the crop windows and pixel affines of all pairs (host code, the float
// operations of the Python form in the same order), mvs_register_pairs runs mvs_register_views for every job on `n_lanes` native
// worker threads, each driving its own context lane (stream, scratch, lock) of the GPU.
#include <atomic>
#include <condition_variable>
#include <thread>

namespace {
// Python's bisect.bisect_left / bisect_right on a float64 array
inline int64_t bisect_left(const double* c, int64_t n, double x) {
    int64_t lo = 0, hi = n;
    while (lo < hi) { const int64_t mid = (lo + hi) / 2; if (c[mid] < x) lo = mid + 1; else hi = mid; }
    return lo;
}
inline int64_t bisect_right(const double* c, int64_t n, double x) {
    int64_t lo = 0, hi = n;
    while (lo < hi) { const int64_t mid = (lo + hi) / 2; if (x < c[mid]) hi = mid; else lo = mid + 1; }
    return lo;
}
inline double around10(double x) { return std::rint(x * 1e10) / 1e10; }      // np.around(x, 10)
inline double pymax(double a, double b) { return b > a ? b : a; }             // max(a, b): the first unless the second is larger
inline double pymin(double a, double b) { return b < a ? b : a; }
}   // namespace

// One plan per pair of translated views (register_pair_of_msims -> sims_to_intrinsic_coord_system -> get_pixel_affine,
// registration.py:194-350 with transformation.py:37-83, for views whose transform is a pure translation): the index window of
// each view the overlap (+ one sample and 1e-6 on either side) selects, the output grid = the fixed view's overlap grid, and
// the diagonal pixel affines (rounded to 10 decimals, near-integer offsets snapped) of both crops.
extern "C" int mvs_plan_pairs(int32_t ndim, int32_t n_views, const double* const* coords, const int64_t* coord_len, const double* translation,
                              const double* tol, int32_t n_pairs, const int32_t* pairs, int64_t* windows_out, double* out_origin_out,
                              double* out_spacing_out, int64_t* out_shape_out, double* matrix_diag_out, double* offset_out, int32_t* status_out) {
    if (ndim < 1 || ndim > 3 || n_views < 1 || n_pairs < 0 || !coords || !coord_len || !translation || (n_pairs > 0 && !pairs) || !windows_out ||
        !out_origin_out || !out_spacing_out || !out_shape_out || !matrix_diag_out || !offset_out || !status_out)
        return MVS_ERR_INVALID_ARG;
    for (int v = 0; v < n_views * ndim; ++v)
        if (!coords[v] || coord_len[v] < 1) return MVS_ERR_INVALID_ARG;
    for (int p = 0; p < n_pairs; ++p) {
        const int v[2] = {pairs[2 * p], pairs[2 * p + 1]};
        status_out[p] = 1;                                  // no overlap until proven otherwise
        if (v[0] < 0 || v[1] < 0 || v[0] >= n_views || v[1] >= n_views) return MVS_ERR_INVALID_ARG;
        double lowers[2][3], uppers[2][3];
        bool ok = true;
        for (int k = 0; k < ndim && ok; ++k) {
            double lo2[2], hi2[2];
            for (int i = 0; i < 2; ++i) {
                const double* c = coords[(size_t)v[i] * ndim + k];
                int64_t n = coord_len[(size_t)v[i] * ndim + k];
                double o = c[0];
                const double s = n > 1 ? c[1] - c[0] : 1.0;
                if (tol) { n = n + (int64_t)std::ceil(2 * tol[k] / s); o = o - tol[k]; }
                const double t = translation[(size_t)v[i] * ndim + k];
                lo2[i] = o + t;
                hi2[i] = ((double)(n - 1) * 1.0 * s + o) + t;
            }
            const double lo = pymax(lo2[0], lo2[1]), hi = pymin(hi2[0], hi2[1]);
            if (hi < lo) { ok = false; break; }
            const double up = 1.0 * (hi - lo) + lo;
            for (int i = 0; i < 2; ++i) {
                const double t = translation[(size_t)v[i] * ndim + k];
                lowers[i][k] = lo + (-t);
                uppers[i][k] = up + (-t);
            }
        }
        if (!ok) continue;
        double origins[2][3], spacings[2][3];
        for (int i = 0; i < 2 && ok; ++i)
            for (int k = 0; k < ndim; ++k) {
                const double* c = coords[(size_t)v[i] * ndim + k];
                const int64_t n = coord_len[(size_t)v[i] * ndim + k];
                const double gs = n > 1 ? c[1] - c[0] : 1.0;
                const double start = lowers[i][k] - 1e-6 - gs, stop = uppers[i][k] + 1e-6 + gs;
                const int64_t a = bisect_left(c, n, start), b = bisect_right(c, n, stop);
                if (b <= a) { ok = false; break; }
                windows_out[(((size_t)p * 2 + i) * 3 + k) * 2] = a;
                windows_out[(((size_t)p * 2 + i) * 3 + k) * 2 + 1] = b;
                origins[i][k] = c[a];
                spacings[i][k] = (b - a > 1) ? c[a + 1] - c[a] : 1.0;
            }
        if (!ok) continue;
        for (int k = 0; k < ndim; ++k) {
            const double osp = pymax(spacings[0][k], spacings[1][k]);
            const double oo = lowers[0][k];
            out_spacing_out[(size_t)p * 3 + k] = osp;
            out_origin_out[(size_t)p * 3 + k] = oo;
            out_shape_out[(size_t)p * 3 + k] = (int64_t)std::floor((uppers[0][k] - lowers[0][k]) / osp + 1);
            const double t_rel = translation[(size_t)v[0] * ndim + k] + (-translation[(size_t)v[1] * ndim + k]);     // inv(A2) @ A1
            for (int i = 0; i < 2; ++i) {
                const double tt = i == 0 ? 0.0 : t_rel;
                matrix_diag_out[((size_t)p * 2 + i) * 3 + k] = around10((1.0 * osp) / spacings[i][k]);
                const double val = around10(((tt + 0.0) - (origins[i][k] - oo)) / spacings[i][k]);
                const double r = std::rint(val);
                offset_out[((size_t)p * 2 + i) * 3 + k] = std::fabs(val - r) <= 1e-6 ? r : val;
            }
        }
        status_out[p] = 0;
    }
    return MVS_OK;
}

namespace {
// persistent worker threads of mvs_register_pairs (one set per process; a batch is handed over under the mutex)
struct PairBatch {
    int device = 0;
    int n_pairs = 0;
    mvs_pair_job_t* jobs = nullptr;
    int ndim = 3, upsample = 2, region_mode = -1, constant_check = 0;
    double* t_out = nullptr;
    double* quality_out = nullptr;
    int32_t* status_out = nullptr;
    int32_t* ncand_out = nullptr;
    int32_t* rc_out = nullptr;
    std::atomic<int> next{0};
};

void run_pairs_on_lane(PairBatch* b, int lane) {
    const int dev = (b->device & 0xff) | (lane << 8);
    int init_rc = mvs_init(dev);
    for (;;) {
        const int p = b->next.fetch_add(1);
        if (p >= b->n_pairs) break;
        mvs_pair_job_t& j = b->jobs[p];
        int rc = init_rc;
        for (int k = 0; k < 2 && !rc; ++k)
            if (j.wait_ticket[k]) rc = mvs_event_wait(dev, j.wait_ticket[k]);
        int32_t status = 0, ncand = 0;
        double t[3] = {0.0, 0.0, 0.0}, q = NAN;
        if (!rc)
            rc = register_views_impl(dev, &j.fixed, &j.moving, b->ndim, j.out_shape, b->upsample, b->region_mode, b->constant_check,
                                     j.bin[0] > 0 ? j.bin : nullptr, t, &q, &status, &ncand);
        for (int k = 0; k < 3; ++k) b->t_out[(size_t)p * 3 + k] = t[k];
        b->quality_out[p] = q;
        b->status_out[p] = status;
        if (b->ncand_out) b->ncand_out[p] = ncand;
        b->rc_out[p] = rc;
        if (rc) b->status_out[p] = -(lane + 1);      // (which lane's mvs_last_error holds the message)
        if ((j.flags & 1) && !rc) {                  // timeline: when this pair's last kernel finished
            uint64_t m = 0;
            if (mvs_mark(dev, &m) == MVS_OK) j.wait_ticket[0] = m;
        }
    }
}

struct PairPool {
    std::mutex mu;
    std::condition_variable cv_work, cv_done;
    std::vector<std::thread> threads;
    PairBatch* batch = nullptr;
    uint64_t generation = 0;
    int active_lanes = 0, pending = 0;
    std::mutex call_mu;              // one batch at a time

    void worker(int lane) {
        uint64_t seen = 0;
        for (;;) {
            PairBatch* b;
            {
                std::unique_lock<std::mutex> lk(mu);
                cv_work.wait(lk, [&] { return generation != seen; });
                seen = generation;
                b = lane < active_lanes ? batch : nullptr;
            }
            if (!b) continue;
            run_pairs_on_lane(b, lane);
            {
                std::lock_guard<std::mutex> lk(mu);
                if (--pending == 0) cv_done.notify_all();
            }
        }
    }
};
PairPool* pair_pool(int device) {
    // one set of workers per GPU: host threads that drive different devices of one process (executors.DevicePairExecutor) do not queue
    // behind each other (never destroyed: the threads wait for work until the process ends)
    static PairPool* pools[MVS_MAX_DEVICES] = {};
    static std::mutex pools_mu;
    std::lock_guard<std::mutex> lk(pools_mu);
    PairPool*& pool = pools[mvs_hip_device(device) % MVS_MAX_DEVICES];
    if (!pool) pool = new PairPool();
    return pool;
}
}   // namespace

extern "C" int mvs_register_pairs(int device, int32_t n_pairs, mvs_pair_job_t* jobs, int32_t ndim, int32_t upsample_factor,
                                  int32_t region_mode, int32_t constant_check, int32_t n_lanes, double* t_out, double* quality_out,
                                  int32_t* status_out, int32_t* n_candidates_out, int32_t* rc_out) {
    MvsContext* c;
    int rc = mvs_check_ready(device, &c);
    if (rc) return rc;
    if (n_pairs < 0 || (n_pairs > 0 && (!jobs || !t_out || !quality_out || !status_out || !rc_out)))
        return mvs_fail(c, MVS_ERR_INVALID_ARG, "mvs_register_pairs: NULL argument");
    if (ndim != 2 && ndim != 3) return mvs_fail(c, MVS_ERR_INVALID_ARG, "mvs_register_pairs: ndim must be 2 or 3");
    if (device >> 8) return mvs_fail(c, MVS_ERR_INVALID_ARG, "mvs_register_pairs: pass the device without a lane (the call uses lanes 0 .. n_lanes - 1)");
    if (n_pairs == 0) return MVS_OK;
    n_lanes = std::max(1, std::min(std::min((int)n_lanes, (int)MVS_MAX_LANES), (int)n_pairs));
    PairBatch b;
    b.device = device; b.n_pairs = n_pairs; b.jobs = jobs; b.ndim = ndim; b.upsample = upsample_factor; b.region_mode = region_mode;
    b.constant_check = constant_check; b.t_out = t_out; b.quality_out = quality_out; b.status_out = status_out; b.ncand_out = n_candidates_out;
    b.rc_out = rc_out;
    if (n_lanes == 1) {
        run_pairs_on_lane(&b, 0);
    } else {
        PairPool* pool = pair_pool(device);
        std::lock_guard<std::mutex> call(pool->call_mu);
        {
            std::lock_guard<std::mutex> lk(pool->mu);
            while ((int)pool->threads.size() < n_lanes - 1) {
                const int lane = (int)pool->threads.size() + 1;          // (the calling thread drives lane 0)
                pool->threads.emplace_back([pool, lane] { pool->worker(lane); });
                pool->threads.back().detach();
            }
            pool->batch = &b;
            pool->active_lanes = n_lanes;
            pool->pending = n_lanes - 1;
            ++pool->generation;
        }
        pool->cv_work.notify_all();
        run_pairs_on_lane(&b, 0);
        std::unique_lock<std::mutex> lk(pool->mu);
        pool->cv_done.wait(lk, [&] { return pool->pending == 0; });
        pool->batch = nullptr;
    }
    for (int p = 0; p < n_pairs; ++p)
        if (rc_out[p]) {
            const int lane = -status_out[p] - 1;
            MvsContext* lc = mvs_ctx((device & 0xff) | (std::max(lane, 0) << 8));
            return mvs_fail(c, rc_out[p], "mvs_register_pairs: pair %d failed: %s", p, lc ? lc->last_error.c_str() : "?");
        }
    return MVS_OK;
}
