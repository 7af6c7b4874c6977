// mvs_internal.h -- per-device context shared by the translation units of libmvs_hip.so.
// Not part of the ABI (that is include/mvs_hip.h).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <map>
#include <mutex>
#include <unordered_map>
#include <string>
#include <vector>

#include "mvs_hip.h"

#define MVS_MAX_DEVICES 16
// A context id is `device | lane << 8`: up to MVS_MAX_LANES independent contexts (stream, scratch, pool, lock) on one
// GPU, so that host threads working on independent units (pairs) overlap on the device instead of queueing on one lock.
#define MVS_MAX_LANES 16
static inline int mvs_hip_device(int id) { return id & 0xff; }
static inline int mvs_ctx_index(int id) { return ((id >> 8) & 0xff) * MVS_MAX_DEVICES + (id & 0xff); }

struct MvsScratch {
    void* ptr = nullptr;
    size_t cap = 0;
};

struct MvsContext {
    int device = -1;
    bool ready = false;
    hipStream_t own_stream = nullptr;
    hipStream_t stream = nullptr;  // the stream work is issued on (own or external)
    hipEvent_t ev_start = nullptr, ev_stop = nullptr;
    bool timing_valid = false;
    int ablate = 0;               // profiling only (see FuseParams)
    bool force_generic = false;   // debug/test switch: never take the translation fast path
    // measurement: algorithmic HBM bytes of the pairwise registrations run on this context since the last reset (SURVEY 8d:
    // 28 n per phase-correlation variant, 20 n per scored candidate, 64 n for the rank correlation; n = crop voxels)
    double reg_alg_bytes = 0.0;
    double reg_alg_bytes_full = 0.0;   // the same model with every scored candidate counted whole (what the reference's formulation moves)
    long long reg_pairs = 0, reg_candidates = 0, reg_pruned = 0;   // reg_pruned: candidates the pruned arg-max search left unfinished
    long long reg_slab_pairs = 0;      // phase correlations that took the three-pass form (mvs_fft_slab.hip)
    double reg_cand_volumes = 0.0;     // candidate volumes the SSIM walk actually went through (a pruned candidate counts its fraction)
    // the class kernels of one fuse launch run on side streams next to the main one (fork / join by events)
    hipStream_t aux_stream[4] = {nullptr, nullptr, nullptr, nullptr};
    hipEvent_t ev_fork = nullptr, ev_join[4] = {nullptr, nullptr, nullptr, nullptr};
    bool fuse_mixed = false;           // option "fuse_mixed": copy / one-view / two-view bricks in ONE launch over a space-ordered list
    bool serial_classes = false;       // option "serial_classes": launch them one after the other on the main stream instead
    bool reg_unfused = false;          // test switch: the phase correlation runs its separate launches (pack, cross power, peak search, min / max) instead of the fused passes
    bool ssim_f32 = true;              // option "ssim_f32" (default 1): the pruned arg-max search walks in float32 and re-walks near ties in float64
    long long reg_rewalks = 0;         // candidates the pruned search walked again in float64 (counter "reg_rewalks")
    bool ssim_prune = true;            // option "ssim_prune" (default 1; environment MVS_SSIM_PRUNE=0 turns it off for new contexts)
    bool fft_no_pair = false;          // test switch: the first inverse pass of the phase correlation takes its lines in flat order (no partner pairs)
    bool ssim_two_pass = false;        // test switch: batched candidates go through the separate z and y/x SSIM launches instead of the fused walk
    bool materialize_shifts = false;   // test switch: candidate scoring always writes the shifted copies (no on-the-fly SSIM z pass)
    std::recursive_mutex mu;      // recursive: composite entry points (mvs_pair.hip) call the public ones under the lock
    std::string last_error;
    // grow-only device scratch buffers (slot-indexed) and one pinned host staging buffer
    MvsScratch dev[16];
    void* pinned = nullptr;
    size_t pinned_cap = 0;
    void* pinned2 = nullptr;
    size_t pinned2_cap = 0;
    // an async upload out of a pinned staging buffer may still be in flight when the next call wants to refill it
    // (calls that leave their result on the device return without synchronising): one event per staging slot
    hipEvent_t pinned_ev[2] = {nullptr, nullptr};
    // tickets of mvs_event_record: a ring of events other lanes (or other devices' contexts) make their streams wait for
    hipEvent_t ticket_ev[32] = {};
    uint32_t ticket_next = 0;
    bool pinned_pending[2] = {false, false};
    // "mailbox": pinned host memory the device writes small results into directly (reduction partials, peak candidates): the host
    // reads them after the stream wait -- no copy launch, no staging through pageable memory
    int last_code = 0;            // code of the most recent mvs_fail on this context
    void* mbox_host = nullptr;
    void* mbox_dev = nullptr;
    size_t mbox_cap = 0;
    uint64_t mbox_gen = 0;        // bumped whenever the mailbox is reallocated (its old contents are gone)
    bool cb_mask_closed_form = false;  // option "cb_mask_closed_form" (default 0, see profiles/round5_cb_mask.txt): a view whose valid mask is a box gets its mask filters from tables
    bool cb_mask_count = false;        // test switch "cb_mask_count": count the views whose mask is a box (counters cb_mask_views / cb_mask_boxes)
    long long cb_mask_views = 0, cb_mask_boxes = 0;
    bool cb_exact = false;        // option "cb_exact": content-based weights through the bit-faithful passes (float64 taps, filtered masks) instead of
                                  // the default fast path (mvs_gauss_fast.inc: mask = box + list, one quantity per pass, float32 taps)
    bool cb_blend_generic = false; // test switch: the fast path takes its blending weights from the generic (float64 coordinate) blend launch instead of the closed form
    int cb_taps = 1;              // option "cb_taps_f64": accumulators of the fast path's taps: 0 float32, 1 float64 (default), 2 / 3 float64 for the first / second filter only
    int* cb_flag_host = nullptr;  // mapped pinned word the device raises when a view's mask list overflows (fast path) ...
    int* cb_flag_dev = nullptr;   // ... its device alias; read by the call itself (host results) or through counter "cb_overflow"
    long long cb_overflows = 0, cb_line_launches = 0;
    bool cb_nosplit = false;      // test switch: the y / z passes of the paired path keep both quantities in one workgroup
    bool cb_unpaired = false;     // test switch: content-based weights through the separate value / mask line passes of rounds 1-3
    int fft_slab_axes = 4;        // short axes (bit k = axis k of (z, y, x)) whose crops take the three-pass phase correlation: by default only
                                  // the contiguous one -- measured inside the 8-lane pair loop (profiles/round5_fft_slab.txt); 7 = all
    bool fft_no_slab = false;     // test switch: crops of the slab kind (mvs_fft_slab.hip) run the six single-axis passes instead
    bool fft_no_line = false;     // test switch: lengths of the whole-line DFT kernel run on the Bluestein kernels instead
    bool no_regions = false;      // test switch: skip the region kernel (use the column kernel)
    bool rows_v1 = false;         // opt-in: direct-load row kernels (mvs_fuse_rows.hip) for every dtype (default: float tiles only)
    // caching device allocator behind mvs_malloc / mvs_free: freed blocks are kept (size-keyed) and handed out
    // again, because hipMalloc / hipFree cost ~0.4 ms each and the registration path allocates per pair.
    // Work of this library is stream-ordered on `stream`, so a recycled block is never touched out of order.
    std::mutex pool_mu;
    std::multimap<size_t, void*> pool_free;
    std::unordered_map<void*, size_t> pool_live;
    size_t pool_cached_bytes = 0;
    long long pool_misses = 0, pool_releases = 0;     // counters "pool_misses" / "pool_releases": hipMalloc / hipFree calls the pool could not avoid
    double pool_miss_bytes = 0.0;
    size_t pool_cache_limit = (size_t)32 << 30;
};

// ---- per-call options of the internal entry points (arguments, not context state: a failing inner call cannot leave them behind) ----
// what mvs_register_crops knows about its two crops when it scores candidates (mvs_score_candidates_impl)
struct MvsScoreOpts {
    bool both_crops_finite = false;    // neither NaN nor inf in either crop: every voxel valid, boxes = the volume, no NaN handling
    bool argmax_only = false;          // only the arg-max candidate (and its rank correlation) is needed: the pruned search may run
    double value_bound = INFINITY;     // ... with the largest absolute value of the two (rescaled) crops
    const float* raw_u16_keys[2] = {nullptr, nullptr};   // integer-valued originals of the two crops (16-bit rank keys), or NULL
    float raw_range[4] = {0.f, 0.f, 0.f, 0.f};           // with raw_u16_keys: min, max of the fixed crop, min, max of the moving crop
};
// min / max / #valid partials the integer crop kernel leaves while it writes a crop (mvs_register_views -> mvs_resample_impl ->
// mvs_rescale_pair_device): per-block partials in the layout of nanminmax_pair_kernel, `nb` blocks per image, image k at
// base + k * nb * 16; done[k] is set by the launch that actually wrote them; gen = the mailbox generation they were parked in
// Profiling builds only (-DMVS_PROFILING_ABLATIONS): a kernel whose tag is listed in the environment variable MVS_DUP_KERNELS is
// launched twice (idempotent launches only: the second one rewrites the same results).  The difference in the pair loop's wall time
// is that kernel's marginal cost among the concurrent context lanes -- which its duration alone in a trace does not tell.
#ifdef MVS_PROFILING_ABLATIONS
bool mvs_dup_kernel(const char* tag);
#define MVS_DUP(tag, ...) do { __VA_ARGS__; if (mvs_dup_kernel(tag)) { __VA_ARGS__; } } while (0)
#else
#define MVS_DUP(tag, ...) do { __VA_ARGS__; } while (0)
#endif
struct MvsCropStats {
    char* base = nullptr;
    int nb = 0;
    uint64_t gen = 0;
    bool done[2] = {false, false};
};
struct MvsResampleOpts {
    bool defer_sync = false;           // a result in device memory is not waited for (the composite caller orders later work)
    MvsCropStats* stats = nullptr;     // with stats_k: also reduce the crop's statistics (integer crops with NaN outside only)
    int stats_k = 0;
};
int mvs_score_candidates_impl(int device, const float* fixed, const float* moving, int32_t mem, int32_t ndim, const int64_t shape[3],
                              const double* t_candidates, int32_t n_candidates, int32_t region_mode, double data_range, double im1_min,
                              int32_t quality_for_all, double* ssim_out, double* spearman_out, int32_t* code_out, const MvsScoreOpts& so);
int mvs_crop_bin_impl(int device, const mvs_view_t* view, const int32_t bin[3], const int64_t out_shape[3], float* out, const MvsResampleOpts& ro);
int mvs_resample_impl(int device, const mvs_view_t* view, const int64_t out_shape[3], int32_t order, float cval, float* out, int32_t out_mem,
                      const MvsResampleOpts& ro);

MvsContext* mvs_ctx(int device);                       // nullptr if out of range
// mvs_transfer.hip: timed tickets of the copy stream (mvs_copy_async / mvs_mark)
bool mvs_transfer_is_ticket(uint64_t ticket);
int mvs_transfer_event(MvsContext* c, uint64_t ticket, hipEvent_t* ev);
void mvs_transfer_shutdown(int dev);
int mvs_fail(MvsContext* c, int code, const char* fmt, ...);
int mvs_check_ready(int device, MvsContext** out);     // locks nothing; returns code
void* mvs_scratch(MvsContext* c, int slot, size_t nbytes);   // nullptr on failure (error set)
void* mvs_pinned(MvsContext* c, size_t nbytes);
void* mvs_pinned_slot(MvsContext* c, int slot, size_t nbytes);   // slot 0 == mvs_pinned; waits for the slot's last upload
void mvs_pinned_mark(MvsContext* c, int slot);                    // call after enqueueing the copies that read the slot
// nbytes of the context's mailbox: *host for the CPU, *dev for kernels (same memory).  Valid until the next call that asks for more;
// the caller waits for the stream before it reads.  Returns an MVS_* code.
int mvs_mailbox(MvsContext* c, size_t nbytes, void** host, void** dev);
// the context's four low-priority side streams + fork / join events (created at first use)
int mvs_ensure_aux_streams(MvsContext* c);
// stream-ordered copy of a few KB out of the mailbox (device-visible pointer) by a kernel, not by the copy engine; 16-byte granularity
int mvs_upload_from_mapped(MvsContext* c, void* dst_dev, const void* src_mapped_dev, size_t nbytes);
int mvs_upload_small(MvsContext* c, void* dst_dev, const void* src_pinned, size_t nbytes);   // parameter blocks out of a pinned staging slot: by a kernel, not a copy engine

// the code of the failure mvs_scratch / mvs_pinned recorded when they returned NULL (MVS_ERR_OUT_OF_MEMORY or MVS_ERR_HIP)
static inline int mvs_alloc_failed(const MvsContext* c) { return c->last_code ? c->last_code : MVS_ERR_HIP; }

#define MVS_HIP_TRY(c, expr)                                                         \
    do {                                                                             \
        hipError_t _e = (expr);                                                      \
        if (_e != hipSuccess) {                                                      \
            (void)hipGetLastError(); /* the runtime keeps the error for the next hipGetLastError(): it is reported here, once */ \
            return mvs_fail((c), _e == hipErrorOutOfMemory ? MVS_ERR_OUT_OF_MEMORY : MVS_ERR_HIP,  \
                            "%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__); \
        }                                                                            \
    } while (0)

static inline size_t mvs_dtype_size(int dtype) {
    switch (dtype) {
        case MVS_U8: return 1;
        case MVS_U16: return 2;
        case MVS_F32: return 4;
        default: return 0;
    }
}
