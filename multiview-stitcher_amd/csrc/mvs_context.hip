// mvs_context.hip -- device contexts, error reporting, memory helpers of libmvs_hip.so.
#include "mvs_internal.h"
#include <vector>
#include <algorithm>

#include <algorithm>
#include <cstring>

static MvsContext g_ctx[MVS_MAX_DEVICES * MVS_MAX_LANES];

MvsContext* mvs_ctx(int device) {
    if (device < 0 || (device & 0xff) >= MVS_MAX_DEVICES || (device >> 8) >= MVS_MAX_LANES) return nullptr;
    return &g_ctx[mvs_ctx_index(device)];
}

int mvs_fail(MvsContext* c, int code, const char* fmt, ...) {
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    if (c) { c->last_error = buf; c->last_code = code; }
    return code;
}

int mvs_check_ready(int device, MvsContext** out) {
    MvsContext* c = mvs_ctx(device);
    if (!c) return MVS_ERR_INVALID_ARG;
    if (!c->ready) {
        int rc = mvs_init(device);
        if (rc != MVS_OK) return rc;
    }
    *out = c;
    return MVS_OK;
}

void* mvs_scratch(MvsContext* c, int slot, size_t nbytes) {
    MvsScratch& s = c->dev[slot];
    if (nbytes <= s.cap && s.ptr) return s.ptr;
    if (s.ptr) {
        hipStreamSynchronize(c->stream);
        hipFree(s.ptr);
        s.ptr = nullptr;
        s.cap = 0;
    }
    size_t cap = nbytes + (nbytes >> 3) + 4096;
    hipError_t e = hipMalloc(&s.ptr, cap);
    if (e != hipSuccess) {
        (void)hipGetLastError();      // reported here; must not surface again at the next launch check of this thread
        mvs_fail(c, e == hipErrorOutOfMemory ? MVS_ERR_OUT_OF_MEMORY : MVS_ERR_HIP, "hipMalloc(%zu) failed: %s", cap, hipGetErrorString(e));
        s.ptr = nullptr;
        return nullptr;
    }
    s.cap = cap;
    return s.ptr;
}

void* mvs_pinned_slot(MvsContext* c, int slot, size_t nbytes) {
    void*& p = slot ? c->pinned2 : c->pinned;
    size_t& pc = slot ? c->pinned2_cap : c->pinned_cap;
    if (c->pinned_pending[slot ? 1 : 0]) {
        hipEventSynchronize(c->pinned_ev[slot ? 1 : 0]);
        c->pinned_pending[slot ? 1 : 0] = false;
    }
    if (nbytes <= pc && p) return p;
    if (p) {
        hipStreamSynchronize(c->stream);
        hipHostFree(p);
        p = nullptr;
        pc = 0;
    }
    size_t cap = nbytes * 2 + 4096;
    hipError_t e = hipHostMalloc(&p, cap, hipHostMallocDefault);
    if (e != hipSuccess) {
        (void)hipGetLastError();
        mvs_fail(c, e == hipErrorOutOfMemory ? MVS_ERR_OUT_OF_MEMORY : MVS_ERR_HIP, "hipHostMalloc(%zu) failed: %s", cap, hipGetErrorString(e));
        p = nullptr;
        return nullptr;
    }
    pc = cap;
    return p;
}

void* mvs_pinned(MvsContext* c, size_t nbytes) { return mvs_pinned_slot(c, 0, nbytes); }

int mvs_mailbox(MvsContext* c, size_t nbytes, void** host, void** dev) {
    if (nbytes > c->mbox_cap || !c->mbox_host) {
        if (c->mbox_host) {
            hipStreamSynchronize(c->stream);      // queued kernels may still write the old one
            hipHostFree(c->mbox_host);
            c->mbox_host = c->mbox_dev = nullptr;
            c->mbox_cap = 0;
        }
        const size_t cap = std::max<size_t>(nbytes * 2, (size_t)256 << 10);
        MVS_HIP_TRY(c, hipHostMalloc(&c->mbox_host, cap, hipHostMallocMapped));
        MVS_HIP_TRY(c, hipHostGetDevicePointer(&c->mbox_dev, c->mbox_host, 0));
        c->mbox_cap = cap;
        ++c->mbox_gen;
    }
    *host = c->mbox_host;
    *dev = c->mbox_dev;
    return MVS_OK;
}

// Small host -> device uploads as a KERNEL that reads the mapped pinned block (mailbox) instead of a DMA copy: a copy-engine
// transfer queues behind every large transfer in flight on that engine -- with tiles streaming in over PCIe (bench.py's
// PCIe-inclusive pipeline) each pair's 27 KB of kernel vectors waited for ALL queued 268 MB tile uploads, and the registration
// did not overlap with the uploads at all (first wave of pairs: 243 ms instead of ~11).
__global__ void mvs_small_copy_kernel(const uint4* __restrict__ src, uint4* __restrict__ dst, size_t n16) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (size_t)gridDim.x * blockDim.x) dst[i] = src[i];
}
int mvs_upload_from_mapped(MvsContext* c, void* dst_dev, const void* src_mapped_dev, size_t nbytes) {
    if ((((uintptr_t)dst_dev | (uintptr_t)src_mapped_dev) & 15) || (nbytes & 15))
        return mvs_fail(c, MVS_ERR_INVALID_ARG, "mvs_upload_from_mapped: 16-byte granularity");
    const size_t n16 = nbytes / 16;
    if (!n16) return MVS_OK;
    const unsigned blocks = (unsigned)std::min<size_t>((n16 + 255) / 256, 64);
    hipLaunchKernelGGL(mvs_small_copy_kernel, dim3(blocks), dim3(256), 0, c->stream, (const uint4*)src_mapped_dev, (uint4*)dst_dev, n16);
    MVS_HIP_TRY(c, hipGetLastError());
    return MVS_OK;
}

int mvs_ensure_aux_streams(MvsContext* c) {
    if (c->ev_fork) return MVS_OK;
    MVS_HIP_TRY(c, hipEventCreateWithFlags(&c->ev_fork, hipEventDisableTiming));
    int prio_lo = 0, prio_hi = 0;
    hipDeviceGetStreamPriorityRange(&prio_lo, &prio_hi);
    for (int a = 0; a < 4; ++a) {
        // lowest priority: the work on the main stream is the longest and ends the call, the side streams fill in around it
        // (fuse launch: 11.7 -> 11.3 ms on the jittered mosaic against equal priorities)
        MVS_HIP_TRY(c, hipStreamCreateWithPriority(&c->aux_stream[a], hipStreamNonBlocking, prio_lo));
        MVS_HIP_TRY(c, hipEventCreateWithFlags(&c->ev_join[a], hipEventDisableTiming));
    }
    return MVS_OK;
}

// Host -> device upload of a parameter block out of a pinned staging slot (mvs_pinned_slot), stream-ordered like the hipMemcpyAsync it
// replaces: up to 32 MiB travel as a KERNEL that reads the pinned block over the link (see mvs_small_copy_kernel), because a DMA copy
// queues behind whatever large transfer its copy engine is busy with -- the view records and the brick list of a slab's fuse launch
// (a few hundred KB) sat behind the previous slab's 1.3 GB download for its whole 24 ms, i.e. fuse_to_host / the block pipeline of a
// streamed fuse() overlapped kernels and transfers only when the two happened to land on different engines (round 6: in most runs
// for some slabs, in one of ten for none).  Larger blocks, odd alignments and memory the device cannot map take the DMA copy.
__global__ void mvs_copy_bytes_kernel(const unsigned char* __restrict__ src, unsigned char* __restrict__ dst, size_t nbytes) {
    const size_t n16 = nbytes >> 4, tid = (size_t)blockIdx.x * blockDim.x + threadIdx.x, nth = (size_t)gridDim.x * blockDim.x;
    for (size_t i = tid; i < n16; i += nth) reinterpret_cast<uint4*>(dst)[i] = reinterpret_cast<const uint4*>(src)[i];
    for (size_t i = (n16 << 4) + tid; i < nbytes; i += nth) dst[i] = src[i];
}
int mvs_upload_small(MvsContext* c, void* dst_dev, const void* src_pinned, size_t nbytes) {
    if (!nbytes) return MVS_OK;
    void* sdev = nullptr;
    if (nbytes <= ((size_t)32 << 20) && !(((uintptr_t)dst_dev | (uintptr_t)src_pinned) & 15) &&
        hipHostGetDevicePointer(&sdev, const_cast<void*>(src_pinned), 0) == hipSuccess && sdev) {
        const unsigned blocks = (unsigned)std::min<size_t>(((nbytes >> 4) + 255) / 256 + 1, 512);
        hipLaunchKernelGGL(mvs_copy_bytes_kernel, dim3(blocks), dim3(256), 0, c->stream, (const unsigned char*)sdev, (unsigned char*)dst_dev, nbytes);
        MVS_HIP_TRY(c, hipGetLastError());
        return MVS_OK;
    }
    (void)hipGetLastError();
    MVS_HIP_TRY(c, hipMemcpyAsync(dst_dev, src_pinned, nbytes, hipMemcpyHostToDevice, c->stream));
    return MVS_OK;
}

void mvs_pinned_mark(MvsContext* c, int slot) {
    const int k = slot ? 1 : 0;
    if (hipEventRecord(c->pinned_ev[k], c->stream) == hipSuccess) c->pinned_pending[k] = true;
    else hipStreamSynchronize(c->stream);
}

static size_t pool_round(uint64_t nbytes) {
    if (nbytes == 0) nbytes = 1;
    const size_t q = nbytes >= ((size_t)1 << 20) ? ((size_t)2 << 20) : 4096;
    return (nbytes + q - 1) / q * q;
}

// really release every cached block (stream first: a cached block may still be read by queued work)
static void pool_flush(MvsContext* c) {
    if (c->pool_free.empty()) return;
    hipStreamSynchronize(c->stream);
    for (auto& kv : c->pool_free) hipFree(kv.second);
    c->pool_free.clear();
    c->pool_cached_bytes = 0;
}

double mvs_rows_last_plan_ms(MvsContext* c);       // mvs_fuse_rows.hip
double mvs_regions_last_plan_ms(MvsContext* c);    // mvs_fuse_region.hip
double mvs_regions_class_stat(MvsContext* c, int what, int cls);   // mvs_fuse_region.hip

extern "C" {

const char* mvs_version(void) { return "mvs_hip 0.1 (gfx950)"; }

int mvs_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

int mvs_init(int device) {
    MvsContext* c = mvs_ctx(device);
    if (!c) return MVS_ERR_INVALID_ARG;
    std::lock_guard<std::recursive_mutex> lock(c->mu);
    if (c->ready) return MVS_OK;
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || mvs_hip_device(device) >= n)
        return mvs_fail(c, MVS_ERR_HIP, "no HIP device %d (count=%d, %s)", mvs_hip_device(device), n,
                        hipGetErrorString(e));
    MVS_HIP_TRY(c, hipSetDevice(mvs_hip_device(device)));
    {
        // Experiment (profiles/round5_cu_mask.txt): a lane's stream restricted to a share of the compute units, so that the pairs of
        // different lanes run next to each other instead of time-slicing the whole chip.  MVS_LANE_CU_MASK=xcd<k>: lane l gets the CUs
        // of k XCDs starting at XCD (l * k) % 8 (CU i sits on XCD i % 8); spread<k>: k / 8 of the CUs, taken evenly from all XCDs.
        // Profiling builds only (-DMVS_PROFILING_ABLATIONS), and only for values of the form xcd[1-8] / spread[1-8]: a stray
        // environment variable must not change how every lane's stream is created.
        const char* ev = nullptr;
#ifdef MVS_PROFILING_ABLATIONS
        ev = getenv("MVS_LANE_CU_MASK");
        if (ev) {
            const size_t pre = !strncmp(ev, "xcd", 3) ? 3 : (!strncmp(ev, "spread", 6) ? 6 : 0);
            if (!pre || ev[pre] < '1' || ev[pre] > '8' || ev[pre + 1] != '\0') ev = nullptr;
        }
#endif
        const int lane = (device >> 8) & 0xff;
        bool masked = false;
        if (ev && lane < 15) {
            hipDeviceProp_t prop;
            MVS_HIP_TRY(c, hipGetDeviceProperties(&prop, mvs_hip_device(device)));
            const int ncu = prop.multiProcessorCount;
            const bool xcd = !strncmp(ev, "xcd", 3);
            const int k = std::max(1, std::min(8, atoi(ev + (xcd ? 3 : 6))));
            std::vector<uint32_t> mask((size_t)(ncu + 31) / 32, 0u);
            for (int i = 0; i < ncu; ++i) {
                bool on;
                if (xcd) { const int x = i % 8, first = (lane * k) % 8; on = ((x - first + 8) % 8) < k; }
                else { const int slot = (i / 8) % 8, first = (lane * k) % 8; on = ((slot - first + 8) % 8) < k; }
                if (on) mask[(size_t)i / 32] |= 1u << (i % 32);
            }
            if (hipExtStreamCreateWithCUMask(&c->own_stream, (uint32_t)mask.size(), mask.data()) == hipSuccess) masked = true;
            else (void)hipGetLastError();
        }
        if (!masked) MVS_HIP_TRY(c, hipStreamCreateWithFlags(&c->own_stream, hipStreamNonBlocking));
    }
    MVS_HIP_TRY(c, hipEventCreate(&c->ev_start));
    MVS_HIP_TRY(c, hipEventCreate(&c->ev_stop));
    MVS_HIP_TRY(c, hipEventCreateWithFlags(&c->pinned_ev[0], hipEventDisableTiming));
    MVS_HIP_TRY(c, hipEventCreateWithFlags(&c->pinned_ev[1], hipEventDisableTiming));
    c->pinned_pending[0] = c->pinned_pending[1] = false;
    c->stream = c->own_stream;
    c->device = device;
    // extra lanes serve pair-sized work: keep their allocation caches small (lane 0 may hold whole mosaics)
    c->pool_cache_limit = ((device >> 8) ? (size_t)4 : (size_t)32) << 30;
    if (((device >> 8) & 0xff) == 0) {
        // lane 0 holds whole mosaics and the tiles of asynchronous uploads: its cache may keep half of the device's memory (144 GB of
        // an MI355X's 288; at least 32 GB).  A block freed above the limit costs a stream wait + hipFree now and a hipMalloc (10-25 ms
        // per GB, depending on the box) at its next use; one mosaic of the PCIe-inclusive pipeline alone parks 17 GB of tiles + 11 GB
        // of slabs between two runs, next to the previous mosaic.  mvs_malloc hands the cache back before it fails
        // (counters pool_misses / pool_releases).
        size_t f = 0, t = 0;
        if (hipMemGetInfo(&f, &t) == hipSuccess) c->pool_cache_limit = std::max(c->pool_cache_limit, t / 2);
        else (void)hipGetLastError();
    }
    {   // A/B switch for all context lanes of a process (bench.py, tools/): MVS_SSIM_PRUNE=0 scores every candidate in full
        const char* ev = getenv("MVS_SSIM_PRUNE");
        if (ev && *ev) c->ssim_prune = atoi(ev) != 0;
        ev = getenv("MVS_SSIM_F32");      // (the same for the float32 walk of the pruned search)
        if (ev && *ev) c->ssim_f32 = atoi(ev) != 0;
        ev = getenv("MVS_FUSE_MIXED");          // A/B switch of the one-launch fuse list (profiles/round5_fuse_mixed.txt)
        if (ev && *ev) c->fuse_mixed = atoi(ev) != 0;
        ev = getenv("MVS_FFT_SLAB_AXES");       // ... and of the crop orientations that take it (bit k = short axis k of (z, y, x))
        if (ev && *ev && atoi(ev) >= 0 && atoi(ev) <= 7) c->fft_slab_axes = atoi(ev);
        ev = getenv("MVS_FFT_NO_SLAB");         // A/B switch of the three-pass phase correlation (mvs_fft_slab.hip)
        if (ev && *ev) c->fft_no_slab = atoi(ev) != 0;
        ev = getenv("MVS_FFT_NO_PAIR");
        if (ev && *ev) c->fft_no_pair = atoi(ev) != 0;
    }
    // (The side streams of the fuse launch are NOT created here but at the first large launch (mvs_ensure_aux_streams, ~70 ms once).
    // Round 6 created them at init to take that one-off out of a one-mosaic caller's first fuse(): HIP maps streams onto its hardware
    // queues in creation order, and with the four low-priority streams made BEFORE the lanes' streams and the copy stream the
    // PCIe-inclusive pipeline ran 570-1150 ms instead of 507-514 -- the fuse launches of the slabs stalled behind the transfers
    // (alternating runs on one box, profiles/round6_summary.md).  Creation order: lanes, side streams, copy stream.)
    c->ready = true;
    c->last_error.clear();
    return MVS_OK;
}

#ifdef MVS_PROFILING_ABLATIONS
extern "C++" bool mvs_dup_kernel(const char* tag) {
    static const std::string list = [] { const char* e = getenv("MVS_DUP_KERNELS"); return std::string(",") + (e ? e : "") + ","; }();
    return list.find(std::string(",") + tag + ",") != std::string::npos;
}
#endif

void mvs_shutdown(int device) {
    MvsContext* c = mvs_ctx(device);
    if (!c || !c->ready) return;
    std::lock_guard<std::recursive_mutex> lock(c->mu);
    hipSetDevice(mvs_hip_device(device));
    hipStreamSynchronize(c->stream);
    if (((device >> 8) & 0xff) == 0) mvs_transfer_shutdown(mvs_hip_device(device));      // the device's copy stream and timed tickets go with lane 0
    {
        std::lock_guard<std::mutex> plock(c->pool_mu);
        for (auto& kv : c->pool_free) hipFree(kv.second);
        c->pool_free.clear();
        c->pool_cached_bytes = 0;
        c->pool_live.clear();     // blocks still held by the caller die with the context's device memory owner
    }
    for (auto& s : c->dev) {
        if (s.ptr) hipFree(s.ptr);
        s.ptr = nullptr;
        s.cap = 0;
    }
    if (c->mbox_host) hipHostFree(c->mbox_host);
    c->mbox_host = c->mbox_dev = nullptr;
    c->mbox_cap = 0;
    if (c->cb_flag_host) hipHostFree(c->cb_flag_host);
    c->cb_flag_host = c->cb_flag_dev = nullptr;
    if (c->pinned) hipHostFree(c->pinned);
    if (c->pinned2) hipHostFree(c->pinned2);
    c->pinned = c->pinned2 = nullptr;
    c->pinned_cap = c->pinned2_cap = 0;
    hipEventDestroy(c->ev_start);
    hipEventDestroy(c->ev_stop);
    for (int a = 0; a < 4; ++a) { if (c->aux_stream[a]) hipStreamDestroy(c->aux_stream[a]); if (c->ev_join[a]) hipEventDestroy(c->ev_join[a]); c->aux_stream[a] = nullptr; c->ev_join[a] = nullptr; }
    if (c->ev_fork) { hipEventDestroy(c->ev_fork); c->ev_fork = nullptr; }
    hipEventDestroy(c->pinned_ev[0]);
    hipEventDestroy(c->pinned_ev[1]);
    for (hipEvent_t& e : c->ticket_ev) { if (e) hipEventDestroy(e); e = nullptr; }
    c->pinned_pending[0] = c->pinned_pending[1] = false;
    hipStreamDestroy(c->own_stream);
    c->own_stream = c->stream = nullptr;
    c->ready = false;
}

const char* mvs_last_error(int device) {
    MvsContext* c = mvs_ctx(device);
    if (!c) return "invalid device index";
    return c->last_error.c_str();
}

int mvs_set_stream(int device, void* hip_stream) {
    MvsContext* c;
    int rc = mvs_check_ready(device, &c);
    if (rc) return rc;
    std::lock_guard<std::recursive_mutex> lock(c->mu);
    hipStream_t next = hip_stream ? (hipStream_t)hip_stream : c->own_stream;
    if (next != c->stream) {
        // recycled allocations and scratch are only ordered within one stream: drain the old one first
        MVS_HIP_TRY(c, hipSetDevice(mvs_hip_device(device)));
        MVS_HIP_TRY(c, hipStreamSynchronize(c->stream));
    }
    c->stream = next;
    return MVS_OK;
}

int mvs_set_option(int device, const char* key, int64_t value) {
    MvsContext* c;
    int rc = mvs_check_ready(device, &c);
    if (rc) return rc;
    if (!key) return mvs_fail(c, MVS_ERR_INVALID_ARG, "mvs_set_option: NULL key");
    std::lock_guard<std::recursive_mutex> lock(c->mu);
    if (!strcmp(key, "force_generic")) {
        c->force_generic = value != 0;
        return MVS_OK;
    }
    if (!strcmp(key, "pool_cache_limit_mb")) {
        std::lock_guard<std::mutex> plock(c->pool_mu);
        c->pool_cache_limit = (size_t)std::max<int64_t>(value, 0) << 20;
        if (c->pool_cached_bytes > c->pool_cache_limit) pool_flush(c);
        return MVS_OK;
    }
    if (!strcmp(key, "no_regions")) {
        c->no_regions = value != 0;
        return MVS_OK;
    }
    if (!strcmp(key, "rows_v1")) {
        c->rows_v1 = value != 0;
        return MVS_OK;
    }
    if (!strcmp(key, "cb_mask_count")) {
        c->cb_mask_count = value != 0;
        return MVS_OK;
    }
    if (!strcmp(key, "cb_mask_closed_form")) {
        c->cb_mask_closed_form = value != 0;
        return MVS_OK;
    }
    if (!strcmp(key, "fuse_mixed")) {
        c->fuse_mixed = value != 0;
        return MVS_OK;
    }
    if (!strcmp(key, "serial_classes")) {
        c->serial_classes = value != 0;
        return MVS_OK;
    }
    if (!strcmp(key, "reg_unfused")) {
        c->reg_unfused = value != 0;
        return MVS_OK;
    }
    if (!strcmp(key, "ssim_two_pass")) {
        c->ssim_two_pass = value != 0;
        return MVS_OK;
    }
    if (!strcmp(key, "ssim_f32")) {
        c->ssim_f32 = value != 0;
        return MVS_OK;
    }
    if (!strcmp(key, "ssim_prune")) {
        c->ssim_prune = value != 0;
        return MVS_OK;
    }
    if (!strcmp(key, "materialize_shifts")) {
        c->materialize_shifts = value != 0;
        return MVS_OK;
    }
    if (!strcmp(key, "cb_exact")) {
        c->cb_exact = value != 0;
        return MVS_OK;
    }
    if (!strcmp(key, "cb_blend_generic")) {
        c->cb_blend_generic = value != 0;
        return MVS_OK;
    }
    if (!strcmp(key, "cb_taps_f64")) {
        if (value < 0 || value > 3) return mvs_fail(c, MVS_ERR_INVALID_ARG, "mvs_set_option: cb_taps_f64 is 0, 1, 2 or 3");
        c->cb_taps = (int)value;
        return MVS_OK;
    }
    if (!strcmp(key, "cb_nosplit")) {
        c->cb_nosplit = value != 0;
        return MVS_OK;
    }
    if (!strcmp(key, "cb_unpaired")) {
        c->cb_unpaired = value != 0;
        return MVS_OK;
    }
    if (!strcmp(key, "fft_no_pair")) {
        c->fft_no_pair = value != 0;
        return MVS_OK;
    }
    if (!strcmp(key, "fft_slab_axes")) {
        if (value < 0 || value > 7) return mvs_fail(c, MVS_ERR_INVALID_ARG, "mvs_set_option: fft_slab_axes is a mask of 3 bits");
        c->fft_slab_axes = (int)value;
        return MVS_OK;
    }
    if (!strcmp(key, "fft_no_slab")) {
        c->fft_no_slab = value != 0;
        return MVS_OK;
    }
    if (!strcmp(key, "fft_no_line")) {
        c->fft_no_line = value != 0;
        return MVS_OK;
    }
    if (!strcmp(key, "ablate")) {
        c->ablate = (int)value;
        return MVS_OK;
    }
    return mvs_fail(c, MVS_ERR_INVALID_ARG, "mvs_set_option: unknown key '%s'", key);
}

// Measurement counters of one context: "reg_alg_bytes" (algorithmic HBM bytes of the pairwise registrations),
// "reg_pairs", "reg_candidates" (scored candidates), "fuse_plan_ms" (host time of the last fuse decomposition; 0 when the
// cached plan was reused).  reset != 0 clears the accumulating ones after reading.
int mvs_get_counter(int device, const char* key, int32_t reset, double* value_out) {
    MvsContext* c;
    int rc = mvs_check_ready(device, &c);
    if (rc) return rc;
    if (!key || !value_out) return mvs_fail(c, MVS_ERR_INVALID_ARG, "mvs_get_counter: NULL argument");
    std::lock_guard<std::recursive_mutex> lock(c->mu);
    if (!strcmp(key, "reg_alg_bytes")) { *value_out = c->reg_alg_bytes; if (reset) c->reg_alg_bytes = 0.0; return MVS_OK; }
    if (!strcmp(key, "reg_alg_bytes_full")) { *value_out = c->reg_alg_bytes_full; if (reset) c->reg_alg_bytes_full = 0.0; return MVS_OK; }
    if (!strcmp(key, "reg_pairs")) { *value_out = (double)c->reg_pairs; if (reset) c->reg_pairs = 0; return MVS_OK; }
    if (!strcmp(key, "reg_candidates")) { *value_out = (double)c->reg_candidates; if (reset) c->reg_candidates = 0; return MVS_OK; }
    if (!strcmp(key, "reg_rewalks")) { *value_out = (double)c->reg_rewalks; if (reset) c->reg_rewalks = 0; return MVS_OK; }
    if (!strcmp(key, "reg_pruned")) { *value_out = (double)c->reg_pruned; if (reset) c->reg_pruned = 0; return MVS_OK; }
    if (!strcmp(key, "reg_cand_volumes")) { *value_out = c->reg_cand_volumes; if (reset) c->reg_cand_volumes = 0.0; return MVS_OK; }
    if (!strcmp(key, "reg_slab_pairs")) { *value_out = (double)c->reg_slab_pairs; if (reset) c->reg_slab_pairs = 0; return MVS_OK; }
    if (!strcmp(key, "cb_mask_views")) { *value_out = (double)c->cb_mask_views; if (reset) c->cb_mask_views = 0; return MVS_OK; }
    if (!strcmp(key, "cb_mask_boxes")) { *value_out = (double)c->cb_mask_boxes; if (reset) c->cb_mask_boxes = 0; return MVS_OK; }
    if (!strcmp(key, "cb_line_launches")) { *value_out = (double)c->cb_line_launches; if (reset) c->cb_line_launches = 0; return MVS_OK; }
    if (!strcmp(key, "cb_overflow")) {
        // chunks of the fast content-based path whose mask list overflowed: the ones a host-result call redid by itself are counted,
        // a raised flag means device-result chunks that still hold a wrong result (the caller redoes them with option cb_exact)
        double pending = 0.0;
        if (c->cb_flag_host) {
            MVS_HIP_TRY(c, hipSetDevice(mvs_hip_device(device)));
            MVS_HIP_TRY(c, hipStreamSynchronize(c->stream));
            pending = *c->cb_flag_host ? 1.0 : 0.0;
            if (reset) *c->cb_flag_host = 0;
        }
        *value_out = pending;
        return MVS_OK;
    }
    // allocation pool of this context: hipMalloc calls mvs_malloc had to make (no cached block fitted) and their bytes, blocks
    // mvs_free released to the runtime because the cache was full
    if (!strcmp(key, "pool_misses")) { std::lock_guard<std::mutex> pl(c->pool_mu); *value_out = (double)c->pool_misses; if (reset) c->pool_misses = 0; return MVS_OK; }
    if (!strcmp(key, "pool_miss_bytes")) { std::lock_guard<std::mutex> pl(c->pool_mu); *value_out = c->pool_miss_bytes; if (reset) c->pool_miss_bytes = 0.0; return MVS_OK; }
    if (!strcmp(key, "pool_releases")) { std::lock_guard<std::mutex> pl(c->pool_mu); *value_out = (double)c->pool_releases; if (reset) c->pool_releases = 0; return MVS_OK; }
    if (!strcmp(key, "cb_overflows_redone")) { *value_out = (double)c->cb_overflows; if (reset) c->cb_overflows = 0; return MVS_OK; }
    if (!strcmp(key, "fuse_plan_ms")) {
        *value_out = mvs_rows_last_plan_ms(c) + mvs_regions_last_plan_ms(c);
        return MVS_OK;
    }
    // per-class figures of the last region-kernel launch of this context, <k> = 0 (one-view rim boxes), 1 (NV = 2), 2 (NV <= 4),
    // 3 (NV <= 8), 4 (copy): "fuse_class_in_vox_<k>" / "fuse_class_out_vox_<k>" = input voxel reads (sum over the class's boxes of
    // voxels x views) and output voxels of the class; "fuse_class_ms_<k>" = the class kernel's own duration, measured only by a
    // launch made with option "serial_classes" = 1 (the kernels then run one after the other between timing events; -1 otherwise)
    {
        static const char* const names[3] = {"fuse_class_in_vox_", "fuse_class_out_vox_", "fuse_class_ms_"};
        for (int w = 0; w < 3; ++w) {
            const size_t n = strlen(names[w]);
            if (!strncmp(key, names[w], n) && key[n] >= '0' && key[n] <= '4' && key[n + 1] == '\0') {
                if (w == 2) MVS_HIP_TRY(c, hipSetDevice(mvs_hip_device(device)));
                *value_out = mvs_regions_class_stat(c, w, key[n] - '0');
                return MVS_OK;
            }
        }
    }
    return mvs_fail(c, MVS_ERR_INVALID_ARG, "mvs_get_counter: unknown key '%s'", key);
}

int mvs_synchronize(int device) {
    MvsContext* c;
    int rc = mvs_check_ready(device, &c);
    if (rc) return rc;
    MVS_HIP_TRY(c, hipSetDevice(mvs_hip_device(device)));
    MVS_HIP_TRY(c, hipStreamSynchronize(c->stream));
    return MVS_OK;
}

// Stream-ordered dependencies between context lanes without a host wait.  mvs_event_record marks the work queued so far on this
// lane's stream and hands out a ticket; mvs_event_wait makes everything queued LATER on another lane's stream (same GPU) wait for
// that mark -- hipStreamWaitEvent, the host returns at once.  Tickets live in a ring of 32 events per lane: one that has been
// overwritten by a later mvs_event_record stands for that later mark (waiting longer, never shorter).
int mvs_event_record(int device, uint64_t* ticket_out) {
    MvsContext* c;
    int rc = mvs_check_ready(device, &c);
    if (rc) return rc;
    std::lock_guard<std::recursive_mutex> lock(c->mu);
    if (!ticket_out) return mvs_fail(c, MVS_ERR_INVALID_ARG, "mvs_event_record: NULL argument");
    MVS_HIP_TRY(c, hipSetDevice(mvs_hip_device(device)));
    const uint32_t slot = c->ticket_next++ % 32u;
    if (!c->ticket_ev[slot]) MVS_HIP_TRY(c, hipEventCreateWithFlags(&c->ticket_ev[slot], hipEventDisableTiming));
    MVS_HIP_TRY(c, hipEventRecord(c->ticket_ev[slot], c->stream));
    *ticket_out = ((uint64_t)(uint32_t)mvs_ctx_index(device) << 8) | slot | (1ull << 40);
    return MVS_OK;
}

int mvs_event_wait(int device, uint64_t ticket) {
    MvsContext* c;
    int rc = mvs_check_ready(device, &c);
    if (rc) return rc;
    if (mvs_transfer_is_ticket(ticket)) {      // a timed ticket of mvs_copy_async / mvs_mark (mvs_transfer.hip)
        hipEvent_t tev;
        rc = mvs_transfer_event(c, ticket, &tev);
        if (rc) return rc;
        std::lock_guard<std::recursive_mutex> lock(c->mu);
        MVS_HIP_TRY(c, hipSetDevice(mvs_hip_device(device)));
        MVS_HIP_TRY(c, hipStreamWaitEvent(c->stream, tev, 0));
        return MVS_OK;
    }
    if (!(ticket >> 40 & 1ull)) return mvs_fail(c, MVS_ERR_INVALID_ARG, "mvs_event_wait: not a ticket");
    const uint32_t slot = (uint32_t)(ticket & 0xffu), idx = (uint32_t)((ticket >> 8) & 0xffffffu);
    if (slot >= 32u || idx >= (uint32_t)(MVS_MAX_DEVICES * MVS_MAX_LANES)) return mvs_fail(c, MVS_ERR_INVALID_ARG, "mvs_event_wait: bad ticket");
    MvsContext* src = &g_ctx[idx];
    hipEvent_t ev;
    {
        std::lock_guard<std::recursive_mutex> lock(src->mu);
        if (!src->ready || !src->ticket_ev[slot]) return mvs_fail(c, MVS_ERR_INVALID_ARG, "mvs_event_wait: the recording context is gone");
        ev = src->ticket_ev[slot];
    }
    std::lock_guard<std::recursive_mutex> lock(c->mu);
    MVS_HIP_TRY(c, hipSetDevice(mvs_hip_device(device)));
    MVS_HIP_TRY(c, hipStreamWaitEvent(c->stream, ev, 0));
    return MVS_OK;
}

double mvs_last_kernel_ms(int device) {
    MvsContext* c = mvs_ctx(device);
    if (!c || !c->ready || !c->timing_valid) return -1.0;
    std::lock_guard<std::recursive_mutex> lock(c->mu);
    hipSetDevice(mvs_hip_device(device));
    if (hipEventSynchronize(c->ev_stop) != hipSuccess) return -1.0;
    float ms = -1.f;
    if (hipEventElapsedTime(&ms, c->ev_start, c->ev_stop) != hipSuccess) return -1.0;
    return (double)ms;
}

int mvs_malloc(int device, uint64_t nbytes, void** dev_ptr) {
    MvsContext* c;
    int rc = mvs_check_ready(device, &c);
    if (rc) return rc;
    if (!dev_ptr) return mvs_fail(c, MVS_ERR_INVALID_ARG, "mvs_malloc: dev_ptr is NULL");
    MVS_HIP_TRY(c, hipSetDevice(mvs_hip_device(device)));
    const size_t want = pool_round(nbytes);
    std::lock_guard<std::mutex> lock(c->pool_mu);
    auto it = c->pool_free.lower_bound(want);
    if (it != c->pool_free.end() && it->first <= want + want / 4 + ((size_t)1 << 20)) {
        *dev_ptr = it->second;
        c->pool_live[it->second] = it->first;
        c->pool_cached_bytes -= it->first;
        c->pool_free.erase(it);
        return MVS_OK;
    }
    c->pool_misses += 1;
    c->pool_miss_bytes += (double)want;
    hipError_t e = hipMalloc(dev_ptr, want);
    if (e != hipSuccess) {           // out of memory: give the cache back and retry once
        (void)hipGetLastError();
        pool_flush(c);
        MVS_HIP_TRY(c, hipMalloc(dev_ptr, want));
    }
    c->pool_live[*dev_ptr] = want;
    return MVS_OK;
}

int mvs_mem_info(int device, uint64_t* free_bytes, uint64_t* total_bytes) {
    MvsContext* c;
    int rc = mvs_check_ready(device, &c);
    if (rc) return rc;
    MVS_HIP_TRY(c, hipSetDevice(mvs_hip_device(device)));
    size_t f = 0, t = 0;
    MVS_HIP_TRY(c, hipMemGetInfo(&f, &t));
    {   // memory parked in the allocation cache is handed back on demand (mvs_malloc flushes it before it fails)
        std::lock_guard<std::mutex> lock(c->pool_mu);
        f += c->pool_cached_bytes;
    }
    if (free_bytes) *free_bytes = (uint64_t)f;
    if (total_bytes) *total_bytes = (uint64_t)t;
    return MVS_OK;
}

int mvs_free(int device, void* dev_ptr) {
    MvsContext* c;
    int rc = mvs_check_ready(device, &c);
    if (rc) return rc;
    if (!dev_ptr) return MVS_OK;
    MVS_HIP_TRY(c, hipSetDevice(mvs_hip_device(device)));
    std::lock_guard<std::mutex> lock(c->pool_mu);
    auto it = c->pool_live.find(dev_ptr);
    if (it == c->pool_live.end()) {   // not ours (or freed twice): behave like hipFree
        MVS_HIP_TRY(c, hipStreamSynchronize(c->stream));
        MVS_HIP_TRY(c, hipFree(dev_ptr));
        return MVS_OK;
    }
    const size_t sz = it->second;
    c->pool_live.erase(it);
    if (c->pool_cached_bytes + sz > c->pool_cache_limit) {
        c->pool_releases += 1;
        MVS_HIP_TRY(c, hipStreamSynchronize(c->stream));
        MVS_HIP_TRY(c, hipFree(dev_ptr));
        return MVS_OK;
    }
    c->pool_free.emplace(sz, dev_ptr);
    c->pool_cached_bytes += sz;
    return MVS_OK;
}

int mvs_memcpy_h2d(int device, void* dst_dev, const void* src_host, uint64_t nbytes) {
    MvsContext* c;
    int rc = mvs_check_ready(device, &c);
    if (rc) return rc;
    MVS_HIP_TRY(c, hipSetDevice(mvs_hip_device(device)));
    MVS_HIP_TRY(c, hipMemcpyAsync(dst_dev, src_host, nbytes, hipMemcpyHostToDevice, c->stream));
    MVS_HIP_TRY(c, hipStreamSynchronize(c->stream));
    return MVS_OK;
}

int mvs_memcpy_d2h(int device, void* dst_host, const void* src_dev, uint64_t nbytes) {
    MvsContext* c;
    int rc = mvs_check_ready(device, &c);
    if (rc) return rc;
    MVS_HIP_TRY(c, hipSetDevice(mvs_hip_device(device)));
    MVS_HIP_TRY(c, hipMemcpyAsync(dst_host, src_dev, nbytes, hipMemcpyDeviceToHost, c->stream));
    MVS_HIP_TRY(c, hipStreamSynchronize(c->stream));
    return MVS_OK;
}

// Device-to-device copy between two GPUs of the node (xGMI peer access when the driver grants it, staged by the runtime
// otherwise).  Ordered after everything queued on the SOURCE context's stream, issued on the destination's stream, returns
// when the bytes have arrived: the one-tile halo a chunk or pair owner fetches from a neighbour's resident tiles.
int mvs_memcpy_peer(int dst_device, void* dst_dev, int src_device, const void* src_dev, uint64_t nbytes) {
    MvsContext *cd, *cs;
    int rc = mvs_check_ready(dst_device, &cd);
    if (rc) return rc;
    rc = mvs_check_ready(src_device, &cs);
    if (rc) return rc;
    if (!dst_dev || !src_dev) return mvs_fail(cd, MVS_ERR_INVALID_ARG, "mvs_memcpy_peer: NULL pointer");
    const int dd = mvs_hip_device(dst_device), sd = mvs_hip_device(src_device);
    MVS_HIP_TRY(cs, hipSetDevice(sd));
    MVS_HIP_TRY(cs, hipStreamSynchronize(cs->stream));
    MVS_HIP_TRY(cd, hipSetDevice(dd));
    if (dd == sd) {
        MVS_HIP_TRY(cd, hipMemcpyAsync(dst_dev, src_dev, nbytes, hipMemcpyDeviceToDevice, cd->stream));
    } else {
        static std::mutex peer_mu;
        static bool peer_tried[MVS_MAX_DEVICES][MVS_MAX_DEVICES] = {{false}};
        {
            std::lock_guard<std::mutex> lk(peer_mu);
            if (!peer_tried[dd][sd]) {     // best effort: without peer access the runtime stages the copy
                int can = 0;
                if (hipDeviceCanAccessPeer(&can, dd, sd) == hipSuccess && can) {
                    hipError_t e = hipDeviceEnablePeerAccess(sd, 0);
                    if (e != hipSuccess && e != hipErrorPeerAccessAlreadyEnabled) (void)hipGetLastError();
                }
                peer_tried[dd][sd] = true;
            }
        }
        MVS_HIP_TRY(cd, hipMemcpyPeerAsync(dst_dev, dd, src_dev, sd, nbytes, cd->stream));
    }
    MVS_HIP_TRY(cd, hipStreamSynchronize(cd->stream));
    return MVS_OK;
}

int mvs_upload_tile(int device, const void* host, int32_t dtype, const int64_t shape[3],
                    void** dev_ptr) {
    MvsContext* c;
    int rc = mvs_check_ready(device, &c);
    if (rc) return rc;
    size_t es = mvs_dtype_size(dtype);
    if (!es || !host || !dev_ptr || !shape)
        return mvs_fail(c, MVS_ERR_INVALID_ARG, "mvs_upload_tile: bad argument");
    uint64_t n = (uint64_t)shape[0] * (uint64_t)shape[1] * (uint64_t)shape[2] * es;
    rc = mvs_malloc(device, n, dev_ptr);
    if (rc) return rc;
    return mvs_memcpy_h2d(device, *dev_ptr, host, n);
}

// rows of `nbytes_row` bytes: src rows sit at (z * src_pitch_z + y * src_pitch_y), dst rows at (z * dst_pitch_z + y * dst_pitch_y).
// A thread moves 16-byte pieces (work items = (row, piece), so short rows fill the launch as well as long ones); neither end
// needs any alignment (rows of a uint16 mosaic of odd width start at odd 2-byte offsets): global memory takes unaligned vectors.
__global__ __launch_bounds__(256) void copy_box_kernel(const unsigned char* __restrict__ src, unsigned char* __restrict__ dst, int nz, int ny,
                                                       long long nbytes_row, long long src_pitch_y, long long src_pitch_z, long long dst_pitch_y,
                                                       long long dst_pitch_z) {
    typedef unsigned int u32x4_a1 __attribute__((ext_vector_type(4), aligned(1)));
    const long long rows = (long long)nz * ny, ppr = (nbytes_row + 15) >> 4, total = rows * ppr;
    for (long long w = (long long)blockIdx.x * blockDim.x + threadIdx.x; w < total; w += (long long)gridDim.x * blockDim.x) {
        const long long r = w / ppr, pc = w - r * ppr;
        const int z = (int)(r / ny), y = (int)(r - (long long)z * ny);
        const unsigned char* s = src + (long long)z * src_pitch_z + (long long)y * src_pitch_y + pc * 16;
        unsigned char* d = dst + (long long)z * dst_pitch_z + (long long)y * dst_pitch_y + pc * 16;
        const long long left = nbytes_row - pc * 16;
        if (left >= 16) *reinterpret_cast<u32x4_a1*>(d) = *reinterpret_cast<const u32x4_a1*>(s);
        else
            for (int i = 0; i < (int)left; ++i) d[i] = s[i];
    }
}

int mvs_memset(int device, void* dst_dev, int32_t byte_value, uint64_t nbytes) {
    MvsContext* c;
    int rc = mvs_check_ready(device, &c);
    if (rc) return rc;
    if (!dst_dev && nbytes) return mvs_fail(c, MVS_ERR_INVALID_ARG, "mvs_memset: NULL pointer");
    std::lock_guard<std::recursive_mutex> lock(c->mu);
    MVS_HIP_TRY(c, hipSetDevice(mvs_hip_device(device)));
    if (nbytes) MVS_HIP_TRY(c, hipMemsetAsync(dst_dev, byte_value, nbytes, c->stream));
    return MVS_OK;
}

int mvs_copy_into(int device, const void* src_dev, int32_t dtype, const int64_t shape[3], void* dst_dev, const int64_t dst_shape[3],
                  const int64_t dst_offset[3]) {
    MvsContext* c;
    int rc = mvs_check_ready(device, &c);
    if (rc) return rc;
    const size_t es = mvs_dtype_size(dtype);
    if (!es || !src_dev || !dst_dev || !shape || !dst_shape || !dst_offset)
        return mvs_fail(c, MVS_ERR_INVALID_ARG, "mvs_copy_into: bad argument");
    for (int k = 0; k < 3; ++k)
        if (shape[k] < 0 || dst_offset[k] < 0 || dst_offset[k] + shape[k] > dst_shape[k])
            return mvs_fail(c, MVS_ERR_INVALID_ARG, "mvs_copy_into: box does not fit into the destination");
    if (shape[0] * shape[1] * shape[2] == 0) return MVS_OK;
    if (shape[0] * shape[1] >= (1ll << 31)) return mvs_fail(c, MVS_ERR_UNSUPPORTED, "mvs_copy_into: too many rows");
    std::lock_guard<std::recursive_mutex> lock(c->mu);
    MVS_HIP_TRY(c, hipSetDevice(mvs_hip_device(device)));
    const long long py = dst_shape[2] * (long long)es, pz = dst_shape[1] * py;
    unsigned char* d0 = (unsigned char*)dst_dev + dst_offset[0] * pz + dst_offset[1] * py + dst_offset[2] * (long long)es;
    const int rows = (int)(shape[0] * shape[1]);
    const long long pieces = (long long)rows * ((shape[2] * (long long)es + 15) >> 4);
    const long long row = (long long)shape[2] * (long long)es;
    hipLaunchKernelGGL(copy_box_kernel, dim3((unsigned)std::min<long long>((pieces + 255) / 256, 65536)), dim3(256), 0, c->stream, (const unsigned char*)src_dev, d0, (int)shape[0],
                       (int)shape[1], row, row, row * shape[1], py, pz);
    MVS_HIP_TRY(c, hipGetLastError());
    return MVS_OK;
}

int mvs_copy_box(int device, const void* src_dev, const int64_t src_pitch[2], void* dst_dev, const int64_t dst_pitch[2], const int64_t box[3]) {
    MvsContext* c;
    int rc = mvs_check_ready(device, &c);
    if (rc) return rc;
    if (!src_dev || !dst_dev || !src_pitch || !dst_pitch || !box) return mvs_fail(c, MVS_ERR_INVALID_ARG, "mvs_copy_box: NULL argument");
    if (box[0] < 0 || box[1] < 0 || box[2] < 0 || src_pitch[0] < box[2] || dst_pitch[0] < box[2] || src_pitch[1] < 0 || dst_pitch[1] < 0)
        return mvs_fail(c, MVS_ERR_INVALID_ARG, "mvs_copy_box: negative extent or a row pitch shorter than the row");
    if (box[0] * box[1] * box[2] == 0) return MVS_OK;
    if (box[0] * box[1] >= (1ll << 31)) return mvs_fail(c, MVS_ERR_UNSUPPORTED, "mvs_copy_box: too many rows");
    std::lock_guard<std::recursive_mutex> lock(c->mu);
    MVS_HIP_TRY(c, hipSetDevice(mvs_hip_device(device)));
    const long long pieces = box[0] * box[1] * ((box[2] + 15) >> 4);
    hipLaunchKernelGGL(copy_box_kernel, dim3((unsigned)std::min<long long>((pieces + 255) / 256, 65536)), dim3(256), 0, c->stream, (const unsigned char*)src_dev,
                       (unsigned char*)dst_dev, (int)box[0], (int)box[1], (long long)box[2], (long long)src_pitch[0], (long long)src_pitch[1], (long long)dst_pitch[0],
                       (long long)dst_pitch[1]);
    MVS_HIP_TRY(c, hipGetLastError());
    return MVS_OK;
}

}  // extern "C"
