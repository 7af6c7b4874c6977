// mvs_transfer.hip -- host <-> device transfers that overlap with the kernels (gfx950 only).
//
// The reference's users hand fuse() host / Zarr-backed arrays and stream the fused chunks out (fusion/_core.py:1068-1170, 2044-2156;
// spatial_image_utils.py:712-860 reads tiles lazily); SURVEY 8d(2) defines the end-to-end figure INCLUDING H2D / D2H and asks for
// pinned asynchronous copies overlapped with the work.  Rounds 2-5 had that pipeline only inside bench.py (torch streams); this
// unit puts its pieces behind the C ABI:
//   * mvs_host_alloc / mvs_host_free: pinned host memory (the only kind an asynchronous copy really is asynchronous from);
//   * mvs_copy_async: one copy on the device's COPY STREAM of its direction (uploads and downloads have one each) -- created with a priority of its own, because HIP multiplexes the
//     streams of a process onto a few hardware queues PER PRIORITY LEVEL and a stream holding the barrier packets of 64 queued tile
//     uploads stalls every compute stream that shares its queue (measured in round 4: the first wave of pairs took 242 ms instead
//     of ~11 with a normal-priority copy stream) -- optionally after a ticket, returning a ticket of its own;
//   * tickets of this unit are TIMED events from a ring of 4096 per device: mvs_event_wait (mvs_context.hip) accepts them next to
//     the 32 per-lane tickets of mvs_event_record, so a pair job of mvs_register_pairs (wait_ticket) starts when its two tiles have
//     landed and a slab's download starts when its fuse launch is done; mvs_mark puts one on a lane's stream; mvs_ticket_sync
//     makes the host wait; mvs_ticket_elapsed_ms reads the time between two of them (the overlap test reads its timeline there).
#include "mvs_internal.h"

#include <cstdlib>
#include <mutex>

namespace {

constexpr uint32_t kRing = 4096;
struct TransferQueue {
    std::mutex mu;
    hipStream_t stream[2] = {nullptr, nullptr};      // [0] host -> device, [1] device -> host: one copy stream per DIRECTION, so that the
                                                     // uploads of block k + 1 and the download of block k - 1 of a streamed fuse() run side by
                                                     // side (the link is full duplex: 96 GB/s both ways at once against 55 one way); each
                                                     // direction keeps the order in which its copies were queued.  (Measured on the C5
                                                     // z-slab, whose pipeline is bound by the host's chunk-file copies: 0.27-0.33 s with one
                                                     // stream or two.)
    hipEvent_t ev[kRing] = {};
    uint32_t next = 0;
};
TransferQueue g_tq[MVS_MAX_DEVICES];

constexpr uint64_t kTransferTag = 1ull << 41;

int ensure_stream(MvsContext* c, TransferQueue& q, int dev) {
    if (q.stream[0]) return MVS_OK;
    int lo = 0, hi = 0;
    MVS_HIP_TRY(c, hipDeviceGetStreamPriorityRange(&lo, &hi));      // (lo: numerically greatest = least urgent)
    for (int k = 0; k < 2; ++k) MVS_HIP_TRY(c, hipStreamCreateWithPriority(&q.stream[k], hipStreamNonBlocking, hi));
    (void)dev;
    return MVS_OK;
}

// a fresh timed event of the device's ring (the ticket of the slot's previous use becomes the newer mark: waits get longer, never shorter)
int next_event(MvsContext* c, TransferQueue& q, int dev, hipEvent_t* ev, uint64_t* ticket) {
    const uint32_t slot = q.next++ % kRing;
    if (!q.ev[slot]) MVS_HIP_TRY(c, hipEventCreate(&q.ev[slot]));
    *ev = q.ev[slot];
    *ticket = kTransferTag | ((uint64_t)(uint32_t)dev << 32) | slot;
    return MVS_OK;
}

}  // namespace

// (mvs_context.hip: mvs_event_wait hands tickets of this unit over)
bool mvs_transfer_is_ticket(uint64_t ticket) { return (ticket & kTransferTag) != 0; }
int mvs_transfer_event(MvsContext* c, uint64_t ticket, hipEvent_t* ev) {
    const uint32_t dev = (uint32_t)((ticket >> 32) & 0xffu), slot = (uint32_t)(ticket & 0xffffffffu);
    if (!(ticket & kTransferTag) || dev >= MVS_MAX_DEVICES || slot >= kRing) return mvs_fail(c, MVS_ERR_INVALID_ARG, "not a transfer ticket");
    std::lock_guard<std::mutex> lock(g_tq[dev].mu);
    if (!g_tq[dev].ev[slot]) return mvs_fail(c, MVS_ERR_INVALID_ARG, "transfer ticket of a slot that was never used");
    *ev = g_tq[dev].ev[slot];
    return MVS_OK;
}

extern "C" int mvs_host_alloc(uint64_t nbytes, void** host_ptr) {
    if (!host_ptr) return MVS_ERR_INVALID_ARG;
    *host_ptr = nullptr;
    const hipError_t e = hipHostMalloc(host_ptr, nbytes ? (size_t)nbytes : 1, hipHostMallocDefault);
    if (e != hipSuccess) { (void)hipGetLastError(); return e == hipErrorOutOfMemory ? MVS_ERR_OUT_OF_MEMORY : MVS_ERR_HIP; }
    return MVS_OK;
}

extern "C" int mvs_host_free(void* host_ptr) {
    if (!host_ptr) return MVS_OK;
    const hipError_t e = hipHostFree(host_ptr);
    if (e != hipSuccess) { (void)hipGetLastError(); return MVS_ERR_HIP; }
    return MVS_OK;
}

extern "C" int mvs_copy_async(int device, void* dst, const void* src, uint64_t nbytes, int32_t kind, uint64_t after, uint64_t* done_out) {
    MvsContext* c;
    int rc = mvs_check_ready(device, &c);
    if (rc) return rc;
    if (!dst || !src || (kind != 0 && kind != 1)) return mvs_fail(c, MVS_ERR_INVALID_ARG, "mvs_copy_async: dst, src, kind 0 (host -> device) | 1 (device -> host)");
    const int dev = mvs_hip_device(device);
    TransferQueue& q = g_tq[dev];
    hipEvent_t after_ev = nullptr;
    if (after) {
        if (mvs_transfer_is_ticket(after)) {
            rc = mvs_transfer_event(c, after, &after_ev);
            if (rc) return rc;
        } else {
            // a lane ticket (mvs_event_record): let the lane of THIS call's context id wait for it and mark that -- keeps the 32-slot
            // ring's bookkeeping in one place (mvs_context.hip)
            rc = mvs_event_wait(device, after);
            if (rc) return rc;
            uint64_t m = 0;
            rc = mvs_mark(device, &m);
            if (rc) return rc;
            rc = mvs_transfer_event(c, m, &after_ev);
            if (rc) return rc;
        }
    }
    std::lock_guard<std::mutex> lock(q.mu);
    MVS_HIP_TRY(c, hipSetDevice(dev));
    rc = ensure_stream(c, q, dev);
    if (rc) return rc;
    hipStream_t cs = q.stream[kind];
    if (after_ev) MVS_HIP_TRY(c, hipStreamWaitEvent(cs, after_ev, 0));
    if (nbytes) MVS_HIP_TRY(c, hipMemcpyAsync(dst, src, (size_t)nbytes, kind == 0 ? hipMemcpyHostToDevice : hipMemcpyDeviceToHost, cs));
    hipEvent_t ev;
    uint64_t ticket;
    rc = next_event(c, q, dev, &ev, &ticket);
    if (rc) return rc;
    MVS_HIP_TRY(c, hipEventRecord(ev, cs));
    if (done_out) *done_out = ticket;
    return MVS_OK;
}

extern "C" int mvs_mark(int device, uint64_t* ticket_out) {
    MvsContext* c;
    int rc = mvs_check_ready(device, &c);
    if (rc) return rc;
    if (!ticket_out) return mvs_fail(c, MVS_ERR_INVALID_ARG, "mvs_mark: NULL argument");
    const int dev = mvs_hip_device(device);
    TransferQueue& q = g_tq[dev];
    hipEvent_t ev;
    {
        std::lock_guard<std::mutex> lock(q.mu);
        MVS_HIP_TRY(c, hipSetDevice(dev));
        rc = next_event(c, q, dev, &ev, ticket_out);
        if (rc) return rc;
    }
    std::lock_guard<std::recursive_mutex> lock(c->mu);
    MVS_HIP_TRY(c, hipEventRecord(ev, c->stream));
    return MVS_OK;
}

extern "C" int mvs_ticket_sync(uint64_t ticket) {
    MvsContext* c = mvs_ctx((int)((ticket >> 32) & 0xffu));
    if (!c) return MVS_ERR_INVALID_ARG;
    hipEvent_t ev;
    int rc = mvs_transfer_event(c, ticket, &ev);
    if (rc) return rc;
    MVS_HIP_TRY(c, hipEventSynchronize(ev));
    return MVS_OK;
}

extern "C" int mvs_ticket_elapsed_ms(uint64_t t0, uint64_t t1, double* ms_out) {
    MvsContext* c = mvs_ctx((int)((t0 >> 32) & 0xffu));
    if (!c) return MVS_ERR_INVALID_ARG;
    if (!ms_out) return mvs_fail(c, MVS_ERR_INVALID_ARG, "mvs_ticket_elapsed_ms: NULL argument");
    hipEvent_t a, b;
    int rc = mvs_transfer_event(c, t0, &a);
    if (rc) return rc;
    rc = mvs_transfer_event(c, t1, &b);
    if (rc) return rc;
    MVS_HIP_TRY(c, hipEventSynchronize(a));
    MVS_HIP_TRY(c, hipEventSynchronize(b));
    float ms = 0.f;
    MVS_HIP_TRY(c, hipEventElapsedTime(&ms, a, b));
    *ms_out = (double)ms;
    return MVS_OK;
}

// mvs_shutdown of the device's lane 0 releases the queue
void mvs_transfer_shutdown(int dev) {
    if (dev < 0 || dev >= MVS_MAX_DEVICES) return;
    TransferQueue& q = g_tq[dev];
    std::lock_guard<std::mutex> lock(q.mu);
    for (hipStream_t& st : q.stream)
        if (st) { hipStreamSynchronize(st); hipStreamDestroy(st); st = nullptr; }
    for (auto& e : q.ev)
        if (e) { hipEventDestroy(e); e = nullptr; }
    q.next = 0;
}
