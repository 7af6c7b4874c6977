/* mvs_hip.h -- C ABI of libmvs_hip.so: the MI355X (gfx950) register+fuse hot path
 * for multiview-stitcher workflows.
 *
 * Every entry point below replaces (sits directly beneath) one Python call
 * site of the reference; citations are relative to the reference checkout
 * (src/multiview_stitcher/...).  Nothing in these signatures is a torch type:
 * plain pointers, sizes and POD structs, so the library can be bound from
 * ctypes (what this repo ships), cffi, pybind11 or cgo alike.
 *
 * Conventions
 *   - Arrays are C-contiguous along x; axis order is z,y,x.  2D data is passed
 *     as 3D with shape[0] == 1 (the reference drops a singleton z the same way,
 *     registration.py:2414-2464, fusion/_core.py:701-704).
 *   - "matrix"/"offset" follow scipy.ndimage.affine_transform: they map OUTPUT
 *     pixel indices to INPUT pixel coordinates, in_coord = matrix @ out_idx + offset,
 *     and are produced on the host exactly as transformation.py:37-83 does
 *     (double precision, rounded to 10 decimals, near-integer offsets snapped).
 *   - Return value: 0 = ok, negative = error; the message is available from
 *     mvs_last_error(device).  The Python shim raises RuntimeError.
 *   - Threading: one context per device; calls on the same device serialise on
 *     an internal mutex and run on that context's HIP stream; calls on
 *     different devices are fully concurrent.
 *   - Context lanes: a device argument may carry a lane number in bits 8..11
 *     (`device | lane << 8`, lane < 16): every lane is an independent context on the
 *     same GPU (own stream, scratch buffers, allocation pool and lock), so host
 *     threads that work on independent units (image pairs) overlap on the device
 *     instead of serialising on one lock.  Memory allocated through one lane can be
 *     read by the others once the producing call has returned.
 *   - Ownership: the caller owns every pointer it passes for the duration of
 *     the call.  Device allocations made by mvs_malloc / mvs_upload_tile belong
 *     to the library until mvs_free.
 */
#ifndef MVS_HIP_H
#define MVS_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MVS_OK 0
#define MVS_ERR_INVALID_ARG (-1)
#define MVS_ERR_HIP (-2)
#define MVS_ERR_NOT_INITIALISED (-3)
#define MVS_ERR_UNSUPPORTED (-4)
/* a HIP call failed with hipErrorOutOfMemory (device or pinned host allocation): the caller may retry with smaller units */
#define MVS_ERR_OUT_OF_MEMORY (-5)

enum mvs_dtype { MVS_U8 = 0, MVS_U16 = 1, MVS_F32 = 2 };
enum mvs_mem { MVS_MEM_HOST = 0, MVS_MEM_DEVICE = 1 };
/* fusion_func: fusion/_core.py:61-94 (weighted_average_fusion, the default),
 * :42-58 (max_fusion), :97-131 (simple_average_fusion) */
enum mvs_fusion { MVS_FUSE_WEIGHTED_AVERAGE = 0, MVS_FUSE_MAX = 1, MVS_FUSE_SIMPLE_AVERAGE = 2 };
/* weights_func: None | weights.content_based (weights.py:22-74) */
enum mvs_weights { MVS_WEIGHTS_NONE = 0, MVS_WEIGHTS_CONTENT_BASED = 1 };

/* One input view slab of one output chunk: what fuse_np receives per view as
 * (sims[i], params[i], full_view_bbs[i])  -- fusion/_core.py:1513-1531, 1621-1646. */
typedef struct mvs_view_t {
    const void* data;      /* slab voxels, host or device pointer (see mem)            */
    int32_t dtype;         /* enum mvs_dtype; all views of one call share one dtype    */
    int32_t mem;           /* enum mvs_mem                                            */
    int64_t shape[3];      /* slab shape z,y,x                                        */
    int64_t stride[3];     /* element strides z,y,x; stride[2] must be 1             */
    double matrix[9];      /* out px -> slab px  (transformation.py:53-83)           */
    double offset[3];
    double w_matrix[9];    /* out px -> coordinates on the 5^ndim blending support grid */
    double w_offset[3];    /*   (weights.py:448-481 through transformation.py:53-83)  */
    float edt[125];        /* support table, (nz,5,5) row-major, nz = 5 (3D) or 1 (2D);
                              distance_transform_edt of weights.py:459-464 cast to f32 */
    int32_t reserved;
    int64_t index_offset[3]; /* index frame (see mvs_fuse_opts_t.index_origin): first pixel of this slab in the pixel grid
                              that `offset` refers to; 0 = `offset` refers to the slab itself (the reference's per-chunk form) */
} mvs_view_t;

typedef struct mvs_fuse_opts_t {
    int32_t ndim;          /* 2 or 3                                                  */
    int32_t order;         /* interpolation_order 0 | 1  (_core.py:1521, 1627)        */
    int32_t fusion;        /* enum mvs_fusion                                         */
    int32_t weights;       /* enum mvs_weights                                        */
    int64_t out_shape[3];  /* chunk shape INCLUDING halo (output_properties["shape"]) */
    int64_t trim[3];       /* trim_overlap_in_pixels per axis (_core.py:1687-1711)    */
    float sigma_1;         /* content_based sigmas (weights.py:26-27)                 */
    float sigma_2;
    int32_t out_dtype;     /* dtype of the result = input dtype (_core.py:1713)       */
    int32_t out_mem;       /* where `out` lives                                       */
    int64_t index_origin[3]; /* INDEX FRAME.  The reference derives matrix / offset / w_offset per chunk from the chunk's and
                              the slab's origins and rounds them to 10 decimals (transformation.py:72-83), so two chunkings of
                              one mosaic differ by ~1e-9 px in the blend weights -- one count on ~1e-5 of the voxels.  A caller
                              that fuses a stack chunk by chunk (fusion.fuse, the multi-GPU shards) may instead derive every
                              view's parameters ONCE, for output index 0 = a common frame origin (the stack's first voxel) and
                              for pixel 0 of the WHOLE view, and pass here the index of this chunk's first voxel (incl. halo)
                              in that frame and in mvs_view_t.index_offset the first pixel of each slab.  The integer parts
                              are then shifted as integers: every voxel gets the same parameters whatever chunk it falls in.
                              All zeros (the default) = the reference's per-chunk form. */
} mvs_fuse_opts_t;

/* ---- context ---------------------------------------------------------------- */
const char* mvs_version(void);
int mvs_device_count(void);
/* Creates the per-device context (stream, events, scratch pool). Idempotent. */
int mvs_init(int device);
void mvs_shutdown(int device);
const char* mvs_last_error(int device);
/* Run this device's work on an externally owned hipStream_t (e.g. torch's
 * current stream) instead of the context's own stream; NULL restores it. */
int mvs_set_stream(int device, void* hip_stream);
int mvs_synchronize(int device);
/* Tuning / test switches. "force_generic" = 1: mvs_fuse_chunk never takes the translation fast
 * path (both paths must agree; tests compare them).  "no_regions" = 1: skip the region kernels.
 * "pool_cache_limit_mb": bytes (MiB) mvs_free may keep cached for later mvs_malloc calls
 * (default 32768; 0 = release immediately).  "materialize_shifts" = 1: mvs_score_candidates / mvs_register_crops
 * always write the shifted copies of the moving image (by default finite-only crops evaluate them inside the SSIM z
 * pass; both ways must agree bit for bit) and mvs_phasecorr_multi runs one inverse transform per normalisation (by default
 * two normalisations share one); tests compare the plain and the default paths.  "serial_classes" = 1: the class kernels of
 * the translation fast path of mvs_fuse_chunk run one after the other on the context's stream (default: side by side on
 * side streams, joined before the call's work is considered done).  "fuse_mixed" = 1 (default 0): the copy class and the
 * one- / two-view classes of that path run as ONE launch over a brick list ordered in space across the classes (measured: no
 * faster, 2 % less HBM traffic -- profiles/round5_fuse_mixed.txt; same voxels).  "rows_v1" = 1: the direct-load row-owning kernels
 * (whole output rows per workgroup, mvs_fuse_rows.hip) are tried before the region kernels for every dtype (default: for
 * float32 tiles only, where they reproduce scipy's NaN propagation through zero-weight taps); it must agree with the
 * default path (tests compare both with the oracle).  Unknown keys -- among them the retired "rowlds" and "stream_rows"
 * of rounds 1-2 -- return MVS_ERR_INVALID_ARG.  "ssim_two_pass" = 1: batched SSIM candidates (and the fixed
 * image's window means) go through the separate z and y / x launches instead of the fused z walks (equal to 1e-9).
 * "reg_unfused" = 1: the phase correlation runs its separate launches (pack, cross power, stored correlation + peak search,
 * one refinement stage per normalisation, a min / max pass over the crops) instead of the fused passes at the ends of the two
 * transforms (bit for bit the same peaks and shifts; tests compare).  "fft_no_line" = 1: axes of 17-64 samples with prime factors
 * <= 19 run on the Bluestein kernels instead of the whole-line register transforms (mvs_dft_small.h; equal to float32 rounding).
 * "fft_no_pair" = 1 (environment MVS_FFT_NO_PAIR for new contexts): the first pass of the inverse transform of the phase correlation takes
 * its lines in flat order instead of as partner pairs (kz, ky), (-kz, -ky) (bit for bit the same; the pairs fetch the packed spectrum once).
 * "fft_slab_axes" = mask (bit k = axis k of (z, y, x); default 4; environment MVS_FFT_SLAB_AXES), "fft_no_slab" = 1 (MVS_FFT_NO_SLAB):
 * crops with ONE short axis (17-64 samples, a whole-line length) and two axes of 64 / 128 / 256 samples whose short axis is in the mask
 * run the phase correlation in three passes over HBM instead of six -- two axes per workgroup, the third forward / cross power /
 * inverse in one kernel (mvs_fft_slab.hip); same peaks and shifts, peak heights to float32 rounding (tests compare).  By default only
 * crops whose short axis is the contiguous one take it: inside the 8-lane pair loop the others are not faster for having half the traffic
 * (profiles/round5_fft_slab.txt).
 * "cb_unpaired" = 1: content-based weights filter value and mask lines in separate launches with separate preparation / quotient
 * kernels (rounds 1-3) instead of gauss1d_pair_kernel; "cb_nosplit" = 1: the paired path keeps both quantities in one workgroup
 * on every pass (all three bit for bit equal; tests compare).  "ssim_prune" = 0: mvs_register_crops / mvs_register_views score
 * every candidate completely (default 1: the arg-max search below; the environment variable MVS_SSIM_PRUNE=0 sets the default
 * of every context created afterwards).  The reference keeps only the candidate with the best SSIM and its rank correlation
 * (registration.py:558-565), and the per-voxel SSIM is <= 1, so a candidate whose partial sum plus (1 + slack) per voxel not
 * yet visited is below the sum of a completely scored candidate cannot win: the candidates of a pair are walked in rounds over
 * growing parts of the volume and dropped as soon as that holds (same selected translation, same quality, same status --
 * tests compare both settings and the oracle).  mvs_score_candidates itself always scores in full. */
int mvs_set_option(int device, const char* key, int64_t value);
/* Measurement counters of one context (bench.py): "reg_alg_bytes" = algorithmic HBM bytes of the pairwise registrations
 * since the last reset (28 n per phase-correlation variant + 20 n per scored candidate + 64 n for the rank correlation, n = crop
 * voxels; a candidate the arg-max search stopped counts the fraction of its volume it was scored on; "reg_alg_bytes_full": every
 * scored candidate counted whole, the reference's formulation), "reg_pairs",
 * "reg_candidates" (candidates that entered the scoring), "reg_pruned" (of these, left unfinished), "reg_cand_volumes"
 * (candidate volumes the SSIM passes went through), "reg_slab_pairs" (phase correlations that ran in three passes: crops with one
 * short and two power-of-two axes; option "fft_no_slab" switches that form off), "fuse_plan_ms" = host time the last mvs_fuse_chunk spent decomposing the chunk
 * (0 when the plan cached for the same geometry was reused).  Per class k of the last region-kernel launch (0 one-view rim
 * boxes, 1 NV = 2, 2 NV <= 4, 3 NV <= 8, 4 copy): "fuse_class_in_vox_<k>" (voxels x views of the class's boxes),
 * "fuse_class_out_vox_<k>", and "fuse_class_ms_<k>" = the class kernel's own duration when that launch ran with option
 * "serial_classes" = 1 (-1 otherwise).  "pool_misses" / "pool_miss_bytes" / "pool_releases": hipMalloc calls (and their bytes) that
 * mvs_malloc could not serve from its cache, blocks mvs_free handed back to the runtime.  reset != 0 clears an accumulating
 * counter after reading. */
int mvs_get_counter(int device, const char* key, int32_t reset, double* value_out);
/* Device time (ms, hipEvent) spent in the kernels of the most recent compute
 * call on this device; blocks until that work has finished. */
double mvs_last_kernel_ms(int device);

/* ---- device memory / tile residency ----------------------------------------- *
 * The reference moves every chunk across PCIe twice (cp.asarray on entry,
 * cp.asnumpy on exit: fusion/_core.py:1584-1587, 1716-1721).  These handles let
 * register() and fuse() share ONE upload per tile. */
/* mvs_malloc / mvs_free go through a caching pool (a pairwise registration allocates a handful of
 * overlap-sized buffers per pair; hipMalloc/hipFree would cost more than its kernels): memory is
 * uninitialised, and a freed block may be handed out again while earlier work on the context's
 * stream is still queued -- which is safe because all work of this library is ordered on that stream. */
int mvs_malloc(int device, uint64_t nbytes, void** dev_ptr);
int mvs_free(int device, void* dev_ptr);
/* Free / total device memory of the GPU behind `device` (hipMemGetInfo + what the allocation cache of this context would
 * give back).  Used by fusion.fuse to size its launch blocks: output + staged view slabs must fit (the reference sizes its
 * work by output_chunksize alone, fusion/_core.py:248-277, because every chunk there is a separate host task). */
int mvs_mem_info(int device, uint64_t* free_bytes, uint64_t* total_bytes);
int mvs_memcpy_h2d(int device, void* dst_dev, const void* src_host, uint64_t nbytes);
int mvs_memcpy_d2h(int device, void* dst_host, const void* src_dev, uint64_t nbytes);
int mvs_upload_tile(int device, const void* host, int32_t dtype, const int64_t shape[3], void** dev_ptr);
/* Copy between two GPUs of the node (peer access over xGMI when available): ordered after the work queued on the
 * source context, returns when the bytes have arrived.  The multi-GPU farms fetch the halo tiles of a pair / chunk
 * owner from the neighbour that holds them with this (reference precedent: browser/executors.py:166-194, 267-281,
 * where every worker re-reads its inputs from storage). */
int mvs_memcpy_peer(int dst_device, void* dst_dev, int src_device, const void* src_dev, uint64_t nbytes);
/* Stream-ordered byte fill of device memory (returns without waiting). */
int mvs_memset(int device, void* dst_dev, int32_t byte_value, uint64_t nbytes);
/* Device-to-device copy of a contiguous (z,y,x) box into a window of a larger contiguous array (both on this
 * device; stream-ordered, returns without waiting): how the chunks of a chunked fuse() are assembled into one
 * device-resident mosaic (the reference's chunks each go to their own zarr region, fusion/_core.py:1123-1141). */
int mvs_copy_into(int device, const void* src_dev, int32_t dtype, const int64_t shape[3],
                  void* dst_dev, const int64_t dst_shape[3], const int64_t dst_offset[3]);
/* Device-to-device copy of a box of box[0] planes x box[1] rows x box[2] BYTES per row between two pitched arrays on this device
 * (pitch[0]: bytes from row to row, pitch[1]: from plane to plane; rows contiguous; stream-ordered, returns without waiting).  A
 * streamed fuse() re-tiles a fused launch block with it so that every Zarr chunk of the block is ONE contiguous piece of the
 * download and goes to its chunk file without a gather on the host (fusion/_core.py:1123-1171 writes regions of a dask array). */
int mvs_copy_box(int device, const void* src_dev, const int64_t src_pitch[2], void* dst_dev, const int64_t dst_pitch[2], const int64_t box[3]);

/* ---- fusion ------------------------------------------------------------------ *
 * mvs_fuse_chunk == the body of fusion.fuse_np (fusion/_core.py:1608-1713) for
 * the built-in fusion_func / weights_func: per view affine resample
 * (transformation.py:136-139 -> scipy.ndimage.affine_transform, cval = NaN),
 * blending weights (weights.py:391-511), mask by ~isnan + normalise
 * (_core.py:1648-1649, weights.py:325-345), optional content-based weights,
 * fusion_func, trim halo, nan_to_num, cast to the input dtype -- as ONE fused
 * kernel (no per-view float32 temporaries).  `out` has shape out_shape - 2*trim. */
int mvs_fuse_chunk(int device, const mvs_view_t* views, int32_t n_views,
                   const mvs_fuse_opts_t* opts, void* out);

/* Single-view resample == transformation.transform_sim's scipy call
 * (transformation.py:136-139) with order 0|1, mode="constant", cval; float32
 * output.  Used for registration's pre-transform (registration.py:318-338) and
 * for fusion with user-supplied fusion_func/weights_func callables.
 * Only matrix/offset/data/shape/stride/dtype/mem of `view` are read. */
int mvs_resample(int device, const mvs_view_t* view, const int64_t out_shape[3],
                 int32_t order, float cval, float* out, int32_t out_mem);

/* Blending-weight volume of one view == weights.get_blending_weights
 * (weights.py:391-511): resampled 5^ndim support + cosine ramp, NOT normalised. */
int mvs_blend_weights(int device, const mvs_view_t* view, int32_t ndim,
                      const int64_t out_shape[3], float* out, int32_t out_mem);

/* ---- registration ------------------------------------------------------------ *
 * mvs_phasecorr == skimage.registration.phase_cross_correlation(fixed, moving,
 * normalization=None|"phase", upsample_factor=u, disambiguate=False)[0] as
 * called from registration.py:422-431:  F=fftn(a), G=fftn(b), P=F*conj(G),
 * [P /= max(|P|, 100 eps)], cc = ifftn(P), integer peak = argmax|cc| (lowest
 * flat index wins ties), wrap to signed shift, then upsampled-DFT refinement.
 * Inputs are float32, NaN-free, same shape; complex64 arithmetic.
 *   shift_out[3]      final (sub-pixel) shift, z,y,x (0 for the unused z in 2D)
 *   peak_index_out[3] integer argmax index before wrapping (bit-exact target)
 *   peak_abs_out      |cc| at that index                                        */
int mvs_phasecorr(int device, const float* fixed, const float* moving, int32_t mem,
                  int32_t ndim, const int64_t shape[3], int32_t normalization,
                  int32_t upsample_factor, double shift_out[3],
                  int64_t peak_index_out[3], float* peak_abs_out);
/* In-place complex64 (interleaved re, im) transform of one C-contiguous (z,y,x) array == numpy.fft.fftn /
 * ifftn (inverse != 0: conjugate transform WITHOUT the 1/N).  Any axis length up to 2^22 (powers of two: register /
 * LDS Stockham up to 4096, four-step in device scratch beyond; other lengths: Bluestein on the same cores, four-step above
 * 2048; longer axes return MVS_ERR_UNSUPPORTED).  The building block of mvs_phasecorr, exposed for tests and custom
 * pairwise registration functions. */
int mvs_fft_c2c(int device, void* data, int32_t mem, int32_t ndim, const int64_t shape[3], int32_t inverse);
/* The same for n_norm normalisations of ONE image pair (the reference calls phase_cross_correlation
 * with "phase" and None on the same inputs, registration.py:413-431): the two forward transforms
 * are computed once.  shifts_out: n_norm x 3, peak_indices_out: n_norm x 3 (may be NULL),
 * peak_abs_out: n_norm (may be NULL). */
int mvs_phasecorr_multi(int device, const float* fixed, const float* moving, int32_t mem,
                        int32_t ndim, const int64_t shape[3], const int32_t* normalizations,
                        int32_t n_norm, int32_t upsample_factor, double* shifts_out,
                        int64_t* peak_indices_out, float* peak_abs_out);

/* Intensity normalisation of one registration input == skimage.exposure.rescale_intensity(im,
 * in_range=(nanmin(im), nanmax(im)), out_range=(0, 1)) as called at registration.py:381-389:
 * float32 in/out, NaN preserved; also reports nanmin, nanmax and the number of non-NaN voxels
 * (valid_pixels1 of registration.py:400). */
int mvs_rescale_intensity(int device, const float* in, int32_t mem, int64_t n, float* out, int32_t out_mem,
                          float* min_out, float* max_out, int64_t* nvalid_out);

/* Block-mean binning of one view == sim.coarsen(bins, boundary="trim").mean().astype(dtype)
 * (registration.py:1732-1741): out shape = shape // bin, mean in double, cast like astype. */
int mvs_bin_mean(int device, const void* in, int32_t dtype, int32_t mem, const int64_t shape[3],
                 const int64_t stride[3], const int64_t bin[3], void* out, int32_t out_mem);

/* mvs_bin_mean between device buffers without the final wait: the kernel is queued on the context lane's stream and the
 * call returns; mvs_synchronize(device) orders later use (other lanes read the result only after it).  Lets a host that
 * has other work -- registration.register builds and prunes its overlap graph -- bin all tiles of a mosaic meanwhile
 * (reference: the same coarsen().mean() of registration.py:1732-1741, there part of the lazy dask graph). */
int mvs_bin_mean_async(int device, const void* in, int32_t dtype, const int64_t shape[3], const int64_t stride[3],
                       const int64_t bin[3], void* out);

/* mvs_bin_mean_async for n_views views of one shape, stride and dtype (in[v] -> out[v], device memory) in one call: the
 * binning of all tiles of a mosaic is queued with one library call (16-bit tiles binned by 2 along x: one launch per 32 views). */
int mvs_bin_mean_batch_async(int device, int32_t n_views, const void* const* in, int32_t dtype, const int64_t shape[3],
                             const int64_t stride[3], const int64_t bin[3], void* const* out);

/* Stream-ordered dependencies between context lanes, no host wait (the reference's dask graph orders the binning of a view
 * before the pairs that read it, registration.py:1732-1741 -> 2657-2664; here the binning runs on one lane and the pairs on
 * the others).  mvs_event_record marks the work queued so far on this lane and returns a ticket; mvs_event_wait(device,
 * ticket) makes the work queued LATER on `device`'s lane wait for that mark.  A lane keeps 32 tickets: an older one stands for
 * the mark that replaced it (the wait is longer, never shorter). */
int mvs_event_record(int device, uint64_t* ticket_out);
int mvs_event_wait(int device, uint64_t ticket);

/* ---- Transfers that overlap with the kernels (csrc/mvs_transfer.hip).  The reference's users hand fuse() host- or Zarr-backed
 * arrays and stream the fused chunks out (fusion/_core.py:1068-1170, 2044-2156; spatial_image_utils.py:712-860); SURVEY 8d(2)
 * counts H2D / D2H into the end-to-end figure.  mvs_host_alloc: pinned host memory (what an asynchronous copy needs to be
 * asynchronous).  mvs_copy_async: one copy (kind 0: host -> device, 1: device -> host) on the device's COPY STREAM of that direction -- one
 * per device and direction (uploads and downloads run side by side, each direction in the order queued), created with a priority of
 * their own so that queued copies do not stall the compute streams -- started after ticket
 * `after` (0: at once; a ticket of mvs_event_record, mvs_mark or an earlier mvs_copy_async) and marked by the ticket *done_out.
 * Tickets of this group are timed events from a ring of 4096 per device; mvs_event_wait accepts them, so a pair job of
 * mvs_register_pairs (wait_ticket) starts when its two tiles have landed, and a download starts when the launch that produced
 * its data is done (mvs_mark on the producing lane).  mvs_ticket_sync: the host waits for a ticket; mvs_ticket_elapsed_ms: the
 * time between two tickets of one device (both must have passed or the call waits for them).  Host pointers must stay valid, and
 * unchanged for uploads, until the ticket has passed. */
int mvs_host_alloc(uint64_t nbytes, void** host_ptr);
int mvs_host_free(void* host_ptr);
int mvs_copy_async(int device, void* dst, const void* src, uint64_t nbytes, int32_t kind, uint64_t after, uint64_t* done_out);
int mvs_mark(int device, uint64_t* ticket_out);
int mvs_ticket_sync(uint64_t ticket);
int mvs_ticket_elapsed_ms(uint64_t t0, uint64_t t1, double* ms_out);

/* Candidate scoring == the loop of registration.py:493-556 for n translation
 * candidates t (z,y,x rows): moving resampled by t (order 1, NaN outside),
 * masks, bounding-box region (region_mode 0 = "union", 1 = "intersection"),
 * SSIM (ssim_out) and masked Spearman correlation (spearman_out).  A skipped
 * candidate reports code_out = 1 (mask empty / <10% valid -> both metrics -1),
 * 2 (the `continue` of registration.py:530-533, no metric appended), else 0.
 * quality_for_all = 0: the rank correlation is evaluated only for the candidate(s)
 * holding the best SSIM -- the only one the reference reports (registration.py:543-556) --
 * and spearman_out is NaN for the other scored candidates; 1: for every candidate. */
int mvs_score_candidates(int device, const float* fixed, const float* moving, int32_t mem,
                         int32_t ndim, const int64_t shape[3],
                         const double* t_candidates, int32_t n_candidates,
                         int32_t region_mode, double data_range, double im1_min,
                         int32_t quality_for_all,
                         double* ssim_out, double* spearman_out, int32_t* code_out);

/* The whole of registration.phase_correlation_registration (registration.py:353-565) for two same-shape
 * float32 overlap crops (NaN = outside the view) in one call: intensity normalisation, the two phase
 * correlations, the zero-shift candidate when a crop holds NaNs (quirk Q1), candidate enumeration, scoring and
 * the nanargmax selection with the reference's list bookkeeping (quirk Q3).  upsample_factor: the reference uses
 * 10 in 2D and 2 in 3D; region_mode -1 = the reference's choice ("intersection" with NaNs, else "union");
 * constant_check != 0 adds dispatch_pairwise_reg_func's guard (registration.py:1500-1520).
 * t_out: translation (z,y,x; fixed px -> moving px), quality_out: Spearman coefficient of the selected candidate.
 * status_out: 0 = ok, 1 = no admissible candidate (the reference returns [zeros(ndim)], quirk Q2),
 * 2 = constant crop (identity, quality NaN), 3 = no finite SSIM (np.nanargmax raises in the reference). */
int mvs_register_crops(int device, const float* fixed, const float* moving, int32_t mem, int32_t ndim,
                       const int64_t shape[3], int32_t upsample_factor, int32_t region_mode,
                       int32_t constant_check, double t_out[3], double* quality_out,
                       int32_t* status_out, int32_t* n_candidates_out);

/* One call per image pair == sims_to_intrinsic_coord_system + dispatch_pairwise_reg_func(phase_correlation_registration)
 * (registration.py:280-350, 1477-1544, 353-565): both views (mvs_view_t geometry as for mvs_resample: matrix / offset map
 * pixels of the fixed view's overlap grid to pixels of the view's slab) are resampled with order 1 and NaN outside onto
 * out_shape (float32, library scratch) and handed to mvs_register_crops without a host round trip in between.  Outputs as
 * mvs_register_crops. */
int mvs_register_views(int device, const mvs_view_t* fixed_view, const mvs_view_t* moving_view, int32_t ndim,
                       const int64_t out_shape[3], int32_t upsample_factor, int32_t region_mode, int32_t constant_check,
                       double t_out[3], double* quality_out, int32_t* status_out, int32_t* n_candidates_out);

/* All pairs of a mosaic in one call (registration.compute_pairwise_registrations, registration.py:2622-2714: one task per
 * pair; here one library call, the pairs farmed over context lanes by native worker threads -- no interpreter in the loop).
 *
 * mvs_plan_pairs (host only): for pairs of views whose transform is a pure translation, what register_pair_of_msims ->
 * sims_to_intrinsic_coord_system -> get_pixel_affine derive per pair (registration.py:194-350, 1547-2058; transformation.py:37-83).
 * coords[v * ndim + k]: the coordinate array of view v along axis k (coord_len entries; origin = c[0], spacing = c[1] - c[0]);
 * translation: n_views x ndim; tol: ndim overlap tolerances or NULL; pairs: n_pairs x 2 (fixed, moving).  Per pair p:
 * windows_out[((p * 2 + i) * 3 + k) * 2 + {0, 1}] = first / one-past-last index of view i's crop window on axis k (the overlap
 * + one sample + 1e-6 on either side), out_origin / out_spacing / out_shape (p * 3 + k) = the fixed view's overlap grid,
 * matrix_diag / offset ((p * 2 + i) * 3 + k) = the pixel affine of crop i (output pixel -> window pixel; rounded to 10
 * decimals, offsets within 1e-6 of an integer snapped), status_out[p] = 0 ok, 1 the views do not overlap.  Axes k < ndim.
 *
 * mvs_register_pairs: mvs_register_views for every job on n_lanes (<= 16) native worker threads, thread w driving context lane
 * `device | w << 8`; `device` carries no lane.  wait_ticket: mvs_event_record tickets (0 = none) the lane's stream waits for
 * before it reads the fixed / moving view (binned tiles still being produced on another lane).  Outputs per pair as
 * mvs_register_views (t_out: n_pairs x 3); rc_out[p] = that pair's return code.  Returns 0, or the first failing pair's code
 * with its message in mvs_last_error(device). */
typedef struct mvs_pair_job_t {
    mvs_view_t fixed;
    mvs_view_t moving;
    int64_t out_shape[3];
    uint64_t wait_ticket[2];
    int32_t bin[3];          /* bin[0] > 0: `fixed` / `moving` are windows of RAW uint8 / uint16 tiles (shape = a whole number of bins
                                per axis, identity matrix, offset = the whole-pixel translation in BINNED pixels) and the crops are
                                taken with this registration binning applied on the fly -- sim.coarsen(bin).mean().astype(dtype)
                                (registration.py:1732-1741) and the crop in one pass, no binned copy of the tiles; all zero: the
                                views are used as they are (mvs_register_views) */
    int32_t flags;           /* bit 0: when the pair is done, wait_ticket[0] is OVERWRITTEN with a timed ticket (mvs_mark) of the pair's
                                lane -- when its last kernel finished, for timelines (mvs_ticket_elapsed_ms); 0: the job is only read */
} mvs_pair_job_t;
int mvs_plan_pairs(int32_t ndim, int32_t n_views, const double* const* coords, const int64_t* coord_len, const double* translation,
                   const double* tol, int32_t n_pairs, const int32_t* pairs, int64_t* windows_out, double* out_origin_out,
                   double* out_spacing_out, int64_t* out_shape_out, double* matrix_diag_out, double* offset_out, int32_t* status_out);
int mvs_register_pairs(int device, int32_t n_pairs, mvs_pair_job_t* jobs, int32_t ndim, int32_t upsample_factor,
                       int32_t region_mode, int32_t constant_check, int32_t n_lanes, double* t_out, double* quality_out,
                       int32_t* status_out, int32_t* n_candidates_out, int32_t* rc_out);

/* Host-only (no device, no mvs_init needed): the inner loop of the reference's global optimisation for the translation
 * model -- optimize_bead_subgraph, param_resolution/global_optimization.py:313-417 with transforms.py:45-53 as estimator.
 * Edge e joins nodes edge_nodes[2e], edge_nodes[2e+1] and carries n_beads virtual beads in each node's frame
 * (beads_a / beads_b: n_edges x n_beads x ndim doubles, param_resolution/utils.py:42-78).  Sweep: the nodes in `order`
 * (ref_node and nodes without edges are skipped) each add mean(adjacent bead - own bead), taken in world coordinates over
 * all their beads, to their translation.  After each sweep the bead residuals |a + t_a - b - t_b| are stored
 * (edge_residuals: n_edges x n_beads), their mean of edge means and maximum are appended to mean_hist / max_hist
 * (max_iter doubles each); from the 7th sweep on the loop stops when max |r - r_previous| / max r < rel_tol.
 * translations (n_nodes x ndim) is in/out; n_iter_out = sweeps done.
 * Contract with the numpy form of the same sweeps (param_resolution.py, used for the models with a linear part): the node
 * update is evaluated in affine form (sum over a node's edges of +-(D0_e + n_beads (T_a - T_b))), i.e. its additions run in
 * another order than the bead-by-bead sum -- equal to 1e-10 in translations, residuals and both histories, NOT bit for bit
 * (tests/test_param_resolution.py).  The AVX2 residual pass (x86-64 hosts that have it) equals the scalar loops bit for bit. */
int mvs_beads_translation_sweeps(int32_t ndim, int32_t n_nodes, int32_t n_edges, const int32_t* edge_nodes,
                                 const double* beads_a, const double* beads_b, int32_t n_beads, const int32_t* order,
                                 int32_t ref_node, int32_t max_iter, double rel_tol, double* translations,
                                 double* edge_residuals, double* mean_hist, double* max_hist, int32_t* n_iter_out);

/* Host-only: the chunk -> view-slab plan of fusion.fuse (fusion/_core.py:354-722 with mv_graph.py:934-1117 and the label
 * selection of _core.py:1371-1386).  All per-axis arrays hold `ndim` entries per item in (z,) y, x order: views as
 * origin / spacing / shape of their stacks and (ndim + 1)^2 row-major affines `params` (view -> world; `inv_params` = their
 * inverses, needed -- and only read -- when some view is not a pure translation), the output stack, the chunk size and the
 * halo (overlap_in_pixels).  Decides which axes are pure translations for every view (dim_masks_out[0], bit d = axis d) and on
 * which of those the output samples fall on every view's sampling grid (dim_masks_out[1]: no interpolation taps there,
 * _core.py:354-459), then lists, in block order (first axis slowest) and ascending view index, one entry per (output chunk,
 * contributing view): the integer window lo[d] .. lo[d] + n[d] - 1 of the view that the chunk (grown by the halo and by
 * `interpolation_order` taps on the axes that interpolate) needs (_core.py:462-533 for translations, mv_graph.py:989-1117
 * for general affines), exactly the samples the reference's `sims[iview].sel(...)` selects.  planewise = 1 when the chunk is a
 * single z plane on the views' z grid (fused with the 2D parameters, _core.py:694-703).  Call with entries = NULL to get
 * the count in n_entries_out, then with capacity >= that count. */
typedef struct mvs_plan_entry {
    int64_t block[3];
    int32_t view;
    int32_t planewise;
    int64_t lo[3];
    int64_t n[3];
} mvs_plan_entry_t;
int mvs_fuse_plan(int32_t ndim, int32_t n_views, const double* view_origin, const double* view_spacing, const int64_t* view_shape,
                  const double* params, const double* inv_params, const double* out_origin, const double* out_spacing,
                  const int64_t* out_shape, const int64_t* chunk_size, const int64_t* halo, int32_t interpolation_order,
                  mvs_plan_entry_t* entries, int64_t capacity, int64_t* n_entries_out, int32_t* dim_masks_out);

/* Host-only: edge betweenness centrality of the view adjacency graph -- networkx.edge_betweenness_centrality(g) as
 * prune_graph_to_alternating_colors calls it (mv_graph.py:664-741; Brandes, unweighted, normalised by n (n - 1)).  Nodes are
 * 0 .. n_nodes - 1 in the graph's node order, node v's neighbours adj_nodes[adj_offsets[v] .. adj_offsets[v + 1]) in
 * adjacency order, adj_edge[a] = index of the edge adjacency entry a belongs to; bet_out: n_edges doubles.  Traversal and
 * accumulation order are networkx's (the reference compares the derived edge values with <=). */
int mvs_edge_betweenness(int32_t n_nodes, int32_t n_edges, const int32_t* adj_offsets, const int32_t* adj_nodes,
                         const int32_t* adj_edge, double* bet_out);

/* Host-only: the pairs registration.register registers, for views whose world frames are axis-aligned boxes -- the overlap
 * graph of mv_graph.build_view_adjacency_graph_from_msims (mv_graph.py:35-180) and, with method 1, its pruning by
 * prune_graph_to_alternating_colors (mv_graph.py:664-741, the default pre_registration_pruning_method) in one call.
 * box_lo / box_hi: n_views x ndim world coordinates of the first / last sample of every view (after any overlap tolerance);
 * pairs: n_pairs x 2 candidate view indices in the order the reference's cKDTree ball query lists them (i != j; (j, i) after
 * (i, j) changes nothing).  An unordered pair becomes an edge at its first appearance when its intersection volume
 * prod(min(hi) - max(lo)) is positive on every axis (the value Qhull returns for the box).  method 0: all edges; method 1:
 * edges are removed level by level in rising order of overlap + betweenness bonus (an edge with an end point of degree 1 stays)
 * until a greedy largest-first colouring needs at most n_colors colours.  Output: the surviving edges in networkx's edges()
 * order (node by node, neighbours in insertion order), each (i, j) with i < j, and their overlap volumes; edges_out holds
 * 2 * n_pairs int32, overlap_out n_pairs doubles at most.  n_graph_edges_out (optional): edges before pruning.
 * MVS_ERR_UNSUPPORTED: a case the Python form decides (NaN volumes, no colouring within the levels). */
int mvs_view_graph_prune(int32_t ndim, int32_t n_views, const double* box_lo, const double* box_hi, int64_t n_pairs,
                         const int32_t* pairs, int32_t method, int32_t n_colors, int32_t* edges_out, double* overlap_out,
                         int32_t* n_edges_out, int32_t* n_graph_edges_out);

/* Host-only: param_resolution.groupwise_resolution(method="global_optimization", transform="translation")
 * (param_resolution/__init__.py:44-150, global_optimization.py:16-511, utils.py:42-101) for a CONNECTED mosaic whose pairwise
 * results are pure translations, in one call.  edges: n_edges x 2 view indices (i < j) in the order the pairs were added to the
 * registration graph; pair_t: n_edges x ndim translation of each pair's transform; quality: n_edges; bbox_lo / bbox_hi:
 * n_edges x ndim overlap box of each pair in the fixed view's frame; spacing: n_views x ndim (abs_tol < 0: the largest voxel
 * diagonal is used).  reference_view < 0: the view with the largest sum of edge qualities.  Outputs: translations_out
 * (n_views x ndim, the reference view keeps 0), edge_rms_out (n_edges, RMS bead residual per edge in input edge order),
 * mean_hist / max_hist (max_iter doubles each: mean of edge means / maximum of the bead residuals after every sweep),
 * n_iter_out, ref_out (optional).  Returns MVS_ERR_UNSUPPORTED -- and the caller takes the Python form -- when the graph is not
 * one component over all views, an edge is duplicated, an input is not finite, or the final maximal residual is not below
 * abs_tol (the reference then starts removing edges: global_optimization.py:419-505). */
int mvs_resolve_translations(int32_t ndim, int32_t n_views, int32_t n_edges, const int32_t* edges, const double* pair_t,
                             const double* quality, const double* bbox_lo, const double* bbox_hi, const double* spacing,
                             int32_t reference_view, int32_t max_iter, double rel_tol, double abs_tol, double* translations_out,
                             double* edge_rms_out, double* mean_hist, double* max_hist, int32_t* n_iter_out, int32_t* ref_out);

#ifdef __cplusplus
}
#endif
#endif /* MVS_HIP_H */
