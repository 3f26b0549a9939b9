/*
 * klt_gpu.h -- C ABI of libkltgpu.so, the MI355X (gfx950) backend for the PyFeatureTrack KLT
 * hot path: pyramid build -> min-eigenvalue corner selection -> per-feature Newton tracking.
 *
 * The reference (TimSC/PyFeatureTrack) has no FFI of its own; its only native boundary is the
 * Cython `def` layer (setup.py:8-9).  Each entry point below names the reference interface it
 * replaces (file:line relative to the reference root).  Host code stays Python and binds this
 * header with ctypes (pyfeaturetrack_amd/_abi.py; INTEGRATION.md shows the stub a reference
 * maintainer would add).
 *
 * Conventions
 *   - plain C types only; no torch / HIP types cross the boundary;
 *   - every function returns 0 on success or a negative klt_status (klt_timing_read: the entry count; klt_slot_state: the
 *     state bits; klt_select_finish: also 1, "done, and the list was rewritten on the way" -- see there); the message is available
 *     from klt_last_error(); nothing exits the process or throws across the ABI
 *     (the reference's KLTError prints and calls exit(1), error.py:12-14);
 *   - per-feature failures are data (klt_feat.val < 0, klt.py:23-29), not errors;
 *   - host buffers are caller-owned and only borrowed for the duration of the call;
 *   - device buffers are owned by the context: frames live in numbered *slots* (u8/f32 image +
 *     the three pyramids), feature lists in numbered *feature buffers*;
 *   - a context owns one HIP stream; *_async entry points only enqueue, everything else has
 *     completed on return.  One context per host thread / device; no shared mutable globals;
 *   - there is NO CPU path: klt_create fails if the device cannot be opened.
 */
#ifndef KLT_GPU_H
#define KLT_GPU_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define KLT_ABI_VERSION 11
#define KLT_MAX_KERNEL_WIDTH 71   /* convolve.py:28 */
#define KLT_MAX_LEVELS 8

typedef enum {
    KLT_OK = 0,
    KLT_ERR_ARG = -1,        /* bad argument / unsupported parameter combination */
    KLT_ERR_DEVICE = -2,     /* HIP runtime error */
    KLT_ERR_STATE = -3,      /* call order (e.g. tracking a slot whose pyramids were never built) */
    KLT_ERR_NOMEM = -4,      /* a device or pinned-host allocation could not be had (klt_last_error names the size asked for); nothing is left
                              * half-allocated and no stale error is left with the HIP runtime: free something and repeat the call */
    KLT_ERR_TIMEOUT = -5     /* a host-side wait for a collective gave up (klt_comm_set_timeout): a peer is gone or stuck */
} klt_status;

/* feature status codes, klt.py:23-29 (kltState) */
enum { KLT_TRACKED = 0, KLT_NOT_FOUND = -1, KLT_SMALL_DET = -2, KLT_MAX_ITERATIONS = -3,
       KLT_OOB = -4, KLT_LARGE_RESIDUE = -5 };

/* selection modes, selectGoodFeatures.py:11-13 */
enum { KLT_SELECTING_ALL = 1, KLT_REPLACING_SOME = 2 };

/* 16-byte feature record; replaces the KLT_Feature attribute bag (klt.py:249-263).
 * x, y hold the reference's Python floats (integers after selection, f32-valued after tracking,
 * -1 when lost); val is the status / int(eigenvalue). */
typedef struct { float x, y; int32_t val; int32_t aux; } klt_feat;

/* POD mirror of KLT_TrackingContext (klt.py:45-73) plus the values the reference derives from it.
 * borderx/bordery are Python floats in the reference (e.g. 30.0, klt.py:186-189). */
typedef struct {
    int32_t mindist, window_width, window_height;
    int32_t smoothBeforeSelecting, retainTrackers, nSkippedPixels;
    int32_t max_iterations, nPyramidLevels, subsampling, use_max_residue;
    float   min_determinant, min_displacement, step_factor, max_residue;
    double  min_eigenvalue;
    double  grad_sigma, smooth_sigma, pyramid_sigma;
    double  borderx, bordery;
} klt_params;

typedef struct klt_ctx klt_ctx;

/* ---- lifetime ------------------------------------------------------------------------------ */
int         klt_abi_version(void);
int         klt_device_count(void);
int         klt_create(int device, klt_ctx **out);          /* replaces KLT_TrackingContext() device state, klt.py:43-81 */
void        klt_destroy(klt_ctx *ctx);
const char *klt_last_error(klt_ctx *ctx);                   /* ctx may be NULL: error of the last failed klt_create */
int         klt_sync(klt_ctx *ctx);                         /* wait for everything enqueued on the context's stream */
void       *klt_stream_handle(klt_ctx *ctx);                /* the context's hipStream_t, for callers that order RCCL collectives after it */

/* ---- options ------------------------------------------------------------------------------ */
#define KLT_OPT_FUSED_KERNELS 1   /* 1 (default): LDS-tiled fused pyramid kernels; 0: generic two-pass kernels (any tap count) */
/* affine state (klt_affine_alloc id, or -1) whose records klt_select* resets for every slot it fills
 * (selectGoodFeatures.py:120-128 clears the aff_* fields of a newly placed feature) */
#define KLT_OPT_SELECT_AFFINE_STATE 4
/* 1 (default): klt_select* only considers the highest-scoring candidates (histogram threshold keeping about 64 per
 * requested feature) and repeats with every candidate if they run out before the list is full; 0: always every candidate. */
#define KLT_OPT_TOPK_PREFILTER 5
/* 1 (default): minimum-distance enforcement as parallel passes over all candidates (a candidate is accepted once every
 * higher-ranked neighbour is rejected; same result as the walk); 0: sort all candidates and walk them in order.
 * Either way klt_select_async synchronises once internally (it reads back whether the list was filled / settled). */
#define KLT_OPT_SELECT_PARALLEL_NMS 8
#define KLT_OPT_SAT_VARIANT 10           /* summed-area tables: 1 (default) step-synchronous wavefront pipelines (frames with ncols % 4 == 0), 0 barrier-coupled kernels */
#define KLT_OPT_TRACK_VARIANT 11         /* 4 (default): quad-load tracker kernels (7x7: four features per wavefront; 15x15: one 16-byte load per lane and image); 0: plain one-feature-per-wavefront kernel for every window (same records; process-wide) */
#define KLT_OPT_FUSED_HREDUCE 12         /* 1 (default): the level-0 kernel also runs the horizontal pass of the first pyramid reduction (subsampling 4); 0: separate reduction kernel */
#define KLT_OPT_TRACK_XCD_ORDER 13        /* 1 (default): features are handed to the tracker sorted by row, one band of the image per XCD (single-pair and batched launches; the order is refreshed every 64 launches); 0: list order */
/* 1: pyramid builds are enqueued on a second HIP stream of the context (a second hardware queue), behind everything enqueued on
 * the main stream so far; trackers / selections / uploads that touch a slot wait (on the device) for the build that fills it.
 * A sequence can then enqueue the build of frame t+1 BEFORE the tracker and the replacement pass of frame t and the GPU overlaps
 * them (one event each way per frame).  0 (default): one stream.  The caller must give frame t+1 a slot that no call still to be
 * enqueued reads (a ring of three slots for a sequence). */
#define KLT_OPT_BUILD_STREAM 15
#define KLT_OPT_TRACK_TREE_SUMS 18        /* 0 (default): the tracker adds its five window sums (and the residue) in the reference's order -- records identical to the reference's bit for bit; 1: butterfly sums in registers (7x7 / 15x15 quad kernels; same precision, other order of the additions): positions agree to 1e-3 px, a status word can differ where a feature sits on a threshold */
#define KLT_OPT_SCORE_SETS 16            /* how many sets of prepared selection scores (klt_select_prepare_async) the context keeps: 2 (default) .. 256; a selection frees the set it uses */
/* 1 .. 8 (default 2; KLT_COPY_STREAMS in the environment sets the initial value): the copy streams consecutive klt_upload_u8_async calls
 * alternate between.  Two let the frames of a PAIR travel side by side (45 GB/s against 28-39 on one or three); a SEQUENCE loop -- one new
 * frame per step -- is faster on ONE (0.280 against 0.307 ms per 4K frame, 0.141-0.152 against 0.148-0.155 at 1080p: with two, the host's look
 * at every other frame's selection waits 211 instead of 102 us, profiles/README.md): KLTTrackSequence and bench.py's sequence loops set 1. */
#define KLT_OPT_COPY_STREAMS 20
/* test hook: >= 0: the library's (value + 1)-th device / pinned-host allocation from now is refused as if memory had run out (KLT_ERR_NOMEM; the
 * context stays usable, the call can be repeated); -1 (default): off.  Lets the tests walk every allocation site of a call sequence. */
#define KLT_OPT_FAIL_ALLOC_AFTER 19
int klt_set_option(klt_ctx *ctx, int option, int value);

/* ---- parameters and taps ------------------------------------------------------------------- */
int klt_set_params(klt_ctx *ctx, const klt_params *p);      /* klt.py:45-73, :84-128, :137-189 */
/* FP64 taps computed on the host exactly as convolve.py:27-93 (_computeKernels).
 * which: 0 = smoothing (sigma = smooth_sigma_fact*max(w,h)), 1 = pyramid (pyramid_sigma_fact*ss), 2 = gradient (grad_sigma) */
int klt_set_kernels(klt_ctx *ctx, int which, const double *gauss, int ng, const double *deriv, int nd);

/* ---- frames (slots) ------------------------------------------------------------------------ */
/* replaces `np.array(img.convert("F"))`, trackFeatures.py:165,176 / selectGoodFeatures.py:190.
 * pitch is in elements.  The upload is enqueued; the host buffer may be reused on return.
 * Limits: 1 <= ncols, nrows <= 65535 and ncols * nrows < 2^28 (the kernels address a plane with 32-bit byte offsets below 2 GB, and
 * the interleaved gradient plane of level 0 has 8 bytes per pixel); a frame
 * beyond them is KLT_ERR_ARG here, at klt_upload_u8_async and at the stand-alone convolution / pyramid calls. */
int klt_upload_u8(klt_ctx *ctx, int slot, const uint8_t *px, int ncols, int nrows, int pitch);
int klt_upload_f32(klt_ctx *ctx, int slot, const float *px, int ncols, int nrows, int pitch);
/* Asynchronous ingest: the frame is copied from PINNED host memory on a dedicated copy stream and overlaps
 * the kernels of earlier frames; the next build / selection of the slot waits for it on the device, the copy itself
 * waits for queued work that still reads the slot.  `px` must stay untouched until klt_upload_wait (or klt_sync) has
 * returned after the call. */
int klt_host_alloc(klt_ctx *ctx, size_t bytes, void **out);     /* pinned host memory, freed by klt_host_free / klt_destroy */
int klt_host_free(klt_ctx *ctx, void *p);
int klt_upload_u8_async(klt_ctx *ctx, int slot, const uint8_t *px, int ncols, int nrows, int pitch);
/* the same for a float32 frame (what `img.convert("F")` of a colour or float image gives, trackFeatures.py:165,176): pinned host memory,
 * copy streams, the slot's two alternating raw buffers -- 4 bytes per pixel on the link */
int klt_upload_f32_async(klt_ctx *ctx, int slot, const float *px, int ncols, int nrows, int pitch);
int klt_upload_wait(klt_ctx *ctx);      /* host waits for the copies issued so far (only the copy stream; kernels keep running) */
/* Host side of the ingest, no context involved (thread-safe; a process-wide pool of parked worker threads, KLT_HOST_THREADS lanes in
 * all, default 4; a caller that finds the pool busy does its own work): the reference converts and rebuilds both images on every call
 * (trackFeatures.py:146-196); a caller that keeps a frame's pyramids while the image still has EXACTLY the pixels the slot was filled
 * from pays one pass over the frame per call instead -- klt_host_compare: 0 = every byte equal, 1 = different --, and stages a new
 * frame into the pinned buffer klt_upload_u8_async reads with klt_host_copy. */
int klt_host_compare(const void *a, const void *b, size_t bytes);
int klt_host_copy(void *dst, const void *src, size_t bytes);
int klt_host_lanes(void);               /* lanes a large compare / copy is spread over (workers + the caller) */
/* The same two for a frame that is NOT one contiguous array but a table of row addresses -- Pillow's ImagingMemoryInstance.image8, which the
 * reference's callers hand in (trackFeatures.py:165,176 / selectGoodFeatures.py:190: `img.convert("F")` of a PIL image; the Python layer
 * takes the table from Image.getim(), pyfeaturetrack_amd/_pil.py): rows[r] = address of row r, row_bytes bytes each; `b` / `dst` a contiguous
 * [nrows x row_bytes] buffer (the pinned copy a slot was filled from).  Rows that follow each other in memory are treated as one range.
 * klt_host_sample_rows: rows[y][x] for y = 0, ystep, ... and x = 0, xstep, ... (row-major) into out; returns the count (the 1024-pixel
 * lattice of the frame cache without making an array of the image). */
int klt_host_compare_rows(const uint8_t *const *rows, int nrows, size_t row_bytes, const void *b);
int klt_host_copy_rows(void *dst, const uint8_t *const *rows, int nrows, size_t row_bytes);
/* on != 0: compares / copies issued by the CALLING THREAD from now on stay on that thread (a background thread that stages frames while
 * the main thread enqueues must not take the pool's lanes away from it: measured slower, profiles/README.md); returns the previous setting */
int klt_host_thread_serial(int on);
int klt_host_sample_rows(const uint8_t *const *rows, int nrows, int ncols, int ystep, int xstep, uint8_t *out, size_t cap);
/* `img.convert("F")` of a colour image without Pillow's intermediate image and numpy's copy of it: rows of 4-byte pixels (R, G, B, X --
 * Pillow's storage of "RGB" / "RGBA" / "RGBX" images, ImagingMemoryInstance.image32) -> dst[nrows][ncols] float32 =
 * (float)(299 R + 587 G + 114 B) / 1000.0f, Pillow's own expression (libImaging/Convert.c, rgb2f), spread over the pool's lanes */
int klt_host_luma_rows(float *dst, const uint8_t *const *rows, int nrows, int ncols);
/* Frames that are ALREADY in device memory (a hardware decoder's output, a clip kept resident): the slot's frame becomes this caller-owned
 * buffer -- read in place by the next build / selection of the slot, never copied, written or freed by the library (zero-copy
 * ingest).  pitch must equal ncols.  Host-only call: nothing is enqueued; the buffer must hold the frame already and stay unchanged until
 * the work that reads it has completed (klt_sync, or the download of what was computed from it).  The next upload into the slot ends the
 * adoption.  klt_device_alloc / klt_device_write / klt_device_free give callers without a HIP binding of their own (Python) such memory:
 * owned by the context, freed with it. */
int klt_slot_adopt_u8(klt_ctx *ctx, int slot, const uint8_t *dev_px, int ncols, int nrows, int pitch);
int klt_device_alloc(klt_ctx *ctx, size_t bytes, void **out);
int klt_device_write(klt_ctx *ctx, void *dev_dst, const void *host_src, size_t bytes);   /* synchronous host-to-device copy */
int klt_device_free(klt_ctx *ctx, void *dev);
/* smooth -> pyramid -> gradients of every level: ComputeImagePyramids for one image,
 * trackFeatures.py:165-172 + pyramid.py:37-77 + convolve.py:208-264 */
int klt_build_pyramids_async(klt_ctx *ctx, int slot);
/* the same for several slots at once: frames of equal size share kernel launches (both frames of a pair,
 * or every pair of a batch -- BASELINE cfg-4) */
int klt_build_pyramids_batch_async(klt_ctx *ctx, const int *slots, int n);
int klt_build_pyramids(klt_ctx *ctx, int slot);
/* bit 0: the slot holds a frame; bit 1: its pyramids are built and match the current parameters / taps (what
 * `tc.pyramid_last is not None` means in the reference, trackFeatures.py:152); 0 for a slot never used */
int klt_slot_state(klt_ctx *ctx, int slot);
/* free / total memory of the context's device (hipMemGetInfo): what a long-running sequence watches to see that slots, score sets and
 * descriptor tables are recycled, not accumulated */
int klt_device_memory(klt_ctx *ctx, size_t *free_bytes, size_t *total_bytes);
/* which build filled the slot's pyramids: a number unique per build within the context that travels with klt_swap_slots; 0 while the
 * slot has no valid pyramids.  The host layer's pyramid handles (what ComputeImagePyramids returns and tc.pyramid_last holds,
 * trackFeatures.py:146-196, :401-404) use it to tell whether the planes they stand for are still on the device. */
int klt_slot_generation(klt_ctx *ctx, int slot, uint64_t *gen);
/* releases the slot's device memory (raw frames, pyramids); the index can be used again.  The reference frees images and
 * pyramids when the Python objects die; the host layer calls this from a finalizer of the tracking context. */
int klt_slot_free(klt_ctx *ctx, int slot);
/* sequentialMode: the frame-2 pyramids become frame 1 (trackFeatures.py:152-161, :401-404) */
int klt_swap_slots(klt_ctx *ctx, int a, int b);

/* ---- feature buffers ----------------------------------------------------------------------- */
int   klt_featbuf_upload(klt_ctx *ctx, int fb, const klt_feat *src, int n);
/* the same without waiting for the copy: `src` must stay untouched until the next synchronising call on the context (klt_sync,
 * klt_featbuf_download, ...) */
int   klt_featbuf_upload_async(klt_ctx *ctx, int fb, const klt_feat *src, int n);
int   klt_featbuf_download(klt_ctx *ctx, int fb, klt_feat *dst, int n);
/* the same without draining the pipeline: the copy is enqueued in stream order behind the kernels that wrote the records; `dst` must be
 * pinned (klt_host_alloc) and holds the records once klt_download_wait (or any synchronising call) has returned.  A loop that reads a
 * table back every N steps issues this at the window's end and waits for it at the NEXT window's end: the host never waits for queued work. */
int   klt_featbuf_download_async(klt_ctx *ctx, int fb, klt_feat *dst, int n);
int   klt_download_wait(klt_ctx *ctx);                        /* host waits for the latest klt_featbuf_download_async / klt_download_mark_async */
/* marks this point of the main stream for klt_download_wait without copying anything: for records the kernels write straight into pinned
 * memory (klt_featbuf_map_host) when more work is enqueued behind the kernel that writes them -- KLTTrackFeatures (trackFeatures.py:205-409)
 * waits for its tracker only while the scores of the KLTReplaceLostFeatures call that follows are already being computed */
int   klt_download_mark_async(klt_ctx *ctx);
/* Feature buffer `fb` becomes `n` records of PINNED host memory (klt_host_alloc), read and written in place by the kernels over the
 * link: a call that sends a list, tracks it and waits (KLTTrackFeatures, trackFeatures.py:205-409) then needs no copy command in either
 * direction -- the records are in `host` once klt_sync (or any synchronising call) has returned.  Meant for the lists of such one-shot
 * calls; tables that kernels revisit belong in device memory.  host == NULL ends the mapping (the buffer is empty afterwards).  The
 * memory must outlive the mapping. */
int   klt_featbuf_map_host(klt_ctx *ctx, int fb, klt_feat *host, int n);
int   klt_featbuf_alloc(klt_ctx *ctx, int fb, int n);         /* n records, all marked lost (val = -1) */
/* fb_view becomes a window [offset, offset+n) of fb_parent (a device-side [frames x features] table, cf. the
 * KLT_FeatureTable stub at klt.py:278-283, can then be gathered with one collective).  The parent must outlive the
 * view and must not be resized while it exists. */
int   klt_featbuf_view(klt_ctx *ctx, int fb_view, int fb_parent, int offset, int n);
void *klt_featbuf_devptr(klt_ctx *ctx, int fb);             /* device address (for RCCL gathers); NULL if unset */

/* ---- selection: _KLTSelectGoodFeatures, selectGoodFeatures.py:141-261 ---------------------- */
/* ScanImageForGoodFeatures (goodFeaturesUtils.pyx:35-73) + sort (:234-236) + _enforceMinimumDistance
 * (:45-135).  mode KLT_REPLACING_SOME keeps features with val >= 0.  If use_pyramid != 0 the
 * level-0 image/gradients of the slot's pyramids are reused (selectGoodFeatures.py:176-181),
 * otherwise the slot's raw frame is smoothed/differentiated afresh (:183-197). */
int klt_select_async(klt_ctx *ctx, int slot, int mode, int use_pyramid, int fb, int n);
/* klt_select_async in two halves.  klt_select_begin_async enqueues everything up to the point where the host has to look at the
 * outcome (how many passes the minimum-distance stage needed, whether the candidate cut held); klt_select_finish waits, looks and --
 * rarely -- enqueues more passes or the repeat with every candidate.  Between the two the caller may enqueue other work that does not
 * WRITE the list or select again: the next frame's upload, build and klt_select_prepare_async (a sequence overlaps the host side of
 * frame t+1 with the GPU side of frame t's replacement this way) -- and work that only READS the list, such as the tracker launch of the
 * next frame into another buffer (klt_track_async: the GPU then has the tracker queued while the host looks at the outcome and enqueues
 * the next selection, instead of idling through that turn-around).  klt_select_finish returns KLT_OK, or 1 (not an error) when it had to
 * rewrite the list after the launches of klt_select_begin_async had run (more passes, or the repeat with every candidate): whatever
 * read the list in between must then be enqueued again.  klt_select_finish without a pending selection returns KLT_OK. */
int klt_select_begin_async(klt_ctx *ctx, int slot, int mode, int use_pyramid, int fb, int n);
int klt_select_finish(klt_ctx *ctx);
/* The list-independent half of a later klt_select_async(slot, KLT_REPLACING_SOME, use_pyramid = 1, ...): the summed-area tables and
 * the eigenvalue of every candidate window (goodFeaturesUtils.pyx:17-73, called from selectGoodFeatures.py:199-232) of the slot's
 * level-0 gradients, kept with the slot's contents (KLT_OPT_SCORE_SETS sets per context; a set follows klt_swap_slots, is used by one
 * selection, and dies with the next build
 * of the slot or a change of the selection parameters).  With KLT_OPT_BUILD_STREAM it runs on the build stream, i.e. while the main
 * stream tracks into the previous frame and replaces its lost features; the selection then only masks the live features'
 * squares, cuts and runs the minimum-distance passes.  Same result with or without this call. */
int klt_select_prepare_async(klt_ctx *ctx, int slot);
int klt_select(klt_ctx *ctx, int slot, int mode, int use_pyramid, klt_feat *inout, int n, int *n_placed);
/* replaces _enforceMinimumDistance(pointlist, featurelist, ncols, nrows, mindist, min_eigenvalue, overwriteAllFeatures) called on its own
 * (selectGoodFeatures.py:45-135): the greedy walk over a GIVEN candidate list in the GIVEN order.  keys[i] = f32 bits of val << 32 |
 * x << 16 | y (what klt_download_sorted_candidates returns, re-packed); the caller has dropped the candidates the walk skips without
 * effect (val < min_eigenvalue; positions inside the (2 mindist - 1)-squares of live features when overwrite_all == 0).  Free slots of
 * `inout` are filled in list order -- every slot by rank when overwrite_all, the lost ones (val < 0) otherwise; with overwrite_all the
 * slots the candidates did not reach become (-1, -1, KLT_NOT_FOUND).  Every candidate must lie inside the image (x < ncols, y < nrows;
 * KLT_ERR_ARG otherwise, checked here: the reference asserts the same, :90-91) and no key may be zero (value 0.0 at (0, 0): the walk's own
 * end mark; the reference never accepts a value below 1, :95).  Synchronous; needs no parameters and no frame. */
int klt_min_distance_walk(klt_ctx *ctx, const uint64_t *keys, int nkeys, int ncols, int nrows, int mindist, int overwrite_all,
                          klt_feat *inout, int n, int *n_placed);

/* ---- tracking: KLTTrackFeatures, trackFeatures.py:205-409 (translation model) -------------- */
/* _trackFeature (:67-136) + trackFeatureIterateCKLT (trackFeaturesUtils.pyx:393-459) for every
 * live feature, coarse to fine, one wavefront per feature.  Pyramids of both slots must be built. */
int klt_track_async(klt_ctx *ctx, int slot1, int slot2, int fb_in, int fb_out, int n);
int klt_track(klt_ctx *ctx, int slot1, int slot2, klt_feat *inout, int n, int *n_tracked);
/* npairs independent frame pairs of equal size in ONE tracker launch (BASELINE cfg-4: a shard of the 256 pairs);
 * pair i tracks feature buffer fb_in[i] (n records) from slot1[i] to slot2[i] into fb_out[i] */
int klt_track_batch_async(klt_ctx *ctx, const int *slot1, const int *slot2, const int *fb_in, const int *fb_out,
                          int npairs, int n);

/* ---- affine consistency check (BASELINE cfg-3) -- PARITY UNPINNED ------------------------- */
/* The reference calls _am_trackFeatureAffine / _am_getSubFloatImage at trackFeatures.py:347-399 but defines neither
 * (NameError): the call site pins the interface, upstream KLT 1.3.4 the behaviour.
 * mode = tc.affineConsistencyCheck (klt.py:67): -1 off, 0 translation, 1 similarity, 2 affine. */
typedef struct {
    int32_t mode, window_width, window_height, max_iterations;       /* klt.py:67-70 */
    float   max_residue, min_displacement, max_displacement_differ;   /* klt.py:71-73 */
} klt_affine_params;
/* per-feature state the reference keeps in KLT_Feature (klt.py:255-263); the three (window+2)^2 templates live in a
 * device array next to these records */
typedef struct { float aff_x, aff_y, Axx, Ayx, Axy, Ayy; int32_t valid, pad /* Newton iterations of the feature's last check */; } klt_affine_rec;
int klt_set_affine_params(klt_ctx *ctx, const klt_affine_params *p);
int klt_affine_alloc(klt_ctx *ctx, int state, int n);          /* n feature slots, no templates yet */
int klt_affine_download(klt_ctx *ctx, int state, klt_affine_rec *dst, int n);
/* device-side snapshot: the records (and, with_templates != 0, the templates) of the first n features of state `src` into `dst`
 * (allocated if needed), on the context's stream.  The templates of a feature never change while its record is valid. */
int klt_affine_copy_async(klt_ctx *ctx, int dst, int src, int n, int with_templates);
int klt_affine_free(klt_ctx *ctx, int state);                  /* releases records + templates; the id can be allocated again */
/* klt_track_async followed by the consistency check of the features it tracked; fb_in != fb_out.  `state` travels
 * with the feature list (same index = same feature). */
int klt_track_affine_async(klt_ctx *ctx, int slot1, int slot2, int fb_in, int fb_out, int n, int state);
int klt_track_affine(klt_ctx *ctx, int slot1, int slot2, klt_feat *inout, int n, int state, int *n_tracked);

typedef struct {
    uint64_t features;                       /* live features entering the kernel */
    uint64_t level_visits[KLT_MAX_LEVELS];   /* _trackFeature calls per pyramid level */
    uint64_t iterations[KLT_MAX_LEVELS];     /* Newton iterations per pyramid level */
} klt_track_stats;
/* reset zeroes the counters and starts collecting (a small reduction kernel after every tracker launch);
 * read returns the totals since the reset and stops collecting.  klt_feat.aux of a tracked record holds
 * 4 bits per level: 0 = level not visited, v = v-1 Newton iterations (saturating at 14). */
int klt_track_stats_reset(klt_ctx *ctx);
int klt_track_stats_read(klt_ctx *ctx, klt_track_stats *out);

/* ---- multi-GPU: one process per GPU, RCCL over xGMI (BASELINE cfg-4) --------------------- */
/* The reference has no counterpart (single process, single thread; its only concurrency is the GUI's multiprocessing
 * queue, examplegui.py:22-29).  The path shards by frame pair, so ranks never exchange pixels: the only collective is
 * the gather of 16-byte feature records at the end of a shard.  librccl is opened on first use (dlopen); a process that
 * never calls these entry points never loads it.  Collectives run on a side stream of the context, event-ordered behind
 * everything enqueued on the context's stream when they are issued; the context's stream itself never waits for RCCL
 * unless klt_comm_fence_async is called. */
#define KLT_COMM_ID_BYTES 128
/* rank 0: a fresh RCCL unique id (ncclGetUniqueId); the caller hands the 128 bytes to the other ranks (file, socket, env) */
int klt_comm_unique_id(void *out128);
/* every rank, same id: joins the communicator of `nranks` processes (ncclCommInitRank on the context's device) */
int klt_comm_init_rank(klt_ctx *ctx, int nranks, int rank, const void *unique_id);
int klt_comm_destroy(klt_ctx *ctx);
int klt_comm_info(klt_ctx *ctx, int *nranks, int *rank);       /* 1 / 0 when the context has no communicator */
/* all-gather / gather of the first n records of feature buffer fb_src into fb_dst (nranks * n records in rank order;
 * allocated here if needed; for the gather only `root` needs / gets fb_dst, other ranks may pass -1).  n may cover a
 * whole device-side [pairs x features] table (klt_featbuf_view). */
int klt_allgather_featbuf_async(klt_ctx *ctx, int fb_src, int fb_dst, int n);
int klt_gather_featbuf_async(klt_ctx *ctx, int fb_src, int fb_dst, int n, int root);
/* gather with a count per rank (shards of unequal size: 7 or 257 pairs over 8 GPUs): rank r contributes the first counts[r]
 * records of its fb_src, the root's fb_dst receives sum(counts) records back to back in rank order.  Every rank passes the same
 * table of nranks counts; a rank with count 0 takes no part in the transfer (its fb_src may be -1). */
int klt_gatherv_featbuf_async(klt_ctx *ctx, int fb_src, int fb_dst, const int *counts, int root);
/* The feature list as the baton of ONE temporal sequence cut into blocks of frames, one block per GPU (the tracker and the
 * replacement pass are a serial chain, trackFeatures.py:250-346 / selectGoodFeatures.py:45-135; everything that depends on the pixels
 * only is prepared by the block's owner meanwhile): n records of fb_send go to rank `to` and / or n records from rank `from` arrive in
 * fb_recv (-1: no such side; to == from == own rank: a device copy), on the side stream behind everything enqueued on the context's
 * stream so far; the context's stream then waits for the arrival.  fb_send must stay untouched until klt_comm_fence_featbuf_async. */
int klt_sendrecv_featbuf_async(klt_ctx *ctx, int fb_send, int to, int fb_recv, int from, int n);
int klt_comm_fence_async(klt_ctx *ctx);    /* the context's stream waits (on the device) for the collectives issued so far */
/* ... only for the last collective that read or wrote feature buffer fb (call it before overwriting a table whose
 * gather may still be in flight; collectives on other tables keep overlapping) */
int klt_comm_fence_featbuf_async(klt_ctx *ctx, int fb);
int klt_comm_wait(klt_ctx *ctx);           /* the host waits for them -- at most the communicator's timeout, then KLT_ERR_TIMEOUT */
/* Host-side waits on the communicator (klt_comm_wait, klt_sync, klt_comm_allreduce_max, and the waits before device memory that a
 * collective may touch is freed) give up after `ms` milliseconds with KLT_ERR_TIMEOUT instead of hanging when a peer has died;
 * default 300 000, or KLT_COMM_TIMEOUT_MS in the environment; <= 0 waits for ever.  After a timeout the communicator is unusable:
 * the rank should report and exit non-zero (never restart in place a process that has touched the GPU). */
int klt_comm_set_timeout(klt_ctx *ctx, double ms);
/* element-wise maximum over all ranks of n <= 16 doubles (host in, host out; synchronous -- also the barrier the
 * benchmark brackets its timed region with) */
int klt_comm_allreduce_max(klt_ctx *ctx, double *inout, int n);

/* ---- test / inspection hooks --------------------------------------------------------------- */
/* pyramid: 0 = image, 1 = gradx, 2 = grady; dst holds level_ncols*level_nrows floats.  (On the device the two gradient planes of a level are
 * stored interleaved -- gradx, grady of a pixel side by side, DESIGN.md section 4 --; this call and klt_download_select_f32 hand out separate planes.) */
int klt_level_dims(klt_ctx *ctx, int slot, int level, int *ncols, int *nrows);
int klt_download_f32(klt_ctx *ctx, int slot, int pyramid, int level, float *dst);
/* selection intermediates of the last klt_select*: 0 = smoothed image, 1 = gradx, 2 = grady (full frame),
 * 3 = eigenvalue map [ny][nx] (scan order, goodFeaturesUtils.pyx:53-54; after a KLT_REPLACING_SOME selection the pixels inside
 * the exclusion squares of the live features read 0: they can never be placed and are not scored; a selection that used the
 * scores of klt_select_prepare_async writes no map, and 3 is refused after it).  dims via klt_select_dims. */
int klt_select_dims(klt_ctx *ctx, int what, int *ncols, int *nrows);
int klt_download_select_f32(klt_ctx *ctx, int what, float *dst);
/* the NEXT klt_select* takes `count` = nx*ny eigenvalues from `val` (scan order, as what = 3 above returns them) instead
 * of computing them: lets the tests drive _enforceMinimumDistance (selectGoodFeatures.py:45-135) with equal scores and
 * long dependency chains that real frames rarely contain.  The override is consumed by one selection. */
int klt_set_score_override(klt_ctx *ctx, const float *val, int count);
/* first `n` sorted candidates of the last klt_select* as (val, x, y), selectGoodFeatures.py:234-236 */
int klt_download_sorted_candidates(klt_ctx *ctx, float *val, int32_t *x, int32_t *y, int n, int *n_valid);

/* ---- standalone convolutions (host buffers in / out, synchronous) ---------------------------- */
/* _convolveSeparate(imgin, horiz_kernel, vert_kernel), convolve.py:208-219 (SciPy branch: scipy.ndimage.convolve1d along axis 1, then
 * along axis 0; FP64 accumulation in correlate1d's operation order, f32 after each pass, `reflect` borders): the reference's one general
 * separable convolution -- ANY two tap lists of 1 .. 71 taps, odd or even (an even count shifts the window as convolve1d does), symmetric,
 * antisymmetric or neither.  f32 image in, f32 image out. */
int klt_convolve_separate_f32(klt_ctx *ctx, const float *src, int ncols, int nrows, const double *horiz, int nh, const double *vert, int nv,
                              float *dst);
/* KLTComputeSmoothedImage, convolve.py:254-264: _convolveSeparate(img, gauss, gauss) */
int klt_smooth_f32(klt_ctx *ctx, const float *src, int ncols, int nrows, const double *gauss, int ng, float *dst);
/* KLTComputeGradients, convolve.py:226-248: gradx = (deriv, gauss), grady = (gauss, deriv) */
int klt_gradients_f32(klt_ctx *ctx, const float *src, int ncols, int nrows, const double *gauss, int ng,
                      const double *deriv, int nd, float *gradx, float *grady);

/* KLTPyramid.Compute, pyramid.py:37-77: level 0 = src unchanged; level i = level i-1 smoothed with `gauss` (the taps of sigma =
 * subsampling * sigma_fact, convolve.py:27-93) and sampled at (ss y + ss/2, ss x + ss/2), dimensions int(n / ss).  dst receives
 * levels 1 .. nlevels-1 back to back (nothing for nlevels = 1); all levels stay on the device until the one download. */
int klt_pyramid_f32(klt_ctx *ctx, const float *src, int ncols, int nrows, int nlevels, int subsampling, const double *gauss, int ng,
                    float *dst);

/* ---- the reference's literal native boundary (host arrays in / out, synchronous) --------------- */
/* The only FFI the reference really has is its Cython `def` layer (setup.py:8-9).  These three are its entry points on the hot
 * path, bound by pyfeaturetrack_amd/compat/goodFeaturesUtils.py and trackFeaturesUtils.py under the reference's module and
 * function names.  Every call uploads the planes it is given: a boundary for drop-in use and for parity checks at the reference's
 * own granularity, not the fast path (that is klt_select* / klt_track*). */
/* ScanImageForGoodFeatures, goodFeaturesUtils.pyx:35-73: summed-area tables of gradx^2, gradx grady, grady^2 (f32 sequential scans,
 * :49-51) and the minimum eigenvalue of every candidate window (:17-31), candidates at x = borderx, borderx + step, ... < ncols - borderx
 * (the same in y), step = nSkippedPixels + 1.  val receives nx * ny eigenvalues in scan order (y outer, x inner: the order of the
 * reference's three lists); *nx, *ny the grid.  Arguments are the C ints Cython truncates the Python floats to (30.0 -> 30, 3.5 -> 3). */
int klt_scan_good_features_f32(klt_ctx *ctx, const float *gradx, const float *grady, int ncols, int nrows, int borderx, int bordery,
                               int window_hw, int window_hh, int nSkippedPixels, float *val, int val_cap, int *nx, int *ny);
/* extractImagePatchSlow, trackFeaturesUtils.pyx:14-51: width x height bilinear samples around (x, y), weights in FP64.
 * Square patches only (the reference swaps the roles of rows and columns, :38-49); a footprint that leaves the image is KLT_ERR_ARG
 * (the reference asserts, :35). */
int klt_extract_patch_f32(klt_ctx *ctx, const float *img, int ncols, int nrows, float x, float y, int width, int height, float *patch);
/* trackFeatureIterateCKLT, trackFeaturesUtils.pyx:393-459: the Newton loop of one feature at one pyramid level on template patches
 * the caller extracted; returns the position, the status (KLT_TRACKED / KLT_OOB / KLT_SMALL_DET) and the iteration count.  The tests
 * after the loop and the status priority are _trackFeature's (trackFeatures.py:67-136), i.e. the caller's.  Square windows only
 * (the reference's jacobian is strided by shape[0], :128). */
int klt_track_iterate_f32(klt_ctx *ctx, float x2, float y2, const float *gradx_patch, const float *grady_patch, const float *img_patch,
                          int width, int height, const float *img2, const float *gradx2, const float *grady2, int ncols, int nrows,
                          float step_factor, float min_determinant, float min_displacement, int max_iterations,
                          float *x2_out, float *y2_out, int *status, int *iterations);

/* ---- per-kernel timing (HIP events on the context's stream) -------------------------------- */
typedef struct { char name[32]; uint32_t launches; float total_ms; double bytes; } klt_kernel_time;
int klt_timing_enable(klt_ctx *ctx, int on);                /* resets the accumulated figures.  1: an event pair around every launch;
                                                             * 2: the same, but the kernels that are one launch per call (smooth_grad_l0, pyramid_reduce, gradients, track,
                                                             * affine_check, sat_rows, sat_cols, eigen_keys) are timed by their dispatch's own begin / end timestamps
                                                             * (hipExtLaunchKernelGGL events) -- the kernel duration a profiler reports, without the boundary between
                                                             * two dependent launches that an event pair also holds */
int klt_timing_read(klt_ctx *ctx, klt_kernel_time *out, int max_entries);   /* returns the entry count.  Mode 2: a launch of a stamped family that took
                                                                              * a path without dispatch timestamps (a fallback kernel) is not measured; such
                                                                              * launches are counted in an extra entry "<family>!unstamped" (no time) */

#ifdef __cplusplus
}
#endif
#endif /* KLT_GPU_H */
