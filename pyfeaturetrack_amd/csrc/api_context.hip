// api_context.hip -- the context of the C ABI (include/klt_gpu.h): lifetime, parameters, options, the state of slots, and the helpers the
// other api_*.hip files lean on (errors, device buffers that grow, the ring of ordering events, cross-stream marks), timing read-back.
#include "klt_context.h"

thread_local hipEvent_t g_klt_stamp_start = nullptr, g_klt_stamp_stop = nullptr;      // klt_internal.h: timing by dispatch timestamps

namespace kltapi {

std::string g_create_error;

int fail(klt_ctx *c, int code, const std::string &msg)
{
    if (c) c->err = msg;
    return code;
}

int drain_timers(klt_ctx *c)
{
    if (c->pending.empty()) return 0;
    HIPCHK(c, hipStreamSynchronize(c->stream));
    if (c->bstream) HIPCHK(c, hipStreamSynchronize(c->bstream));
    for (Timed &t : c->pending) {
        float ms = 0.f;
        hipEventElapsedTime(&ms, t.a, t.b);
        c->acc_ms[t.fam] += ms;
        c->acc_bytes[t.fam] += t.bytes;
        c->acc_n[t.fam]++;
        c->pool.push_back(t.a);
        c->pool.push_back(t.b);
    }
    c->pending.clear();
    return 0;
}

// every stream idle -- the communicator's side stream included: a gather still in flight reads / writes feature tables -- : required
// before freeing anything a queued kernel or collective may still use
int sync_all(klt_ctx *c)
{
    // A communicator whose collective timed out: the main stream may be fenced behind the dead collective (klt_comm_fence_async, a
    // feature buffer's comm_done wait), so waiting for it here would hang the rank that is trying to report and leave.  Nothing is
    // freed or re-laid-out in that state -- the rank exits non-zero and is never restarted in place (klt_comm_set_timeout).
    if (c->comm && comm_poisoned(c->comm))
        return fail(c, KLT_ERR_TIMEOUT, "the communicator timed out earlier: device memory is left to process exit (klt_destroy does not wait either)");
    HIPCHK(c, hipStreamSynchronize(c->stream));
    if (c->cstream) HIPCHK(c, hipStreamSynchronize(c->cstream));
    for (hipStream_t x : c->cextra) if (x) HIPCHK(c, hipStreamSynchronize(x));
    if (c->bstream) HIPCHK(c, hipStreamSynchronize(c->bstream));
    if (c->comm) { std::string err; if (int rc = comm_wait(c->comm, err)) return fail(c, rc, err); }
    return 0;
}
int dev_alloc(klt_ctx *c, void **p, size_t bytes, const char *what)
{
    *p = nullptr;
    hipError_t e = hipErrorOutOfMemory;
    if (c->fail_alloc_in == 0) c->fail_alloc_in = -1;                 // the test hook: this allocation is refused as if memory had run out
    else {
        if (c->fail_alloc_in > 0) c->fail_alloc_in--;
        e = hipMalloc(p, bytes ? bytes : 1);
    }
    if (e == hipSuccess) return 0;
    *p = nullptr;
    (void)hipGetLastError();                                          // out of memory has been answered: nothing stale for the next launch check
    char msg[160];
    snprintf(msg, sizeof msg, "out of device memory: %zu bytes asked for (%s)%s%s", bytes, what, e == hipErrorOutOfMemory ? "" : ": ",
             e == hipErrorOutOfMemory ? "" : hipGetErrorString(e));
    return fail(c, e == hipErrorOutOfMemory ? KLT_ERR_NOMEM : KLT_ERR_DEVICE, msg);
}

int host_alloc(klt_ctx *c, void **p, size_t bytes, const char *what)
{
    *p = nullptr;
    hipError_t e = hipErrorOutOfMemory;
    if (c->fail_alloc_in == 0) c->fail_alloc_in = -1;
    else {
        if (c->fail_alloc_in > 0) c->fail_alloc_in--;
        e = hipHostMalloc(p, bytes ? bytes : 1, hipHostMallocDefault);
    }
    if (e == hipSuccess) return 0;
    *p = nullptr;
    (void)hipGetLastError();
    char msg[160];
    snprintf(msg, sizeof msg, "out of pinned host memory: %zu bytes asked for (%s)%s%s", bytes, what, e == hipErrorOutOfMemory ? "" : ": ",
             e == hipErrorOutOfMemory ? "" : hipGetErrorString(e));
    return fail(c, e == hipErrorOutOfMemory ? KLT_ERR_NOMEM : KLT_ERR_DEVICE, msg);
}

int ensure_tmp(klt_ctx *c, size_t pixels)
{
    if (pixels <= c->tmp_cap && c->tmpA) return 0;
    if (c->tmpA) { if (int rc = sync_all(c)) return rc; hipFree(c->tmpA); hipFree(c->tmpB); c->tmpA = c->tmpB = nullptr; c->tmp_cap = 0; }
    DEVALLOC(c, c->tmpA, pixels * sizeof(float));
    DEVALLOC(c, c->tmpB, pixels * sizeof(float));       // (tmpA alone with tmp_cap 0 is freed by the next call)
    c->tmp_cap = pixels;
    return 0;
}

int ensure_h1(klt_ctx *c, size_t floats)
{
    if (floats <= c->h1_cap && c->h1) return 0;
    if (c->h1) { if (int rc = sync_all(c)) return rc; hipFree(c->h1); c->h1 = nullptr; c->h1_cap = 0; }
    DEVALLOC(c, c->h1, floats * sizeof(float));
    c->h1_cap = floats;
    return 0;
}

int get_slot(klt_ctx *c, int slot, Slot **out, bool create)
{
    if (slot < 0 || slot > 65535) return fail(c, KLT_ERR_ARG, "slot index out of range");
    if ((size_t)slot >= c->slots.size()) {
        if (!create) return fail(c, KLT_ERR_STATE, "slot has no frame");
        c->slots.resize(slot + 1);
    }
    *out = &c->slots[slot];
    return 0;
}

int get_fb(klt_ctx *c, int fb, int n, FeatBuf **out)
{
    if (fb < 0 || fb > 65535) return fail(c, KLT_ERR_ARG, "feature buffer index out of range");
    if ((size_t)fb >= c->fbs.size()) c->fbs.resize(fb + 1);
    FeatBuf &b = c->fbs[fb];
    if (n > b.cap && b.view) return fail(c, KLT_ERR_ARG, "feature buffer is a view and too small");
    if (n > b.cap) {
        klt_feat *nd = nullptr;
        DEVALLOC(c, nd, (size_t)n * sizeof(klt_feat));
        if (b.d) {
            HIPCHK(c, hipMemcpyAsync(nd, b.d, (size_t)b.cap * sizeof(klt_feat), hipMemcpyDeviceToDevice, c->stream));
            if (int rc = sync_all(c)) return rc;
            hipFree(b.d);
        }
        b.d = nd;
        b.cap = n;
    }
    *out = &b;
    return 0;
}

// ordering events for the asynchronous ingest come from a ring (see klt_ctx::ring)
constexpr size_t kEventRing = 256;

// `serial` (optional) receives the running number of the hand-out: an event pointer kept by a slot is only meaningful while
// fewer than kEventRing events have been handed out since (event_live); after that the ring has re-recorded it elsewhere.
int fresh_event(klt_ctx *c, hipEvent_t *out, uint64_t *serial)
{
    if (serial) *serial = c->ring_serial;
    c->ring_serial++;
    if (c->ring.size() < kEventRing) {
        hipEvent_t e = nullptr;
        HIPCHK(c, hipEventCreateWithFlags(&e, hipEventDisableTiming));
        c->ring.push_back(e);
        *out = e;
        return 0;
    }
    *out = c->ring[c->ring_next];
    c->ring_next = (c->ring_next + 1) % kEventRing;
    return 0;
}

bool event_live(const klt_ctx *c, uint64_t serial) { return c->ring_serial - serial < kEventRing; }

// The frame copied by klt_upload_u8_async has landed before anything enqueued on `stream` after this call reads it.  A slot
// left alone for more than a ring's worth of events no longer owns its event: the host waits for the copy stream instead.
int wait_upload(klt_ctx *c, Slot *s, hipStream_t consumer)
{
    if (!s->upload_pending) return 0;
    if (c->capturing) { }                                   // (the capturing caller has waited for the copy on the host)
    else if (event_live(c, s->upload_serial)) HIPCHK(c, hipStreamWaitEvent(consumer, s->ev_upload, 0));
    else { HIPCHK(c, hipStreamSynchronize(c->cstream)); for (hipStream_t x : c->cextra) if (x) HIPCHK(c, hipStreamSynchronize(x)); }
    s->upload_pending = false;
    return 0;
}

// everything enqueued on `stream` so far has read the raw frames of these slots: the next asynchronous copy into them waits for it
int mark_consumed(klt_ctx *c, Slot *const *slots, int n, hipStream_t reader)
{
    if (!c->cstream) return 0;
    hipEvent_t e;
    uint64_t serial;
    if (int rc = fresh_event(c, &e, &serial)) return rc;
    HIPCHK(c, hipEventRecord(e, reader));
    for (int i = 0; i < n; i++) { slots[i]->ev_consumed = e; slots[i]->consumed_serial = serial; slots[i]->consumed_valid = true; }
    return 0;
}

// the tracker launch just enqueued on the main stream reads the pyramids of these slots: the next build of any of them on the build
// stream waits for it (and for nothing else on the main stream)
int mark_read(klt_ctx *c, Slot *const *slots, int n)
{
    if (!c->build_stream_on) { for (int i = 0; i < n; i++) slots[i]->read_valid = false; return 0; }    // main-stream builds are in order
    hipEvent_t e;
    uint64_t serial;
    if (int rc = fresh_event(c, &e, &serial)) return rc;
    HIPCHK(c, hipEventRecord(e, c->stream));
    for (int i = 0; i < n; i++) { slots[i]->ev_read = e; slots[i]->read_serial = serial; slots[i]->read_valid = true; }
    return 0;
}

// KLT_OPT_BUILD_STREAM: work on the main stream that reads (or overwrites) a slot's frame / pyramids waits for the build that is
// filling them on the build stream.  One wait per build: later main-stream work is ordered behind the first.
int wait_built(klt_ctx *c, Slot *s)
{
    if (!s->built_pending) return 0;
    // the slots of a batched build share its event: the main stream waits for it once (64 slots = 64 barrier packets otherwise, a few
    // microseconds of queue time each)
    if (s->built_serial != c->waited_built_serial) {
        if (event_live(c, s->built_serial)) HIPCHK(c, hipStreamWaitEvent(c->stream, s->ev_built, 0));
        else if (!c->capturing) HIPCHK(c, hipStreamSynchronize(c->bstream));
        c->waited_built_serial = s->built_serial;
    }
    s->built_pending = false;
    return 0;
}
int check_ready(klt_ctx *c)
{
    if (!c) return KLT_ERR_ARG;
    if (!c->have_params) return fail(c, KLT_ERR_STATE, "klt_set_params has not been called");
    for (int i = 0; i < 3; i++)
        if (!c->have_taps[i]) return fail(c, KLT_ERR_STATE, "klt_set_kernels has not been called for all three tap sets");
    return 0;
}


}  // namespace kltapi

extern "C" {

int klt_abi_version(void) { return KLT_ABI_VERSION; }

int klt_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

int klt_create(int device, klt_ctx **out)
{
    if (!out) return KLT_ERR_ARG;
    *out = nullptr;
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0) {
        g_create_error = std::string("no HIP device available (") + hipGetErrorString(e) + "); libkltgpu has no CPU path";
        return KLT_ERR_DEVICE;
    }
    if (device < 0 || device >= n) {
        g_create_error = "device index out of range";
        return KLT_ERR_ARG;
    }
    klt_ctx *c = new klt_ctx();
    c->device = device;
    if ((e = hipSetDevice(device)) != hipSuccess || (e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking)) != hipSuccess ||
        (e = hipMalloc((void **)&c->stats_d, (1 + 2 * KLT_MAX_LEVELS) * sizeof(unsigned long long))) != hipSuccess ||
        (e = hipMalloc((void **)&c->placed_d, 4 * sizeof(int))) != hipSuccess ||
        (e = hipMemset(c->stats_d, 0, (1 + 2 * KLT_MAX_LEVELS) * sizeof(unsigned long long))) != hipSuccess) {
        g_create_error = std::string("device setup failed: ") + hipGetErrorString(e);
        delete c;
        return KLT_ERR_DEVICE;
    }
    c->work = c->stream;
    if (const char *v = getenv("KLT_FUSED_HREDUCE")) c->fuse_hreduce = atoi(v) != 0;      // experiment hook (initial value of the option)
    if (const char *v = getenv("KLT_TRACK_XCD_ORDER")) c->track_xcd_order = atoi(v) != 0;
    if (const char *v = getenv("KLT_COPY_STREAMS")) { const int k = atoi(v); c->ncopy = k < 1 ? 1 : (k > klt_ctx::kMaxCopyStreams ? klt_ctx::kMaxCopyStreams : k); }
    *out = c;
    return KLT_OK;
}

void klt_destroy(klt_ctx *c)
{
    if (!c) return;
    hipSetDevice(c->device);
    if (c->comm && comm_poisoned(c->comm)) {
        // A collective of this context can never complete (klt_comm_wait timed out: a peer is gone).  The main stream may be fenced
        // behind it, and hipStreamSynchronize / hipFree (which synchronises the device) would hang the rank that is trying to report and
        // exit non-zero.  Abort the communicator and leave the device memory to the process's end.
        comm_destroy(c->comm);
        c->comm = nullptr;
        delete c;
        return;
    }
    if (c->stream) hipStreamSynchronize(c->stream);
    if (c->cstream) { hipStreamSynchronize(c->cstream); hipStreamDestroy(c->cstream); }
    for (hipStream_t x : c->cextra) if (x) { hipStreamSynchronize(x); hipStreamDestroy(x); }
    if (c->bstream) { hipStreamSynchronize(c->bstream); hipStreamDestroy(c->bstream); }
    if (c->comm) { comm_destroy(c->comm); c->comm = nullptr; }
    for (void *p : c->pinned) hipHostFree(p);
    for (void *p : c->dev_allocs) hipFree(p);
    for (hipEvent_t e : c->ring) hipEventDestroy(e);
    if (c->ev_sel) hipEventDestroy(c->ev_sel);
    for (Slot &s : c->slots) { hipFree(s.u8); hipFree(s.u8_alt); hipFree(s.f32); hipFree(s.planes); }
    for (FeatBuf &b : c->fbs)
        if (!b.view) hipFree(b.d);
    hipFree(c->tmpA); hipFree(c->tmpB); hipFree(c->h1);
    hipFree(c->sel_img); hipFree(c->sel_gx); hipFree(c->sat); hipFree(c->valmap);      // (sel_gy points into sel_gx's allocation)
    for (auto &e : c->pre) hipFree(e.keys);
    hipFree(c->sat_pre);
    hipFree(c->keys); hipFree(c->seedmap); hipFree(c->grid); hipFree(c->nms_slots); for (auto &bt : c->batch_tables) hipFree(bt.dev); for (auto &bo : c->batch_orders) hipFree(bo.order); hipFree(c->shared_order.order); hipFree(c->keys2); hipFree(c->topk_hist); hipFree(c->fl_snapshot); hipFree(c->mis_st); hipFree(c->mis_list); hipFree(c->mis_cnt); hipFree(c->score_override); hipFree(c->mis_tile_keys);
    for (AffState &a : c->aff) { hipFree(a.rec); hipFree(a.tpl); } hipFree(c->placed_d); hipFree(c->stats_d);
    for (Timed &t : c->pending) { hipEventDestroy(t.a); hipEventDestroy(t.b); }
    for (hipEvent_t e : c->pool) hipEventDestroy(e);
    if (c->stream) hipStreamDestroy(c->stream);
    delete c;
}

const char *klt_last_error(klt_ctx *c) { return c ? c->err.c_str() : g_create_error.c_str(); }

int klt_sync(klt_ctx *c)
{
    if (!c) return KLT_ERR_ARG;
    if (c->cstream) HIPCHK(c, hipStreamSynchronize(c->cstream));
    for (hipStream_t x : c->cextra) if (x) HIPCHK(c, hipStreamSynchronize(x));
    if (c->bstream) HIPCHK(c, hipStreamSynchronize(c->bstream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    if (c->comm) { std::string err; if (int rc = comm_wait(c->comm, err)) return fail(c, rc, err); }
    return KLT_OK;
}

void *klt_stream_handle(klt_ctx *c) { return c ? (void *)c->stream : nullptr; }

int klt_set_params(klt_ctx *c, const klt_params *p)
{
    if (!c || !p) return fail(c, KLT_ERR_ARG, "null argument");
    if (p->window_width != p->window_height || (p->window_width & 1) == 0 || p->window_width < 3 || p->window_width > 31)
        return fail(c, KLT_ERR_ARG, "window must be square, odd and between 3 and 31");
    if (p->nPyramidLevels < 1 || p->nPyramidLevels > KLT_MAX_LEVELS) return fail(c, KLT_ERR_ARG, "nPyramidLevels out of range");
    const int ss = p->subsampling;
    if (p->nPyramidLevels > 1 && ss != 2 && ss != 4 && ss != 8 && ss != 16 && ss != 32)
        return fail(c, KLT_ERR_ARG, "subsampling must be 2, 4, 8, 16 or 32");      // pyramid.py:17-20
    if (p->nSkippedPixels < 0 || p->max_iterations < 0) return fail(c, KLT_ERR_ARG, "negative count");
    const bool relayout = !c->have_params || c->p.nPyramidLevels != p->nPyramidLevels || c->p.subsampling != ss;
    c->p = *p;
    if (c->p.nPyramidLevels == 1 && (ss < 2)) c->p.subsampling = 2;
    c->have_params = true;
    if (relayout)
        for (Slot &s : c->slots) s.pyr_valid = false;
    return KLT_OK;
}


int klt_set_option(klt_ctx *c, int option, int value)
{
    if (!c) return KLT_ERR_ARG;
    if (option == KLT_OPT_FUSED_KERNELS) { c->use_fused = value != 0; return KLT_OK; }
    if (option == KLT_OPT_SAT_VARIANT) { c->sat_variant = value; return KLT_OK; }
    if (option == KLT_OPT_TRACK_VARIANT) { g_track_variant = value; return KLT_OK; }
    if (option == KLT_OPT_FUSED_HREDUCE) { c->fuse_hreduce = value != 0; return KLT_OK; }
    if (option == KLT_OPT_TRACK_XCD_ORDER) { c->track_xcd_order = value != 0; return KLT_OK; }
    if (option == KLT_OPT_BUILD_STREAM) {
        if (!value && c->bstream) HIPCHK(c, hipStreamSynchronize(c->bstream));      // pending builds finish; their events stay valid
        if (c->build_stream_on != (value != 0)) c->last_build_on_bstream = -1;   // trackers launched meanwhile carry no read marks: the next
        c->build_stream_on = value != 0;                                          // build over there waits for the whole main stream
        return KLT_OK;
    }
    if (option == KLT_OPT_SCORE_SETS) {
        if (value < 2 || value > 256) return fail(c, KLT_ERR_ARG, "KLT_OPT_SCORE_SETS takes 2..256");
        if (c->sel_job) return fail(c, KLT_ERR_STATE, "a selection is pending (it may hold one of the score sets): klt_select_finish first");
        if (int rc = sync_all(c)) return rc;
        for (size_t i = (size_t)value; i < c->pre.size(); i++) hipFree(c->pre[i].keys);
        c->pre.resize((size_t)value);
        return KLT_OK;
    }
    if (option == KLT_OPT_COPY_STREAMS) {
        if (value < 1 || value > klt_ctx::kMaxCopyStreams) return fail(c, KLT_ERR_ARG, "KLT_OPT_COPY_STREAMS takes 1..8");
        if (c->cstream) HIPCHK(c, hipStreamSynchronize(c->cstream));          // uploads in flight finish on the streams they were given
        for (hipStream_t x : c->cextra) if (x) HIPCHK(c, hipStreamSynchronize(x));
        c->ncopy = value;
        return KLT_OK;
    }
    if (option == KLT_OPT_FAIL_ALLOC_AFTER) { c->fail_alloc_in = value; return KLT_OK; }
    if (option == KLT_OPT_TRACK_TREE_SUMS) { c->track_tree_sums = value != 0; return KLT_OK; }
    if (option == KLT_OPT_TOPK_PREFILTER) { c->use_topk = value != 0; return KLT_OK; }
    if (option == KLT_OPT_SELECT_PARALLEL_NMS) { c->use_mis = value != 0; return KLT_OK; }
    if (option == KLT_OPT_SELECT_AFFINE_STATE) {
        if (value >= 0 && ((size_t)value >= c->aff.size() || !c->aff[value].rec)) return fail(c, KLT_ERR_STATE, "affine state not allocated");
        c->select_aff_state = value;
        return KLT_OK;
    }
    return fail(c, KLT_ERR_ARG, "unknown option");
}


int klt_slot_state(klt_ctx *c, int slot)
{
    if (!c) return KLT_ERR_ARG;
    if (slot < 0 || (size_t)slot >= c->slots.size()) return 0;
    const Slot &s = c->slots[slot];
    return (s.raw_kind != 0 ? 1 : 0) | (s.pyr_valid ? 2 : 0);
}

int klt_device_memory(klt_ctx *c, size_t *free_bytes, size_t *total_bytes)
{
    if (!c) return KLT_ERR_ARG;
    HIPCHK(c, hipSetDevice(c->device));
    size_t f = 0, t = 0;
    HIPCHK(c, hipMemGetInfo(&f, &t));
    if (free_bytes) *free_bytes = f;
    if (total_bytes) *total_bytes = t;
    return KLT_OK;
}

int klt_slot_generation(klt_ctx *c, int slot, uint64_t *gen)
{
    if (!c || !gen) return KLT_ERR_ARG;
    *gen = (slot >= 0 && (size_t)slot < c->slots.size() && c->slots[slot].pyr_valid) ? c->slots[slot].gen : 0;
    return KLT_OK;
}

int klt_slot_free(klt_ctx *c, int slot)
{
    if (!c) return KLT_ERR_ARG;
    if (slot < 0 || (size_t)slot >= c->slots.size()) return KLT_OK;
    HIPCHK(c, hipSetDevice(c->device));
    if (int rc = sync_all(c)) return rc;
    Slot &s = c->slots[slot];
    hipFree(s.u8); hipFree(s.u8_alt); hipFree(s.f32); hipFree(s.planes);
    s = Slot();
    return KLT_OK;
}

int klt_swap_slots(klt_ctx *c, int a, int b)
{
    if (!c) return KLT_ERR_ARG;
    Slot *sa, *sb;
    const int hi = a > b ? a : b;
    if (hi >= 0 && (size_t)hi >= c->slots.size() && hi <= 65535) c->slots.resize(hi + 1);
    if (int rc = get_slot(c, a, &sa, true)) return rc;
    if (int rc = get_slot(c, b, &sb, true)) return rc;
    std::swap(*sa, *sb);
    return KLT_OK;
}


// ------------------------------------------------------------------------------------------ timing
int klt_timing_enable(klt_ctx *c, int on)
{
    if (!c) return KLT_ERR_ARG;
    if (int rc = drain_timers(c)) return rc;
    for (int f = 0; f < F_COUNT; f++) { c->acc_ms[f] = 0; c->acc_bytes[f] = 0; c->acc_n[f] = 0; c->acc_unstamped[f] = 0; }
    c->timing = on != 0;
    c->timing_stamps = on == 2;
    return KLT_OK;
}

int klt_timing_read(klt_ctx *c, klt_kernel_time *out, int max_entries)
{
    if (!c || !out) return fail(c, KLT_ERR_ARG, "null argument");
    if (int rc = drain_timers(c)) return rc;
    int k = 0;
    for (int f = 0; f < F_COUNT && k < max_entries; f++) {
        if (!c->acc_n[f]) continue;
        std::memset(&out[k], 0, sizeof(out[k]));
        std::snprintf(out[k].name, sizeof(out[k].name), "%s", kFamilyName[f]);
        out[k].launches = c->acc_n[f];
        out[k].total_ms = (float)c->acc_ms[f];
        out[k].bytes = c->acc_bytes[f];
        k++;
    }
    // mode 2: launches of a stamped family that took a path without dispatch timestamps are reported, not dropped: "<family>!unstamped"
    // carries their number (no time, no bytes) -- a reader must not quote the family's figures as covering every launch
    for (int f = 0; f < F_COUNT && k < max_entries; f++) {
        if (!c->acc_unstamped[f]) continue;
        std::memset(&out[k], 0, sizeof(out[k]));
        std::snprintf(out[k].name, sizeof(out[k].name), "%s!unstamped", kFamilyName[f]);
        out[k].launches = c->acc_unstamped[f];
        k++;
    }
    return k;
}


}  // extern "C"
