// api_featbuf.hip -- feature buffers of a context (numbered lists of 16-byte klt_feat records): copies either way, with and without a
// host wait, views into tables, buffers that ARE pinned host memory.
#include "klt_context.h"

extern "C" int klt_comm_fence_async(klt_ctx *c);        // api_comm.hip: a gathered table is complete before it is read back

namespace {
// a view may be pinned host memory (klt_featbuf_map_host): the copy's direction is then left to the runtime, which knows both pointers
inline hipMemcpyKind up(const FeatBuf &b) { return b.view ? hipMemcpyDefault : hipMemcpyHostToDevice; }
inline hipMemcpyKind down(const FeatBuf &b) { return b.view ? hipMemcpyDefault : hipMemcpyDeviceToHost; }
}  // namespace

extern "C" {

int klt_featbuf_upload(klt_ctx *c, int fb, const klt_feat *src, int n)
{
    if (!c || !src || n < 0) return fail(c, KLT_ERR_ARG, "bad argument");
    HIPCHK(c, hipSetDevice(c->device));
    FeatBuf *b;
    if (int rc = get_fb(c, fb, n > 0 ? n : 1, &b)) return rc;
    HIPCHK(c, hipMemcpyAsync(b->d, src, (size_t)n * sizeof(klt_feat), up(*b), c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return KLT_OK;
}

int klt_featbuf_upload_async(klt_ctx *c, int fb, const klt_feat *src, int n)
{
    if (!c || !src || n < 0) return fail(c, KLT_ERR_ARG, "bad argument");
    HIPCHK(c, hipSetDevice(c->device));
    FeatBuf *b;
    if (int rc = get_fb(c, fb, n > 0 ? n : 1, &b)) return rc;
    HIPCHK(c, hipMemcpyAsync(b->d, src, (size_t)n * sizeof(klt_feat), up(*b), c->stream));
    return KLT_OK;
}

int klt_featbuf_download(klt_ctx *c, int fb, klt_feat *dst, int n)
{
    if (!c || !dst || n < 0) return fail(c, KLT_ERR_ARG, "bad argument");
    if (fb < 0 || (size_t)fb >= c->fbs.size() || c->fbs[fb].cap < n) return fail(c, KLT_ERR_STATE, "feature buffer not that large");
    HIPCHK(c, hipSetDevice(c->device));
    if (int rc = klt_comm_fence_async(c)) return rc;        // a gathered table is complete before it is read back
    HIPCHK(c, hipMemcpyAsync(dst, c->fbs[fb].d, (size_t)n * sizeof(klt_feat), down(c->fbs[fb]), c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return KLT_OK;
}

// Records on their way to the host WITHOUT draining the pipeline: the copy is enqueued on the context's stream (stream order: behind the
// kernels that wrote the records, in front of whatever is enqueued next) and an event behind it is what klt_download_wait waits for.  A
// synchronous download at a window boundary makes the host wait for everything it has enqueued -- up to a millisecond of queued steps --
// and nothing new (uploads of the next frames included) is issued meanwhile: tools/trace_copies.py showed the link idle for 0.3-0.9 ms
// per 16 pairs.
int klt_featbuf_download_async(klt_ctx *c, int fb, klt_feat *dst, int n)
{
    if (!c || !dst || n < 0) return fail(c, KLT_ERR_ARG, "bad argument");
    if (fb < 0 || (size_t)fb >= c->fbs.size() || c->fbs[fb].cap < n) return fail(c, KLT_ERR_STATE, "feature buffer not that large");
    HIPCHK(c, hipSetDevice(c->device));
    hipPointerAttribute_t attr;                           // a pageable destination would be staged synchronously
    if (hipPointerGetAttributes(&attr, dst) != hipSuccess || attr.type != hipMemoryTypeHost) {
        (void)hipGetLastError();
        return fail(c, KLT_ERR_ARG, "klt_featbuf_download_async needs pinned host memory (klt_host_alloc)");
    }
    if (int rc = klt_comm_fence_async(c)) return rc;        // a gathered table is complete before it is read back
    HIPCHK(c, hipMemcpyAsync(dst, c->fbs[fb].d, (size_t)n * sizeof(klt_feat), down(c->fbs[fb]), c->stream));
    if (int rc = fresh_event(c, &c->ev_download, &c->download_serial)) return rc;
    HIPCHK(c, hipEventRecord(c->ev_download, c->stream));
    c->download_pending = true;
    return KLT_OK;
}

// The same event without a copy in front of it: "the host wants to wait for THIS point of the main stream" (klt_download_wait), not for
// the stream -- a caller whose records are written straight into pinned memory (klt_featbuf_map_host) marks the stream behind the
// kernel that writes them and may enqueue more work (the next selection's scores) before it waits.
int klt_download_mark_async(klt_ctx *c)
{
    if (!c) return KLT_ERR_ARG;
    HIPCHK(c, hipSetDevice(c->device));
    if (int rc = fresh_event(c, &c->ev_download, &c->download_serial)) return rc;
    HIPCHK(c, hipEventRecord(c->ev_download, c->stream));
    c->download_pending = true;
    return KLT_OK;
}

int klt_download_wait(klt_ctx *c)
{
    if (!c) return KLT_ERR_ARG;
    if (!c->download_pending) return KLT_OK;
    HIPCHK(c, hipSetDevice(c->device));
    if (!event_live(c, c->download_serial)) HIPCHK(c, hipStreamSynchronize(c->stream));     // (the ring has re-used the event since)
    else {
        hipError_t q = hipEventQuery(c->ev_download);       // poll: normally long complete
        for (long spins = 0; q == hipErrorNotReady; spins++) {
            (void)hipGetLastError();
            if (spins > 2000000) { HIPCHK(c, hipEventSynchronize(c->ev_download)); q = hipSuccess; break; }
            q = hipEventQuery(c->ev_download);
        }
        if (q != hipSuccess) return fail(c, KLT_ERR_DEVICE, std::string("hipEventQuery: ") + hipGetErrorString(q));
    }
    c->download_pending = false;
    return KLT_OK;
}

int klt_featbuf_alloc(klt_ctx *c, int fb, int n)
{
    if (!c || n <= 0) return fail(c, KLT_ERR_ARG, "bad argument");
    HIPCHK(c, hipSetDevice(c->device));
    FeatBuf *b;
    if (int rc = get_fb(c, fb, n, &b)) return rc;
    HIPCHK(c, hipMemsetAsync(b->d, 0xff, (size_t)n * sizeof(klt_feat), c->stream));     // val = -1 everywhere
    return KLT_OK;
}

int klt_featbuf_view(klt_ctx *c, int fb_view, int fb_parent, int offset, int n)
{
    if (!c || offset < 0 || n <= 0 || fb_view == fb_parent) return fail(c, KLT_ERR_ARG, "bad argument");
    if (fb_parent < 0 || (size_t)fb_parent >= c->fbs.size() || c->fbs[fb_parent].cap < offset + n || c->fbs[fb_parent].view)
        return fail(c, KLT_ERR_STATE, "parent feature buffer too small (or itself a view)");
    if (fb_view < 0 || fb_view > 65535) return fail(c, KLT_ERR_ARG, "feature buffer index out of range");
    if ((size_t)fb_view >= c->fbs.size()) c->fbs.resize(fb_view + 1);
    FeatBuf &v = c->fbs[fb_view];
    if (v.d && !v.view) { if (int rc = sync_all(c)) return rc; hipFree(v.d); }
    v.d = c->fbs[fb_parent].d + offset;
    v.cap = n;
    v.view = true;
    return KLT_OK;
}

// A feature buffer that IS pinned host memory: kernels read and write the caller's records over the link (16 bytes per feature -- 80 KB
// for 5000 features, a few microseconds), so a call that sends a list, tracks it and waits for the result needs no copy command in
// either direction (each one costs 8-15 us of queue latency on its own, a third of the tracker's run time at cfg-2's list length).
int klt_featbuf_map_host(klt_ctx *c, int fb, klt_feat *host, int n)
{
    if (!c || n < 0 || (host && n == 0)) return fail(c, KLT_ERR_ARG, "bad argument");
    if (fb < 0 || fb > 65535) return fail(c, KLT_ERR_ARG, "feature buffer index out of range");
    HIPCHK(c, hipSetDevice(c->device));
    void *dev = nullptr;
    if (host) {
        hipPointerAttribute_t attr;
        if (hipPointerGetAttributes(&attr, host) != hipSuccess || attr.type != hipMemoryTypeHost) {
            (void)hipGetLastError();
            return fail(c, KLT_ERR_ARG, "klt_featbuf_map_host needs pinned host memory (klt_host_alloc)");
        }
        HIPCHK(c, hipHostGetDevicePointer(&dev, host, 0));
    }
    if ((size_t)fb >= c->fbs.size()) c->fbs.resize(fb + 1);
    FeatBuf &v = c->fbs[fb];
    if (v.d) {                                             // queued work may still use what the buffer was so far
        if (int rc = sync_all(c)) return rc;
        if (!v.view) hipFree(v.d);
    }
    v.d = (klt_feat *)dev;
    v.cap = host ? n : 0;
    v.view = host != nullptr;
    return KLT_OK;
}

void *klt_featbuf_devptr(klt_ctx *c, int fb)
{
    if (!c || fb < 0 || (size_t)fb >= c->fbs.size()) return nullptr;
    return c->fbs[fb].d;
}


}  // extern "C"
