// api_frames.hip -- frames on their way in and the pyramids built from them: taps, uploads (synchronous, asynchronous from pinned memory on
// the copy streams, adoption of frames already in device memory), pinned / device memory for callers without a HIP binding, the launch
// sequence of the pyramid build (klt_build_pyramids*), the stand-alone convolutions and the read-back of planes.
#include "klt_context.h"

namespace kltapi {

void make_taps(const double *k, int n, Taps &t)
{
    // scipy.ndimage.convolve1d: weights[::-1], then correlate1d's symmetry test (|a -+ b| <= DBL_EPSILON)
    std::memset(&t, 0, sizeof(t));
    t.n = n;
    for (int i = 0; i < n; i++) t.k[i] = k[n - 1 - i];
    t.sym = 0;
    if (n & 1) {
        const int half = n / 2;
        t.sym = 1;
        for (int ii = 1; ii <= half; ii++)
            if (std::fabs(t.k[half + ii] - t.k[half - ii]) > 2.220446049250313e-16) { t.sym = 0; break; }
        if (t.sym == 0) {
            t.sym = -1;
            for (int ii = 1; ii <= half; ii++)
                if (std::fabs(t.k[half + ii] + t.k[half - ii]) > 2.220446049250313e-16) { t.sym = 0; break; }
        }
    }
}


int upload_raw(klt_ctx *c, int slot, const void *px, int ncols, int nrows, int pitch, int kind)
{
    if (!c || !px) return fail(c, KLT_ERR_ARG, "null argument");
    if (ncols <= 0 || nrows <= 0 || ncols > 65535 || nrows > 65535 || pitch < ncols)
        return fail(c, KLT_ERR_ARG, "bad image geometry");
    if ((long long)ncols * nrows > kMaxFramePixels) return fail(c, KLT_ERR_ARG, "frame too large (2^28 pixels or more: a plane must stay below 2 GB)");
    HIPCHK(c, hipSetDevice(c->device));
    Slot *s;
    if (int rc = get_slot(c, slot, &s, true)) return rc;
    if (s->upload_pending) {
        HIPCHK(c, hipStreamSynchronize(c->cstream));
        for (hipStream_t x : c->cextra) if (x) HIPCHK(c, hipStreamSynchronize(x));
        s->upload_pending = false;
    }
    if (int rc = wait_built(c, s)) return rc;                 // a build on the build stream may still read the old frame
    const size_t px_count = (size_t)ncols * nrows;
    s->u8_ext = nullptr;
    if (kind == 1) { if (int rc = ensure(c, s->u8, s->u8_cap, px_count)) return rc; }
    else { if (int rc = ensure(c, s->f32, s->f32_cap, px_count)) return rc; }
    const size_t esz = kind == 1 ? 1 : sizeof(float);
    void *dst = kind == 1 ? (void *)s->u8 : (void *)s->f32;
    if (pitch == ncols) HIPCHK(c, hipMemcpyAsync(dst, px, px_count * esz, hipMemcpyHostToDevice, c->stream));
    else HIPCHK(c, hipMemcpy2DAsync(dst, (size_t)ncols * esz, px, (size_t)pitch * esz, (size_t)ncols * esz, nrows,
                                    hipMemcpyHostToDevice, c->stream));
    // pageable host memory: the copy above is staged before returning, but make it explicit
    HIPCHK(c, hipStreamSynchronize(c->stream));
    s->nc = ncols;
    s->nr = nrows;
    s->raw_kind = kind;
    s->f32_in_raw = false;
    s->pyr_valid = false;
    return 0;
}

int layout_pyramid(klt_ctx *c, Slot *s)
{
    const int L = c->p.nPyramidLevels, ss = c->p.subsampling;
    // [image planes of all levels][interleaved gradient planes of all levels]; every level starts on a 16-byte boundary
    auto padded = [](int nc_, int nr_) { return ((size_t)nc_ * nr_ + 3) & ~(size_t)3; };
    size_t total = 0;
    int nc = s->nc, nr = s->nr;
    for (int l = 0; l < L; l++) {
        if (nc <= 0 || nr <= 0) return fail(c, KLT_ERR_ARG, "image too small for the requested pyramid");
        total += padded(nc, nr);
        nc /= ss;
        nr /= ss;
    }
    if (3 * total > s->planes_cap) {
        if (s->planes) { if (int rc = sync_all(c)) return rc; hipFree(s->planes); s->planes = nullptr; s->planes_cap = 0; }
        s->pyr_valid = false;                              // (the old planes are gone whether or not the new ones can be had)
        DEVALLOC(c, s->planes, 3 * total * sizeof(float));
        s->planes_cap = 3 * total;
    }
    size_t off = 0;
    nc = s->nc;
    nr = s->nr;
    for (int l = 0; l < L; l++) {
        s->lv[l].nc = nc;
        s->lv[l].nr = nr;
        s->lv[l].img = s->planes + off;
        s->lv[l].gx = s->planes + total + KLT_GRAD_STRIDE * off;      // gradx and grady of a pixel side by side (klt_internal.h)
        s->lv[l].gy = s->lv[l].gx + 1;
        off += padded(nc, nr);
        nc /= ss;
        nr /= ss;
    }
    s->nlev = L;
    s->ss = ss;
    return 0;
}

int enqueue_smooth_raw(klt_ctx *c, Slot *s, float *dst)
{
    const int nc = s->nc, nr = s->nr;
    const double N = (double)nc * nr;
    const Taps &g = c->gauss[0];
    {
        TimerScope t(c, F_SMOOTH_H, N * ((s->raw_kind == 1 ? 1 : 4) + 4));
        if (s->raw_kind == 1) launch_hconv_u8(c->work, raw8(s), nc, nr, c->tmpA, nullptr, nc, 1, 0, g, nullptr);
        else launch_hconv_f32(c->work, rawf(s), nc, nr, c->tmpA, nullptr, nc, 1, 0, g, nullptr);
    }
    {
        TimerScope t(c, F_SMOOTH_V, N * 8);
        launch_vconv(c->work, c->tmpA, nullptr, nc, nr, dst, nullptr, nr, 1, 0, g, nullptr);
    }
    return 0;
}

// KLTComputeGradients, convolve.py:226-248: gx = (deriv horizontally, gauss vertically), gy = (gauss, deriv)
int enqueue_gradients(klt_ctx *c, const float *img, int nc, int nr, float *gx, float *gy)
{
    const double N = (double)nc * nr;
    {
        TimerScope t(c, F_GRAD_H, N * 12);
        launch_hconv_f32(c->work, img, nc, nr, c->tmpA, c->tmpB, nc, 1, 0, c->deriv[2], &c->gauss[2]);
    }
    {
        TimerScope t(c, F_GRAD_V, N * 16);
        launch_vconv(c->work, c->tmpA, c->tmpB, nc, nr, gx, gy, nr, 1, 0, c->gauss[2], &c->deriv[2], KLT_GRAD_STRIDE);   // a level's interleaved planes
    }
    return 0;
}

constexpr size_t kMaxLds = 150 * 1024;     // leave headroom below the 160 KiB of a CU

int grad_radius(const klt_ctx *c) { return (c->gauss[2].n > c->deriv[2].n ? c->gauss[2].n : c->deriv[2].n) / 2; }

bool fused_smooth_ok(const klt_ctx *c)
{
    return c->use_fused && c->gauss[0].sym == 1 && smooth_grad_lds_bytes(c->gauss[0].n / 2, grad_radius(c)) <= kMaxLds;
}
// the register-blocked kernels take per-entry geometry (levels of different size in one launch); dims must fit a short
bool merged_grad_ok(const klt_ctx *c)
{
    return c->gauss[2].sym == 1 && c->deriv[2].sym == -1 && c->gauss[2].n == 7 && c->deriv[2].n == 7;
}
bool fused_grad_ok(const klt_ctx *c) { return c->use_fused && smooth_grad_lds_bytes(-1, grad_radius(c)) <= kMaxLds; }
bool fused_reduce_ok(const klt_ctx *c) { return c->use_fused && pyr_reduce_lds_bytes(c->p.subsampling, c->gauss[1].n) <= kMaxLds; }

// 2 when every entry's grady plane starts one element behind its gradx plane (the interleaved planes of slots and of the selection), else 1
static int grad_stride_of(float *const *gx, float *const *gy, int batch)
{
    for (int b = 0; b < batch; b++)
        if (gy[b] != gx[b] + 1) return 1;
    return KLT_GRAD_STRIDE;
}

// smooth(raw frame) + gradients for up to KLT_MAX_BATCH same-sized frames in one launch
// *fused_h1 (optional, in/out): in = the caller wants the horizontal pass of the first reduction fused into this launch; out =
// whether it was (then c->h1 holds one H1 plane of nr x (nc / ss) floats per frame)
int enqueue_fused_smooth_grad(klt_ctx *c, int batch, const void *const *raw, int raw_kind, float *const *img,
                              float *const *gx, float *const *gy, int nc, int nr, bool *fused_h1 /* = nullptr */)
{
    SmoothGradArgs a;
    std::memset(&a, 0, sizeof(a));
    for (int b = 0; b < batch; b++) { a.raw[b] = raw[b]; a.img[b] = img[b]; a.gx[b] = gx[b]; a.gy[b] = gy[b]; }
    a.gstride = grad_stride_of(gx, gy, batch);
    a.smooth = c->gauss[0]; a.ggauss = c->gauss[2]; a.gderiv = c->deriv[2];
    a.ncols = nc; a.nrows = nr; a.R = grad_radius(c);
    const int kind = raw_kind == 1 ? 0 : 1;
    bool hred = fused_h1 && *fused_h1 && c->fuse_hreduce && smooth_grad_hred_ok(a, batch, kind, c->gauss[1], c->p.subsampling);
    if (hred) {
        const size_t plane = (size_t)nr * (nc / c->p.subsampling);
        if (int rc = ensure_h1(c, plane * batch)) return rc;
        a.reduce = c->gauss[1];
        a.h1_nc = nc / c->p.subsampling;
        for (int b = 0; b < batch; b++) a.h1[b] = c->h1 + plane * b;
    }
    if (fused_h1) *fused_h1 = hred;
    const double N = (double)nc * nr * batch;
    // algorithmic bytes: smoothing b_in + 4, gradients 12 per pixel; with the fused horizontal reduction this
    // launch also consumes the reduction stage's input (4 per pixel of level 0 -- the part of 4 (N0 + N1) that no longer
    // touches HBM); pyr_vreduce is charged the stage's output, so the step total is unchanged
    TimerScope t(c, F_SMOOTH_GRAD, N * ((raw_kind == 1 ? 1 : 4) + 4) + N * 12 + (hred ? 4.0 * N : 0.0));
    if (int e = launch_smooth_grad(c->work, a, batch, kind, hred))
        return fail(c, KLT_ERR_DEVICE, std::string("smooth_grad launch: ") + hipGetErrorString((hipError_t)e));
    return 0;
}

// gradients of up to KLT_MAX_BATCH same-sized f32 images in one launch
int enqueue_fused_grad(klt_ctx *c, int batch, const float *const *img, float *const *gx, float *const *gy, int nc, int nr,
                       bool u8_input /* = false */)
{
    SmoothGradArgs a;
    std::memset(&a, 0, sizeof(a));
    for (int b = 0; b < batch; b++) { a.raw[b] = img[b]; a.gx[b] = gx[b]; a.gy[b] = gy[b]; }
    a.gstride = grad_stride_of(gx, gy, batch);
    a.smooth = c->gauss[0]; a.ggauss = c->gauss[2]; a.gderiv = c->deriv[2];
    a.ncols = nc; a.nrows = nr; a.R = grad_radius(c);
    TimerScope t(c, F_GRAD, (double)nc * nr * batch * 12);
    if (int e = launch_smooth_grad(c->work, a, batch, u8_input ? 3 : 2))
        return fail(c, KLT_ERR_DEVICE, std::string("gradient launch: ") + hipGetErrorString((hipError_t)e));
    return 0;
}

int build_pyramids_batch(klt_ctx *c, const int *slot_ids, int n)
{
    if (int rc = check_ready(c)) return rc;
    if (!slot_ids || n <= 0) return fail(c, KLT_ERR_ARG, "empty slot list");
    HIPCHK(c, hipSetDevice(c->device));
    // KLT_OPT_BUILD_STREAM: the whole build goes to the build stream, behind everything enqueued on the main stream so far (the
    // earlier readers of these slots, synchronous uploads) -- one event each way per build.  The generic two-pass kernels share
    // scratch with the selection, so they stay on the main stream.
    struct WorkScope {
        klt_ctx *c;
        ~WorkScope() { c->work = c->stream; }
    } work_scope{c};
    // (Round 3 measured the level-0 kernel alone on the build stream, levels >= 1 and the tracker on the main stream -- "KLT_OPT_L0_STREAM",
    // commit 72f1f4e: bit-identical, and no faster: one context 0.0452 ms per pair against 0.0462 on one stream with two pairs per launch.
    // The level-0 kernel and the tracker are both bound by VALU issue: running side by side they take 70 us where they take 48 + 26 one
    // after the other, profiles/README.md.)
    const bool on_bstream = c->build_stream_on && c->use_fused && fused_smooth_ok(c) && fused_grad_ok(c) && fused_reduce_ok(c);
    if (on_bstream) {
        if (!c->bstream) HIPCHK(c, hipStreamCreateWithFlags(&c->bstream, hipStreamNonBlocking));
        if (c->last_build_on_bstream != 1) {
            // the first build over here: behind everything on the main stream (earlier builds there share the H1 scratch)
            hipEvent_t mark;
            if (int rc = fresh_event(c, &mark)) return rc;
            HIPCHK(c, hipEventRecord(mark, c->stream));
            HIPCHK(c, hipStreamWaitEvent(c->bstream, mark, 0));
        }
        c->work = c->bstream;
    } else if (c->last_build_on_bstream == 1 && c->ev_bbuild) {
        if (event_live(c, c->bbuild_serial)) HIPCHK(c, hipStreamWaitEvent(c->stream, c->ev_bbuild, 0));
        else HIPCHK(c, hipStreamSynchronize(c->bstream));
    }
    std::vector<Slot *> sl((size_t)n);
    std::vector<uint64_t> read_waited;
    for (int i = 0; i < n; i++) {
        if (int rc = get_slot(c, slot_ids[i], &sl[i], false)) return rc;
        if (sl[i]->raw_kind == 0) return fail(c, KLT_ERR_STATE, "slot has no frame");
        for (int j = 0; j < i; j++)
            if (sl[j] == sl[i]) return fail(c, KLT_ERR_ARG, "slot listed twice");
        if (int rc = layout_pyramid(c, sl[i])) return rc;
        if (int rc = wait_upload(c, sl[i], c->work)) return rc;      // asynchronous ingest: the frame must have landed
        if (!on_bstream) { if (int rc = wait_built(c, sl[i])) return rc; }
        else if (sl[i]->read_valid) {
            // a tracker on the main stream may still be reading the pyramids this build overwrites: wait for that launch only (a mark on
            // the whole main stream would put the build behind a tracker enqueued just before it -- the overlap the stream is for)
            // (once per launch: the slots of a batch that one launch read share its event)
            if (std::find(read_waited.begin(), read_waited.end(), sl[i]->read_serial) == read_waited.end()) {
                if (event_live(c, sl[i]->read_serial)) HIPCHK(c, hipStreamWaitEvent(c->bstream, sl[i]->ev_read, 0));
                else HIPCHK(c, hipStreamSynchronize(c->stream));
                read_waited.push_back(sl[i]->read_serial);
            }
        }
        sl[i]->read_valid = false;
    }
    c->last_build_on_bstream = on_bstream ? 1 : 0;
    const int ss = c->p.subsampling;
    // groups of frames with the same geometry and input type share launches
    std::vector<bool> done((size_t)n, false);
    for (int i0 = 0; i0 < n; i0++) {
        if (done[i0]) continue;
        std::vector<Slot *> g;
        for (int i = i0; i < n && (int)g.size() < KLT_MAX_BATCH; i++)
            if (!done[i] && sl[i]->nc == sl[i0]->nc && sl[i]->nr == sl[i0]->nr && sl[i]->raw_kind == sl[i0]->raw_kind) {
                g.push_back(sl[i]);
                done[i] = true;
            }
        const int B = (int)g.size();
        Slot *s0 = g[0];
        const void *raw[KLT_MAX_BATCH];
        const float *src[KLT_MAX_BATCH];
        float *img[KLT_MAX_BATCH], *gx[KLT_MAX_BATCH], *gy[KLT_MAX_BATCH];
        if (int rc = ensure_tmp(c, (size_t)s0->nc * s0->nr)) return rc;

        // level 0: smoothed frame (trackFeatures.py:165-166) and its gradients (:171-172)
        bool h1_fused = false;              // level 1 comes from the H1 planes written by the level-0 kernel
        if (fused_smooth_ok(c)) {
            for (int b = 0; b < B; b++) {
                raw[b] = g[b]->raw_kind == 1 ? (const void *)raw8(g[b]) : (const void *)rawf(g[b]);
                img[b] = g[b]->lv[0].img; gx[b] = g[b]->lv[0].gx; gy[b] = g[b]->lv[0].gy;
            }
            h1_fused = s0->nlev > 1 && fused_reduce_ok(c);
            if (int rc = enqueue_fused_smooth_grad(c, B, raw, s0->raw_kind, img, gx, gy, s0->nc, s0->nr, &h1_fused)) return rc;
        } else {
            for (int b = 0; b < B; b++) {
                enqueue_smooth_raw(c, g[b], g[b]->lv[0].img);
                enqueue_gradients(c, g[b]->lv[0].img, g[b]->nc, g[b]->nr, g[b]->lv[0].gx, g[b]->lv[0].gy);
            }
        }
        // levels 1..L-1: smooth with the pyramid sigma, keep pixel (ss*y + ss/2, ss*x + ss/2) (pyramid.py:59-72),
        // then the gradients of the new level.  Only surviving columns / rows are evaluated.
        for (int l = 1; l < s0->nlev; l++) {
            const Level &ls = s0->lv[l - 1];
            const Level &ld = s0->lv[l];
            if (fused_reduce_ok(c)) {
                PyrReduceArgs a;
                std::memset(&a, 0, sizeof(a));
                for (int b = 0; b < B; b++) { a.src[b] = g[b]->lv[l - 1].img; a.dst[b] = g[b]->lv[l].img; }
                a.taps = c->gauss[1];
                a.src_nc = ls.nc; a.src_nr = ls.nr; a.dst_nc = ld.nc; a.dst_nr = ld.nr; a.ss = ss;
                a.log2ss = 0;
                while ((1 << a.log2ss) < ss) a.log2ss++;
                if (l == 1 && h1_fused) {
                    // vertical pass only: H1 (level-0 rows x level-1 columns) -> level 1
                    const size_t plane = (size_t)ls.nr * ld.nc;
                    for (int b = 0; b < B; b++) a.src[b] = c->h1 + plane * b;
                    TimerScope t(c, F_PYR_REDUCE, 4.0 * B * ((double)ld.nc * ld.nr));
                    if (int e = launch_pyr_vreduce(c->work, a, B))
                        return fail(c, KLT_ERR_DEVICE, std::string("pyr_vreduce launch: ") + hipGetErrorString((hipError_t)e));
                } else {
                    TimerScope t(c, F_PYR_REDUCE, 4.0 * B * ((double)ls.nc * ls.nr + (double)ld.nc * ld.nr));
                    if (int e = launch_pyr_reduce(c->work, a, B))
                        return fail(c, KLT_ERR_DEVICE, std::string("pyr_reduce launch: ") + hipGetErrorString((hipError_t)e));
                }
            } else {
                for (int b = 0; b < B; b++) {
                    {
                        TimerScope t(c, F_PYR_H, 4.0 * ((double)ls.nc * ls.nr + (double)ld.nc * ls.nr));
                        launch_hconv_f32(c->work, g[b]->lv[l - 1].img, ls.nc, ls.nr, c->tmpA, nullptr, ld.nc, ss, ss / 2, c->gauss[1], nullptr);
                    }
                    {
                        TimerScope t(c, F_PYR_V, 4.0 * ((double)ld.nc * ls.nr + (double)ld.nc * ld.nr));
                        launch_vconv(c->work, c->tmpA, nullptr, ld.nc, ls.nr, g[b]->lv[l].img, nullptr, ld.nr, ss, ss / 2, c->gauss[1], nullptr);
                    }
                }
            }
            const bool merged = fused_grad_ok(c) && merged_grad_ok(c) && B * (s0->nlev - 1) <= KLT_MAX_BATCH && s0->nc / s0->ss <= 32767 && s0->nr / s0->ss <= 32767;
            if (merged) continue;          // gradients of all levels >= 1 go out in one launch below
            if (fused_grad_ok(c)) {
                for (int b = 0; b < B; b++) { src[b] = g[b]->lv[l].img; gx[b] = g[b]->lv[l].gx; gy[b] = g[b]->lv[l].gy; }
                if (int rc = enqueue_fused_grad(c, B, src, gx, gy, ld.nc, ld.nr)) return rc;
            } else {
                for (int b = 0; b < B; b++) enqueue_gradients(c, g[b]->lv[l].img, ld.nc, ld.nr, g[b]->lv[l].gx, g[b]->lv[l].gy);
            }
        }
        if (s0->nlev > 1 && fused_grad_ok(c) && merged_grad_ok(c) && B * (s0->nlev - 1) <= KLT_MAX_BATCH && s0->nc / s0->ss <= 32767 && s0->nr / s0->ss <= 32767) {
            // one launch for the gradients of every level >= 1 of every frame: entry = (frame, level), per-entry geometry
            SmoothGradArgs a;
            std::memset(&a, 0, sizeof(a));
            int e = 0;
            double bytes = 0;
            for (int l = 1; l < s0->nlev; l++)
                for (int b = 0; b < B; b++, e++) {
                    a.raw[e] = g[b]->lv[l].img; a.gx[e] = g[b]->lv[l].gx; a.gy[e] = g[b]->lv[l].gy;
                    a.dim_c[e] = (short)g[b]->lv[l].nc; a.dim_r[e] = (short)g[b]->lv[l].nr;
                    bytes += 12.0 * g[b]->lv[l].nc * g[b]->lv[l].nr;
                }
            a.gstride = KLT_GRAD_STRIDE;                              // slot planes: gradx / grady interleaved
            a.smooth = c->gauss[0]; a.ggauss = c->gauss[2]; a.gderiv = c->deriv[2];
            a.ncols = s0->lv[1].nc; a.nrows = s0->lv[1].nr; a.R = grad_radius(c);
            TimerScope t(c, F_GRAD, bytes);
            if (int er = launch_smooth_grad(c->work, a, e, 2))
                return fail(c, KLT_ERR_DEVICE, std::string("gradient launch: ") + hipGetErrorString((hipError_t)er));
        }
        for (Slot *s : g) { s->pyr_valid = true; s->gen = ++c->gen_counter; }
    }
    // the next asynchronous copy into these slots waits for this build
    if (int rc = mark_consumed(c, sl.data(), n, c->work)) return rc;
    if (on_bstream) {
        hipEvent_t e;
        uint64_t serial;
        if (int rc = fresh_event(c, &e, &serial)) return rc;
        HIPCHK(c, hipEventRecord(e, c->bstream));
        for (Slot *s : sl) { s->ev_built = e; s->built_serial = serial; s->built_pending = true; }
        c->ev_bbuild = e; c->bbuild_serial = serial;
    }
    for (Slot *s : sl) s->built_on_bstream = on_bstream;
    HIPCHK(c, hipGetLastError());
    return KLT_OK;
}



int download_plane(klt_ctx *c, const float *src, int stride, size_t cnt, float *dst)
{
    float *tmp = nullptr;
    if (stride != 1) {
        DEVALLOC(c, tmp, cnt * sizeof(float));
        launch_take_strided(c->stream, src, tmp, cnt, stride);
        src = tmp;
    }
    hipError_t e = hipMemcpyAsync(dst, src, cnt * sizeof(float), hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    if (tmp) hipFree(tmp);
    HIPCHK(c, e);
    return KLT_OK;
}

}  // namespace kltapi

extern "C" {

int klt_set_kernels(klt_ctx *c, int which, const double *gauss, int ng, const double *deriv, int nd)
{
    if (!c || !gauss || !deriv) return fail(c, KLT_ERR_ARG, "null argument");
    if (which < 0 || which > 2) return fail(c, KLT_ERR_ARG, "which must be 0, 1 or 2");
    if (ng < 1 || nd < 1 || ng > KLT_MAX_KERNEL_WIDTH || nd > KLT_MAX_KERNEL_WIDTH || !(ng & 1) || !(nd & 1))
        return fail(c, KLT_ERR_ARG, "tap counts must be odd and at most 71");
    Taps g, d;
    make_taps(gauss, ng, g);
    make_taps(deriv, nd, d);
    // resident pyramids (sequentialMode: tc.pyramid_last, trackFeatures.py:152-161) stay valid unless the taps really change
    const bool same = c->have_taps[which] && std::memcmp(&g, &c->gauss[which], sizeof(Taps)) == 0 &&
                      std::memcmp(&d, &c->deriv[which], sizeof(Taps)) == 0;
    if (same) return KLT_OK;
    c->gauss[which] = g;
    c->deriv[which] = d;
    c->have_taps[which] = true;
    for (Slot &s : c->slots) s.pyr_valid = false;
    return KLT_OK;
}


int klt_upload_u8(klt_ctx *c, int slot, const uint8_t *px, int ncols, int nrows, int pitch)
{
    return upload_raw(c, slot, px, ncols, nrows, pitch, 1);
}

int klt_upload_f32(klt_ctx *c, int slot, const float *px, int ncols, int nrows, int pitch)
{
    return upload_raw(c, slot, px, ncols, nrows, pitch, 2);
}

int klt_host_alloc(klt_ctx *c, size_t bytes, void **out)
{
    if (!c || !out || bytes == 0) return fail(c, KLT_ERR_ARG, "bad argument");
    HIPCHK(c, hipSetDevice(c->device));
    void *p = nullptr;
    if (int rc = host_alloc(c, &p, bytes, "klt_host_alloc")) return rc;
    c->pinned.push_back(p);
    *out = p;
    return KLT_OK;
}

int klt_host_free(klt_ctx *c, void *p)
{
    if (!c || !p) return fail(c, KLT_ERR_ARG, "bad argument");
    for (size_t i = 0; i < c->pinned.size(); i++)
        if (c->pinned[i] == p) {
            if (int rc = sync_all(c)) return rc;
            HIPCHK(c, hipHostFree(p));
            c->pinned.erase(c->pinned.begin() + (long)i);
            return KLT_OK;
        }
    return fail(c, KLT_ERR_ARG, "pointer was not allocated with klt_host_alloc");
}

// one frame from pinned host memory into the slot's other raw buffer, on one of the copy streams; esz = bytes per pixel (1: u8, 4: f32;
// pitch in pixels)
static int upload_async(klt_ctx *c, int slot, const void *px, int ncols, int nrows, int pitch, size_t esz)
{
    if (!c || !px) return fail(c, KLT_ERR_ARG, "null argument");
    if (ncols <= 0 || nrows <= 0 || ncols > 65535 || nrows > 65535 || pitch < ncols) return fail(c, KLT_ERR_ARG, "bad image geometry");
    if ((long long)ncols * nrows > kMaxFramePixels) return fail(c, KLT_ERR_ARG, "frame too large (2^28 pixels or more: a plane must stay below 2 GB)");
    HIPCHK(c, hipSetDevice(c->device));
    hipPointerAttribute_t attr;                           // the source must be pinned: a pageable copy would be staged synchronously
    if (hipPointerGetAttributes(&attr, px) != hipSuccess || attr.type != hipMemoryTypeHost) {
        (void)hipGetLastError();
        return fail(c, KLT_ERR_ARG, "klt_upload_u8_async / klt_upload_f32_async need pinned host memory (klt_host_alloc)");
    }
    if (!c->cstream) HIPCHK(c, hipStreamCreateWithFlags(&c->cstream, hipStreamNonBlocking));
    const int lane = (int)(c->upload_count++ % (unsigned)c->ncopy);                 // the two frames of a pair travel side by side
    if (lane > 0 && !c->cextra[lane - 1]) HIPCHK(c, hipStreamCreateWithFlags(&c->cextra[lane - 1], hipStreamNonBlocking));
    const hipStream_t cs = lane == 0 ? c->cstream : c->cextra[lane - 1];
    Slot *s;
    if (int rc = get_slot(c, slot, &s, true)) return rc;
    const size_t px_count = (size_t)ncols * nrows * esz;      // bytes: the raw buffers are byte buffers (`u8_cap` counts bytes)
    // write into the buffer the build before last read (normally long finished: poll, block only if it is not)
    s->u8_ext = nullptr;
    std::swap(s->u8, s->u8_alt);
    std::swap(s->u8_cap, s->u8_alt_cap);
    std::swap(s->ev_consumed, s->ev_consumed_alt);
    std::swap(s->consumed_serial, s->consumed_alt_serial);
    std::swap(s->consumed_valid, s->consumed_alt_valid);
    std::swap(s->ev_wr, s->ev_wr_alt);
    std::swap(s->wr_serial, s->wr_alt_serial);
    std::swap(s->wr_lane, s->wr_alt_lane);
    if (int rc = ensure(c, s->u8, s->u8_cap, px_count)) return rc;
    if (s->wr_lane >= 0 && s->wr_lane != lane) {
        // an earlier copy into this very buffer went through another copy stream (a slot uploaded twice without a build in between, an
        // upload abandoned by klt_slot_adopt_u8): this one is ordered behind it -- a copy-stream event, normally long complete
        if (event_live(c, s->wr_serial)) HIPCHK(c, hipStreamWaitEvent(cs, s->ev_wr, 0));
        else HIPCHK(c, hipStreamSynchronize(s->wr_lane == 0 ? c->cstream : c->cextra[s->wr_lane - 1]));
    }
    if (s->consumed_valid) {
        if (!event_live(c, s->consumed_serial)) {
            // the event has been re-used since: wait for the reading streams themselves (a build on the build stream reads raw frames too)
            HIPCHK(c, hipStreamSynchronize(c->stream));
            if (c->bstream) HIPCHK(c, hipStreamSynchronize(c->bstream));
        } else {
            // poll: the build in question is at most a few launches from done, and hipEventSynchronize wakes the host 50-100 us late --
            // long enough for the queues to run dry behind it (tools/ingest_probe.py: 246 us per pair with the blocking wait)
            hipError_t q = hipEventQuery(s->ev_consumed);
            for (long spins = 0; q == hipErrorNotReady; spins++) {
                (void)hipGetLastError();                          // ("not ready" must not surface in a later hipGetLastError check)
                if (spins > 2000000) { HIPCHK(c, hipEventSynchronize(s->ev_consumed)); q = hipSuccess; break; }   // (seconds: something else is wrong)
                q = hipEventQuery(s->ev_consumed);
            }
            if (q != hipSuccess) return fail(c, KLT_ERR_DEVICE, std::string("hipEventQuery: ") + hipGetErrorString(q));
        }
        s->consumed_valid = false;
    }
    if (pitch == ncols) HIPCHK(c, hipMemcpyAsync(s->u8, px, px_count, hipMemcpyHostToDevice, cs));
    else HIPCHK(c, hipMemcpy2DAsync(s->u8, (size_t)ncols * esz, px, (size_t)pitch * esz, (size_t)ncols * esz, nrows, hipMemcpyHostToDevice, cs));
    if (int rc = fresh_event(c, &s->ev_upload, &s->upload_serial)) return rc;
    HIPCHK(c, hipEventRecord(s->ev_upload, cs));
    s->upload_pending = true;
    s->ev_wr = s->ev_upload; s->wr_serial = s->upload_serial; s->wr_lane = lane;
    s->nc = ncols;
    s->nr = nrows;
    s->raw_kind = esz == 1 ? 1 : 2;
    s->f32_in_raw = esz != 1;
    s->pyr_valid = false;
    return KLT_OK;
}

int klt_upload_u8_async(klt_ctx *c, int slot, const uint8_t *px, int ncols, int nrows, int pitch)
{
    return upload_async(c, slot, px, ncols, nrows, pitch, 1);
}

int klt_upload_f32_async(klt_ctx *c, int slot, const float *px, int ncols, int nrows, int pitch)
{
    return upload_async(c, slot, px, ncols, nrows, pitch, sizeof(float));
}

int klt_upload_wait(klt_ctx *c)
{
    if (!c) return KLT_ERR_ARG;
    HIPCHK(c, hipSetDevice(c->device));
    if (c->cstream) HIPCHK(c, hipStreamSynchronize(c->cstream));
    for (hipStream_t x : c->cextra) if (x) HIPCHK(c, hipStreamSynchronize(x));
    return KLT_OK;
}

int klt_device_alloc(klt_ctx *c, size_t bytes, void **out)
{
    if (!c || !out || !bytes) return fail(c, KLT_ERR_ARG, "bad argument");
    HIPCHK(c, hipSetDevice(c->device));
    void *p = nullptr;
    if (int rc = dev_alloc(c, &p, bytes, "klt_device_alloc")) return rc;
    c->dev_allocs.push_back(p);
    c->dev_alloc_bytes.push_back(bytes);
    *out = p;
    return KLT_OK;
}

int klt_device_free(klt_ctx *c, void *p)
{
    if (!c) return KLT_ERR_ARG;
    for (size_t i = 0; i < c->dev_allocs.size(); i++)
        if (c->dev_allocs[i] == p) {
            HIPCHK(c, hipSetDevice(c->device));
            if (int rc = sync_all(c)) return rc;              // a build may still read a frame adopted from it
            // slots that had adopted a frame INSIDE this allocation hold no frame any more (not the slot's own raw buffer, which may be
            // smaller than the adopted frame and holds an older image); slots adopted from other memory keep theirs
            const uint8_t *lo = (const uint8_t *)p, *hi = lo + c->dev_alloc_bytes[i];
            for (Slot &s : c->slots)
                if (s.u8_ext && s.u8_ext >= lo && s.u8_ext < hi) {
                    s.u8_ext = nullptr; s.raw_kind = 0; s.pyr_valid = false; s.gen = 0; s.nc = s.nr = 0;
                }
            hipFree(p);
            c->dev_allocs.erase(c->dev_allocs.begin() + (long)i);
            c->dev_alloc_bytes.erase(c->dev_alloc_bytes.begin() + (long)i);
            return KLT_OK;
        }
    return fail(c, KLT_ERR_ARG, "not a klt_device_alloc allocation");
}

int klt_device_write(klt_ctx *c, void *dst, const void *src, size_t bytes)
{
    if (!c || !dst || !src) return fail(c, KLT_ERR_ARG, "null argument");
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return KLT_OK;
}

int klt_slot_adopt_u8(klt_ctx *c, int slot, const uint8_t *dev_px, int ncols, int nrows, int pitch)
{
    if (!c || !dev_px) return fail(c, KLT_ERR_ARG, "null argument");
    if (ncols <= 0 || nrows <= 0 || ncols > 65535 || nrows > 65535) return fail(c, KLT_ERR_ARG, "bad image geometry");
    if (pitch != ncols) return fail(c, KLT_ERR_ARG, "an adopted frame must have contiguous rows (pitch == ncols): it is read in place");
    if ((long long)ncols * nrows > kMaxFramePixels) return fail(c, KLT_ERR_ARG, "frame too large (2^28 pixels or more: a plane must stay below 2 GB)");
    HIPCHK(c, hipSetDevice(c->device));
    hipPointerAttribute_t attr;
    if (hipPointerGetAttributes(&attr, dev_px) != hipSuccess || attr.type != hipMemoryTypeDevice) {
        (void)hipGetLastError();
        return fail(c, KLT_ERR_ARG, "klt_slot_adopt_u8 needs device memory");
    }
    Slot *s;
    if (int rc = get_slot(c, slot, &s, true)) return rc;
    // nothing is enqueued: the pointer is what the next build / selection of the slot reads.  Work already enqueued read the slot's
    // previous frame through its own pointer and is unaffected; a pending asynchronous upload into the slot is abandoned
    s->upload_pending = false;
    s->u8_ext = dev_px;
    s->nc = ncols;
    s->nr = nrows;
    s->raw_kind = 1;
    s->f32_in_raw = false;
    s->pyr_valid = false;
    return KLT_OK;
}

int klt_build_pyramids_async(klt_ctx *c, int slot) { return build_pyramids_batch(c, &slot, 1); }

int klt_build_pyramids_batch_async(klt_ctx *c, const int *slots, int n) { return build_pyramids_batch(c, slots, n); }


int klt_build_pyramids(klt_ctx *c, int slot)
{
    if (int rc = klt_build_pyramids_async(c, slot)) return rc;
    return klt_sync(c);
}


// -------------------------------------------------------------------------------------- inspection
int klt_level_dims(klt_ctx *c, int slot, int level, int *ncols, int *nrows)
{
    if (!c) return KLT_ERR_ARG;
    Slot *s;
    if (int rc = get_slot(c, slot, &s, false)) return rc;
    if (!s->pyr_valid || level < 0 || level >= s->nlev) return fail(c, KLT_ERR_STATE, "no such pyramid level");
    if (ncols) *ncols = s->lv[level].nc;
    if (nrows) *nrows = s->lv[level].nr;
    return KLT_OK;
}

// a plane to the host; the gradient planes are stored interleaved (klt_internal.h) and leave through a plane of their own

int klt_download_f32(klt_ctx *c, int slot, int pyramid, int level, float *dst)
{
    if (!c || !dst) return fail(c, KLT_ERR_ARG, "null argument");
    Slot *s;
    if (int rc = get_slot(c, slot, &s, false)) return rc;
    if (!s->pyr_valid || level < 0 || level >= s->nlev || pyramid < 0 || pyramid > 2) return fail(c, KLT_ERR_STATE, "no such pyramid level");
    const Level &l = s->lv[level];
    const float *src = pyramid == 0 ? l.img : (pyramid == 1 ? l.gx : l.gy);
    HIPCHK(c, hipSetDevice(c->device));
    if (int rc = wait_built(c, s)) return rc;
    return download_plane(c, src, pyramid == 0 ? 1 : KLT_GRAD_STRIDE, (size_t)l.nc * l.nr, dst);
}


// ------------------------------------------------------------------------- standalone convolutions
// _convolveSeparate, convolve.py:208-219 (SciPy branch): convolve1d along axis 1 with the horizontal taps, f32, then along axis 0 with the
// vertical taps -- ANY two tap lists (1 .. 71 taps each, odd or even, no symmetry assumed: make_taps classifies them as correlate1d does).
int klt_convolve_separate_f32(klt_ctx *c, const float *src, int ncols, int nrows, const double *horiz, int nh, const double *vert, int nv, float *dst)
{
    if (!c || !src || !dst || !horiz || !vert) return fail(c, KLT_ERR_ARG, "null argument");
    if (nh < 1 || nv < 1 || nh > KLT_MAX_KERNEL_WIDTH || nv > KLT_MAX_KERNEL_WIDTH) return fail(c, KLT_ERR_ARG, "1 to 71 taps per direction");
    if (ncols <= 0 || nrows <= 0 || (long long)ncols * nrows > kMaxFramePixels) return fail(c, KLT_ERR_ARG, "bad image geometry");
    HIPCHK(c, hipSetDevice(c->device));
    const size_t N = (size_t)ncols * nrows;
    if (int rc = ensure_tmp(c, N)) return rc;
    float *d_in = nullptr;
    DEVALLOC(c, d_in, 2 * N * sizeof(float));
    float *d_out = d_in + N;
    Taps th, tv;
    make_taps(horiz, nh, th);
    make_taps(vert, nv, tv);
    hipError_t e = hipMemcpyAsync(d_in, src, N * sizeof(float), hipMemcpyHostToDevice, c->stream);
    if (e == hipSuccess) {
        launch_hconv_f32(c->stream, d_in, ncols, nrows, c->tmpA, nullptr, ncols, 1, 0, th, nullptr);
        launch_vconv(c->stream, c->tmpA, nullptr, ncols, nrows, d_out, nullptr, nrows, 1, 0, tv, nullptr);
        e = hipMemcpyAsync(dst, d_out, N * sizeof(float), hipMemcpyDeviceToHost, c->stream);
    }
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    hipFree(d_in);
    if (e != hipSuccess) return fail(c, KLT_ERR_DEVICE, hipGetErrorString(e));
    return KLT_OK;
}

int klt_smooth_f32(klt_ctx *c, const float *src, int ncols, int nrows, const double *gauss, int ng, float *dst)
{
    if (!c || !src || !dst || !gauss) return fail(c, KLT_ERR_ARG, "null argument");
    if (ng < 1 || ng > KLT_MAX_KERNEL_WIDTH || !(ng & 1)) return fail(c, KLT_ERR_ARG, "tap count must be odd and at most 71");
    return klt_convolve_separate_f32(c, src, ncols, nrows, gauss, ng, gauss, ng, dst);      // KLTComputeSmoothedImage, convolve.py:263
}

int klt_gradients_f32(klt_ctx *c, const float *src, int ncols, int nrows, const double *gauss, int ng,
                      const double *deriv, int nd, float *gx, float *gy)
{
    if (!c || !src || !gx || !gy || !gauss || !deriv) return fail(c, KLT_ERR_ARG, "null argument");
    if (ng < 1 || nd < 1 || ng > KLT_MAX_KERNEL_WIDTH || nd > KLT_MAX_KERNEL_WIDTH || !(ng & 1) || !(nd & 1))
        return fail(c, KLT_ERR_ARG, "tap counts must be odd and at most 71");
    if (ncols <= 0 || nrows <= 0 || (long long)ncols * nrows > kMaxFramePixels) return fail(c, KLT_ERR_ARG, "bad image geometry");
    HIPCHK(c, hipSetDevice(c->device));
    const size_t N = (size_t)ncols * nrows;
    if (int rc = ensure_tmp(c, N)) return rc;
    float *d_in = nullptr;
    DEVALLOC(c, d_in, 3 * N * sizeof(float));
    float *d_gx = d_in + N, *d_gy = d_in + 2 * N;
    Taps g, d;
    make_taps(gauss, ng, g);
    make_taps(deriv, nd, d);
    hipError_t e = hipMemcpyAsync(d_in, src, N * sizeof(float), hipMemcpyHostToDevice, c->stream);
    if (e == hipSuccess) {
        launch_hconv_f32(c->stream, d_in, ncols, nrows, c->tmpA, c->tmpB, ncols, 1, 0, d, &g);
        launch_vconv(c->stream, c->tmpA, c->tmpB, ncols, nrows, d_gx, d_gy, nrows, 1, 0, g, &d);
        e = hipMemcpyAsync(gx, d_gx, N * sizeof(float), hipMemcpyDeviceToHost, c->stream);
        if (e == hipSuccess) e = hipMemcpyAsync(gy, d_gy, N * sizeof(float), hipMemcpyDeviceToHost, c->stream);
    }
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    hipFree(d_in);
    if (e != hipSuccess) return fail(c, KLT_ERR_DEVICE, hipGetErrorString(e));
    return KLT_OK;
}

// KLTPyramid.Compute, pyramid.py:37-77: level 0 = src as it is; level i = level i-1 smoothed with `gauss` (sigma = subsampling *
// sigma_fact, computed by the caller) and sampled at (ss y + ss/2, ss x + ss/2), dims int(n / ss).  Levels 1 .. nlevels-1 come back
// concatenated in dst.  Only the surviving columns / rows are evaluated; every level stays on the device until the one download.
int klt_pyramid_f32(klt_ctx *c, const float *src, int ncols, int nrows, int nlevels, int subsampling, const double *gauss, int ng, float *dst)
{
    if (!c || !src || !gauss || (nlevels > 1 && !dst)) return fail(c, KLT_ERR_ARG, "null argument");
    if (ng < 1 || ng > KLT_MAX_KERNEL_WIDTH || !(ng & 1)) return fail(c, KLT_ERR_ARG, "tap count must be odd and at most 71");
    if (ncols <= 0 || nrows <= 0 || (long long)ncols * nrows > kMaxFramePixels || nlevels < 1 || nlevels > KLT_MAX_LEVELS) return fail(c, KLT_ERR_ARG, "bad pyramid geometry");
    const int ss = subsampling;
    if (nlevels > 1 && ss != 2 && ss != 4 && ss != 8 && ss != 16 && ss != 32) return fail(c, KLT_ERR_ARG, "subsampling must be 2, 4, 8, 16 or 32");
    if (nlevels == 1) return KLT_OK;
    HIPCHK(c, hipSetDevice(c->device));
    const size_t N = (size_t)ncols * nrows;
    size_t total = 0;
    {
        int nc = ncols, nr = nrows;
        for (int l = 1; l < nlevels; l++) {
            nc /= ss; nr /= ss;
            if (nc <= 0 || nr <= 0) return fail(c, KLT_ERR_ARG, "image too small for the requested pyramid");
            total += (size_t)nc * nr;
        }
    }
    if (int rc = ensure_tmp(c, N)) return rc;
    float *d_in = nullptr;
    DEVALLOC(c, d_in, (N + total) * sizeof(float));
    float *d_lv = d_in + N;
    Taps g;
    make_taps(gauss, ng, g);
    hipError_t e = hipMemcpyAsync(d_in, src, N * sizeof(float), hipMemcpyHostToDevice, c->stream);
    if (e == hipSuccess) {
        const float *cur = d_in;
        float *out = d_lv;
        int nc = ncols, nr = nrows;
        for (int l = 1; l < nlevels; l++) {
            const int dc = nc / ss, dr = nr / ss;
            launch_hconv_f32(c->stream, cur, nc, nr, c->tmpA, nullptr, dc, ss, ss / 2, g, nullptr);
            launch_vconv(c->stream, c->tmpA, nullptr, dc, nr, out, nullptr, dr, ss, ss / 2, g, nullptr);
            cur = out;
            out += (size_t)dc * dr;
            nc = dc; nr = dr;
        }
        e = hipMemcpyAsync(dst, d_lv, total * sizeof(float), hipMemcpyDeviceToHost, c->stream);
    }
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    hipFree(d_in);
    if (e != hipSuccess) return fail(c, KLT_ERR_DEVICE, hipGetErrorString(e));
    return KLT_OK;
}


}  // extern "C"
