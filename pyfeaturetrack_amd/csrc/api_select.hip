// api_select.hip -- selection (goodFeaturesUtils.pyx:17-73, selectGoodFeatures.py:45-135, :141-261): scores prepared ahead of a
// replacement, the two-halves protocol around the host's look at the outcome (klt_select_begin_async / klt_select_finish), the walk over
// a given candidate list, and the hooks the parity tests inspect a selection through.
#include "klt_context.h"

namespace kltapi {

int enqueue_sat(klt_ctx *c, hipStream_t st, const float *gx, const float *gy, float *sat, int nc, int nr, bool rows_only)
{
    const bool pipe = c->sat_variant == 1;
    const double N = (double)nc * nr;
    { TimerScope t(c, F_SAT_ROWS, N * (8 + 12));
      const int e = pipe ? launch_sat_rows_pipe(st, gx, gy, sat, nc, nr) : -1;
      if (e > 0) return fail(c, KLT_ERR_DEVICE, hipGetErrorString((hipError_t)e));
      if (e < 0) launch_sat_rows(st, gx, gy, sat, nc, nr); }
    if (rows_only) return 0;
    { TimerScope t(c, F_SAT_COLS, N * 24);
      const int e = pipe ? launch_sat_cols_pipe(st, sat, nc, nr) : -1;
      if (e > 0) return fail(c, KLT_ERR_DEVICE, hipGetErrorString((hipError_t)e));
      if (e < 0) launch_sat_cols(st, sat, nc, nr); }
    return 0;
}

}  // namespace kltapi

namespace {
// start of an attempt: free slots + snapshot of the list, scores / histogram / cut (attempt 0) or "every candidate" (attempt 1), tile lists
int select_job_start(klt_ctx *c, SelectJob &j)
{
    j.filtered = j.prefilter && j.attempt == 0;
    SelectArgs &sa = j.sa;
    if (j.attempt == 0) {
        launch_mis_prepare(c->stream, j.fl, j.n, j.pa.overwrite_all, j.pa.slots, j.nfill_d, c->fl_snapshot, j.zero_from, j.zero_n);
        if (j.filtered) { sa.hist = j.hist_d; sa.ticket = j.ticket_d; sa.info = j.info_d; sa.hist_target = (unsigned)((j.target + 3) / 4); }
        if (j.filtered && j.mode == KLT_REPLACING_SOME) {
            // only the lost features' slots are filled and the live features' squares are not scored at all: 64 candidates per
            // LOST feature (at least 4096) instead of 64 per list entry -- most of a frame's candidates never enter the passes.
            // (cfg-5, 50-95 lost of 20000 per frame: a floor of 65536 / 16384 / 4096 / 1024 candidates reads 0.424 / 0.387 /
            // 0.365 / 0.364 ms per frame; too tight a cut only costs the repeat below, never the result)
            sa.hist_target = 4096 / 4; sa.hist_slots = j.nfill_d; sa.hist_per_slot = 64 / 4;
        }
        if (j.pre) {
            // scored ahead of time without the seed map: histogram of the keys outside it here, the mask itself in mis_init
            sa.keys = j.pre->keys;
            TimerScope t(c, F_EIGEN, (double)j.ncand * 2);
            launch_mask_hist(c->stream, sa);
        } else {
            // SURVEY 8(d) [score]: the three table planes once + eigenvalue and key per candidate
            TimerScope t(c, F_EIGEN, 12.0 * sa.ncols * sa.nrows + (double)j.ncand * (4 + 8));
            launch_eigen_hist(c->stream, sa);
        }
    } else {
        launch_zero_words(c->stream, j.zero_from, j.zero_n);      // threshold bin 0: every candidate
        j.ma.sparse = 0;
    }
    if (!j.by_rank) HIPCHK(c, hipMemsetAsync(c->keys2, 0, (size_t)j.np2 * sizeof(unsigned long long), c->stream));
    j.round = 0;
    TimerScope t(c, F_NMS, (double)j.ncand * 12);
    launch_mis_init(c->stream, j.ma);
    return 0;
}

// a batch of passes, then the accepted candidates ranked and placed, and the few words the host looks at written to pinned memory
int select_job_rounds(klt_ctx *c, SelectJob &j)
{
    {
        TimerScope t(c, F_NMS, (double)j.n * 16);
        for (int r = 0; r < j.rounds_per_look; r++, j.round++)
            if (const int e = launch_mis_round(c->stream, j.ma, j.round))
                return fail(c, KLT_ERR_DEVICE, std::string("minimum-distance pass: ") + hipGetErrorString((hipError_t)e));
    }
    j.look = j.round < 64 ? j.round : 64;                        // the last `look` passes
    const unsigned *rem_d = j.ma.remaining + j.round - j.look;
    launch_mis_compact(c->stream, j.ma, c->keys2, j.acc_count_d);
    if (j.by_rank) {
        TimerScope t(c, F_NMS, (double)j.n * 16);
        launch_mis_place(c->stream, j.pa, j.acc_count_d, j.rank_d, j.nfill_d, (int)j.bound, c->readback, rem_d, j.look, j.info_d);
    } else {
        { TimerScope t(c, F_SORT, (double)j.np2 * 16); launch_sort_desc(c->stream, c->keys2, (int)j.np2); }
        TimerScope t(c, F_NMS, (double)j.n * 16);
        const int e = launch_nms(c->stream, j.pa);
        if (e) return fail(c, KLT_ERR_DEVICE, std::string("nms launch: ") + hipGetErrorString((hipError_t)e));
        launch_mis_results(c->stream, c->readback, rem_d, j.look, j.info_d, c->placed_d);
    }
    // the host's look waits for THIS point of the stream, not for the stream: a caller may enqueue work that only reads the list (the next
    // frame's tracker) between the two halves, and the GPU keeps it queued while the host looks and enqueues the next selection
    if (!c->ev_sel) HIPCHK(c, hipEventCreateWithFlags(&c->ev_sel, hipEventDisableTiming));
    HIPCHK(c, hipEventRecord(c->ev_sel, c->stream));
    return 0;
}

// the host's look at the outcome, and whatever it asks for; returns when the selection is complete
int select_job_finish(klt_ctx *c, SelectJob &j)
{
    const unsigned *const rem = c->readback, *const info = c->readback + 64, *const res = c->readback + 72;
    int looks = 0;                       // > 1: the list was rewritten after the launches of klt_select_begin_async had run
    for (;;) {
        HIPCHK(c, hipEventSynchronize(c->ev_sel));
        looks++;
        if (rem[j.look - 1] != 0u) {
            // a dependency chain longer than the passes run so far: put the list back and keep going
            if (j.round + j.rounds_per_look > SelectJob::kMaxRounds) return fail(c, KLT_ERR_DEVICE, "minimum-distance passes did not settle");
            HIPCHK(c, hipMemcpyAsync(j.fl, c->fl_snapshot, (size_t)j.n * sizeof(klt_feat), hipMemcpyDefault /* the list may be pinned host memory (klt_featbuf_map_host) */, c->stream));
            if (j.by_rank) launch_zero_words(c->stream, j.rank_d, (size_t)j.bound);
            else HIPCHK(c, hipMemsetAsync(c->keys2, 0, (size_t)j.np2 * sizeof(unsigned long long), c->stream));
            if (int rc = select_job_rounds(c, j)) return rc;
            continue;
        }
        int needed = j.look;                                         // learn how many passes were needed
        while (needed > 1 && rem[needed - 2] == 0u) needed--;
        needed += j.round - j.look;
        // replacement runs frame after frame: one spare pass, because a frame that needs one pass more than the last
        // one costs a host round trip and another batch of passes, an idle pass 2-5 us
        if (j.mode == KLT_REPLACING_SOME) needed += 1;
        c->mis_rounds_hint = needed < 2 ? 2 : (needed > 32 ? 32 : needed);
        c->sorted_keys = j.by_rank ? nullptr : c->keys2; c->sorted_count = j.by_rank ? 0 : (int)j.np2;
        // ran out of accepted candidates although the prefilter dropped some: repeat with every candidate
        if (j.filtered && res[1] && info[1] < info[2]) {
            HIPCHK(c, hipMemcpyAsync(j.fl, c->fl_snapshot, (size_t)j.n * sizeof(klt_feat), hipMemcpyDefault, c->stream));
            j.attempt = 1;
            if (int rc = select_job_start(c, j)) return rc;
            if (int rc = select_job_rounds(c, j)) return rc;
            continue;
        }
        break;
    }
    if (j.pre) j.pre->gen = 0;                                       // a score set is used once
    HIPCHK(c, hipGetLastError());
    return looks > 1 ? 1 : KLT_OK;
}
}  // namespace

namespace {
// borders / half-windows as ScanImageForGoodFeatures receives them: Python floats truncated to C ints
// (selectGoodFeatures.py:168-169, :215-221, goodFeaturesUtils.pyx:35-37)
struct SelGeom { int bx, by, hw, hh, step, nx, ny; long long ncand, npow2; };
int select_geometry(klt_ctx *c, int nc, int nr, SelGeom *g)
{
    const klt_params &p = c->p;
    double bxd = p.borderx, byd = p.bordery;
    if (bxd < p.window_width / 2.0) bxd = p.window_width / 2.0;
    if (byd < p.window_height / 2.0) byd = p.window_height / 2.0;
    g->bx = (int)bxd; g->by = (int)byd; g->hw = p.window_width / 2; g->hh = p.window_height / 2;
    g->step = p.nSkippedPixels + 1;
    if (g->bx - g->hw - 1 < 0 || g->by - g->hh - 1 < 0)
        return fail(c, KLT_ERR_ARG, "border must be at least window/2 + 1 (the reference reads outside the image otherwise)");
    g->nx = (nc - g->bx > g->bx) ? (nc - 2 * g->bx + g->step - 1) / g->step : 0;
    g->ny = (nr - g->by > g->by) ? (nr - 2 * g->by + g->step - 1) / g->step : 0;
    g->ncand = (long long)g->nx * g->ny;
    g->npow2 = 2048;
    while (g->npow2 < g->ncand) g->npow2 <<= 1;
    if (g->npow2 > (1LL << 30)) return fail(c, KLT_ERR_ARG, "too many candidates");
    return 0;
}

// summed-area tables of the gradient products (goodFeaturesUtils.pyx:49-51): step-synchronous wavefront pipelines (sat_pipeline.hip)
// where whole aligned quads can be moved, else the barrier-coupled kernels of select_kernels.hip

ScoreCache *find_scores(klt_ctx *c, const Slot *s, const SelGeom &g, double min_eig)
{
    if (!s->pyr_valid || !s->gen) return nullptr;
    for (auto &e : c->pre)
        if (e.gen == s->gen && e.nc == s->nc && e.nr == s->nr && e.bx == g.bx && e.by == g.by && e.hw == g.hw && e.hh == g.hh &&
            e.step == g.step && e.nx == g.nx && e.ny == g.ny && e.min_eig == min_eig)
            return &e;
    return nullptr;
}
}  // namespace

extern "C" {


// The half of a selection that depends on the pixels only -- summed-area tables and the eigenvalue of every candidate window
// (goodFeaturesUtils.pyx:17-73) -- for the level-0 images of `slot`, ahead of the selection itself: on the build stream when
// KLT_OPT_BUILD_STREAM is on, where it overlaps the tracker and the minimum-distance passes of the previous frame.
int klt_select_prepare_async(klt_ctx *c, int slot)
{
    if (int rc = check_ready(c)) return rc;
    HIPCHK(c, hipSetDevice(c->device));
    Slot *s;
    if (int rc = get_slot(c, slot, &s, false)) return rc;
    if (!s->pyr_valid) return fail(c, KLT_ERR_STATE, "klt_select_prepare_async: the slot's pyramids are not built");
    const int nc = s->nc, nr = s->nr;
    const size_t N = (size_t)nc * nr;
    SelGeom g;
    if (int rc = select_geometry(c, nc, nr, &g)) return rc;
    if (g.ncand <= 0) return KLT_OK;
    struct WorkScope {
        klt_ctx *c;
        ~WorkScope() { c->work = c->stream; }
    } work_scope{c};
    if (c->build_stream_on) {
        if (!c->bstream) HIPCHK(c, hipStreamCreateWithFlags(&c->bstream, hipStreamNonBlocking));
        if (!s->built_on_bstream) {             // built on the main stream: behind everything there
            hipEvent_t mark;
            if (int rc = fresh_event(c, &mark)) return rc;
            HIPCHK(c, hipEventRecord(mark, c->stream));
            HIPCHK(c, hipStreamWaitEvent(c->bstream, mark, 0));
        }
        // (the set being replaced was last read by a selection, and selections synchronise the main stream before they return)
        c->work = c->bstream;
    } else {
        if (int rc = wait_built(c, s)) return rc;
        for (auto &e : c->pre)                  // an earlier preparation on the build stream shares the table scratch
            if (e.ev && event_live(c, e.ev_serial)) HIPCHK(c, hipStreamWaitEvent(c->stream, e.ev, 0));
    }
    // the set that already belongs to these contents, else a free one, else the oldest
    ScoreCache *e = nullptr;
    for (auto &x : c->pre) if (x.gen == s->gen) { e = &x; break; }
    if (!e) for (auto &x : c->pre) if (!x.gen) { e = &x; break; }
    if (!e) {
        const ScoreCache *busy = c->sel_job ? c->sel_job->pre : nullptr;      // (a pending selection still reads its set)
        for (auto &x : c->pre) if (&x != busy && (!e || x.stamp < e->stamp)) e = &x;
    }
    if (int rc = ensure(c, c->sat_pre, c->sat_pre_cap, 3 * N + KLT_SAT_PAD)) return rc;
    if (int rc = ensure(c, e->keys, e->cap, (size_t)g.npow2)) return rc;
    e->gen = 0;
    SelectArgs sa;
    std::memset(&sa, 0, sizeof(sa));
    sa.sat = c->sat_pre; sa.keys = e->keys;
    sa.min_eig = c->p.min_eigenvalue < 1 ? 1.0 : c->p.min_eigenvalue;          // selectGoodFeatures.py:53
    sa.ncols = nc; sa.nrows = nr; sa.bx = g.bx; sa.by = g.by; sa.step = g.step; sa.nx = g.nx; sa.ny = g.ny;
    sa.hw = g.hw; sa.hh = g.hh; sa.npow2 = (int)g.npow2;
    // the tables' column pass and the eigenvalue keys in one launch where that applies (sat_pipeline.hip: the column-summed tables never
    // reach HBM); KLT_FUSED_COLS_EIGEN=0: the two separate kernels
    static const bool fused_cols_eigen = !(getenv("KLT_FUSED_COLS_EIGEN") && atoi(getenv("KLT_FUSED_COLS_EIGEN")) == 0);
    if (fused_cols_eigen && c->sat_variant == 1 && sat_cols_eigen_ok(sa)) {
        if (int rc = enqueue_sat(c, c->work, s->lv[0].gx, s->lv[0].gy, c->sat_pre, nc, nr, true)) return rc;
        int e2;
        { TimerScope t(c, F_EIGEN, 12.0 * N + (double)g.ncand * 8);
          e2 = launch_sat_cols_eigen_pipe(c->work, c->sat_pre, sa); }
        if (e2 > 0) return fail(c, KLT_ERR_DEVICE, std::string("column pass + eigenvalue keys: ") + hipGetErrorString((hipError_t)e2));
        if (e2 < 0) {
            // the fused kernel does not take this geometry after all: the column pass and the keys as two launches (no keys were written,
            // and the score set is only stamped below, after a launch that did write them)
            { TimerScope t(c, F_SAT_COLS, N * 24);
              const int e3 = launch_sat_cols_pipe(c->work, c->sat_pre, nc, nr);
              if (e3 > 0) return fail(c, KLT_ERR_DEVICE, hipGetErrorString((hipError_t)e3));
              if (e3 < 0) launch_sat_cols(c->work, c->sat_pre, nc, nr); }
            TimerScope t(c, F_EIGEN, 12.0 * N + (double)g.ncand * 8);
            launch_eigen_hist(c->work, sa);
        }
    } else {
        if (int rc = enqueue_sat(c, c->work, s->lv[0].gx, s->lv[0].gy, c->sat_pre, nc, nr)) return rc;
        TimerScope t(c, F_EIGEN, 12.0 * N + (double)g.ncand * 8);
        launch_eigen_hist(c->work, sa);
    }
    if (int rc = fresh_event(c, &e->ev, &e->ev_serial)) return rc;
    HIPCHK(c, hipEventRecord(e->ev, c->work));
    e->gen = s->gen; e->stamp = ++c->pre_stamp;
    e->nc = nc; e->nr = nr; e->bx = g.bx; e->by = g.by; e->hw = g.hw; e->hh = g.hh; e->step = g.step; e->nx = g.nx; e->ny = g.ny;
    e->min_eig = sa.min_eig;
    HIPCHK(c, hipGetLastError());
    return KLT_OK;
}

int klt_select_begin_async(klt_ctx *c, int slot, int mode, int use_pyramid, int fb, int n)
{
    if (int rc = check_ready(c)) return rc;
    if (c->sel_job) return fail(c, KLT_ERR_STATE, "a selection is pending: klt_select_finish first");
    if (mode != KLT_SELECTING_ALL && mode != KLT_REPLACING_SOME) return fail(c, KLT_ERR_ARG, "bad selection mode");
    if (n <= 0) return fail(c, KLT_ERR_ARG, "nFeatures must be positive");
    HIPCHK(c, hipSetDevice(c->device));
    Slot *s;
    if (int rc = get_slot(c, slot, &s, false)) return rc;
    const int nc = s->nc, nr = s->nr;
    const size_t N = (size_t)nc * nr;
    FeatBuf *b;
    if (int rc = get_fb(c, fb, n, &b)) return rc;

    const klt_params &p = c->p;
    SelGeom geom;
    if (int rc = select_geometry(c, nc, nr, &geom)) return rc;
    const int bx = geom.bx, by = geom.by, hw = geom.hw, hh = geom.hh, step = geom.step, nx = geom.nx, ny = geom.ny;
    const long long ncand = geom.ncand, npow2 = geom.npow2;

    // scratch
    if (N > c->sel_cap) {
        if (c->sel_img) { if (int rc = sync_all(c)) return rc; hipFree(c->sel_img); hipFree(c->sel_gx); hipFree(c->sat); hipFree(c->valmap); }
        c->sel_img = c->sel_gx = c->sel_gy = c->sat = c->valmap = nullptr;
        c->sel_cap = 0;                                       // (nothing is held until all four planes are: a failure below frees what it got)
        int rc_alloc = dev_alloc(c, (void **)&c->sel_img, N * sizeof(float), "selection scratch: image");
        if (!rc_alloc) rc_alloc = dev_alloc(c, (void **)&c->sel_gx, KLT_GRAD_STRIDE * N * sizeof(float), "selection scratch: gradients");      // gradx / grady interleaved, like a slot's planes
        if (!rc_alloc) rc_alloc = dev_alloc(c, (void **)&c->sat, 3 * N * sizeof(float), "selection scratch: summed-area tables");
        if (!rc_alloc) rc_alloc = dev_alloc(c, (void **)&c->valmap, N * sizeof(float), "selection scratch: eigenvalue map");
        if (rc_alloc) {
            hipFree(c->sel_img); hipFree(c->sel_gx); hipFree(c->sat); hipFree(c->valmap);
            c->sel_img = c->sel_gx = c->sel_gy = c->sat = c->valmap = nullptr;
            return rc_alloc;
        }
        c->sel_gy = c->sel_gx + 1;
        c->sel_cap = N;
    }
    if (int rc = ensure(c, c->keys, c->keys_cap, (size_t)npow2)) return rc;
    if (int rc = ensure_tmp(c, N)) return rc;

    // images: reuse the slot's level-0 pyramid (selectGoodFeatures.py:176-181) or compute afresh (:183-197)
    const float *img, *gx, *gy;
    if (use_pyramid) {
        if (!s->pyr_valid) return fail(c, KLT_ERR_STATE, "use_pyramid requested but the slot's pyramids are not built");
        if (int rc = wait_built(c, s)) return rc;
        img = s->lv[0].img; gx = s->lv[0].gx; gy = s->lv[0].gy;
    } else {
        if (s->raw_kind == 0) return fail(c, KLT_ERR_STATE, "slot has no frame");
        if (int rc = wait_upload(c, s, c->stream)) return rc;
        if (int rc = wait_built(c, s)) return rc;              // a build of this slot may still read the raw frame's buffers
        bool grads_done = false;
        if (p.smoothBeforeSelecting && fused_smooth_ok(c)) {
            const void *raw = s->raw_kind == 1 ? (const void *)raw8(s) : (const void *)rawf(s);
            if (int rc = enqueue_fused_smooth_grad(c, 1, &raw, s->raw_kind, &c->sel_img, &c->sel_gx, &c->sel_gy, nc, nr)) return rc;
            img = c->sel_img;
            grads_done = true;
        } else if (p.smoothBeforeSelecting) {
            enqueue_smooth_raw(c, s, c->sel_img);
            img = c->sel_img;
        } else if (s->raw_kind == 2) {
            img = rawf(s);
        } else {
            // u8 -> f32 with a 1-tap identity kernel is overkill; widen with a 1-tap correlate (exact)
            Taps one;
            std::memset(&one, 0, sizeof(one));
            one.n = 1; one.sym = 1; one.k[0] = 1.0;
            launch_hconv_u8(c->stream, raw8(s), nc, nr, c->sel_img, nullptr, nc, 1, 0, one, nullptr);
            img = c->sel_img;
        }
        if (!grads_done) {
            if (fused_grad_ok(c)) { if (int rc = enqueue_fused_grad(c, 1, &img, &c->sel_gx, &c->sel_gy, nc, nr)) return rc; }
            else enqueue_gradients(c, img, nc, nr, c->sel_gx, c->sel_gy);
        }
        gx = c->sel_gx; gy = c->sel_gy;
        // the kernels above read the raw frame: the second-next asynchronous copy into this slot (its raw buffers alternate) waits
        if (int rc = mark_consumed(c, &s, 1, c->stream)) return rc;
    }
    c->last_sel[0] = img; c->last_sel[1] = gx; c->last_sel[2] = gy;
    c->sel_nc = nc; c->sel_nr = nr; c->sel_nx = nx; c->sel_ny = ny; c->sel_npow2 = (int)npow2;

    int mindist = p.mindist < 0 ? 0 : p.mindist;          // selectGoodFeatures.py:241-243
    const int d = mindist - 1;                            // :61
    const int R = d >= 0 ? d / step : -1;                 // exclusion radius in candidate cells
    const bool parallel_nms = c->use_mis && ncand > 0 && mis_stage_bytes(R) <= 120 * 1024;
    long long target = 64LL * n;
    if (target < 65536) target = 65536;
    const bool prefilter = c->use_topk && ncand > 262144 && target < ncand / 2;
    const double min_eig = p.min_eigenvalue < 1 ? 1.0 : p.min_eigenvalue;          // :53

    // scores prepared ahead of time (klt_select_prepare_async) are used by the replacement pass of the parallel path; everything else
    // computes them here
    ScoreCache *pre = nullptr;
    if (mode == KLT_REPLACING_SOME && use_pyramid && parallel_nms && prefilter && d >= 0 && !c->score_override_n)
        pre = find_scores(c, s, geom, min_eig);
    c->sel_valmap = !pre;
    struct Consume {                                       // a set is used once: the selection frees it when it is through with it
        ScoreCache *e;
        ~Consume() { if (e) e->gen = 0; }
    } consume{pre};
    if (pre) {
        if (event_live(c, pre->ev_serial)) HIPCHK(c, hipStreamWaitEvent(c->stream, pre->ev, 0));
        else if (c->bstream && !c->capturing) HIPCHK(c, hipStreamSynchronize(c->bstream));
    } else {
        // summed-area tables (goodFeaturesUtils.pyx:49-51)
        if (int rc = enqueue_sat(c, c->stream, gx, gy, c->sat, nc, nr)) return rc;
    }

    const uint8_t *seed = nullptr;
    // REPLACING_SOME: the squares of the live features are marked first; the eigenvalue kernels skip marked pixels, so neither the
    // scoring nor the minimum-distance stage ever sees them
    if (mode == KLT_REPLACING_SOME && d >= 0) {
        const uint8_t *before = c->seedmap;
        if (int rc = ensure(c, c->seedmap, c->seed_cap, N)) return rc;
        if (c->seedmap != before || c->seed_n != N || c->seed_stamp == 255) {       // new map, other frame size, or the stamps wrapped
            HIPCHK(c, hipMemsetAsync(c->seedmap, 0, N, c->stream));
            c->seed_n = N;
            c->seed_stamp = 0;
        }
        c->seed_stamp++;
        TimerScope t(c, F_SEED, (double)n * 16);
        launch_seed_fill(c->stream, b->d, n, c->seedmap, nc, nr, d, c->seed_stamp);
        seed = c->seedmap;
    }

    SelectArgs sa;
    sa.sat = c->sat; sa.valmap = c->valmap; sa.keys = c->keys; sa.seedmap = seed; sa.seed_stamp = c->seed_stamp;
    sa.val_in = nullptr;
    sa.hist = sa.ticket = sa.info = nullptr; sa.hist_target = 0; sa.hist_slots = nullptr; sa.hist_per_slot = 0;
    if (c->score_override_n) {
        const int given = c->score_override_n;
        c->score_override_n = 0;
        if (given != ncand) return fail(c, KLT_ERR_ARG, "score override does not match the candidate grid");
        sa.val_in = c->score_override;
    }
    sa.min_eig = min_eig;
    sa.ncols = nc; sa.nrows = nr; sa.bx = bx; sa.by = by; sa.step = step; sa.nx = nx; sa.ny = ny;
    sa.hw = hw; sa.hh = hh; sa.npow2 = (int)npow2;
    if (!parallel_nms) { TimerScope t(c, F_EIGEN, 12.0 * N + (double)ncand * (4 + 8)); launch_eigen(c->stream, sa); }
    NmsArgs na;
    std::memset(&na, 0, sizeof(na));
    na.fl = b->d; na.placed_out = c->placed_d;
    na.nfeat = n; na.overwrite_all = (mode == KLT_SELECTING_ALL);
    na.d = d; na.cell = d >= 0 ? d + 1 : 1;
    na.cell_magic = na.cell == 1 ? 0u : (unsigned)((1ull << 32) / (unsigned)na.cell) + 1u;
    if (int rc = ensure(c, c->nms_slots, c->nms_slots_cap, (size_t)n)) return rc;
    na.slots = c->nms_slots;
    na.aff_rec = nullptr;
    if (c->select_aff_state >= 0) {
        AffState &as = c->aff[c->select_aff_state];
        if (as.n < n) return fail(c, KLT_ERR_STATE, "affine state smaller than the feature list");
        na.aff_rec = as.rec;
    }
    na.gw = (nc + na.cell - 1) / na.cell; na.gh = (nr + na.cell - 1) / na.cell;
    if (d < 0) { na.gw = na.gh = 1; }
    const size_t grid_bytes = (size_t)na.gw * na.gh * sizeof(uint32_t);
    na.grid_in_lds = grid_bytes <= 128 * 1024;     // + 12.4 KiB of static LDS in the kernel
    na.grid_global = nullptr;
    if (!na.grid_in_lds) {
        if (int rc = ensure(c, c->grid, c->grid_cap, (size_t)na.gw * na.gh)) return rc;
        na.grid_global = c->grid;
    }
    auto run_nms = [&](const unsigned long long *keys, int nkeys) -> int {
        na.keys = keys;
        na.nkeys = nkeys;
        if (!na.grid_in_lds) HIPCHK(c, hipMemsetAsync(c->grid, 0, grid_bytes, c->stream));
        TimerScope t(c, F_NMS, (double)n * 16);
        const int e = launch_nms(c->stream, na);
        if (e) return fail(c, KLT_ERR_DEVICE, std::string("nms launch: ") + hipGetErrorString((hipError_t)e));
        return 0;
    };

    // ---- parallel minimum distance (default): decide every candidate in a few passes, rank the accepted ones, and
    // fill the free slots with the best of them (same result as the sorted serial walk below)
    if (parallel_nms) {
        auto job = std::make_unique<SelectJob>();
        SelectJob &j = *job;
        // passes enqueued before the host looks at the outcome: what the previous selection needed (frames of a sequence
        // behave alike); an idle pass costs 5 us, a second look costs a host round trip
        j.rounds_per_look = c->mis_rounds_hint;
        const int tiles = mis_tiles(nx, ny);
        // two accepted candidates are more than R cells apart in x or in y: at most one per (R+1)x(R+1) block of cells
        j.bound = R < 0 ? ncand : (long long)((nx + R) / (R + 1)) * ((ny + R) / (R + 1));
        j.by_rank = j.bound <= 98304;                       // rank by counting; beyond that sort the accepted keys
        j.np2 = 2048;
        while (j.np2 < j.bound) j.np2 <<= 1;
        // one allocation of counters: [tiles] list lengths | [kMaxRounds] "undecided left after pass r" | accepted count |
        // workgroup ticket | [tiles] accepted per tile | 8192 histogram bins + 4 words of prefilter info | [bound] ranks
        const size_t off_rem = (size_t)tiles, off_acc = off_rem + SelectJob::kMaxRounds, off_ticket = off_acc + 1, off_tacc = off_ticket + 1,
                     off_hist = off_tacc + tiles, off_rank = off_hist + 8192 + 4;
        const int tile_cap = mis_tile_capacity(R);
        const size_t n_cnt = off_rank + (j.by_rank ? (size_t)j.bound : 0);
        if (int rc = ensure(c, c->keys2, c->keys2_cap, (size_t)(j.np2 > npow2 ? j.np2 : npow2))) return rc;
        if (int rc = ensure(c, c->mis_st, c->mis_st_cap, (size_t)ncand)) return rc;
        if (int rc = ensure(c, c->mis_list, c->mis_list_cap, (size_t)tiles * 1024)) return rc;
        if (int rc = ensure(c, c->mis_cnt, c->mis_cnt_cap, n_cnt)) return rc;
        if (int rc = ensure(c, c->mis_tile_keys, c->mis_tile_keys_cap, (size_t)tiles * tile_cap)) return rc;
        if (int rc = ensure(c, c->fl_snapshot, c->fl_snapshot_cap, (size_t)n)) return rc;
        // results come back through pinned host memory the kernels write to directly
        if (!c->readback) {
            void *hp = nullptr;
            if (int rc = host_alloc(c, &hp, 128 * sizeof(unsigned), "selection read-back words")) return rc;
            c->readback = (unsigned *)hp;
            c->pinned.push_back(hp);
        }
        j.fl = b->d; j.n = n; j.ncand = ncand; j.mode = mode; j.prefilter = prefilter; j.target = target; j.pre = pre;
        j.zero_from = c->mis_cnt + off_rem; j.zero_n = n_cnt - off_rem;
        j.hist_d = c->mis_cnt + off_hist; j.ticket_d = c->mis_cnt + off_ticket;
        j.info_d = c->mis_cnt + off_hist + 8192; j.rank_d = c->mis_cnt + off_rank;
        j.acc_count_d = c->mis_cnt + off_acc;
        j.nfill_d = c->placed_d + 2;
        MisArgs &ma = j.ma;
        ma.keys = pre ? pre->keys : c->keys; ma.seed = pre ? seed : nullptr; ma.seed_stamp = c->seed_stamp; ma.ncols = nc; ma.st = c->mis_st; ma.list = c->mis_list; ma.cnt = c->mis_cnt;
        ma.remaining = c->mis_cnt + off_rem; ma.acc_cnt = c->mis_cnt + off_tacc; ma.acc_cap = tile_cap;
        ma.acc_keys = c->mis_tile_keys; ma.info = j.info_d;
        ma.nx = nx; ma.ny = ny; ma.R = R; ma.stage = 1; ma.bx = bx; ma.by = by; ma.step = step;
        ma.sparse = mode == KLT_REPLACING_SOME && prefilter ? 1 : 0;
        j.pa = na;                                          // placement: the accepted candidates never exclude each other
        j.pa.d = -1; j.pa.cell = 1; j.pa.cell_magic = 0u; j.pa.gw = j.pa.gh = 1; j.pa.grid_in_lds = 1; j.pa.grid_global = nullptr;
        j.pa.keys = c->keys2; j.pa.nkeys = (int)j.np2;
        j.sa = sa;
        consume.e = nullptr;                                // the job frees the score set when it is through with it
        if (int rc = select_job_start(c, j)) return rc;
        if (int rc = select_job_rounds(c, j)) return rc;
        c->sel_job = std::move(job);
        HIPCHK(c, hipGetLastError());
        return KLT_OK;
    }

    // ---- sorted serial walk (KLT_OPT_SELECT_PARALLEL_NMS = 0, or an exclusion square too large for the LDS tile)
    // top-K prefilter: sort only the candidates the greedy walk can plausibly reach (one small D2H read-back)
    if (prefilter) {
        if (int rc = ensure(c, c->keys2, c->keys2_cap, (size_t)npow2)) return rc;
        size_t hcap = c->topk_hist ? 8192 + 4 : 0;
        if (int rc = ensure(c, c->topk_hist, hcap, (size_t)8192 + 4)) return rc;
        if (int rc = ensure(c, c->fl_snapshot, c->fl_snapshot_cap, (size_t)n)) return rc;
        HIPCHK(c, hipMemsetAsync(c->topk_hist, 0, (8192 + 4) * sizeof(unsigned), c->stream));
        unsigned info[4] = {0, 0, 0, 0};
        {
            TimerScope t(c, F_SORT, (double)ncand * 16);
            launch_topk_prefilter(c->stream, c->keys, (int)ncand, (unsigned)target, c->topk_hist, c->topk_hist + 8192, c->keys2);
        }
        HIPCHK(c, hipMemcpyAsync(info, c->topk_hist + 8192, sizeof(info), hipMemcpyDeviceToHost, c->stream));
        HIPCHK(c, hipStreamSynchronize(c->stream));
        const long long kept = info[3], valid = info[2];
        long long np2 = 2048;
        while (np2 < kept) np2 <<= 1;
        if (kept < np2) HIPCHK(c, hipMemsetAsync(c->keys2 + kept, 0, (size_t)(np2 - kept) * sizeof(unsigned long long), c->stream));
        { TimerScope t(c, F_SORT, (double)np2 * 16); launch_sort_desc(c->stream, c->keys2, (int)np2); }
        HIPCHK(c, hipMemcpyAsync(c->fl_snapshot, b->d, (size_t)n * sizeof(klt_feat), hipMemcpyDefault, c->stream));
        if (int rc = run_nms(c->keys2, (int)kept)) return rc;
        c->sorted_keys = c->keys2; c->sorted_count = (int)kept;
        int res[2] = {0, 0};
        HIPCHK(c, hipMemcpyAsync(res, c->placed_d, sizeof(res), hipMemcpyDeviceToHost, c->stream));
        HIPCHK(c, hipStreamSynchronize(c->stream));
        if (!(res[1] && kept < valid)) { HIPCHK(c, hipGetLastError()); return KLT_OK; }
        // the kept candidates ran out before the list was full: restore the list and take the full sort
        HIPCHK(c, hipMemcpyAsync(b->d, c->fl_snapshot, (size_t)n * sizeof(klt_feat), hipMemcpyDefault, c->stream));
    }
    { TimerScope t(c, F_SORT, (double)npow2 * 16); launch_sort_desc(c->stream, c->keys, (int)npow2); }
    if (int rc = run_nms(c->keys, (int)(ncand < npow2 ? ncand : npow2))) return rc;
    c->sorted_keys = c->keys; c->sorted_count = (int)(ncand < npow2 ? ncand : npow2);
    HIPCHK(c, hipGetLastError());
    return KLT_OK;
}

int klt_select_finish(klt_ctx *c)
{
    if (!c) return KLT_ERR_ARG;
    if (!c->sel_job) return KLT_OK;                       // nothing pending (or a path that completes in klt_select_begin_async)
    HIPCHK(c, hipSetDevice(c->device));
    std::unique_ptr<SelectJob> job = std::move(c->sel_job);
    return select_job_finish(c, *job);
}

int klt_select_async(klt_ctx *c, int slot, int mode, int use_pyramid, int fb, int n)
{
    if (int rc = klt_select_begin_async(c, slot, mode, use_pyramid, fb, n)) return rc;
    const int rc = klt_select_finish(c);
    return rc > 0 ? KLT_OK : rc;
}

int klt_select(klt_ctx *c, int slot, int mode, int use_pyramid, klt_feat *inout, int n, int *n_placed)
{
    if (!c || !inout) return fail(c, KLT_ERR_ARG, "null argument");
    const int fb = 65535;       // private staging buffer
    if (int rc = klt_featbuf_upload(c, fb, inout, n)) return rc;
    if (int rc = klt_select_async(c, slot, mode, use_pyramid, fb, n)) return rc;
    if (int rc = klt_featbuf_download(c, fb, inout, n)) return rc;
    if (n_placed) {
        HIPCHK(c, hipMemcpy(n_placed, c->placed_d, sizeof(int), hipMemcpyDeviceToHost));
    }
    return KLT_OK;
}

// _enforceMinimumDistance (selectGoodFeatures.py:45-135) as the reference's callers may use it on its own: the greedy walk over a
// GIVEN candidate list in the GIVEN order (keys as klt_download_sorted_candidates describes them: f32 bits of val << 32 | x << 16 | y;
// the caller has dropped the candidates the walk would skip without effect -- val below min_eigenvalue, positions inside the squares of
// live features when these are kept), filling the list's free slots: every slot in rank order when overwrite_all, the lost ones otherwise.
int klt_min_distance_walk(klt_ctx *c, const uint64_t *keys, int nkeys, int ncols, int nrows, int mindist, int overwrite_all,
                          klt_feat *inout, int n, int *n_placed)
{
    if (!c || !inout || (!keys && nkeys > 0)) return fail(c, KLT_ERR_ARG, "null argument");
    if (nkeys < 0 || n <= 0 || ncols <= 0 || nrows <= 0 || ncols > 65535 || nrows > 65535) return fail(c, KLT_ERR_ARG, "bad argument");
    if (c->sel_job) return fail(c, KLT_ERR_STATE, "a selection is pending: klt_select_finish first");
    // every candidate inside the image (the reference asserts the same when the walk reaches the point, selectGoodFeatures.py:90-91): the
    // kernel marks accepted candidates in a grid of ncols x nrows cells and checks nothing, a position outside would be a write outside it
    for (int i = 0; i < nkeys; i++) {
        if (keys[i] == 0ull) return fail(c, KLT_ERR_ARG, "a zero key (value 0.0 at (0, 0)) is the walk's end mark, not a candidate");
        const int x = (int)((keys[i] >> 16) & 0xffffull), y = (int)(keys[i] & 0xffffull);
        if (x >= ncols || y >= nrows) {
            char msg[128];
            snprintf(msg, sizeof msg, "candidate %d at (%d, %d) lies outside the %d x %d image", i, x, y, ncols, nrows);
            return fail(c, KLT_ERR_ARG, msg);
        }
    }
    HIPCHK(c, hipSetDevice(c->device));
    const int fb = 65535;                                  // the synchronous entry points' staging buffer
    if (int rc = klt_featbuf_upload(c, fb, inout, n)) return rc;
    if (int rc = ensure(c, c->keys2, c->keys2_cap, (size_t)nkeys + 1)) return rc;
    if (nkeys) HIPCHK(c, hipMemcpyAsync(c->keys2, keys, (size_t)nkeys * sizeof(uint64_t), hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipMemsetAsync(c->keys2 + nkeys, 0, sizeof(uint64_t), c->stream));            // a zero key ends the walk
    NmsArgs na;
    std::memset(&na, 0, sizeof(na));
    const int d = (mindist < 0 ? 0 : mindist) - 1;        // :61 (and :241-243 for a negative minimum distance)
    na.fl = c->fbs[fb].d; na.placed_out = c->placed_d;
    na.nfeat = n; na.overwrite_all = overwrite_all != 0;
    na.d = d; na.cell = d >= 0 ? d + 1 : 1;
    na.cell_magic = na.cell == 1 ? 0u : (unsigned)((1ull << 32) / (unsigned)na.cell) + 1u;
    if (int rc = ensure(c, c->nms_slots, c->nms_slots_cap, (size_t)n)) return rc;
    na.slots = c->nms_slots;
    na.gw = d >= 0 ? (ncols + na.cell - 1) / na.cell : 1;
    na.gh = d >= 0 ? (nrows + na.cell - 1) / na.cell : 1;
    const size_t grid_bytes = (size_t)na.gw * na.gh * sizeof(uint32_t);
    na.grid_in_lds = grid_bytes <= 128 * 1024;
    if (!na.grid_in_lds) {
        if (int rc = ensure(c, c->grid, c->grid_cap, (size_t)na.gw * na.gh)) return rc;
        na.grid_global = c->grid;
        HIPCHK(c, hipMemsetAsync(c->grid, 0, grid_bytes, c->stream));
    }
    na.keys = c->keys2; na.nkeys = nkeys + 1;
    if (const int e = launch_nms(c->stream, na)) return fail(c, KLT_ERR_DEVICE, std::string("nms launch: ") + hipGetErrorString((hipError_t)e));
    c->sorted_keys = nullptr; c->sorted_count = 0;
    if (int rc = klt_featbuf_download(c, fb, inout, n)) return rc;
    if (n_placed) HIPCHK(c, hipMemcpy(n_placed, c->placed_d, sizeof(int), hipMemcpyDeviceToHost));
    return KLT_OK;
}


int klt_select_dims(klt_ctx *c, int what, int *ncols, int *nrows)
{
    if (!c || what < 0 || what > 3) return fail(c, KLT_ERR_ARG, "bad argument");
    if (c->sel_nc == 0) return fail(c, KLT_ERR_STATE, "no selection has run");
    if (ncols) *ncols = what == 3 ? c->sel_nx : c->sel_nc;
    if (nrows) *nrows = what == 3 ? c->sel_ny : c->sel_nr;
    return KLT_OK;
}

int klt_download_select_f32(klt_ctx *c, int what, float *dst)
{
    if (!c || !dst || what < 0 || what > 3) return fail(c, KLT_ERR_ARG, "bad argument");
    if (c->sel_nc == 0) return fail(c, KLT_ERR_STATE, "no selection has run");
    if (what == 3 && !c->sel_valmap) return fail(c, KLT_ERR_STATE, "the last selection used prepared scores: no eigenvalue map was written");
    const float *src = what == 3 ? c->valmap : c->last_sel[what];
    const size_t cnt = what == 3 ? (size_t)c->sel_nx * c->sel_ny : (size_t)c->sel_nc * c->sel_nr;
    HIPCHK(c, hipSetDevice(c->device));
    return download_plane(c, src, (what == 1 || what == 2) ? KLT_GRAD_STRIDE : 1, cnt, dst);
}

int klt_set_score_override(klt_ctx *c, const float *val, int count)
{
    if (!c || !val || count <= 0) return fail(c, KLT_ERR_ARG, "bad argument");
    HIPCHK(c, hipSetDevice(c->device));
    if (int rc = ensure(c, c->score_override, c->score_override_cap, (size_t)count)) return rc;
    HIPCHK(c, hipMemcpyAsync(c->score_override, val, (size_t)count * sizeof(float), hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    c->score_override_n = count;
    return KLT_OK;
}

int klt_download_sorted_candidates(klt_ctx *c, float *val, int32_t *x, int32_t *y, int n, int *n_valid)
{
    if (!c || !val || !x || !y || n < 0) return fail(c, KLT_ERR_ARG, "bad argument");
    if (c->sel_nc == 0) return fail(c, KLT_ERR_STATE, "no selection has run");
    if (!c->sorted_keys) return fail(c, KLT_ERR_STATE, "the last selection kept no sorted candidate list (KLT_OPT_SELECT_PARALLEL_NMS = 0 keeps one)");
    if (n > c->sorted_count) n = c->sorted_count;
    std::vector<unsigned long long> h((size_t)(n > 0 ? n : 1));
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipMemcpyAsync(h.data(), c->sorted_keys, (size_t)n * sizeof(unsigned long long), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    int k = 0;
    for (; k < n && h[k] != 0ull; k++) {
        const uint32_t bits = (uint32_t)(h[k] >> 32);
        std::memcpy(&val[k], &bits, 4);
        x[k] = (int32_t)((h[k] >> 16) & 0xffffull);
        y[k] = (int32_t)(h[k] & 0xffffull);
    }
    if (n_valid) *n_valid = k;
    return KLT_OK;
}


}  // extern "C"
