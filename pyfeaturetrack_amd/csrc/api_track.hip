// api_track.hip -- the tracker (trackFeatures.py:205-409): single-pair and batched launches with their XCD-aware feature orders, the
// affine consistency check and its per-feature state, the iteration counters behind the roofline figures.
#include "klt_context.h"

extern "C" {

static void fill_track_params(const klt_ctx *c, const Slot *s1, TrackArgs &a, int n)
{
    const klt_params &p = c->p;
    a.half_window = p.window_width / 2.0;
    a.borderx = p.borderx; a.bordery = p.bordery;
    a.n = n; a.nlevels = s1->nlev; a.window = p.window_width; a.max_iterations = p.max_iterations;
    a.use_max_residue = p.use_max_residue; a.retain = p.retainTrackers; a.ncols = s1->nc; a.nrows = s1->nr;
    a.small = p.min_determinant; a.th = p.min_displacement; a.step = p.step_factor; a.max_residue = p.max_residue;
    a.ss = (float)s1->ss;
    a.inv_ss = 1.0f / (float)s1->ss;
    a.tree_sums = c->track_tree_sums ? 1 : 0;
}

// XCD-aware feature order (KLT_OPT_TRACK_XCD_ORDER): one permutation of 0..n-1 per pair of the launch.  It is only a locality
// hint (any permutation tracks every feature exactly once), so it is recomputed when the shape of the launch changes and every
// 64th launch (a sequence's features drift, and its lists alternate between two buffers); in between the stored one is reused.
// The orders are kept per set of INPUT lists (at most kBatchOrders sets, least recently used one replaced): a caller that rotates
// through many resident pairs -- each with its own list -- finds every list's own order again instead of tracking pair B in the
// order of pair A's rows.  A single-pair launch on a list seen for the FIRST time takes the context's shared order instead (refreshed every
// 64 launches): the rows of a sequence's feature table are all new buffers holding nearly the same
// positions, and an order kernel per frame would buy nothing; the second launch on the same buffer gives it its own entry.
static int set_track_order(klt_ctx *c, TrackArgs &a, int n, const std::vector<const klt_feat *> &ins)
{
    if (!c->track_xcd_order || n < 64) return 0;
    const int npairs = (int)ins.size();
    klt_ctx::BatchOrder *bo = nullptr;
    for (auto &e : c->batch_orders)
        if (e.in == ins) { bo = &e; break; }
    if (npairs == 1) {
        bool seen = false;
        for (const klt_feat *p : c->seen_once) seen = seen || p == ins[0];
        if (!bo && !seen) {
            if (c->seen_once.size() >= 256) c->seen_once.erase(c->seen_once.begin());
            c->seen_once.push_back(ins[0]);
            bo = &c->shared_order;
        }
    }
    if (!bo) {
        if (c->batch_orders.size() < klt_ctx::kBatchOrders) {
            c->batch_orders.emplace_back();
            bo = &c->batch_orders.back();
        } else {
            bo = &c->batch_orders[0];
            for (auto &e : c->batch_orders)
                if (e.used < bo->used) bo = &e;
        }
        bo->in = ins;
        bo->n = -1;
    }
    bo->used = ++c->batch_clock;
    const size_t cap_before = bo->cap;
    if (int rc = ensure(c, bo->order, bo->cap, (size_t)n * npairs)) return rc;
    if (bo->cap != cap_before) bo->n = -1;               // a new buffer holds no order yet
    a.order = bo->order;
    a.order_chunk = (n + 7) / 8;
    a.order_refresh = (bo->n != n || bo->age >= 64) ? 1 : 0;
    if (a.order_refresh) { bo->n = n; bo->age = 0; }
    bo->age++;
    return 0;
}

static int check_pair(klt_ctx *c, int slot1, int slot2, Slot **p1, Slot **p2)
{
    if (int rc = get_slot(c, slot1, p1, false)) return rc;
    if (int rc = get_slot(c, slot2, p2, false)) return rc;
    Slot *s1 = *p1, *s2 = *p2;
    if (!s1->pyr_valid || !s2->pyr_valid) return fail(c, KLT_ERR_STATE, "pyramids of both slots must be built before tracking");
    if (int rc = wait_built(c, s1)) return rc;
    if (int rc = wait_built(c, s2)) return rc;
    if (s1->nc != s2->nc || s1->nr != s2->nr || s1->nlev != s2->nlev || s1->ss != s2->ss)
        return fail(c, KLT_ERR_ARG, "the two frames differ in size");            // trackFeatures.py:156-159, :217
    return 0;
}

static void fill_levels(const Slot *s1, const Slot *s2, TrackLevel *lv)
{
    for (int l = 0; l < s1->nlev; l++) {
        lv[l].i1 = s1->lv[l].img; lv[l].gx1 = s1->lv[l].gx; lv[l].gy1 = s1->lv[l].gy;
        lv[l].i2 = s2->lv[l].img; lv[l].gx2 = s2->lv[l].gx; lv[l].gy2 = s2->lv[l].gy;
        lv[l].nc = s1->lv[l].nc; lv[l].nr = s1->lv[l].nr;
    }
}

int klt_track_async(klt_ctx *c, int slot1, int slot2, int fb_in, int fb_out, int n)
{
    if (int rc = check_ready(c)) return rc;
    if (n < 0) return fail(c, KLT_ERR_ARG, "negative feature count");
    HIPCHK(c, hipSetDevice(c->device));
    Slot *s1, *s2;
    if (int rc = check_pair(c, slot1, slot2, &s1, &s2)) return rc;
    if (fb_in < 0 || (size_t)fb_in >= c->fbs.size() || c->fbs[fb_in].cap < n) return fail(c, KLT_ERR_STATE, "input feature buffer not set");
    FeatBuf *bo;
    if (int rc = get_fb(c, fb_out, n > 0 ? n : 1, &bo)) return rc;
    TrackArgs a;
    std::memset(&a, 0, sizeof(a));
    fill_levels(s1, s2, a.lv);
    a.in = c->fbs[fb_in].d; a.out = bo->d;
    fill_track_params(c, s1, a, n);
    if (int rc = set_track_order(c, a, n, std::vector<const klt_feat *>{a.in})) return rc;
    {
        const double foot = 12.0 * (c->p.window_width + 1) * (c->p.window_width + 1);
        TimerScope t(c, F_TRACK, (double)n * (foot * 2 * s1->nlev + 32), c->stream);   // refined by the caller from klt_track_stats
        if (launch_track(c->stream, a)) return fail(c, KLT_ERR_ARG, "unsupported window size");
    }
    if (c->collect_stats) launch_track_stats(c->stream, a.in, a.out, n, s1->nlev, c->stats_d);
    {
        Slot *both[2] = {s1, s2};
        if (int rc = mark_read(c, both, 2)) return rc;
    }
    HIPCHK(c, hipGetLastError());
    return KLT_OK;
}

int klt_track_batch_async(klt_ctx *c, const int *slot1, const int *slot2, const int *fb_in, const int *fb_out, int npairs, int n)
{
    if (int rc = check_ready(c)) return rc;
    if (!slot1 || !slot2 || !fb_in || !fb_out || npairs <= 0 || npairs > 65535 || n < 0) return fail(c, KLT_ERR_ARG, "bad argument");
    HIPCHK(c, hipSetDevice(c->device));
    std::vector<TrackPairDesc> table((size_t)npairs);
    std::vector<Slot *> used;
    Slot *first = nullptr;
    for (int i = 0; i < npairs; i++) {
        FeatBuf *bo;                      // may grow c->fbs: do it before taking pointers into it
        if (int rc = get_fb(c, fb_out[i], n > 0 ? n : 1, &bo)) return rc;
    }
    for (int i = 0; i < npairs; i++) {
        Slot *s1, *s2;
        if (int rc = check_pair(c, slot1[i], slot2[i], &s1, &s2)) return rc;
        if (!first) first = s1;
        used.push_back(s1);
        used.push_back(s2);
        if (s1->nc != first->nc || s1->nr != first->nr || s1->nlev != first->nlev)
            return fail(c, KLT_ERR_ARG, "all pairs of a batch must have the same frame size");
        if (fb_in[i] < 0 || (size_t)fb_in[i] >= c->fbs.size() || c->fbs[fb_in[i]].cap < n)
            return fail(c, KLT_ERR_STATE, "input feature buffer not set");
        std::memset(&table[i], 0, sizeof(TrackPairDesc));
        fill_levels(s1, s2, table[i].lv);
        table[i].in = c->fbs[fb_in[i]].d;
        table[i].out = c->fbs[fb_out[i]].d;
    }
    // the descriptor table is uploaded only when none of the tables kept on the device holds it (found by hash; at most 256 tables,
    // the least recently used one is replaced).  Pageable source: the runtime stages it before returning; stream order protects the
    // launch that read the replaced table
    uint64_t hash = 1469598103934665603ull;
    {
        const unsigned char *bytes = reinterpret_cast<const unsigned char *>(table.data());
        for (size_t i = 0; i < table.size() * sizeof(TrackPairDesc); i++) hash = (hash ^ bytes[i]) * 1099511628211ull;
    }
    klt_ctx::BatchTable *bt = nullptr;
    for (auto &e : c->batch_tables)
        if (e.hash == hash && e.host.size() == table.size() && std::memcmp(e.host.data(), table.data(), table.size() * sizeof(TrackPairDesc)) == 0) { bt = &e; break; }
    if (!bt) {
        if (c->batch_tables.size() < klt_ctx::kBatchTables) {
            c->batch_tables.emplace_back();
            bt = &c->batch_tables.back();
        } else {
            bt = &c->batch_tables[0];
            for (auto &e : c->batch_tables)
                if (e.used < bt->used) bt = &e;
        }
        if (int rc = ensure(c, bt->dev, bt->cap, (size_t)npairs)) return rc;
        HIPCHK(c, hipMemcpyAsync(bt->dev, table.data(), (size_t)npairs * sizeof(TrackPairDesc), hipMemcpyHostToDevice, c->stream));
        bt->host = table;
        bt->hash = hash;
    }
    bt->used = ++c->batch_clock;
    TrackArgs a;
    std::memset(&a, 0, sizeof(a));
    a.pairs = bt->dev;
    a.npairs = npairs;
    fill_track_params(c, first, a, n);
    {
        // one permutation per pair, kept with the set of input lists (see set_track_order)
        std::vector<const klt_feat *> ins((size_t)npairs);
        for (int i = 0; i < npairs; i++) ins[i] = table[i].in;
        if (int rc = set_track_order(c, a, n, ins)) return rc;
    }
    {
        const double foot = 12.0 * (c->p.window_width + 1) * (c->p.window_width + 1);
        TimerScope t(c, F_TRACK, (double)npairs * n * (foot * 2 * first->nlev + 32), c->stream);
        if (launch_track(c->stream, a)) return fail(c, KLT_ERR_ARG, "unsupported window size");
    }
    if (c->collect_stats)
        for (int i = 0; i < npairs; i++) launch_track_stats(c->stream, table[i].in, table[i].out, n, first->nlev, c->stats_d);
    if (int rc = mark_read(c, used.data(), (int)used.size())) return rc;
    HIPCHK(c, hipGetLastError());
    return KLT_OK;
}

int klt_track(klt_ctx *c, int slot1, int slot2, klt_feat *inout, int n, int *n_tracked)
{
    if (!c || !inout) return fail(c, KLT_ERR_ARG, "null argument");
    const int fi = 65534, fo = 65535;
    if (int rc = klt_featbuf_upload_async(c, fi, inout, n)) return rc;      // (the download below synchronises: `inout` is ours until then)
    if (int rc = klt_track_async(c, slot1, slot2, fi, fo, n)) return rc;
    if (int rc = klt_featbuf_download(c, fo, inout, n)) return rc;
    if (n_tracked) {
        int k = 0;
        for (int i = 0; i < n; i++) k += inout[i].val >= 0;
        *n_tracked = k;
    }
    return KLT_OK;
}

// ------------------------------------------------------------------------- affine consistency check
int klt_set_affine_params(klt_ctx *c, const klt_affine_params *p)
{
    if (!c || !p) return fail(c, KLT_ERR_ARG, "null argument");
    if (p->mode < -1 || p->mode > 2) return fail(c, KLT_ERR_ARG, "affineConsistencyCheck must be -1, 0, 1 or 2");
    if (p->mode >= 0 && (p->window_width < 3 || p->window_height < 3 || !(p->window_width & 1) || !(p->window_height & 1) ||
                         p->window_width > 63 || p->window_height > 63 || p->max_iterations < 1))
        return fail(c, KLT_ERR_ARG, "affine window must be odd, 3..63; max_iterations >= 1");
    if (c->ap.window_width != p->window_width || c->ap.window_height != p->window_height)
        for (AffState &a : c->aff)
            if (a.rec) return fail(c, KLT_ERR_STATE, "affine window cannot change while affine states exist");
    c->ap = *p;
    return KLT_OK;
}

int klt_affine_alloc(klt_ctx *c, int state, int n)
{
    if (!c || state < 0 || state > 4095 || n <= 0) return fail(c, KLT_ERR_ARG, "bad argument");
    HIPCHK(c, hipSetDevice(c->device));
    if ((size_t)state >= c->aff.size()) c->aff.resize(state + 1);
    AffState &a = c->aff[state];
    const int tn = (c->ap.window_width + 2) * (c->ap.window_height + 2);
    if (a.n < n || a.tn != tn) {
        if (a.rec) { if (int rc = sync_all(c)) return rc; hipFree(a.rec); hipFree(a.tpl); a.rec = nullptr; a.tpl = nullptr; a.n = 0; }
        DEVALLOC(c, a.rec, (size_t)n * sizeof(klt_affine_rec));
        if (int rc = dev_alloc(c, (void **)&a.tpl, (size_t)n * 3 * tn * sizeof(float), "affine templates")) {
            hipFree(a.rec);                                   // a state is its records AND its templates, or nothing
            a.rec = nullptr;
            return rc;
        }
        a.n = n;
        a.tn = tn;
    }
    launch_affine_reset(c->stream, a.rec, a.n);
    HIPCHK(c, hipGetLastError());
    return KLT_OK;
}

int klt_affine_free(klt_ctx *c, int state)
{
    if (!c) return KLT_ERR_ARG;
    if (state < 0 || (size_t)state >= c->aff.size() || !c->aff[state].rec) return KLT_OK;
    HIPCHK(c, hipSetDevice(c->device));
    if (int rc = sync_all(c)) return rc;
    AffState &a = c->aff[state];
    hipFree(a.rec); hipFree(a.tpl);
    a = AffState();
    if (c->select_aff_state == state) c->select_aff_state = -1;
    return KLT_OK;
}

// records (and, if asked, templates) of the first n features of `src` into `dst` (allocated here if needed), on the context's stream:
// a snapshot of the per-feature state, e.g. to replay a step of a sequence from the same state
int klt_affine_copy_async(klt_ctx *c, int dst, int src, int n, int with_templates)
{
    if (!c || dst == src || n <= 0) return fail(c, KLT_ERR_ARG, "bad argument");
    if (src < 0 || (size_t)src >= c->aff.size() || !c->aff[src].rec || c->aff[src].n < n)
        return fail(c, KLT_ERR_STATE, "source affine state not allocated (or smaller than requested)");
    if (dst < 0 || (size_t)dst >= c->aff.size() || !c->aff[dst].rec || c->aff[dst].n < n) {
        if (int rc = klt_affine_alloc(c, dst, c->aff[src].n)) return rc;
    }
    HIPCHK(c, hipSetDevice(c->device));
    const AffState &s = c->aff[src];
    AffState &d = c->aff[dst];
    if (d.tn != s.tn) return fail(c, KLT_ERR_STATE, "affine states of different window sizes");
    HIPCHK(c, hipMemcpyAsync(d.rec, s.rec, (size_t)n * sizeof(klt_affine_rec), hipMemcpyDeviceToDevice, c->stream));
    if (with_templates)
        HIPCHK(c, hipMemcpyAsync(d.tpl, s.tpl, (size_t)n * 3 * s.tn * sizeof(float), hipMemcpyDeviceToDevice, c->stream));
    return KLT_OK;
}

int klt_affine_download(klt_ctx *c, int state, klt_affine_rec *dst, int n)
{
    if (!c || !dst || state < 0 || (size_t)state >= c->aff.size() || !c->aff[state].rec || c->aff[state].n < n)
        return fail(c, KLT_ERR_STATE, "affine state not allocated (or smaller than requested)");
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipMemcpyAsync(dst, c->aff[state].rec, (size_t)n * sizeof(klt_affine_rec), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return KLT_OK;
}

int klt_track_affine_async(klt_ctx *c, int slot1, int slot2, int fb_in, int fb_out, int n, int state)
{
    if (!c) return KLT_ERR_ARG;
    if (fb_in == fb_out) return fail(c, KLT_ERR_ARG, "the consistency check needs the records before and after: fb_in != fb_out");
    if (c->ap.mode >= 0 && (state < 0 || (size_t)state >= c->aff.size() || !c->aff[state].rec || c->aff[state].n < n))
        return fail(c, KLT_ERR_STATE, "affine state not allocated (klt_affine_alloc) or smaller than the feature list");
    if (int rc = klt_track_async(c, slot1, slot2, fb_in, fb_out, n)) return rc;
    if (c->ap.mode < 0 || n == 0) return KLT_OK;
    Slot *s1 = &c->slots[slot1], *s2 = &c->slots[slot2];
    AffState &as = c->aff[state];
    AffineArgs a;
    std::memset(&a, 0, sizeof(a));
    a.in = c->fbs[fb_in].d; a.out = c->fbs[fb_out].d; a.rec = as.rec; a.tpl = as.tpl;
    a.i1 = s1->lv[0].img; a.gx1 = s1->lv[0].gx; a.gy1 = s1->lv[0].gy;
    a.i2 = s2->lv[0].img; a.gx2 = s2->lv[0].gx; a.gy2 = s2->lv[0].gy;
    a.n = n; a.ncols = s1->nc; a.nrows = s1->nr; a.mode = c->ap.mode;
    a.width = c->ap.window_width; a.height = c->ap.window_height; a.max_iterations = c->ap.max_iterations;
    a.step = c->p.step_factor; a.small = c->p.min_determinant; a.th = c->p.min_displacement;
    a.th_aff = c->ap.min_displacement; a.max_residue = c->ap.max_residue; a.max_differ = c->ap.max_displacement_differ;
    {
        TimerScope t(c, F_AFFINE, (double)n * 12.0 * (a.width + 1) * (a.height + 1) * 3, c->stream);
        launch_affine(c->stream, a);
    }
    {
        Slot *both[2] = {s1, s2};
        if (int rc = mark_read(c, both, 2)) return rc;
    }
    HIPCHK(c, hipGetLastError());
    return KLT_OK;
}

int klt_track_affine(klt_ctx *c, int slot1, int slot2, klt_feat *inout, int n, int state, int *n_tracked)
{
    if (!c || !inout) return fail(c, KLT_ERR_ARG, "null argument");
    const int fi = 65534, fo = 65535;
    if (int rc = klt_featbuf_upload(c, fi, inout, n)) return rc;
    if (int rc = klt_track_affine_async(c, slot1, slot2, fi, fo, n, state)) return rc;
    if (int rc = klt_featbuf_download(c, fo, inout, n)) return rc;
    if (n_tracked) {
        int k = 0;
        for (int i = 0; i < n; i++) k += inout[i].val >= 0;
        *n_tracked = k;
    }
    return KLT_OK;
}

int klt_track_stats_reset(klt_ctx *c)
{
    if (!c) return KLT_ERR_ARG;
    HIPCHK(c, hipMemsetAsync(c->stats_d, 0, (1 + 2 * KLT_MAX_LEVELS) * sizeof(unsigned long long), c->stream));
    c->collect_stats = true;
    return KLT_OK;
}

int klt_track_stats_read(klt_ctx *c, klt_track_stats *out)
{
    if (!c || !out) return fail(c, KLT_ERR_ARG, "null argument");
    unsigned long long h[1 + 2 * KLT_MAX_LEVELS];
    HIPCHK(c, hipMemcpyAsync(h, c->stats_d, sizeof(h), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    c->collect_stats = false;
    out->features = h[0];
    for (int l = 0; l < KLT_MAX_LEVELS; l++) { out->level_visits[l] = h[1 + l]; out->iterations[l] = h[1 + KLT_MAX_LEVELS + l]; }
    return KLT_OK;
}


}  // extern "C"
