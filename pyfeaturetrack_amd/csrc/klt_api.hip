// C ABI of libkltgpu.so (include/klt_gpu.h): context, slots, feature buffers, and the launch
// sequences for pyramid build, selection and tracking.  Host-side C++ only; all kernels live in
// conv_kernels.hip / select_kernels.hip / track_kernels.hip.
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <algorithm>
#include <cstring>
#include <memory>
#include <string>
#include <utility>
#include <vector>

#include "klt_internal.h"

thread_local hipEvent_t g_klt_stamp_start = nullptr, g_klt_stamp_stop = nullptr;      // klt_internal.h: timing by dispatch timestamps

namespace {

enum Family { F_SMOOTH_GRAD, F_PYR_REDUCE, F_GRAD, F_SMOOTH_H, F_SMOOTH_V, F_PYR_H, F_PYR_V, F_GRAD_H, F_GRAD_V, F_TRACK,
              F_SAT_ROWS, F_SAT_COLS, F_EIGEN, F_SORT, F_NMS, F_SEED, F_AFFINE, F_COUNT };
const char *const kFamilyName[F_COUNT] = {"smooth_grad_l0", "pyramid_reduce", "gradients", "smooth_h", "smooth_v", "pyramid_h",
                                          "pyramid_v", "gradient_h", "gradient_v", "track", "sat_rows", "sat_cols",
                                          "eigen_keys", "sort", "nms", "seed_map", "affine_check"};

struct Level { int nc = 0, nr = 0; float *img = nullptr, *gx = nullptr, *gy = nullptr; };

struct Slot {
    int nc = 0, nr = 0;
    int raw_kind = 0;                 // 0 none, 1 u8, 2 f32
    uint8_t *u8 = nullptr;
    float *f32 = nullptr;
    size_t u8_cap = 0, f32_cap = 0;   // pixels
    float *planes = nullptr;          // 3 pyramids, level-concatenated
    size_t planes_cap = 0;            // floats
    Level lv[KLT_MAX_LEVELS];
    int nlev = 0, ss = 0;
    bool pyr_valid = false;
    uint64_t gen = 0;                 // which build filled the pyramids (unique per build; travels with klt_swap_slots)
    hipEvent_t ev_upload = nullptr;   // asynchronous ingest: frame copy finished (the build waits for it)
    hipEvent_t ev_consumed = nullptr; // last kernel on `stream` that read the raw frame u8
    uint64_t upload_serial = 0, consumed_serial = 0, consumed_alt_serial = 0;   // when the ring handed those events out (event_live)
    hipEvent_t ev_built = nullptr;    // KLT_OPT_BUILD_STREAM: end of the build that filled this slot's pyramids (recorded on the build stream)
    uint64_t built_serial = 0;
    bool built_pending = false;       // the main stream has not waited for ev_built yet
    bool built_on_bstream = false;    // which stream the last build of this slot ran on
    hipEvent_t ev_read = nullptr;     // last tracker launch on the main stream that reads this slot's pyramids (a build on the build
    uint64_t read_serial = 0;         // stream waits for it; selections synchronise before they return and need no mark)
    bool read_valid = false;
    bool upload_pending = false, consumed_valid = false;
    // asynchronous ingest alternates between two raw buffers so that a copy never has to wait (on the device) for
    // kernels still reading the previous frame: making the copy stream wait on a compute-stream event blocks the
    // HOST for the duration of the queued work on this runtime (measured 150-800 us per step)
    uint8_t *u8_alt = nullptr;
    size_t u8_alt_cap = 0;
    const uint8_t *u8_ext = nullptr;  // klt_slot_adopt_u8: the frame IS this caller-owned device buffer (read in place, never written or freed here)
    hipEvent_t ev_consumed_alt = nullptr;
    bool consumed_alt_valid = false;
    // last asynchronous copy INTO each raw buffer (consecutive uploads go round-robin over the copy streams: the next copy into the
    // same buffer may be issued on another stream and must wait for this one)
    hipEvent_t ev_wr = nullptr, ev_wr_alt = nullptr;
    uint64_t wr_serial = 0, wr_alt_serial = 0;
    int wr_lane = -1, wr_alt_lane = -1;
};

inline const uint8_t *raw8(const Slot *s) { return s->u8_ext ? s->u8_ext : s->u8; }

struct FeatBuf { klt_feat *d = nullptr; int cap = 0; bool view = false; hipEvent_t comm_done = nullptr; /* last collective that touched it */ };

struct AffState { klt_affine_rec *rec = nullptr; float *tpl = nullptr; int n = 0, tn = 0; };

struct Timed { int fam; hipEvent_t a, b; double bytes; };

std::string g_create_error;

}  // namespace

// one set of prepared selection scores (klt_select_prepare_async)
struct ScoreCache {
    unsigned long long *keys = nullptr;
    size_t cap = 0;
    uint64_t gen = 0, stamp = 0;      // generation of the slot contents the keys were scored on (0 = empty); age
    int nc = 0, nr = 0, bx = 0, by = 0, hw = 0, hh = 0, step = 0, nx = 0, ny = 0;
    double min_eig = 0;
    hipEvent_t ev = nullptr;
    uint64_t ev_serial = 0;
};

// The parallel minimum-distance selection in two halves: everything up to the point where the host has to look at the outcome
// (klt_select_begin_async) and the rest (klt_select_finish: wait, look, and -- rarely -- more passes or the repeat with every
// candidate).  Between the two the caller may enqueue other work (the next frame's upload, build and score preparation).
struct SelectJob {
    static constexpr int kMaxRounds = 512;
    MisArgs ma;
    NmsArgs pa;
    SelectArgs sa;
    klt_feat *fl = nullptr;
    ScoreCache *pre = nullptr;
    unsigned *zero_from = nullptr, *hist_d = nullptr, *ticket_d = nullptr, *info_d = nullptr, *rank_d = nullptr, *acc_count_d = nullptr;
    int *nfill_d = nullptr;
    size_t zero_n = 0;
    long long bound = 0, np2 = 0, ncand = 0, target = 0;
    int n = 0, mode = 0, rounds_per_look = 0, attempt = 0, round = 0, look = 0;
    bool by_rank = false, prefilter = false, filtered = false;
};


struct klt_ctx {
    int device = -1;
    hipStream_t stream = nullptr;     // uploads, pyramid build, selection, tracker
    hipStream_t cstream = nullptr;    // asynchronous frame ingest from pinned host memory (created on first use)
    // ... and more copy streams: consecutive uploads go round-robin over all of them, i.e. over several DMA engines (2 MB frames: 44 GB/s
    // on one stream, 55 GB/s on two -- profiles/r04_h2d_probe.json), and a stream's next copy is not queued right behind the event of its last one
    static constexpr int kMaxCopyStreams = 8;
    hipStream_t cextra[kMaxCopyStreams - 1] = {nullptr};
    int ncopy = 2;                    // copy streams in use (KLT_COPY_STREAMS)
    unsigned upload_count = 0;
    hipStream_t bstream = nullptr;    // KLT_OPT_BUILD_STREAM: pyramid builds run here, overlapping the tracker / selection of earlier frames
    hipStream_t work = nullptr;       // stream the pyramid-build helpers enqueue on: `stream`, or `bstream` inside a build
    bool build_stream_on = false;
    KltComm *comm = nullptr;          // RCCL communicator + side stream (klt_comm_init_rank), comm.hip
    std::vector<void *> pinned;       // klt_host_alloc allocations
    std::vector<void *> dev_allocs;   // klt_device_alloc allocations
    std::vector<size_t> dev_alloc_bytes;
    // Ordering events come from one ring and are never re-recorded while a waiter may still be queued on them
    // (re-recording a pending event makes hipEventRecord block the host until the device has caught up -- measured:
    // 400-800 us per step).  256 events ~ 25 steps of history.
    std::vector<hipEvent_t> ring;
    size_t ring_next = 0;
    uint64_t ring_serial = 0;         // events handed out so far
    std::string err;
    klt_params p{};
    bool have_params = false;
    Taps gauss[3], deriv[3];
    bool have_taps[3] = {false, false, false};
    std::vector<Slot> slots;
    std::vector<FeatBuf> fbs;
    float *tmpA = nullptr, *tmpB = nullptr;
    size_t tmp_cap = 0;
    float *h1 = nullptr;                      // H1 planes of the fused first reduction (one per frame of a batch)
    size_t h1_cap = 0;
    bool fuse_hreduce = true;                 // KLT_OPT_FUSED_HREDUCE
    bool track_xcd_order = true;              // KLT_OPT_TRACK_XCD_ORDER
    uint64_t waited_built_serial = ~0ull;     // the build event the main stream waited for last (wait_built)
    // selection scratch
    float *sel_img = nullptr, *sel_gx = nullptr, *sel_gy = nullptr, *sat = nullptr, *valmap = nullptr;
    size_t sel_cap = 0;               // pixels
    unsigned long long *keys = nullptr;
    size_t keys_cap = 0;
    uint8_t *seedmap = nullptr;
    size_t seed_cap = 0, seed_n = 0;          // pixels the stamps in the map are valid for
    uint8_t seed_stamp = 0;                   // stamp of the latest replacement pass (1..255)
    uint32_t *grid = nullptr;
    size_t grid_cap = 0;
    int *nms_slots = nullptr;
    size_t nms_slots_cap = 0;
    unsigned long long *keys2 = nullptr;      // compacted top-K keys
    size_t keys2_cap = 0;
    unsigned *topk_hist = nullptr;            // 8192 bins + 4 words of info
    klt_feat *fl_snapshot = nullptr;
    size_t fl_snapshot_cap = 0;
    bool track_tree_sums = false;     // KLT_OPT_TRACK_TREE_SUMS
    bool use_topk = true;
    bool use_mis = true;                      // parallel minimum-distance passes instead of the sorted serial walk
    int mis_rounds_hint = 6;
    int sat_variant = 1;                      // 1: step-synchronous wavefront pipelines (sat_pipeline.hip), 0: barrier-coupled SAT kernels
    unsigned *readback = nullptr;             // pinned scratch for small results
    float *score_override = nullptr;          // test hook (klt_set_score_override)
    size_t score_override_cap = 0;
    int score_override_n = 0;
    uint32_t *mis_st = nullptr, *mis_list = nullptr;
    unsigned long long *mis_tile_keys = nullptr;
    size_t mis_tile_keys_cap = 0;
    unsigned *mis_cnt = nullptr;              // [tiles] + kMisRounds remaining counters + accepted counter
    size_t mis_st_cap = 0, mis_list_cap = 0, mis_cnt_cap = 0;
    // klt_select_prepare_async: the list-independent half of a selection (summed-area tables, eigenvalue keys) of a slot's level 0, computed
    // ahead of time (on the build stream when that is on).  Two sets by default: the next frame's keys are written while this frame's are read;
    // a rank that prepares a whole block of frames while it waits for the feature list of the previous block keeps one per frame.
    std::vector<ScoreCache> pre = std::vector<ScoreCache>(2);      // KLT_OPT_SCORE_SETS
    std::unique_ptr<SelectJob> sel_job;       // a selection between klt_select_begin_async and klt_select_finish
    hipEvent_t ev_download = nullptr;         // behind the latest klt_featbuf_download_async (an event of the ring)
    uint64_t download_serial = 0;
    bool download_pending = false;
    hipEvent_t ev_sel = nullptr;              // behind the last launch of the pending selection's latest batch: what klt_select_finish waits for
    float *sat_pre = nullptr;
    size_t sat_pre_cap = 0;
    uint64_t gen_counter = 0, pre_stamp = 0;
    int last_build_on_bstream = -1;           // -1: no build yet
    hipEvent_t ev_bbuild = nullptr;           // end of the latest build on the build stream (shares the H1 scratch with main-stream builds)
    uint64_t bbuild_serial = 0;
    const unsigned long long *sorted_keys = nullptr;   // what the last selection walked (test hook)
    int sorted_count = 0;
    // batched tracker launches: the descriptor tables (and XCD-aware feature orders) of the last few distinct batches stay on the device --
    // a shard that goes through in sub-shards, step after step, uploads each table once
    struct BatchTable {
        std::vector<TrackPairDesc> host;          // what `dev` holds
        uint64_t hash = 0;
        TrackPairDesc *dev = nullptr;
        size_t cap = 0;
        uint64_t used = 0;
    };
    // the XCD-aware feature orders depend on the INPUT lists only: one set of permutations per distinct (inputs, length), whatever
    // the frames and the output buffers of the launch
    struct BatchOrder {
        std::vector<const klt_feat *> in;
        uint32_t *order = nullptr;
        size_t cap = 0;
        int n = -1, age = 0;
        uint64_t used = 0;
    };
    std::vector<BatchTable> batch_tables;         // at most kBatchTables, least recently used one replaced
    std::vector<BatchOrder> batch_orders;         // at most kBatchOrders
    BatchOrder shared_order;                      // single-pair launches on lists seen for the first time
    std::vector<const klt_feat *> seen_once;      // input lists of single-pair launches seen once so far (set_track_order)
    uint64_t batch_clock = 0;
    static constexpr size_t kBatchTables = 256, kBatchOrders = 128;
    klt_affine_params ap{-1, 15, 15, 10, 10.f, 0.02f, 1.5f};      // klt.py:67-73 defaults
    std::vector<AffState> aff;
    int select_aff_state = -1;
    int *placed_d = nullptr;
    const float *last_sel[3] = {nullptr, nullptr, nullptr};
    int sel_nc = 0, sel_nr = 0, sel_nx = 0, sel_ny = 0, sel_npow2 = 0;
    bool sel_valmap = false;          // the last selection wrote the eigenvalue map (one that used prepared scores did not)
    unsigned long long *stats_d = nullptr;
    bool collect_stats = false;
    bool use_fused = true;            // LDS-tiled fused kernels (pyramid_kernels.hip); off = generic two-pass kernels
    // timing
    bool timing = false;
    bool timing_stamps = false;               // klt_timing_enable(ctx, 2): single-launch kernel families are timed by their dispatch timestamps
    std::vector<Timed> pending;
    std::vector<hipEvent_t> pool;
    double acc_ms[F_COUNT] = {0}, acc_bytes[F_COUNT] = {0};
    unsigned acc_n[F_COUNT] = {0};
    unsigned acc_unstamped[F_COUNT] = {0};    // timing mode 2: scopes of a family whose launch did not go through klt_launch (nothing measured)
    // experiment (tools/graph_frame_probe.py, profiles/README.md "HIP graphs"): one frame's launch set captured into a HIP graph and replayed
    bool capturing = false;                   // the streams are being captured: nothing may synchronise or allocate
    hipGraphExec_t probe_graph = nullptr;
};

namespace {

int fail(klt_ctx *c, int code, const std::string &msg)
{
    if (c) c->err = msg;
    return code;
}

#define HIPCHK(c, call)                                                                              \
    do {                                                                                             \
        hipError_t e_ = (call);                                                                      \
        if (e_ != hipSuccess)                                                                        \
            return fail((c), KLT_ERR_DEVICE, std::string(#call) + ": " + hipGetErrorString(e_));     \
    } while (0)

struct TimerScope {
    klt_ctx *c;
    Timed t;
    bool on;
    hipStream_t st;
    bool stamps = false;
    TimerScope(klt_ctx *c_, int fam, double bytes, hipStream_t st_ = nullptr) : c(c_), on(c_->timing), st(st_ ? st_ : c_->work)
    {
        if (!on) return;
        t.fam = fam;
        t.bytes = bytes;
        for (hipEvent_t *e : {&t.a, &t.b}) {
            if (!c->pool.empty()) { *e = c->pool.back(); c->pool.pop_back(); }
            else if (hipEventCreate(e) != hipSuccess) { on = false; return; }
        }
        // families whose scope holds ONE launch that goes through klt_launch (the order kernel in front of a tracker and the threshold
        // kernel behind the eigenvalue pass do not): timed by that dispatch's own timestamps.  Scopes with several launches keep the pair
        stamps = c->timing_stamps && (fam == F_SMOOTH_GRAD || fam == F_PYR_REDUCE || fam == F_GRAD || fam == F_TRACK || fam == F_AFFINE ||
                                      fam == F_SAT_ROWS || fam == F_SAT_COLS || fam == F_EIGEN);
        if (stamps) { g_klt_stamp_start = t.a; g_klt_stamp_stop = t.b; }      // filled by the launch itself (klt_launch)
        else hipEventRecord(t.a, st);
    }
    ~TimerScope()
    {
        if (!on) return;
        if (stamps) {
            if (g_klt_stamp_start) {                 // no launch took them (an error path, or a launcher that does not go through
                c->acc_unstamped[t.fam]++;           // klt_launch): nothing was measured -- counted, klt_timing_read says so
                g_klt_stamp_start = g_klt_stamp_stop = nullptr;
                c->pool.push_back(t.a); c->pool.push_back(t.b);
                return;
            }
        } else {
            hipEventRecord(t.b, st);
        }
        c->pending.push_back(t);
    }
};

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

template <typename T>
int ensure(klt_ctx *c, T *&ptr, size_t &cap, size_t want)
{
    if (want <= cap && ptr) return 0;
    if (c->capturing) return fail(c, KLT_ERR_STATE, "a buffer would have to grow during a stream capture");
    if (ptr) { if (int rc = sync_all(c)) return rc; HIPCHK(c, hipFree(ptr)); ptr = nullptr; cap = 0; }
    HIPCHK(c, hipMalloc((void **)&ptr, want * sizeof(T)));
    cap = want;
    return 0;
}

int ensure_tmp(klt_ctx *c, size_t pixels)
{
    if (pixels <= c->tmp_cap && c->tmpA) return 0;
    if (c->tmpA) { if (int rc = sync_all(c)) return rc; hipFree(c->tmpA); hipFree(c->tmpB); c->tmpA = c->tmpB = nullptr; c->tmp_cap = 0; }
    HIPCHK(c, hipMalloc((void **)&c->tmpA, pixels * sizeof(float)));
    HIPCHK(c, hipMalloc((void **)&c->tmpB, pixels * sizeof(float)));
    c->tmp_cap = pixels;
    return 0;
}

int ensure_h1(klt_ctx *c, size_t floats)
{
    if (floats <= c->h1_cap && c->h1) return 0;
    if (c->h1) { if (int rc = sync_all(c)) return rc; hipFree(c->h1); c->h1 = nullptr; c->h1_cap = 0; }
    HIPCHK(c, hipMalloc((void **)&c->h1, floats * sizeof(float)));
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
        HIPCHK(c, hipMalloc((void **)&nd, (size_t)n * sizeof(klt_feat)));
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
int fresh_event(klt_ctx *c, hipEvent_t *out, uint64_t *serial = nullptr)
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

// The kernels address a plane with 32-bit BYTE offsets into a buffer descriptor of 2 GB (raw buffer operations, klt_internal.h), and the
// largest plane is the interleaved gradient plane of level 0 with 8 bytes per pixel: a frame must stay below 2^28 pixels.
// 2^28 - 1 pixels is a 16 384 x 16 383 frame; the largest frame of the test suite is 7680 x 4320.
constexpr long long kMaxFramePixels = (1LL << 28) - 1;

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
        HIPCHK(c, hipMalloc((void **)&s->planes, 3 * total * sizeof(float)));
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

int check_ready(klt_ctx *c)
{
    if (!c) return KLT_ERR_ARG;
    if (!c->have_params) return fail(c, KLT_ERR_STATE, "klt_set_params has not been called");
    for (int i = 0; i < 3; i++)
        if (!c->have_taps[i]) return fail(c, KLT_ERR_STATE, "klt_set_kernels has not been called for all three tap sets");
    return 0;
}

// smooth(raw) -> dst, convolve.py:254-264 with the smoothing taps
int enqueue_smooth_raw(klt_ctx *c, Slot *s, float *dst)
{
    const int nc = s->nc, nr = s->nr;
    const double N = (double)nc * nr;
    const Taps &g = c->gauss[0];
    {
        TimerScope t(c, F_SMOOTH_H, N * ((s->raw_kind == 1 ? 1 : 4) + 4));
        if (s->raw_kind == 1) launch_hconv_u8(c->work, raw8(s), nc, nr, c->tmpA, nullptr, nc, 1, 0, g, nullptr);
        else launch_hconv_f32(c->work, s->f32, nc, nr, c->tmpA, nullptr, nc, 1, 0, g, nullptr);
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
                              float *const *gx, float *const *gy, int nc, int nr, bool *fused_h1 = nullptr)
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
                       bool u8_input = false)
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
                raw[b] = g[b]->raw_kind == 1 ? (const void *)raw8(g[b]) : (const void *)g[b]->f32;
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

}  // namespace

// =============================================================================================== ABI

extern "C" {

int klt_comm_fence_async(klt_ctx *c);

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
    HIPCHK(c, hipHostMalloc(&p, bytes, hipHostMallocDefault));
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

int klt_upload_u8_async(klt_ctx *c, int slot, const uint8_t *px, int ncols, int nrows, int pitch)
{
    if (!c || !px) return fail(c, KLT_ERR_ARG, "null argument");
    if (ncols <= 0 || nrows <= 0 || ncols > 65535 || nrows > 65535 || pitch < ncols) return fail(c, KLT_ERR_ARG, "bad image geometry");
    if ((long long)ncols * nrows > kMaxFramePixels) return fail(c, KLT_ERR_ARG, "frame too large (2^28 pixels or more: a plane must stay below 2 GB)");
    HIPCHK(c, hipSetDevice(c->device));
    hipPointerAttribute_t attr;                           // the source must be pinned: a pageable copy would be staged synchronously
    if (hipPointerGetAttributes(&attr, px) != hipSuccess || attr.type != hipMemoryTypeHost) {
        (void)hipGetLastError();
        return fail(c, KLT_ERR_ARG, "klt_upload_u8_async needs pinned host memory (klt_host_alloc)");
    }
    if (!c->cstream) HIPCHK(c, hipStreamCreateWithFlags(&c->cstream, hipStreamNonBlocking));
    const int lane = (int)(c->upload_count++ % (unsigned)c->ncopy);                 // the two frames of a pair travel side by side
    if (lane > 0 && !c->cextra[lane - 1]) HIPCHK(c, hipStreamCreateWithFlags(&c->cextra[lane - 1], hipStreamNonBlocking));
    const hipStream_t cs = lane == 0 ? c->cstream : c->cextra[lane - 1];
    Slot *s;
    if (int rc = get_slot(c, slot, &s, true)) return rc;
    const size_t px_count = (size_t)ncols * nrows;
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
    else HIPCHK(c, hipMemcpy2DAsync(s->u8, (size_t)ncols, px, (size_t)pitch, (size_t)ncols, nrows, hipMemcpyHostToDevice, cs));
    if (int rc = fresh_event(c, &s->ev_upload, &s->upload_serial)) return rc;
    HIPCHK(c, hipEventRecord(s->ev_upload, cs));
    s->upload_pending = true;
    s->ev_wr = s->ev_upload; s->wr_serial = s->upload_serial; s->wr_lane = lane;
    s->nc = ncols;
    s->nr = nrows;
    s->raw_kind = 1;
    s->pyr_valid = false;
    return KLT_OK;
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
    if (hipMalloc(&p, bytes) != hipSuccess) { (void)hipGetLastError(); return fail(c, KLT_ERR_NOMEM, "klt_device_alloc: out of device memory"); }
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
    s->pyr_valid = false;
    return KLT_OK;
}

int klt_build_pyramids_async(klt_ctx *c, int slot) { return build_pyramids_batch(c, &slot, 1); }

int klt_build_pyramids_batch_async(klt_ctx *c, const int *slots, int n) { return build_pyramids_batch(c, slots, n); }

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

int klt_build_pyramids(klt_ctx *c, int slot)
{
    if (int rc = klt_build_pyramids_async(c, slot)) return rc;
    return klt_sync(c);
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

int klt_featbuf_upload(klt_ctx *c, int fb, const klt_feat *src, int n)
{
    if (!c || !src || n < 0) return fail(c, KLT_ERR_ARG, "bad argument");
    HIPCHK(c, hipSetDevice(c->device));
    FeatBuf *b;
    if (int rc = get_fb(c, fb, n > 0 ? n : 1, &b)) return rc;
    HIPCHK(c, hipMemcpyAsync(b->d, src, (size_t)n * sizeof(klt_feat), hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return KLT_OK;
}

int klt_featbuf_upload_async(klt_ctx *c, int fb, const klt_feat *src, int n)
{
    if (!c || !src || n < 0) return fail(c, KLT_ERR_ARG, "bad argument");
    HIPCHK(c, hipSetDevice(c->device));
    FeatBuf *b;
    if (int rc = get_fb(c, fb, n > 0 ? n : 1, &b)) return rc;
    HIPCHK(c, hipMemcpyAsync(b->d, src, (size_t)n * sizeof(klt_feat), hipMemcpyHostToDevice, c->stream));
    return KLT_OK;
}

int klt_featbuf_download(klt_ctx *c, int fb, klt_feat *dst, int n)
{
    if (!c || !dst || n < 0) return fail(c, KLT_ERR_ARG, "bad argument");
    if (fb < 0 || (size_t)fb >= c->fbs.size() || c->fbs[fb].cap < n) return fail(c, KLT_ERR_STATE, "feature buffer not that large");
    HIPCHK(c, hipSetDevice(c->device));
    if (int rc = klt_comm_fence_async(c)) return rc;        // a gathered table is complete before it is read back
    HIPCHK(c, hipMemcpyAsync(dst, c->fbs[fb].d, (size_t)n * sizeof(klt_feat), hipMemcpyDeviceToHost, c->stream));
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
    HIPCHK(c, hipMemcpyAsync(dst, c->fbs[fb].d, (size_t)n * sizeof(klt_feat), hipMemcpyDeviceToHost, c->stream));
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

// ---------------------------------------------------------------------------------------- selection
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
            HIPCHK(c, hipMemcpyAsync(j.fl, c->fl_snapshot, (size_t)j.n * sizeof(klt_feat), hipMemcpyDeviceToDevice, c->stream));
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
            HIPCHK(c, hipMemcpyAsync(j.fl, c->fl_snapshot, (size_t)j.n * sizeof(klt_feat), hipMemcpyDeviceToDevice, c->stream));
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
int enqueue_sat(klt_ctx *c, hipStream_t st, const float *gx, const float *gy, float *sat, int nc, int nr, bool rows_only = false)
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
        HIPCHK(c, hipMalloc((void **)&c->sel_img, N * sizeof(float)));
        HIPCHK(c, hipMalloc((void **)&c->sel_gx, KLT_GRAD_STRIDE * N * sizeof(float)));      // gradx / grady interleaved, like a slot's planes
        c->sel_gy = c->sel_gx + 1;
        HIPCHK(c, hipMalloc((void **)&c->sat, 3 * N * sizeof(float)));
        HIPCHK(c, hipMalloc((void **)&c->valmap, N * sizeof(float)));
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
            const void *raw = s->raw_kind == 1 ? (const void *)raw8(s) : (const void *)s->f32;
            if (int rc = enqueue_fused_smooth_grad(c, 1, &raw, s->raw_kind, &c->sel_img, &c->sel_gx, &c->sel_gy, nc, nr)) return rc;
            img = c->sel_img;
            grads_done = true;
        } else if (p.smoothBeforeSelecting) {
            enqueue_smooth_raw(c, s, c->sel_img);
            img = c->sel_img;
        } else if (s->raw_kind == 2) {
            img = s->f32;
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
            HIPCHK(c, hipHostMalloc(&hp, 128 * sizeof(unsigned), hipHostMallocDefault));
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
        HIPCHK(c, hipMemcpyAsync(c->fl_snapshot, b->d, (size_t)n * sizeof(klt_feat), hipMemcpyDeviceToDevice, c->stream));
        if (int rc = run_nms(c->keys2, (int)kept)) return rc;
        c->sorted_keys = c->keys2; c->sorted_count = (int)kept;
        int res[2] = {0, 0};
        HIPCHK(c, hipMemcpyAsync(res, c->placed_d, sizeof(res), hipMemcpyDeviceToHost, c->stream));
        HIPCHK(c, hipStreamSynchronize(c->stream));
        if (!(res[1] && kept < valid)) { HIPCHK(c, hipGetLastError()); return KLT_OK; }
        // the kept candidates ran out before the list was full: restore the list and take the full sort
        HIPCHK(c, hipMemcpyAsync(b->d, c->fl_snapshot, (size_t)n * sizeof(klt_feat), hipMemcpyDeviceToDevice, c->stream));
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

// ----------------------------------------------------------------------------------------- tracking
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
        HIPCHK(c, hipMalloc((void **)&a.rec, (size_t)n * sizeof(klt_affine_rec)));
        HIPCHK(c, hipMalloc((void **)&a.tpl, (size_t)n * 3 * tn * sizeof(float)));
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

// ------------------------------------------------------------------------------------- multi-GPU
static std::string g_comm_error;      // klt_comm_unique_id has no context to report into

int klt_comm_unique_id(void *out128)
{
    const int rc = comm_unique_id(out128, g_comm_error);
    if (rc) g_create_error = g_comm_error;          // readable through klt_last_error(NULL)
    return rc;
}

int klt_comm_init_rank(klt_ctx *c, int nranks, int rank, const void *unique_id)
{
    if (!c) return KLT_ERR_ARG;
    if (c->comm) return fail(c, KLT_ERR_STATE, "the context already has a communicator");
    std::string err;
    if (int rc = comm_create(c->device, nranks, rank, unique_id, &c->comm, err)) return fail(c, rc, err);
    return KLT_OK;
}

int klt_comm_destroy(klt_ctx *c)
{
    if (!c) return KLT_ERR_ARG;
    if (c->comm) {
        comm_destroy(c->comm);                    // (waits for the side stream, then destroys the communicator's events)
        c->comm = nullptr;
        for (FeatBuf &b : c->fbs) b.comm_done = nullptr;      // ... which the feature buffers must not keep
    }
    return KLT_OK;
}

int klt_comm_info(klt_ctx *c, int *nranks, int *rank)
{
    if (!c) return KLT_ERR_ARG;
    if (nranks) *nranks = comm_nranks(c->comm);
    if (rank) *rank = comm_rank(c->comm);
    return KLT_OK;
}

static int gather_common(klt_ctx *c, int fb_src, int fb_dst, int n, int root /* -1: all-gather */)
{
    if (!c) return KLT_ERR_ARG;
    if (!c->comm) return fail(c, KLT_ERR_STATE, "klt_comm_init_rank has not been called");
    if (n <= 0 || fb_src == fb_dst) return fail(c, KLT_ERR_ARG, "bad gather arguments");
    if (fb_src < 0 || (size_t)fb_src >= c->fbs.size() || c->fbs[fb_src].cap < n) return fail(c, KLT_ERR_STATE, "source feature buffer not that large");
    HIPCHK(c, hipSetDevice(c->device));
    const int nranks = comm_nranks(c->comm), rank = comm_rank(c->comm);
    const bool need_dst = root < 0 || rank == root;
    if ((long long)n * nranks > 0x7fffffffLL) return fail(c, KLT_ERR_ARG, "gathered table too large");
    klt_feat *dst = nullptr;
    if (need_dst) {
        FeatBuf *bd;
        if (int rc = get_fb(c, fb_dst, n * nranks, &bd)) return rc;      // may grow c->fbs: take the source pointer afterwards
        dst = bd->d;
    }
    const klt_feat *src = c->fbs[fb_src].d;
    std::string err;
    const size_t bytes = (size_t)n * sizeof(klt_feat);
    const int rc = root < 0 ? comm_allgather(c->comm, c->stream, src, dst, bytes, err)
                            : comm_gather(c->comm, c->stream, src, dst, bytes, root, err);
    if (rc) return fail(c, rc, err);
    c->fbs[fb_src].comm_done = comm_last_done(c->comm);
    if (need_dst) c->fbs[fb_dst].comm_done = c->fbs[fb_src].comm_done;
    return KLT_OK;
}

// gather with a count per rank: rank r contributes the first counts[r] records of its fb_src, the root's fb_dst receives them back to
// back in rank order.  Every rank passes the same table of nranks counts (the caller's shard arithmetic).
int klt_gatherv_featbuf_async(klt_ctx *c, int fb_src, int fb_dst, const int *counts, int root)
{
    if (!c) return KLT_ERR_ARG;
    if (!c->comm) return fail(c, KLT_ERR_STATE, "klt_comm_init_rank has not been called");
    const int nranks = comm_nranks(c->comm), rank = comm_rank(c->comm);
    if (!counts || root < 0 || root >= nranks || fb_src == fb_dst) return fail(c, KLT_ERR_ARG, "bad gatherv arguments");
    long long total = 0;
    std::vector<size_t> bytes((size_t)nranks);
    for (int r = 0; r < nranks; r++) {
        if (counts[r] < 0) return fail(c, KLT_ERR_ARG, "negative count");
        total += counts[r];
        bytes[r] = (size_t)counts[r] * sizeof(klt_feat);
    }
    if (total <= 0 || total > 0x7fffffffLL) return fail(c, KLT_ERR_ARG, "gathered table empty or too large");
    const int n = counts[rank];
    if (n > 0 && (fb_src < 0 || (size_t)fb_src >= c->fbs.size() || c->fbs[fb_src].cap < n)) return fail(c, KLT_ERR_STATE, "source feature buffer not that large");
    HIPCHK(c, hipSetDevice(c->device));
    klt_feat *dst = nullptr;
    if (rank == root) {
        FeatBuf *bd;
        if (int rc = get_fb(c, fb_dst, (int)total, &bd)) return rc;      // may grow c->fbs: take the source pointer afterwards
        dst = bd->d;
    }
    const klt_feat *src = n > 0 ? c->fbs[fb_src].d : nullptr;
    std::string err;
    if (const int rc = comm_gatherv(c->comm, c->stream, src, dst, bytes.data(), root, err)) return fail(c, rc, err);
    if (n > 0) c->fbs[fb_src].comm_done = comm_last_done(c->comm);
    if (rank == root) c->fbs[fb_dst].comm_done = comm_last_done(c->comm);
    return KLT_OK;
}

int klt_comm_set_timeout(klt_ctx *c, double ms)
{
    if (!c) return KLT_ERR_ARG;
    if (!c->comm) return fail(c, KLT_ERR_STATE, "klt_comm_init_rank has not been called");
    comm_set_timeout(c->comm, ms);
    return KLT_OK;
}

int klt_comm_fence_featbuf_async(klt_ctx *c, int fb)
{
    if (!c) return KLT_ERR_ARG;
    if (fb < 0 || (size_t)fb >= c->fbs.size() || !c->fbs[fb].comm_done) return KLT_OK;
    HIPCHK(c, hipStreamWaitEvent(c->stream, c->fbs[fb].comm_done, 0));
    return KLT_OK;
}

int klt_allgather_featbuf_async(klt_ctx *c, int fb_src, int fb_dst, int n) { return gather_common(c, fb_src, fb_dst, n, -1); }

int klt_gather_featbuf_async(klt_ctx *c, int fb_src, int fb_dst, int n, int root)
{
    if (c && c->comm && (root < 0 || root >= comm_nranks(c->comm))) return fail(c, KLT_ERR_ARG, "gather root out of range");
    return gather_common(c, fb_src, fb_dst, n, root < 0 ? 0 : root);
}

// The feature list as a baton between the GPUs of one temporal sequence: n records of fb_send go to rank `to` and / or
// n records arrive in fb_recv from rank `from` (-1: no such side), on the communicator's side stream behind everything enqueued on the
// main stream so far; the main stream then waits for the arrival, so whatever is enqueued next reads the received list.
int klt_sendrecv_featbuf_async(klt_ctx *c, int fb_send, int to, int fb_recv, int from, int n)
{
    if (!c) return KLT_ERR_ARG;
    if (!c->comm) return fail(c, KLT_ERR_STATE, "klt_comm_init_rank has not been called");
    if (n <= 0 || (to < 0 && from < 0)) return fail(c, KLT_ERR_ARG, "bad send / receive arguments");
    HIPCHK(c, hipSetDevice(c->device));
    klt_feat *dst = nullptr;
    if (from >= 0) {
        if (fb_recv == fb_send && to >= 0) return fail(c, KLT_ERR_ARG, "send and receive buffers must differ");
        FeatBuf *bd;
        if (int rc = get_fb(c, fb_recv, n, &bd)) return rc;             // may grow c->fbs: take the source pointer afterwards
        dst = bd->d;
    }
    const klt_feat *src = nullptr;
    if (to >= 0) {
        if (fb_send < 0 || (size_t)fb_send >= c->fbs.size() || c->fbs[fb_send].cap < n) return fail(c, KLT_ERR_STATE, "source feature buffer not that large");
        src = c->fbs[fb_send].d;
    }
    std::string err;
    if (const int rc = comm_sendrecv(c->comm, c->stream, src, to, dst, from, (size_t)n * sizeof(klt_feat), err)) return fail(c, rc, err);
    if (to >= 0) c->fbs[fb_send].comm_done = comm_last_done(c->comm);
    if (from >= 0) {
        c->fbs[fb_recv].comm_done = comm_last_done(c->comm);
        HIPCHK(c, hipStreamWaitEvent(c->stream, c->fbs[fb_recv].comm_done, 0));
    }
    return KLT_OK;
}

int klt_comm_fence_async(klt_ctx *c)
{
    if (!c) return KLT_ERR_ARG;
    if (!c->comm) return KLT_OK;
    std::string err;
    const int rc = comm_fence(c->comm, c->stream, err);
    return rc ? fail(c, rc, err) : KLT_OK;
}

int klt_comm_wait(klt_ctx *c)
{
    if (!c) return KLT_ERR_ARG;
    if (!c->comm) return KLT_OK;
    std::string err;
    const int rc = comm_wait(c->comm, err);
    return rc ? fail(c, rc, err) : KLT_OK;
}

int klt_comm_allreduce_max(klt_ctx *c, double *inout, int n)
{
    if (!c) return KLT_ERR_ARG;
    if (!c->comm) return fail(c, KLT_ERR_STATE, "klt_comm_init_rank has not been called");
    std::string err;
    const int rc = comm_allreduce_max(c->comm, inout, n, err);
    return rc ? fail(c, rc, err) : KLT_OK;
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
static int download_plane(klt_ctx *c, const float *src, int stride, size_t cnt, float *dst)
{
    float *tmp = nullptr;
    if (stride != 1) {
        HIPCHK(c, hipMalloc((void **)&tmp, cnt * sizeof(float)));
        launch_take_strided(c->stream, src, tmp, cnt, stride);
        src = tmp;
    }
    hipError_t e = hipMemcpyAsync(dst, src, cnt * sizeof(float), hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    if (tmp) hipFree(tmp);
    HIPCHK(c, e);
    return KLT_OK;
}

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

// ------------------------------------------------------------------------- the reference's literal native boundary
// (setup.py:8-9: the Cython `def` functions of goodFeaturesUtils / trackFeaturesUtils; host arrays in and out, synchronous)
int klt_scan_good_features_f32(klt_ctx *c, const float *gradx, const float *grady, int ncols, int nrows, int borderx, int bordery,
                               int window_hw, int window_hh, int nSkippedPixels, float *val, int val_cap, int *nx_out, int *ny_out)
{
    if (!c || !gradx || !grady) return fail(c, KLT_ERR_ARG, "null argument");
    if (ncols <= 0 || nrows <= 0 || (long long)ncols * nrows > kMaxFramePixels) return fail(c, KLT_ERR_ARG, "bad image geometry");
    if (nSkippedPixels < 0 || window_hw < 0 || window_hh < 0) return fail(c, KLT_ERR_ARG, "bad window / skip");
    // the reference reads cumSum[y - hh - 1][x - hw - 1] with bounds checks off (goodFeaturesUtils.pyx:3, :26): defined only from here on
    if (borderx - window_hw - 1 < 0 || bordery - window_hh - 1 < 0)
        return fail(c, KLT_ERR_ARG, "border must be at least window/2 + 1 (the reference reads outside the image otherwise)");
    const int step = nSkippedPixels + 1;
    const int nx = (ncols - borderx > borderx) ? (ncols - 2 * borderx + step - 1) / step : 0;
    const int ny = (nrows - bordery > bordery) ? (nrows - 2 * bordery + step - 1) / step : 0;
    if (nx_out) *nx_out = nx;
    if (ny_out) *ny_out = ny;
    const long long ncand = (long long)nx * ny;
    if (ncand == 0) return KLT_OK;
    if (!val || val_cap < ncand) return fail(c, KLT_ERR_ARG, "val holds fewer than nx * ny floats");
    HIPCHK(c, hipSetDevice(c->device));
    const size_t N = (size_t)ncols * nrows;
    long long npow2 = 2048;
    while (npow2 < ncand) npow2 <<= 1;
    // gradx / grady interleaved, as the table kernels read a slot's planes
    std::vector<float> inter(2 * N);
    for (size_t i = 0; i < N; i++) { inter[2 * i] = gradx[i]; inter[2 * i + 1] = grady[i]; }
    float *d = nullptr;                         // [2N] gradients | [3N] tables | [ncand] eigenvalues
    unsigned long long *keys = nullptr;
    HIPCHK(c, hipMalloc((void **)&d, (5 * N + (size_t)ncand) * sizeof(float)));
    hipError_t e = hipMalloc((void **)&keys, (size_t)npow2 * sizeof(unsigned long long));
    int rc = 0;
    if (e == hipSuccess) e = hipMemcpyAsync(d, inter.data(), 2 * N * sizeof(float), hipMemcpyHostToDevice, c->stream);
    if (e == hipSuccess) {
        c->work = c->stream;
        rc = enqueue_sat(c, c->stream, d, d + 1, d + 2 * N, ncols, nrows);
    }
    if (e == hipSuccess && rc == 0) {
        SelectArgs sa;
        std::memset(&sa, 0, sizeof(sa));
        sa.sat = d + 2 * N; sa.valmap = d + 5 * N; sa.keys = keys;
        sa.min_eig = 1.0;
        sa.ncols = ncols; sa.nrows = nrows; sa.bx = borderx; sa.by = bordery; sa.step = step; sa.nx = nx; sa.ny = ny;
        sa.hw = window_hw; sa.hh = window_hh; sa.npow2 = (int)npow2;
        launch_eigen(c->stream, sa);
        e = hipMemcpyAsync(val, d + 5 * N, (size_t)ncand * sizeof(float), hipMemcpyDeviceToHost, c->stream);
    }
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    hipFree(d);
    hipFree(keys);
    if (rc) return rc;
    if (e != hipSuccess) return fail(c, KLT_ERR_DEVICE, hipGetErrorString(e));
    return KLT_OK;
}

int klt_extract_patch_f32(klt_ctx *c, const float *img, int ncols, int nrows, float x, float y, int width, int height, float *patch)
{
    if (!c || !img || !patch) return fail(c, KLT_ERR_ARG, "null argument");
    if (ncols <= 0 || nrows <= 0 || (long long)ncols * nrows > kMaxFramePixels) return fail(c, KLT_ERR_ARG, "bad image geometry");
    // trackFeaturesUtils.pyx:38-49 swaps the roles of rows and columns: only square patches are defined behaviour there
    if (width != height || width < 1 || width > 31) return fail(c, KLT_ERR_ARG, "square patches of side 1..31 only");
    HIPCHK(c, hipSetDevice(c->device));
    const size_t N = (size_t)ncols * nrows, n = (size_t)width * width;
    float *d = nullptr;
    HIPCHK(c, hipMalloc((void **)&d, (N + n + 1) * sizeof(float)));
    int bad = 0;
    hipError_t e = hipMemcpyAsync(d, img, N * sizeof(float), hipMemcpyHostToDevice, c->stream);
    if (e == hipSuccess) {
        launch_extract_patch(c->stream, d, ncols, nrows, x, y, width, d + N, (int *)(d + N + n));
        e = hipMemcpyAsync(patch, d + N, n * sizeof(float), hipMemcpyDeviceToHost, c->stream);
    }
    if (e == hipSuccess) e = hipMemcpyAsync(&bad, d + N + n, sizeof(int), hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    hipFree(d);
    if (e != hipSuccess) return fail(c, KLT_ERR_DEVICE, hipGetErrorString(e));
    if (bad) return fail(c, KLT_ERR_ARG, "patch footprint leaves the image (the reference asserts: trackFeaturesUtils.pyx:35)");
    return KLT_OK;
}

int klt_track_iterate_f32(klt_ctx *c, float x2, float y2, const float *gradx_patch, const float *grady_patch, const float *img_patch,
                          int width, int height, const float *img2, const float *gradx2, const float *grady2, int ncols, int nrows,
                          float step_factor, float min_determinant, float min_displacement, int max_iterations,
                          float *x2_out, float *y2_out, int *status, int *iterations)
{
    if (!c || !gradx_patch || !grady_patch || !img_patch || !img2 || !gradx2 || !grady2) return fail(c, KLT_ERR_ARG, "null argument");
    if (ncols <= 0 || nrows <= 0 || (long long)ncols * nrows > kMaxFramePixels) return fail(c, KLT_ERR_ARG, "bad image geometry");
    // _computeGradientSum strides the jacobian by shape[0] (trackFeaturesUtils.pyx:128): square windows only
    if (width != height || width < 1 || width > 31) return fail(c, KLT_ERR_ARG, "square windows of side 1..31 only");
    HIPCHK(c, hipSetDevice(c->device));
    const size_t N = (size_t)ncols * nrows, n = (size_t)width * width;
    float *d = nullptr;                         // three planes | three patches | result
    HIPCHK(c, hipMalloc((void **)&d, (3 * N + 3 * n + 4) * sizeof(float)));
    float res[4] = {0, 0, 0, 0};
    const float *src[6] = {img2, gradx2, grady2, gradx_patch, grady_patch, img_patch};
    const size_t off[6] = {0, N, 2 * N, 3 * N, 3 * N + n, 3 * N + 2 * n}, len[6] = {N, N, N, n, n, n};
    hipError_t e = hipSuccess;
    for (int k = 0; k < 6 && e == hipSuccess; k++)
        e = hipMemcpyAsync(d + off[k], src[k], len[k] * sizeof(float), hipMemcpyHostToDevice, c->stream);
    if (e == hipSuccess) {
        launch_track_iterate(c->stream, d + off[3], d + off[4], d + off[5], d, d + N, d + 2 * N, ncols, nrows, width, x2, y2, step_factor,
                             min_determinant, min_displacement, max_iterations, d + 3 * N + 3 * n);
        e = hipMemcpyAsync(res, d + 3 * N + 3 * n, sizeof(res), hipMemcpyDeviceToHost, c->stream);
    }
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    hipFree(d);
    if (e != hipSuccess) return fail(c, KLT_ERR_DEVICE, hipGetErrorString(e));
    if (x2_out) *x2_out = res[0];
    if (y2_out) *y2_out = res[1];
    if (status) *status = (int)res[2];
    if (iterations) *iterations = (int)res[3];
    return KLT_OK;
}

// ------------------------------------------------------------------------- standalone convolutions
int klt_smooth_f32(klt_ctx *c, const float *src, int ncols, int nrows, const double *gauss, int ng, float *dst)
{
    if (!c || !src || !dst || !gauss) return fail(c, KLT_ERR_ARG, "null argument");
    if (ng < 1 || ng > KLT_MAX_KERNEL_WIDTH || !(ng & 1)) return fail(c, KLT_ERR_ARG, "tap count must be odd and at most 71");
    if (ncols <= 0 || nrows <= 0 || (long long)ncols * nrows > kMaxFramePixels) return fail(c, KLT_ERR_ARG, "bad image geometry");
    HIPCHK(c, hipSetDevice(c->device));
    const size_t N = (size_t)ncols * nrows;
    if (int rc = ensure_tmp(c, N)) return rc;
    float *d_in = nullptr;
    HIPCHK(c, hipMalloc((void **)&d_in, 2 * N * sizeof(float)));
    float *d_out = d_in + N;
    Taps g;
    make_taps(gauss, ng, g);
    hipError_t e = hipMemcpyAsync(d_in, src, N * sizeof(float), hipMemcpyHostToDevice, c->stream);
    if (e == hipSuccess) {
        launch_hconv_f32(c->stream, d_in, ncols, nrows, c->tmpA, nullptr, ncols, 1, 0, g, nullptr);
        launch_vconv(c->stream, c->tmpA, nullptr, ncols, nrows, d_out, nullptr, nrows, 1, 0, g, nullptr);
        e = hipMemcpyAsync(dst, d_out, N * sizeof(float), hipMemcpyDeviceToHost, c->stream);
    }
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    hipFree(d_in);
    if (e != hipSuccess) return fail(c, KLT_ERR_DEVICE, hipGetErrorString(e));
    return KLT_OK;
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
    HIPCHK(c, hipMalloc((void **)&d_in, 3 * N * sizeof(float)));
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
    HIPCHK(c, hipMalloc((void **)&d_in, (N + total) * sizeof(float)));
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

// ------------------------------------------------------------------------------------------ experiment: a frame as a HIP graph
#ifdef KLT_GRAPH_PROBE
// Not part of include/klt_gpu.h and not in the product library: tools/graph_frame_probe.py compiles this file with -DKLT_GRAPH_PROBE into a
// private copy (profiles/README.md, "HIP graphs").  op 0: every stream idle, the cross-stream bookkeeping forgotten (no event recorded before the capture
// is waited for inside it), capture begins on the main stream; the caller then enqueues ONE frame's work through the ordinary *_async
// entry points -- the build first, so that the build stream forks off the main stream at the graph's root.  op 1: the build stream joins,
// the capture ends, the graph is instantiated; the selection the capture left pending is dropped (its launches are in the graph).
// op 2: hipGraphLaunch on the main stream.  op 3: destroy.  Returns the node count from op 1.
int klt_debug_graph(klt_ctx *c, int op)
{
    if (!c) return KLT_ERR_ARG;
    HIPCHK(c, hipSetDevice(c->device));
    auto forget = [&]() {
        for (Slot &s : c->slots) { s.upload_pending = s.built_pending = s.read_valid = s.consumed_valid = s.consumed_alt_valid = false; }
        c->ring_serial += kEventRing;                 // every event handed out so far counts as re-used: nobody waits for it any more
        c->last_build_on_bstream = -1;                // the next build on the build stream orders itself behind the main stream
        c->waited_built_serial = ~0ull;
    };
    if (op == 0) {
        if (c->capturing) return fail(c, KLT_ERR_STATE, "already capturing");
        if (int rc = sync_all(c)) return rc;
        forget();
        HIPCHK(c, hipStreamBeginCapture(c->stream, hipStreamCaptureModeThreadLocal));
        c->capturing = true;
        return KLT_OK;
    }
    if (op == 1) {
        if (!c->capturing) return fail(c, KLT_ERR_STATE, "not capturing");
        c->capturing = false;
        hipError_t e = hipSuccess;
        if (c->bstream && c->last_build_on_bstream == 1) {         // the build stream took part: its tail joins the main stream
            hipEvent_t j;
            if (int rc = fresh_event(c, &j)) return rc;
            e = hipEventRecord(j, c->bstream);
            if (e == hipSuccess) e = hipStreamWaitEvent(c->stream, j, 0);
        }
        hipGraph_t g = nullptr;
        const hipError_t e2 = hipStreamEndCapture(c->stream, &g);
        c->sel_job.reset();
        forget();
        if (e != hipSuccess) return fail(c, KLT_ERR_DEVICE, std::string("join: ") + hipGetErrorString(e));
        if (e2 != hipSuccess) return fail(c, KLT_ERR_DEVICE, std::string("hipStreamEndCapture: ") + hipGetErrorString(e2));
        size_t nodes = 0;
        hipGraphGetNodes(g, nullptr, &nodes);
        if (c->probe_graph) { hipGraphExecDestroy(c->probe_graph); c->probe_graph = nullptr; }
        const hipError_t e3 = hipGraphInstantiate(&c->probe_graph, g, nullptr, nullptr, 0);
        hipGraphDestroy(g);
        if (e3 != hipSuccess) return fail(c, KLT_ERR_DEVICE, std::string("hipGraphInstantiate: ") + hipGetErrorString(e3));
        return (int)nodes;
    }
    if (op == 2) {
        if (!c->probe_graph) return fail(c, KLT_ERR_STATE, "no graph");
        HIPCHK(c, hipGraphLaunch(c->probe_graph, c->stream));
        return KLT_OK;
    }
    if (op == 3) {
        if (c->probe_graph) { hipGraphExecDestroy(c->probe_graph); c->probe_graph = nullptr; }
        return KLT_OK;
    }
    return fail(c, KLT_ERR_ARG, "unknown op");
}
#endif  // KLT_GRAPH_PROBE

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
