// klt_context.h -- what the files of the C ABI (api_*.hip) share: the context object behind `klt_ctx` with its slots, feature buffers and
// selection state, the error / timing helpers, and the declarations of the helpers one file defines and another uses.  Host-side C++
// only; the kernels and their launchers are declared in klt_internal.h.  Everything lives in namespace kltapi except klt_ctx itself
// (the opaque type of include/klt_gpu.h).
//
//   api_context.hip   lifetime, parameters, options, slots' state, events and buffers every other file leans on, timing
//   api_frames.hip    frames in (uploads, pinned / device memory, adoption), pyramid build, stand-alone convolutions, plane read-back
//   api_featbuf.hip   feature buffers: copies either way, views, host-mapped buffers
//   api_select.hip    selection: score preparation, the begin / finish protocol, the given-list walk, its inspection hooks
//   api_track.hip     tracker launches (single pair, batched), affine consistency check, iteration counters
//   api_comm.hip      the multi-GPU entry points (RCCL itself: comm.hip)
//   api_compat.hip    the reference's literal native boundary (one call = one wavefront), the HIP-graph experiment
#pragma once
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

struct klt_ctx;

namespace kltapi {


enum Family { F_SMOOTH_GRAD, F_PYR_REDUCE, F_GRAD, F_SMOOTH_H, F_SMOOTH_V, F_PYR_H, F_PYR_V, F_GRAD_H, F_GRAD_V, F_TRACK,
              F_SAT_ROWS, F_SAT_COLS, F_EIGEN, F_SORT, F_NMS, F_SEED, F_AFFINE, F_COUNT };
inline const char *const kFamilyName[F_COUNT] = {"smooth_grad_l0", "pyramid_reduce", "gradients", "smooth_h", "smooth_v", "pyramid_h",
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
    bool f32_in_raw = false;          // klt_upload_f32_async: the f32 frame lives in the alternating raw buffers (`u8`, 4 bytes per pixel), not in `f32`
};

inline const uint8_t *raw8(const Slot *s) { return s->u8_ext ? s->u8_ext : s->u8; }
inline const float *rawf(const Slot *s) { return s->f32_in_raw ? reinterpret_cast<const float *>(s->u8) : s->f32; }

struct FeatBuf { klt_feat *d = nullptr; int cap = 0; bool view = false; hipEvent_t comm_done = nullptr; /* last collective that touched it */ };

struct AffState { klt_affine_rec *rec = nullptr; float *tpl = nullptr; int n = 0, tn = 0; };

struct Timed { int fam; hipEvent_t a, b; double bytes; };

extern std::string g_create_error;      // klt_create has no context to report into (api_context.hip)


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


}  // namespace kltapi

using namespace kltapi;      // (this header is for the api_*.hip files only)

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
    int fail_alloc_in = -1;                   // KLT_OPT_FAIL_ALLOC_AFTER (test hook): >= 0 counts allocations down, the one that finds 0 fails
};

namespace kltapi {

int fail(klt_ctx *c, int code, const std::string &msg);      // sets the context's message, returns `code`

#define HIPCHK(c, call)                                                                              \
    do {                                                                                             \
        hipError_t e_ = (call);                                                                      \
        if (e_ != hipSuccess)                                                                        \
            return fail((c), KLT_ERR_DEVICE, std::string(#call) + ": " + hipGetErrorString(e_));     \
    } while (0)

// Every device / pinned-host allocation of the library goes through these two (api_context.hip).  Out of memory is an ANSWER, not a device
// error: KLT_ERR_NOMEM with the size that was asked for, the pointer left null, and the runtime's sticky last-error cleared -- otherwise
// the next HIPCHK(c, hipGetLastError()) behind a launch would report a stale out-of-memory after the caller has already dealt with it.
int dev_alloc(klt_ctx *c, void **p, size_t bytes, const char *what);
int host_alloc(klt_ctx *c, void **p, size_t bytes, const char *what);
#define DEVALLOC(c, ptr, bytes)                                                                       \
    do {                                                                                             \
        if (int rc_ = kltapi::dev_alloc((c), (void **)&(ptr), (bytes), #ptr)) return rc_;            \
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

// ---- api_context.hip
int drain_timers(klt_ctx *c);
int sync_all(klt_ctx *c);
int ensure_tmp(klt_ctx *c, size_t pixels);
int ensure_h1(klt_ctx *c, size_t floats);
int get_slot(klt_ctx *c, int slot, Slot **out, bool create);
int get_fb(klt_ctx *c, int fb, int n, FeatBuf **out);
int fresh_event(klt_ctx *c, hipEvent_t *out, uint64_t *serial = nullptr);
bool event_live(const klt_ctx *c, uint64_t serial);
int wait_upload(klt_ctx *c, Slot *s, hipStream_t consumer);
int mark_consumed(klt_ctx *c, Slot *const *slots, int n, hipStream_t reader);
int mark_read(klt_ctx *c, Slot *const *slots, int n);
int wait_built(klt_ctx *c, Slot *s);
int check_ready(klt_ctx *c);

template <typename T>
int ensure(klt_ctx *c, T *&ptr, size_t &cap, size_t want)
{
    if (want <= cap && ptr) return 0;
    if (c->capturing) return fail(c, KLT_ERR_STATE, "a buffer would have to grow during a stream capture");
    if (ptr) { if (int rc = sync_all(c)) return rc; HIPCHK(c, hipFree(ptr)); ptr = nullptr; cap = 0; }
    DEVALLOC(c, ptr, want * sizeof(T));
    cap = want;
    return 0;
}

// ---- api_frames.hip
// The kernels address a plane with 32-bit BYTE offsets into a buffer descriptor of 2 GB (raw buffer operations, klt_internal.h), and the
// largest plane is the interleaved gradient plane of level 0 with 8 bytes per pixel: a frame must stay below 2^28 pixels.
// 2^28 - 1 pixels is a 16 384 x 16 383 frame; the largest frame of the test suite is 7680 x 4320.
constexpr long long kMaxFramePixels = (1LL << 28) - 1;
void make_taps(const double *k, int n, Taps &t);
int upload_raw(klt_ctx *c, int slot, const void *px, int ncols, int nrows, int pitch, int kind);
int layout_pyramid(klt_ctx *c, Slot *s);
int enqueue_smooth_raw(klt_ctx *c, Slot *s, float *dst);
int enqueue_gradients(klt_ctx *c, const float *img, int nc, int nr, float *gx, float *gy);
bool fused_smooth_ok(const klt_ctx *c);
bool fused_grad_ok(const klt_ctx *c);
int enqueue_fused_smooth_grad(klt_ctx *c, int batch, const void *const *raw, int raw_kind, float *const *img,
                              float *const *gx, float *const *gy, int nc, int nr, bool *fused_h1 = nullptr);
int enqueue_fused_grad(klt_ctx *c, int batch, const float *const *img, float *const *gx, float *const *gy, int nc, int nr,
                       bool u8_input = false);
int build_pyramids_batch(klt_ctx *c, const int *slot_ids, int n);
int download_plane(klt_ctx *c, const float *src, int stride, size_t cnt, float *dst);      // every stride-th float of a device plane to the host

// ---- api_select.hip
// the three summed-area tables of a gradient pair (goodFeaturesUtils.pyx:49-51); rows_only: the column pass is left to the caller
int enqueue_sat(klt_ctx *c, hipStream_t st, const float *gx, const float *gy, float *sat, int nc, int nr, bool rows_only = false);

}  // namespace kltapi
