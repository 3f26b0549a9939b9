// host_pool.hip -- host side of the frame ingest: comparing a frame with the copy its slot was filled from, and staging a frame into
// pinned memory, by a few parked worker threads.  Pure host code (no kernel in this file).
//
// Why: the reference converts and rebuilds both images on every KLTTrackFeatures call (trackFeatures.py:146-196).  The Python layer
// keeps a frame's pyramids while a call names an image with EXACTLY the pixels the slot holds, which costs one pass over the frame
// per call and image (2 MB at 1080p): 0.034 ms for one core on the GPU box, next to 0.03-0.05 ms of device work for the tracker --
// with both frames resident the two comparisons ARE the call.  Four lanes bring a frame to about 0.012 ms; the same lanes stage a
// new frame into the pinned buffer the DMA reads (klt_upload_u8_async).
//
// Design: one process-wide pool; a job is a range cut into 128 KB chunks that lanes claim from one atomic word tagged with the job's
// number and closed when the job ends (a lane that wakes late can never claim a chunk of another job); the caller is a lane itself and always finishes the job
// even if no worker shows up; a second caller that finds the pool busy simply does its own work.  Workers spin for ~40 us after a
// job (the second frame's comparison follows the first one's) and then park on a condition variable.  KLT_HOST_THREADS = total
// lanes (default 4, 1 = no workers).
#include "../../include/klt_gpu.h"

#include <pthread.h>

#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <new>
#include <thread>

namespace {

constexpr size_t kChunk = 128 << 10;
constexpr size_t kParallelFrom = 512 << 10;      // below this one core is done before a second one has started

struct Pool {
    std::mutex job_mutex;                        // one job at a time
    std::mutex m;                                // parking
    std::condition_variable cv;
    std::atomic<int> parked{0};
    std::atomic<uint64_t> next{0};               // (job number << 32) | next chunk to claim
    std::atomic<uint32_t> done{0};               // chunks of the current job completed
    std::atomic<int> differ{0};
    // the job (written between jobs only; read by a lane only while it holds a claim of that job)
    std::atomic<int> kind{0};                    // 0 compare, 1 copy, 2 luma (rows of R, G, B, X bytes -> f32: klt_host_luma_rows)
    std::atomic<const uint8_t *> a{nullptr}, b{nullptr};
    std::atomic<uint8_t *> dst{nullptr};
    std::atomic<size_t> bytes{0};
    std::atomic<uint32_t> nchunks{0};
    // a frame given as a table of row addresses (klt_host_compare_rows / klt_host_copy_rows): rows != nullptr, a chunk = rows_per_chunk rows,
    // `b` / `dst` the contiguous side
    std::atomic<const uint8_t *const *> rows{nullptr};
    std::atomic<size_t> row_bytes{0};
    std::atomic<uint32_t> nrows{0}, rows_per_chunk{0};
    uint32_t job_no = 0;                         // guarded by job_mutex
    int workers = 0;                             // guarded by job_mutex
    int lanes = 4;
};

Pool *g_pool = nullptr;
thread_local bool tl_serial = false;            // klt_host_thread_serial: this thread's compares / copies stay on this thread
std::once_flag g_once;

inline void cpu_relax() { __builtin_ia32_pause(); }

// Pillow's rgb2f (libImaging/Convert.c: `(float)L(in) / 1000.0F`, L = 299 R + 587 G + 114 B in integers) for one row of 4-byte pixels
inline void luma_row(float *dst, const uint8_t *px, size_t npx)
{
    for (size_t i = 0; i < npx; i++, px += 4)
        dst[i] = (float)((int)px[0] * 299 + (int)px[1] * 587 + (int)px[2] * 114) / 1000.0f;
}

// claims and runs chunks of job `tag` until none is left (or the tag has moved on); returns the number of chunks this lane completed
uint32_t run_chunks(Pool *p, uint32_t tag)
{
    uint32_t mine = 0;
    for (;;) {
        uint64_t v = p->next.load(std::memory_order_acquire);
        if ((uint32_t)(v >> 32) != tag) return mine;
        const uint32_t idx = (uint32_t)v, n = p->nchunks.load(std::memory_order_relaxed);
        if (idx >= n) return mine;
        const int kind = p->kind.load(std::memory_order_relaxed);
        const uint8_t *a = p->a.load(std::memory_order_relaxed), *b = p->b.load(std::memory_order_relaxed);
        uint8_t *dst = p->dst.load(std::memory_order_relaxed);
        const size_t bytes = p->bytes.load(std::memory_order_relaxed);
        const uint8_t *const *rows = p->rows.load(std::memory_order_relaxed);
        const size_t row_bytes = p->row_bytes.load(std::memory_order_relaxed);
        const uint32_t nrows = p->nrows.load(std::memory_order_relaxed), rpc = p->rows_per_chunk.load(std::memory_order_relaxed);
        if (!p->next.compare_exchange_weak(v, v + 1, std::memory_order_acq_rel)) continue;
        // the claim succeeded while the tag was still ours: the fields read above are this job's
        if (rows) {
            const uint32_t r0 = idx * rpc, r1 = r0 + rpc < nrows ? r0 + rpc : nrows;
            for (uint32_t r = r0; r < r1; r++) {
                if (kind == 0) {
                    if (p->differ.load(std::memory_order_relaxed)) break;
                    if (memcmp(rows[r], b + (size_t)r * row_bytes, row_bytes) != 0) p->differ.store(1, std::memory_order_relaxed);
                } else if (kind == 1) {
                    memcpy(dst + (size_t)r * row_bytes, rows[r], row_bytes);
                } else {
                    luma_row(reinterpret_cast<float *>(dst + (size_t)r * row_bytes), rows[r], row_bytes / 4);      // (4 bytes in, 4 bytes out per pixel)
                }
            }
            p->done.fetch_add(1, std::memory_order_release);
            mine++;
            continue;
        }
        const size_t off = (size_t)idx * kChunk, len = bytes - off < kChunk ? bytes - off : kChunk;
        if (kind == 0) {
            if (!p->differ.load(std::memory_order_relaxed) && memcmp(a + off, b + off, len) != 0)
                p->differ.store(1, std::memory_order_relaxed);
        } else {
            memcpy(dst + off, a + off, len);
        }
        p->done.fetch_add(1, std::memory_order_release);
        mine++;
    }
}

void worker(Pool *p)
{
    uint32_t seen = 0;
    for (;;) {
        // spin for a while: the next job usually follows at once
        const auto until = std::chrono::steady_clock::now() + std::chrono::microseconds(40);
        uint32_t tag = (uint32_t)(p->next.load(std::memory_order_acquire) >> 32);
        while (tag == seen && std::chrono::steady_clock::now() < until) {
            for (int i = 0; i < 32; i++) cpu_relax();
            tag = (uint32_t)(p->next.load(std::memory_order_acquire) >> 32);
        }
        if (tag == seen) {
            std::unique_lock<std::mutex> lk(p->m);
            p->parked.fetch_add(1, std::memory_order_seq_cst);
            p->cv.wait(lk, [&] { return (uint32_t)(p->next.load(std::memory_order_acquire) >> 32) != seen; });
            p->parked.fetch_sub(1, std::memory_order_seq_cst);
            tag = (uint32_t)(p->next.load(std::memory_order_acquire) >> 32);
        }
        seen = tag;
        run_chunks(p, tag);
    }
}

Pool *fresh_pool()
{
    Pool *p = new Pool();                        // never destroyed: detached workers may be parked on it at exit
    const char *e = getenv("KLT_HOST_THREADS");
    int lanes = e ? atoi(e) : 4;
    const int hw = (int)std::thread::hardware_concurrency();
    if (hw > 0 && lanes > hw) lanes = hw;
    p->lanes = lanes < 1 ? 1 : (lanes > 16 ? 16 : lanes);
    return p;
}

void in_child_after_fork()
{
    // the worker threads do not exist in the child, and the locks may have been held: a fresh pool (the old one is abandoned)
    if (g_pool) g_pool = fresh_pool();
}

Pool *pool()
{
    std::call_once(g_once, [] {
        g_pool = fresh_pool();
        pthread_atfork(nullptr, nullptr, in_child_after_fork);
    });
    return g_pool;
}

// 0 / 1 for a comparison (equal / different), 0 for a copy.  rows != nullptr: side `a` is nrows rows of row_bytes bytes at rows[0 .. nrows-1]
// (bytes = nrows * row_bytes), `b` / `dst` the contiguous side
int run_job(int kind, const uint8_t *a, const uint8_t *b, uint8_t *dst, size_t bytes, const uint8_t *const *rows = nullptr, uint32_t nrows = 0,
            size_t row_bytes = 0)
{
    Pool *p = bytes >= kParallelFrom && !tl_serial ? pool() : nullptr;
    if (!p || p->lanes <= 1 || !p->job_mutex.try_lock()) {
        if (rows) {
            for (uint32_t r = 0; r < nrows; r++) {
                if (kind == 0) { if (memcmp(rows[r], b + (size_t)r * row_bytes, row_bytes) != 0) return 1; }
                else if (kind == 1) memcpy(dst + (size_t)r * row_bytes, rows[r], row_bytes);
                else luma_row(reinterpret_cast<float *>(dst + (size_t)r * row_bytes), rows[r], row_bytes / 4);
            }
            return 0;
        }
        if (kind == 0) return memcmp(a, b, bytes) != 0;
        memcpy(dst, a, bytes);
        return 0;
    }
    std::lock_guard<std::mutex> job(p->job_mutex, std::adopt_lock);
    while (p->workers < p->lanes - 1) {          // started on first use
        try { std::thread(worker, p).detach(); } catch (...) { break; }
        p->workers++;
    }
    uint32_t rpc = 0;
    if (rows) { rpc = (uint32_t)(kChunk / (row_bytes ? row_bytes : 1)); if (!rpc) rpc = 1; }
    const uint32_t n = rows ? (nrows + rpc - 1) / rpc : (uint32_t)((bytes + kChunk - 1) / kChunk);
    p->rows.store(rows, std::memory_order_relaxed);
    p->row_bytes.store(row_bytes, std::memory_order_relaxed);
    p->nrows.store(nrows, std::memory_order_relaxed);
    p->rows_per_chunk.store(rpc, std::memory_order_relaxed);
    p->kind.store(kind, std::memory_order_relaxed);
    p->a.store(a, std::memory_order_relaxed);
    p->b.store(b, std::memory_order_relaxed);
    p->dst.store(dst, std::memory_order_relaxed);
    p->bytes.store(bytes, std::memory_order_relaxed);
    p->nchunks.store(n, std::memory_order_relaxed);
    p->differ.store(0, std::memory_order_relaxed);
    p->done.store(0, std::memory_order_relaxed);
    const uint32_t tag = ++p->job_no ? p->job_no : ++p->job_no;      // never 0 (a worker's initial `seen`)
    p->next.store((uint64_t)tag << 32, std::memory_order_seq_cst);
    if (p->parked.load(std::memory_order_seq_cst) > 0) {
        std::lock_guard<std::mutex> lk(p->m);
        p->cv.notify_all();
    }
    run_chunks(p, tag);
    while (p->done.load(std::memory_order_acquire) < n) cpu_relax();  // chunks other lanes have claimed
    // closed: no chunk of this job can be claimed any more, whatever the next job writes into the fields above before it publishes
    // its own tag (a lane that wakes late compares its chunk index with the NEW job's chunk count otherwise)
    p->next.store(((uint64_t)tag << 32) | 0xffffffffu, std::memory_order_seq_cst);
    return kind == 0 ? p->differ.load(std::memory_order_relaxed) : 0;
}

}  // namespace

extern "C" {

int klt_host_compare(const void *a, const void *b, size_t bytes)
{
    if ((!a || !b) && bytes) return KLT_ERR_ARG;
    if (!bytes || a == b) return 0;
    return run_job(0, (const uint8_t *)a, (const uint8_t *)b, nullptr, bytes);
}

int klt_host_copy(void *dst, const void *src, size_t bytes)
{
    if ((!dst || !src) && bytes) return KLT_ERR_ARG;
    if (!bytes || dst == src) return KLT_OK;
    return run_job(1, (const uint8_t *)src, nullptr, (uint8_t *)dst, bytes);
}

// rows that follow each other in memory (an image allocated in one block; an image mapped onto a contiguous array) are one range
static bool rows_contiguous(const uint8_t *const *rows, int nrows, size_t row_bytes)
{
    for (int r = 1; r < nrows; r++)
        if (rows[r] != rows[r - 1] + row_bytes) return false;
    return true;
}

int klt_host_compare_rows(const uint8_t *const *rows, int nrows, size_t row_bytes, const void *b)
{
    if (nrows < 0 || ((!rows || !b) && nrows && row_bytes)) return KLT_ERR_ARG;
    if (!nrows || !row_bytes) return 0;
    for (int r = 0; r < nrows; r++) if (!rows[r]) return KLT_ERR_ARG;
    if (rows_contiguous(rows, nrows, row_bytes)) return klt_host_compare(rows[0], b, (size_t)nrows * row_bytes);
    return run_job(0, nullptr, (const uint8_t *)b, nullptr, (size_t)nrows * row_bytes, rows, (uint32_t)nrows, row_bytes);
}

int klt_host_copy_rows(void *dst, const uint8_t *const *rows, int nrows, size_t row_bytes)
{
    if (nrows < 0 || ((!rows || !dst) && nrows && row_bytes)) return KLT_ERR_ARG;
    if (!nrows || !row_bytes) return KLT_OK;
    for (int r = 0; r < nrows; r++) if (!rows[r]) return KLT_ERR_ARG;
    if (rows_contiguous(rows, nrows, row_bytes)) return klt_host_copy(dst, rows[0], (size_t)nrows * row_bytes);
    return run_job(1, nullptr, nullptr, (uint8_t *)dst, (size_t)nrows * row_bytes, rows, (uint32_t)nrows, row_bytes);
}

int klt_host_luma_rows(float *dst, const uint8_t *const *rows, int nrows, int ncols)
{
    if (nrows < 0 || ncols < 0 || ((!rows || !dst) && nrows && ncols)) return KLT_ERR_ARG;
    if (!nrows || !ncols) return KLT_OK;
    for (int r = 0; r < nrows; r++) if (!rows[r]) return KLT_ERR_ARG;
    return run_job(2, nullptr, nullptr, reinterpret_cast<uint8_t *>(dst), (size_t)nrows * ncols * 4, rows, (uint32_t)nrows, (size_t)ncols * 4);
}

int klt_host_sample_rows(const uint8_t *const *rows, int nrows, int ncols, int ystep, int xstep, uint8_t *out, size_t cap)
{
    if (!rows || !out || nrows <= 0 || ncols <= 0 || ystep <= 0 || xstep <= 0) return KLT_ERR_ARG;
    const size_t per_row = ((size_t)ncols + xstep - 1) / xstep, nr = ((size_t)nrows + ystep - 1) / ystep;
    if (per_row * nr > cap) return KLT_ERR_ARG;
    size_t k = 0;
    for (int y = 0; y < nrows; y += ystep) {
        const uint8_t *row = rows[y];
        if (!row) return KLT_ERR_ARG;
        for (int x = 0; x < ncols; x += xstep) out[k++] = row[x];
    }
    return (int)k;
}

int klt_host_lanes(void) { return pool()->lanes; }

int klt_host_thread_serial(int on) { const int was = tl_serial; tl_serial = on != 0; return was; }

}  // extern "C"
