// Multi-GPU plumbing of libkltgpu.so: one process per GPU, RCCL over xGMI.
//
// The KLT path shards by frame pair -- ranks never exchange pixels or pyramids -- so the only
// communication is the gather of 16-byte feature records (klt_feat) at the end of a shard, plus a
// barrier / max-reduction for timing.  The reference has nothing here (its only concurrency is the
// GUI's multiprocessing queue, examplegui.py:22-29, out of scope).
//
// librccl is opened with dlopen the first time a communicator is needed: a single-GPU process never
// loads it, and the library has no link-time dependency on it.  Collectives run on a side stream of
// the communicator that is event-ordered behind the producer stream (the context's stream), so the
// tracker's stream never waits for RCCL unless the caller asks (comm_fence).
#include <dlfcn.h>
#include <rccl/rccl.h>

#include <unistd.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "klt_internal.h"

namespace {

struct Rccl {
    void *handle = nullptr;
    decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
    decltype(&ncclCommInitRank) CommInitRank = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclCommAbort) CommAbort = nullptr;          // optional: used to tear down a communicator whose collective can never complete
    decltype(&ncclAllGather) AllGather = nullptr;
    decltype(&ncclAllReduce) AllReduce = nullptr;
    decltype(&ncclSend) Send = nullptr;
    decltype(&ncclRecv) Recv = nullptr;
    decltype(&ncclGroupStart) GroupStart = nullptr;
    decltype(&ncclGroupEnd) GroupEnd = nullptr;
    decltype(&ncclGetErrorString) GetErrorString = nullptr;
    std::string error;
};

Rccl g_rccl;

template <typename F>
bool sym(F &fn, const char *name)
{
    fn = reinterpret_cast<F>(dlsym(g_rccl.handle, name));
    if (!fn) g_rccl.error = std::string("librccl lacks ") + name;
    return fn != nullptr;
}

bool load_rccl(std::string &err)
{
    if (g_rccl.handle && g_rccl.error.empty()) return true;
    if (!g_rccl.handle) {
        // KLT_RCCL_LIB names THE library to use (no fallback to the system's: a wrong path must not go unnoticed)
        const char *forced = getenv("KLT_RCCL_LIB");
        // the ROCm installation's own library first, by path: a bare soname would resolve to whatever copy the process has already
        // loaded -- e.g. the librccl a PyTorch wheel bundles next to its own HIP runtime, which is not the runtime this library talks to
        // ("ncclCommInitRank: unhandled cuda error" in a process that had imported torch)
        std::string rocm = getenv("ROCM_PATH") ? getenv("ROCM_PATH") : "/opt/rocm";
        rocm += "/lib/librccl.so.1";
        const char *defaults[] = {rocm.c_str(), "librccl.so.1", "librccl.so"};
        std::vector<const char *> names;
        if (forced && *forced) names.push_back(forced);
        else names.assign(defaults, defaults + 3);
        std::string tried;
        for (const char *n : names) {
            g_rccl.handle = dlopen(n, RTLD_NOW | RTLD_LOCAL);
            if (g_rccl.handle) break;
            const char *m = dlerror();                    // (dlerror() clears its state: read it once per failure)
            tried += std::string(tried.empty() ? "" : "; ") + n + ": " + (m ? m : "not found");
        }
        if (!g_rccl.handle) {
            err = "cannot open librccl (" + tried + ")";
            return false;
        }
        g_rccl.error.clear();
        sym(g_rccl.GetUniqueId, "ncclGetUniqueId") && sym(g_rccl.CommInitRank, "ncclCommInitRank") &&
            sym(g_rccl.CommDestroy, "ncclCommDestroy") && sym(g_rccl.AllGather, "ncclAllGather") &&
            sym(g_rccl.AllReduce, "ncclAllReduce") && sym(g_rccl.Send, "ncclSend") && sym(g_rccl.Recv, "ncclRecv") &&
            sym(g_rccl.GroupStart, "ncclGroupStart") && sym(g_rccl.GroupEnd, "ncclGroupEnd") &&
            sym(g_rccl.GetErrorString, "ncclGetErrorString");
        g_rccl.CommAbort = reinterpret_cast<decltype(g_rccl.CommAbort)>(dlsym(g_rccl.handle, "ncclCommAbort"));
    }
    if (!g_rccl.error.empty()) { err = g_rccl.error; return false; }
    return true;
}

}  // namespace

struct KltComm {
    int device = 0, nranks = 1, rank = 0;
    ncclComm_t comm = nullptr;
    hipStream_t side = nullptr;            // collectives run here
    std::vector<hipEvent_t> ring;          // ordering events; never re-recorded while a waiter may be queued
    size_t ring_next = 0;
    hipEvent_t last_done = nullptr;        // end of the most recent collective
    double *scratch = nullptr;             // device scratch for the small reductions (16 doubles)
    double timeout_ms = 300000.0;          // host-side waits give up after this long (KLT_COMM_TIMEOUT_MS / comm_set_timeout; <= 0: never)
    bool poisoned = false;                 // a host-side wait timed out: a collective on the side stream can never complete.  Nothing waits
                                           // for that stream again -- not even the teardown -- and every further call fails at once
};

// Collectives of DIFFERENT communicators of one process (bench.py gives every context its own) must not run side by side: two RCCL
// kernels on one GPU that each wait for their peers on the other GPUs can starve each other when the GPUs start them in different order.
// Every collective therefore waits (on the device) for the previous collective of this process, whatever communicator issued it: one at
// a time, in issue order -- which is the same on every rank.  The records are small (megabytes per step); the chain costs nothing.
namespace {
hipEvent_t g_chain_done = nullptr;       // end of the most recent collective of this process
const KltComm *g_chain_owner = nullptr;  // whose ring the event lives in
}

#define COMM_HIP(call)                                                                  \
    do {                                                                                \
        hipError_t e_ = (call);                                                         \
        if (e_ != hipSuccess) { err = std::string(#call) + ": " + hipGetErrorString(e_); return KLT_ERR_DEVICE; } \
    } while (0)
#define COMM_NCCL(call)                                                                 \
    do {                                                                                \
        ncclResult_t r_ = (call);                                                       \
        if (r_ != ncclSuccess) { err = std::string(#call) + ": " + g_rccl.GetErrorString(r_); return KLT_ERR_DEVICE; } \
    } while (0)

static int comm_event(KltComm *k, hipEvent_t *out, std::string &err)
{
    constexpr size_t kRing = 128;
    if (k->ring.size() < kRing) {
        hipEvent_t e = nullptr;
        COMM_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
        k->ring.push_back(e);
        *out = e;
        return 0;
    }
    *out = k->ring[k->ring_next];
    k->ring_next = (k->ring_next + 1) % kRing;
    return 0;
}

int comm_unique_id(void *out128, std::string &err)
{
    if (!out128) { err = "null argument"; return KLT_ERR_ARG; }
    if (!load_rccl(err)) return KLT_ERR_DEVICE;
    ncclUniqueId id;
    COMM_NCCL(g_rccl.GetUniqueId(&id));
    static_assert(sizeof(id) == KLT_COMM_ID_BYTES, "ncclUniqueId is 128 bytes");
    std::memcpy(out128, &id, sizeof(id));
    return 0;
}

int comm_create(int device, int nranks, int rank, const void *unique_id, KltComm **out, std::string &err)
{
    if (!unique_id || !out || nranks < 1 || rank < 0 || rank >= nranks) { err = "bad communicator arguments"; return KLT_ERR_ARG; }
    if (!load_rccl(err)) return KLT_ERR_DEVICE;
    COMM_HIP(hipSetDevice(device));
    KltComm *k = new KltComm();
    k->device = device; k->nranks = nranks; k->rank = rank;
    if (const char *t = getenv("KLT_COMM_TIMEOUT_MS")) k->timeout_ms = atof(t);
    ncclUniqueId id;
    std::memcpy(&id, unique_id, sizeof(id));
    ncclResult_t r = g_rccl.CommInitRank(&k->comm, nranks, id, rank);
    if (r != ncclSuccess) {
        err = std::string("ncclCommInitRank: ") + g_rccl.GetErrorString(r);
        delete k;
        return KLT_ERR_DEVICE;
    }
    hipError_t e = hipStreamCreateWithFlags(&k->side, hipStreamNonBlocking);
    if (e == hipSuccess) e = hipMalloc((void **)&k->scratch, 16 * sizeof(double));
    if (e != hipSuccess) {
        err = std::string("communicator stream / scratch: ") + hipGetErrorString(e);
        comm_destroy(k);
        return KLT_ERR_DEVICE;
    }
    *out = k;
    return 0;
}

bool comm_poisoned(const KltComm *k) { return k && k->poisoned; }

void comm_destroy(KltComm *k)
{
    if (!k) return;
    hipSetDevice(k->device);
    if (k->poisoned) {
        // The side stream holds a collective whose peer is gone: synchronising it, or ncclCommDestroy (which does), would hang the
        // rank that is trying to report and exit.  Abort the communicator where the library can, and leak the stream, its events and
        // the scratch -- the process is on its way out (comm_wait's caller exits non-zero; nothing is restarted in place).
        if (g_chain_owner == k) { g_chain_done = nullptr; g_chain_owner = nullptr; }
        if (k->comm && g_rccl.CommAbort) g_rccl.CommAbort(k->comm);
        delete k;
        return;
    }
    if (k->side) hipStreamSynchronize(k->side);
    if (g_chain_owner == k) { g_chain_done = nullptr; g_chain_owner = nullptr; }      // (its events are about to go; the stream above is idle)
    if (k->comm && g_rccl.CommDestroy) g_rccl.CommDestroy(k->comm);
    for (hipEvent_t e : k->ring) hipEventDestroy(e);
    if (k->scratch) hipFree(k->scratch);
    if (k->side) hipStreamDestroy(k->side);
    delete k;
}

hipEvent_t comm_last_done(const KltComm *k) { return k ? k->last_done : nullptr; }
int comm_nranks(const KltComm *k) { return k ? k->nranks : 1; }
int comm_rank(const KltComm *k) { return k ? k->rank : 0; }

// side stream waits for everything enqueued on `producer` so far
static int comm_order_behind(KltComm *k, hipStream_t producer, std::string &err)
{
    if (k->poisoned) { err = "the communicator is unusable: an earlier wait for a collective timed out (a peer is gone or stuck)"; return KLT_ERR_TIMEOUT; }
    hipEvent_t ready;
    if (int rc = comm_event(k, &ready, err)) return rc;
    COMM_HIP(hipEventRecord(ready, producer));
    COMM_HIP(hipStreamWaitEvent(k->side, ready, 0));
    if (g_chain_done && g_chain_owner != k) COMM_HIP(hipStreamWaitEvent(k->side, g_chain_done, 0));     // (the own side stream is in order anyway)
    return 0;
}

static int comm_mark_done(KltComm *k, std::string &err)
{
    hipEvent_t done;
    if (int rc = comm_event(k, &done, err)) return rc;
    COMM_HIP(hipEventRecord(done, k->side));
    k->last_done = done;
    g_chain_done = done;
    g_chain_owner = k;
    return 0;
}

int comm_allgather(KltComm *k, hipStream_t producer, const void *src, void *dst, size_t bytes, std::string &err)
{
    COMM_HIP(hipSetDevice(k->device));
    if (int rc = comm_order_behind(k, producer, err)) return rc;
    COMM_NCCL(g_rccl.AllGather(src, dst, bytes, ncclUint8, k->comm, k->side));
    return comm_mark_done(k, err);
}

// every rank sends `bytes` to `root`, which receives them in rank order (its own part is a device copy)
int comm_gather(KltComm *k, hipStream_t producer, const void *src, void *dst, size_t bytes, int root, std::string &err)
{
    if (root < 0 || root >= k->nranks) { err = "gather root out of range"; return KLT_ERR_ARG; }
    COMM_HIP(hipSetDevice(k->device));
    if (int rc = comm_order_behind(k, producer, err)) return rc;
    if (k->rank == root) {
        if (!dst) { err = "the gather root needs a destination buffer"; return KLT_ERR_ARG; }
        COMM_HIP(hipMemcpyAsync((char *)dst + (size_t)root * bytes, src, bytes, hipMemcpyDeviceToDevice, k->side));
        if (k->nranks > 1) {
            COMM_NCCL(g_rccl.GroupStart());
            ncclResult_t first = ncclSuccess;             // an open group is always closed, whatever failed inside it
            for (int r = 0; r < k->nranks && first == ncclSuccess; r++)
                if (r != root) first = g_rccl.Recv((char *)dst + (size_t)r * bytes, bytes, ncclUint8, r, k->comm, k->side);
            const ncclResult_t end = g_rccl.GroupEnd();
            COMM_NCCL(first);
            COMM_NCCL(end);
        }
    } else {
        COMM_NCCL(g_rccl.Send(src, bytes, ncclUint8, root, k->comm, k->side));
    }
    return comm_mark_done(k, err);
}

// the same with a count per rank (shards of unequal size, SURVEY 8(e): any contiguous / round-robin partition): rank r contributes
// counts[r] bytes, the root receives them back to back in rank order.  Every rank passes the same table.
int comm_gatherv(KltComm *k, hipStream_t producer, const void *src, void *dst, const size_t *counts, int root, std::string &err)
{
    if (root < 0 || root >= k->nranks || !counts) { err = "gatherv: bad root / counts"; return KLT_ERR_ARG; }
    COMM_HIP(hipSetDevice(k->device));
    if (int rc = comm_order_behind(k, producer, err)) return rc;
    const size_t mine = counts[k->rank];
    if (k->rank == root) {
        if (!dst) { err = "the gather root needs a destination buffer"; return KLT_ERR_ARG; }
        std::vector<size_t> off((size_t)k->nranks + 1, 0);
        for (int r = 0; r < k->nranks; r++) off[r + 1] = off[r] + counts[r];
        if (mine) COMM_HIP(hipMemcpyAsync((char *)dst + off[root], src, mine, hipMemcpyDeviceToDevice, k->side));
        if (k->nranks > 1) {
            COMM_NCCL(g_rccl.GroupStart());
            ncclResult_t first = ncclSuccess;
            for (int r = 0; r < k->nranks && first == ncclSuccess; r++)
                if (r != root && counts[r]) first = g_rccl.Recv((char *)dst + off[r], counts[r], ncclUint8, r, k->comm, k->side);
            const ncclResult_t end = g_rccl.GroupEnd();
            COMM_NCCL(first);
            COMM_NCCL(end);
        }
    } else if (mine) {
        COMM_NCCL(g_rccl.Send(src, mine, ncclUint8, root, k->comm, k->side));
    }
    return comm_mark_done(k, err);
}

// point to point, one group: `bytes` at src go to rank `to` and / or `bytes` from rank `from` arrive at dst (-1: no such side).
// Sending to oneself (to == from == own rank) is a device copy.
int comm_sendrecv(KltComm *k, hipStream_t producer, const void *src, int to, void *dst, int from, size_t bytes, std::string &err)
{
    if (to >= k->nranks || from >= k->nranks || (to < 0 && from < 0)) { err = "send / receive peer out of range"; return KLT_ERR_ARG; }
    if ((to >= 0 && !src) || (from >= 0 && !dst)) { err = "send / receive buffer missing"; return KLT_ERR_ARG; }
    if ((to == k->rank) != (from == k->rank)) { err = "a rank can only send to itself what it receives from itself"; return KLT_ERR_ARG; }
    COMM_HIP(hipSetDevice(k->device));
    if (int rc = comm_order_behind(k, producer, err)) return rc;
    if (to == k->rank) {
        COMM_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, k->side));
    } else {
        COMM_NCCL(g_rccl.GroupStart());
        ncclResult_t first = ncclSuccess;
        if (to >= 0) first = g_rccl.Send(src, bytes, ncclUint8, to, k->comm, k->side);
        if (from >= 0 && first == ncclSuccess) first = g_rccl.Recv(dst, bytes, ncclUint8, from, k->comm, k->side);
        const ncclResult_t end = g_rccl.GroupEnd();
        COMM_NCCL(first);
        COMM_NCCL(end);
    }
    return comm_mark_done(k, err);
}

// `consumer` waits (on the device) for every collective issued so far
int comm_fence(KltComm *k, hipStream_t consumer, std::string &err)
{
    if (k->last_done) COMM_HIP(hipStreamWaitEvent(consumer, k->last_done, 0));
    return 0;
}

void comm_set_timeout(KltComm *k, double ms) { if (k) k->timeout_ms = ms; }

// The host waits for the side stream -- but never for ever: a peer that died leaves a collective that cannot complete, and a rank
// stuck in hipStreamSynchronize cannot even report it.  Polls (spinning for the first 200 us: the timing barrier of bench.py sits
// here) and gives up with KLT_ERR_TIMEOUT after timeout_ms; the caller is expected to exit non-zero (a process that has touched
// the GPU is never restarted in place).
int comm_wait(KltComm *k, std::string &err)
{
    if (k->poisoned) { err = "the communicator is unusable: an earlier wait for a collective timed out (a peer is gone or stuck)"; return KLT_ERR_TIMEOUT; }
    COMM_HIP(hipSetDevice(k->device));
    if (k->timeout_ms <= 0) { COMM_HIP(hipStreamSynchronize(k->side)); return 0; }
    const auto t0 = std::chrono::steady_clock::now();
    for (;;) {
        const hipError_t q = hipStreamQuery(k->side);
        if (q == hipSuccess) return 0;
        (void)hipGetLastError();                 // "not ready" is an answer, not an error: it must not surface in a later hipGetLastError check
        if (q != hipErrorNotReady) { err = std::string("hipStreamQuery(side stream): ") + hipGetErrorString(q); return KLT_ERR_DEVICE; }
        const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
        if (ms > k->timeout_ms) {
            char buf[160];
            std::snprintf(buf, sizeof(buf), "collective still pending after %.0f ms (rank %d of %d): a peer is gone or stuck", ms, k->rank, k->nranks);
            err = buf;
            k->poisoned = true;
            return KLT_ERR_TIMEOUT;
        }
        if (ms > 0.2) usleep(50);
    }
}

// max over ranks of up to 16 doubles, host in / host out, synchronous (bench timing: MAX over ranks); doubles as the barrier
int comm_allreduce_max(KltComm *k, double *inout, int n, std::string &err)
{
    if (!inout || n < 1 || n > 16) { err = "allreduce takes 1..16 doubles"; return KLT_ERR_ARG; }
    if (k->poisoned) { err = "the communicator is unusable: an earlier wait for a collective timed out (a peer is gone or stuck)"; return KLT_ERR_TIMEOUT; }
    COMM_HIP(hipSetDevice(k->device));
    if (g_chain_done && g_chain_owner != k) COMM_HIP(hipStreamWaitEvent(k->side, g_chain_done, 0));     // one collective of the process at a time
    COMM_HIP(hipMemcpyAsync(k->scratch, inout, n * sizeof(double), hipMemcpyHostToDevice, k->side));
    COMM_NCCL(g_rccl.AllReduce(k->scratch, k->scratch, (size_t)n, ncclDouble, ncclMax, k->comm, k->side));
    COMM_HIP(hipMemcpyAsync(inout, k->scratch, n * sizeof(double), hipMemcpyDeviceToHost, k->side));
    return comm_wait(k, err);
}
