// TEST INFRASTRUCTURE ONLY -- a stand-in for librccl that lets several ranks of libkltgpu.so's multi-GPU path run on ONE GPU.
//
// RCCL refuses two ranks of a communicator on the same device, and the GPU boxes this repository is developed on have one GPU,
// so the N > 1 code of csrc/comm.hip, klt_api.hip and bench.py (per-rank counts of the gather, the feature list as a baton between
// ranks, all-gather of the record tables, barrier / max over ranks) would only ever meet a single rank.  This library exports the ten
// RCCL entry points comm.hip binds, with the same signatures (rccl.h), and moves the bytes through files in a directory all ranks
// share (KLT_STUB_RCCL_DIR).  Point KLT_RCCL_LIB at the built library and give every rank the same device (KLT_RANKS_SHARE_DEVICE=0):
// tests/test_gpu_multirank.py.
//
// ASYNCHRONOUS, like the library it stands in for (round 4; the first version waited for the stream and copied synchronously inside every
// call, which serialised everything and could hide ordering mistakes of the caller):
//   * an entry point only ENQUEUES: device -> pinned copies of what is sent, ONE host function on the caller's stream (hipLaunchHostFunc)
//     that exchanges the files, pinned -> device copies of what is received.  The call returns at once; the stream -- and nothing else --
//     is held until the exchange is through, so work the caller enqueues behind the collective waits for it and work on other streams
//     does not;
//   * operations between ncclGroupStart and ncclGroupEnd form one batch per (communicator, stream): inside the host function they are
//     carried out in RANDOM order with RANDOM delays (0 .. KLT_STUB_RCCL_MAX_DELAY_US, default 1500 us; every send before any receive, so
//     that no order of a group's operations can deadlock -- a group's operations progress together in the real library too);
//   * a receive may be posted long before its send exists: it polls; a send never waits for its receiver;
//   * communicators share nothing: progress on one says nothing about another.
// What it is NOT: a communication library -- no performance, one node, trusted peers.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <dirent.h>
#include <sys/stat.h>
#include <unistd.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <random>
#include <string>
#include <thread>
#include <vector>

struct ncclComm {
    std::string dir;
    int nranks = 1, rank = 0;
    std::vector<unsigned long long> send_seq, recv_seq;
    unsigned long long coll_seq = 0;
    std::atomic<bool> failed{false};
};

namespace {

const double kTimeoutS = 120.0;

std::string base_dir()
{
    const char *d = getenv("KLT_STUB_RCCL_DIR");
    return d && *d ? d : "/tmp/klt_stub_rccl";
}

bool write_file(const std::string &path, const void *data, size_t n)
{
    const std::string tmp = path + ".tmp" + std::to_string((long)getpid());
    FILE *f = fopen(tmp.c_str(), "wb");
    if (!f) return false;
    const bool ok = n == 0 || fwrite(data, 1, n, f) == n;
    fclose(f);
    return ok && rename(tmp.c_str(), path.c_str()) == 0;      // readers see nothing or everything
}

bool read_file_when_there(const std::string &path, void *data, size_t n)
{
    const auto t0 = std::chrono::steady_clock::now();
    for (;;) {
        struct stat st;
        if (stat(path.c_str(), &st) == 0 && (size_t)st.st_size == n) {
            FILE *f = fopen(path.c_str(), "rb");
            if (f) {
                const bool ok = n == 0 || fread(data, 1, n, f) == n;
                fclose(f);
                if (ok) return true;
            }
        }
        if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > kTimeoutS) return false;
        std::this_thread::sleep_for(std::chrono::microseconds(200));
    }
}

size_t type_size(ncclDataType_t t)
{
    switch (t) {
    case ncclInt8: case ncclUint8: return 1;
    case ncclFloat16: case ncclBfloat16: return 2;
    case ncclInt32: case ncclUint32: case ncclFloat32: return 4;
    case ncclInt64: case ncclUint64: case ncclFloat64: return 8;
    default: return 0;
    }
}

// ---- one exchange = the operations of a group (or a single call) on one communicator and stream
enum Kind { SEND, RECV, PUBLISH /* own part of a collective */, COLLECT /* a peer's part of a collective */, REDUCE /* after the COLLECTs: combine */ };

struct Op {
    Kind kind;
    std::string path, unlink_first;
    const void *dev_src = nullptr;
    void *dev_dst = nullptr;
    size_t bytes = 0;
    char *host = nullptr;          // pinned staging
    bool remove_after = false;
    int red_op = 0, red_count = 0;
};

struct Batch {
    ncclComm *comm = nullptr;
    std::vector<Op> ops;
    unsigned long long seed = 0;
};

std::mutex g_pool_mutex;
std::vector<std::pair<char *, size_t>> g_pool;      // pinned staging buffers not in use

char *staging(size_t bytes)
{
    bytes = bytes ? bytes : 1;
    {
        std::lock_guard<std::mutex> lock(g_pool_mutex);
        for (size_t i = 0; i < g_pool.size(); i++)
            if (g_pool[i].second >= bytes && g_pool[i].second <= 4 * bytes + 4096) {
                char *p = g_pool[i].first;
                g_pool.erase(g_pool.begin() + (long)i);
                return p;
            }
    }
    // the size travels in front of the buffer (returned to the pool by give_back)
    char *p = nullptr;
    if (hipHostMalloc((void **)&p, bytes + 64, hipHostMallocDefault) != hipSuccess) return nullptr;
    *reinterpret_cast<size_t *>(p) = bytes;
    return p + 64;
}

void give_back(char *p)
{
    if (!p) return;
    std::lock_guard<std::mutex> lock(g_pool_mutex);
    g_pool.emplace_back(p, *reinterpret_cast<size_t *>(p - 64));
}

unsigned max_delay_us()
{
    static const unsigned v = [] { const char *e = getenv("KLT_STUB_RCCL_MAX_DELAY_US"); return e ? (unsigned)atoi(e) : 1500u; }();
    return v;
}

std::atomic<unsigned long long> g_batches{0};

// runs on a thread of the HIP runtime, in stream order: no HIP call in here
void exchange(void *arg)
{
    Batch *b = static_cast<Batch *>(arg);
    std::mt19937_64 rng(b->seed);
    auto nap = [&] {
        const unsigned m = max_delay_us();
        if (m) std::this_thread::sleep_for(std::chrono::microseconds(rng() % (m + 1)));
    };
    std::vector<size_t> first, second;
    for (size_t i = 0; i < b->ops.size(); i++) {
        const Kind k = b->ops[i].kind;
        if (k == SEND || k == PUBLISH) first.push_back(i);
        else if (k == RECV || k == COLLECT) second.push_back(i);
    }
    std::shuffle(first.begin(), first.end(), rng);
    std::shuffle(second.begin(), second.end(), rng);
    bool ok = !b->comm->failed.load();
    for (size_t i : first) {
        Op &o = b->ops[i];
        nap();
        if (!o.unlink_first.empty()) unlink(o.unlink_first.c_str());
        ok = ok && write_file(o.path, o.host, o.bytes);
    }
    for (size_t i : second) {
        Op &o = b->ops[i];
        nap();
        ok = ok && read_file_when_there(o.path, o.host, o.bytes);
        if (ok && o.remove_after) unlink(o.path.c_str());
    }
    for (Op &o : b->ops) {
        if (o.kind != REDUCE || !ok) continue;
        // o.host: the result; the parts are the COLLECT staging buffers of this batch, in rank order
        double *res = reinterpret_cast<double *>(o.host);
        bool have = false;
        for (const Op &p : b->ops) {
            if (p.kind != COLLECT) continue;
            const double *v = reinterpret_cast<const double *>(p.host);
            for (int i = 0; i < o.red_count; i++)
                res[i] = !have ? v[i] : (o.red_op == (int)ncclMax ? (v[i] > res[i] ? v[i] : res[i]) : res[i] + v[i]);
            have = true;
        }
    }
    if (!ok) b->comm->failed.store(true);
}

void retire(void *arg)
{
    Batch *b = static_cast<Batch *>(arg);
    for (Op &o : b->ops) give_back(o.host);
    delete b;
}

// enqueue one batch on `s`: copies out, the exchange, copies in, clean-up
ncclResult_t submit(Batch *b, hipStream_t s)
{
    b->seed = 0x9e3779b97f4a7c15ull * (g_batches.fetch_add(1) + 1) + (unsigned long long)getpid();
    if (const char *e = getenv("KLT_STUB_RCCL_SEED")) b->seed ^= strtoull(e, nullptr, 10) * 0xbf58476d1ce4e5b9ull;
    for (Op &o : b->ops) {
        o.host = staging(o.bytes);
        if (!o.host) return ncclSystemError;
        if ((o.kind == SEND || o.kind == PUBLISH) && o.bytes &&
            hipMemcpyAsync(o.host, o.dev_src, o.bytes, hipMemcpyDeviceToHost, s) != hipSuccess) return ncclUnhandledCudaError;
    }
    if (hipLaunchHostFunc(s, exchange, b) != hipSuccess) return ncclUnhandledCudaError;
    for (Op &o : b->ops)
        if (o.dev_dst && o.bytes && (o.kind == RECV || o.kind == COLLECT || o.kind == REDUCE) &&
            hipMemcpyAsync(o.dev_dst, o.host, o.bytes, hipMemcpyHostToDevice, s) != hipSuccess) return ncclUnhandledCudaError;
    if (hipLaunchHostFunc(s, retire, b) != hipSuccess) return ncclUnhandledCudaError;
    return ncclSuccess;
}

// ---- groups: per calling thread
thread_local int t_group_depth = 0;
struct Pending { ncclComm *comm; hipStream_t stream; Batch *batch; };
thread_local std::vector<Pending> t_pending;

ncclResult_t add_op(ncclComm *c, hipStream_t s, Op &&op)
{
    if (c->failed.load()) return ncclSystemError;
    if (t_group_depth == 0) {
        Batch *b = new Batch();
        b->comm = c;
        b->ops.push_back(std::move(op));
        return submit(b, s);
    }
    for (Pending &p : t_pending)
        if (p.comm == c && p.stream == s) { p.batch->ops.push_back(std::move(op)); return ncclSuccess; }
    Batch *b = new Batch();
    b->comm = c;
    b->ops.push_back(std::move(op));
    t_pending.push_back({c, s, b});
    return ncclSuccess;
}

}  // namespace

extern "C" {

ncclResult_t ncclGetUniqueId(ncclUniqueId *id)
{
    if (!id) return ncclInvalidArgument;
    std::memset(id, 0, sizeof(*id));
    FILE *f = fopen("/dev/urandom", "rb");
    unsigned char r[12] = {0};
    if (f) { (void)!fread(r, 1, sizeof(r), f); fclose(f); }
    char *p = id->internal;
    for (unsigned char c : r) p += std::snprintf(p, 3, "%02x", c);
    return ncclSuccess;
}

ncclResult_t ncclCommInitRank(ncclComm_t *out, int nranks, ncclUniqueId id, int rank)
{
    if (!out || nranks < 1 || rank < 0 || rank >= nranks) return ncclInvalidArgument;
    ncclComm *c = new ncclComm();
    c->nranks = nranks; c->rank = rank;
    c->send_seq.assign((size_t)nranks, 0); c->recv_seq.assign((size_t)nranks, 0);
    mkdir(base_dir().c_str(), 0777);
    c->dir = base_dir() + "/" + std::string(id.internal, strnlen(id.internal, 32));
    mkdir(c->dir.c_str(), 0777);
    char one = 1;
    if (!write_file(c->dir + "/joined_" + std::to_string(rank), &one, 1)) { delete c; return ncclSystemError; }
    for (int r = 0; r < nranks; r++)
        if (!read_file_when_there(c->dir + "/joined_" + std::to_string(r), &one, 1)) { delete c; return ncclSystemError; }
    *out = c;
    return ncclSuccess;
}

// (the caller has synchronised the stream its collectives ran on -- comm.hip does -- so no host function still holds the communicator)
ncclResult_t ncclCommDestroy(ncclComm_t c)
{
    delete c;
    return ncclSuccess;
}

// a communicator whose exchange can never complete: its host functions may still be polling -- leaked, as the caller is about to exit
ncclResult_t ncclCommAbort(ncclComm_t c)
{
    if (c) c->failed.store(true);
    return ncclSuccess;
}

const char *ncclGetErrorString(ncclResult_t r)
{
    switch (r) {
    case ncclSuccess: return "no error";
    case ncclUnhandledCudaError: return "stub rccl: HIP error";
    case ncclSystemError: return "stub rccl: file exchange failed or timed out";
    case ncclInvalidArgument: return "stub rccl: invalid argument";
    default: return "stub rccl: error";
    }
}

ncclResult_t ncclGroupStart()
{
    t_group_depth++;
    return ncclSuccess;
}

ncclResult_t ncclGroupEnd()
{
    if (t_group_depth <= 0) return ncclInvalidUsage;
    if (--t_group_depth > 0) return ncclSuccess;
    ncclResult_t res = ncclSuccess;
    for (Pending &p : t_pending) {
        const ncclResult_t r = submit(p.batch, p.stream);
        if (r != ncclSuccess && res == ncclSuccess) res = r;
    }
    t_pending.clear();
    return res;
}

ncclResult_t ncclSend(const void *buf, size_t count, ncclDataType_t t, int peer, ncclComm_t c, hipStream_t s)
{
    if (!c || peer < 0 || peer >= c->nranks || !type_size(t)) return ncclInvalidArgument;
    Op o;
    o.kind = SEND;
    o.dev_src = buf;
    o.bytes = count * type_size(t);
    o.path = c->dir + "/msg_" + std::to_string(c->rank) + "_" + std::to_string(peer) + "_" + std::to_string(c->send_seq[peer]++);
    return add_op(c, s, std::move(o));
}

ncclResult_t ncclRecv(void *buf, size_t count, ncclDataType_t t, int peer, ncclComm_t c, hipStream_t s)
{
    if (!c || peer < 0 || peer >= c->nranks || !type_size(t)) return ncclInvalidArgument;
    Op o;
    o.kind = RECV;
    o.dev_dst = buf;
    o.bytes = count * type_size(t);
    o.remove_after = true;
    o.path = c->dir + "/msg_" + std::to_string(peer) + "_" + std::to_string(c->rank) + "_" + std::to_string(c->recv_seq[peer]++);
    return add_op(c, s, std::move(o));
}

// every rank publishes its part as coll_<seq>_<rank>; a rank removes its own file of operation seq - 2 when it carries out operation seq
// (everybody has read it by then: to publish seq - 1 a rank must have finished seq - 2)
static void add_collective_ops(ncclComm_t c, Batch *b, const void *send, size_t bytes, char *recv_base /* null: staging only */)
{
    const unsigned long long seq = c->coll_seq++;
    Op pub;
    pub.kind = PUBLISH;
    pub.dev_src = send;
    pub.bytes = bytes;
    pub.path = c->dir + "/coll_" + std::to_string(seq) + "_" + std::to_string(c->rank);
    if (seq >= 2) pub.unlink_first = c->dir + "/coll_" + std::to_string(seq - 2) + "_" + std::to_string(c->rank);
    b->ops.push_back(std::move(pub));
    for (int r = 0; r < c->nranks; r++) {
        Op col;
        col.kind = COLLECT;
        col.bytes = bytes;
        col.dev_dst = recv_base ? recv_base + (size_t)r * bytes : nullptr;
        col.path = c->dir + "/coll_" + std::to_string(seq) + "_" + std::to_string(r);
        b->ops.push_back(std::move(col));
    }
}

ncclResult_t ncclAllGather(const void *send, void *recv, size_t count, ncclDataType_t t, ncclComm_t c, hipStream_t s)
{
    if (!c || !type_size(t)) return ncclInvalidArgument;
    if (c->failed.load()) return ncclSystemError;
    Batch *b = new Batch();
    b->comm = c;
    add_collective_ops(c, b, send, count * type_size(t), (char *)recv);
    return submit(b, s);
}

ncclResult_t ncclAllReduce(const void *send, void *recv, size_t count, ncclDataType_t t, ncclRedOp_t op, ncclComm_t c, hipStream_t s)
{
    if (!c || t != ncclFloat64 || (op != ncclMax && op != ncclSum)) return ncclInvalidArgument;       // what comm.hip uses
    if (c->failed.load()) return ncclSystemError;
    Batch *b = new Batch();
    b->comm = c;
    add_collective_ops(c, b, send, count * sizeof(double), nullptr);
    Op red;
    red.kind = REDUCE;
    red.bytes = count * sizeof(double);
    red.dev_dst = recv;
    red.red_op = (int)op;
    red.red_count = (int)count;
    b->ops.push_back(std::move(red));
    return submit(b, s);
}

}  // extern "C"
