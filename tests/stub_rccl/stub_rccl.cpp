// TEST INFRASTRUCTURE ONLY -- a stand-in for librccl that lets several ranks of libkltgpu.so's multi-GPU path run on ONE GPU.
//
// RCCL refuses two ranks of a communicator on the same device, and the GPU boxes this repository is developed on have one GPU,
// so the N > 1 code of csrc/comm.hip, klt_api.hip and bench.py (per-rank counts of the gather, the feature list as a baton between
// ranks, all-gather of the record tables, barrier / max over ranks) would only ever meet a single rank.  This library exports the ten
// RCCL entry points comm.hip binds, with the same signatures (rccl.h), and moves the bytes through files in a directory all ranks
// share (KLT_STUB_RCCL_DIR): every operation first waits for the stream it was given (so the producer's work is complete), copies device
// -> host -> file, and the receiving side polls for the file and copies host -> device, synchronously.  Point KLT_RCCL_LIB at the built
// library and give every rank the same device (KLT_RANKS_SHARE_DEVICE=0): tests/test_gpu_multirank.py.
//
// It is NOT a communication library: no overlap, no performance, one node, trusted peers.  It exists so that rank arithmetic, buffer
// offsets, ordering and teardown of the real code are exercised by more than one process.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <dirent.h>
#include <sys/stat.h>
#include <unistd.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

struct ncclComm {
    std::string dir;
    int nranks = 1, rank = 0;
    std::vector<unsigned long long> send_seq, recv_seq;
    unsigned long long coll_seq = 0;
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

ncclResult_t to_host(const void *dev, size_t bytes, hipStream_t s, std::vector<char> &h)
{
    h.resize(bytes ? bytes : 1);
    if (hipStreamSynchronize(s) != hipSuccess) return ncclUnhandledCudaError;          // the producer's work is complete
    if (bytes && hipMemcpy(h.data(), dev, bytes, hipMemcpyDeviceToHost) != hipSuccess) return ncclUnhandledCudaError;
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

ncclResult_t ncclCommDestroy(ncclComm_t c)
{
    delete c;
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

ncclResult_t ncclGroupStart() { return ncclSuccess; }     // every operation completes before it returns: a group is its operations in order
ncclResult_t ncclGroupEnd() { return ncclSuccess; }

ncclResult_t ncclSend(const void *buf, size_t count, ncclDataType_t t, int peer, ncclComm_t c, hipStream_t s)
{
    if (!c || peer < 0 || peer >= c->nranks || !type_size(t)) return ncclInvalidArgument;
    std::vector<char> h;
    if (ncclResult_t r = to_host(buf, count * type_size(t), s, h)) return r;
    const std::string path = c->dir + "/msg_" + std::to_string(c->rank) + "_" + std::to_string(peer) + "_" + std::to_string(c->send_seq[peer]++);
    return write_file(path, h.data(), count * type_size(t)) ? ncclSuccess : ncclSystemError;      // never blocks: no send / receive order can deadlock
}

ncclResult_t ncclRecv(void *buf, size_t count, ncclDataType_t t, int peer, ncclComm_t c, hipStream_t s)
{
    if (!c || peer < 0 || peer >= c->nranks || !type_size(t)) return ncclInvalidArgument;
    const size_t bytes = count * type_size(t);
    std::vector<char> h(bytes ? bytes : 1);
    const std::string path = c->dir + "/msg_" + std::to_string(peer) + "_" + std::to_string(c->rank) + "_" + std::to_string(c->recv_seq[peer]++);
    if (!read_file_when_there(path, h.data(), bytes)) return ncclSystemError;
    unlink(path.c_str());
    if (hipStreamSynchronize(s) != hipSuccess) return ncclUnhandledCudaError;
    if (bytes && hipMemcpy(buf, h.data(), bytes, hipMemcpyHostToDevice) != hipSuccess) return ncclUnhandledCudaError;
    return ncclSuccess;
}

// every rank publishes its part as coll_<seq>_<rank>; a rank removes its own file of operation seq - 2 when it starts operation seq
// (everybody has read it by then: to publish seq - 1 a rank must have finished seq - 2)
static ncclResult_t publish_and_collect(ncclComm_t c, const std::vector<char> &mine, size_t bytes, std::vector<std::vector<char>> &all)
{
    const unsigned long long seq = c->coll_seq++;
    if (seq >= 2) unlink((c->dir + "/coll_" + std::to_string(seq - 2) + "_" + std::to_string(c->rank)).c_str());
    if (!write_file(c->dir + "/coll_" + std::to_string(seq) + "_" + std::to_string(c->rank), mine.data(), bytes)) return ncclSystemError;
    all.assign((size_t)c->nranks, std::vector<char>(bytes ? bytes : 1));
    for (int r = 0; r < c->nranks; r++)
        if (!read_file_when_there(c->dir + "/coll_" + std::to_string(seq) + "_" + std::to_string(r), all[r].data(), bytes)) return ncclSystemError;
    return ncclSuccess;
}

ncclResult_t ncclAllGather(const void *send, void *recv, size_t count, ncclDataType_t t, ncclComm_t c, hipStream_t s)
{
    if (!c || !type_size(t)) return ncclInvalidArgument;
    const size_t bytes = count * type_size(t);
    std::vector<char> mine;
    if (ncclResult_t r = to_host(send, bytes, s, mine)) return r;
    std::vector<std::vector<char>> all;
    if (ncclResult_t r = publish_and_collect(c, mine, bytes, all)) return r;
    for (int r = 0; r < c->nranks; r++)
        if (bytes && hipMemcpy((char *)recv + (size_t)r * bytes, all[r].data(), bytes, hipMemcpyHostToDevice) != hipSuccess) return ncclUnhandledCudaError;
    return ncclSuccess;
}

ncclResult_t ncclAllReduce(const void *send, void *recv, size_t count, ncclDataType_t t, ncclRedOp_t op, ncclComm_t c, hipStream_t s)
{
    if (!c || t != ncclFloat64 || (op != ncclMax && op != ncclSum)) return ncclInvalidArgument;       // what comm.hip uses
    const size_t bytes = count * sizeof(double);
    std::vector<char> mine;
    if (ncclResult_t r = to_host(send, bytes, s, mine)) return r;
    std::vector<std::vector<char>> all;
    if (ncclResult_t r = publish_and_collect(c, mine, bytes, all)) return r;
    std::vector<double> res(count ? count : 1);
    for (size_t i = 0; i < count; i++) {
        double v = reinterpret_cast<const double *>(all[0].data())[i];
        for (int r = 1; r < c->nranks; r++) {
            const double w = reinterpret_cast<const double *>(all[r].data())[i];
            v = op == ncclMax ? (w > v ? w : v) : v + w;
        }
        res[i] = v;
    }
    if (bytes && hipMemcpy(recv, res.data(), bytes, hipMemcpyHostToDevice) != hipSuccess) return ncclUnhandledCudaError;
    return ncclSuccess;
}

}  // extern "C"
