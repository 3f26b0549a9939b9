// api_comm.hip -- the multi-GPU entry points of the C ABI: communicator lifetime, gathers / all-gathers / the baton of feature buffers,
// fences and waits.  RCCL itself (dlopen, the side stream, the process-wide chain of collectives) is comm.hip.
#include "klt_context.h"

extern "C" {

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


}  // extern "C"
