// api_compat.hip -- the reference's literal native boundary (setup.py:8-9: ScanImageForGoodFeatures, extractImagePatchSlow,
// trackFeatureIterateCKLT) as synchronous one-shot calls, and the HIP-graph experiment (compiled only with -DKLT_GRAPH_PROBE).
#include "klt_context.h"


extern "C" {

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
    DEVALLOC(c, d, (5 * N + (size_t)ncand) * sizeof(float));
    if (int rc_keys = dev_alloc(c, (void **)&keys, (size_t)npow2 * sizeof(unsigned long long), "candidate keys")) { hipFree(d); return rc_keys; }
    hipError_t e = hipSuccess;
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
    DEVALLOC(c, d, (N + n + 1) * sizeof(float));
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
    DEVALLOC(c, d, (3 * N + 3 * n + 4) * sizeof(float));
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


}  // extern "C"
