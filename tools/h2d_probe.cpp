// What the host link of the GPU box sustains for frame ingest: pinned host-to-device copies of one 1920x1080 (2.07 MB) and one 3840x2160
// (8.29 MB) u8 frame, on one and on two copy streams, with and without a compute kernel running next to them -- the ceiling
// bench.py's `pcie_pipelined` and `sequence_from_host` figures are held against -- plus what a pageable source costs and the
// device-to-host rate of a record table.  One JSON line on stdout.
// Build: hipcc -O3 --offload-arch=gfx950 tools/h2d_probe.cpp -o tools/h2d_probe
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>

static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

__global__ void busy(float *p, int iters)
{
    float v = p[threadIdx.x];
    for (int i = 0; i < iters; i++) v = v * 1.0001f + 0.5f;
    p[threadIdx.x + blockIdx.x * blockDim.x] = v;
}

int main()
{
    const size_t sizes[2] = {1920ul * 1080ul, 3840ul * 2160ul};
    const char *names[2] = {"1080p_2.07MB", "4k_8.29MB"};
    const size_t NMAX = sizes[1];
    const int NBUF = 4;
    unsigned char *d[NBUF], *pin[NBUF], *page = (unsigned char *)malloc(NMAX);
    for (int i = 0; i < NBUF; i++) {
        if (hipMalloc((void **)&d[i], NMAX) != hipSuccess || hipHostMalloc((void **)&pin[i], NMAX, hipHostMallocDefault) != hipSuccess) { printf("{\"error\": \"alloc\"}\n"); return 1; }
        memset(pin[i], i + 1, NMAX);
    }
    memset(page, 7, NMAX);
    hipStream_t s[4], k;
    for (auto &x : s) hipStreamCreateWithFlags(&x, hipStreamNonBlocking);
    hipStreamCreateWithFlags(&k, hipStreamNonBlocking);
    float *work;
    hipMalloc((void **)&work, 256 * 1024 * sizeof(float));
    hipEvent_t ev[64];
    for (auto &e : ev) hipEventCreateWithFlags(&e, hipEventDisableTiming);
    std::string out = "{";
    char buf[256];
    for (int si = 0; si < 2; si++) {
        const size_t N = sizes[si];
        const int reps = si == 0 ? 400 : 150;
        double last_enqueue_us = 0;
        auto run = [&](int nstreams, bool with_kernel, bool pageable, bool with_events = false) {
            double best = 1e9;
            for (int rep = 0; rep < 4; rep++) {
                hipDeviceSynchronize();
                if (with_kernel) hipLaunchKernelGGL(busy, dim3(1024), dim3(256), 0, k, work, 20000000);   // keeps the CUs busy for the whole pass
                const double t = now();
                for (int i = 0; i < reps; i++)
                {
                    hipMemcpyAsync(d[i % NBUF], pageable ? page : pin[i % NBUF], N, hipMemcpyHostToDevice, s[i % nstreams]);
                    if (with_events) hipEventRecord(ev[i % 64], s[i % nstreams]);      // what klt_upload_u8_async does behind every copy
                }
                const double enq = (now() - t) / reps;
                for (auto &x : s) hipStreamSynchronize(x);
                const double dt = (now() - t) / reps;
                last_enqueue_us = enq * 1e6;
                if (dt < best) best = dt;
                hipDeviceSynchronize();
            }
            return best;
        };
        const double one = run(1, false, false), two = run(2, false, false), onek = run(1, true, false), twok = run(2, true, false), pg = run(1, false, true);
        snprintf(buf, sizeof(buf), "%s\"h2d_%s\": {\"one_stream_GBps\": %.2f, \"two_streams_GBps\": %.2f, \"one_stream_next_to_a_kernel_GBps\": %.2f, "
                 "\"two_streams_next_to_a_kernel_GBps\": %.2f, \"pageable_GBps\": %.2f, \"us_per_frame_one_stream\": %.1f}",
                 si ? ", " : "", names[si], N / one / 1e9, N / two / 1e9, N / onek / 1e9, N / twok / 1e9, N / pg / 1e9, one * 1e6);
        out += buf;
        // more streams, and an event recorded behind every copy
        std::string more = std::string(", \"h2d_") + names[si] + "_streams\": {";
        for (int ns = 1; ns <= 4; ns++) {
            const double plain = run(ns, false, false, false); const double e1 = last_enqueue_us;
            const double evd = run(ns, false, false, true); const double e2 = last_enqueue_us;
            snprintf(buf, sizeof(buf), "%s\"%d\": {\"GBps\": %.2f, \"host_us_per_copy\": %.1f, \"with_an_event_per_copy_GBps\": %.2f, \"with_an_event_host_us\": %.1f}",
                     ns > 1 ? ", " : "", ns, N / plain / 1e9, e1, N / evd / 1e9, e2);
            more += buf;
        }
        out += more + "}";
    }
    // device-to-host: 16 rows of 5000 records (1.28 MB) and 16 rows of 20000 (5.12 MB)
    for (size_t bytes : {16ul * 5000 * 16, 16ul * 20000 * 16}) {
        double best = 1e9;
        for (int rep = 0; rep < 4; rep++) {
            hipDeviceSynchronize();
            const double t = now();
            for (int i = 0; i < 100; i++) hipMemcpyAsync(pin[i % NBUF], d[i % NBUF], bytes, hipMemcpyDeviceToHost, s[0]);
            hipStreamSynchronize(s[0]);
            const double dt = (now() - t) / 100;
            if (dt < best) best = dt;
        }
        snprintf(buf, sizeof(buf), ", \"d2h_%zuKB\": {\"GBps\": %.2f, \"us\": %.1f}", bytes / 1000, bytes / best / 1e9, best * 1e6);
        out += buf;
    }
    out += "}";
    printf("%s\n", out.c_str());
    return 0;
}
