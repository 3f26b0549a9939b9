// What a captured HIP graph buys a chain of ~25 small dependent launches on gfx950 / ROCm 7.2, and which capture features work:
//   (1) host time of one hipGraphLaunch against the 25 hipLaunchKernelGGL calls it replaces, and the GPU time of the chain either way;
//   (2) fork / join over two streams inside a capture (hipEventRecord + hipStreamWaitEvent), hipMemsetAsync and a device-to-device
//       hipMemcpyAsync as captured nodes;
//   (3) a kernel in the middle of a graph publishing a word to pinned host memory (system-scope store) that the host polls while the rest
//       of the graph is still running -- the host's "look" at a selection's outcome without an event;
//   (4) kernels that index through a device-side counter they advance themselves, so that one graph serves every frame;
//   (5) the same eager frame with the side stream's launches issued by a second host thread (two enqueue threads, one per stream).
// Build: hipcc -O3 --offload-arch=gfx950 tools/mb/graph_probe.hip -o tools/mb/graph_probe
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstring>
#include <vector>
#include <atomic>
#include <thread>

#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s (line %d)\n", #x, hipGetErrorString(e_), __LINE__); return 1; } } while (0)

__global__ void spin(long long ticks, unsigned *acc)
{
    const long long t0 = wall_clock64();
    while (wall_clock64() - t0 < ticks) { }
    if (threadIdx.x == 0 && blockIdx.x == 0) atomicAdd(acc, 1u);
}

// publishes *counter (after advancing it) to pinned host memory: what the host polls
__global__ void publish(unsigned *counter, volatile unsigned *host_word, unsigned *acc)
{
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        const unsigned v = atomicAdd(counter, 1u) + 1u;
        host_word[1] = *acc;                        // payload first
        __threadfence_system();
        host_word[0] = v;                           // then the sequence number
    }
}

// appends `n` words of src to table row *row, and advances the row
__global__ void append_row(const unsigned *src, unsigned *table, unsigned *row, int n)
{
    const unsigned r = *row;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) table[(size_t)r * n + i] = src[i] + r;
    __syncthreads();
    if (blockIdx.x == 0 && threadIdx.x == 0) { __threadfence(); atomicAdd(row, 1u); }   // single block in this probe
}

static double now_us() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main()
{
    hipStream_t s0, s1;
    CHK(hipStreamCreateWithFlags(&s0, hipStreamNonBlocking));
    CHK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking));
    unsigned *acc, *counter, *row, *src, *table, *scratch;
    CHK(hipMalloc(&acc, 4)); CHK(hipMalloc(&counter, 4)); CHK(hipMalloc(&row, 4));
    const int N = 4096, ROWS = 2000;
    CHK(hipMalloc(&src, N * 4)); CHK(hipMalloc(&table, (size_t)N * ROWS * 4)); CHK(hipMalloc(&scratch, 1 << 20));
    CHK(hipMemset(acc, 0, 4)); CHK(hipMemset(counter, 0, 4)); CHK(hipMemset(row, 0, 4)); CHK(hipMemset(src, 0, N * 4));
    unsigned *host_word;
    CHK(hipHostMalloc((void **)&host_word, 64, hipHostMallocDefault));
    host_word[0] = host_word[1] = 0;
    hipEvent_t fork, join;
    CHK(hipEventCreateWithFlags(&fork, hipEventDisableTiming));
    CHK(hipEventCreateWithFlags(&join, hipEventDisableTiming));
    const long long T3 = 300;                      // 3 us at 100 MHz wall clock

    // the frame's launch set: main: 12 x spin(3 us) -> publish -> [join] -> 2 x spin; side: memset, 8 x spin(6 us), D2D copy; append_row at the end
    auto enqueue_frame = [&](bool capturing) -> int {
        CHK(hipEventRecord(fork, s0));
        CHK(hipStreamWaitEvent(s1, fork, 0));
        CHK(hipMemsetAsync(scratch, 0, 1 << 20, s1));
        for (int i = 0; i < 8; i++) hipLaunchKernelGGL(spin, dim3(64), dim3(256), 0, s1, 2 * T3, acc);
        CHK(hipMemcpyAsync(scratch + 1024, scratch, 4096, hipMemcpyDeviceToDevice, s1));
        CHK(hipEventRecord(join, s1));
        for (int i = 0; i < 12; i++) hipLaunchKernelGGL(spin, dim3(64), dim3(256), 0, s0, T3, acc);
        hipLaunchKernelGGL(publish, dim3(1), dim3(64), 0, s0, counter, host_word, acc);
        CHK(hipStreamWaitEvent(s0, join, 0));
        for (int i = 0; i < 2; i++) hipLaunchKernelGGL(spin, dim3(64), dim3(256), 0, s0, 5 * T3, acc);
        hipLaunchKernelGGL(append_row, dim3(1), dim3(256), 0, s0, src, table, row, N);
        (void)capturing;
        return 0;
    };

    // ---- eager
    for (int w = 0; w < 3; w++) if (enqueue_frame(false)) return 1;
    CHK(hipStreamSynchronize(s0)); CHK(hipStreamSynchronize(s1));
    const int F = 200;
    double host_eager = 0;
    double t0 = now_us();
    for (int f = 0; f < F; f++) {
        const double a = now_us();
        if (enqueue_frame(false)) return 1;
        host_eager += now_us() - a;
        const unsigned want = 3 + f + 1;
        while (*(volatile unsigned *)host_word < want) { }          // the look: mid-frame
    }
    CHK(hipStreamSynchronize(s0)); CHK(hipStreamSynchronize(s1));
    const double eager_frame = (now_us() - t0) / F;
    printf("eager : %.1f us per frame wall, %.1f us of host enqueue per frame (26 launches, 2 copies, 4 event operations)\n", eager_frame, host_eager / F);

    // ---- captured once, replayed
    hipGraph_t graph; hipGraphExec_t exec;
    CHK(hipStreamBeginCapture(s0, hipStreamCaptureModeThreadLocal));
    if (enqueue_frame(true)) return 1;
    CHK(hipStreamEndCapture(s0, &graph));
    size_t nn = 0;
    CHK(hipGraphGetNodes(graph, nullptr, &nn));
    CHK(hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0));
    printf("graph : %zu nodes captured over two streams (fork / join, memset, device copy)\n", nn);
    unsigned before = 0;
    CHK(hipMemcpy(&before, counter, 4, hipMemcpyDeviceToHost));
    for (int w = 0; w < 3; w++) CHK(hipGraphLaunch(exec, s0));
    CHK(hipStreamSynchronize(s0));
    double host_graph = 0, look_lead = 0;
    t0 = now_us();
    for (int f = 0; f < F; f++) {
        const double a = now_us();
        CHK(hipGraphLaunch(exec, s0));
        host_graph += now_us() - a;
        const unsigned want = before + 3 + f + 1;
        while (*(volatile unsigned *)host_word < want) { }          // the look arrives while the tail of the graph (and the side branch) still runs
        const double seen = now_us();
        if (f == F - 1) { CHK(hipStreamSynchronize(s0)); look_lead = now_us() - seen; }
    }
    CHK(hipStreamSynchronize(s0));
    const double graph_frame = (now_us() - t0) / F;
    printf("graph : %.1f us per frame wall, %.1f us of host time per hipGraphLaunch; the last frame's look came %.1f us before its graph ended\n",
           graph_frame, host_graph / F, look_lead);
    unsigned rows = 0, payload = host_word[1], accv = 0;
    CHK(hipMemcpy(&rows, row, 4, hipMemcpyDeviceToHost));
    CHK(hipMemcpy(&accv, acc, 4, hipMemcpyDeviceToHost));
    std::vector<unsigned> t((size_t)N * rows);
    CHK(hipMemcpy(t.data(), table, t.size() * 4, hipMemcpyDeviceToHost));
    bool ok = rows == (unsigned)(3 + F + 3 + F);
    for (unsigned r = 0; r < rows && ok; r++) ok = t[(size_t)r * N] == r && t[(size_t)r * N + N - 1] == r;
    printf("check : %u rows appended through the device-side row counter (%s), %u kernels counted, last published payload %u\n", rows, ok ? "every row holds its index" : "WRONG", accv, payload);
    // the same graph instantiated FOUR times, launched in rotation (a sequence would rotate through one graph per ring phase): does the host
    // cost of a launch come from re-launching an executable graph whose previous launch is still running?
    {
        hipGraphExec_t ex[4];
        for (auto &e : ex) CHK(hipGraphInstantiate(&e, graph, nullptr, nullptr, 0));
        for (int w = 0; w < 8; w++) CHK(hipGraphLaunch(ex[w % 4], s0));
        CHK(hipStreamSynchronize(s0));
        unsigned base2 = 0;
        CHK(hipMemcpy(&base2, counter, 4, hipMemcpyDeviceToHost));
        double host_rot = 0;
        t0 = now_us();
        for (int f = 0; f < F; f++) {
            const double a = now_us();
            CHK(hipGraphLaunch(ex[f % 4], s0));
            host_rot += now_us() - a;
            while (*(volatile unsigned *)host_word < base2 + f + 1) { }
        }
        CHK(hipStreamSynchronize(s0));
        printf("graph : four executable graphs in rotation: %.1f us per frame wall (look per frame), %.1f us of host time per hipGraphLaunch\n", (now_us() - t0) / F, host_rot / F);
        t0 = now_us();
        host_rot = 0;
        for (int f = 0; f < F; f++) { const double a = now_us(); CHK(hipGraphLaunch(ex[f % 4], s0)); host_rot += now_us() - a; }
        CHK(hipStreamSynchronize(s0));
        printf("graph : four executable graphs in rotation, back to back: %.1f us per frame, %.1f us of host time per launch\n", (now_us() - t0) / F, host_rot / F);
    }
    // back-to-back graph launches without a look: the floor
    t0 = now_us();
    for (int f = 0; f < F; f++) CHK(hipGraphLaunch(exec, s0));
    CHK(hipStreamSynchronize(s0));
    printf("graph : %.1f us per frame when launched back to back (no look)\n", (now_us() - t0) / F);
    t0 = now_us();
    for (int f = 0; f < F; f++) if (enqueue_frame(false)) return 1;
    CHK(hipStreamSynchronize(s0)); CHK(hipStreamSynchronize(s1));
    printf("eager : %.1f us per frame when enqueued back to back (no look)\n", (now_us() - t0) / F);

    // ---- eager with two enqueue threads: the helper issues the side stream's work of frame f as soon as the main thread has recorded
    // the frame's fork event; the main thread waits for the helper's join event to be recorded before it makes its stream wait for it
    {
        std::vector<hipEvent_t> forks(F + 8), joins(F + 8);
        for (auto &e : forks) CHK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
        for (auto &e : joins) CHK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
        std::atomic<int> forked{0}, joined{0};
        std::atomic<bool> stop{false};
        double helper_busy = 0;
        std::thread helper([&] {
            hipSetDevice(0);
            for (int f = 0; f < F; f++) {
                while (forked.load(std::memory_order_acquire) <= f) { if (stop.load()) return; }
                const double a = now_us();
                hipStreamWaitEvent(s1, forks[f], 0);
                hipMemsetAsync(scratch, 0, 1 << 20, s1);
                for (int i = 0; i < 8; i++) hipLaunchKernelGGL(spin, dim3(64), dim3(256), 0, s1, 2 * T3, acc);
                hipMemcpyAsync(scratch + 1024, scratch, 4096, hipMemcpyDeviceToDevice, s1);
                hipEventRecord(joins[f], s1);
                helper_busy += now_us() - a;
                joined.store(f + 1, std::memory_order_release);
            }
        });
        unsigned base = 0;
        CHK(hipMemcpy(&base, counter, 4, hipMemcpyDeviceToHost));
        double main_busy = 0;
        t0 = now_us();
        for (int f = 0; f < F; f++) {
            const double a = now_us();
            CHK(hipEventRecord(forks[f], s0));
            forked.store(f + 1, std::memory_order_release);
            for (int i = 0; i < 12; i++) hipLaunchKernelGGL(spin, dim3(64), dim3(256), 0, s0, T3, acc);
            hipLaunchKernelGGL(publish, dim3(1), dim3(64), 0, s0, counter, host_word, acc);
            while (joined.load(std::memory_order_acquire) <= f) { }
            CHK(hipStreamWaitEvent(s0, joins[f], 0));
            for (int i = 0; i < 2; i++) hipLaunchKernelGGL(spin, dim3(64), dim3(256), 0, s0, 5 * T3, acc);
            hipLaunchKernelGGL(append_row, dim3(1), dim3(256), 0, s0, src, table, row, N);
            main_busy += now_us() - a;
            while (*(volatile unsigned *)host_word < base + f + 1) { }
        }
        CHK(hipStreamSynchronize(s0)); CHK(hipStreamSynchronize(s1));
        const double two = (now_us() - t0) / F;
        stop.store(true);
        helper.join();
        printf("eager, two enqueue threads: %.1f us per frame wall; main thread %.1f us, helper %.1f us of enqueue per frame\n", two, main_busy / F, helper_busy / F);
    }
    return 0;
}
