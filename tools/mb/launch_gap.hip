// How long is the gap between dependent kernels of one stream, and does hipExtAnyOrderLaunch remove it on gfx950?
// Build: hipcc -O3 --offload-arch=gfx950 tools/mb/launch_gap.hip -o tools/mb/launch_gap
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <cstdio>

__global__ void spin(long long ticks, int *sink)
{
    const long long t0 = wall_clock64();
    while (wall_clock64() - t0 < ticks) { }
    if (sink && threadIdx.x == 0 && blockIdx.x == 0) *sink = 1;
}

static double run(int n, int blocks, long long ticks, unsigned flags_even, unsigned flags_odd, int *sink)
{
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int i = 0; i < n; i++)
        hipExtLaunchKernelGGL(spin, dim3(blocks), dim3(256), 0, 0, nullptr, nullptr, (i & 1) ? flags_odd : flags_even, ticks, sink);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    return ms * 1000.0 / n;
}

int main()
{
    int *sink; hipMalloc(&sink, 4);
    run(10, 1, 0, 0, 0, sink);
    printf("empty kernel, 1 block:          ordered %.2f us/launch   any-order %.2f us/launch\n", run(400, 1, 0, 0, 0, sink), run(400, 1, 0, 1, 1, sink));
    printf("empty kernel, 2048 blocks:      ordered %.2f us/launch   any-order %.2f us/launch\n", run(400, 2048, 0, 0, 0, sink), run(400, 2048, 0, 1, 1, sink));
    printf("10 us spin, 256 blocks:         ordered %.2f us/launch   any-order %.2f   alternate (odd launches any-order) %.2f\n",
           run(200, 256, 1000, 0, 0, sink), run(200, 256, 1000, 1, 1, sink), run(200, 256, 1000, 0, 1, sink));
    printf("10 us spin, 2048 blocks:        ordered %.2f us/launch   any-order %.2f   alternate %.2f\n",
           run(200, 2048, 1000, 0, 0, sink), run(200, 2048, 1000, 1, 1, sink), run(200, 2048, 1000, 0, 1, sink));
    return 0;
}
