// How fast does the GPU start workgroups?  N single-wavefront workgroups (like the tracker: one feature each, 1 KB of LDS)
// against the same wavefronts packed four per workgroup.  Build: hipcc -O3 --offload-arch=gfx950 tools/mb/dispatch_rate.hip -o tools/mb/dispatch_rate
#include <hip/hip_runtime.h>
#include <cstdio>

__global__ void work(long long ticks, float *out)
{
    extern __shared__ float l[];
    l[threadIdx.x] = threadIdx.x;
    const long long t0 = wall_clock64();
    while (wall_clock64() - t0 < ticks) { }
    if (out && l[threadIdx.x] < 0) out[0] = 1;
}

static double run(int blocks, int threads, long long ticks, size_t lds)
{
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 3; i++) hipLaunchKernelGGL(work, dim3(blocks), dim3(threads), lds, 0, ticks, (float *)nullptr);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int i = 0; i < 50; i++) hipLaunchKernelGGL(work, dim3(blocks), dim3(threads), lds, 0, ticks, (float *)nullptr);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    return ms * 1000.0 / 50;
}

int main()
{
    for (long long ticks : {0LL, 300LL, 1000LL}) {
        printf("each wavefront busy for %.1f us:\n", ticks / 100.0);
        printf("   5000 workgroups x  64 threads, 1 KB LDS: %6.2f us per launch\n", run(5000, 64, ticks, 1024));
        printf("   5000 workgroups x  64 threads, no LDS  : %6.2f us per launch\n", run(5000, 64, ticks, 0));
        printf("   2500 workgroups x 128 threads, 2 KB LDS: %6.2f us per launch\n", run(2500, 128, ticks, 2048));
        printf("   1250 workgroups x 256 threads, 4 KB LDS: %6.2f us per launch\n", run(1250, 256, ticks, 4096));
        printf("  20000 workgroups x  64 threads, 1 KB LDS: %6.2f us per launch\n", run(20000, 64, ticks, 1024));
        printf("   5000 workgroups x 256 threads, 4 KB LDS: %6.2f us per launch\n", run(5000, 256, ticks, 4096));
    }
    return 0;
}
