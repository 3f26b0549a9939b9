// Latency of a dependent v_add_f32 chain (what bounds the summed-area-table kernels): one wavefront per workgroup runs N dependent adds;
// ns per add for a few / many workgroups, alone and with a second independent chain in the same wavefront.
// Build: hipcc -O3 --offload-arch=gfx950 dep_add.hip -o dep_add
#include <hip/hip_runtime.h>
#include <cstdio>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
constexpr int N = 1 << 16;
template <int CHAINS>
__global__ __launch_bounds__(64) void chain(float *out, float y)
{
    float x[CHAINS];
    for (int c = 0; c < CHAINS; c++) x[c] = threadIdx.x + c;
    for (int i = 0; i < N / 16; i++) {
#pragma unroll
        for (int u = 0; u < 16; u++)
#pragma unroll
            for (int c = 0; c < CHAINS; c++) asm volatile("v_add_f32 %0, %0, %1" : "+v"(x[c]) : "v"(y));
    }
    float s = 0;
    for (int c = 0; c < CHAINS; c++) s += x[c];
    out[blockIdx.x * 64 + threadIdx.x] = s;
}
template <int CHAINS>
static int run(int wgs, float *out)
{
    hipEvent_t a, b;
    CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
    hipLaunchKernelGGL(chain<CHAINS>, dim3(wgs), dim3(64), 0, 0, out, 1.0f);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(a));
    for (int r = 0; r < 5; r++) hipLaunchKernelGGL(chain<CHAINS>, dim3(wgs), dim3(64), 0, 0, out, 1.0f);
    CHECK(hipEventRecord(b)); CHECK(hipEventSynchronize(b));
    float ms; CHECK(hipEventElapsedTime(&ms, a, b));
    printf("%5d workgroups x 1 wavefront, %d chain(s): %.2f ns per dependent add (%.1f clocks at 2.4 GHz)\n", wgs, CHAINS, ms / 5 * 1e6 / N, ms / 5 * 1e6 / N * 2.4);
    return 0;
}
int main()
{
    float *out; CHECK(hipMalloc(&out, 65536 * 64 * sizeof(float)));
    for (int wgs : {1, 68, 256, 1024, 4096}) { if (run<1>(wgs, out)) return 1; }
    for (int wgs : {68, 1024}) { if (run<2>(wgs, out)) return 1; if (run<4>(wgs, out)) return 1; }
    return 0;
}
