// Cost of a device-wide barrier between the workgroups of one (co-resident) launch: every workgroup adds 1 to a counter and spins until
// it reaches a multiple of the grid size (agent-scope atomics); K barriers per launch, against K + 1 dependent empty launches.
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 tools/mb/grid_barrier.hip -o tools/mb/grid_barrier
#include <hip/hip_runtime.h>
#include <cstdio>

__device__ __forceinline__ void grid_barrier(unsigned *counter, unsigned &epoch)
{
    __syncthreads();
    if (threadIdx.x == 0) {
        epoch += gridDim.x;
        __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        while (__hip_atomic_load(counter, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < epoch) __builtin_amdgcn_s_sleep(1);
    }
    __syncthreads();
}

__global__ void barriers(unsigned *counter, unsigned base, int k, float *sink)
{
    unsigned epoch = base;
    float v = threadIdx.x;
    for (int i = 0; i < k; i++) {
        v = v * 1.0001f + 1.f;
        grid_barrier(counter, epoch);
    }
    if (v == 12345.f) *sink = v;
}

__global__ void empty(float *sink) { if (threadIdx.x == 12345) *sink = 1.f; }

int main()
{
    unsigned *counter; float *sink;
    hipMalloc(&counter, 4); hipMalloc(&sink, 4); hipMemset(counter, 0, 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int K = 64;
    for (int threads : {256, 1024})
        for (int g : {32, 64, 128, 256, 512}) {
            unsigned base = 0;
            hipMemset(counter, 0, 4);
            hipLaunchKernelGGL(barriers, dim3(g), dim3(threads), 0, 0, counter, base, K, sink); base += (unsigned)g * K;
            hipDeviceSynchronize();
            hipEventRecord(e0);
            for (int r = 0; r < 10; r++) { hipLaunchKernelGGL(barriers, dim3(g), dim3(threads), 0, 0, counter, base, K, sink); base += (unsigned)g * K; }
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            printf("%4d workgroups x %4d threads: %.2f us per launch of %d barriers -> %.2f us per barrier\n", g, threads, ms * 100, K, ms * 100 / K);
        }
    hipEventRecord(e0);
    for (int r = 0; r < 640; r++) hipLaunchKernelGGL(empty, dim3(64), dim3(256), 0, 0, sink);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("dependent empty launches: %.2f us each\n", ms * 1000 / 640);
    return 0;
}
