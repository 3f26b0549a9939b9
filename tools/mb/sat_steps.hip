// Step timing of the SAT row pipeline: builds the product source with SAT_PIPE_DEBUG and prints, for workgroup 0, when each
// wavefront started a step, finished its work and passed the barrier (ticks of 10 ns).
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off -DSAT_PIPE_DEBUG -Iinclude -Ipyfeaturetrack_amd/csrc tools/mb/sat_steps.hip -o tools/mb/sat_steps
#include "../../pyfeaturetrack_amd/csrc/sat_pipeline.hip"
thread_local hipEvent_t g_klt_stamp_start = nullptr, g_klt_stamp_stop = nullptr;     // the timing hooks of klt_launch (api_context.hip in the library)
#include <cstdio>
#include <cstdlib>
#include <vector>
int main(int argc, char **argv)
{
    const int nc = argc > 2 ? atoi(argv[1]) : 1920, nr = argc > 2 ? atoi(argv[2]) : 1080;
    float *gx, *gy, *sat;
    hipMalloc(&gx, 8ull * nc * nr); gy = gx + 1;                    // the interleaved gradient planes (klt_internal.h)
    hipMalloc(&sat, 12ull * nc * nr);
    std::vector<float> h(2 * (size_t)nc * nr);
    for (size_t i = 0; i < h.size(); i++) h[i] = (float)((i * 2654435761u) >> 20) * 1e-3f;
    hipMemcpy(gx, h.data(), 4 * h.size(), hipMemcpyHostToDevice);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 3; i++) launch_sat_rows_pipe(0, gx, gy, sat, nc, nr);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int i = 0; i < 20; i++) launch_sat_rows_pipe(0, gx, gy, sat, nc, nr);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("sat_rows_pipe: %.2f us per launch\n", ms * 1000 / 20);
    static long long d[6 * 64 * 3];
    hipMemcpyFromSymbol(d, HIP_SYMBOL(g_sat_dbg), sizeof(d));
    const long long t0 = d[(0 * 64 + 0) * 3];
    printf("step | start | chain: work wait | loader 1 | loader 2 | loader 3 | storer 1 | storer 2   (ticks of 10 ns)\n");
    for (int s = 0; s < 20; s++) {
        printf("%3d  %6lld |", s, d[(0 * 64 + s) * 3] - t0);
        for (int w = 0; w < 6; w++) printf(" %4lld %4lld |", d[(w * 64 + s) * 3 + 1] - d[(w * 64 + s) * 3], d[(w * 64 + s) * 3 + 2] - d[(w * 64 + s) * 3 + 1]);
        printf("\n");
    }
    // the column pass (in place on the row pass's output): same table for workgroup (0, 0)
    static long long z[6 * 64 * 3];
    hipMemcpyToSymbol(HIP_SYMBOL(g_sat_dbg), z, sizeof(z));
    for (int i = 0; i < 3; i++) launch_sat_cols_pipe(0, sat, nc, nr);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int i = 0; i < 20; i++) launch_sat_cols_pipe(0, sat, nc, nr);
    hipEventRecord(e1); hipEventSynchronize(e1);
    hipEventElapsedTime(&ms, e0, e1);
    printf("sat_cols_pipe: %.2f us per launch\n", ms * 1000 / 20);
    hipMemcpyFromSymbol(d, HIP_SYMBOL(g_sat_dbg), sizeof(d));
    const long long t1 = d[(1 * 64 + 0) * 3];
    for (int s = 0; s < 19; s++) {
        printf("%3d  %6lld |", s, d[(0 * 64 + s) * 3] - t1);
        for (int w = 0; w < 6; w++) printf(" %4lld %4lld |", d[(w * 64 + s) * 3 + 1] - d[(w * 64 + s) * 3], d[(w * 64 + s) * 3 + 2] - d[(w * 64 + s) * 3 + 1]);
        printf("\n");
    }
    return 0;
}
