// Step timing of the fused column pass + eigenvalue kernel (sat_pipeline.hip, cols_eigen_pipe): the product source built with
// SAT_PIPE_DEBUG; prints the launch time alone on the GPU (against sat_cols_pipe alone) and, for workgroup 0, how long every wavefront
// worked and waited in each step (ticks of 10 ns).
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off -DSAT_PIPE_DEBUG -Iinclude -Ipyfeaturetrack_amd/csrc tools/mb/cols_eigen_steps.hip -o tools/mb/cols_eigen_steps
#include "../../pyfeaturetrack_amd/csrc/sat_pipeline.hip"
thread_local hipEvent_t g_klt_stamp_start = nullptr, g_klt_stamp_stop = nullptr;     // the timing hooks of klt_launch (api_context.hip in the library)
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
int main(int argc, char **argv)
{
    const int nc = argc > 2 ? atoi(argv[1]) : 3840, nr = argc > 2 ? atoi(argv[2]) : 2160, border = argc > 3 ? atoi(argv[3]) : 120;
    float *sat;
    unsigned long long *keys;
    const size_t N = (size_t)nc * nr;
    hipMalloc(&sat, 12 * N + 4 * KLT_SAT_PAD);
    hipMalloc(&keys, 8 * N);
    std::vector<float> h(3 * N);
    for (size_t i = 0; i < h.size(); i++) h[i] = (float)((i * 2654435761u) >> 20) * 1e-3f;
    hipMemcpy(sat, h.data(), 4 * h.size(), hipMemcpyHostToDevice);
    SelectArgs a;
    memset(&a, 0, sizeof(a));
    a.sat = sat; a.keys = keys; a.min_eig = 1.0; a.ncols = nc; a.nrows = nr; a.bx = a.by = border; a.step = 1;
    a.nx = nc - 2 * border; a.ny = nr - 2 * border; a.hw = a.hh = 3;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float ms;
    for (int i = 0; i < 3; i++) launch_sat_cols_eigen_pipe(0, sat, a);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int i = 0; i < 20; i++) launch_sat_cols_eigen_pipe(0, sat, a);
    hipEventRecord(e1); hipEventSynchronize(e1);
    hipEventElapsedTime(&ms, e0, e1);
    printf("cols_eigen_pipe alone: %.2f us per launch (%d x %d, %d strips, %d waves per workgroup: %d chain, %d loaders, %d eigen)\n", ms * 1000 / 20, nc, nr,
           (a.nx + FW - 8) / (FW - 7), FUSED_THREADS / 64, FN_CHAIN, FN_LOAD, FN_EIGEN);
    static long long d[16 * 96 * 3];
    hipMemcpyFromSymbol(d, HIP_SYMBOL(g_fused_dbg), sizeof(d));
    const int W = FUSED_THREADS / 64;
    const long long t0 = d[0];
    printf("step | start | per wavefront: work wait (ticks of 10 ns)\n");
    for (int s = 0; s < 24; s++) {
        printf("%3d %6lld |", s, d[(0 * 96 + s) * 3] - t0);
        for (int w = 0; w < W; w++) printf(" %3lld %3lld |", d[(w * 96 + s) * 3 + 1] - d[(w * 96 + s) * 3], d[(w * 96 + s) * 3 + 2] - d[(w * 96 + s) * 3 + 1]);
        printf("\n");
    }
    for (int i = 0; i < 3; i++) launch_sat_cols_pipe(0, sat, nc, nr);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int i = 0; i < 20; i++) launch_sat_cols_pipe(0, sat, nc, nr);
    hipEventRecord(e1); hipEventSynchronize(e1);
    hipEventElapsedTime(&ms, e0, e1);
    printf("sat_cols_pipe alone: %.2f us per launch\n", ms * 1000 / 20);
    return 0;
}
