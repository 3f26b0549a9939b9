// Cost of the LDS access patterns of smooth_grad_rb (ds_read_b128 / ds_write_b128 with the lane -> address maps of its stages)
// against a contiguous pattern.  Reports clocks per wavefront-instruction with all four SIMDs of a CU reading.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr int ITER = 2000, UNR = 8;

__global__ __launch_bounds__(256) void rd(float *out, const int *addr_tab, int write)
{
    extern __shared__ __attribute__((aligned(16))) float l[];
    for (int i = threadIdx.x; i < 10240; i += 256) l[i] = i;
    __syncthreads();
    const unsigned a = (unsigned)(size_t)l + 4u * (unsigned)addr_tab[threadIdx.x];     // byte address of this lane's quad
    f32x4 s = {0, 0, 0, 0};
    if (write) {
        for (int it = 0; it < ITER; it++) {
#pragma unroll
            for (int u = 0; u < UNR; u++) asm volatile("ds_write_b128 %0, %1" :: "v"(a), "v"(s) : "memory");
            asm volatile("s_waitcnt lgkmcnt(0)");
        }
    } else {
        for (int it = 0; it < ITER; it++) {
#pragma unroll
            for (int u = 0; u < UNR; u++) { f32x4 v; asm volatile("ds_read_b128 %0, %1" : "=v"(v) : "v"(a) : "memory"); s += v; }
            asm volatile("s_waitcnt lgkmcnt(0)");
        }
    }
    out[blockIdx.x * 256 + threadIdx.x] = s.x + s.y;
}

int main()
{
    hipDeviceProp_t prop; hipGetDeviceProperties(&prop, 0);
    const int cus = prop.multiProcessorCount; const double mhz = prop.clockRate / 1000.0;
    float *out; hipMalloc(&out, 4 * 256 * cus * 4);
    int *tab; hipMalloc(&tab, 4 * 256);
    struct { const char *name; int rows_step, Q, stride; } pats[] = {
        {"contiguous (lane * 4 floats)", 1, 256, 0},
        {"stage 1 read A : r = i / 18, q = i % 18, row stride 80", 1, 18, 80},
        {"stage 2 read B : r = 2 (i / 18), q = i % 18, row stride 72", 2, 18, 72},
        {"stage 3 read C : r = i / 16, q = i % 16, row stride 72", 1, 16, 72},
        {"stage 4 read D : r = 2 (i / 16), q = i % 16, row stride 64", 2, 16, 64},
        {"stage 2 with row stride 76", 2, 18, 76},
        {"stage 2 with row stride 84", 2, 18, 84},
        {"stage 1 with row stride 84", 1, 18, 84},
        {"stage 1 with row stride 88", 1, 18, 88},
    };
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (auto &p : pats) {
        int h[256];
        for (int i = 0; i < 256; i++) h[i] = p.stride ? (p.rows_step * (i / p.Q)) * p.stride + 4 * (i % p.Q) : 4 * i;
        hipMemcpy(tab, h, sizeof(h), hipMemcpyHostToDevice);
        for (int write = 0; write < 2; write++) {
            hipLaunchKernelGGL(rd, dim3(cus * 2), dim3(256), 40960, 0, out, tab, write);
            hipDeviceSynchronize();
            hipEventRecord(e0);
            for (int r = 0; r < 3; r++) hipLaunchKernelGGL(rd, dim3(cus * 2), dim3(256), 40960, 0, out, tab, write);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            const double cycles = ms * 1000 / 3 * mhz;
            printf("%-62s %s: %.1f clocks per wavefront-instruction per CU\n", p.name, write ? "write" : "read ", cycles / (ITER * UNR * 8.0));
        }
    }
    return 0;
}
