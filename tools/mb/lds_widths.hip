// LDS instruction cost by access width (contiguous lanes): clocks per wavefront-instruction per CU and bytes per clock.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
constexpr int ITER = 2000, UNR = 8;
template <int W, bool WRITE>
__global__ __launch_bounds__(256) void k(float *out)
{
    extern __shared__ __attribute__((aligned(16))) float l[];
    for (int i = threadIdx.x; i < 4096; i += 256) l[i] = i;
    __syncthreads();
    const unsigned a = (unsigned)(size_t)l + (unsigned)(W * 4) * threadIdx.x;
    f32x4 s4 = {1, 2, 3, 4}; f32x2 s2 = {1, 2}; float s1 = 1;
    for (int it = 0; it < ITER; it++) {
#pragma unroll
        for (int u = 0; u < UNR; u++) {
            if (WRITE) {
                if (W == 4) asm volatile("ds_write_b128 %0, %1" :: "v"(a), "v"(s4) : "memory");
                if (W == 2) asm volatile("ds_write_b64 %0, %1" :: "v"(a), "v"(s2) : "memory");
                if (W == 1) asm volatile("ds_write_b32 %0, %1" :: "v"(a), "v"(s1) : "memory");
            } else {
                if (W == 4) asm volatile("ds_read_b128 %0, %1" : "=v"(s4) : "v"(a) : "memory");
                if (W == 2) asm volatile("ds_read_b64 %0, %1" : "=v"(s2) : "v"(a) : "memory");
                if (W == 1) asm volatile("ds_read_b32 %0, %1" : "=v"(s1) : "v"(a) : "memory");
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)");
    }
    out[blockIdx.x * 256 + threadIdx.x] = s4.x + s2.x + s1;
}
template <int W, bool WRITE>
static void run(const char *name, int cus, double mhz, float *out)
{
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((k<W, WRITE>), dim3(cus * 2), dim3(256), 16384, 0, out); hipDeviceSynchronize();
    hipEventRecord(e0); for (int r = 0; r < 3; r++) hipLaunchKernelGGL((k<W, WRITE>), dim3(cus * 2), dim3(256), 16384, 0, out);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double clk = ms * 1000 / 3 * mhz / (ITER * UNR * 8.0);
    printf("%-16s %5.1f clocks per wavefront-instruction per CU = %5.1f bytes per clock\n", name, clk, 64.0 * W * 4 / clk);
}
int main()
{
    hipDeviceProp_t prop; hipGetDeviceProperties(&prop, 0);
    const int cus = prop.multiProcessorCount; const double mhz = prop.clockRate / 1000.0;
    float *out; hipMalloc(&out, 4 * 256 * cus * 4);
    run<1, false>("ds_read_b32", cus, mhz, out); run<2, false>("ds_read_b64", cus, mhz, out); run<4, false>("ds_read_b128", cus, mhz, out);
    run<1, true>("ds_write_b32", cus, mhz, out); run<2, true>("ds_write_b64", cus, mhz, out); run<4, true>("ds_write_b128", cus, mhz, out);
    return 0;
}
