// Does the FP64 instruction MIX of the convolution kernels issue as fast as the single-instruction streams of valu_rate.hip?
// Body = the horizontal smoothing step of smooth_grad_rb (8 widenings, 8 pair adds, 12 multiplies by taps held in SGPRs,
// 8 accumulations, 4 narrowings = 40 FP64-rate instructions for 4 outputs), in registers only.  Variant B keeps the taps in VGPRs.
#include <hip/hip_runtime.h>
#include <cstdio>
#pragma clang fp contract(off)
constexpr int ITER = 4000;

template <bool TAPS_IN_VGPR>
__global__ __launch_bounds__(256) void body(float *out, double k0, double k1, double k2, float seed)
{
    double t0 = k0, t1 = k1, t2 = k2;
    if (TAPS_IN_VGPR) { asm volatile("" : "+v"(t0), "+v"(t1), "+v"(t2)); }
    float f[8];
    for (int i = 0; i < 8; i++) f[i] = seed + threadIdx.x + i;
    for (int it = 0; it < ITER; it++) {
        double v[8];
#pragma unroll
        for (int i = 0; i < 8; i++) v[i] = (double)f[i];
        float o[4];
#pragma unroll
        for (int q = 0; q < 4; q++) {
            double acc = v[q + 2] * t0;
            acc = acc + (v[q] + v[q + 4]) * t2;
            acc = acc + (v[q + 1] + v[q + 3]) * t1;
            o[q] = (float)acc;
        }
#pragma unroll
        for (int q = 0; q < 4; q++) { f[q] = o[q]; f[q + 4] = o[3 - q]; }
        asm volatile("" : "+v"(f[0]), "+v"(f[1]), "+v"(f[2]), "+v"(f[3]), "+v"(f[4]), "+v"(f[5]), "+v"(f[6]), "+v"(f[7]));
    }
    out[blockIdx.x * 256 + threadIdx.x] = f[0] + f[5];
}

int main()
{
    hipDeviceProp_t prop; hipGetDeviceProperties(&prop, 0);
    const int cus = prop.multiProcessorCount; const double mhz = prop.clockRate / 1000.0;
    float *out; hipMalloc(&out, sizeof(float) * 256 * cus * 8);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int variant = 0; variant < 2; variant++)
        for (int w : {1, 2, 4, 8}) {
            auto go = [&]() {
                if (variant) hipLaunchKernelGGL(body<true>, dim3(cus * w), dim3(256), 0, 0, out, 0.4, 0.25, 0.05, 1.f);
                else hipLaunchKernelGGL(body<false>, dim3(cus * w), dim3(256), 0, 0, out, 0.4, 0.25, 0.05, 1.f);
            };
            go(); hipDeviceSynchronize();
            hipEventRecord(e0); for (int r = 0; r < 5; r++) go(); hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            const double us = ms * 1000 / 5, cycles = us * mhz;
            printf("taps in %s, %d wavefronts per SIMD: %.1f us -> %.2f clocks per FP64-rate instruction per SIMD (40 per iteration)\n",
                   variant ? "VGPRs" : "SGPRs", w, us, cycles / (ITER * 40.0 * w));
        }
    return 0;
}
