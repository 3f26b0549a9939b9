// Instruction-rate probe for gfx950: how many lane-ops per clock per CU do the FP64 / conversion / LDS instructions
// of the convolution kernels sustain?  Build: hipcc -O3 --offload-arch=gfx950 valu_rate.hip -o valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

constexpr int ITER = 2000, UNR = 16;

#define KERNEL(name, decl, body, sink)                                                         \
    __global__ __launch_bounds__(256) void name(float *out, float seed)                        \
    {                                                                                          \
        decl;                                                                                  \
        for (int it = 0; it < ITER; it++) {                                                    \
            _Pragma("unroll") for (int u = 0; u < UNR; u++) { body; }                          \
        }                                                                                      \
        sink;                                                                                  \
    }

KERNEL(k_cvt_f64_f32, float f[UNR]; double d[UNR]; for (int u = 0; u < UNR; u++) f[u] = seed + u + threadIdx.x,
       asm volatile("v_cvt_f64_f32 %0, %1" : "=v"(d[u]) : "v"(f[u])),
       double s = 0; for (int u = 0; u < UNR; u++) s += d[u]; out[blockIdx.x * 256 + threadIdx.x] = (float)s)
KERNEL(k_cvt_f32_f64, double f[UNR]; float d[UNR]; for (int u = 0; u < UNR; u++) f[u] = seed + u + threadIdx.x,
       asm volatile("v_cvt_f32_f64 %0, %1" : "=v"(d[u]) : "v"(f[u])),
       float s = 0; for (int u = 0; u < UNR; u++) s += d[u]; out[blockIdx.x * 256 + threadIdx.x] = s)
KERNEL(k_add_f64, double f[UNR]; for (int u = 0; u < UNR; u++) f[u] = seed + u + threadIdx.x,
       asm volatile("v_add_f64 %0, %0, %1" : "+v"(f[u]) : "v"(f[(u + 1) % UNR])),
       double s = 0; for (int u = 0; u < UNR; u++) s += f[u]; out[blockIdx.x * 256 + threadIdx.x] = (float)s)
KERNEL(k_mul_f64, double f[UNR]; double g = 1.0000001 + seed; for (int u = 0; u < UNR; u++) f[u] = seed + u + threadIdx.x,
       asm volatile("v_mul_f64 %0, %0, %1" : "+v"(f[u]) : "v"(g)),
       double s = 0; for (int u = 0; u < UNR; u++) s += f[u]; out[blockIdx.x * 256 + threadIdx.x] = (float)s)
KERNEL(k_fma_f64, double f[UNR]; double g = 1.0000001 + seed; for (int u = 0; u < UNR; u++) f[u] = seed + u + threadIdx.x,
       asm volatile("v_fma_f64 %0, %0, %1, %1" : "+v"(f[u]) : "v"(g)),
       double s = 0; for (int u = 0; u < UNR; u++) s += f[u]; out[blockIdx.x * 256 + threadIdx.x] = (float)s)
KERNEL(k_add_f32, float f[UNR]; for (int u = 0; u < UNR; u++) f[u] = seed + u + threadIdx.x,
       asm volatile("v_add_f32 %0, %0, %1" : "+v"(f[u]) : "v"(f[(u + 1) % UNR])),
       float s = 0; for (int u = 0; u < UNR; u++) s += f[u]; out[blockIdx.x * 256 + threadIdx.x] = s)
KERNEL(k_and_or_b32, unsigned f[UNR]; for (int u = 0; u < UNR; u++) f[u] = (unsigned)seed + u + threadIdx.x,
       asm volatile("v_and_or_b32 %0, %0, %1, %1" : "+v"(f[u]) : "v"(f[(u + 1) % UNR])),
       unsigned s = 0; for (int u = 0; u < UNR; u++) s += f[u]; out[blockIdx.x * 256 + threadIdx.x] = (float)s)
KERNEL(k_pk_add_f32, double f[UNR]; for (int u = 0; u < UNR; u++) f[u] = seed + u + threadIdx.x,
       asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(f[u]) : "v"(f[(u + 1) % UNR])),
       double s = 0; for (int u = 0; u < UNR; u++) s += f[u]; out[blockIdx.x * 256 + threadIdx.x] = (float)s)
// dependent chains: one accumulator, UNR ops in sequence per iteration
KERNEL(k_dep_add_f32, float acc = seed; float g = seed + threadIdx.x,
       asm volatile("v_add_f32 %0, %0, %1" : "+v"(acc) : "v"(g)),
       out[blockIdx.x * 256 + threadIdx.x] = acc)
KERNEL(k_dep_add_f64, double acc = seed; double g = seed + threadIdx.x,
       asm volatile("v_add_f64 %0, %0, %1" : "+v"(acc) : "v"(g)),
       out[blockIdx.x * 256 + threadIdx.x] = (float)acc)

__global__ __launch_bounds__(256) void k_lds_b128(float *out, float seed)
{
    __shared__ __attribute__((aligned(16))) float l[256 * 4 + 64];
    for (int i = threadIdx.x; i < 256 * 4 + 64; i += 256) l[i] = seed + i;
    __syncthreads();
    float4 s = {0, 0, 0, 0};
    const float4 *p = reinterpret_cast<const float4 *>(l) + threadIdx.x;
    for (int it = 0; it < ITER; it++) {
#pragma unroll
        for (int u = 0; u < UNR; u++) {
            float4 v;
            asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"((unsigned)(size_t)p), "n"(0));
            asm volatile("s_waitcnt lgkmcnt(8)");
            s.x += v.x;
        }
    }
    asm volatile("s_waitcnt lgkmcnt(0)");
    out[blockIdx.x * 256 + threadIdx.x] = s.x + s.y;
}
__global__ __launch_bounds__(256) void k_lds_b32(float *out, float seed)
{
    __shared__ __attribute__((aligned(16))) float l[256 * 4 + 64];
    for (int i = threadIdx.x; i < 256 * 4 + 64; i += 256) l[i] = seed + i;
    __syncthreads();
    float s = 0;
    const float *p = l + threadIdx.x;
    for (int it = 0; it < ITER; it++) {
#pragma unroll
        for (int u = 0; u < UNR; u++) {
            float v;
            asm volatile("ds_read_b32 %0, %1" : "=v"(v) : "v"((unsigned)(size_t)p));
            asm volatile("s_waitcnt lgkmcnt(8)");
            s += v;
        }
    }
    asm volatile("s_waitcnt lgkmcnt(0)");
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

typedef void (*kern_t)(float *, float);

int main()
{
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    const double mhz = prop.clockRate / 1000.0;
    printf("%s: %d CUs, %.0f MHz\n", prop.name, cus, mhz);
    float *out;
    const int waves_per_simd[] = {1, 2, 4};
    CHECK(hipMalloc(&out, sizeof(float) * 256 * cus * 16));
    struct { const char *name; kern_t k; double lanes_per_instr; } ks[] = {
        {"v_cvt_f64_f32", k_cvt_f64_f32, 64}, {"v_cvt_f32_f64", k_cvt_f32_f64, 64}, {"v_add_f64", k_add_f64, 64}, {"v_mul_f64", k_mul_f64, 64},
        {"v_fma_f64", k_fma_f64, 64}, {"v_add_f32", k_add_f32, 64}, {"v_and_or_b32", k_and_or_b32, 64}, {"v_pk_add_f32", k_pk_add_f32, 64},
        {"dep v_add_f32", k_dep_add_f32, 64}, {"dep v_add_f64", k_dep_add_f64, 64}, {"ds_read_b128", k_lds_b128, 64}, {"ds_read_b32", k_lds_b32, 64}};
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    for (auto &k : ks) {
        for (int w : waves_per_simd) {
            const int blocks = cus * w;             // 256 threads = 4 waves = one per SIMD; w blocks per CU
            hipLaunchKernelGGL(k.k, dim3(blocks), dim3(256), 0, 0, out, 1.0f);
            CHECK(hipDeviceSynchronize());
            CHECK(hipEventRecord(e0));
            for (int r = 0; r < 5; r++) hipLaunchKernelGGL(k.k, dim3(blocks), dim3(256), 0, 0, out, 1.0f);
            CHECK(hipEventRecord(e1));
            CHECK(hipEventSynchronize(e1));
            float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
            const double us = ms * 1000.0 / 5;
            const double instr_per_wave = (double)ITER * UNR;
            const double cycles = us * mhz;         // at the reported clock
            // cycles per instruction per wave slot on one SIMD with w waves resident
            printf("%-16s waves/SIMD %d: %8.1f us  -> %.2f clk per wave-instr per SIMD (%.1f lane-ops/clk/CU)\n", k.name, w, us,
                   cycles / (instr_per_wave * w), 4.0 * 64.0 * instr_per_wave * w / cycles);
        }
    }
    return 0;
}
