// Summed-area tables as a wavefront pipeline (KLT_OPT_SAT_VARIANT = 1).
//
// numpy's cumsum(1).cumsum(0) on f32 (goodFeaturesUtils.pyx:49-51) is one strictly sequential f32 chain per row and
// then per column; the chains cannot be split, so the time of a pass is (chain length) x (time per element of the one
// wavefront that runs a band of chains).  Here that wavefront does nothing but read products from LDS, add, and write
// prefixes to LDS.  Loader wavefronts stream tiles from HBM into a ring of LDS slots, storer wavefronts drain a ring of
// finished tiles; the three kinds of wavefront are coupled only by counters in LDS (no workgroup barrier, so nobody
// waits for anybody's global memory traffic, and no register ring for the compiler to rotate).  Every wait is bounded:
// if a counter does not move for 2^20 polls the workgroup gives up and raises a flag the host checks.
#include "klt_internal.h"

#pragma clang fp contract(off)

#ifdef SAT_PIPE_DEBUG
__device__ long long g_sat_dbg[8 * 256];      // block 0: per tile {wait loaded, wait stored, compute done} clocks
#endif

namespace {

constexpr int P_NS = 8;            // input slots = loader wavefronts (each owns a slot)
constexpr int P_NO = 4;            // output slots = storer wavefronts
constexpr int P_T = 64 * (1 + P_NS + P_NO);
constexpr int P_SPIN = 1 << 20;

__device__ __forceinline__ void lds_done() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }

// wait until *p >= need (written by another wavefront of the workgroup); false: timed out
__device__ __forceinline__ bool wait_ge(volatile int *p, int need, volatile int *err)
{
    int guard = 0;
    while (*p < need) {
        __builtin_amdgcn_s_sleep(1);
        if (++guard > P_SPIN || *err) { *err = 1; return false; }
    }
    asm volatile("" ::: "memory");
    return true;
}

// ------------------------------------------------------------------ row pass (+ products)
constexpr int R_ROWS = 16, R_LD = 68;

__global__ __launch_bounds__(P_T) void sat_rows_pipe(const float *__restrict__ gx, const float *__restrict__ gy, float *__restrict__ sat,
                                                     int ncols, int nrows, int *__restrict__ error_flag)
{
    extern __shared__ __attribute__((aligned(16))) float pipe_lds[];
    typedef float Tile[R_ROWS * R_LD];
    Tile *const in_gx = reinterpret_cast<Tile *>(pipe_lds);                       // [P_NS]
    Tile *const in_gy = in_gx + P_NS;                                             // [P_NS]
    Tile(*const outp)[3] = reinterpret_cast<Tile(*)[3]>(in_gy + P_NS);            // [P_NO][3]
    __shared__ int loaded[P_NS], stored[P_NO], chain_pos, err;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int row0 = blockIdx.x * R_ROWS;
    const int ntiles = (ncols + 63) / 64;
    const size_t plane = (size_t)ncols * nrows;
    if (tid < P_NS) loaded[tid] = 0;
    if (tid < P_NO) stored[tid] = 0;
    if (tid == 0) { chain_pos = 0; err = 0; }
    __syncthreads();

    if (wave == 0) {                                            // ---- the chains: lane = plane * 16 + row
        __builtin_amdgcn_s_setprio(3);
        const int pl = lane >> 4, r = lane & 15;
        float carry = 0.f;
        for (int t = 0; t < ntiles; t++) {
            const int slot = t % P_NS, oslot = t % P_NO;
#ifdef SAT_PIPE_DEBUG
            const long long c0 = wall_clock64();
#endif
            if (!wait_ge(&loaded[slot], t + 1, &err)) break;
#ifdef SAT_PIPE_DEBUG
            const long long c1 = wall_clock64();
#endif
            if (t >= P_NO && !wait_ge(&stored[oslot], t - P_NO + 1, &err)) break;
#ifdef SAT_PIPE_DEBUG
            const long long c2 = wall_clock64();
#endif
            if (lane < 48) {
                const float4 *a4 = reinterpret_cast<const float4 *>(pl == 2 ? &in_gy[slot][r * R_LD] : &in_gx[slot][r * R_LD]);
                const float4 *b4 = reinterpret_cast<const float4 *>(pl == 0 ? &in_gx[slot][r * R_LD] : &in_gy[slot][r * R_LD]);
                float4 *o4 = reinterpret_cast<float4 *>(&outp[oslot][pl][r * R_LD]);
                float4 p[16];
#pragma unroll
                for (int c = 0; c < 16; c++) {
                    const float4 a = a4[c], b = b4[c];
                    p[c].x = a.x * b.x; p[c].y = a.y * b.y; p[c].z = a.z * b.z; p[c].w = a.w * b.w;
                }
#pragma unroll
                for (int c = 0; c < 16; c++) {
                    float4 o;
                    carry = carry + p[c].x; o.x = carry;
                    carry = carry + p[c].y; o.y = carry;
                    carry = carry + p[c].z; o.z = carry;
                    carry = carry + p[c].w; o.w = carry;
                    o4[c] = o;
                }
            }
            lds_done();
            if (lane == 0) *(volatile int *)&chain_pos = t + 1;
#ifdef SAT_PIPE_DEBUG
            if (blockIdx.x == 0 && lane == 0 && t < 256) {
                g_sat_dbg[8 * t] = c0; g_sat_dbg[8 * t + 1] = c1; g_sat_dbg[8 * t + 2] = c2; g_sat_dbg[8 * t + 3] = wall_clock64();
            }
#endif
        }
    } else if (wave <= P_NS) {                                  // ---- loaders: lane = column, slot = wavefront
        const int slot = wave - 1;
        for (int t = slot; t < ntiles; t += P_NS) {
            if (t >= P_NS && !wait_ge(&chain_pos, t - P_NS + 1, &err)) break;
            const int col = t * 64 + lane, colc = min(col, ncols - 1);
            float vx[R_ROWS], vy[R_ROWS];
#pragma unroll
            for (int r = 0; r < R_ROWS; r++) {                  // clamped, unconditional
                const size_t o = (size_t)min(row0 + r, nrows - 1) * ncols + colc;
                vx[r] = gx[o];
                vy[r] = gy[o];
            }
#pragma unroll
            for (int r = 0; r < R_ROWS; r++) {
                const bool ok = col < ncols && row0 + r < nrows;
                in_gx[slot][r * R_LD + lane] = ok ? vx[r] : 0.f;
                in_gy[slot][r * R_LD + lane] = ok ? vy[r] : 0.f;
            }
            lds_done();
            if (lane == 0) *(volatile int *)&loaded[slot] = t + 1;
        }
    } else {                                                    // ---- storers: lane = column, output slot = wavefront
        const int oslot = wave - 1 - P_NS;
        for (int t = oslot; t < ntiles; t += P_NO) {
            if (!wait_ge(&chain_pos, t + 1, &err)) break;
            float v[3][R_ROWS];
#pragma unroll
            for (int pl = 0; pl < 3; pl++)
#pragma unroll
                for (int r = 0; r < R_ROWS; r++) v[pl][r] = outp[oslot][pl][r * R_LD + lane];
            lds_done();
            if (lane == 0) *(volatile int *)&stored[oslot] = t + 1;
            const int col = t * 64 + lane;
            if (col < ncols) {
#pragma unroll
                for (int r = 0; r < R_ROWS; r++) {
                    const int row = row0 + r;
                    if (row < nrows) {
                        sat[(size_t)row * ncols + col] = v[0][r];
                        sat[plane + (size_t)row * ncols + col] = v[1][r];
                        sat[2 * plane + (size_t)row * ncols + col] = v[2][r];
                    }
                }
            }
        }
    }
    if (err && lane == 0) *error_flag = 1;
}

// ------------------------------------------------------------------ column pass (in place)
constexpr int C_COLS = 32, C_ROWS = 64;

__global__ __launch_bounds__(P_T) void sat_cols_pipe(float *__restrict__ sat, int ncols, int nrows, int *__restrict__ error_flag)
{
    extern __shared__ __attribute__((aligned(16))) float pipe_lds[];
    typedef float CTile[C_ROWS * C_COLS];
    CTile *const tin = reinterpret_cast<CTile *>(pipe_lds);                       // [P_NS]
    CTile *const tout = tin + P_NS;                                               // [P_NO]
    __shared__ int loaded[P_NS], stored[P_NO], chain_pos, err;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int x0 = blockIdx.x * C_COLS;
    float *s = sat + (size_t)blockIdx.y * ncols * nrows;
    const int ntiles = (nrows + C_ROWS - 1) / C_ROWS;
    if (tid < P_NS) loaded[tid] = 0;
    if (tid < P_NO) stored[tid] = 0;
    if (tid == 0) { chain_pos = 0; err = 0; }
    __syncthreads();
    const int lc = lane & 31, lh = lane >> 5;                   // loaders / storers: column, row parity

    if (wave == 0) {                                            // ---- the chains: lane = column (lanes 32..63 idle)
        __builtin_amdgcn_s_setprio(3);
        float carry = 0.f;
        for (int t = 0; t < ntiles; t++) {
            const int slot = t % P_NS, oslot = t % P_NO;
            if (!wait_ge(&loaded[slot], t + 1, &err)) break;
            if (t >= P_NO && !wait_ge(&stored[oslot], t - P_NO + 1, &err)) break;
            if (lane < C_COLS) {
                float v[C_ROWS];
#pragma unroll
                for (int r = 0; r < C_ROWS; r++) v[r] = tin[slot][r * C_COLS + lane];
#pragma unroll
                for (int r = 0; r < C_ROWS; r++) {
                    carry = carry + v[r];
                    tout[oslot][r * C_COLS + lane] = carry;
                }
            }
            lds_done();
            if (lane == 0) *(volatile int *)&chain_pos = t + 1;
        }
    } else if (wave <= P_NS) {
        const int slot = wave - 1;
        const int col = min(x0 + lc, ncols - 1);
        for (int t = slot; t < ntiles; t += P_NS) {
            if (t >= P_NS && !wait_ge(&chain_pos, t - P_NS + 1, &err)) break;
            float v[C_ROWS / 2];
#pragma unroll
            for (int i = 0; i < C_ROWS / 2; i++) v[i] = s[(size_t)min(t * C_ROWS + 2 * i + lh, nrows - 1) * ncols + col];
#pragma unroll
            for (int i = 0; i < C_ROWS / 2; i++) tin[slot][(2 * i + lh) * C_COLS + lc] = v[i];
            lds_done();
            if (lane == 0) *(volatile int *)&loaded[slot] = t + 1;
        }
    } else {
        const int oslot = wave - 1 - P_NS;
        for (int t = oslot; t < ntiles; t += P_NO) {
            if (!wait_ge(&chain_pos, t + 1, &err)) break;
            float v[C_ROWS / 2];
#pragma unroll
            for (int i = 0; i < C_ROWS / 2; i++) v[i] = tout[oslot][(2 * i + lh) * C_COLS + lc];
            lds_done();
            if (lane == 0) *(volatile int *)&stored[oslot] = t + 1;
            if (x0 + lc < ncols) {
#pragma unroll
                for (int i = 0; i < C_ROWS / 2; i++) {
                    const int row = t * C_ROWS + 2 * i + lh;
                    if (row < nrows) s[(size_t)row * ncols + x0 + lc] = v[i];
                }
            }
        }
    }
    if (err && lane == 0) *error_flag = 1;
}

}  // namespace

int launch_sat_rows_pipe(hipStream_t s, const float *gx, const float *gy, float *sat, int ncols, int nrows, int *error_flag)
{
    constexpr size_t lds = sizeof(float) * R_ROWS * R_LD * (2 * P_NS + 3 * P_NO);
    static bool set = false;
    if (!set) {
        hipError_t e = hipFuncSetAttribute((const void *)sat_rows_pipe, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return (int)e;
        set = true;
    }
    hipLaunchKernelGGL(sat_rows_pipe, dim3((nrows + R_ROWS - 1) / R_ROWS), dim3(P_T), lds, s, gx, gy, sat, ncols, nrows, error_flag);
    return 0;
}

int launch_sat_cols_pipe(hipStream_t s, float *sat, int ncols, int nrows, int *error_flag)
{
    constexpr size_t lds = sizeof(float) * C_ROWS * C_COLS * (P_NS + P_NO);
    static bool set = false;
    if (!set) {
        hipError_t e = hipFuncSetAttribute((const void *)sat_cols_pipe, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return (int)e;
        set = true;
    }
    hipLaunchKernelGGL(sat_cols_pipe, dim3((ncols + C_COLS - 1) / C_COLS, 3), dim3(P_T), lds, s, sat, ncols, nrows, error_flag);
    return 0;
}
