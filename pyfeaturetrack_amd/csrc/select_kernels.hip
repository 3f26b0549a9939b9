// Corner selection on gfx950: summed-area tables, min-eigenvalue score, candidate sort, greedy
// minimum-distance suppression.
//
// Parity notes:
//   * numpy's `cumsum(1).cumsum(0)` on f32 is a strictly sequential f32 prefix per row, then per column
//     (goodFeaturesUtils.pyx:49-51).  A parallel scan rounds differently and changes which pixels win,
//     so each row / column chain stays sequential here; parallelism comes from the number of chains
//     (rows x 3 planes, columns x 3 planes), with 64x64 tiles transposed through LDS so that HBM sees
//     coalesced 256-byte segments in both passes.
//   * window sum ((c + a) - b) - d in f32 (:23-31); eigenvalue with the reference's mixed f32/f64
//     arithmetic (:17-19).
//   * candidates are ordered by (val, x, y) descending (selectGoodFeatures.py:234-236): one 64-bit key
//     [f32 bits of val | x:16 | y:16] sorted descending gives exactly that order for val > 0.
//   * _enforceMinimumDistance (selectGoodFeatures.py:45-135) is a sequential greedy pass.  Equivalent
//     form used here: a candidate is accepted iff no previously accepted feature lies within Chebyshev
//     distance mindist-1.  With cells of side mindist at most one accepted feature fits in a cell, so
//     the test reads the 3x3 neighbouring cells of a small grid held in LDS instead of a full-frame map.
#include "klt_internal.h"

#pragma clang fp contract(off)

namespace {

// ------------------------------------------------------------------ SAT, row pass (+ products)
// block = one wavefront; blockIdx.x = band of 64 rows; blockIdx.y = plane (0: gx*gx, 1: gx*gy, 2: gy*gy)
__global__ __launch_bounds__(64) void sat_rows_kernel(const float *__restrict__ gx, const float *__restrict__ gy,
                                                       float *__restrict__ sat, int ncols, int nrows)
{
    __shared__ float tile[64][65];
    const int lane = threadIdx.x;
    const int row0 = blockIdx.x * 64;
    const int plane = blockIdx.y;
    float *out = sat + (size_t)plane * ncols * nrows;
    float carry = 0.f;
    for (int x0 = 0; x0 < ncols; x0 += 64) {
        const int col = x0 + lane;
#pragma unroll 8
        for (int r = 0; r < 64; r++) {
            const int row = row0 + r;
            float v = 0.f;
            if (row < nrows && col < ncols) {
                const float a = gx[(size_t)row * ncols + col];
                const float b = gy[(size_t)row * ncols + col];
                v = plane == 0 ? a * a : (plane == 1 ? a * b : b * b);
            }
            tile[r][lane] = v;
        }
        __syncthreads();
#pragma unroll 8
        for (int c = 0; c < 64; c++) {      // lane = row of the band: sequential f32 prefix along x
            carry = carry + tile[lane][c];
            tile[lane][c] = carry;
        }
        __syncthreads();
#pragma unroll 8
        for (int r = 0; r < 64; r++) {
            const int row = row0 + r;
            if (row < nrows && col < ncols) out[(size_t)row * ncols + col] = tile[r][lane];
        }
        __syncthreads();
    }
}

// ------------------------------------------------------------------ SAT, column pass (in place)
// thread = one column of one plane; sequential f32 prefix along y; loads are issued 8 rows ahead
__global__ __launch_bounds__(64) void sat_cols_kernel(float *__restrict__ sat, int ncols, int nrows)
{
    const int x = blockIdx.x * 64 + threadIdx.x;
    if (x >= ncols) return;
    float *s = sat + (size_t)blockIdx.y * ncols * nrows + x;
    float carry = 0.f;
    int y = 0;
    for (; y + 8 <= nrows; y += 8) {
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; u++) v[u] = s[(size_t)(y + u) * ncols];
#pragma unroll
        for (int u = 0; u < 8; u++) {
            carry = carry + v[u];
            s[(size_t)(y + u) * ncols] = carry;
        }
    }
    for (; y < nrows; y++) {
        carry = carry + s[(size_t)y * ncols];
        s[(size_t)y * ncols] = carry;
    }
}

// ------------------------------------------------------------------ seed map for REPLACING_SOME
// selectGoodFeatures.py:64-69: every live feature blocks the square of half-size d around (int(x), int(y))
__global__ void seed_fill_kernel(const klt_feat *__restrict__ fl, int nfeat, uint8_t *__restrict__ seedmap,
                                 int ncols, int nrows, int d)
{
    const int f = blockIdx.x;
    if (f >= nfeat) return;
    const klt_feat ft = fl[f];
    if (ft.val < 0) return;
    const int cx = (int)ft.x, cy = (int)ft.y, side = 2 * d + 1;
    for (int k = threadIdx.x; k < side * side; k += blockDim.x) {
        const int ix = cx - d + k % side, iy = cy - d + k / side;
        if (ix >= 0 && ix < ncols && iy >= 0 && iy < nrows) seedmap[(size_t)iy * ncols + ix] = 1;
    }
}

// ------------------------------------------------------------------ eigenvalue map + sort keys
__device__ __forceinline__ float window_sum(const float *__restrict__ s, int ncols, int x, int y, int hw, int hh)
{
    const float a = s[(size_t)(y - hh - 1) * ncols + (x - hw - 1)];
    const float b = s[(size_t)(y - hh - 1) * ncols + (x + hw)];
    const float c = s[(size_t)(y + hh) * ncols + (x + hw)];
    const float d = s[(size_t)(y + hh) * ncols + (x - hw - 1)];
    return ((c + a) - b) - d;
}

__global__ __launch_bounds__(256) void eigen_kernel(SelectArgs a)
{
    const int k = blockIdx.x * 256 + threadIdx.x;
    if (k >= a.npow2) return;
    const int ncand = a.nx * a.ny;
    if (k >= ncand) { a.keys[k] = 0ull; return; }     // padding sorts last
    const int xi = k % a.nx, yi = k / a.nx;
    const int x = a.bx + xi * a.step, y = a.by + yi * a.step;
    const size_t plane = (size_t)a.ncols * a.nrows;
    const float gxx = window_sum(a.sat, a.ncols, x, y, a.hw, a.hh);
    const float gxy = window_sum(a.sat + plane, a.ncols, x, y, a.hw, a.hh);
    const float gyy = window_sum(a.sat + 2 * plane, a.ncols, x, y, a.hw, a.hh);
    // goodFeaturesUtils.pyx:17-19 as compiled: (gxx-gyy)^2 in f32, 4*gxy*gxy and the sum in f64,
    // pow(.,0.5) -> f32, (gxx+gyy-s) in f32, /2 exact
    const float dif = gxx - gyy;
    const float sq = dif * dif;
    const double t = (double)sq + (4.0 * (double)gxy) * (double)gxy;
    const float s = (float)sqrt(t);
    const float sum = gxx + gyy;
    const float num = sum - s;
    const float val = (float)((double)num / 2.0);
    a.valmap[k] = val;
    bool ok = (double)val >= a.min_eig;                // val >= max(min_eigenvalue, 1) > 0
    if (ok && a.seedmap) ok = a.seedmap[(size_t)y * a.ncols + x] == 0;
    a.keys[k] = ok ? (((unsigned long long)__float_as_uint(val) << 32) | ((unsigned long long)x << 16) |
                      (unsigned long long)y)
                   : 0ull;
}

// ------------------------------------------------------------------ bitonic sort, descending, u64 keys
constexpr int SORT_E = 2048;      // keys per workgroup (16 KiB of LDS)
constexpr int SORT_T = 1024;

__device__ __forceinline__ void cmpx(unsigned long long &a, unsigned long long &b, bool desc)
{
    const bool sw = desc ? (a < b) : (a > b);
    if (sw) { const unsigned long long t = a; a = b; b = t; }
}

// full sort of each 2048-key chunk; chunk direction follows the global network (bit 11 of the index)
__global__ __launch_bounds__(SORT_T) void bitonic_local_sort(unsigned long long *__restrict__ keys)
{
    __shared__ unsigned long long s[SORT_E];
    const int t = threadIdx.x;
    const size_t base = (size_t)blockIdx.x * SORT_E;
    s[t] = keys[base + t];
    s[t + SORT_T] = keys[base + t + SORT_T];
    __syncthreads();
    for (int k = 2; k <= SORT_E; k <<= 1) {
        for (int j = k >> 1; j > 0; j >>= 1) {
            const int i = ((t & ~(j - 1)) << 1) | (t & (j - 1));
            const int l = i | j;
            const bool desc = (((base + i) & (size_t)k) == 0);
            unsigned long long a = s[i], b = s[l];
            cmpx(a, b, desc);
            s[i] = a;
            s[l] = b;
            __syncthreads();
        }
    }
    keys[base + t] = s[t];
    keys[base + t + SORT_T] = s[t + SORT_T];
}

// one compare-exchange step with partner distance j >= SORT_E, inside the stage that builds runs of length k
__global__ __launch_bounds__(256) void bitonic_global_step(unsigned long long *__restrict__ keys, int j, int k, int half_n)
{
    const int t = blockIdx.x * 256 + threadIdx.x;
    if (t >= half_n) return;
    const size_t i = ((size_t)(t & ~(j - 1)) << 1) | (size_t)(t & (j - 1));
    const size_t l = i | (size_t)j;
    const bool desc = ((i & (size_t)k) == 0);
    unsigned long long a = keys[i], b = keys[l];
    const bool sw = desc ? (a < b) : (a > b);
    if (sw) { keys[i] = b; keys[l] = a; }
}

// the remaining steps (j = 1024 .. 1) of stage k, inside LDS
__global__ __launch_bounds__(SORT_T) void bitonic_local_merge(unsigned long long *__restrict__ keys, int k)
{
    __shared__ unsigned long long s[SORT_E];
    const int t = threadIdx.x;
    const size_t base = (size_t)blockIdx.x * SORT_E;
    s[t] = keys[base + t];
    s[t + SORT_T] = keys[base + t + SORT_T];
    __syncthreads();
    for (int j = SORT_E >> 1; j > 0; j >>= 1) {
        const int i = ((t & ~(j - 1)) << 1) | (t & (j - 1));
        const int l = i | j;
        const bool desc = (((base + i) & (size_t)k) == 0);
        unsigned long long a = s[i], b = s[l];
        cmpx(a, b, desc);
        s[i] = a;
        s[l] = b;
        __syncthreads();
    }
    keys[base + t] = s[t];
    keys[base + t + SORT_T] = s[t + SORT_T];
}

// ------------------------------------------------------------------ greedy minimum-distance pass
// One wavefront walks the sorted keys 64 at a time.  Every lane tests its candidate against the cell
// grid, then the free lanes are resolved in rank order (lowest lane = highest rank) with ballots.
__global__ __launch_bounds__(64) void nms_kernel(NmsArgs a)
{
    extern __shared__ uint32_t lds_grid[];
    uint32_t *grid = a.grid_in_lds ? lds_grid : a.grid_global;   // flat pointer: LDS or global
    const int lane = threadIdx.x;
    const int ncell = a.gw * a.gh;
    if (a.grid_in_lds) {
        for (int i = lane; i < ncell; i += 64) grid[i] = 0u;
    }
    __syncthreads();

    int indx = 0, placed = 0;
    bool list_full = false, keys_done = false;
    for (int pos = 0; pos < a.nkeys && !list_full && !keys_done; pos += 64) {
        const int kidx = pos + lane;
        const unsigned long long key = kidx < a.nkeys ? a.keys[kidx] : 0ull;
        const bool valid = key != 0ull;
        const int x = (int)((key >> 16) & 0xffffull), y = (int)(key & 0xffffull);
        const float val = __uint_as_float((uint32_t)(key >> 32));
        bool is_free = valid;
        if (valid && a.d >= 0) {
            const int cx = x / a.cell, cy = y / a.cell;
            for (int dy = -1; dy <= 1; dy++)
                for (int dx = -1; dx <= 1; dx++) {
                    const int gx = cx + dx, gy = cy + dy;
                    if (gx < 0 || gy < 0 || gx >= a.gw || gy >= a.gh) continue;
                    const uint32_t v = __hip_atomic_load(&grid[gy * a.gw + gx], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    if (v) {
                        const int ax = (int)((v - 1u) >> 16), ay = (int)((v - 1u) & 0xffffu);
                        if (abs(x - ax) <= a.d && abs(y - ay) <= a.d) is_free = false;
                    }
                }
        }
        unsigned long long mask = __ballot(is_free);
        if (__ballot(!valid) != 0ull) keys_done = true;    // a zero key: everything after it is padding / rejected
        while (mask != 0ull) {
            // next slot to fill (selectGoodFeatures.py:109-112)
            if (!a.overwrite_all)
                while (indx < a.nfeat && a.fl[indx].val >= 0) indx++;
            if (indx >= a.nfeat) { list_full = true; break; }
            const int l = __ffsll((long long)mask) - 1;
            const int ax = __shfl(x, l), ay = __shfl(y, l);
            const float aval = __shfl(val, l);
            if (lane == l) {
                klt_feat ft;
                ft.x = (float)ax;
                ft.y = (float)ay;
                ft.val = (int32_t)aval;        // int(val): truncation (selectGoodFeatures.py:119)
                ft.aux = 0;
                a.fl[indx] = ft;
                if (a.d >= 0)
                    __hip_atomic_store(&grid[(ay / a.cell) * a.gw + (ax / a.cell)], (((uint32_t)ax << 16) | (uint32_t)ay) + 1u,
                                       __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            indx++;
            placed++;
            mask &= ~(1ull << l);
            // later candidates of this batch that the new feature blocks
            const bool hit = a.d >= 0 && is_free && lane > l && abs(x - ax) <= a.d && abs(y - ay) <= a.d;
            mask &= ~__ballot(hit);
        }
        __syncthreads();       // grid / feature-list writes visible to the next batch
    }
    // candidates exhausted: selectGoodFeatures.py:78-94 (SELECTING_ALL only; see DESIGN.md for the deviation)
    if (!list_full && a.overwrite_all) {
        for (int i = indx + lane; i < a.nfeat; i += 64) {
            klt_feat ft;
            ft.x = -1.f;
            ft.y = -1.f;
            ft.val = KLT_NOT_FOUND;
            ft.aux = 0;
            a.fl[i] = ft;
        }
    }
    if (lane == 0 && a.placed_out) *a.placed_out = placed;
}

__global__ void unpack_candidates_kernel(const unsigned long long *__restrict__ keys, int n, float *__restrict__ val,
                                         int *__restrict__ x, int *__restrict__ y)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const unsigned long long key = keys[i];
    val[i] = __uint_as_float((uint32_t)(key >> 32));
    x[i] = (int)((key >> 16) & 0xffffull);
    y[i] = (int)(key & 0xffffull);
}

}  // namespace

void launch_sat_rows(hipStream_t s, const float *gx, const float *gy, float *sat, int ncols, int nrows)
{
    hipLaunchKernelGGL(sat_rows_kernel, dim3((nrows + 63) / 64, 3), dim3(64), 0, s, gx, gy, sat, ncols, nrows);
}

void launch_sat_cols(hipStream_t s, float *sat, int ncols, int nrows)
{
    hipLaunchKernelGGL(sat_cols_kernel, dim3((ncols + 63) / 64, 3), dim3(64), 0, s, sat, ncols, nrows);
}

void launch_seed_fill(hipStream_t s, const klt_feat *fl, int nfeat, uint8_t *seedmap, int ncols, int nrows, int d)
{
    if (nfeat <= 0 || d < 0) return;
    hipLaunchKernelGGL(seed_fill_kernel, dim3(nfeat), dim3(64), 0, s, fl, nfeat, seedmap, ncols, nrows, d);
}

void launch_eigen(hipStream_t s, const SelectArgs &a)
{
    hipLaunchKernelGGL(eigen_kernel, dim3((a.npow2 + 255) / 256), dim3(256), 0, s, a);
}

void launch_sort_desc(hipStream_t s, unsigned long long *keys, int n)
{
    // n is a power of two >= SORT_E
    hipLaunchKernelGGL(bitonic_local_sort, dim3(n / SORT_E), dim3(SORT_T), 0, s, keys);
    for (long long k = 2LL * SORT_E; k <= n; k <<= 1) {
        for (long long j = k >> 1; j >= SORT_E; j >>= 1)
            hipLaunchKernelGGL(bitonic_global_step, dim3((n / 2 + 255) / 256), dim3(256), 0, s, keys, (int)j, (int)k, n / 2);
        hipLaunchKernelGGL(bitonic_local_merge, dim3(n / SORT_E), dim3(SORT_T), 0, s, keys, (int)k);
    }
}

int launch_nms(hipStream_t s, const NmsArgs &a)
{
    size_t lds = a.grid_in_lds ? (size_t)a.gw * a.gh * sizeof(uint32_t) : 0;
    if (lds > 48 * 1024) {
        hipError_t e = hipFuncSetAttribute((const void *)nms_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return (int)e;
    }
    hipLaunchKernelGGL(nms_kernel, dim3(1), dim3(64), lds, s, a);
    return 0;
}

void launch_unpack_candidates(hipStream_t s, const unsigned long long *keys, int n, float *val, int *x, int *y)
{
    if (n <= 0) return;
    hipLaunchKernelGGL(unpack_candidates_kernel, dim3((n + 255) / 256), dim3(256), 0, s, keys, n, val, x, y);
}
