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
// The only serial work is the chain itself: 64 dependent f32 adds per 64-column tile and row.  Everything else is kept
// off the chain's wavefront.  block = 5 wavefronts = one band of 16 rows: wavefront 0 runs the 48 chains of the band
// (lane = plane * 16 + row; planes gx*gx, gx*gy, gy*gy) from LDS to LDS; wavefronts 1..4 load gx / gy (coalesced
// 256-byte row segments, a ring of SR_D tiles in flight per lane to cover HBM latency), write the three products of
// the next tile into LDS and store the finished prefix tile of the previous step.  One barrier per tile.
constexpr int SR_ROWS = 16, SR_LD = 68, SR_D = 8, SR_T = 320;

__global__ __launch_bounds__(SR_T) void sat_rows_kernel(const float *__restrict__ gx, const float *__restrict__ gy,
                                                         float *__restrict__ sat, int ncols, int nrows)
{
    __shared__ __attribute__((aligned(16))) float prod[2][3][SR_ROWS * SR_LD];
    __shared__ __attribute__((aligned(16))) float outp[2][3][SR_ROWS * SR_LD];
    const int tid = threadIdx.x, lane = tid & 63;
    const bool chain = tid < 64;
    const int row0 = blockIdx.x * SR_ROWS;
    const int ntiles = (ncols + 63) / 64;
    const size_t plane = (size_t)ncols * nrows;
    const int lr = (tid - 64) >> 6, lc = lane;                 // loader lanes: rows lr, lr + 4, lr + 8, lr + 12 of the band
    float rgx[SR_D][4], rgy[SR_D][4];
    auto fetch = [&](int slot, int t) {
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const int row = row0 + lr + 4 * k, col = t * 64 + lc;
            const bool ok = t < ntiles && row < nrows && col < ncols;
            rgx[slot][k] = ok ? gx[KLT_GRAD_STRIDE * ((size_t)row * ncols + col)] : 0.f;      // interleaved gradient planes (klt_internal.h)
            rgy[slot][k] = ok ? gy[KLT_GRAD_STRIDE * ((size_t)row * ncols + col)] : 0.f;
        }
    };
    auto put = [&](int slot, int buf) {
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const int o = (lr + 4 * k) * SR_LD + lc;
            const float a = rgx[slot][k], b = rgy[slot][k];
            prod[buf][0][o] = a * a;
            prod[buf][1][o] = a * b;
            prod[buf][2][o] = b * b;
        }
    };
    auto store = [&](int buf, int t) {
        const int col = t * 64 + lc;
        if (col < ncols) {
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const int r = lr + 4 * k, row = row0 + r;
                if (row < nrows) {
                    sat[(size_t)row * ncols + col] = outp[buf][0][r * SR_LD + lc];
                    sat[plane + (size_t)row * ncols + col] = outp[buf][1][r * SR_LD + lc];
                    sat[2 * plane + (size_t)row * ncols + col] = outp[buf][2][r * SR_LD + lc];
                }
            }
        }
    };
    float carry = 0.f;
    const int cpl = lane >> 4, cr = lane & 15;                 // chain lanes: plane, row (lanes 48..63 idle)
    auto scan = [&](int buf) {
        if (lane < 48) {
            const float4 *a4 = reinterpret_cast<const float4 *>(&prod[buf][cpl][cr * SR_LD]);
            float4 *o4 = reinterpret_cast<float4 *>(&outp[buf][cpl][cr * SR_LD]);
            float4 v[16];
#pragma unroll
            for (int c = 0; c < 16; c++) v[c] = a4[c];
#pragma unroll
            for (int c = 0; c < 16; c++) {
                float4 o;
                carry = carry + v[c].x; o.x = carry;
                carry = carry + v[c].y; o.y = carry;
                carry = carry + v[c].z; o.z = carry;
                carry = carry + v[c].w; o.w = carry;
                o4[c] = o;
            }
        }
    };
    if (!chain) {
#pragma unroll
        for (int s = 0; s < SR_D; s++) fetch(s, s);
        put(0, 0);
        fetch(0, SR_D);
    }
    __syncthreads();
    for (int t0 = 0; t0 < ntiles; t0 += SR_D) {
#pragma unroll
        for (int k = 0; k < SR_D; k++) {
            const int t = t0 + k;
            if (t < ntiles) {                                   // uniform
                if (chain) {
                    scan(k & 1);
                } else {
                    if (t + 1 < ntiles) {
                        put((k + 1) % SR_D, (k + 1) & 1);
                        fetch((k + 1) % SR_D, t + 1 + SR_D);
                    }
                    if (t >= 1) store((k + 1) & 1, t - 1);
                }
                __syncthreads();
            }
        }
    }
    if (!chain) store((ntiles - 1) & 1, ntiles - 1);
}

// ------------------------------------------------------------------ SAT, column pass (in place)
// Same structure: block = 32 columns of one plane; wavefront 0 runs the 32 column chains (lane = column) on 32-row
// tiles in LDS, wavefronts 1..4 stream the tiles in (ring of SC_D tiles per lane) and the finished tiles out.
constexpr int SC_COLS = 32, SC_ROWS = 32, SC_D = 12, SC_T = 320;

__global__ __launch_bounds__(SC_T) void sat_cols_kernel(float *__restrict__ sat, int ncols, int nrows)
{
    __shared__ float tin[2][SC_ROWS * SC_COLS], tout[2][SC_ROWS * SC_COLS];
    const int tid = threadIdx.x, lane = tid & 63;
    const bool chain = tid < 64;
    const int x0 = blockIdx.x * SC_COLS;
    float *s = sat + (size_t)blockIdx.y * ncols * nrows;
    const int ntiles = (nrows + SC_ROWS - 1) / SC_ROWS;
    const int lr = (tid - 64) >> 5, lc = (tid - 64) & 31;       // loader lanes: rows lr, lr + 8, lr + 16, lr + 24 of the tile
    float rg[SC_D][4];
    auto fetch = [&](int slot, int t) {
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const int row = t * SC_ROWS + lr + 8 * k, col = x0 + lc;
            rg[slot][k] = (t < ntiles && row < nrows && col < ncols) ? s[(size_t)row * ncols + col] : 0.f;
        }
    };
    auto put = [&](int slot, int buf) {
#pragma unroll
        for (int k = 0; k < 4; k++) tin[buf][(lr + 8 * k) * SC_COLS + lc] = rg[slot][k];
    };
    auto store = [&](int buf, int t) {
        const int col = x0 + lc;
        if (col < ncols) {
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const int r = lr + 8 * k, row = t * SC_ROWS + r;
                if (row < nrows) s[(size_t)row * ncols + col] = tout[buf][r * SC_COLS + lc];
            }
        }
    };
    float carry = 0.f;
    auto scan = [&](int buf) {
        if (lane < SC_COLS) {
            float v[SC_ROWS];
#pragma unroll
            for (int r = 0; r < SC_ROWS; r++) v[r] = tin[buf][r * SC_COLS + lane];
#pragma unroll
            for (int r = 0; r < SC_ROWS; r++) {
                carry = carry + v[r];
                tout[buf][r * SC_COLS + lane] = carry;
            }
        }
    };
    if (!chain) {
#pragma unroll
        for (int k = 0; k < SC_D; k++) fetch(k, k);
        put(0, 0);
        fetch(0, SC_D);
    }
    __syncthreads();
    for (int t0 = 0; t0 < ntiles; t0 += SC_D) {
#pragma unroll
        for (int k = 0; k < SC_D; k++) {
            const int t = t0 + k;
            if (t < ntiles) {                                   // uniform
                if (chain) {
                    scan(k & 1);
                } else {
                    if (t + 1 < ntiles) {
                        put((k + 1) % SC_D, (k + 1) & 1);
                        fetch((k + 1) % SC_D, t + 1 + SC_D);
                    }
                    if (t >= 1) store((k + 1) & 1, t - 1);
                }
                __syncthreads();
            }
        }
    }
    if (!chain) store((ntiles - 1) & 1, ntiles - 1);
}

// ------------------------------------------------------------------ seed map for REPLACING_SOME
// selectGoodFeatures.py:64-69: every live feature blocks the square of half-size d around (int(x), int(y))
// The map is stamped, not cleared: a pixel is blocked when it carries THIS selection's stamp (1..255; the host clears the map when the
// stamps wrap or the frame size changes), so that a replacement pass does not begin by zeroing a frame-sized map.
__global__ void seed_fill_kernel(const klt_feat *__restrict__ fl, int nfeat, uint8_t *__restrict__ seedmap,
                                 int ncols, int nrows, int d, uint8_t stamp)
{
    const int f = blockIdx.x;
    if (f >= nfeat) return;
    const klt_feat ft = fl[f];
    if (ft.val < 0) return;
    const int cx = (int)ft.x, cy = (int)ft.y, side = 2 * d + 1;
    for (int k = threadIdx.x; k < side * side; k += blockDim.x) {
        const int ix = cx - d + k % side, iy = cy - d + k / side;
        if (ix >= 0 && ix < ncols && iy >= 0 && iy < nrows) seedmap[(size_t)iy * ncols + ix] = stamp;
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

// key of candidate k (0 = not a candidate); also stores the eigenvalue
__device__ __forceinline__ unsigned long long eigen_key(const SelectArgs &a, int k)
{
    const int xi = k % a.nx, yi = k / a.nx;
    const int x = a.bx + xi * a.step, y = a.by + yi * a.step;
    // REPLACING_SOME: a pixel inside the exclusion square of a live feature can never be placed (selectGoodFeatures.py:64-69 marks
    // the feature map before the walk), so it is not scored at all -- most of the frame when few features were lost
    if (a.seedmap && a.seedmap[(size_t)y * a.ncols + x] == a.seed_stamp) { if (a.valmap) a.valmap[k] = 0.f; return 0ull; }
    const size_t plane = (size_t)a.ncols * a.nrows;
    const float gxx = window_sum(a.sat, a.ncols, x, y, a.hw, a.hh);
    const float gxy = window_sum(a.sat + plane, a.ncols, x, y, a.hw, a.hh);
    const float gyy = window_sum(a.sat + 2 * plane, a.ncols, x, y, a.hw, a.hh);
    float val;
    unsigned long long key = klt_window_key(gxx, gxy, gyy, a.min_eig, x, y, &val);       // goodFeaturesUtils.pyx:17-19 (klt_internal.h)
    if (a.val_in) {                                                                      // test hook: the eigenvalue is given
        val = a.val_in[k];
        key = (double)val >= a.min_eig ? (((unsigned long long)__float_as_uint(val) << 32) | ((unsigned long long)x << 16) | (unsigned long long)y) : 0ull;
    }
    if (a.valmap) a.valmap[k] = val;
    return key;
}

__global__ __launch_bounds__(256) void eigen_kernel(SelectArgs a)
{
    const int k = blockIdx.x * 256 + threadIdx.x;
    if (k >= a.npow2) return;
    a.keys[k] = k < a.nx * a.ny ? eigen_key(a, k) : 0ull;     // padding sorts last
}

// ------------------------------------------------------------------ top-K prefilter for the sort
// The greedy pass only ever looks at a prefix of the sorted candidates (197 775 of 1 411 200 at 1080p for 5000
// features).  A 8192-bin histogram over the top 13 bits of the (positive) f32 eigenvalue picks the smallest bin
// boundary that keeps at least `target` candidates; only those are compacted and sorted.  Order inside the kept set
// is unchanged (same keys), so the walk sees exactly the same prefix; if it ever runs off the end of the kept set
// before the list is full, the host repeats the selection with the full sort.
constexpr int HIST_BINS = 8192;
__device__ __forceinline__ unsigned key_bin(unsigned long long key) { return (unsigned)(key >> 50) & (HIST_BINS - 1); }

// stride > 1: every stride-th key only (an estimate is enough where the threshold is just a work-saving cut)
__global__ __launch_bounds__(256) void key_hist_kernel(const unsigned long long *__restrict__ keys, int n, unsigned *__restrict__ hist, int stride)
{
    __shared__ unsigned h[HIST_BINS];
    for (int i = threadIdx.x; i < HIST_BINS; i += 256) h[i] = 0u;
    __syncthreads();
    for (int i = (blockIdx.x * 256 + threadIdx.x) * stride; i < n; i += gridDim.x * 256 * stride) {
        const unsigned long long key = keys[i];
        if (key) atomicAdd(&h[key_bin(key)], 1u);
    }
    __syncthreads();
    for (int i = threadIdx.x; i < HIST_BINS; i += 256)
        if (h[i]) atomicAdd(&hist[i], h[i]);
}

// info[0] = threshold bin, info[1] = number of keys in bins >= threshold, info[2] = number of valid keys
// slots != nullptr: target = max(target, per_slot * *slots) -- REPLACING_SOME sizes the cut by the number of lost features, which
// only the device knows (mis_prepare_kernel)
__global__ __launch_bounds__(1024) void key_threshold_kernel(const unsigned *__restrict__ hist, unsigned target, unsigned *__restrict__ info,
                                                             const int *__restrict__ slots, unsigned per_slot)
{
    __shared__ unsigned suf[1025];
    const int t = threadIdx.x;
    if (slots) {
        const unsigned want = per_slot * (unsigned)max(*slots, 0);
        target = want > target ? want : target;
    }
    unsigned mine = 0;
    for (int b = 0; b < 8; b++) mine += hist[8 * t + b];
    suf[t] = mine;
    if (t == 0) suf[1024] = 0u;
    __syncthreads();
    for (int off = 1; off < 1024; off <<= 1) {             // inclusive suffix scan: suf[t] = sum of parts t..1023
        const unsigned v = t + off < 1024 ? suf[t + off] : 0u;
        __syncthreads();
        suf[t] += v;
        __syncthreads();
    }
    const unsigned total = suf[0];
    if (total < target) {
        if (t == 0) { info[0] = 0u; info[1] = total; info[2] = total; }
        return;
    }
    if (suf[t] >= target && suf[t + 1] < target) {         // the boundary lies inside this thread's 8 bins
        unsigned acc = suf[t + 1];
        int b = 8 * t + 7;
        for (; b > 8 * t; b--) {
            acc += hist[b];
            if (acc >= target) break;
        }
        if (b == 8 * t) acc = suf[t];
        info[0] = (unsigned)b; info[1] = acc; info[2] = total;
    }
}

// Eigenvalues, keys, and -- for the parallel minimum-distance passes -- the histogram behind the prefilter threshold in the
// same launch: every 4th workgroup histograms its keys (an estimate is enough: the threshold only saves work, any value
// is correct); key_threshold_kernel turns it into the threshold bin.  a.hist == nullptr: keys only.
__global__ __launch_bounds__(256) void eigen_hist_kernel(SelectArgs a)
{
    __shared__ unsigned h[HIST_BINS];
    const int tid = threadIdx.x, ncand = a.nx * a.ny;
    // (XCD-contiguous eighths of the candidates instead of this interleaved sweep -- the table rows shared by windows 2 hh + 1 rows
    // apart then meet in one L2 -- were measured in round 2: no change, 14.6 us at 1080p either way)
    const bool sampler = a.hist != nullptr && (blockIdx.x & 3) == 0;
    if (sampler) {
        for (int i = tid; i < HIST_BINS; i += 256) h[i] = 0u;
        __syncthreads();
    }
    for (int k = blockIdx.x * 256 + tid; k < ncand; k += gridDim.x * 256) {
        const unsigned long long key = eigen_key(a, k);
        a.keys[k] = key;
        if (sampler && key) atomicAdd(&h[key_bin(key)], 1u);
    }
    if (!a.hist) return;
    if (sampler) {
        __syncthreads();
        for (int i = tid; i < HIST_BINS; i += 256)
            if (h[i]) atomicAdd(&a.hist[i], h[i]);
    }
}

// Keys scored ahead of time without the seed map (klt_select_prepare_async): the histogram behind the cut counts the keys outside
// the live features' squares, on a sample of every 16th block of 256 keys (the cut only saves work; its targets are scaled to the
// sample).  A workgroup takes eight sampled blocks in one step: the seed-map loads, then the key loads of the cells outside the squares,
// are in flight together, and only ~250 workgroups flush their bins (one block per step and every 4th block, as eigen_hist_kernel
// samples, read 29-35 us for a 4K frame: two dependent round trips per block, then 1024 workgroups adding to the same few hundred words).
constexpr int MASK_HIST_SAMPLE = 16;
__global__ __launch_bounds__(256) void mask_hist_kernel(SelectArgs a)
{
    __shared__ unsigned h[HIST_BINS];
    const int tid = threadIdx.x, ncand = a.nx * a.ny;
    for (int i = tid; i < HIST_BINS; i += 256) h[i] = 0u;
    __syncthreads();
    constexpr int U = 8, S = MASK_HIST_SAMPLE;
    for (int k0 = S * U * blockIdx.x * 256 + tid; k0 < ncand; k0 += S * U * gridDim.x * 256) {
        unsigned long long key[U];
        uint8_t masked[U];
#pragma unroll
        for (int u = 0; u < U; u++) {
            const int k = k0 + S * u * 256, xi = k % a.nx, yi = k / a.nx;
            masked[u] = k < ncand ? (uint8_t)(a.seedmap[(size_t)(a.by + yi * a.step) * a.ncols + a.bx + xi * a.step] == a.seed_stamp) : (uint8_t)1;
        }
#pragma unroll
        for (int u = 0; u < U; u++) {
            key[u] = masked[u] ? 0ull : a.keys[k0 + S * u * 256];
            if (!key[u]) masked[u] = 1;
        }
#pragma unroll
        for (int u = 0; u < U; u++)
            if (!masked[u]) atomicAdd(&h[key_bin(key[u])], 1u);
    }
    __syncthreads();
    for (int i = tid; i < HIST_BINS; i += 256)
        if (h[i]) atomicAdd(&a.hist[i], h[i]);
}

// Order inside the kept set does not matter (it is sorted next), so every workgroup counts the keys it keeps in its
// contiguous chunk, reserves a range with ONE global atomic (a returning atomic on a single word sustains only ~90 per
// microsecond chip-wide), and fills the range using an LDS cursor.
__global__ __launch_bounds__(256) void key_compact_kernel(const unsigned long long *__restrict__ keys, int n,
                                                           const unsigned *__restrict__ info, unsigned long long *__restrict__ out,
                                                           unsigned *__restrict__ counter)
{
    __shared__ unsigned s_count, s_base, s_cursor;
    const unsigned thr = info[0];
    const int lane = threadIdx.x & 63;
    const int chunk = ((n + gridDim.x - 1) / gridDim.x + 255) & ~255;
    const int begin = blockIdx.x * chunk, end = min(n, begin + chunk);
    if (threadIdx.x == 0) { s_count = 0u; s_cursor = 0u; }
    __syncthreads();
    unsigned mine = 0;
    for (int i = begin + threadIdx.x; i < end; i += 256) {
        const unsigned long long key = keys[i];
        mine += (key != 0ull && key_bin(key) >= thr) ? 1u : 0u;
    }
    for (int m = 32; m >= 1; m >>= 1) mine += __shfl_xor(mine, m);
    if (lane == 0 && mine) atomicAdd(&s_count, mine);
    __syncthreads();
    if (threadIdx.x == 0) s_base = s_count ? atomicAdd(counter, s_count) : 0u;
    __syncthreads();
    if (s_count == 0u) return;
    const unsigned base = s_base;
    for (int i0 = begin; i0 < end; i0 += 256) {
        const int i = i0 + threadIdx.x;
        const unsigned long long key = i < end ? keys[i] : 0ull;
        const bool keep = key != 0ull && key_bin(key) >= thr;
        const unsigned long long m = __ballot(keep);
        unsigned wbase = 0;
        if (lane == 0 && m) wbase = atomicAdd(&s_cursor, (unsigned)__popcll(m));
        wbase = __shfl(wbase, 0);
        if (keep) out[base + wbase + __popcll(m & ((1ull << lane) - 1ull))] = key;
    }
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
// One workgroup of 16 wavefronts walks the sorted keys 1024 at a time:
//   1. every thread tests its candidate against the cell grid as it stood at the start of the super-batch
//      (almost every candidate dies here: it is blocked by a feature accepted long before);
//   2. the survivors are compacted in rank order;
//   3. wave 0 resolves them sequentially with ballots -- a survivor can only be blocked by a feature accepted
//      inside the same super-batch, which is a register compare (plus a grid re-test for survivors beyond the
//      first 64, whose predecessors in the super-batch are already in the grid).
// The result is exactly the sequential walk of selectGoodFeatures.py:71-134.
constexpr int NMS_T = 1024;

template <bool LDSGRID>
__device__ __forceinline__ uint32_t grid_load(const uint32_t *p)
{
    if (LDSGRID) return *p;
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);     // L2-served: never a stale L1 line
}

// x / cell for x < 65536 without an integer division: cell_magic = floor(2^32 / cell) + 1
__device__ __forceinline__ int cell_of(int v, const NmsArgs &a)
{
    return a.cell == 1 ? v : (int)__umulhi((unsigned)v, a.cell_magic);
}

template <bool LDSGRID>
__device__ __forceinline__ bool grid_free(const uint32_t *grid, const NmsArgs &a, int x, int y)
{
    const int cx = cell_of(x, a), cy = cell_of(y, a);
    bool ok = true;
#pragma unroll
    for (int dy = -1; dy <= 1; dy++)
#pragma unroll
        for (int dx = -1; dx <= 1; dx++) {
            const int gx = cx + dx, gy = cy + dy;
            if (gx < 0 || gy < 0 || gx >= a.gw || gy >= a.gh) continue;
            const uint32_t v = grid_load<LDSGRID>(&grid[gy * a.gw + gx]);
            if (v) {
                const int ax = (int)((v - 1u) >> 16), ay = (int)((v - 1u) & 0xffffu);
                if (abs(x - ax) <= a.d && abs(y - ay) <= a.d) ok = false;
            }
        }
    return ok;
}

template <bool LDSGRID>
__global__ __launch_bounds__(NMS_T) void nms_kernel(NmsArgs a)
{
    extern __shared__ uint32_t lds_grid[];
    __shared__ unsigned long long surv[NMS_T];
    __shared__ int wave_cnt[NMS_T / 64];
    __shared__ int scan[NMS_T];
    __shared__ int s_stop, s_nfill;
    uint32_t *grid = LDSGRID ? lds_grid : a.grid_global;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (LDSGRID)
        for (int i = tid; i < a.gw * a.gh; i += NMS_T) grid[i] = 0u;
    if (tid == 0) s_stop = 0;

    // slots that may be filled, in list order (selectGoodFeatures.py:109-110): every slot when overwriting,
    // otherwise the slots whose feature is lost.  a.slots[k] = index of the k-th fillable slot.
    int nfill = a.nfeat;
    if (!a.overwrite_all) {
        const int per = (a.nfeat + NMS_T - 1) / NMS_T, lo = tid * per, hi = min(lo + per, a.nfeat);
        int cnt = 0;
        for (int i = lo; i < hi; i++) cnt += a.fl[i].val < 0;
        scan[tid] = cnt;
        __syncthreads();
        for (int off = 1; off < NMS_T; off <<= 1) {          // Hillis-Steele inclusive scan
            const int v = tid >= off ? scan[tid - off] : 0;
            __syncthreads();
            scan[tid] += v;
            __syncthreads();
        }
        int k = scan[tid] - cnt;
        for (int i = lo; i < hi; i++)
            if (a.fl[i].val < 0) a.slots[k++] = i;
        if (tid == NMS_T - 1) s_nfill = scan[tid];
        __syncthreads();
        nfill = s_nfill;
    }
    __syncthreads();

    int placed = 0;                               // meaningful in wave 0 only
    bool list_full = nfill == 0;
    unsigned long long key = tid < a.nkeys ? a.keys[tid] : 0ull;
    for (int pos = 0; pos < a.nkeys && nfill > 0; pos += NMS_T) {
        const unsigned long long cur = key;
        const int nxt = pos + NMS_T + tid;
        key = nxt < a.nkeys ? a.keys[nxt] : 0ull;                 // prefetch the next super-batch
        const bool valid = cur != 0ull;
        const int x = (int)((cur >> 16) & 0xffffull), y = (int)(cur & 0xffffull);
        bool ok = valid;
        if (valid && a.d >= 0) ok = grid_free<LDSGRID>(grid, a, x, y);
        const unsigned long long m = __ballot(ok);
        if (lane == 0) wave_cnt[wave] = __popcll(m);
        const int ended = __syncthreads_or(!valid);               // a zero key: everything after it is padding
        int offset = 0, total = 0;
#pragma unroll
        for (int w = 0; w < NMS_T / 64; w++) {
            const int c = wave_cnt[w];
            offset += w < wave ? c : 0;
            total += c;
        }
        if (ok) surv[offset + __popcll(m & ((1ull << lane) - 1ull))] = cur;
        __syncthreads();
        if (wave == 0) {
            for (int base = 0; base < total && !list_full; base += 64) {
                const unsigned long long k = base + lane < total ? surv[base + lane] : 0ull;
                const int sx = (int)((k >> 16) & 0xffffull), sy = (int)(k & 0xffffull);
                bool is_free = k != 0ull;
                if (base > 0 && is_free && a.d >= 0) is_free = grid_free<LDSGRID>(grid, a, sx, sy);
                unsigned long long mask = __ballot(is_free), accepted = 0ull;
                int room = nfill - placed;
                if (a.d < 0) {                   // nothing excludes anything: the first `room` free lanes are accepted
                    const bool take = ((mask >> lane) & 1ull) && __popcll(mask & ((1ull << lane) - 1ull)) < room;
                    accepted = __ballot(take);
                    mask = 0ull;
                }
                // rank order = lane order: the first free lane is accepted and blocks the later lanes near it
                while (mask != 0ull && room > 0) {
                    const int l = __ffsll((long long)mask) - 1;
                    accepted |= 1ull << l;
                    room--;
                    mask &= ~(1ull << l);
                    if (a.d >= 0) {
                        const int ax = __builtin_amdgcn_readlane(sx, l), ay = __builtin_amdgcn_readlane(sy, l);
                        mask &= ~__ballot(abs(sx - ax) <= a.d && abs(sy - ay) <= a.d);
                    }
                }
                if ((accepted >> lane) & 1ull) {
                    const int rank = placed + __popcll(accepted & ((1ull << lane) - 1ull));
                    const int slot = a.overwrite_all ? rank : a.slots[rank];
                    klt_feat ft;
                    ft.x = (float)sx;
                    ft.y = (float)sy;
                    ft.val = (int32_t)__uint_as_float((uint32_t)(k >> 32));      // int(val), selectGoodFeatures.py:119
                    ft.aux = 0;
                    a.fl[slot] = ft;
                    if (a.aff_rec) {
                        klt_affine_rec r;
                        r.aff_x = -1.f; r.aff_y = -1.f; r.Axx = 1.f; r.Ayx = 0.f; r.Axy = 0.f; r.Ayy = 1.f; r.valid = 0; r.pad = 0;
                        a.aff_rec[slot] = r;
                    }
                    if (a.d >= 0) {
                        const uint32_t code = (((uint32_t)sx << 16) | (uint32_t)sy) + 1u;
                        uint32_t *cellp = &grid[cell_of(sy, a) * a.gw + cell_of(sx, a)];
                        if (LDSGRID) *cellp = code;
                        else __hip_atomic_store(cellp, code, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    }
                }
                placed += __popcll(accepted);
                if (placed >= nfill) list_full = true;                            // :112
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            }
            if (lane == 0 && (list_full || ended)) s_stop = 1;
        }
        __syncthreads();
        if (s_stop) break;
    }
    // candidates exhausted: selectGoodFeatures.py:78-94 (SELECTING_ALL only; DESIGN.md lists the deviation)
    if (wave == 0) {
        if (!list_full && a.overwrite_all) {
            for (int i = placed + lane; i < a.nfeat; i += 64) {
                klt_feat ft;
                ft.x = -1.f;
                ft.y = -1.f;
                ft.val = KLT_NOT_FOUND;
                ft.aux = 0;
                a.fl[i] = ft;
            }
        }
        if (lane == 0 && a.placed_out) {
            a.placed_out[0] = placed;
            a.placed_out[1] = list_full ? 0 : 1;      // 1: the walk ran out of candidates before the list was full
        }
    }
}

// ------------------------------------------------------------------ minimum distance without the serial walk
// _enforceMinimumDistance (selectGoodFeatures.py:45-135) accepts a candidate iff no candidate of higher rank within
// the exclusion square was accepted.  That fixed point is unique, so it can be reached in any order: a candidate is
// ACCEPTED as soon as every higher-ranked candidate within the square is rejected, and REJECTED as soon as one
// accepted candidate lies within the square.  Every decision taken is final and equals the sequential walk's; stale
// reads of a neighbour (still "undecided" although it has just been decided) only postpone a decision.  One launch =
// one such pass over the undecided candidates; 6 passes settle 320 000 candidates of a 1080p frame.  The accepted
// candidates are then sorted by rank and the first `free slots` of them are exactly what the walk would have placed.
//
// State per candidate cell (u32): 0 = no candidate / rejected, f32 bits of the eigenvalue (>= 1.0f) = undecided,
// bits | 0x80000000 = accepted.  Rank order = (value, x, y) descending, so for equal values the neighbour to the right
// wins, and in the same column the neighbour below.
// Each workgroup owns one 32x32 tile of the candidate grid and the list of its undecided cells (compacted in place).
constexpr int MIS_TILE = 32, MIS_CAP = MIS_TILE * MIS_TILE, MIS_T = 256;

// Workgroup -> tile.  Workgroup b runs on XCD b % 8: XCD k takes the k-th contiguous eighth of the (row-major) tiles, so that the
// halo cells a tile shares with its neighbours come out of the same L2.
__device__ __forceinline__ unsigned xcd_tile(unsigned b, unsigned g)
{
    const unsigned k = b & 7u, i = b >> 3, base = g >> 3, rem = g & 7u;
    return k * base + min(k, rem) + i;
}

__global__ __launch_bounds__(MIS_T) void mis_init_kernel(MisArgs a)
{
    __shared__ unsigned s_cursor;
    const int tiles_x = (a.nx + MIS_TILE - 1) / MIS_TILE;
    const unsigned tile = xcd_tile(blockIdx.x, gridDim.x);
    const int tx = tile % tiles_x, ty = tile / tiles_x;
    const unsigned thr = a.info[0];
    const int lane = threadIdx.x & 63;
    if (threadIdx.x == 0) s_cursor = 0u;
    __syncthreads();
    uint32_t *list = a.list + (size_t)tile * MIS_CAP;
    // the four cells of a thread: seed-map bytes first (keys scored ahead of time carry no mask), then the keys of the cells outside
    // the live features' squares, each group of loads in flight together
    constexpr int U = MIS_CAP / MIS_T;
    bool inside[U];
    int pcell[U];
    unsigned long long key[U];
#pragma unroll
    for (int u = 0; u < U; u++) {
        const int k = u * MIS_T + threadIdx.x;
        const int xi = tx * MIS_TILE + (k & (MIS_TILE - 1)), yi = ty * MIS_TILE + k / MIS_TILE;
        inside[u] = xi < a.nx && yi < a.ny;
        pcell[u] = yi * a.nx + xi;
        key[u] = 1ull;
        if (inside[u] && a.seed) key[u] = a.seed[(size_t)(a.by + yi * a.step) * a.ncols + a.bx + xi * a.step] == a.seed_stamp ? 0ull : 1ull;
    }
#pragma unroll
    for (int u = 0; u < U; u++) key[u] = inside[u] && key[u] ? a.keys[pcell[u]] : 0ull;
#pragma unroll
    for (int u = 0; u < U; u++) {
        const int p = pcell[u];
        const bool keep = key[u] != 0ull && key_bin(key[u]) >= thr;
        if (inside[u]) a.st[p] = keep ? (uint32_t)(key[u] >> 32) : 0u;
        const unsigned long long m = __ballot(keep);
        unsigned wbase = 0;
        if (lane == 0 && m) wbase = atomicAdd(&s_cursor, (unsigned)__popcll(m));
        wbase = __shfl(wbase, 0);
        if (keep) list[wbase + __popcll(m & ((1ull << lane) - 1ull))] = (uint32_t)p;
    }
    __syncthreads();
    if (threadIdx.x == 0) a.cnt[tile] = s_cursor;
}

// (Several passes per launch -- every tile looping over re-staged states read with agent-scope loads, so that decisions of
// other XCDs are seen inside the launch -- were measured in round 2: 2 launches x 41 us instead of 6 x 13 us at 1080p, i.e. the
// same; at 4K, where the 5700 tiles are not all resident, the early tiles spin on neighbours that have not started: 218 vs 99 us.
// A pass costs what it computes (staging + window maxima of every active tile, ~4.6 M wavefront-instructions), not its launch.)
// Timeline of one workgroup in the first pass (1080p cfg-2 frame, 1431 tiles, ~220 candidates each; medians, wall clock read by
// thread 0 at the stage boundaries): list length 0.4 us, staging the tile 4.1, window maxima 1.3, decisions 2.2, tie checks 1.7, push
// 1.0, keys 0.3 -- 11 us, and 16.8 us from the first workgroup's start to the last one's end; in the fifth pass (145 tiles, ~3
// candidates) 0.4 / 2.0 / 0.8 / 0.9 / 0.2 / 0.1 / 0.1 -- 4.4 us in a 7.4 us launch.  Staging is state-map reads that missed the L2
// (every kernel boundary invalidates it): hence the XCD-contiguous tile order below (-4 % on a 4K frame).
// RC > 0: the exclusion radius as a compile-time constant (the reference's default mindist 10 -> 9 cells: tile geometry, window
// loops and the index arithmetic fold); RC = 0: any radius
template <int RC>
__device__ __forceinline__ void mis_round_tile(const MisArgs &a, int round, unsigned tile, unsigned n)
{
    // staged tiles: S = states of the tile + halo ((32 + 2R)^2 u32), H = per row of S and interior column the maximum
    // state over the horizontal window; a candidate then needs the 2R + 1 values of H above and below it.  The maximum
    // carries the accepted bit.  States compare like the rank except for equal eigenvalues: a candidate whose value
    // equals the window maximum scans the window for an equal neighbour of higher rank (right of it, or below it in
    // the same column).
    extern __shared__ uint32_t lds32[];
    __shared__ unsigned short top_local[MIS_CAP], wait_local[MIS_CAP];
    __shared__ unsigned s_cursor, s_acc, s_top, s_left;
    const unsigned have = a.acc_cnt[tile];             // accepted candidates of this tile so far
    const int tiles_x = (a.nx + MIS_TILE - 1) / MIS_TILE;
    const int tx = tile % tiles_x, ty = tile / tiles_x;
    const int R = RC > 0 ? RC : a.R, W = MIS_TILE + 2 * R, L = 2 * R + 1;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    // (Deciding the last few candidates of a tile without staging it -- a wavefront reading each candidate's window straight from
    // the states -- was measured in round 2: every pass got slower, 11.3 vs 7.4 us for the late ones; the dependent L2 round trips
    // per candidate cost more than staging the tile once.)
    const bool staged = a.stage && R > 0;
    uint32_t *H = lds32;                      // [W][32]
    uint32_t *S = lds32 + W * MIS_TILE;       // [W][W]
    // candidates accepted in this pass: at most a.acc_cap per tile (pairwise more than R cells apart), behind the staged tiles
    // (eight workgroups per CU instead of seven -- 2048 slots for the tiles of a 1080p frame -- were measured this way: no change)
    unsigned short *acc_local = reinterpret_cast<unsigned short *>(lds32 + (staged ? W * MIS_TILE + W * W : 0));
    const int ox = tx * MIS_TILE - R, oy = ty * MIS_TILE - R;
    uint32_t *list = a.list + (size_t)tile * MIS_CAP;
    unsigned next_p = threadIdx.x < n ? list[threadIdx.x] : 0u;           // first batch, in flight during the staging
    if (threadIdx.x == 0) { s_acc = 0u; s_cursor = 0u; s_top = 0u; s_left = 0u; }
    // tile + halo inside the frame (all but the tiles along its edges): no test per cell -- column test once per lane, row test on the
    // wavefront's (scalar) number; the general loop below spends a dozen instructions per load on them
    const bool interior = staged && ox >= 0 && oy >= 0 && ox + W <= a.nx && oy + W <= a.ny;
    if (interior) {
        const int wave_u = __builtin_amdgcn_readfirstlane(wave);
        for (int cx0 = 0; cx0 < W; cx0 += 64) {
            const int cx = cx0 + lane;
            if (cx < W) {
                const uint32_t *col = a.st + ((unsigned)oy * (unsigned)a.nx + (unsigned)(ox + cx));
                for (int r0 = 0; r0 < W; r0 += 64) {
                    uint32_t v[16];
#pragma unroll
                    for (int u = 0; u < 16; u++) {
                        const int r = min(r0 + wave_u + 4 * u, W - 1);          // (rows past the stage repeat its last one; not written)
                        v[u] = col[(unsigned)r * (unsigned)a.nx];
                    }
#pragma unroll
                    for (int u = 0; u < 16; u++) {
                        const int r = r0 + wave_u + 4 * u;
                        if (r < W) S[r * W + cx] = v[u];
                    }
                }
            }
        }
    } else if (staged) {
        for (int cx0 = 0; cx0 < W; cx0 += 64)
            for (int r0 = 0; r0 < W; r0 += 64) {            // 16 rows per wavefront in flight at once
                const int cx = cx0 + lane, gx = ox + cx;
                uint32_t v[16];
#pragma unroll
                for (int u = 0; u < 16; u++) {
                    const int r = r0 + wave + 4 * u, gy = oy + r;
                    v[u] = (cx < W && r < W && gx >= 0 && gx < a.nx && gy >= 0 && gy < a.ny) ? a.st[(size_t)gy * a.nx + gx] : 0u;
                }
#pragma unroll
                for (int u = 0; u < 16; u++) {
                    const int r = r0 + wave + 4 * u;
                    if (cx < W && r < W) S[r * W + cx] = v[u];
                }
            }
    }
    if (staged) {
        __syncthreads();
        // H[r][c] = max of S[r][c .. c + 2R].  A thread makes 8 adjacent outputs of a row: their windows share the columns
        // c0 + 7 .. c0 + L - 1 (one running maximum), output j adds the last 7 - j columns before and the first j columns after that
        // range -- L + 7 LDS reads for 8 outputs instead of 8 L (the passes are bound by LDS traffic: this loop was 30400 of the
        // ~37000 LDS accesses a tile costs).
        if (L >= 8) {
            for (int it = threadIdx.x; it < W * (MIS_TILE / 8); it += MIS_T) {
                const int r = it % W, c0 = (it / W) * 8;
                const uint32_t *row = S + r * W + c0;
                uint32_t lo[7], hi[7];
#pragma unroll
                for (int j = 0; j < 7; j++) { lo[j] = row[j]; hi[j] = row[L + j]; }
                uint32_t common = row[7];
#pragma unroll 4
                for (int k = 8; k < L; k++) common = max(common, row[k]);
                uint32_t o[8];
                o[7] = common;
#pragma unroll
                for (int j = 6; j >= 0; j--) o[j] = max(o[j + 1], lo[j]);           // + columns c0 + j .. c0 + 6
                uint32_t m = 0u;
#pragma unroll
                for (int j = 1; j < 8; j++) { m = max(m, hi[j - 1]); o[j] = max(o[j], m); }      // + columns c0 + L .. c0 + L + j - 1
                uint4 *dst = reinterpret_cast<uint4 *>(H + r * MIS_TILE + c0);
                dst[0] = make_uint4(o[0], o[1], o[2], o[3]);
                dst[1] = make_uint4(o[4], o[5], o[6], o[7]);
            }
        } else
        for (int k = threadIdx.x; k < W * MIS_TILE; k += MIS_T) {
            const uint32_t *row = &S[(k / MIS_TILE) * W + (k % MIS_TILE)];     // window = columns c .. c + 2R of the row
            uint32_t m = row[0];
#pragma unroll 8
            for (int dx = 1; dx <= 2 * R; dx++) m = max(m, row[dx]);
            H[k] = m;
        }
    }
    __syncthreads();
    for (unsigned base = 0; base < n; base += MIS_T) {
        const unsigned i = base + threadIdx.x;
        const bool valid = i < n;
        const int p = (int)next_p;
        if (base + MIS_T + threadIdx.x < n) next_p = list[base + MIS_T + threadIdx.x];
        const int xi = p % a.nx, yi = p / a.nx;
        const int lx = xi - tx * MIS_TILE, ly = yi - ty * MIS_TILE;              // interior coordinates
        bool near_accepted = false, blocked = false, gone = false, top = false;
        uint32_t sp = 0u;
        if (valid && staged) {
            sp = S[(ly + R) * W + lx + R];
            gone = sp == 0u;                                                     // rejected by a neighbour's push
            uint32_t m = 0u;
            const uint32_t *col = &H[ly * MIS_TILE + lx];                        // rows ly .. ly + 2R of H
#pragma unroll 8
            for (int dy = 0; dy <= 2 * R; dy++) m = max(m, col[dy * MIS_TILE]);
            near_accepted = (m >> 31) != 0u;
            blocked = m > sp;
            top = !gone && m == sp;              // holds the window maximum: accepted unless an equal neighbour outranks it
        } else if (valid) {
            sp = a.st[p];
            gone = sp == 0u;
            const int x0 = max(xi - R, 0), x1 = min(xi + R, a.nx - 1), y0 = max(yi - R, 0), y1 = min(yi + R, a.ny - 1);
            for (int yy = y0; yy <= y1 && !near_accepted && !gone; yy++) {
                const uint32_t *row = a.st + (size_t)yy * a.nx;
                for (int xx = x0; xx <= x1; xx++) {
                    const uint32_t sv = row[xx];
                    near_accepted |= (sv >> 31) != 0u;
                    blocked |= sv > sp || (sv == sp && (xx > xi || (xx == xi && yy > yi)));
                }
            }
        }
        const bool live = valid && !gone;
        const bool reject = live && near_accepted;
        const bool wait = live && !near_accepted && blocked;
        const bool accept = live && !near_accepted && !blocked && !top;
        const unsigned short q = (unsigned short)(ly * MIS_TILE + lx);
        if (reject) a.st[p] = 0u;
        if (top) top_local[atomicAdd(&s_top, 1u)] = q;       // decided after the loop
        if (accept) {
            a.st[p] = sp | 0x80000000u;
            acc_local[atomicAdd(&s_acc, 1u)] = q;
        }
        // the candidates that stay undecided are collected in LDS and go back to the list once this pass's accepted ones are known
        const unsigned long long m = __ballot(wait);
        unsigned wbase = 0;
        if (lane == 0 && m) wbase = atomicAdd(&s_cursor, (unsigned)__popcll(m));
        wbase = __shfl(wbase, 0);
        if (wait) wait_local[wbase + __popcll(m & ((1ull << lane) - 1ull))] = q;
    }
    __syncthreads();
    const unsigned ntop = s_top;
    if (ntop) {
        // equal eigenvalues inside one window are ranked by position: a wavefront per candidate scans its window for an equal
        // neighbour of higher rank (one candidate after the other with the whole workgroup, and a barrier each, before: 1.7 of
        // the 11 us a dense tile lives)
        const unsigned magic = (unsigned)((1ull << 32) / (unsigned)L) + 1u;      // w / L for w < 65536
        for (unsigned e = (unsigned)wave; e < ntop; e += MIS_T / 64) {
            const int q = (int)top_local[e], lx = q % MIS_TILE, ly = q / MIS_TILE;
            const uint32_t sp = S[(ly + R) * W + lx + R];
            bool outranked = false;
            for (int w = lane; w < L * L; w += 64) {
                const int wy = (int)__umulhi((unsigned)w, magic), dx = w - wy * L - R, dy = wy - R;
                outranked |= S[(ly + R + dy) * W + lx + R + dx] == sp && (dx > 0 || (dx == 0 && dy > 0));
            }
            const bool any = __ballot(outranked) != 0ull;
            if (lane == 0) {
                if (any) {
                    wait_local[atomicAdd(&s_cursor, 1u)] = (unsigned short)q;
                } else {
                    a.st[(ty * MIS_TILE + ly) * a.nx + tx * MIS_TILE + lx] = sp | 0x80000000u;
                    acc_local[atomicAdd(&s_acc, 1u)] = (unsigned short)q;
                }
            }
        }
        __syncthreads();
    }
    // no atomics on shared words here: a word that every workgroup increments serialises the whole launch (~90 returning
    // atomics per microsecond chip-wide).  "Something is left" is a plain store of 1; the accepted candidates go to the
    // tile's own slots and are compacted once, after the last pass.
    unsigned nacc = s_acc;
    const unsigned nwait = s_cursor;
    if (have + nacc > (unsigned)a.acc_cap) nacc = (unsigned)a.acc_cap - have;      // cannot happen (one per (R+1)^2 cells)
    // back to the list: the undecided candidates outside the squares of the candidates accepted in this pass.  (The push below
    // clears the squares in the state map; a candidate of this tile inside one used to stay listed until the next pass found it
    // gone -- after the first pass that was 85 % of the list.)
    for (unsigned i0 = 0; i0 < nwait; i0 += MIS_T) {
        const unsigned i = i0 + threadIdx.x;
        const int q = i < nwait ? (int)wait_local[i] : 0, lx = q % MIS_TILE, ly = q / MIS_TILE;
        bool keep = i < nwait;
        if (R > 0)
            for (unsigned k = 0; k < nacc; k++) {
                const int qa = (int)acc_local[k];
                keep = keep && (abs(lx - qa % MIS_TILE) > R || abs(ly - qa / MIS_TILE) > R);
            }
        const unsigned long long m = __ballot(keep);
        unsigned wbase = 0;
        if (lane == 0 && m) wbase = atomicAdd(&s_left, (unsigned)__popcll(m));
        wbase = __shfl(wbase, 0);
        if (keep) list[wbase + __popcll(m & ((1ull << lane) - 1ull))] = (uint32_t)((ty * MIS_TILE + ly) * a.nx + tx * MIS_TILE + lx);
    }
    __syncthreads();
    const unsigned left = s_left;
    if (threadIdx.x == 0) {
        a.cnt[tile] = left;
        if (left) a.remaining[round] = 1u;
        a.acc_cnt[tile] = have + nacc;
    }
    const size_t s_base = (size_t)tile * a.acc_cap + have;
    // push: every cell within the exclusion square of a newly accepted candidate is rejected right away.  None of them
    // can be accepted (two accepted candidates never lie within each other's square), so the whole square is cleared.
    if (R > 0 && nacc) {
        const unsigned magic = (unsigned)((1ull << 32) / (unsigned)L) + 1u;      // w / L for w < 65536
        for (int w = threadIdx.x; w < L * L; w += MIS_T) {
            const int wy = (int)__umulhi((unsigned)w, magic), dx = w - wy * L - R, dy = wy - R;
            if ((dx | dy) == 0) continue;
            for (unsigned k = 0; k < nacc; k++) {
                const int q = (int)acc_local[k];
                const int gx = tx * MIS_TILE + (q % MIS_TILE) + dx, gy = ty * MIS_TILE + q / MIS_TILE + dy;
                if (staged && S[(q / MIS_TILE + R + dy) * W + q % MIS_TILE + R + dx] == 0u) continue;
                if (gx >= 0 && gy >= 0 && gx < a.nx && gy < a.ny) a.st[(size_t)gy * a.nx + gx] = 0u;
            }
        }
    }
    __syncthreads();
    for (unsigned i = threadIdx.x; i < nacc; i += MIS_T) {
        const int q = (int)acc_local[i];
        const int xi = tx * MIS_TILE + q % MIS_TILE, yi = ty * MIS_TILE + q / MIS_TILE;
        const uint32_t sp = staged ? S[(q / MIS_TILE + R) * W + q % MIS_TILE + R] : a.st[(size_t)yi * a.nx + xi] & 0x7fffffffu;
        const unsigned long long key = ((unsigned long long)sp << 32) | ((unsigned long long)(a.bx + xi * a.step) << 16) |
                                       (unsigned long long)(a.by + yi * a.step);
        a.acc_keys[s_base + i] = key;
    }
}

// One workgroup = TPW consecutive tiles of its XCD's eighth (see xcd_tile), one after the other.  After the first pass most tiles have
// nothing left to decide, and a workgroup whose only act is to find that out costs a launch slot and a round trip to its counter: 6780
// such workgroups (4K) were 5-8 us of every later pass.  Four tiles per workgroup, their counters requested together, make that a
// quarter of the workgroups with one round trip each; the first pass, where almost every tile has work, keeps one tile per workgroup.
constexpr int MIS_TPW_MAX = 4;
template <int RC>
__global__ __launch_bounds__(MIS_T) void mis_round_kernel(MisArgs a, int round, int tpw)
{
    const unsigned ntile = (unsigned)(((a.nx + MIS_TILE - 1) / MIS_TILE) * ((a.ny + MIS_TILE - 1) / MIS_TILE));
    const unsigned k = blockIdx.x & 7u, i = blockIdx.x >> 3, base = ntile >> 3, rem = ntile & 7u;
    const unsigned first = k * base + min(k, rem), len = base + (k < rem ? 1u : 0u);      // XCD k's tiles
    unsigned cnt[MIS_TPW_MAX];
#pragma unroll
    for (int sub = 0; sub < MIS_TPW_MAX; sub++) {
        const unsigned j = i * (unsigned)tpw + (unsigned)sub;
        cnt[sub] = sub < tpw && j < len ? a.cnt[first + j] : 0u;
    }
#pragma unroll
    for (int sub = 0; sub < MIS_TPW_MAX; sub++) {
        if (cnt[sub] == 0u) continue;                                                     // (uniform: the whole workgroup skips the tile)
        mis_round_tile<RC>(a, round, first + i * (unsigned)tpw + (unsigned)sub, cnt[sub]);
        __syncthreads();
    }
}

// block-wide helpers (256 threads): sum, and inclusive scan, of one unsigned per thread
__device__ __forceinline__ unsigned block_sum_256(unsigned v, unsigned *red /* [4] */)
{
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    const unsigned t = red[0] + red[1] + red[2] + red[3];
    __syncthreads();
    return t;
}

__device__ __forceinline__ unsigned block_scan_256(unsigned v, unsigned *red /* [4] */)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    unsigned inc = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const unsigned o = __shfl_up(inc, d);
        if (lane >= d) inc += o;
    }
    if (lane == 63) red[wave] = inc;
    __syncthreads();
    unsigned base = 0;
    for (int w = 0; w < wave; w++) base += red[w];
    __syncthreads();
    return base + inc;
}

// accepted candidates of all tiles -> one dense array (tile order) + their number.  One thread per tile, any number of
// workgroups: a workgroup first adds up the counts of the tiles before its own 256 (a few thousand words from L2), scans its
// own, and every thread copies its tile's keys.  (The single-workgroup version took 10.7 us at 1080p and 30.7 us at 4K.
// Keeping only the keys that can reach a free slot -- a histogram of the accepted keys, the bin that leaves `free slots` above
// it -- would shrink the O(n^2) ranking behind it, but building that histogram costs more than it saves: 1.4 us per pass with
// one global atomic per accepted key, 16 / 37 us with every compaction workgroup histogramming all keys in LDS.)
__global__ __launch_bounds__(256) void mis_compact_kernel(const unsigned long long *__restrict__ tile_keys, const unsigned *__restrict__ acc_cnt,
                                                          int tiles, int cap, unsigned long long *__restrict__ out, unsigned *__restrict__ out_count)
{
    __shared__ unsigned red[4];
    const int tid = threadIdx.x, t0 = blockIdx.x * 256, t = t0 + tid;
    unsigned before = 0;
    {
        // sixteen loads in flight: the last workgroup of a 4K frame walks 22 strides of 256 counts, and a loop that waits for every
        // load in turn made this kernel 22 us long (round-3 kernel trace) for 5 us of work
        int i = tid;
        for (; i + 15 * 256 < t0; i += 16 * 256) {
            unsigned v[16];
#pragma unroll
            for (int u = 0; u < 16; u++) v[u] = acc_cnt[i + u * 256];
#pragma unroll
            for (int u = 0; u < 16; u++) before += v[u];
        }
        unsigned v[16];                                         // the tail: clamped, unconditional loads (a guard is a branch)
#pragma unroll
        for (int u = 0; u < 16; u++) v[u] = acc_cnt[min(i + u * 256, max(t0 - 1, 0))];
#pragma unroll
        for (int u = 0; u < 16; u++) before += i + u * 256 < t0 ? v[u] : 0u;
    }
    const unsigned base = block_sum_256(before, red);
    const unsigned c = t < tiles ? acc_cnt[t] : 0u;
    const unsigned incl = block_scan_256(c, red);
    {                                                                      // thread = tile; 8 loads in flight at a time
        const unsigned long long *src = tile_keys + (size_t)t * cap;
        unsigned long long *dst = out + base + incl - c;
        for (unsigned k0 = 0; k0 < c; k0 += 8) {
            unsigned long long v[8];
#pragma unroll
            for (int u = 0; u < 8; u++) v[u] = k0 + u < c ? src[k0 + u] : 0ull;
#pragma unroll
            for (int u = 0; u < 8; u++)
                if (k0 + u < c) dst[k0 + u] = v[u];
        }
    }
    if (blockIdx.x == gridDim.x - 1 && tid == 255) *out_count = base + incl;
}

// ---- placement of the accepted candidates: rank by counting, then the first `free slots` fill the list
// free slots in list order (selectGoodFeatures.py:109-110): every slot when overwriting, else the lost features
// one launch before the passes: the first workgroups list the free slots and keep a copy of the list (a repeated attempt
// starts from it), the others clear the counters
__global__ __launch_bounds__(256) void mis_prepare_kernel(const klt_feat *__restrict__ fl, int nfeat, int overwrite_all, int *__restrict__ slots,
                                                          int *__restrict__ nfill_out, klt_feat *__restrict__ snapshot,
                                                          unsigned *__restrict__ zero, size_t zero_n, int feat_blocks)
{
    if ((int)blockIdx.x >= feat_blocks) {
        const size_t zb = blockIdx.x - feat_blocks, nz = gridDim.x - feat_blocks;
        for (size_t i = zb * 256 + threadIdx.x; i < zero_n; i += nz * 256) zero[i] = 0u;
        return;
    }
    // workgroup b owns features [256 b, 256 b + 256): copies them, and (REPLACING_SOME) lists the lost ones among them behind
    // the lost ones of all earlier features, which it counts itself
    __shared__ unsigned red[4];
    const int tid = threadIdx.x, f0 = blockIdx.x * 256, f = f0 + tid;
    klt_feat ft;
    ft.val = 0;
    if (f < nfeat) { ft = fl[f]; snapshot[f] = ft; }
    if (overwrite_all) { if (f == 0) *nfill_out = nfeat; return; }
    unsigned before = 0;
    {
        // (eight loads in flight: the last workgroup of a 20 000-feature list walks 78 strides of 256 records)
        int i = tid;
        for (; i + 15 * 256 < f0; i += 16 * 256) {
            int v[16];
#pragma unroll
            for (int u = 0; u < 16; u++) v[u] = fl[i + u * 256].val;
#pragma unroll
            for (int u = 0; u < 16; u++) before += v[u] < 0 ? 1u : 0u;
        }
        {                                                       // the tail as one more batch: clamped, unconditional loads
            int v[16];
#pragma unroll
            for (int u = 0; u < 16; u++) v[u] = fl[min(i + u * 256, max(f0 - 1, 0))].val;
#pragma unroll
            for (int u = 0; u < 16; u++) before += (i + u * 256 < f0 && v[u] < 0) ? 1u : 0u;
        }
    }
    const unsigned base = block_sum_256(before, red);
    const unsigned lost = (f < nfeat && ft.val < 0) ? 1u : 0u;
    const unsigned incl = block_scan_256(lost, red);
    if (lost) slots[base + incl - 1] = f;
    if ((int)blockIdx.x == feat_blocks - 1 && tid == 255) *nfill_out = (int)(base + incl);
}

// the few words the host looks at, written straight into pinned host memory: [0, look) "undecided left" flags of the
// last passes, [64, 68) prefilter info, [72] placed, [73] candidates ran out
__device__ __forceinline__ void write_results(unsigned *host_out, const unsigned *rem, int look, const unsigned *info, int placed, int ranout)
{
    for (int i = threadIdx.x; i < look; i += blockDim.x) host_out[i] = rem[i];
    if (threadIdx.x < 4) host_out[64 + threadIdx.x] = info[threadIdx.x];
    if (threadIdx.x == 0) { host_out[72] = (unsigned)placed; host_out[73] = (unsigned)ranout; }
}

__global__ void mis_results_kernel(unsigned *host_out, const unsigned *rem, int look, const unsigned *info, const int *placed)
{
    write_results(host_out, rem, look, info, placed[0], placed[1]);
}

// rank[i] = number of accepted keys greater than key i (keys are distinct).  The number of accepted keys is only known on the
// device, so the grid is fixed and every workgroup walks the (i-chunk, j-chunk) pairs with its stride: a REPLACING_SOME pass that
// accepted a few hundred candidates does one pair, where a grid sized for the worst case launched 73 000 empty workgroups at 4K.
constexpr int RANK_T = 256, RANK_GX = 96, RANK_GY = 96;
__global__ __launch_bounds__(RANK_T) void mis_rank_kernel(const unsigned long long *__restrict__ keys, const unsigned *__restrict__ count,
                                                          unsigned *__restrict__ rank)
{
    __shared__ unsigned long long other[RANK_T];
    const unsigned n = *count, nchunks = (n + RANK_T - 1) / RANK_T;
    for (unsigned ic = blockIdx.x; ic < nchunks; ic += gridDim.x) {
        const unsigned i = ic * RANK_T + threadIdx.x;
        const unsigned long long key = i < n ? keys[i] : ~0ull;
        unsigned greater = 0;
        for (unsigned jc = blockIdx.y; jc < nchunks; jc += gridDim.y) {
            __syncthreads();
            other[threadIdx.x] = jc * RANK_T + threadIdx.x < n ? keys[jc * RANK_T + threadIdx.x] : 0ull;
            __syncthreads();
#pragma unroll 8
            for (int j = 0; j < RANK_T; j++) greater += other[j] > key ? 1u : 0u;
        }
        if (i < n && greater) atomicAdd(&rank[i], greater);
    }
}

__global__ __launch_bounds__(256) void mis_place_kernel(NmsArgs a, const unsigned *__restrict__ count, const unsigned *__restrict__ rank,
                                                        const int *__restrict__ nfill_in, unsigned *host_out, const unsigned *rem, int look,
                                                        const unsigned *info)
{
    const int n = (int)*count, nfill = *nfill_in;
    const int placed = min(n, nfill);
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < n) {
        const int r = (int)rank[i];
        if (r < nfill) {
            const unsigned long long k = a.keys[i];
            const int slot = a.overwrite_all ? r : a.slots[r];
            klt_feat ft;
            ft.x = (float)(int)((k >> 16) & 0xffffull);
            ft.y = (float)(int)(k & 0xffffull);
            ft.val = (int32_t)__uint_as_float((uint32_t)(k >> 32));          // int(val), selectGoodFeatures.py:119
            ft.aux = 0;
            a.fl[slot] = ft;
            if (a.aff_rec) {
                klt_affine_rec rec;
                rec.aff_x = -1.f; rec.aff_y = -1.f; rec.Axx = 1.f; rec.Ayx = 0.f; rec.Axy = 0.f; rec.Ayy = 1.f; rec.valid = 0; rec.pad = 0;
                a.aff_rec[slot] = rec;
            }
        }
    }
    // candidates exhausted: selectGoodFeatures.py:78-94 (SELECTING_ALL only; DESIGN.md lists the deviation)
    if (a.overwrite_all)
        for (int s = placed + i; s < a.nfeat; s += gridDim.x * 256) {
            klt_feat ft;
            ft.x = -1.f; ft.y = -1.f; ft.val = KLT_NOT_FOUND; ft.aux = 0;
            a.fl[s] = ft;
        }
    if (i == 0 && a.placed_out) {
        a.placed_out[0] = placed;
        a.placed_out[1] = placed < nfill ? 1 : 0;       // 1: the candidates ran out before the list was full
    }
    if (blockIdx.x == 0) write_results(host_out, rem, look, info, placed, placed < nfill ? 1 : 0);
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
    hipLaunchKernelGGL(sat_rows_kernel, dim3((nrows + SR_ROWS - 1) / SR_ROWS), dim3(SR_T), 0, s, gx, gy, sat, ncols, nrows);
}

void launch_sat_cols(hipStream_t s, float *sat, int ncols, int nrows)
{
    hipLaunchKernelGGL(sat_cols_kernel, dim3((ncols + SC_COLS - 1) / SC_COLS, 3), dim3(SC_T), 0, s, sat, ncols, nrows);
}

void launch_seed_fill(hipStream_t s, const klt_feat *fl, int nfeat, uint8_t *seedmap, int ncols, int nrows, int d, uint8_t stamp)
{
    if (nfeat <= 0 || d < 0) return;
    hipLaunchKernelGGL(seed_fill_kernel, dim3(nfeat), dim3(64), 0, s, fl, nfeat, seedmap, ncols, nrows, d, stamp);
}

void launch_eigen(hipStream_t s, const SelectArgs &a)
{
    hipLaunchKernelGGL(eigen_kernel, dim3((a.npow2 + 255) / 256), dim3(256), 0, s, a);
}

void launch_topk_prefilter(hipStream_t s, const unsigned long long *keys, int n, unsigned target, unsigned *hist,
                           unsigned *info /* [0..2] + counter at [3] */, unsigned long long *out)
{
    hipLaunchKernelGGL(key_hist_kernel, dim3(256), dim3(256), 0, s, keys, n, hist, 1);
    hipLaunchKernelGGL(key_threshold_kernel, dim3(1), dim3(1024), 0, s, hist, target, info, (const int *)nullptr, 0u);
    hipLaunchKernelGGL(key_compact_kernel, dim3(512), dim3(256), 0, s, keys, n, info, out, info + 3);
}

// threshold bin for the parallel passes from a histogram of every 4th key (counts in info[] are in sampled units)
void launch_key_threshold(hipStream_t s, const unsigned long long *keys, int n, unsigned target, unsigned *hist, unsigned *info)
{
    hipLaunchKernelGGL(key_hist_kernel, dim3(256), dim3(256), 0, s, keys, n, hist, 4);
    hipLaunchKernelGGL(key_threshold_kernel, dim3(1), dim3(1024), 0, s, hist, (target + 3u) / 4u, info, (const int *)nullptr, 0u);
}

__global__ __launch_bounds__(256) void zero_words_kernel(unsigned *__restrict__ p, size_t n)
{
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) p[i] = 0u;
}

void launch_zero_words(hipStream_t s, unsigned *p, size_t n)
{
    const size_t blocks = (n + 255) / 256;
    hipLaunchKernelGGL(zero_words_kernel, dim3((unsigned)(blocks < 1024 ? (blocks ? blocks : 1) : 1024)), dim3(256), 0, s, p, n);
}

int mis_tiles(int nx, int ny) { return ((nx + MIS_TILE - 1) / MIS_TILE) * ((ny + MIS_TILE - 1) / MIS_TILE); }

size_t mis_stage_bytes(int R)
{
    const size_t W = MIS_TILE + 2 * (R > 0 ? R : 0);
    return R > 0 ? (W * MIS_TILE + W * W) * sizeof(uint32_t) : 0;
}

void launch_mis_init(hipStream_t s, const MisArgs &a)
{
    hipLaunchKernelGGL(mis_init_kernel, dim3(mis_tiles(a.nx, a.ny)), dim3(MIS_T), 0, s, a);
}

template <int RC>
static int launch_mis_round_t(hipStream_t s, const MisArgs &a, int round, size_t lds)
{
    if (lds > 48 * 1024) {
        hipError_t e = hipFuncSetAttribute((const void *)mis_round_kernel<RC>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return (int)e;
    }
    static const int tpw_later = getenv("KLT_MIS_TPW") ? atoi(getenv("KLT_MIS_TPW")) : MIS_TPW_MAX;      // 1: every pass one tile per workgroup
    // (a selection of all features has hundreds of candidates per tile for several passes: four tiles per workgroup one after the other
    // took its 5000-feature selection at 1080p from 0.149 to 0.185 ms)
    const int tpw = round == 0 || !a.sparse ? 1 : (tpw_later < 1 ? 1 : (tpw_later > MIS_TPW_MAX ? MIS_TPW_MAX : tpw_later));
    const int tiles = mis_tiles(a.nx, a.ny), per_xcd = (tiles + 7) / 8;
    hipLaunchKernelGGL(mis_round_kernel<RC>, dim3(8 * ((per_xcd + tpw - 1) / tpw)), dim3(MIS_T), lds, s, a, round, tpw);
    return 0;
}

int launch_mis_round(hipStream_t s, const MisArgs &a, int round)
{
    const size_t lds = (a.stage ? mis_stage_bytes(a.R) : 0) + (((size_t)a.acc_cap * sizeof(unsigned short) + 15) & ~(size_t)15);
    return a.R == 9 ? launch_mis_round_t<9>(s, a, round, lds) : launch_mis_round_t<0>(s, a, round, lds);
}

void launch_mis_compact(hipStream_t s, const MisArgs &a, unsigned long long *out, unsigned *out_count)
{
    const int tiles = mis_tiles(a.nx, a.ny);
    hipLaunchKernelGGL(mis_compact_kernel, dim3((tiles + 255) / 256), dim3(256), 0, s, a.acc_keys, a.acc_cnt, tiles, a.acc_cap, out, out_count);
}

int mis_tile_capacity(int R)
{
    if (R < 0) return MIS_CAP;
    const int per_side = (MIS_TILE + R) / (R + 1);      // accepted candidates are more than R cells apart in x or in y
    return per_side * per_side < MIS_CAP ? per_side * per_side : MIS_CAP;
}

void launch_mis_prepare(hipStream_t s, const klt_feat *fl, int nfeat, int overwrite_all, int *slots, int *nfill_out, klt_feat *snapshot,
                        unsigned *zero, size_t zero_n)
{
    const size_t zb = (zero_n + 1023) / 1024;
    const int feat_blocks = nfeat > 0 ? (nfeat + 255) / 256 : 1;
    hipLaunchKernelGGL(mis_prepare_kernel, dim3(feat_blocks + (unsigned)(zb < 1 ? 1 : (zb > 64 ? 64 : zb))), dim3(256), 0, s, fl, nfeat, overwrite_all,
                       slots, nfill_out, snapshot, zero, zero_n, feat_blocks);
}

void launch_eigen_hist(hipStream_t s, const SelectArgs &a)
{
    const int blocks = (a.nx * a.ny + 255) / 256;
    klt_launch(eigen_hist_kernel, dim3(blocks < 1024 ? blocks : 1024), dim3(256), 0, s, a);
    if (a.hist) hipLaunchKernelGGL(key_threshold_kernel, dim3(1), dim3(1024), 0, s, a.hist, a.hist_target, a.info, a.hist_slots, a.hist_per_slot);
}

void launch_mask_hist(hipStream_t s, const SelectArgs &a)
{
    // a.hist_target / a.hist_per_slot arrive in units of eigen_hist_kernel's sample (every 4th block)
    constexpr unsigned scale = MASK_HIST_SAMPLE / 4;
    const int per_wg = MASK_HIST_SAMPLE * 8 * 256;
    const int blocks = (a.nx * a.ny + per_wg - 1) / per_wg;
    hipLaunchKernelGGL(mask_hist_kernel, dim3(blocks < 1024 ? (blocks ? blocks : 1) : 1024), dim3(256), 0, s, a);
    const unsigned per_slot = (a.hist_per_slot + scale - 1) / scale;
    hipLaunchKernelGGL(key_threshold_kernel, dim3(1), dim3(1024), 0, s, a.hist, (a.hist_target + scale - 1) / scale, a.info, a.hist_slots,
                       per_slot ? per_slot : 1u);
}

void launch_mis_results(hipStream_t s, unsigned *host_out, const unsigned *rem, int look, const unsigned *info, const int *placed)
{
    hipLaunchKernelGGL(mis_results_kernel, dim3(1), dim3(64), 0, s, host_out, rem, look, info, placed);
}

// keys[0 .. *count) unsorted accepted candidates, *count <= bound; rank[] zeroed by the caller
void launch_mis_place(hipStream_t s, const NmsArgs &a, const unsigned *count, unsigned *rank, const int *nfill, int bound,
                      unsigned *host_out, const unsigned *rem, int look, const unsigned *info)
{
    const int chunks = (bound + RANK_T - 1) / RANK_T;
    hipLaunchKernelGGL(mis_rank_kernel, dim3(chunks < RANK_GX ? chunks : RANK_GX, chunks < RANK_GY ? chunks : RANK_GY), dim3(RANK_T), 0, s,
                       a.keys, count, rank);
    hipLaunchKernelGGL(mis_place_kernel, dim3((bound + 255) / 256), dim3(256), 0, s, a, count, rank, nfill, host_out, rem, look, info);
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
    if (a.grid_in_lds) {
        const size_t lds = (size_t)a.gw * a.gh * sizeof(uint32_t);
        hipError_t e = hipFuncSetAttribute((const void *)nms_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return (int)e;
        hipLaunchKernelGGL(nms_kernel<true>, dim3(1), dim3(NMS_T), lds, s, a);
    } else {
        hipLaunchKernelGGL(nms_kernel<false>, dim3(1), dim3(NMS_T), 0, s, a);
    }
    return 0;
}

void launch_unpack_candidates(hipStream_t s, const unsigned long long *keys, int n, float *val, int *x, int *y)
{
    if (n <= 0) return;
    hipLaunchKernelGGL(unpack_candidates_kernel, dim3((n + 255) / 256), dim3(256), 0, s, keys, n, val, x, y);
}
