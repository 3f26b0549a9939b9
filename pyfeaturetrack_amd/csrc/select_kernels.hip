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
// block = 256 threads = one band of 64 rows; all four waves move 64x64 tiles of gx / gy between HBM and LDS with
// coalesced 256-byte row segments (the next tile is prefetched into registers while the current one is scanned);
// waves 0..2 each run the 64 sequential row chains of one plane (0: gx*gx, 1: gx*gy, 2: gy*gy), lane = row.
constexpr int SAT_LD = 68;      // tile row stride in floats: 16-byte aligned rows, conflict-free b128 column walks

__global__ __launch_bounds__(256) void sat_rows_kernel(const float *__restrict__ gx, const float *__restrict__ gy,
                                                        float *__restrict__ sat, int ncols, int nrows)
{
    __shared__ __attribute__((aligned(16))) float tin[2][64 * SAT_LD];    // gx, gy tiles
    __shared__ __attribute__((aligned(16))) float tout[3][64 * SAT_LD];   // row-prefix tiles of the three planes
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int row0 = blockIdx.x * 64;
    const size_t plane = (size_t)ncols * nrows;
    float carry = 0.f;
    float pgx[16], pgy[16];
    auto fetch = [&](int x0) {
#pragma unroll
        for (int k = 0; k < 16; k++) {
            const int row = row0 + k * 4 + wave, col = x0 + lane;
            const bool ok = row < nrows && col < ncols;
            pgx[k] = ok ? gx[(size_t)row * ncols + col] : 0.f;
            pgy[k] = ok ? gy[(size_t)row * ncols + col] : 0.f;
        }
    };
    fetch(0);
    for (int x0 = 0; x0 < ncols; x0 += 64) {
#pragma unroll
        for (int k = 0; k < 16; k++) {
            tin[0][(k * 4 + wave) * SAT_LD + lane] = pgx[k];
            tin[1][(k * 4 + wave) * SAT_LD + lane] = pgy[k];
        }
        __syncthreads();
        if (x0 + 64 < ncols) fetch(x0 + 64);                 // in flight during the scan below
        if (wave < 3) {
            const float4 *a4 = reinterpret_cast<const float4 *>(&tin[wave == 2 ? 1 : 0][lane * SAT_LD]);
            const float4 *b4 = reinterpret_cast<const float4 *>(&tin[wave == 0 ? 0 : 1][lane * SAT_LD]);
            float4 *o4 = reinterpret_cast<float4 *>(&tout[wave][lane * SAT_LD]);
#pragma unroll 4
            for (int c = 0; c < 16; c++) {
                const float4 a = a4[c], b = b4[c];
                float4 o;
                { const float p = a.x * b.x; carry = carry + p; o.x = carry; }
                { const float p = a.y * b.y; carry = carry + p; o.y = carry; }
                { const float p = a.z * b.z; carry = carry + p; o.z = carry; }
                { const float p = a.w * b.w; carry = carry + p; o.w = carry; }
                o4[c] = o;
            }
        }
        __syncthreads();
        const int col = x0 + lane;
        if (col < ncols) {
#pragma unroll
            for (int k = 0; k < 16; k++) {
                const int r = k * 4 + wave, row = row0 + r;
                if (row < nrows) {
                    sat[(size_t)row * ncols + col] = tout[0][r * SAT_LD + lane];
                    sat[plane + (size_t)row * ncols + col] = tout[1][r * SAT_LD + lane];
                    sat[2 * plane + (size_t)row * ncols + col] = tout[2][r * SAT_LD + lane];
                }
            }
        }
        // the next iteration's tin writes are ordered behind this iteration's scan by the barrier above;
        // tout is rewritten only after the next barrier
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

// ------------------------------------------------------------------ top-K prefilter for the sort
// The greedy pass only ever looks at a prefix of the sorted candidates (197 775 of 1 411 200 at 1080p for 5000
// features).  A 8192-bin histogram over the top 13 bits of the (positive) f32 eigenvalue picks the smallest bin
// boundary that keeps at least `target` candidates; only those are compacted and sorted.  Order inside the kept set
// is unchanged (same keys), so the walk sees exactly the same prefix; if it ever runs off the end of the kept set
// before the list is full, the host repeats the selection with the full sort.
constexpr int HIST_BINS = 8192;
__device__ __forceinline__ unsigned key_bin(unsigned long long key) { return (unsigned)(key >> 50) & (HIST_BINS - 1); }

__global__ __launch_bounds__(256) void key_hist_kernel(const unsigned long long *__restrict__ keys, int n, unsigned *__restrict__ hist)
{
    __shared__ unsigned h[HIST_BINS];
    for (int i = threadIdx.x; i < HIST_BINS; i += 256) h[i] = 0u;
    __syncthreads();
    for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
        const unsigned long long key = keys[i];
        if (key) atomicAdd(&h[key_bin(key)], 1u);
    }
    __syncthreads();
    for (int i = threadIdx.x; i < HIST_BINS; i += 256)
        if (h[i]) atomicAdd(&hist[i], h[i]);
}

// info[0] = threshold bin, info[1] = number of keys in bins >= threshold, info[2] = number of valid keys
__global__ __launch_bounds__(1024) void key_threshold_kernel(const unsigned *__restrict__ hist, unsigned target, unsigned *__restrict__ info)
{
    __shared__ unsigned suf[1025];
    const int t = threadIdx.x;
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

__global__ __launch_bounds__(MIS_T) void mis_init_kernel(MisArgs a)
{
    __shared__ unsigned s_cursor;
    const int tiles_x = (a.nx + MIS_TILE - 1) / MIS_TILE;
    const int tx = blockIdx.x % tiles_x, ty = blockIdx.x / tiles_x;
    const unsigned thr = a.info[0];
    const int lane = threadIdx.x & 63;
    if (threadIdx.x == 0) s_cursor = 0u;
    __syncthreads();
    uint32_t *list = a.list + (size_t)blockIdx.x * MIS_CAP;
    for (int k0 = 0; k0 < MIS_CAP; k0 += MIS_T) {
        const int k = k0 + threadIdx.x;
        const int xi = tx * MIS_TILE + (k & (MIS_TILE - 1)), yi = ty * MIS_TILE + k / MIS_TILE;
        const bool inside = xi < a.nx && yi < a.ny;
        const int p = yi * a.nx + xi;
        const unsigned long long key = inside ? a.keys[p] : 0ull;
        const bool keep = key != 0ull && key_bin(key) >= thr;
        if (inside) a.st[p] = keep ? (uint32_t)(key >> 32) : 0u;
        const unsigned long long m = __ballot(keep);
        unsigned wbase = 0;
        if (lane == 0 && m) wbase = atomicAdd(&s_cursor, (unsigned)__popcll(m));
        wbase = __shfl(wbase, 0);
        if (keep) list[wbase + __popcll(m & ((1ull << lane) - 1ull))] = (uint32_t)p;
    }
    __syncthreads();
    if (threadIdx.x == 0) a.cnt[blockIdx.x] = s_cursor;
}

__global__ __launch_bounds__(MIS_T) void mis_round_kernel(MisArgs a, int round)
{
    // staged tiles: K = keys of the tile + halo ((32 + 2R)^2 u64: state << 32 | x << 16 | y, so the accepted bit and the
    // rank order both survive a plain max), H = max of K over the horizontal window, interior columns only.  The tile
    // is iterated a few times per launch: decisions taken inside the tile are visible to the next iteration at once,
    // the halo keeps the states it had when the tile was staged (stale = still undecided = a decision postponed).
    extern __shared__ unsigned long long lds64[];
    __shared__ uint32_t acc_local[MIS_CAP];
    __shared__ unsigned s_cursor, s_acc, s_base, s_progress;
    unsigned n = a.cnt[blockIdx.x];
    if (n == 0u) return;
    const int tiles_x = (a.nx + MIS_TILE - 1) / MIS_TILE;
    const int tx = blockIdx.x % tiles_x, ty = blockIdx.x / tiles_x;
    const int R = a.R, W = MIS_TILE + 2 * R, L = 2 * R + 1;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const bool staged = a.stage && R > 0;
    unsigned long long *K = lds64, *H = lds64 + (staged ? W * W : 0);
    const int ox = tx * MIS_TILE - R, oy = ty * MIS_TILE - R;
    uint32_t *list = a.list + (size_t)blockIdx.x * MIS_CAP;
    if (threadIdx.x == 0) s_acc = 0u;
    if (staged) {
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
                    const int r = r0 + wave + 4 * u, gy = oy + r;
                    if (cx < W && r < W)
                        K[r * W + cx] = v[u] ? ((unsigned long long)v[u] << 32) | ((unsigned long long)gx << 16) | (unsigned long long)gy : 0ull;
                }
            }
    }
    const int iterations = staged ? a.iterations : 1;
    for (int it = 0; it < iterations && n > 0u; it++) {
        __syncthreads();                                     // K complete / updated; s_* of the last iteration consumed
        if (threadIdx.x == 0) { s_cursor = 0u; s_progress = 0u; }
        const unsigned acc_before = s_acc;
        if (staged) {
            for (int k = threadIdx.x; k < W * MIS_TILE; k += MIS_T) {
                const unsigned long long *row = &K[(k / MIS_TILE) * W + (k % MIS_TILE)];     // window [c, c + 2R] of the row
                unsigned long long m = row[0];
#pragma unroll 6
                for (int dx = 1; dx <= 2 * R; dx++) m = max(m, row[dx]);
                H[k] = m;
            }
        }
        __syncthreads();
        for (unsigned base = 0; base < n; base += MIS_T) {
            const unsigned i = base + threadIdx.x;
            const bool valid = i < n;
            const int p = valid ? (int)list[i] : 0;
            const int xi = p % a.nx, yi = p / a.nx;
            bool near_accepted = false, blocked = false, gone = false;
            uint32_t sp = 0u;
            int kidx = 0;
            if (valid && staged) {
                const int lx = xi - tx * MIS_TILE, ly = yi - ty * MIS_TILE;          // interior coordinates
                kidx = (ly + R) * W + lx + R;
                const unsigned long long kp = K[kidx];
                sp = (uint32_t)(kp >> 32);
                gone = kp == 0ull;                                                   // rejected by a neighbour's push
                unsigned long long m = 0ull;
                const unsigned long long *col = &H[ly * MIS_TILE + lx];              // rows ly .. ly + 2R of H
#pragma unroll 6
                for (int dy = 0; dy <= 2 * R; dy++) m = max(m, col[dy * MIS_TILE]);
                near_accepted = (m >> 63) != 0ull;
                blocked = m > kp;
            } else if (valid) {
                sp = a.st[p];
                gone = sp == 0u;
                const int x0 = max(xi - R, 0), x1 = min(xi + R, a.nx - 1), y0 = max(yi - R, 0), y1 = min(yi + R, a.ny - 1);
                for (int yy = y0; yy <= y1 && !near_accepted && !gone; yy++) {
                    const uint32_t *row = a.st + (size_t)yy * a.nx;
                    for (int xx = x0; xx <= x1; xx++) {
                        const uint32_t s = row[xx];
                        near_accepted |= (s >> 31) != 0u;
                        blocked |= s > sp || (s == sp && (xx > xi || (xx == xi && yy > yi)));
                    }
                }
            }
            const bool live = valid && !gone;
            const bool reject = live && near_accepted;
            const bool wait = live && !near_accepted && blocked;
            const bool accept = live && !near_accepted && !blocked;
            if (reject) a.st[p] = 0u;
            if (accept) {
                a.st[p] = sp | 0x80000000u;
                acc_local[atomicAdd(&s_acc, 1u)] = (uint32_t)p;
                if (!staged) {                               // push: every undecided neighbour is rejected right away
                    const int x0 = max(xi - R, 0), x1 = min(xi + R, a.nx - 1), y0 = max(yi - R, 0), y1 = min(yi + R, a.ny - 1);
                    for (int yy = y0; yy <= y1; yy++)
                        for (int xx = x0; xx <= x1; xx++) {
                            uint32_t *q = a.st + (size_t)yy * a.nx + xx;
                            const uint32_t s = *q;
                            if ((xx != xi || yy != yi) && s != 0u && (s >> 31) == 0u) *q = 0u;
                        }
                }
            }
            if ((reject || accept || (valid && gone)) && staged) s_progress = 1u;
            __syncthreads();                                 // every read of this batch of the list (and of K by it) is done
            if (staged && reject) K[kidx] = 0ull;
            if (staged && accept) K[kidx] |= 1ull << 63;
            const unsigned long long m = __ballot(wait);
            unsigned wbase = 0;
            if (lane == 0 && m) wbase = atomicAdd(&s_cursor, (unsigned)__popcll(m));
            wbase = __shfl(wbase, 0);
            if (wait) list[wbase + __popcll(m & ((1ull << lane) - 1ull))] = (uint32_t)p;
        }
        __syncthreads();
        n = s_cursor;
        const unsigned nacc = s_acc;
        if (staged && nacc > acc_before) {                   // push for the staged tile: the block shares the windows
            for (unsigned idx = threadIdx.x; idx < (nacc - acc_before) * (unsigned)(L * L); idx += MIS_T) {
                const int p = (int)acc_local[acc_before + idx / (unsigned)(L * L)], w = (int)(idx % (unsigned)(L * L));
                const int gx = p % a.nx - R + w % L, gy = p / a.nx - R + w / L;
                const int kq = (gy - oy) * W + (gx - ox);
                const unsigned long long k = K[kq];
                if (k != 0ull && (k >> 63) == 0ull) {
                    K[kq] = 0ull;
                    a.st[(size_t)gy * a.nx + gx] = 0u;
                }
            }
        }
        if (!s_progress) break;                              // nothing moved: the rest waits for other tiles
    }
    __syncthreads();
    const unsigned nacc = s_acc;
    if (threadIdx.x == 0) {
        a.cnt[blockIdx.x] = n;
        if (n) atomicAdd(&a.remaining[round], n);
        s_base = nacc ? atomicAdd(a.acc_count, nacc) : 0u;
    }
    __syncthreads();
    for (unsigned i = threadIdx.x; i < nacc; i += MIS_T) a.acc_keys[s_base + i] = a.keys[acc_local[i]];
}

// ---- placement of the accepted candidates: rank by counting, then the first `free slots` fill the list
// free slots in list order (selectGoodFeatures.py:109-110): every slot when overwriting, else the lost features
__global__ __launch_bounds__(1024) void free_slots_kernel(const klt_feat *__restrict__ fl, int nfeat, int overwrite_all,
                                                          int *__restrict__ slots, int *__restrict__ nfill_out)
{
    __shared__ int scan[1024];
    const int tid = threadIdx.x;
    if (overwrite_all) { if (tid == 0) *nfill_out = nfeat; return; }
    const int per = (nfeat + 1023) / 1024, lo = tid * per, hi = min(lo + per, nfeat);
    int cnt = 0;
    for (int i = lo; i < hi; i++) cnt += fl[i].val < 0;
    scan[tid] = cnt;
    __syncthreads();
    for (int off = 1; off < 1024; off <<= 1) {
        const int v = tid >= off ? scan[tid - off] : 0;
        __syncthreads();
        scan[tid] += v;
        __syncthreads();
    }
    int k = scan[tid] - cnt;
    for (int i = lo; i < hi; i++)
        if (fl[i].val < 0) slots[k++] = i;
    if (tid == 1023) *nfill_out = scan[tid];
}

// rank[i] += number of accepted keys in chunk blockIdx.y that are greater than key i (keys are distinct)
constexpr int RANK_T = 256;
__global__ __launch_bounds__(RANK_T) void mis_rank_kernel(const unsigned long long *__restrict__ keys, const unsigned *__restrict__ count,
                                                          unsigned *__restrict__ rank)
{
    __shared__ unsigned long long other[RANK_T];
    const unsigned n = *count;
    const unsigned i = blockIdx.x * RANK_T + threadIdx.x, j0 = blockIdx.y * RANK_T;
    if (blockIdx.x * RANK_T >= n || j0 >= n) return;
    other[threadIdx.x] = j0 + threadIdx.x < n ? keys[j0 + threadIdx.x] : 0ull;
    __syncthreads();
    if (i >= n) return;
    const unsigned long long key = keys[i];
    unsigned greater = 0;
#pragma unroll 8
    for (int j = 0; j < RANK_T; j++) greater += other[j] > key ? 1u : 0u;
    if (greater) atomicAdd(&rank[i], greater);
}

__global__ __launch_bounds__(256) void mis_place_kernel(NmsArgs a, const unsigned *__restrict__ count, const unsigned *__restrict__ rank,
                                                        const int *__restrict__ nfill_in)
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
    hipLaunchKernelGGL(sat_rows_kernel, dim3((nrows + 63) / 64), dim3(256), 0, s, gx, gy, sat, ncols, nrows);
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

void launch_topk_prefilter(hipStream_t s, const unsigned long long *keys, int n, unsigned target, unsigned *hist,
                           unsigned *info /* [0..2] + counter at [3] */, unsigned long long *out)
{
    hipLaunchKernelGGL(key_hist_kernel, dim3(256), dim3(256), 0, s, keys, n, hist);
    hipLaunchKernelGGL(key_threshold_kernel, dim3(1), dim3(1024), 0, s, hist, target, info);
    hipLaunchKernelGGL(key_compact_kernel, dim3(512), dim3(256), 0, s, keys, n, info, out, info + 3);
}

void launch_key_threshold(hipStream_t s, const unsigned long long *keys, int n, unsigned target, unsigned *hist, unsigned *info)
{
    hipLaunchKernelGGL(key_hist_kernel, dim3(256), dim3(256), 0, s, keys, n, hist);
    hipLaunchKernelGGL(key_threshold_kernel, dim3(1), dim3(1024), 0, s, hist, target, info);
}

int mis_tiles(int nx, int ny) { return ((nx + MIS_TILE - 1) / MIS_TILE) * ((ny + MIS_TILE - 1) / MIS_TILE); }

size_t mis_stage_bytes(int R)
{
    const size_t W = MIS_TILE + 2 * (R > 0 ? R : 0);
    return R > 0 ? (W * W + W * MIS_TILE) * sizeof(unsigned long long) : 0;
}

void launch_mis_init(hipStream_t s, const MisArgs &a)
{
    hipLaunchKernelGGL(mis_init_kernel, dim3(mis_tiles(a.nx, a.ny)), dim3(MIS_T), 0, s, a);
}

int launch_mis_round(hipStream_t s, const MisArgs &a, int round)
{
    const size_t lds = a.stage ? mis_stage_bytes(a.R) : 0;
    if (lds > 48 * 1024) {
        hipError_t e = hipFuncSetAttribute((const void *)mis_round_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return (int)e;
    }
    hipLaunchKernelGGL(mis_round_kernel, dim3(mis_tiles(a.nx, a.ny)), dim3(MIS_T), lds, s, a, round);
    return 0;
}

void launch_free_slots(hipStream_t s, const klt_feat *fl, int nfeat, int overwrite_all, int *slots, int *nfill_out)
{
    hipLaunchKernelGGL(free_slots_kernel, dim3(1), dim3(1024), 0, s, fl, nfeat, overwrite_all, slots, nfill_out);
}

// keys[0 .. *count) unsorted accepted candidates, *count <= bound; rank[] zeroed by the caller
void launch_mis_place(hipStream_t s, const NmsArgs &a, const unsigned *count, unsigned *rank, const int *nfill, int bound)
{
    const int chunks = (bound + RANK_T - 1) / RANK_T;
    hipLaunchKernelGGL(mis_rank_kernel, dim3(chunks, chunks), dim3(RANK_T), 0, s, a.keys, count, rank);
    hipLaunchKernelGGL(mis_place_kernel, dim3((bound + 255) / 256), dim3(256), 0, s, a, count, rank, nfill);
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
