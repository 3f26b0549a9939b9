// Summed-area tables as a step-synchronous wavefront pipeline (KLT_OPT_SAT_VARIANT = 1, the default).
//
// numpy's cumsum(1).cumsum(0) on f32 (goodFeaturesUtils.pyx:49-51) is one strictly sequential f32 chain per row and then
// per column; a chain cannot be split without changing the rounding, so the time of a pass is
//     (chain length) x (latency of one dependent v_add_f32, ~9 clocks on gfx950: tools/mb/valu_rate.hip)
// = 7.2 us for a 1920-long row and 4 us for a 1080-long column if the wavefront that runs a band of chains does nothing
// else.  Here it does nothing else: per step it scans one tile that already sits in LDS, in place.  Loader wavefronts
// bring tiles from HBM into a ring of three LDS slots (and form the products gx*gx, gx*gy, gy*gy on the way, row pass),
// storer wavefronts drain scanned tiles.  One s_barrier per step couples them -- and nothing else does:
//   * a loader owns every NLW-th tile and has exactly one tile of loads in flight, requested NLW steps before it is
//     needed, so its wait is for its own oldest loads only (no register ring for the compiler to rotate, the reason the
//     barrier-coupled kernels in select_kernels.hip drain vmcnt every step);
//   * the barrier is `s_waitcnt lgkmcnt(0); s_barrier`, not __syncthreads(): it orders LDS traffic and leaves global
//     loads and stores in flight.
// Tile t is written to slot t % 3 during step t, scanned during step t + 1, read back during step t + 2.
#include "klt_internal.h"

#pragma clang fp contract(off)

#ifdef SAT_PIPE_DEBUG
__device__ long long g_sat_dbg[6 * 64 * 3];       // workgroup 0: per wavefront and step {step start, work done, barrier passed}
#define SAT_MARK(w, s, i) do { if (blockIdx.x == 0 && blockIdx.y == 0 && (threadIdx.x & 63) == 0 && (s) >= 0 && (s) < 64) g_sat_dbg[((w) * 64 + (s)) * 3 + (i)] = wall_clock64(); } while (0)
__device__ long long g_fused_dbg[16 * 96 * 3];   // cols_eigen_pipe, workgroup 0: per wavefront and step {step start, work done, barrier passed}
#define FUSED_MARK(w, s, i) do { if (blockIdx.x == 0 && (threadIdx.x & 63) == 0 && (s) >= 0 && (s) < 96) g_fused_dbg[((w) * 96 + (s)) * 3 + (i)] = wall_clock64(); } while (0)
#else
#define SAT_MARK(w, s, i) do { } while (0)
#define FUSED_MARK(w, s, i) do { } while (0)
#endif

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int NLW = 3;                         // loader wavefronts
constexpr int NSW = 2;                         // storer wavefronts
constexpr int NSLOT = 3;
constexpr int PIPE_THREADS = 64 * (1 + NLW + NSW);

__device__ __forceinline__ void step_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// ------------------------------------------------------------------ row pass (+ products)
// workgroup = band of RB rows, all three planes; tile = RB rows x RT columns; chain wavefront: lane = plane * RB + row
constexpr int RB = 16, RT = 128, RLD = RT + 4;                 // row pitch 132 floats: lanes hit distinct quads of banks
constexpr int RSLOT = 3 * RB * RLD;                            // floats per slot

__global__ __launch_bounds__(PIPE_THREADS) __attribute__((amdgpu_waves_per_eu(1, 2))) void sat_rows_pipe(const float *__restrict__ gx, const float *__restrict__ gy,
                                                                float *__restrict__ sat, int ncols, int nrows)
{
    extern __shared__ __attribute__((aligned(16))) float pipe_lds[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int row0 = blockIdx.x * RB;
    const int ntiles = (ncols + RT - 1) / RT;
    const int nsteps = ntiles + 2;
    const size_t plane = (size_t)ncols * nrows;

    if (wave == 0) {                                            // ---- the chains
        const int pl = lane / RB, r = lane % RB;                // lanes 48..63 idle
        float carry = 0.f;
        for (int s = 0; s < nsteps; s++) {
            SAT_MARK(0, s, 0);
            if (s >= 1 && s <= ntiles && lane < 3 * RB) {
                float4 *row = reinterpret_cast<float4 *>(pipe_lds + ((s - 1) % NSLOT) * RSLOT + (pl * RB + r) * RLD);
                // the whole tile row is requested from LDS up front (128 VGPRs); the 128 dependent adds then run back to
                // back, the writes trailing them
                float4 v[RT / 4];
#pragma unroll
                for (int u = 0; u < RT / 4; u++) v[u] = row[u];
#pragma unroll
                for (int u = 0; u < RT / 4; u++) {
                    float4 o;
                    carry = carry + v[u].x; o.x = carry;
                    carry = carry + v[u].y; o.y = carry;
                    carry = carry + v[u].z; o.z = carry;
                    carry = carry + v[u].w; o.w = carry;
                    row[u] = o;
                }
            }
            SAT_MARK(0, s, 1);
            step_barrier();
            SAT_MARK(0, s, 2);
        }
    } else if (wave <= NLW) {                                   // ---- loaders: lane = (row parity, pixel pair of each half of the tile row)
        // The gradient planes are interleaved (klt_internal.h): a tile row is RT x 8 = 1024 contiguous bytes behind gx.  Lane q takes
        // the 16 bytes at 16 q of each 512-byte half -- pixels 2 q, 2 q + 1 and RT / 2 + 2 q, RT / 2 + 2 q + 1 with gradx and grady side
        // by side -- so every load instruction of a row reads one contiguous 512-byte piece.
        const int j = wave - 1;
        const int half = lane >> 5, q = lane & 31;
        float4 va[RB / 2], vb[RB / 2];                          // {gx, gy, gx, gy} of the lane's pixel pair in the first / second half
        auto request = [&](int t) {
            const int c0 = min(t * RT + 2 * q, ncols - 2), c1 = min(t * RT + RT / 2 + 2 * q, ncols - 2);    // clamped, unconditional (ncols % 4 == 0)
#pragma unroll
            for (int rp = 0; rp < RB / 2; rp++) {
                const size_t o = (size_t)min(row0 + 2 * rp + half, nrows - 1) * ncols;
                va[rp] = *reinterpret_cast<const float4 *>(gx + KLT_GRAD_STRIDE * (o + c0));
                vb[rp] = *reinterpret_cast<const float4 *>(gx + KLT_GRAD_STRIDE * (o + c1));
            }
        };
        for (int s = -NLW; s < nsteps; s++) {                   // (steps -NLW .. -1: the first requests only; one request site)
            SAT_MARK(wave, s, 0);
            if (s >= 0 && s % NLW == j && s < ntiles) {
                float *slot = pipe_lds + (s % NSLOT) * RSLOT;
#pragma unroll
                for (int rp = 0; rp < RB / 2; rp++) {
                    const int r = 2 * rp + half;
#pragma unroll
                    for (int hh = 0; hh < 2; hh++) {
                        const float4 p = hh ? vb[rp] : va[rp];                  // x0 y0 x1 y1
                        float2 xx, xy, yy;
                        xx.x = p.x * p.x; xx.y = p.z * p.z;                     // goodFeaturesUtils.pyx:49
                        xy.x = p.x * p.y; xy.y = p.z * p.w;                     // :50
                        yy.x = p.y * p.y; yy.y = p.w * p.w;                     // :51
                        const int col = hh * (RT / 2) + 2 * q;
                        *reinterpret_cast<float2 *>(slot + (0 * RB + r) * RLD + col) = xx;
                        *reinterpret_cast<float2 *>(slot + (1 * RB + r) * RLD + col) = xy;
                        *reinterpret_cast<float2 *>(slot + (2 * RB + r) * RLD + col) = yy;
                    }
                }
            }
            if ((s + NLW) % NLW == j && s + NLW < ntiles) request(s + NLW);
            SAT_MARK(wave, s, 1);
            if (s >= 0) step_barrier();
            SAT_MARK(wave, s, 2);
        }
    } else {                                                    // ---- storers
        const int k = wave - 1 - NLW;
        const int half = lane >> 5, q = lane & 31;
        for (int s = 0; s < nsteps; s++) {
            const int t = s - 2;
            SAT_MARK(wave, s, 0);
            if (t >= 0 && t % NSW == k) {
                const float *slot = pipe_lds + (t % NSLOT) * RSLOT;
                float4 v[3][RB / 2];
#pragma unroll
                for (int pl = 0; pl < 3; pl++)
#pragma unroll
                    for (int rp = 0; rp < RB / 2; rp++)
                        v[pl][rp] = *reinterpret_cast<const float4 *>(slot + (pl * RB + 2 * rp + half) * RLD + 4 * q);
                const int col = t * RT + 4 * q;
                if (col < ncols) {
#pragma unroll
                    for (int rp = 0; rp < RB / 2; rp++) {
                        const int row = row0 + 2 * rp + half;
                        if (row < nrows) {
#pragma unroll
                            for (int pl = 0; pl < 3; pl++)
                                *reinterpret_cast<float4 *>(sat + pl * plane + (size_t)row * ncols + col) = v[pl][rp];
                        }
                    }
                }
            }
            SAT_MARK(wave, s, 1);
            step_barrier();
            SAT_MARK(wave, s, 2);
        }
    }
}

// ------------------------------------------------------------------ column pass (in place)
// workgroup = strip of 64 columns of one plane; tile = CT rows x 64 columns; chain wavefront: lane = column
constexpr int CT = 64, CSLOT = CT * 64;

__global__ __launch_bounds__(PIPE_THREADS) __attribute__((amdgpu_waves_per_eu(1, 2))) void sat_cols_pipe(float *__restrict__ sat, int ncols, int nrows)
{
    extern __shared__ __attribute__((aligned(16))) float pipe_lds[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float *s_ = sat + (size_t)blockIdx.y * ncols * nrows;
    const int ntiles = (nrows + CT - 1) / CT;
    const int nsteps = ntiles + 2;

    if (wave == 0) {
        float carry = 0.f;
        for (int s = 0; s < nsteps; s++) {
            SAT_MARK(0, s, 0);
            if (s >= 1 && s <= ntiles) {
                float *tile = pipe_lds + ((s - 1) % NSLOT) * CSLOT + lane;
                float v[CT];
#pragma unroll
                for (int u = 0; u < CT; u++) v[u] = tile[u * 64];
#pragma unroll
                for (int u = 0; u < CT; u++) {
                    carry = carry + v[u];
                    tile[u * 64] = carry;
                }
            }
            SAT_MARK(0, s, 1);
            step_barrier();
            SAT_MARK(0, s, 2);
        }
    } else if (wave <= NLW) {                                   // ---- loaders: lane = (row of a group of four, quad of the strip)
        const int j = wave - 1;
        const int rsub = lane >> 4, q = lane & 15;
        const int c = min(blockIdx.x * 64 + 4 * q, ncols - 4);
        f32x4 v[CT / 4];
        for (int s = -NLW; s < nsteps; s++) {
            SAT_MARK(wave, s, 0);
            if (s >= 0 && s % NLW == j && s < ntiles) {
                float *tile = pipe_lds + (s % NSLOT) * CSLOT;
#pragma unroll
                for (int g = 0; g < CT / 4; g++) *reinterpret_cast<f32x4 *>(tile + (4 * g + rsub) * 64 + 4 * q) = v[g];
            }
            if ((s + NLW) % NLW == j && s + NLW < ntiles) {
                const int t = s + NLW;
#pragma unroll
                for (int g = 0; g < CT / 4; g++)
                    v[g] = *reinterpret_cast<const f32x4 *>(s_ + (size_t)min(t * CT + 4 * g + rsub, nrows - 1) * ncols + c);
            }
            SAT_MARK(wave, s, 1);
            if (s >= 0) step_barrier();
            SAT_MARK(wave, s, 2);
        }
    } else {                                                    // ---- storers
        // every storer takes its share of EVERY tile (a storer that drained whole tiles in turn was busy 0.8-0.96 us of a step the chain
        // wavefront needs 0.52-0.76 us of, tools/mb/sat_steps: sixteen 1 KB stores in a row; the other storer idled meanwhile)
        const int k = wave - 1 - NLW;
        const int rsub = lane >> 4, q = lane & 15;
        const int c = blockIdx.x * 64 + 4 * q;
        constexpr int GS = CT / 4 / NSW;                           // groups of four rows per storer and tile
        for (int s = 0; s < nsteps; s++) {
            const int t = s - 2;
            SAT_MARK(wave, s, 0);
            if (t >= 0) {
                const float *tile = pipe_lds + (t % NSLOT) * CSLOT;
                f32x4 v[GS];
#pragma unroll
                for (int g = 0; g < GS; g++) v[g] = *reinterpret_cast<const f32x4 *>(tile + (4 * (k * GS + g) + rsub) * 64 + 4 * q);
                if (c < ncols) {
#pragma unroll
                    for (int g = 0; g < GS; g++) {
                        const int row = t * CT + 4 * (k * GS + g) + rsub;
                        if (row < nrows) *reinterpret_cast<f32x4 *>(s_ + (size_t)row * ncols + c) = v[g];
                    }
                }
            }
            SAT_MARK(wave, s, 1);
            step_barrier();
            SAT_MARK(wave, s, 2);
        }
    }
}


// ------------------------------------------------------------------ column pass + eigenvalue keys (the column-summed tables stay in LDS)
// The column pass above writes 12 bytes per pixel that the eigenvalue kernel reads straight back: at 4K 100 MB each way, 2 x ~20 us of a
// selection whose tables nobody else reads (klt_select_prepare_async).  Here a workgroup owns a strip of FW table columns of all three
// planes and walks down the frame in tiles of FT rows through a ring of four LDS slots:
//   step s: loaders write tile s | chain wavefronts scan tile s - 1 in place | eigen wavefronts score the candidates whose window's
//           BOTTOM row lies in tile s - 2 (its top row, 2 hh + 1 rows higher, lies in tile s - 2 or s - 3: 2 hh + 1 <= FT)
// and the only things that reach HBM are the keys.  A strip of FW table columns serves FW - (2 hw + 1) candidate columns (a window
// needs the columns x - hw - 1 and x + hw), so neighbouring strips overlap by 2 hw + 1 columns that both scan: FW = 32 with two rows
// per wavefront keeps the overlap at 22 % (7x7) and gives a 4K frame 144 workgroups -- scoring a candidate is ~80 instructions (an f64
// square root among them), a 64-column strip would leave the whole frame's scoring to 64 CUs.
#ifndef KLT_F_LOAD
#define KLT_F_LOAD 3
#endif
#ifndef KLT_F_EIGEN
#define KLT_F_EIGEN 8
#endif
#ifndef KLT_F_FT
#define KLT_F_FT 32
#endif
constexpr int FW = 32, FT = KLT_F_FT, FSLOTS = 4 /* a power of two */, FPLANE = FT * FW, FSLOT = 3 * FPLANE;
constexpr int FN_CHAIN = 2, FN_LOAD = KLT_F_LOAD, FN_EIGEN = KLT_F_EIGEN;
constexpr int FUSED_THREADS = 64 * (FN_CHAIN + FN_LOAD + FN_EIGEN);
// (Bands of candidate rows per strip, each band a workgroup of its own whose chains start at the top of the frame -- 288 workgroups for a
// 4K frame instead of 144 -- were measured: 80.7 us alone against 68.7.  A CU's f64 pipes are what a scoring step waits for
// (tools/mb/cols_eigen_steps.hip: the eight eigen wavefronts work 0.5-0.9 us of a 1.0 us step), two workgroups on one CU halve each
// other, and 144 strips do not spread evenly over 256 CUs.)

struct ColsEigenArgs {
    const float *sat;                 // row-summed planes
    unsigned long long *keys;
    float min_eig_f32;                // klt_threshold_f32(max(min_eigenvalue, 1))
    int ncols, nrows, bx, by, nx, ny, hw, hh, per_strip, ntiles;
};

__global__ __launch_bounds__(FUSED_THREADS) __attribute__((amdgpu_waves_per_eu(1, 4))) void cols_eigen_pipe(ColsEigenArgs a)
{
    extern __shared__ __attribute__((aligned(16))) float pipe_lds[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int half = lane >> 5, col = lane & 31;
    const int c0 = a.bx - a.hw - 1 + a.per_strip * (int)blockIdx.x;       // first table column of the strip
    const int ntiles = a.ntiles, nsteps = ntiles + 2;
    const int first_scored = (a.by + a.hh) / FT;                               // the first tile that holds the bottom row of a candidate's window
    const size_t plane = (size_t)a.ncols * a.nrows;

    if (wave < FN_CHAIN) {                                       // ---- the chains: lane = (plane, column); wavefront 1 runs plane 2 on 32 lanes
        const int pl = 2 * wave + half;
        const bool mine = pl < 3;
        float carry = 0.f;
        for (int s = 0; s < nsteps; s++) {
            FUSED_MARK(wave, s, 0);
            if (s >= 1 && s <= ntiles && mine) {
                float *tile = pipe_lds + ((s - 1) % FSLOTS) * FSLOT + pl * FPLANE + col;
                float v[FT];
#pragma unroll
                for (int u = 0; u < FT; u++) v[u] = tile[u * FW];
#pragma unroll
                for (int u = 0; u < FT; u++) {
                    carry = carry + v[u];
                    tile[u * FW] = carry;
                }
            }
            FUSED_MARK(wave, s, 1);
            step_barrier();
            FUSED_MARK(wave, s, 2);
        }
    } else if (wave < FN_CHAIN + FN_LOAD) {                       // ---- loaders: lane = (row of a group of eight, quad of the strip); every FN_LOAD-th tile, all planes
        // 16-byte loads (a strip row is 128 bytes: eight lanes), 12 per tile -- with one dword per lane a tile was 48 load instructions, and
        // the loader whose turn it was held every step's barrier for the time it takes to issue them.  The strip starts at any column, so
        // the loads are only 4-byte aligned; the last strip reads up to 31 floats past a row's end (the next row, or the pad behind the
        // planes): columns that no candidate's window uses.
        const int j = wave - FN_CHAIN;
        const int g = lane >> 3, q = lane & 7;
        typedef float f32x4u __attribute__((ext_vector_type(4), aligned(4)));
        f32x4 v[3][FT / 8];
        for (int s = -FN_LOAD; s < nsteps; s++) {
            FUSED_MARK(wave, s, 0);
            if (s >= 0 && s % FN_LOAD == j && s < ntiles) {
                float *slot = pipe_lds + (s % FSLOTS) * FSLOT + g * FW + 4 * q;
#pragma unroll
                for (int pl = 0; pl < 3; pl++)
#pragma unroll
                    for (int r8 = 0; r8 < FT / 8; r8++) *reinterpret_cast<f32x4 *>(slot + pl * FPLANE + 8 * r8 * FW) = v[pl][r8];
            }
            if ((s + FN_LOAD) % FN_LOAD == j && s + FN_LOAD < ntiles) {
                const int t = s + FN_LOAD;
#pragma unroll
                for (int r8 = 0; r8 < FT / 8; r8++) {
                    const size_t o = (size_t)min(t * FT + 8 * r8 + g, a.nrows - 1) * a.ncols + (c0 + 4 * q);
#pragma unroll
                    for (int pl = 0; pl < 3; pl++) v[pl][r8] = *reinterpret_cast<const f32x4u *>(a.sat + pl * plane + o);
                }
            }
            FUSED_MARK(wave, s, 1);
            if (s >= 0) step_barrier();
            FUSED_MARK(wave, s, 2);
        }
    } else {                                                    // ---- eigen wavefronts: lane = (row parity, strip column); two row pairs per step
        const int e = wave - FN_CHAIN - FN_LOAD;
        const int x = c0 + col;
        const bool col_ok = col >= a.hw + 1 && col < a.hw + 1 + a.per_strip && x < a.bx + a.nx;
        // (lanes that score nothing read the strip's own column: valid LDS addresses, results dropped -- the loop body is straight-line code)
        const int cl = col_ok ? col - a.hw - 1 : col, cr = col_ok ? col + a.hw : col;          // left / right table column of the window
        constexpr int NIT = FT / 2 / FN_EIGEN;
        static_assert(NIT * 2 * FN_EIGEN == FT, "the eigen wavefronts share a tile's rows evenly");
        // everything but the tile number is a constant of the lane: row of the window's bottom inside its tile, row of its top inside
        // that tile or the one before, the candidate's row in tile 0, and where its key goes
        int bot_off[NIT], top_off[NIT], y0[NIT];
        bool top_before[NIT];
        long long kidx[NIT];
#pragma unroll
        for (int it = 0; it < NIT; it++) {
            const int rl = 2 * (e + FN_EIGEN * it) + half, rtl = rl - (2 * a.hh + 1);
            bot_off[it] = rl * FW;
            top_before[it] = rtl < 0;
            top_off[it] = (rtl < 0 ? rtl + FT : rtl) * FW;
            y0[it] = rl - a.hh;
            kidx[it] = (long long)(y0[it] - a.by) * a.nx + (x - a.bx);       // (negative for the rows above the first candidate row: not stored to)
        }
        const long long kstep = (long long)FT * a.nx;
        for (int s = 0; s < nsteps; s++) {
            const int t = s - 2;
            FUSED_MARK(wave, s, 0);
            if (t >= first_scored && t < ntiles) {
                const float *sb = pipe_lds + (t & (FSLOTS - 1)) * FSLOT, *sp = pipe_lds + ((t - 1) & (FSLOTS - 1)) * FSLOT;
                float sum[NIT][3];
#pragma unroll
                for (int it = 0; it < NIT; it++) {
                    const float *top = (top_before[it] ? sp : sb) + top_off[it];
                    const float *bot = sb + bot_off[it];
#pragma unroll
                    for (int pl = 0; pl < 3; pl++) {
                        const float wa = top[pl * FPLANE + cl], wb = top[pl * FPLANE + cr];
                        const float wc = bot[pl * FPLANE + cr], wd = bot[pl * FPLANE + cl];
                        sum[it][pl] = ((wc + wa) - wb) - wd;                    // SumGradientInWindow, goodFeaturesUtils.pyx:23-31
                    }
                }
#pragma unroll
                for (int it = 0; it < NIT; it++) {
                    const int y = t * FT + y0[it];
                    const float val = klt_window_value(sum[it][0], sum[it][1], sum[it][2]);
                    const unsigned long long key = val >= a.min_eig_f32 ? klt_pack_key(val, x, y) : 0ull;
                    if (col_ok && (unsigned)(y - a.by) < (unsigned)a.ny) a.keys[kidx[it]] = key;
                }
            }
#pragma unroll
            for (int it = 0; it < NIT; it++)
                if (t >= 0) kidx[it] += kstep;
            FUSED_MARK(wave, s, 1);
            step_barrier();
            FUSED_MARK(wave, s, 2);
        }
    }
}

}  // namespace

// Both passes move whole aligned quads: they need ncols % 4 == 0 and 16-byte aligned planes; -1 = not applicable (the caller
// falls back to the kernels in select_kernels.hip).
static bool quads_ok(const void *a, const void *b, const void *c, int ncols, int nrows)
{
    return ncols >= 4 && (ncols & 3) == 0 && ((((size_t)a | (size_t)b | (size_t)c) & 15) == 0) && (((size_t)ncols * nrows) & 3) == 0;
}

int launch_sat_rows_pipe(hipStream_t s, const float *gx, const float *gy, float *sat, int ncols, int nrows)
{
    if (gy != gx + 1 || !quads_ok(gx, gx, sat, ncols, nrows)) return -1;      // the interleaved gradient planes (klt_internal.h)
    constexpr size_t lds = sizeof(float) * NSLOT * RSLOT;
    static bool set = false;
    if (!set) {
        hipError_t e = hipFuncSetAttribute((const void *)sat_rows_pipe, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return (int)e;
        set = true;
    }
    klt_launch(sat_rows_pipe, dim3((nrows + RB - 1) / RB), dim3(PIPE_THREADS), (unsigned)lds, s, gx, gy, sat, ncols, nrows);
    return 0;
}

int launch_sat_cols_pipe(hipStream_t s, float *sat, int ncols, int nrows)
{
    if (!quads_ok(sat, sat, sat, ncols, nrows)) return -1;
    constexpr size_t lds = sizeof(float) * NSLOT * CSLOT;
    static bool set = false;
    if (!set) {
        hipError_t e = hipFuncSetAttribute((const void *)sat_cols_pipe, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return (int)e;
        set = true;
    }
    klt_launch(sat_cols_pipe, dim3((ncols + 63) / 64, 3), dim3(PIPE_THREADS), (unsigned)lds, s, sat, ncols, nrows);
    return 0;
}

bool sat_cols_eigen_ok(const SelectArgs &a)
{
    // keys only, every pixel a candidate column / row (step 1), a window whose rows fit one tile and whose columns leave a strip room
    if (a.seedmap || a.valmap || a.val_in || a.hist || a.step != 1 || 2 * a.hh + 1 > FT || FW - (2 * a.hw + 1) < 8 || a.nx <= 0 || a.ny <= 0) return false;
    return a.bx - a.hw - 1 >= 0 && a.by - a.hh - 1 >= 0 && a.by + a.ny + a.hh <= a.nrows;
}

int launch_sat_cols_eigen_pipe(hipStream_t s, const float *sat, const SelectArgs &a)
{
    if (!sat_cols_eigen_ok(a)) return -1;
    ColsEigenArgs k;
    k.sat = sat; k.keys = a.keys; k.min_eig_f32 = klt_threshold_f32(a.min_eig);
    k.ncols = a.ncols; k.nrows = a.nrows; k.bx = a.bx; k.by = a.by; k.nx = a.nx; k.ny = a.ny; k.hw = a.hw; k.hh = a.hh;
    k.per_strip = FW - (2 * a.hw + 1);
    k.ntiles = (a.by + a.ny + a.hh + FT - 1) / FT;              // rows 0 .. by + ny - 1 + hh: the chains start at the top of the frame
    constexpr size_t lds = sizeof(float) * FSLOTS * FSLOT;
    static bool set = false;
    if (!set) {
        hipError_t e = hipFuncSetAttribute((const void *)cols_eigen_pipe, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return (int)e;
        set = true;
    }
    klt_launch(cols_eigen_pipe, dim3((a.nx + k.per_strip - 1) / k.per_strip), dim3(FUSED_THREADS), (unsigned)lds, s, k);
    return 0;
}
