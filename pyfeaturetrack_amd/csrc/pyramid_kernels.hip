// LDS-tiled, fused pyramid-build kernels for gfx950 (the fast path; conv_kernels.hip is the generic one).
//
//   smooth_grad_kernel<TIn, true>   u8/f32 frame -> smoothed level-0 image + gradx + grady   (1 launch, was 4)
//   smooth_grad_kernel<float,false> level image   -> gradx + grady                            (1 launch, was 2)
//   pyr_reduce_kernel               level l-1 image -> level l image (smooth + subsample)      (1 launch, was 2)
//
// blockIdx.z is the frame of a batch (both frames of a pair, or all pairs of a cfg-4 shard, go through one
// launch).  Every intermediate of a tile lives in LDS; HBM sees each input once (plus halo) and each output
// once, in coalesced rows.
//
// Bit-exactness.  Every output sample is computed by the same FP64 expression, in the same
// order, as scipy's correlate1d, with the f32 rounding between the horizontal and the vertical pass.
// Fusing smoothing and differentiation needs smoothed samples *outside* the frame (scipy reflects the
// smoothed image when it differentiates it).  The tile is addressed in virtual coordinates and the raw frame
// is loaded through the reflect map R; for a symmetric tap set k and any virtual x,
//     sum_j k_j raw[R(x + j)]  ==  sum_j k_j raw[R(R(x) + j)]      (bit for bit)
// because R(-1 - t) = R(t), R(2n - 1 - t) = R(t), and scipy's symmetric loop adds the pair (x[-j] + x[j])
// commutatively before multiplying.  So the smoothed value computed at a virtual position equals the smoothed
// image at the reflected position, which is what the differentiation must see.  The host only takes this
// path when the smoothing taps are symmetric (they are Gaussians) and the tile fits in LDS.
#include <cstdlib>

#include "klt_internal.h"

#pragma clang fp contract(off)

// tools/mb/l0_stages.hip builds this file with KLT_STAGE_CLOCKS to time the stages of one workgroup; a no-op otherwise
#ifdef KLT_STAGE_CLOCKS
__device__ long long g_stage_clk[64 * 8];
__device__ long long g_block_clk[8192 * 2];       // start / end of every workgroup, and the XCC / CU it ran on
__device__ unsigned g_block_hw[8192];
#define STAGE_MARK(n)                                                                                   \
    do {                                                                                                \
        if (threadIdx.x == 0 && blockIdx.y == 8 && blockIdx.z == 0 && blockIdx.x < 64) g_stage_clk[blockIdx.x * 8 + (n)] = wall_clock64(); \
        if (threadIdx.x == 0 && ((n) == 0 || (n) == 5)) {                                                \
            const unsigned lin = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z);         \
            if (lin < 8192) {                                                                            \
                g_block_clk[2 * lin + ((n) == 5)] = wall_clock64();                                      \
                unsigned hw, xcc;                                                                        \
                asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));                         \
                asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));                       \
                g_block_hw[lin] = (hw & 0xffff) | (xcc << 16);                                           \
            }                                                                                            \
        }                                                                                                \
    } while (0)
#else
#define STAGE_MARK(n) do { } while (0)
#endif

namespace {

constexpr int TW = 64, TH = 16;        // output tile of smooth_grad_kernel
constexpr int OW = 32, OH = 8;         // output tile of pyr_reduce_kernel
}
constexpr int OW_DEFAULT = 32, OH_DEFAULT = 8;
namespace {

__device__ __forceinline__ int reflect_idx(int i, int n)
{
    const int p = 2 * n;
    int m = i % p;
    if (m < 0) m += p;
    return m < n ? m : p - 1 - m;
}

// correlate1d at `c` (centre sample) of a fully populated LDS line with element stride `stride`
__device__ __forceinline__ float correlate_lds(const float *c, int stride, const Taps &t)
{
    const int size1 = t.n / 2;
    const int size2 = t.n - size1 - 1;
    const double *fw = t.k + size1;
    double acc;
    if (t.sym > 0) {
        acc = (double)c[0] * fw[0];
        for (int jj = -size1; jj < 0; jj++) acc = acc + ((double)c[jj * stride] + (double)c[-jj * stride]) * fw[jj];
    } else if (t.sym < 0) {
        acc = (double)c[0] * fw[0];
        for (int jj = -size1; jj < 0; jj++) acc = acc + ((double)c[jj * stride] - (double)c[-jj * stride]) * fw[jj];
    } else {
        acc = (double)c[size2 * stride] * fw[size2];
        for (int jj = -size1; jj < size2; jj++) acc = acc + (double)c[jj * stride] * fw[jj];
    }
    return (float)acc;
}

// ------------------------------------------------------------------------------------------------------
// frame -> [smoothed image] + gradx + grady.  grid = (ceil(ncols/TW), ceil(nrows/TH), batch), block = 256
template <typename TIn, bool SMOOTH>
__global__ __launch_bounds__(256) void smooth_grad_kernel(SmoothGradArgs a)
{
    extern __shared__ float lds[];
    const int tid = threadIdx.x;
    const int b = blockIdx.z;
    const int tx0 = blockIdx.x * TW, ty0 = blockIdx.y * TH;
    const int nc = a.ncols, nr = a.nrows;
    const int rs = SMOOTH ? a.smooth.n / 2 : 0;
    const int R = a.R;                                   // halo of the image tile = max gradient tap radius
    const int IW = TW + 2 * R, IH = TH + 2 * R;          // image tile
    const int RW = IW + 2 * rs, RH = IH + 2 * rs;        // raw tile
    // LDS carve: [A raw RH*RW][B hsmooth RH*IW] overlaid later by [D IH*TW][E IH*TW]; then [C image IH*IW]
    const int ab = SMOOTH ? RH * RW + RH * IW : 0, de = 2 * IH * TW;
    float *A = lds, *B = lds + RH * RW, *D = lds, *E = lds + IH * TW;
    float *C = lds + (ab > de ? ab : de);
    const TIn *__restrict__ raw = (const TIn *)a.raw[b];

    if (SMOOTH) {
        for (int i = tid; i < RH * RW; i += 256) {
            const int r = i / RW, c = i - r * RW;
            const int gy = reflect_idx(ty0 - R - rs + r, nr), gx = reflect_idx(tx0 - R - rs + c, nc);
            A[i] = (float)raw[(size_t)gy * nc + gx];
        }
        __syncthreads();
        for (int i = tid; i < RH * IW; i += 256) {       // horizontal smoothing
            const int r = i / IW, c = i - r * IW;
            B[i] = correlate_lds(A + r * RW + c + rs, 1, a.smooth);
        }
        __syncthreads();
        float *__restrict__ img = a.img[b];
        for (int i = tid; i < IH * IW; i += 256) {       // vertical smoothing -> image tile (+ store the interior)
            const int r = i / IW, c = i - r * IW;
            const float v = correlate_lds(B + (r + rs) * IW + c, IW, a.smooth);
            C[i] = v;
            const int y = ty0 - R + r, x = tx0 - R + c;
            if (r >= R && r < R + TH && c >= R && c < R + TW && y < nr && x < nc) img[(size_t)y * nc + x] = v;
        }
    } else {
        for (int i = tid; i < IH * IW; i += 256) {
            const int r = i / IW, c = i - r * IW;
            const int gy = reflect_idx(ty0 - R + r, nr), gx = reflect_idx(tx0 - R + c, nc);
            C[i] = (float)raw[(size_t)gy * nc + gx];
        }
    }
    __syncthreads();
    for (int i = tid; i < IH * TW; i += 256) {           // horizontal pass of both gradients
        const int r = i / TW, x = i - r * TW;
        const float *c = C + r * IW + x + R;
        D[i] = correlate_lds(c, 1, a.gderiv);            // gradx: derivative taps along x
        E[i] = correlate_lds(c, 1, a.ggauss);            // grady: Gaussian taps along x
    }
    __syncthreads();
    float *__restrict__ gxo = a.gx[b];
    float *__restrict__ gyo = a.gy[b];
    for (int i = tid; i < TH * TW; i += 256) {           // vertical pass -> coalesced stores
        const int r = i / TW, x = i - r * TW;
        const int y = ty0 + r, xx = tx0 + x;
        if (y < nr && xx < nc) {
            const size_t o = ((size_t)y * nc + xx) * a.gstride;      // gstride 2: gyo == gxo + 1 (interleaved planes)
            gxo[o] = correlate_lds(D + (r + R) * TW + x, TW, a.ggauss);
            gyo[o] = correlate_lds(E + (r + R) * TW + x, TW, a.gderiv);
        }
    }
}

// ------------------------------------------------------------------------------------------------------
// level l-1 image -> level l image: out(ys, xs) = V(H(src))(ss*ys + ss/2, ss*xs + ss/2)  (pyramid.py:59-72).
// Only the surviving columns are smoothed horizontally and only the surviving rows vertically.
// The source tile is stored de-interleaved by column phase (col % ss) so that lanes reading columns
// ss apart hit consecutive LDS addresses.  grid = (ceil(dc/OW), ceil(dr/OH), batch), block = 256
__global__ __launch_bounds__(256) void pyr_reduce_kernel(PyrReduceArgs a)
{
    extern __shared__ float lds[];
    const int tid = threadIdx.x;
    const int b = blockIdx.z;
    const int ss = a.ss, r = a.taps.n / 2;
    const int SW = (OW - 1) * ss + 2 * r + 1, SH = (OH - 1) * ss + 2 * r + 1;
    const int PW = (SW + ss - 1) / ss;                   // entries per phase plane per row
    const int rowlen = PW * ss;
    float *S = lds;                                      // SH rows x ss planes x PW
    float *Hh = lds + SH * rowlen;                       // SH x OW
    const int xs0 = blockIdx.x * OW, ys0 = blockIdx.y * OH;
    const int gx0 = xs0 * ss + ss / 2 - r, gy0 = ys0 * ss + ss / 2 - r;   // source coords of tile element (0,0)
    const float *__restrict__ src = a.src[b];
    const int nc = a.src_nc, nr = a.src_nr;

    for (int i = tid; i < SH * SW; i += 256) {
        const int rr = i / SW, c = i - rr * SW;
        const int gy = reflect_idx(gy0 + rr, nr), gx = reflect_idx(gx0 + c, nc);
        S[rr * rowlen + (c & (ss - 1)) * PW + (c >> a.log2ss)] = src[(size_t)gy * nc + gx];
    }
    __syncthreads();
    const int size1 = a.taps.n / 2, size2 = a.taps.n - size1 - 1;
    const double *fw = a.taps.k + size1;
    for (int i = tid; i < SH * OW; i += 256) {           // horizontal pass at the surviving columns
        const int rr = i / OW, xs = i - rr * OW;
        const float *row = S + rr * rowlen;
        const int cc = xs * ss + r;                      // tile column of the centre sample
        auto at = [&](int j) -> double { const int c = cc + j; return (double)row[(c & (ss - 1)) * PW + (c >> a.log2ss)]; };
        double acc;
        if (a.taps.sym > 0) {
            acc = at(0) * fw[0];
            for (int jj = -size1; jj < 0; jj++) acc = acc + (at(jj) + at(-jj)) * fw[jj];
        } else if (a.taps.sym < 0) {
            acc = at(0) * fw[0];
            for (int jj = -size1; jj < 0; jj++) acc = acc + (at(jj) - at(-jj)) * fw[jj];
        } else {
            acc = at(size2) * fw[size2];
            for (int jj = -size1; jj < size2; jj++) acc = acc + at(jj) * fw[jj];
        }
        Hh[i] = (float)acc;
    }
    __syncthreads();
    const int xs = tid % OW, ys = tid / OW;              // vertical pass at the surviving rows, one output per thread
    const int ox = xs0 + xs, oy = ys0 + ys;
    if (ox < a.dst_nc && oy < a.dst_nr)
        a.dst[b][(size_t)oy * a.dst_nc + ox] = correlate_lds(Hh + (ys * ss + r) * OW + xs, OW, a.taps);
}

// ======================================================================================================
// Compile-time specialisations.  Tap counts, tile geometry and every LDS offset are constants, the taps sit
// in scalar registers for the whole kernel and the correlate loops are fully unrolled (same operation order).
// The runtime-sized kernels above remain the fallback for unusual sigmas.

__device__ __forceinline__ int reflect_fast(int i, int n)
{
    if (i < 0) i = -1 - i;
    else if (i >= n) i = 2 * n - 1 - i;
    if ((unsigned)i >= (unsigned)n) i = reflect_idx(i, n);      // frames smaller than the halo
    return i;
}

template <int NT>
struct TapRegs { double k[NT]; };

template <int NT>
__device__ __forceinline__ void load_taps(TapRegs<NT> &r, const Taps &t)
{
#pragma unroll
    for (int i = 0; i < NT; i++) r.k[i] = t.k[i];
}

// Register-blocked variant (the one dispatched for the default sigmas).  Every thread produces 4 horizontally
// adjacent samples per step: LDS is read 16 bytes at a time (ds_read_b128, one aligned quad per lane), each
// sample is widened to f64 once per quad instead of once per tap, and the integer index math is amortised over
// four outputs.  The per-output FP64 expression and its operation order are unchanged.
// Column frames (relative to the tile's first output column): A (raw) starts at -8, B / C at -4, D / E at 0,
// all row strides are multiples of 4 floats.  Needs tap radii <= 4.
template <int NT, int SYM>
__device__ __forceinline__ float corr_regs(const double *c /* centre */, const TapRegs<NT> &t)
{
    constexpr int H = NT / 2;
    double acc = c[0] * t.k[H];
#pragma unroll
    for (int jj = -H; jj < 0; jj++) {
        const double pr = SYM > 0 ? c[jj] + c[-jj] : c[jj] - c[-jj];
        acc = acc + pr * t.k[H + jj];
    }
    return (float)acc;
}

__device__ __forceinline__ void widen4(const float4 v, double *d)
{
    d[0] = (double)v.x; d[1] = (double)v.y; d[2] = (double)v.z; d[3] = (double)v.w;
}

// Three aligned quads of an LDS row, widened.  The loads are volatile so that they stay three ds_read_b128: when only 8 or
// 10 of the 12 samples are used the compiler otherwise narrows them to ds_read2_b32 pairs, and dword reads at a lane stride
// of 16 bytes are 4-way bank conflicts.
typedef float f32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void widen12(const float *row, double *d)
{
    typedef const volatile __attribute__((address_space(3))) f32x4 *lds_quad_ptr;
    const lds_quad_ptr p = (lds_quad_ptr)row;
    const f32x4 a = p[0], b = p[1], c = p[2];
    d[0] = (double)a.x; d[1] = (double)a.y; d[2] = (double)a.z; d[3] = (double)a.w;
    d[4] = (double)b.x; d[5] = (double)b.y; d[6] = (double)b.z; d[7] = (double)b.w;
    d[8] = (double)c.x; d[9] = (double)c.y; d[10] = (double)c.z; d[11] = (double)c.w;
}

#ifndef KLT_L0_WAVES
#define KLT_L0_WAVES 4
#endif
// HRED: the kernel also runs the HORIZONTAL pass of the first pyramid reduction (subsampling 4, 21 taps) on the smoothed tile while it
// is in LDS, and writes H1[y][x] = hsmooth(image)(y, 4x + 2) (f32, nrows x ncols/4); pyr_vreduce_kernel finishes level 1 from
// H1.  The image tile carries a halo of 12 columns instead of 4 for that (+22 % smoothing work in this kernel), and the separate
// reduction kernel -- with its halo re-reads of the level-0 image, its 1.6x redundant horizontal pass and its latency-bound
// tile loads -- disappears for level 1.  Same values: the tile is addressed in virtual coordinates whose smoothed values equal
// the reflected ones (header of this file), which is what the reference's reduction reads beyond the frame edge.
// EDGE = false: the instantiation for tiles whose outputs all land inside the frame (every tile but the last row / column of tiles of
// a frame whose size is not a multiple of the tile): the per-row and per-column store predicates fold away.
template <typename TIn, bool SMOOTH, int NS, int NG, int ND, int TH_, int NTHR, bool HRED, bool EDGE>
__device__ __forceinline__ void smooth_grad_rb_tile(const SmoothGradArgs &a, float *const lds, const int nc, const int nr)
{
    static_assert(!HRED || SMOOTH, "the fused horizontal reduction needs the smoothing stages");
    constexpr int HB = HRED ? 12 : 4;                           // halo columns of the image tile (B, C) on each side
    constexpr int rs = SMOOTH ? NS / 2 : 0;
    constexpr int R = (NG > ND ? NG : ND) / 2;
    static_assert(rs <= 4 && R <= 4, "register-blocked kernel needs tap radii <= 4");
    constexpr int AW = TW + 2 * HB + 8, BW = TW + 2 * HB, DW = TW;   // floats per row (A starts at column -HB-4, B / C at -HB, D / E at 0)
    constexpr int AQ = AW / 4, BQ = BW / 4, DQ = DW / 4;        // quads per row
    constexpr int IH = TH_ + 2 * R, RH = IH + 2 * rs;
    // LDS regions.  With smoothing: [A | C] then [B | D E] -- the raw tile A is dead once B exists and the smoothed tile C takes its
    // place; B is dead once C exists and the two gradient intermediates D, E take its place (they are written while C is read,
    // so they cannot share C's region).  35.6 KB for the 32-row tile with the fused reduction: four workgroups per CU, and the
    // 2040 tiles of a 1080p pair are two full rounds of the 1024 slots (44 KB / three per CU before: 2.66 rounds).
    // Without smoothing (gradients of levels >= 1): C, then D E.
    constexpr int AC = SMOOTH ? (RH * AW > IH * BW ? RH * AW : IH * BW) : IH * BW;
    float *const A = lds, *const C = lds, *const B = lds + AC, *const D = lds + AC, *const E = lds + AC + IH * DW;
    const int tid = threadIdx.x, b = blockIdx.z;
    const int tx0 = blockIdx.x * TW, ty0 = blockIdx.y * TH_;
    // two adjacent outputs go out as one 8-byte store where every row of the planes keeps them aligned (block-uniform)
    const bool vec2_ok = (nc & 1) == 0 && (reinterpret_cast<uintptr_t>(a.img[b]) & 7) == 0;
    const TIn *__restrict__ raw = (const TIn *)a.raw[b];
    const unsigned row_bytes = 4u * (unsigned)nc;                // of the f32 planes
    const unsigned tile_b0 = (unsigned)ty0 * row_bytes + 4u * (unsigned)tx0;   // byte offset of the tile's first output in them
    TapRegs<NG> kg;
    TapRegs<ND> kd;
    load_taps(kg, a.ggauss);
    load_taps(kd, a.gderiv);

    STAGE_MARK(0);
    // ---- stage 0: frame -> LDS (through the reflect map)
    {
        constexpr int W0 = SMOOTH ? AQ : BQ, H0 = SMOOTH ? RH : IH, X0 = SMOOTH ? -(HB + 4) : -HB, Y0 = -(R + rs);
        float *const dst = SMOOTH ? A : C;
        constexpr int N0 = H0 * W0, U0 = (N0 + NTHR - 1) / NTHR;
        // Tiles whose halo lies inside the frame (most of them): every load of the thread is issued before the first one
        // is used.  The general loop below waits for each of its 3-4 loads in turn -- 2.2 of the 9 us a workgroup lives.
        const bool interior = (nc & 3) == 0 && tx0 + X0 >= 0 && tx0 + X0 + 4 * W0 <= nc && ty0 + Y0 >= 0 && ty0 + Y0 + H0 <= nr;
        if (interior) {
            const plane_rsrc rawp = plane_of(raw);
            const unsigned raw_b0 = ((unsigned)(ty0 + Y0) * (unsigned)nc + (unsigned)(tx0 + X0)) * (unsigned)sizeof(TIn);
            if (sizeof(TIn) == 1) {
                uint32_t w[U0];
#pragma unroll
                for (int u = 0; u < U0; u++) {
                    const int i = min(tid + u * NTHR, N0 - 1);          // clamped, unconditional (the last threads repeat a quad)
                    w[u] = __builtin_amdgcn_raw_buffer_load_b32(rawp, raw_b0 + __umul24((unsigned)(i / W0), (unsigned)nc) + 4u * (unsigned)(i % W0), 0, 0);
                }
#pragma unroll
                for (int u = 0; u < U0; u++) {
                    const int i = tid + u * NTHR;
                    float4 v;
                    v.x = (float)(w[u] & 0xffu); v.y = (float)((w[u] >> 8) & 0xffu);
                    v.z = (float)((w[u] >> 16) & 0xffu); v.w = (float)(w[u] >> 24);
                    if (i < N0) *reinterpret_cast<float4 *>(dst + (size_t)i * 4) = v;
                }
            } else {
                float4 w[U0];
#pragma unroll
                for (int u = 0; u < U0; u++) {
                    const int i = min(tid + u * NTHR, N0 - 1);
                    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
                    const u32x4 q = __builtin_amdgcn_raw_buffer_load_b128(rawp, raw_b0 + __umul24((unsigned)(i / W0), row_bytes) + 16u * (unsigned)(i % W0), 0, 0);
                    w[u] = __builtin_bit_cast(float4, q);
                }
#pragma unroll
                for (int u = 0; u < U0; u++) {
                    const int i = tid + u * NTHR;
                    if (i < N0) *reinterpret_cast<float4 *>(dst + (size_t)i * 4) = w[u];
                }
            }
        } else if (nc >= 2 * TW && nr >= 2 * TH_) {
            // frame-edge tiles of frames large enough that one reflection brings every index inside: branch-free index
            // map, all element loads of the thread in flight together
            TIn e[U0][4];
#pragma unroll
            for (int u = 0; u < U0; u++) {
                const int i = min(tid + u * NTHR, N0 - 1);
                const int y = ty0 + Y0 + i / W0, x = tx0 + X0 + 4 * (i % W0);
                const TIn *row = raw + (size_t)(y < 0 ? -1 - y : y >= nr ? 2 * nr - 1 - y : y) * nc;
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    const int xx = x + k;
                    e[u][k] = row[xx < 0 ? -1 - xx : xx >= nc ? 2 * nc - 1 - xx : xx];
                }
            }
#pragma unroll
            for (int u = 0; u < U0; u++) {
                const int i = tid + u * NTHR;
                float4 v;
                v.x = (float)e[u][0]; v.y = (float)e[u][1]; v.z = (float)e[u][2]; v.w = (float)e[u][3];
                if (i < N0) *reinterpret_cast<float4 *>(dst + (size_t)i * 4) = v;
            }
        } else
        for (int i = tid; i < H0 * W0; i += NTHR) {
            const int r = i / W0, q = i % W0;
            const int gy = reflect_fast(ty0 + Y0 + r, nr);
            const int x = tx0 + X0 + 4 * q;
            const TIn *row = raw + (size_t)gy * nc;
            float4 v;
            if (x >= 0 && x + 3 < nc && (nc & 3) == 0) {          // aligned quad: one 4- or 16-byte load
                if (sizeof(TIn) == 1) {
                    const uint32_t wq = *reinterpret_cast<const uint32_t *>(row + x);
                    v.x = (float)(wq & 0xffu); v.y = (float)((wq >> 8) & 0xffu);
                    v.z = (float)((wq >> 16) & 0xffu); v.w = (float)(wq >> 24);
                } else {
                    v = *reinterpret_cast<const float4 *>(row + x);
                }
            } else if (x >= 0 && x + 3 < nc) {
                v.x = (float)row[x]; v.y = (float)row[x + 1]; v.z = (float)row[x + 2]; v.w = (float)row[x + 3];
            } else {
                v.x = (float)row[reflect_fast(x, nc)]; v.y = (float)row[reflect_fast(x + 1, nc)];
                v.z = (float)row[reflect_fast(x + 2, nc)]; v.w = (float)row[reflect_fast(x + 3, nc)];
            }
            *reinterpret_cast<float4 *>(dst + (size_t)i * 4) = v;
        }
    }
    __syncthreads();
    STAGE_MARK(1);
    if (SMOOTH) {
        TapRegs<NS> ks;
        load_taps(ks, a.smooth);
        // ---- stage 1: horizontal smoothing, A -> B (B column c = A column c + 4)
        for (int i = tid; i < RH * BQ; i += NTHR) {
            const int r = i / BQ, q = i % BQ;
            double v[12];
            widen12(A + r * AW + 4 * q, v);
            float4 o;
            o.x = corr_regs<NS, 1>(v + 4, ks); o.y = corr_regs<NS, 1>(v + 5, ks);
            o.z = corr_regs<NS, 1>(v + 6, ks); o.w = corr_regs<NS, 1>(v + 7, ks);
            *reinterpret_cast<float4 *>(B + r * BW + 4 * q) = o;
        }
        __syncthreads();
        STAGE_MARK(2);
        // ---- stage 2: vertical smoothing, B -> C (+ store the tile interior of the smoothed image).
        // A thread produces two adjacent columns on FOUR consecutive rows: NS + 3 rows of two samples are read (ds_read_b64) and
        // widened for 8 outputs -- 2 widenings per output (a quad on two rows: 3).
        const plane_rsrc img = plane_of(a.img[b]);
        constexpr int BH = BW / 2, G2 = (IH + 3) / 4;           // half-quads per row, groups of four rows (the last one may be partial)
        for (int i = tid; i < G2 * BH; i += NTHR) {
            const int r = 4 * (i / BH), h = i % BH;
            // byte offset of (ty0 - R + r, tx0 - HB + 2 h) modulo 2^32 (rows above the frame are never stored)
            const unsigned img_b0 = tile_b0 + __umul24((unsigned)r, row_bytes) - (unsigned)R * row_bytes + (unsigned)(8 * h) - (unsigned)(4 * HB);
            double v[2][NS + 3];
#pragma unroll
            for (int j = 0; j < NS + 3; j++) {
                const int rj = min(r + j, RH - 1);              // rows past the tile only feed outputs that are not stored
                const float2 t = *reinterpret_cast<const float2 *>(B + rj * BW + 2 * h);
                v[0][j] = (double)t.x; v[1][j] = (double)t.y;
            }
#pragma unroll
            for (int dr = 0; dr < 4; dr++) {
                const int rr = r + dr;
                if (rr >= IH) break;
                float2 o;
                o.x = corr_regs<NS, 1>(v[0] + rs + dr, ks); o.y = corr_regs<NS, 1>(v[1] + rs + dr, ks);
                *reinterpret_cast<float2 *>(C + rr * BW + 2 * h) = o;
                const int y = ty0 - R + rr, x = tx0 - HB + 2 * h;
                if (rr >= R && rr < R + TH_ && h >= HB / 2 && h < HB / 2 + DW / 2 && (!EDGE || y < nr)) {
                    const unsigned ob = img_b0 + (unsigned)dr * row_bytes;          // byte offset of (y, x)
                    if (!EDGE && vec2_ok) plane_store2(img, ob, o);                  // x is even
                    else if (!EDGE) { plane_store(img, ob, o.x); plane_store(img, ob + 4, o.y); }
                    else {
                        if (x < nc) plane_store(img, ob, o.x);
                        if (x + 1 < nc) plane_store(img, ob + 4, o.y);
                    }
                }
            }
        }
        __syncthreads();
    }
    STAGE_MARK(3);
    // ---- stage 3: horizontal pass of both gradients, C -> D (derivative taps), E (Gaussian taps)
    // (the partial last round of this loop goes to wavefronts 2 and 3: stage 2 gave its partial round to wavefronts 0
    // and 1, and wavefront w of every workgroup of a CU sits on SIMD w -- the rotation evens out the SIMDs)
    for (int i = (tid + NTHR / 2) & (NTHR - 1); i < IH * DQ; i += NTHR) {
        const int r = i / DQ, q = i % DQ;
        double v[12];
        widen12(C + r * BW + 4 * q + (HB - 4), v);
        float4 d, e;
        d.x = corr_regs<ND, -1>(v + 4, kd); d.y = corr_regs<ND, -1>(v + 5, kd);
        d.z = corr_regs<ND, -1>(v + 6, kd); d.w = corr_regs<ND, -1>(v + 7, kd);
        e.x = corr_regs<NG, 1>(v + 4, kg); e.y = corr_regs<NG, 1>(v + 5, kg);
        e.z = corr_regs<NG, 1>(v + 6, kg); e.w = corr_regs<NG, 1>(v + 7, kg);
        *reinterpret_cast<float4 *>(D + r * DW + 4 * q) = d;
        *reinterpret_cast<float4 *>(E + r * DW + 4 * q) = e;
    }
    if (HRED) {
        // ---- stage 3b: horizontal pass of the pyramid reduction at the surviving columns 4x + 2 of the tile's own rows
        // (pyramid.py:59-72 -> correlate1d along x, symmetric branch; f32 result, as between the reference's two passes)
        constexpr int NR = 21, HR = NR / 2;
        TapRegs<HR + 1> kr;                                     // k[0..HR]: the taps left of and at the centre (symmetric)
#pragma unroll
        for (int t = 0; t <= HR; t++) kr.k[t] = a.reduce.k[t];
        const plane_rsrc h1 = plane_of(a.h1[b]);
        const int h1_nc = a.h1_nc;
        const unsigned h1_row_bytes = 4u * (unsigned)h1_nc, h1_b0 = (unsigned)ty0 * h1_row_bytes + (unsigned)tx0;   // (ty0, tx0 / 4); tx0 % 4 == 0
        // a thread makes TWO adjacent outputs (columns 4 xs + 2 and 4 xs + 6 of the tile): their 21-sample windows share 17
        // samples, so 28 samples are read and widened for two outputs instead of 24 for each
        static_assert(DQ % 2 == 0, "tile width must be a multiple of eight");
        for (int i = tid; i < TH_ * (DQ / 2); i += NTHR) {
            const int r = i / (DQ / 2), xs = 2 * (i % (DQ / 2));
            // centre = image column 4 xs + 2 = C index HB + 4 xs + 2; samples -10 .. +10 start at C index 4 xs + HB - 8 (a quad)
            typedef const volatile __attribute__((address_space(3))) f32x4 *lds_quad_ptr;
            const lds_quad_ptr p = (lds_quad_ptr)(C + (r + R) * BW + 4 * xs + (HB - 8));
            double v[28];
#pragma unroll
            for (int u = 0; u < 7; u++) {
                const f32x4 t = p[u];
                v[4 * u] = (double)t.x; v[4 * u + 1] = (double)t.y; v[4 * u + 2] = (double)t.z; v[4 * u + 3] = (double)t.w;
            }
            const int y = ty0 + r;
#pragma unroll
            for (int o = 0; o < 2; o++) {
                const double *c = v + HR + 4 * o;                    // centre sample of output o
                double acc = c[0] * kr.k[HR];
#pragma unroll
                for (int jj = -HR; jj < 0; jj++) acc = acc + (c[jj] + c[-jj]) * kr.k[HR + jj];
                const int xg = tx0 / 4 + xs + o;
                if (!EDGE || (y < nr && xg < h1_nc))
                    plane_store(h1, h1_b0 + __umul24((unsigned)r, h1_row_bytes) + 4u * (unsigned)(xs + o), (float)acc);
            }
        }
    }
    __syncthreads();
    STAGE_MARK(4);
    // ---- stage 4: vertical pass, D -> gradx (Gaussian taps), E -> grady (derivative taps).  A thread makes two adjacent columns on
    // FOUR consecutive rows: NG + 3 rows of two samples are read (ds_read_b64) and widened once for 8 outputs per plane -- 2.5
    // widenings per output where a quad on two rows needs 4 (the widening is an FP64-rate instruction like the adds and multiplies).
    const plane_rsrc gxy = plane_of(a.gx[b]);                    // the interleaved gradient plane: gradx, grady of a pixel side by side
    const bool vec4_ok = (nc & 1) == 0 && (reinterpret_cast<uintptr_t>(a.gx[b]) & 15) == 0;    // both pixels of a pair as one 16-byte store
    static_assert(TH_ % 4 == 0, "tile height must be a multiple of four");
    static_assert(NG == ND, "the vertical pass shares its row window between the two planes");
    constexpr int DH = DW / 2;                                   // half-quads per row
    for (int i = tid; i < (TH_ / 4) * DH; i += NTHR) {
        const int r = 4 * (i / DH), h = i % DH;
        const int x = tx0 + 2 * h;
        if (EDGE && (ty0 + r >= nr || x >= nc)) continue;
        float2 ox[4], oy[4];
        {
            double v[2][NG + 3];
#pragma unroll
            for (int j = 0; j < NG + 3; j++) {
                const float2 t = *reinterpret_cast<const float2 *>(D + (r + R - NG / 2 + j) * DW + 2 * h);
                v[0][j] = (double)t.x; v[1][j] = (double)t.y;
            }
#pragma unroll
            for (int dr = 0; dr < 4; dr++) {
                ox[dr].x = corr_regs<NG, 1>(v[0] + NG / 2 + dr, kg); ox[dr].y = corr_regs<NG, 1>(v[1] + NG / 2 + dr, kg);
            }
        }
        {
            double v[2][ND + 3];
#pragma unroll
            for (int j = 0; j < ND + 3; j++) {
                const float2 t = *reinterpret_cast<const float2 *>(E + (r + R - ND / 2 + j) * DW + 2 * h);
                v[0][j] = (double)t.x; v[1][j] = (double)t.y;
            }
#pragma unroll
            for (int dr = 0; dr < 4; dr++) {
                oy[dr].x = corr_regs<ND, -1>(v[0] + ND / 2 + dr, kd); oy[dr].y = corr_regs<ND, -1>(v[1] + ND / 2 + dr, kd);
            }
        }
        const unsigned g_b0 = 2u * (tile_b0 + __umul24((unsigned)r, row_bytes) + (unsigned)(8 * h));   // byte offset of (ty0 + r, x) in the interleaved plane
#pragma unroll
        for (int dr = 0; dr < 4; dr++) {
            const int y = ty0 + r + dr;
            if (EDGE && y >= nr) break;
            const unsigned ob = g_b0 + (unsigned)dr * 2u * row_bytes;
            float2 p0, p1;
            p0.x = ox[dr].x; p0.y = oy[dr].x; p1.x = ox[dr].y; p1.y = oy[dr].y;
            if (!EDGE && vec4_ok) plane_store4(gxy, ob, p0, p1);
            else {
                plane_store2(gxy, ob, p0);
                if (!EDGE || x + 1 < nc) plane_store2(gxy, ob + 8, p1);
            }
        }
    }
    STAGE_MARK(5);
}

template <typename TIn, bool SMOOTH, int NS, int NG, int ND, int TH_, int NTHR = 256, bool HRED = false>
__global__ __launch_bounds__(NTHR, KLT_L0_WAVES) void smooth_grad_rb(SmoothGradArgs a)
{
    constexpr int HB = HRED ? 12 : 4, rs = SMOOTH ? NS / 2 : 0, R = (NG > ND ? NG : ND) / 2;
    constexpr int AW = TW + 2 * HB + 8, BW = TW + 2 * HB, DW = TW, IH = TH_ + 2 * R, RH = IH + 2 * rs;
    constexpr int AC = SMOOTH ? (RH * AW > IH * BW ? RH * AW : IH * BW) : IH * BW;
    constexpr int BDE = SMOOTH ? (RH * BW > 2 * IH * DW ? RH * BW : 2 * IH * DW) : 2 * IH * DW;
    __shared__ __attribute__((aligned(16))) float lds[AC + BDE];
    const int b = blockIdx.z;
    const int tx0 = blockIdx.x * TW, ty0 = blockIdx.y * TH_;
    const int nc = a.dim_c[b] ? a.dim_c[b] : a.ncols, nr = a.dim_r[b] ? a.dim_r[b] : a.nrows;
    if (tx0 >= nc || ty0 >= nr) return;              // entries smaller than the grid extent (mixed pyramid levels)
    // every output of the tile inside the frame (and, with the fused reduction, inside the H1 plane): block-uniform
    const bool inside = tx0 + TW <= nc && ty0 + TH_ <= nr && (!HRED || tx0 / 4 + TW / 4 <= a.h1_nc);
    if (inside) smooth_grad_rb_tile<TIn, SMOOTH, NS, NG, ND, TH_, NTHR, HRED, false>(a, lds, nc, nr);
    else smooth_grad_rb_tile<TIn, SMOOTH, NS, NG, ND, TH_, NTHR, HRED, true>(a, lds, nc, nr);
}

// The f32-rounded horizontal result is kept in LDS as the double it widens to (one widening per sample instead of one
// per tap of the vertical pass); the source tile stays f32 (an f64 tile was measured slower: half the LDS matters more).
template <int NT, int STRIDE>
__device__ __forceinline__ float correlate_sym_f64(const double *c, const TapRegs<NT> &t)
{
    constexpr int H = NT / 2;
    double acc = c[0] * t.k[H];
#pragma unroll
    for (int jj = -H; jj < 0; jj++) acc = acc + (c[jj * STRIDE] + c[-jj * STRIDE]) * t.k[H + jj];
    return (float)acc;
}

// TS = element type of the source tile in LDS: double (widened once at load) or float (half the LDS, widened per tap)
template <int SS, int NT, int NTHR, typename TS, int OW = ::OW_DEFAULT, int OH = ::OH_DEFAULT>
__global__ __launch_bounds__(NTHR) void pyr_reduce_fast(PyrReduceArgs a)
{
    constexpr int r = NT / 2;
    constexpr int SW = (OW - 1) * SS + 2 * r + 1, SH = (OH - 1) * SS + 2 * r + 1;
    constexpr int PW = (SW + SS - 1) / SS, ROWLEN = PW * SS;
    extern __shared__ __attribute__((aligned(16))) double dlds[];
    double *const Hd = dlds;                                     // [SH][OW]
    TS *const S = reinterpret_cast<TS *>(dlds + SH * OW);        // [SH][SS planes][PW]
    const int tid = threadIdx.x, b = blockIdx.z;
    const int xs0 = blockIdx.x * OW, ys0 = blockIdx.y * OH;
    const int gx0 = xs0 * SS + SS / 2 - r, gy0 = ys0 * SS + SS / 2 - r;
    const float *__restrict__ src = a.src[b];
    const int nc = a.src_nc, nr = a.src_nr;
    TapRegs<NT> k;
    load_taps(k, a.taps);
    STAGE_MARK(0);

    // tile load: one aligned 16-byte quad per step where the row allows it (gx0 is a multiple of 4 columns)
    constexpr int SQ = (SW + 3) / 4;
    const bool quads = (nc & 3) == 0;
    // tiles inside the frame: all quads of the thread are requested before the first is used (the general loop waits for
    // each load in turn, and the tile load is 2.3 of the 3.7 us a workgroup lives)
    constexpr int NQ = SH * SQ, QPT = (NQ + NTHR - 1) / NTHR;
    const bool interior = quads && (gx0 & 3) == 0 && gx0 >= 0 && gy0 >= 0 && gx0 + 4 * SQ <= nc && gy0 + SH <= nr;
    if (interior) {
        const plane_rsrc srcp = plane_of(src);                           // raw buffer loads: 32-bit byte offsets
        const unsigned src_b0 = 4u * ((unsigned)gy0 * (unsigned)nc + (unsigned)gx0);
        float4 v[QPT];
#pragma unroll
        for (int u = 0; u < QPT; u++) {
            const int i = min(tid + u * NTHR, NQ - 1);                    // clamped: the last threads repeat the last quad
            const int rr = i / SQ, q = i - rr * SQ;
            typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
            const u32x4 w = __builtin_amdgcn_raw_buffer_load_b128(srcp, src_b0 + 4u * (__umul24((unsigned)rr, (unsigned)nc) + 4u * (unsigned)q), 0, 0);
            v[u] = __builtin_bit_cast(float4, w);
        }
#pragma unroll
        for (int u = 0; u < QPT; u++) {
            const int i = tid + u * NTHR;
            if (i < NQ) {
                const int rr = i / SQ, q = i - rr * SQ;
                TS *dst = S + rr * ROWLEN;
                const float e[4] = {v[u].x, v[u].y, v[u].z, v[u].w};
#pragma unroll
                for (int w = 0; w < 4; w++) {
                    const int c = 4 * q + w;
                    if (c < SW) dst[(c % SS) * PW + c / SS] = (TS)e[w];
                }
            }
        }
    } else if (nc >= 2 * SW && nr >= 2 * SH) {
        // frame-edge tiles of frames large enough that one reflection brings every index inside: branch-free index map,
        // every element load of the thread in flight together
        constexpr int NE = SH * SW, EPT = (NE + NTHR - 1) / NTHR;
        float e[EPT];
#pragma unroll
        for (int u = 0; u < EPT; u++) {
            const int i = min(tid + u * NTHR, NE - 1);
            const int y = gy0 + i / SW, x = gx0 + i % SW;
            e[u] = src[(size_t)(y < 0 ? -1 - y : y >= nr ? 2 * nr - 1 - y : y) * nc + (x < 0 ? -1 - x : x >= nc ? 2 * nc - 1 - x : x)];
        }
#pragma unroll
        for (int u = 0; u < EPT; u++) {
            const int i = tid + u * NTHR;
            const int rr = i / SW, c = i % SW;
            if (i < NE) S[rr * ROWLEN + (c % SS) * PW + c / SS] = (TS)e[u];
        }
    } else
    for (int i = tid; i < SH * SQ; i += NTHR) {
        const int rr = i / SQ, q = i % SQ;
        const int gy = reflect_fast(gy0 + rr, nr);
        const float *row = src + (size_t)gy * nc;
        const int x = gx0 + 4 * q;
        TS *dst = S + rr * ROWLEN;
        if (quads && x >= 0 && x + 3 < nc && 4 * q + 3 < SW) {
            const float4 v = *reinterpret_cast<const float4 *>(row + x);
            dst[((4 * q) % SS) * PW + (4 * q) / SS] = (TS)v.x;
            dst[((4 * q + 1) % SS) * PW + (4 * q + 1) / SS] = (TS)v.y;
            dst[((4 * q + 2) % SS) * PW + (4 * q + 2) / SS] = (TS)v.z;
            dst[((4 * q + 3) % SS) * PW + (4 * q + 3) / SS] = (TS)v.w;
        } else {
#pragma unroll
            for (int e = 0; e < 4; e++) {
                const int c = 4 * q + e;
                if (c < SW) dst[(c % SS) * PW + c / SS] = (TS)row[reflect_fast(gx0 + c, nc)];
            }
        }
    }
    __syncthreads();
    STAGE_MARK(1);
    for (int i = tid; i < SH * OW; i += NTHR) {
        const int rr = i / OW, xs = i % OW;
        const TS *row = S + rr * ROWLEN + xs;            // column xs*SS + r + j lives at plane (r+j)%SS, index xs + (r+j)/SS
        double acc = (double)row[(r % SS) * PW + r / SS] * k.k[r];
#pragma unroll
        for (int jj = -r; jj < 0; jj++) {
            const double lo = (double)row[((r + jj) % SS) * PW + (r + jj) / SS];
            const double hi = (double)row[((r - jj) % SS) * PW + (r - jj) / SS];
            acc = acc + (lo + hi) * k.k[r + jj];
        }
        Hd[i] = (double)(float)acc;                      // the f32 rounding between the passes
    }
    __syncthreads();
    STAGE_MARK(2);
    if (tid < OW * OH) {
        const int xs = tid % OW, ys = tid / OW;
        const int ox = xs0 + xs, oy = ys0 + ys;
        if (ox < a.dst_nc && oy < a.dst_nr)
            a.dst[b][(size_t)oy * a.dst_nc + ox] = correlate_sym_f64<NT, OW>(Hd + (ys * SS + r) * OW + xs, k);
    }
    STAGE_MARK(5);
}

template <int SS, int NT, typename TS, int OW, int OH>
constexpr size_t pyr_reduce_fast_lds()
{
    constexpr int r = NT / 2;
    constexpr int SW = (OW - 1) * SS + 2 * r + 1, SH = (OH - 1) * SS + 2 * r + 1;
    constexpr int PW = (SW + SS - 1) / SS;
    return sizeof(double) * (size_t)(SH * OW) + sizeof(TS) * (size_t)(SH * PW * SS);
}

template <int SS, int NT, int NTHR, typename TS, int OW, int OH>
static int launch_pyr_reduce_fast(hipStream_t s, const PyrReduceArgs &a, int batch)
{
    constexpr size_t l = pyr_reduce_fast_lds<SS, NT, TS, OW, OH>();
    if (int e = set_lds(pyr_reduce_fast<SS, NT, NTHR, TS, OW, OH>, l)) return e;
    const dim3 grid((a.dst_nc + OW - 1) / OW, (a.dst_nr + OH - 1) / OH, batch);
    klt_launch((pyr_reduce_fast<SS, NT, NTHR, TS, OW, OH>), grid, dim3(NTHR), (unsigned)l, s, a);
    return 0;
}

// ------------------------------------------------------------------------------------------------------
// Second half of the fused first reduction: level 1 (ys, xs) = V(H1)(4 ys + 2, xs), H1 = the horizontally reduced level-0 image
// written by smooth_grad_rb<..., HRED>.  Tile = 64 columns x 16 output rows; a thread makes 4 consecutive output rows of one
// column from 33 input rows (widened once).  Rows beyond the frame are read through the reflect map, as the reference does.
template <int SS, int NT>
__global__ __launch_bounds__(256) void pyr_vreduce_kernel(PyrReduceArgs a)
{
    constexpr int r = NT / 2, VW = 64, VH = 16, SH = (VH - 1) * SS + 2 * r + 1;       // 81 source rows
    __shared__ float T[SH * VW];
    const int tid = threadIdx.x, b = blockIdx.z;
    const int xs0 = blockIdx.x * VW, ys0 = blockIdx.y * VH;
    const int nc = a.dst_nc, nr = a.src_nr;                     // H1 is src_nr rows x dst_nc columns
    const plane_rsrc src = plane_of(a.src[b]);                  // raw buffer loads: 32-bit byte offsets, no 64-bit multiply-add per row
    TapRegs<r + 1> k;
#pragma unroll
    for (int t = 0; t <= r; t++) k.k[t] = a.taps.k[t];
    const int gy0 = ys0 * SS + SS / 2 - r;
    {
        constexpr int N = SH * VW, U = (N + 255) / 256;
        const int col = min(xs0 + (tid & 63), nc - 1);
        float v[U];
#pragma unroll
        for (int u = 0; u < U; u++) {                            // clamped, unconditional; all loads in flight together
            const int rr = min((tid >> 6) + 4 * u, SH - 1);
            int y = gy0 + rr;
            y = y < 0 ? -1 - y : y >= nr ? 2 * nr - 1 - y : y;
            y = min(max(y, 0), nr - 1);                          // (frames shorter than the halo never take this kernel)
            v[u] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(src, 4u * (__umul24((unsigned)y, (unsigned)nc) + (unsigned)col), 0, 0));
        }
#pragma unroll
        for (int u = 0; u < U; u++) {
            const int rr = (tid >> 6) + 4 * u;
            if (rr < SH) T[rr * VW + (tid & 63)] = v[u];
        }
    }
    __syncthreads();
    const int xs = tid & 63, yq = tid >> 6;                      // output rows ys0 + 4 yq .. + 3
    const plane_rsrc dst = plane_of(a.dst[b]);
    double v[3 * SS + 2 * r + 1];                                // 33 rows
#pragma unroll
    for (int j = 0; j < 3 * SS + 2 * r + 1; j++) v[j] = (double)T[(4 * yq * SS + j) * VW + xs];
#pragma unroll
    for (int o = 0; o < 4; o++) {
        const double *c = v + o * SS + r;
        double acc = c[0] * k.k[r];
#pragma unroll
        for (int jj = -r; jj < 0; jj++) acc = acc + (c[jj] + c[-jj]) * k.k[r + jj];
        const int oy = ys0 + 4 * yq + o, ox = xs0 + xs;
        if (oy < a.dst_nr && ox < a.dst_nc) plane_store(dst, 4u * (__umul24((unsigned)oy, (unsigned)a.dst_nc) + (unsigned)ox), (float)acc);
    }
}

// (A one-launch-per-level kernel -- vertical reduction from the previous level's H planes, the level's gradients and the next
// level's H planes in one workgroup -- existed in round 1 as KLT_OPT_FUSED_LEVELS: bit-identical, 12 us per level against 5 + 5 + 5
// for the three small launches, because its phases are serial and latency-bound inside a workgroup.  Removed in round 2.)
// (The last level's reduction inside the merged gradient launch -- 1024-thread workgroups, a 32x8 tile + halo of 3 reduced into LDS, halo
// positions beyond the frame reflected there, gradient passes on it; the gradients of the levels in between as further workgroups of
// the same launch -- was built and measured in round 2 as well: bit-identical (all parity tests), 10.5 us per launch against 4.9 + 5.3 for
// the two launches it replaces, 0.0567 vs 0.0557 ms per pair on one stream.  The launch boundary it saves (1.5-2 us) is less than
// what the longer serial chain of a workgroup costs; not kept.)

}  // namespace

size_t smooth_grad_lds_bytes(int rs, int R)
{
    const int IW = TW + 2 * R, IH = TH + 2 * R, RW = IW + 2 * rs, RH = IH + 2 * rs;
    const int ab = rs >= 0 ? RH * RW + RH * IW : 0, de = 2 * IH * TW;
    return sizeof(float) * (size_t)((ab > de ? ab : de) + IH * IW);
}

size_t pyr_reduce_lds_bytes(int ss, int ntaps)
{
    const int r = ntaps / 2;
    const int SW = (OW - 1) * ss + 2 * r + 1, SH = (OH - 1) * ss + 2 * r + 1;
    const int PW = (SW + ss - 1) / ss;
    return sizeof(float) * (size_t)(SH * PW * ss + SH * OW);
}

template <typename K>
static int set_lds(K kernel, size_t lds)
{
    if (lds > 48 * 1024) {
        hipError_t e = hipFuncSetAttribute((const void *)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return (int)e;
    }
    return 0;
}

// kind: 0 = u8 frame + smoothing, 1 = f32 frame + smoothing, 2 = f32 image, gradients only, 3 = u8 image, gradients only
// Is the fused horizontal reduction available for this launch?  (specialised taps, tall tiles, subsampling 4 with 21 taps,
// frames tall enough that the vertical pass needs a single reflection)
bool smooth_grad_hred_ok(const SmoothGradArgs &a, int batch, int kind, const Taps &reduce, int ss)
{
    static const int force_th = getenv("KLT_RB_TH") ? atoi(getenv("KLT_RB_TH")) : 0;
    const bool tall = force_th ? force_th == 32 : (long long)a.ncols * a.nrows * batch >= 1000000;
    return tall && kind < 2 && a.gstride == 2 && a.smooth.sym == 1 && (a.smooth.n == 5 || a.smooth.n == 9) &&
           a.ggauss.sym == 1 && a.gderiv.sym == -1 && a.ggauss.n == 7 && a.gderiv.n == 7 && ss == 4 && reduce.sym == 1 && reduce.n == 21 &&
           a.nrows >= 64 && a.ncols >= 64;
}

int launch_pyr_vreduce(hipStream_t s, const PyrReduceArgs &a, int batch)
{
    const dim3 grid((a.dst_nc + 63) / 64, (a.dst_nr + 15) / 16, batch);
    klt_launch((pyr_vreduce_kernel<4, 21>), grid, dim3(256), 0, s, a);
    return 0;
}

int launch_smooth_grad(hipStream_t s, const SmoothGradArgs &a, int batch, int kind, bool hred)
{
    const bool smooth = kind < 2;
    // compile-time specialisations: Gaussian smoothing (symmetric), Gaussian / derivative gradient taps
    if (a.gstride == 2 && a.ggauss.sym == 1 && a.gderiv.sym == -1 && a.ggauss.n == 7 && a.gderiv.n == 7 && (!smooth || a.smooth.sym == 1)) {
        const dim3 blk(256);
        // register-blocked kernels; small frames take the shorter tile so that the grid still covers the chip
        static const int force_th = getenv("KLT_RB_TH") ? atoi(getenv("KLT_RB_TH")) : 0;   // experiment hook
        const bool tall = force_th ? force_th == 32 : (long long)a.ncols * a.nrows * batch >= 1000000;
        const int th = tall ? 32 : 16;
        const dim3 g((a.ncols + TW - 1) / TW, (a.nrows + th - 1) / th, batch);
#define KLT_RB(T, SM, NSV)                                                                                        \
    do {                                                                                                          \
        if (tall) klt_launch((smooth_grad_rb<T, SM, NSV, 7, 7, 32>), g, blk, 0, s, a);                   \
        else klt_launch((smooth_grad_rb<T, SM, NSV, 7, 7, 16>), g, blk, 0, s, a);                        \
        return 0;                                                                                                 \
    } while (0)
        if (hred) {
            // (28- / 24- / 20-row tiles -- 32.0 / 28.4 / 24.8 KB of LDS, five / five / six workgroups per CU -- were measured in
            // round 2: 29.9 / 29.9 / 30.4 us event-timed against 28.1 for the 32-row tile: the extra halo work outweighs the occupancy)
            // (three / two workgroups per CU instead of four -- dynamic LDS padding, same kernel -- read 0.0379 / 0.0382 ms per pair with
            // three pairs in flight against 0.0365, and 0.0576 against 0.0556 on one stream: round 2)
            if (kind == 0 && a.smooth.n == 5) { klt_launch((smooth_grad_rb<uint8_t, true, 5, 7, 7, 32, 256, true>), g, blk, 0, s, a); return 0; }
            if (kind == 1 && a.smooth.n == 5) { klt_launch((smooth_grad_rb<float, true, 5, 7, 7, 32, 256, true>), g, blk, 0, s, a); return 0; }
            if (kind == 0 && a.smooth.n == 9) { klt_launch((smooth_grad_rb<uint8_t, true, 9, 7, 7, 32, 256, true>), g, blk, 0, s, a); return 0; }
            if (kind == 1 && a.smooth.n == 9) { klt_launch((smooth_grad_rb<float, true, 9, 7, 7, 32, 256, true>), g, blk, 0, s, a); return 0; }
            return -1;
        }
        if (kind == 0 && a.smooth.n == 5) KLT_RB(uint8_t, true, 5);
        if (kind == 1 && a.smooth.n == 5) KLT_RB(float, true, 5);
        if (kind == 0 && a.smooth.n == 9) KLT_RB(uint8_t, true, 9);
        if (kind == 1 && a.smooth.n == 9) KLT_RB(float, true, 9);
        if (kind == 2) KLT_RB(float, false, 1);
        if (kind == 3) KLT_RB(uint8_t, false, 1);
#undef KLT_RB
    }
    const size_t lds = smooth_grad_lds_bytes(smooth ? a.smooth.n / 2 : -1, a.R);
    const dim3 grid((a.ncols + TW - 1) / TW, (a.nrows + TH - 1) / TH, batch), block(256);
    int e = 0;
    switch (kind) {
    case 0: if ((e = set_lds(smooth_grad_kernel<uint8_t, true>, lds))) return e;
            klt_launch((smooth_grad_kernel<uint8_t, true>), grid, block, lds, s, a); break;
    case 1: if ((e = set_lds(smooth_grad_kernel<float, true>, lds))) return e;
            klt_launch((smooth_grad_kernel<float, true>), grid, block, lds, s, a); break;
    case 2: if ((e = set_lds(smooth_grad_kernel<float, false>, lds))) return e;
            klt_launch((smooth_grad_kernel<float, false>), grid, block, lds, s, a); break;
    default: if ((e = set_lds(smooth_grad_kernel<uint8_t, false>, lds))) return e;
            klt_launch((smooth_grad_kernel<uint8_t, false>), grid, block, lds, s, a); break;
    }
    return 0;
}

int launch_pyr_reduce(hipStream_t s, const PyrReduceArgs &a, int batch)
{
    const dim3 grid((a.dst_nc + OW - 1) / OW, (a.dst_nr + OH - 1) / OH, batch), block(256);
    if (a.taps.sym == 1) {
        // measured at cfg-2 (us per launch, two frames, both levels averaged): f32 tile / 1024 threads 11.6,
        // f64 tile / 1024 threads 12.4, f32 / 512 14.1, f64 / 512 15.4 (profiles/README.md)
        // tile shapes 64x8, 32x16, 32x4 and 16x16 were measured too: 32x8 is the fastest (profiles/README.md)
        if (a.ss == 4 && a.taps.n == 21) return launch_pyr_reduce_fast<4, 21, 1024, float, 32, 8>(s, a, batch);
        if (a.ss == 2 && a.taps.n == 11) return launch_pyr_reduce_fast<2, 11, 512, float, 32, 8>(s, a, batch);
    }
    const size_t lds = pyr_reduce_lds_bytes(a.ss, a.taps.n);
    if (int e = set_lds(pyr_reduce_kernel, lds)) return e;
    hipLaunchKernelGGL(pyr_reduce_kernel, grid, block, lds, s, a);
    return 0;
}
