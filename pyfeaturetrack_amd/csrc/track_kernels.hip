// Per-feature coarse-to-fine translational KLT tracker: one wavefront per feature, every pyramid
// level inside one launch (gfx950).
//
// Reference call chain replaced: KLTTrackFeatures per-feature loop (trackFeatures.py:250-346) ->
// _trackFeature (:67-136) -> extractImagePatchSlow (trackFeaturesUtils.pyx:14-51) and
// trackFeatureIterateCKLT (:393-459).  15 000 Python->C calls per 1080p pair become one kernel.
//
// Work split inside the wavefront (window w x w, n = w*w samples):
//   * lane l owns window samples l, l+64, ...; it keeps the image-1 template (intensity, gx, gy) of
//     those samples in registers for the whole level and re-samples image 2 each Newton iteration;
//   * every lane forms the five product terms of its samples (gx*gx, gx*gy, gy*gy, diff*gx, diff*gy) and
//     writes them to LDS; lanes 0..4 then add one array each in the reference's row-major sequential f32
//     order (trackFeaturesUtils.pyx:263-267, :296-302) and the five sums are broadcast with wave
//     shuffles; every lane solves the 2x2 system redundantly so the position stays wave-uniform;
//   * the residue test reproduces numpy's pairwise f32 sum (trackFeatures.py:124).
// The arithmetic mirrors the compiled reference exactly: bilinear weights in FP64
// except the ax*ay*I term which the reference evaluates in f32, products and sums un-fused, f32
// position updates, the in-loop bounds test with the integer half-window and the post-loop one with
// the Python-3 float half-window.
#include <cstdlib>

#include "klt_internal.h"

#pragma clang fp contract(off)

// tools/track_clocks.py builds this file with KLT_TRACK_CLOCKS into a private copy of the library: lane 0 of the first 256
// features of a launch records wall-clock ticks at the marks below (every mark first waits for all outstanding memory
// operations, so the build is for reading the time line, not for timing the kernel).
#ifdef KLT_TRACK_CLOCKS
__device__ long long g_tclk[256 * 32];
#define TCLK(i)                                                                                               \
    do {                                                                                                      \
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");                                            \
        if (threadIdx.x == 0 && blockIdx.x < 256 && (i) < 32) g_tclk[blockIdx.x * 32 + (i)] = wall_clock64();   \
    } while (0)
extern "C" int klt_debug_track_clocks(long long *out) { return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_tclk), sizeof(g_tclk)); }
#else
#define TCLK(i) do { } while (0)
#endif

namespace {

struct Bilinear {
    double w00, w01, w10;
    float w11;
    int ix, iy;
};

// trackFeaturesUtils.pyx:23-31, :44-47
__device__ __forceinline__ Bilinear make_bilinear(float x, float y)
{
    Bilinear b;
    b.ix = (int)x;
    b.iy = (int)y;
    const float ax = (float)((double)x - (double)b.ix);
    const float ay = (float)((double)y - (double)b.iy);
    b.w00 = (1. - (double)ax) * (1. - (double)ay);
    b.w01 = (double)ax * (1. - (double)ay);
    b.w10 = (1. - (double)ax) * (double)ay;
    b.w11 = ax * ay;
    return b;
}

// ST = element stride of the plane: 1 for an image plane, KLT_GRAD_STRIDE for one of the two interleaved gradient planes (qg then
// points at the plane's own first element: gradx at +0, grady at +1)
template <int ST = 1>
__device__ __forceinline__ float sample(const float *__restrict__ qg, int nc, const Bilinear &b)
{
    // the planes live in device memory: global loads (a flat load also counts as an LDS operation)
    const __attribute__((address_space(1))) float *q = (const __attribute__((address_space(1))) float *)qg;
    const float t4 = b.w11 * q[ST * (nc + 1)];
    double v = b.w00 * (double)q[0];
    v = v + b.w01 * (double)q[ST];
    v = v + b.w10 * (double)q[ST * nc];
    v = v + (double)t4;
    return (float)v;
}

// numpy's pairwise summation of n floats in LDS (trackFeatures.py:124), computed by the whole wavefront; the result is valid in
// lane 0.  For a block of 8 <= n <= 128 numpy keeps eight running sums r[j] = a[j] + a[8 + j] + a[16 + j] + ... (each a
// sequential chain, independent of the others), folds them as ((r0 + r1) + (r2 + r3)) + ((r4 + r5) + (r6 + r7)) and then adds
// the n % 8 tail one by one.  Lanes 0..7 run the eight chains side by side and three shuffles do the fold: the same
// additions in the same order as the 1-lane loop, in 5 + 3 + tail steps instead of n.
__device__ __forceinline__ float pairwise_block_wave(const float *a, int n, int lane)
{
    if (n < 8) {
        float res = 0.f;
        for (int i = 0; i < n; i++) res = res + a[i];
        return res;
    }
    const int nn = n - (n % 8);
    float r = 0.f;
    if (lane < 8) {
        r = a[lane];
        for (int i = 8; i < nn; i += 8) r = r + a[i + lane];
    }
    r = r + __shfl_down(r, 1);          // lanes 0, 2, 4, 6: r0 + r1, r2 + r3, r4 + r5, r6 + r7
    r = r + __shfl_down(r, 2);          // lanes 0, 4
    float res = r + __shfl_down(r, 4);  // lane 0
    for (int i = nn; i < n; i++) res = res + a[i];
    return res;
}

template <int DEPTH>
__device__ float pairwise_sum(const float *a, int n, int lane)
{
    if (n <= 128) return pairwise_block_wave(a, n, lane);
    int n2 = n / 2;
    n2 -= n2 % 8;
    return pairwise_sum<DEPTH - 1>(a, n2, lane) + pairwise_sum<DEPTH - 1>(a + n2, n - n2, lane);
}
template <>
__device__ float pairwise_sum<0>(const float *a, int n, int lane)
{
    return pairwise_block_wave(a, n < 128 ? n : 128, lane);
}

// A level of a pair's descriptor table.  The table is written by the host before the launch and never by a kernel, so it is read
// through the constant address space: scalar loads, the six plane pointers stay in SGPRs (as they do for a single-pair launch, whose
// levels travel in the kernarg segment).  Read as generic memory the pointers arrive in VGPRs (the compiler cannot rule out that the
// feature stores alias the table): 12 more vector registers, and flat loads through them.
__device__ __forceinline__ TrackLevel load_level(const TrackLevel *p)
{
    typedef const __attribute__((address_space(4))) TrackLevel *cptr;
    const cptr c = (cptr)p;
    TrackLevel lv;
    lv.i1 = c->i1; lv.gx1 = c->gx1; lv.gy1 = c->gy1;
    lv.i2 = c->i2; lv.gx2 = c->gx2; lv.gy2 = c->gy2;
    lv.nc = c->nc; lv.nr = c->nr;
    return lv;
}

__device__ __forceinline__ void wave_lds_sync() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }

// _trackFeature for one level.  Returns the status; x2/y2 updated in place; `iters` = Newton iterations.
// WCT > 0: window size known at compile time (index math folds, the summation loops unroll and read LDS
// 16 bytes at a time); WCT == 0: any odd window up to 31.
template <int MAXK, int WCT>
__device__ int track_level(const TrackArgs &a, const TrackLevel &lv, float x1, float y1, float &x2r, float &y2r,
                           float *lds, int lane, int &iters, int clk0 = 0)
{
    const int w = WCT > 0 ? WCT : a.window, n = w * w, hw = w / 2;
    const int npad = (n + 3) & ~3;                       // 16-byte aligned sub-arrays
    const int nc = lv.nc, nr = lv.nr;
    float *l_diff = lds;                                 // residue scratch (aliases product array 0)
    iters = 0;

    // image-1 template (trackFeatures.py:102-104)
    const Bilinear b1 = make_bilinear(x1, y1);
    if (!(b1.ix - hw >= 0 && b1.iy - hw >= 0 && b1.ix + hw + 2 <= nc && b1.iy + hw + 2 <= nr))
        return KLT_OOB;      // the reference asserts here (trackFeaturesUtils.pyx:35); see DESIGN.md
    float t_i[MAXK], t_gx[MAXK], t_gy[MAXK];
    int off[MAXK];           // sample offset relative to the window's top-left footprint pixel
#pragma unroll
    for (int kk = 0; kk < MAXK; kk++) {
        const int k = lane + 64 * kk;
        t_i[kk] = t_gx[kk] = t_gy[kk] = 0.f;
        off[kk] = 0;
        if (k < n) {
            off[kk] = (k / w) * nc + (k % w);
            const size_t q = (size_t)(b1.iy - hw) * nc + (b1.ix - hw) + off[kk];
            t_i[kk] = sample(lv.i1 + q, nc, b1);
            t_gx[kk] = sample<KLT_GRAD_STRIDE>(lv.gx1 + KLT_GRAD_STRIDE * q, nc, b1);
            t_gy[kk] = sample<KLT_GRAD_STRIDE>(lv.gy1 + KLT_GRAD_STRIDE * q, nc, b1);
        }
    }

    TCLK(clk0);                                          // template sampled
    float x2 = x2r, y2 = y2r;
    int status;
    const float one_plus_eps = 1.001f;
    for (;;) {
        // trackFeaturesUtils.pyx:428-431 (integer half-window, f32 arithmetic)
        if ((double)(x2 - (float)hw) < 0. || (float)nc - (x2 + (float)hw) < one_plus_eps ||
            (double)(y2 - (float)hw) < 0. || (float)nr - (y2 + (float)hw) < one_plus_eps) {
            status = KLT_OOB;
            break;
        }
        const Bilinear b2 = make_bilinear(x2, y2);
        const size_t base = (size_t)(b2.iy - hw) * nc + (b2.ix - hw);
        // every lane forms the five products of its samples (each product is one rounded f32 multiply, exactly the
        // term the reference adds); LDS then holds five arrays of n terms
#pragma unroll
        for (int kk = 0; kk < MAXK; kk++) {
            const int k = lane + 64 * kk;
            if (k < n) {
                const size_t q = base + off[kk];
                const float diff = t_i[kk] - sample(lv.i2 + q, nc, b2);          // :82-85
                const float sx = t_gx[kk] + sample<KLT_GRAD_STRIDE>(lv.gx2 + KLT_GRAD_STRIDE * q, nc, b2);          // -( -g1 - g2 ), :128 and :297
                const float sy = t_gy[kk] + sample<KLT_GRAD_STRIDE>(lv.gy2 + KLT_GRAD_STRIDE * q, nc, b2);
                lds[k] = sx * sx;                  // gxx terms, :299
                lds[npad + k] = sx * sy;           // gxy terms, :300
                lds[2 * npad + k] = sy * sy;       // gyy terms, :301
                lds[3 * npad + k] = diff * sx;     // ex terms,  :265
                lds[4 * npad + k] = diff * sy;     // ey terms,  :266
            }
        }
        __syncthreads();
        if (iters < 3) TCLK(clk0 + 1 + 2 * iters);       // image 2 sampled, products in LDS
        // lanes 0..4 each add one array in the reference's row-major order (sequential f32 adds)
        float acc = 0.f;
        if (lane < 5) {
            const float *T = lds + lane * npad;
            if (WCT > 0) {
                const float4 *T4 = reinterpret_cast<const float4 *>(T);
#pragma unroll WCT <= 8 ? 16 : 4
                for (int q = 0; q < (WCT * WCT + 3) / 4; q++) {
                    const float4 v = T4[q];
                    acc = acc + v.x;
                    if (4 * q + 1 < WCT * WCT) acc = acc + v.y;
                    if (4 * q + 2 < WCT * WCT) acc = acc + v.z;
                    if (4 * q + 3 < WCT * WCT) acc = acc + v.w;
                }
            } else {
                for (int k = 0; k < n; k++) acc = acc + T[k];
            }
        }
        __syncthreads();
        const float gxx = __shfl(acc, 0), gxy = __shfl(acc, 1), gyy = __shfl(acc, 2);
        const float ex = __shfl(acc, 3) * a.step, ey = __shfl(acc, 4) * a.step;
        // _solveEquation, :318-340
        const float p1 = gxx * gyy, p2 = gxy * gxy;
        const float det = p1 - p2;
        if (det < a.small) { status = KLT_SMALL_DET; break; }
        const float n1 = gyy * ex, n2 = gxy * ey, n3 = gxx * ey, n4 = gxy * ex;
        const float dx = (n1 - n2) / det;
        const float dy = (n3 - n4) / det;
        status = KLT_TRACKED;
        x2 = x2 + dx;
        y2 = y2 + dy;
        if (iters < 3) TCLK(clk0 + 2 + 2 * iters);       // sums, solve, update
        iters++;
        if (!((fabsf(dx) >= a.th || fabsf(dy) >= a.th) && iters < a.max_iterations)) break;
    }
    x2r = x2;
    y2r = y2;
    TCLK(clk0 + 7);                                      // Newton loop left

    // trackFeatures.py:110 -- Python floats: half-window 3.5, eps 1.001 as doubles
    const double x2d = (double)x2, y2d = (double)y2, hwd = a.half_window;
    if (x2d - hwd < 0.0 || (double)nc - (x2d + hwd) < 1.001 || y2d - hwd < 0.0 || (double)nr - (y2d + hwd) < 1.001)
        status = KLT_OOB;

    // residue, trackFeatures.py:118-125
    if (status == KLT_TRACKED && a.use_max_residue) {
        const Bilinear b2 = make_bilinear(x2, y2);
        const size_t base = (size_t)(b2.iy - hw) * nc + (b2.ix - hw);
#pragma unroll
        for (int kk = 0; kk < MAXK; kk++) {
            const int k = lane + 64 * kk;
            if (k < n) l_diff[k] = fabsf(t_i[kk] - sample(lv.i2 + base + off[kk], nc, b2));
        }
        __syncthreads();
        float s = pairwise_sum<3>(l_diff, n, lane);
        __syncthreads();
        s = __shfl(s, 0);
        if (s / (float)n > a.max_residue) status = KLT_LARGE_RESIDUE;
    }
    TCLK(clk0 + 8);                                      // residue test done

    if (a.retain) return KLT_TRACKED;                                   // :127-129
    if (status == KLT_SMALL_DET || status == KLT_OOB || status == KLT_LARGE_RESIDUE) return status;
    if (iters >= a.max_iterations) return KLT_MAX_ITERATIONS;
    return KLT_TRACKED;
}

template <int MAXK, int WCT, bool BATCH>
__global__ __launch_bounds__(64) void track_kernel(TrackArgs a)
{
    extern __shared__ __attribute__((aligned(16))) float lds[];
    int f = blockIdx.x;
    if (a.order) {
        // XCD-aware order: workgroup (x, y) runs on XCD x % 8 (gridDim.x is a multiple of 8); XCD c takes the c-th eighth of the
        // features sorted by row, so that the eight L2s each see one band of the pyramids instead of all of every plane
        const uint32_t *ord = a.order + (BATCH ? (size_t)blockIdx.y * a.n : (size_t)0);
        const int c = blockIdx.x & 7, j = blockIdx.x >> 3, pos = c * a.order_chunk + j;
        if (j >= a.order_chunk || pos >= a.n) return;
        f = (int)ord[pos];
    }
    const int lane = threadIdx.x;
    if (f >= a.n) return;
    const TrackLevel *levels = BATCH ? a.pairs[blockIdx.y].lv : a.lv;
    const klt_feat *fin = BATCH ? a.pairs[blockIdx.y].in : a.in;
    klt_feat *fout = BATCH ? a.pairs[blockIdx.y].out : a.out;
    const klt_feat ft = fin[f];
    if (ft.val < 0) {                       // only live features are tracked, trackFeatures.py:253
        if (lane == 0) fout[f] = ft;
        return;
    }
    const int L = a.nlevels;
    TCLK(0);
    // trackFeatures.py:255-265: position at the coarsest resolution (divisions by a power of two: exact)
    float xloc = ft.x, yloc = ft.y;
    for (int r = 0; r < L; r++) { xloc = xloc * a.inv_ss; yloc = yloc * a.inv_ss; }   // power of two: exact
    float xout = xloc, yout = yloc;
    int val = KLT_TRACKED;
    uint32_t aux = 0;       // 4 bits per level: 0 = level not visited, v = v-1 Newton iterations (saturating at 14)
    for (int r = L - 1; r >= 0; r--) {
        xloc = xloc * a.ss; yloc = yloc * a.ss; xout = xout * a.ss; yout = yout * a.ss;
        int it = 0;
        const TrackLevel lv = BATCH ? load_level(levels + r) : levels[r];
        val = track_level<MAXK, WCT>(a, lv, xloc, yloc, xout, yout, lds, lane, it, 1 + 9 * (L - 1 - r));
        aux |= (uint32_t)(it < 14 ? it + 1 : 15) << (4 * r);      // visited level r with `it` Newton iterations
        if (val == KLT_SMALL_DET || val == KLT_OOB) break;             // :284-285
    }
    if (lane == 0) {
        klt_feat o;
        o.aux = (int32_t)aux;
        const double xd = (double)xout, yd = (double)yout;
        const bool oob = val == KLT_OOB ||
                         xd < a.borderx || xd > (double)(a.ncols - 1) - a.borderx ||
                         yd < a.bordery || yd > (double)(a.nrows - 1) - a.bordery;   // :288-308
        if (oob) { o.x = -1.f; o.y = -1.f; o.val = KLT_OOB; }
        else if (val == KLT_SMALL_DET || val == KLT_LARGE_RESIDUE || val == KLT_MAX_ITERATIONS) {
            o.x = -1.f; o.y = -1.f; o.val = val;
        } else { o.x = xout; o.y = yout; o.val = KLT_TRACKED; }
        fout[f] = o;
    }
}

// ------------------------------------------------------------------------------------------------------
// Quad-load tracker kernels.  The (w+1) x (w+1) footprint of a window is loaded as QUADS of four pixels, one 16-byte load per
// lane and image:
//   7x7   the 8x8 footprint is 16 quads -> a feature owns 16 lanes and a wavefront tracks FOUR features in lock step under
//         per-feature predicates (a finished feature keeps its state and idles);
//   15x15 the 16x16 footprint is exactly 64 quads -> one feature per wavefront.
// Lane (row r, quad h) of a feature loads pixels (r, 4h .. 4h + 3) and computes the window samples (r, 4h .. 4h + 3) from its own
// quad, the quad below (lane + quads-per-row) and the first pixels of the quads to the right (lane + 1, lane + quads-per-row + 1):
// 3 vector loads per footprint and wavefront, where the one-sample-per-lane kernel issues 12 (7x7) or 48 (15x15).  The bounds test
// and the footprint of the NEXT Newton iteration are issued as soon as the position is known -- for the first iteration of a
// level together with the template's loads -- so a level costs one memory round trip per iteration instead of one more.
// Every feature's arithmetic is track_level's, operation for operation: same bilinear expression, the five product arrays in
// LDS added by lanes 0..4 of the feature in the reference's row-major sequential f32 order, numpy's pairwise sum for the residue.

// numpy's pairwise f32 sum of a block of n <= 128 floats, by the lanes s = 0..7 of a feature's lane group (pairwise_block_wave
// with the group lane); the result is valid in the group's lane s == 0
__device__ __forceinline__ float pairwise_block_group(const float *a, int n, int s)
{
    if (n < 8) {
        float res = 0.f;
        for (int i = 0; i < n; i++) res = res + a[i];
        return res;
    }
    const int nn = n - (n % 8);
    float r = 0.f;
    if (s < 8) {
        r = a[s];
        for (int i = 8; i < nn; i += 8) r = r + a[i + s];
    }
    r = r + __shfl_down(r, 1);
    r = r + __shfl_down(r, 2);
    float res = r + __shfl_down(r, 4);
    for (int i = nn; i < n; i++) res = res + a[i];
    return res;
}

// ... of any n: numpy halves blocks of more than 128 elements (n2 = n / 2 rounded down to a multiple of 8)
template <int DEPTH>
__device__ __forceinline__ float pairwise_group(const float *a, int n, int s)
{
    if (n <= 128) return pairwise_block_group(a, n, s);
    int n2 = n / 2;
    n2 -= n2 % 8;
    return pairwise_group<DEPTH - 1>(a, n2, s) + pairwise_group<DEPTH - 1>(a + n2, n - n2, s);
}
template <>
__device__ __forceinline__ float pairwise_group<0>(const float *a, int n, int s)
{
    return pairwise_block_group(a, n < 128 ? n : 128, s);
}

typedef float f32x4 __attribute__((ext_vector_type(4)));

// 4-byte aligned 16-byte load of elements q .. q + 3 of a plane, as a raw buffer load: the plane pointer is wavefront-uniform, so the
// descriptor sits in four SGPRs and the lane's 32-bit byte offset is the whole vector address (a plane is far below 2 GB).  A global
// load of plane + q costs a 64-bit vector add per load and a register pair for the address; a flat load (what a pointer read from
// generic memory gives) also counts as an LDS operation.  Word 3 of the descriptor: data format 32 bits, nothing else (raw dwords).
__device__ __forceinline__ f32x4 load_quad(const float *plane, unsigned q)
{
    const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void *)plane, 0, 0x7fffffff, 0x00020000);
    return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, q << 2, 0, 0));
}

// the gradient quads of pixels q .. q + 3: the two planes are interleaved (gxy = the gradx plane's pointer, grady one element behind),
// so the eight values are 32 contiguous bytes -- two 16-byte loads, as for two separate planes, but ONE piece of memory per footprint row
__device__ __forceinline__ void load_grad_quads(const float *gxy, unsigned q, f32x4 &gx, f32x4 &gy)
{
    const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void *)gxy, 0, 0x7fffffff, 0x00020000);
    const f32x4 a = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, q << 3, 0, 0));
    const f32x4 b = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, (q << 3) + 16u, 0, 0));
    gx.x = a.x; gx.y = a.z; gx.z = b.x; gx.w = b.z;
    gy.x = a.y; gy.y = a.w; gy.z = b.y; gy.w = b.w;
}

// the four window samples of a lane from its quad `a`: pairs (a.x,a.y), (a.y,a.z), (a.z,a.w), (a.w, right neighbour) and the same
// pairs of the row below (QPR = quads per footprint row: the lane holding the quad below is QPR lanes up)
template <int QPR>
__device__ __forceinline__ void sample_quad(const f32x4 a, const Bilinear &b, float out[4])
{
    f32x4 lo;
    lo.x = __shfl_down(a.x, QPR); lo.y = __shfl_down(a.y, QPR); lo.z = __shfl_down(a.z, QPR); lo.w = __shfl_down(a.w, QPR);
    const float rx = __shfl_down(a.x, 1), dx = __shfl_down(a.x, QPR + 1);
    const float v00[4] = {a.x, a.y, a.z, a.w}, v01[4] = {a.y, a.z, a.w, rx};
    const float v10[4] = {lo.x, lo.y, lo.z, lo.w}, v11[4] = {lo.y, lo.z, lo.w, dx};
#pragma unroll
    for (int m = 0; m < 4; m++) {
        const float t4 = b.w11 * v11[m];
        double d = b.w00 * (double)v00[m];
        d = d + b.w01 * (double)v01[m];
        d = d + b.w10 * (double)v10[m];
        d = d + (double)t4;
        out[m] = (float)d;
    }
}

// KLT_OPT_TRACK_TREE_SUMS: the sum of one value per lane over the LPF lanes of a feature, by a butterfly in registers -- within a
// row of 16 lanes four v_add_f32 with DPP operands (quad_perm [1,0,3,2] and [2,3,0,1], row_half_mirror, row_mirror: after each step
// the lanes that were combined hold the same partial sum, so mirroring pairs what xor would pair), across rows ds_bpermute.  Every
// lane ends up with the total.  Same precision as the sequential f32 chain, other ORDER of the additions: not the reference's bits.
template <int LPF>
__device__ __forceinline__ float group_tree_sum(float v)
{
    v = v + __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true));
    v = v + __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, true));
    v = v + __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xF, 0xF, true));
    v = v + __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x140, 0xF, 0xF, true));
    if (LPF > 16) v = v + __shfl_xor(v, 16);
    if (LPF > 32) v = v + __shfl_xor(v, 32);
    return v;
}

template <bool BATCH, int W, int WAVES = 1, bool TREE = false>
__global__ __launch_bounds__(64, WAVES) void track_kernel_quad(TrackArgs a)
{
    static_assert(W == 7 || W == 15, "quad kernels exist for 7x7 and 15x15 windows");
    constexpr int FPW = W == 7 ? 4 : 1;                      // features per wavefront
    constexpr int LPF = 64 / FPW;                            // lanes per feature = quads of its footprint
    constexpr int QPR = (W + 1) / 4;                         // quads per footprint row
    constexpr int w = W, n = W * W, hw = W / 2, npad = (n + 3) & ~3;
    constexpr bool REUSE = W == 7;                           // keep a footprint whose integer corner has not moved (see request_footprint)
    static_assert((W + 1) * QPR == LPF, "one quad per lane");
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int lane = threadIdx.x, g = lane / LPF, s = lane % LPF, glead = lane - s;
    int f = FPW * blockIdx.x + g;
    if (a.order) {
        // XCD-aware order (KLT_OPT_TRACK_XCD_ORDER): workgroups go to the XCDs round-robin in linear order and gridDim.x is a
        // multiple of 8, so workgroup (x, y) runs on XCD x % 8; it takes FPW consecutive features of that XCD's band of the
        // row-sorted list of its pair (batched launches: pair blockIdx.y, permutation blockIdx.y of the table)
        const uint32_t *ord = a.order + (BATCH ? (size_t)blockIdx.y * a.n : (size_t)0);
        const int c = blockIdx.x & 7, j = FPW * (blockIdx.x >> 3) + g, pos = c * a.order_chunk + j;
        f = (j < a.order_chunk && pos < a.n) ? (int)ord[pos] : a.n;
    }
    const TrackLevel *levels = BATCH ? a.pairs[blockIdx.y].lv : a.lv;
    const klt_feat *fin = BATCH ? a.pairs[blockIdx.y].in : a.in;
    klt_feat *fout = BATCH ? a.pairs[blockIdx.y].out : a.out;
    const bool valid = f < a.n;
    const klt_feat ft = fin[valid ? f : a.n - 1];
    const bool tracked_feature = valid && ft.val >= 0;       // only live features are tracked, trackFeatures.py:253
    if (valid && ft.val < 0 && s == 0) fout[f] = ft;
    if (!__any(tracked_feature)) return;
    const int L = a.nlevels;
    float *const gl = lds + g * 5 * npad;                    // this feature's five product arrays
    const int qr = s / QPR, qh = s % QPR;                    // my quad: footprint row qr, columns 4 qh .. 4 qh + 3
    const int k0 = qr * w + 4 * qh;                          // window index of my first sample (qr, 4 qh)
    const float one_plus_eps = 1.001f;

    // trackFeatures.py:255-265
    float xloc = ft.x, yloc = ft.y;
    for (int r = 0; r < L; r++) { xloc = xloc * a.inv_ss; yloc = yloc * a.inv_ss; }
    float xout = xloc, yout = yloc;
    int val = KLT_TRACKED;
    uint32_t aux = 0;
    bool alive = tracked_feature;                            // still descending the pyramid

    for (int r = L - 1; r >= 0; r--) {
        if (!__any(alive)) break;
        const TrackLevel lv = BATCH ? load_level(levels + r) : levels[r];
        const int nc = lv.nc, nr = lv.nr;
        if (alive) { xloc = xloc * a.ss; yloc = yloc * a.ss; xout = xout * a.ss; yout = yout * a.ss; }

        // image-1 template (trackFeatures.py:102-104); a window that leaves image 1 ends the feature
        const Bilinear b1 = make_bilinear(xloc, yloc);
        const bool t_ok = b1.ix - hw >= 0 && b1.iy - hw >= 0 && b1.ix + hw + 2 <= nc && b1.iy + hw + 2 <= nr;
        const bool run = alive && t_ok;
        // (row * nc as a 24-bit multiply: both are far below 2^24, and the 32-bit integer multiply is a quarter-rate instruction)
        const unsigned q1 = run ? __umul24((unsigned)(b1.iy - hw + qr), (unsigned)nc) + (unsigned)(b1.ix - hw + 4 * qh) : 0u;   // 32-bit element offsets: scalar base + vector offset loads
        const f32x4 t_qi = load_quad(lv.i1, q1);
        f32x4 t_qgx, t_qgy;
        load_grad_quads(lv.gx1, q1, t_qgx, t_qgy);

        // the first Newton iteration starts from a position that is already known: its bounds test (trackFeaturesUtils.pyx:428-431)
        // and its footprint loads go out now, behind the template's
        int it = 0, status = KLT_OOB;
        float x2 = xout, y2 = yout;
        bool iterating = run;
        Bilinear b2;
        f32x4 s_qi = {0.f, 0.f, 0.f, 0.f}, s_qgx = s_qi, s_qgy = s_qi;
        unsigned q_held = ~0u;                               // element offset of the footprint quads in s_q*: none of this level yet
        // A Newton step usually moves the window by less than a pixel: when its integer corner stays where it was, the footprint is
        // the one already in registers (only the bilinear weights change) and nothing is requested; neither is anything for a
        // feature that has stopped.  The launch is bound by the L1's requests to the L2 in flight (tools/pmc_mem.sh: ~78 per CU
        // all the time at ~750 clocks each), so a request not made is time saved: 89.2 -> 86.6 us per eight-pair launch of cfg-2,
        // 131.6 -> 125.7 us at cfg-4.  7x7 only: the 15x15 kernel, at its occupancy target of five, pays for the held offset and the
        // conditional loads with 16 more bytes of scratch and goes from 38.9 to 42.4 us at cfg-3.
        auto request_footprint = [&]() {
            const bool oob = (double)(x2 - (float)hw) < 0. || (float)nc - (x2 + (float)hw) < one_plus_eps ||
                             (double)(y2 - (float)hw) < 0. || (float)nr - (y2 + (float)hw) < one_plus_eps;
            if (iterating && oob) { status = KLT_OOB; iterating = false; }
            b2 = make_bilinear(x2, y2);
            if (REUSE) {
                const unsigned q = __umul24((unsigned)(b2.iy - hw + qr), (unsigned)nc) + (unsigned)(b2.ix - hw + 4 * qh);
                if (iterating && q != q_held) {
                    s_qi = load_quad(lv.i2, q); load_grad_quads(lv.gx2, q, s_qgx, s_qgy);
                    q_held = q;
                }
            } else {
                const unsigned q = iterating ? __umul24((unsigned)(b2.iy - hw + qr), (unsigned)nc) + (unsigned)(b2.ix - hw + 4 * qh) : 0u;
                s_qi = load_quad(lv.i2, q); load_grad_quads(lv.gx2, q, s_qgx, s_qgy);
            }
        };
        request_footprint();

        float t_i[4], t_gx[4], t_gy[4];
        sample_quad<QPR>(t_qi, b1, t_i);
        sample_quad<QPR>(t_qgx, b1, t_gx);
        sample_quad<QPR>(t_qgy, b1, t_gy);

        while (__any(iterating)) {
            const bool act = iterating;
            float s_i[4], s_gx[4], s_gy[4];
            sample_quad<QPR>(s_qi, b2, s_i);
            sample_quad<QPR>(s_qgx, b2, s_gx);
            sample_quad<QPR>(s_qgy, b2, s_gy);
            float tree[5] = {0.f, 0.f, 0.f, 0.f, 0.f};     // TREE: this lane's share of the five sums (its samples inside the window)
#pragma unroll
            for (int m = 0; m < 4; m++) {
                if (qr < w && 4 * qh + m < w) {
                    const int k = k0 + m;
                    const float diff = t_i[m] - s_i[m];
                    const float sx = t_gx[m] + s_gx[m];
                    const float sy = t_gy[m] + s_gy[m];
                    if (TREE) {
                        tree[0] = tree[0] + sx * sx;
                        tree[1] = tree[1] + sx * sy;
                        tree[2] = tree[2] + sy * sy;
                        tree[3] = tree[3] + diff * sx;
                        tree[4] = tree[4] + diff * sy;
                    } else {
                        gl[k] = sx * sx;
                        gl[npad + k] = sx * sy;
                        gl[2 * npad + k] = sy * sy;
                        gl[3 * npad + k] = diff * sx;
                        gl[4 * npad + k] = diff * sy;
                    }
                }
            }
            if (!TREE) wave_lds_sync();
            float acc = 0.f;
            if (!TREE && s < 5) {
                const float4 *T4 = reinterpret_cast<const float4 *>(gl + s * npad);
                if (W > 8) {
                    // 15x15: whole quads without a test, the n % 4 tail on its own -- with the tests inside the partly unrolled loop every
                    // quad paid three scalar compares and branches (13 M scalar next to 20 M vector instructions per launch): 49.0 -> 38.5 us
#pragma unroll 8
                    for (int q = 0; q < n / 4; q++) {
                        const float4 v = T4[q];
                        acc = acc + v.x;
                        acc = acc + v.y;
                        acc = acc + v.z;
                        acc = acc + v.w;
                    }
                    if (n % 4) {
                        const float4 v = T4[n / 4];
                        acc = acc + v.x;
                        if (n % 4 > 1) acc = acc + v.y;
                        if (n % 4 > 2) acc = acc + v.z;
                    }
                } else {
                    // 7x7: the 13 quads are unrolled completely and the tests fold (the peeled form measured 0.4 us slower here)
#pragma unroll 16
                    for (int q = 0; q < (n + 3) / 4; q++) {
                        const float4 v = T4[q];
                        acc = acc + v.x;
                        if (4 * q + 1 < n) acc = acc + v.y;
                        if (4 * q + 2 < n) acc = acc + v.z;
                        if (4 * q + 3 < n) acc = acc + v.w;
                    }
                }
            }
            if (!TREE) wave_lds_sync();
            float gxx, gxy, gyy, ex, ey;
            if (TREE) {
                gxx = group_tree_sum<LPF>(tree[0]); gxy = group_tree_sum<LPF>(tree[1]); gyy = group_tree_sum<LPF>(tree[2]);
                ex = group_tree_sum<LPF>(tree[3]) * a.step; ey = group_tree_sum<LPF>(tree[4]) * a.step;
            } else {
                gxx = __shfl(acc, glead); gxy = __shfl(acc, glead + 1); gyy = __shfl(acc, glead + 2);
                ex = __shfl(acc, glead + 3) * a.step; ey = __shfl(acc, glead + 4) * a.step;
            }
            const float p1 = gxx * gyy, p2 = gxy * gxy;
            const float det = p1 - p2;
            const bool small_det = det < a.small;
            if (act && small_det) { status = KLT_SMALL_DET; iterating = false; }
            const float n1 = gyy * ex, n2 = gxy * ey, n3 = gxx * ey, n4 = gxy * ex;
            const float dx = (n1 - n2) / det;
            const float dy = (n3 - n4) / det;
            if (act && !small_det) {
                status = KLT_TRACKED;
                x2 = x2 + dx;
                y2 = y2 + dy;
                it++;
                iterating = (fabsf(dx) >= a.th || fabsf(dy) >= a.th) && it < a.max_iterations;
            }
            if (__any(iterating)) request_footprint();
        }
        if (run) { xout = x2; yout = y2; }

        // trackFeatures.py:110 -- Python floats: half-window 3.5, eps 1.001 as doubles
        const double x2d = (double)x2, y2d = (double)y2, hwd = a.half_window;
        if (run && (x2d - hwd < 0.0 || (double)nc - (x2d + hwd) < 1.001 || y2d - hwd < 0.0 || (double)nr - (y2d + hwd) < 1.001))
            status = KLT_OOB;

        // residue, trackFeatures.py:118-125
        const bool need_res = run && status == KLT_TRACKED && a.use_max_residue;
        if (__any(need_res)) {
            const Bilinear br = make_bilinear(x2, y2);
            f32x4 r_qi;
            if (REUSE) {
                const unsigned q = __umul24((unsigned)(br.iy - hw + qr), (unsigned)nc) + (unsigned)(br.ix - hw + 4 * qh);
                r_qi = s_qi;                                 // the last footprint, if the final position has the same integer corner
                if (need_res && q != q_held) r_qi = load_quad(lv.i2, q);
            } else {
                const unsigned q = need_res ? __umul24((unsigned)(br.iy - hw + qr), (unsigned)nc) + (unsigned)(br.ix - hw + 4 * qh) : 0u;
                r_qi = load_quad(lv.i2, q);
            }
            float s_i[4];
            sample_quad<QPR>(r_qi, br, s_i);
            float sres;
            if (TREE) {
                float part = 0.f;
#pragma unroll
                for (int m = 0; m < 4; m++)
                    if (qr < w && 4 * qh + m < w) part = part + fabsf(t_i[m] - s_i[m]);
                sres = group_tree_sum<LPF>(part);
            } else {
#pragma unroll
                for (int m = 0; m < 4; m++)
                    if (qr < w && 4 * qh + m < w) gl[k0 + m] = fabsf(t_i[m] - s_i[m]);
                wave_lds_sync();
                sres = pairwise_group<3>(gl, n, s);
                wave_lds_sync();
                sres = __shfl(sres, glead);
            }
            if (need_res && sres / (float)n > a.max_residue) status = KLT_LARGE_RESIDUE;
        }

        int lvl_val;
        if (!t_ok) lvl_val = KLT_OOB;
        else if (a.retain) lvl_val = KLT_TRACKED;                                               // :127-129
        else if (status == KLT_SMALL_DET || status == KLT_OOB || status == KLT_LARGE_RESIDUE) lvl_val = status;
        else if (it >= a.max_iterations) lvl_val = KLT_MAX_ITERATIONS;
        else lvl_val = KLT_TRACKED;
        if (alive) {
            val = lvl_val;
            aux |= (uint32_t)(it < 14 ? it + 1 : 15) << (4 * r);
            alive = !(val == KLT_SMALL_DET || val == KLT_OOB);                                  // :284-285
        }
    }
    if (tracked_feature && s == 0) {
        klt_feat o;
        o.aux = (int32_t)aux;
        const double xd = (double)xout, yd = (double)yout;
        const bool oob = val == KLT_OOB ||
                         xd < a.borderx || xd > (double)(a.ncols - 1) - a.borderx ||
                         yd < a.bordery || yd > (double)(a.nrows - 1) - a.bordery;   // :288-308
        if (oob) { o.x = -1.f; o.y = -1.f; o.val = KLT_OOB; }
        else if (val == KLT_SMALL_DET || val == KLT_LARGE_RESIDUE || val == KLT_MAX_ITERATIONS) {
            o.x = -1.f; o.y = -1.f; o.val = val;
        } else { o.x = xout; o.y = yout; o.val = KLT_TRACKED; }
        fout[f] = o;
    }
}

// (15x15, measured and not kept -- round 2: several features per workgroup with ONE wavefront adding the five 225-term chains of
// all of them (5 NW lanes busy instead of 5; the chains are 40 % of an iteration's instructions): 68.4 us (8 features) / 62.7 (4)
// against 59.8 for one independent wavefront per feature at cfg-3.  The barriers and the lock step cost more than the saved issue
// slots.)

// Features sorted by image row (counting sort, one workgroup): order[0..n) = feature indices by ascending (int)y; lost
// features go last.  The order within a row is whatever the atomics give -- every feature is still tracked exactly once and
// written to its own slot, so the result does not depend on it.
constexpr int ORDER_BINS = 4096, ORDER_T = 1024;
__global__ __launch_bounds__(ORDER_T) void track_order_kernel(const klt_feat *__restrict__ in_single, const TrackPairDesc *__restrict__ pairs,
                                                              int n, uint32_t *__restrict__ order_base)
{
    const klt_feat *__restrict__ in = pairs ? pairs[blockIdx.x].in : in_single;      // batched launches: one workgroup per pair
    uint32_t *__restrict__ order = order_base + (size_t)blockIdx.x * n;
    __shared__ unsigned hist[ORDER_BINS], start[ORDER_BINS];
    __shared__ unsigned wsum[ORDER_T / 64];
    const int tid = threadIdx.x;
    for (int i = tid; i < ORDER_BINS; i += ORDER_T) hist[i] = 0u;
    __syncthreads();
    auto bin_of = [&](const klt_feat &ft) {
        if (ft.val < 0) return ORDER_BINS - 1;
        const int y = (int)ft.y;
        return y < 0 ? 0 : y > ORDER_BINS - 2 ? ORDER_BINS - 2 : y;
    };
    for (int i = tid; i < n; i += ORDER_T) atomicAdd(&hist[bin_of(in[i])], 1u);
    __syncthreads();
    // exclusive scan of the 4096 bins: 4 bins per thread, wave scan, then the 16 wave totals
    unsigned v[4], tsum = 0;
#pragma unroll
    for (int k = 0; k < 4; k++) { v[k] = hist[4 * tid + k]; tsum += v[k]; }
    unsigned inc = tsum;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const unsigned o = __shfl_up(inc, d);
        if ((tid & 63) >= d) inc += o;
    }
    if ((tid & 63) == 63) wsum[tid >> 6] = inc;
    __syncthreads();
    unsigned base = 0;
    for (int wv = 0; wv < (tid >> 6); wv++) base += wsum[wv];
    unsigned run = base + inc - tsum;
#pragma unroll
    for (int k = 0; k < 4; k++) { start[4 * tid + k] = run; run += v[k]; }
    __syncthreads();
    for (int i = tid; i < n; i += ORDER_T) order[atomicAdd(&start[bin_of(in[i])], 1u)] = (uint32_t)i;
}

// Iteration statistics from the per-feature aux words (only launched while statistics are being collected;
// per-feature atomics on a handful of shared counters would serialise the whole tracker).
__global__ __launch_bounds__(256) void track_stats_kernel(const klt_feat *__restrict__ in, const klt_feat *__restrict__ out,
                                                           int n, int nlevels, unsigned long long *stats)
{
    __shared__ unsigned int acc[1 + 2 * KLT_MAX_LEVELS];
    if (threadIdx.x < 1 + 2 * KLT_MAX_LEVELS) acc[threadIdx.x] = 0u;
    __syncthreads();
    const int f = blockIdx.x * 256 + threadIdx.x;
    if (f < n && in[f].val >= 0) {
        atomicAdd(&acc[0], 1u);
        const uint32_t aux = (uint32_t)out[f].aux;
        for (int r = 0; r < nlevels; r++) {
            const uint32_t v = (aux >> (4 * r)) & 15u;
            if (v) {
                atomicAdd(&acc[1 + r], 1u);
                atomicAdd(&acc[1 + KLT_MAX_LEVELS + r], v - 1u);
            }
        }
    }
    __syncthreads();
    if (threadIdx.x < 1 + 2 * KLT_MAX_LEVELS && acc[threadIdx.x])
        atomicAdd(&stats[threadIdx.x], (unsigned long long)acc[threadIdx.x]);
}

}  // namespace

void launch_track_stats(hipStream_t s, const klt_feat *in, const klt_feat *out, int n, int nlevels, unsigned long long *stats)
{
    if (n <= 0) return;
    hipLaunchKernelGGL(track_stats_kernel, dim3((n + 255) / 256), dim3(256), 0, s, in, out, n, nlevels, stats);
}

// Tracker kernels: track_kernel_quad (7x7 windows with lists of 2048 features and more: four features per wavefront; 15x15
// windows: one feature per wavefront; one 16-byte load per lane, image and footprint), track_kernel (one feature per wavefront,
// one sample per lane and round: every other window, short 7x7 lists).  All give bit-identical records.
// KLT_OPT_TRACK_VARIANT = 0 (or KLT_TRACK_VARIANT=0 in the environment) forces track_kernel: the plain fallback the parity
// tests compare the quad kernel with.  The measured-slower generations in between (prefetching, per-sample loads with four
// features per wavefront, one pixel per lane) are recorded in profiles/README.md and live in the git history.
int g_track_variant = getenv("KLT_TRACK_VARIANT") ? atoi(getenv("KLT_TRACK_VARIANT")) : 4;

template <bool BATCH>
static int launch_track_t(hipStream_t s, const TrackArgs &a)
{
    const int n = a.window * a.window;
    if (n > 1024) return -1;
    const size_t lds = 5 * (size_t)((n + 3) & ~3) * sizeof(float);
    const unsigned ny = BATCH ? a.npairs : 1;
    const dim3 block(64);
    // the permutation is (re)computed here whenever the caller asks for it, whichever kernel consumes it
    if (a.order && a.order_refresh)
        hipLaunchKernelGGL(track_order_kernel, dim3(ny), dim3(ORDER_T), 0, s, a.in, BATCH ? a.pairs : nullptr, a.n, a.order);
    // (short 7x7 lists keep one feature per wavefront: with a few hundred features the launch is pure latency, which four
    // features in lock step lengthen)
    if (g_track_variant != 0 && a.window == 7 && (long long)a.n * ny >= 2048) {
        const dim3 gq(a.order ? 8 * ((a.order_chunk + 3) / 4) : (a.n + 3) / 4, ny);
        // (occupancy targets 5 / 6 for this kernel -- 96 / 80 VGPRs with 84 / 172 bytes of scratch per lane instead of 122 VGPRs -- were
        // measured again in round 3 on eight-pair launches, 10 000 wavefronts, where a fifth resident wavefront per SIMD could hide
        // latency: 112 / 199 us per launch against 87.5, cfg-4's 32-pair shard 0.535 / 0.687 ms against 0.495.  A scratch reload is a
        // vector memory operation and those return in order: it waits behind the footprint loads in flight, i.e. for the round trip
        // the extra wavefront was meant to hide.)
        if (a.tree_sums) klt_launch((track_kernel_quad<BATCH, 7, 1, true>), gq, block, 0u, s, a);
        else klt_launch((track_kernel_quad<BATCH, 7>), gq, block, (unsigned)(4 * lds), s, a);
        return 0;
    }
    if (g_track_variant != 0 && a.window == 15) {
        const dim3 gq(a.order ? 8 * a.order_chunk : a.n, ny);
        // occupancy target 5 (96 VGPRs, 40 bytes of scratch per lane instead of 107 VGPRs): the 5000 wavefronts of cfg-3 are then
        // resident at once instead of in two rounds -- 57.9 -> 49.3 us.  (The 7x7 kernel loses from the same cap, see above.)
        if (a.tree_sums) klt_launch((track_kernel_quad<BATCH, 15, 5, true>), gq, block, 0u, s, a);
        else klt_launch((track_kernel_quad<BATCH, 15, 5>), gq, block, (unsigned)lds, s, a);
        return 0;
    }
    const dim3 grid(a.order ? 8 * a.order_chunk : a.n, ny);
    if (a.window == 7) klt_launch((track_kernel<1, 7, BATCH>), grid, block, (unsigned)lds, s, a);
    else if (a.window == 15) klt_launch((track_kernel<4, 15, BATCH>), grid, block, (unsigned)lds, s, a);
    else if (n <= 64) klt_launch((track_kernel<1, 0, BATCH>), grid, block, (unsigned)lds, s, a);
    else if (n <= 128) klt_launch((track_kernel<2, 0, BATCH>), grid, block, (unsigned)lds, s, a);
    else if (n <= 256) klt_launch((track_kernel<4, 0, BATCH>), grid, block, (unsigned)lds, s, a);
    else if (n <= 512) klt_launch((track_kernel<8, 0, BATCH>), grid, block, (unsigned)lds, s, a);
    else klt_launch((track_kernel<16, 0, BATCH>), grid, block, (unsigned)lds, s, a);
    return 0;
}

// ------------------------------------------------------------------------------------------------------
// The reference's literal native boundary (setup.py:8-9), one call = one wavefront -- what the compat modules trackFeaturesUtils
// bind (pyfeaturetrack_amd/compat).  Separate (not interleaved) planes: the caller's own arrays.
namespace {

// extractImagePatchSlow, trackFeaturesUtils.pyx:14-51: w x w bilinear samples around (x, y); out[0] = 1 when the footprint leaves
// the image (the reference asserts, :35)
__global__ __launch_bounds__(64) void extract_patch_kernel(const float *__restrict__ img, int nc, int nr, float x, float y, int w,
                                                           float *__restrict__ patch, int *__restrict__ bad)
{
    const int hw = w / 2, n = w * w;
    const Bilinear b = make_bilinear(x, y);
    if (!(b.ix - hw >= 0 && b.iy - hw >= 0 && b.ix + hw + 2 <= nc && b.iy + hw + 2 <= nr)) {
        if (threadIdx.x == 0) *bad = 1;
        return;
    }
    if (threadIdx.x == 0) *bad = 0;
    for (int k = threadIdx.x; k < n; k += 64) {
        const size_t q = (size_t)(b.iy - hw + k / w) * nc + (b.ix - hw + k % w);
        patch[k] = sample(img + q, nc, b);
    }
}

// trackFeatureIterateCKLT, trackFeaturesUtils.pyx:393-459, on template patches the caller extracted: the Newton loop of track_level
// (same operations in the same order) without the template sampling, the post-loop tests and the status priority, which belong to
// _trackFeature (trackFeatures.py:67-136).  res = {x2, y2, status, iterations}
__global__ __launch_bounds__(64) void track_iterate_kernel(const float *__restrict__ t_gx, const float *__restrict__ t_gy,
                                                           const float *__restrict__ t_i, const float *__restrict__ i2,
                                                           const float *__restrict__ gx2, const float *__restrict__ gy2, int nc, int nr,
                                                           int w, float x2, float y2, float step, float small, float th,
                                                           int max_iterations, float *__restrict__ res)
{
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int lane = threadIdx.x, n = w * w, hw = w / 2, npad = (n + 3) & ~3;
    int iters = 0, status;
    const float one_plus_eps = 1.001f;
    for (;;) {
        if ((double)(x2 - (float)hw) < 0. || (float)nc - (x2 + (float)hw) < one_plus_eps ||
            (double)(y2 - (float)hw) < 0. || (float)nr - (y2 + (float)hw) < one_plus_eps) {       // :428-431
            status = KLT_OOB;
            break;
        }
        const Bilinear b2 = make_bilinear(x2, y2);
        const size_t base = (size_t)(b2.iy - hw) * nc + (b2.ix - hw);
        for (int k = lane; k < n; k += 64) {
            const size_t q = base + (size_t)(k / w) * nc + (k % w);
            const float diff = t_i[k] - sample(i2 + q, nc, b2);
            const float sx = t_gx[k] + sample(gx2 + q, nc, b2);
            const float sy = t_gy[k] + sample(gy2 + q, nc, b2);
            lds[k] = sx * sx;
            lds[npad + k] = sx * sy;
            lds[2 * npad + k] = sy * sy;
            lds[3 * npad + k] = diff * sx;
            lds[4 * npad + k] = diff * sy;
        }
        __syncthreads();
        float acc = 0.f;
        if (lane < 5) {
            const float *T = lds + lane * npad;
            for (int k = 0; k < n; k++) acc = acc + T[k];
        }
        __syncthreads();
        const float gxx = __shfl(acc, 0), gxy = __shfl(acc, 1), gyy = __shfl(acc, 2);
        const float ex = __shfl(acc, 3) * step, ey = __shfl(acc, 4) * step;
        const float p1 = gxx * gyy, p2 = gxy * gxy;
        const float det = p1 - p2;
        if (det < small) { status = KLT_SMALL_DET; break; }
        const float n1 = gyy * ex, n2 = gxy * ey, n3 = gxx * ey, n4 = gxy * ex;
        const float dx = (n1 - n2) / det;
        const float dy = (n3 - n4) / det;
        status = KLT_TRACKED;
        x2 = x2 + dx;
        y2 = y2 + dy;
        iters++;
        if (!((fabsf(dx) >= th || fabsf(dy) >= th) && iters < max_iterations)) break;
    }
    if (lane == 0) { res[0] = x2; res[1] = y2; res[2] = (float)status; res[3] = (float)iters; }
}

}  // namespace

void launch_extract_patch(hipStream_t s, const float *img, int nc, int nr, float x, float y, int w, float *patch, int *bad)
{
    hipLaunchKernelGGL(extract_patch_kernel, dim3(1), dim3(64), 0, s, img, nc, nr, x, y, w, patch, bad);
}

void launch_track_iterate(hipStream_t s, const float *t_gx, const float *t_gy, const float *t_i, const float *i2, const float *gx2,
                          const float *gy2, int nc, int nr, int w, float x2, float y2, float step, float small, float th,
                          int max_iterations, float *res)
{
    const size_t lds = 5 * (size_t)((w * w + 3) & ~3) * sizeof(float);
    hipLaunchKernelGGL(track_iterate_kernel, dim3(1), dim3(64), (unsigned)lds, s, t_gx, t_gy, t_i, i2, gx2, gy2, nc, nr, w, x2, y2, step,
                       small, th, max_iterations, res);
}

int launch_track(hipStream_t s, const TrackArgs &a)
{
    if (a.n <= 0) return 0;
    if (a.pairs) return a.npairs > 0 ? launch_track_t<true>(s, a) : 0;
    return launch_track_t<false>(s, a);
}
