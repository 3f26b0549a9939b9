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
#include "klt_internal.h"

#pragma clang fp contract(off)

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

__device__ __forceinline__ float sample(const float *__restrict__ q, int nc, const Bilinear &b)
{
    const float t4 = b.w11 * q[nc + 1];
    double v = b.w00 * (double)q[0];
    v = v + b.w01 * (double)q[1];
    v = v + b.w10 * (double)q[nc];
    v = v + (double)t4;
    return (float)v;
}

// numpy's pairwise summation for one block of n <= 128 floats (and the n < 8 loop)
__device__ float pairwise_block(const float *a, int n)
{
    if (n < 8) {
        float res = 0.f;
        for (int i = 0; i < n; i++) res = res + a[i];
        return res;
    }
    float r[8];
    for (int j = 0; j < 8; j++) r[j] = a[j];
    int i;
    for (i = 8; i < n - (n % 8); i += 8)
        for (int j = 0; j < 8; j++) r[j] = r[j] + a[i + j];
    float res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
    for (; i < n; i++) res = res + a[i];
    return res;
}

template <int DEPTH>
__device__ float pairwise_sum(const float *a, int n)
{
    if (n <= 128) return pairwise_block(a, n);
    int n2 = n / 2;
    n2 -= n2 % 8;
    return pairwise_sum<DEPTH - 1>(a, n2) + pairwise_sum<DEPTH - 1>(a + n2, n - n2);
}
template <>
__device__ float pairwise_sum<0>(const float *a, int n)
{
    return pairwise_block(a, n < 128 ? n : 128);
}

// _trackFeature for one level.  Returns the status; x2/y2 updated in place; `iters` = Newton iterations.
// WCT > 0: window size known at compile time (index math folds, the summation loops unroll and read LDS
// 16 bytes at a time); WCT == 0: any odd window up to 31.
template <int MAXK, int WCT>
__device__ int track_level(const TrackArgs &a, const TrackLevel &lv, float x1, float y1, float &x2r, float &y2r,
                           float *lds, int lane, int &iters)
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
            t_gx[kk] = sample(lv.gx1 + q, nc, b1);
            t_gy[kk] = sample(lv.gy1 + q, nc, b1);
        }
    }

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
                const float sx = t_gx[kk] + sample(lv.gx2 + q, nc, b2);          // -( -g1 - g2 ), :128 and :297
                const float sy = t_gy[kk] + sample(lv.gy2 + q, nc, b2);
                lds[k] = sx * sx;                  // gxx terms, :299
                lds[npad + k] = sx * sy;           // gxy terms, :300
                lds[2 * npad + k] = sy * sy;       // gyy terms, :301
                lds[3 * npad + k] = diff * sx;     // ex terms,  :265
                lds[4 * npad + k] = diff * sy;     // ey terms,  :266
            }
        }
        __syncthreads();
        // lanes 0..4 each add one array in the reference's row-major order (sequential f32 adds)
        float acc = 0.f;
        if (lane < 5) {
            const float *T = lds + lane * npad;
            if (WCT > 0) {
                const float4 *T4 = reinterpret_cast<const float4 *>(T);
#pragma unroll(WCT <= 8 ? 16 : 4)
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
        iters++;
        if (!((fabsf(dx) >= a.th || fabsf(dy) >= a.th) && iters < a.max_iterations)) break;
    }
    x2r = x2;
    y2r = y2;

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
        float s = 0.f;
        if (lane == 0) s = pairwise_sum<3>(l_diff, n);
        __syncthreads();
        s = __shfl(s, 0);
        if (s / (float)n > a.max_residue) status = KLT_LARGE_RESIDUE;
    }

    if (a.retain) return KLT_TRACKED;                                   // :127-129
    if (status == KLT_SMALL_DET || status == KLT_OOB || status == KLT_LARGE_RESIDUE) return status;
    if (iters >= a.max_iterations) return KLT_MAX_ITERATIONS;
    return KLT_TRACKED;
}

template <int MAXK, int WCT, bool BATCH>
__global__ __launch_bounds__(64) void track_kernel(TrackArgs a)
{
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int f = blockIdx.x;
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
    // trackFeatures.py:255-265: position at the coarsest resolution (divisions by a power of two: exact)
    float xloc = ft.x, yloc = ft.y;
    for (int r = 0; r < L; r++) { xloc = xloc * a.inv_ss; yloc = yloc * a.inv_ss; }   // power of two: exact
    float xout = xloc, yout = yloc;
    int val = KLT_TRACKED;
    uint32_t aux = 0;       // 4 bits per level: 0 = level not visited, v = v-1 Newton iterations (saturating at 14)
    for (int r = L - 1; r >= 0; r--) {
        xloc = xloc * a.ss; yloc = yloc * a.ss; xout = xout * a.ss; yout = yout * a.ss;
        int it = 0;
        val = track_level<MAXK, WCT>(a, levels[r], xloc, yloc, xout, yout, lds, lane, it);
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

template <bool BATCH>
static int launch_track_t(hipStream_t s, const TrackArgs &a)
{
    const int n = a.window * a.window;
    const size_t lds = 5 * (size_t)((n + 3) & ~3) * sizeof(float);
    const dim3 grid(a.n, BATCH ? a.npairs : 1), block(64);
    if (a.window == 7) hipLaunchKernelGGL((track_kernel<1, 7, BATCH>), grid, block, lds, s, a);
    else if (a.window == 15) hipLaunchKernelGGL((track_kernel<4, 15, BATCH>), grid, block, lds, s, a);
    else if (n <= 64) hipLaunchKernelGGL((track_kernel<1, 0, BATCH>), grid, block, lds, s, a);
    else if (n <= 128) hipLaunchKernelGGL((track_kernel<2, 0, BATCH>), grid, block, lds, s, a);
    else if (n <= 256) hipLaunchKernelGGL((track_kernel<4, 0, BATCH>), grid, block, lds, s, a);
    else if (n <= 512) hipLaunchKernelGGL((track_kernel<8, 0, BATCH>), grid, block, lds, s, a);
    else if (n <= 1024) hipLaunchKernelGGL((track_kernel<16, 0, BATCH>), grid, block, lds, s, a);
    else return -1;
    return 0;
}

int launch_track(hipStream_t s, const TrackArgs &a)
{
    if (a.n <= 0) return 0;
    if (a.pairs) return a.npairs > 0 ? launch_track_t<true>(s, a) : 0;
    return launch_track_t<false>(s, a);
}
