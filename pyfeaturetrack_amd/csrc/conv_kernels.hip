// Separable FP64-accumulated convolutions for the pyramid build (gfx950).
//
// What is reproduced (reference convolve.py:208-214 -> scipy.ndimage.convolve1d):
//   * f32 (or u8) samples widened to FP64, FP64 taps, FP64 accumulate in correlate1d's exact operation
//     order (centre tap first, then symmetric / antisymmetric pairs from the outside in);
//   * `reflect` (half-sample symmetric) borders;
//   * the result of each 1-D pass rounded to f32 before the next pass reads it;
//   * no FMA contraction anywhere (scipy's C is compiled without FMA).
// Each output sample is an independent expression, so computing only the samples that survive the
// pyramid's subsampling (pyramid.py:70-72) is bit-identical to convolving everything and discarding.
//
// Generic path: one thread per output sample, taps in the kernarg segment (scalar loads), neighbours
// through the vector L1.  Works for every tap count up to 71 and every stride.
#include "klt_internal.h"

#pragma clang fp contract(off)

namespace {

__device__ __forceinline__ int reflect_idx(int i, int n)
{
    const int p = 2 * n;
    int m = i % p;
    if (m < 0) m += p;
    return m < n ? m : p - 1 - m;
}

// one correlate1d output at position `pos` of a line of length n (element stride `stride`)
template <typename TIn>
__device__ __forceinline__ float correlate_at(const TIn *line, size_t stride, int pos, int n, const Taps &t)
{
    const int size1 = t.n / 2;
    const int size2 = t.n - size1 - 1;
    const double *fw = t.k + size1;
    // scipy.ndimage.convolve1d hands correlate1d origin -1 for an EVEN tap count ("if not weights.shape[0] & 1: origin -= 1"): the window of
    // output `pos` then sits one sample to the right.  The reference's own taps are always odd (convolve.py:27-93); only
    // klt_convolve_separate_f32 (_convolveSeparate with a caller's tap lists) can bring an even count, which is never symmetric-classed.
    pos += (t.n & 1) ^ 1;
    double acc;
    if (pos - size1 >= 0 && pos + size2 < n) {          // interior: no index folding
        const TIn *c = line + (size_t)pos * stride;
        if (t.sym > 0) {
            acc = (double)c[0] * fw[0];
            for (int jj = -size1; jj < 0; jj++) {
                const double a = (double)c[(ptrdiff_t)jj * (ptrdiff_t)stride];
                const double b = (double)c[(ptrdiff_t)(-jj) * (ptrdiff_t)stride];
                acc = acc + (a + b) * fw[jj];
            }
        } else if (t.sym < 0) {
            acc = (double)c[0] * fw[0];
            for (int jj = -size1; jj < 0; jj++) {
                const double a = (double)c[(ptrdiff_t)jj * (ptrdiff_t)stride];
                const double b = (double)c[(ptrdiff_t)(-jj) * (ptrdiff_t)stride];
                acc = acc + (a - b) * fw[jj];
            }
        } else {
            acc = (double)c[(ptrdiff_t)size2 * (ptrdiff_t)stride] * fw[size2];
            for (int jj = -size1; jj < size2; jj++)
                acc = acc + (double)c[(ptrdiff_t)jj * (ptrdiff_t)stride] * fw[jj];
        }
    } else {
        if (t.sym > 0) {
            acc = (double)line[(size_t)reflect_idx(pos, n) * stride] * fw[0];
            for (int jj = -size1; jj < 0; jj++) {
                const double a = (double)line[(size_t)reflect_idx(pos + jj, n) * stride];
                const double b = (double)line[(size_t)reflect_idx(pos - jj, n) * stride];
                acc = acc + (a + b) * fw[jj];
            }
        } else if (t.sym < 0) {
            acc = (double)line[(size_t)reflect_idx(pos, n) * stride] * fw[0];
            for (int jj = -size1; jj < 0; jj++) {
                const double a = (double)line[(size_t)reflect_idx(pos + jj, n) * stride];
                const double b = (double)line[(size_t)reflect_idx(pos - jj, n) * stride];
                acc = acc + (a - b) * fw[jj];
            }
        } else {
            acc = (double)line[(size_t)reflect_idx(pos + size2, n) * stride] * fw[size2];
            for (int jj = -size1; jj < size2; jj++)
                acc = acc + (double)line[(size_t)reflect_idx(pos + jj, n) * stride] * fw[jj];
        }
    }
    return (float)acc;
}

// horizontal pass: out[y][xs] = correlate(row y, at x = xs*xstride + xoff); NOUT==2 applies a second tap set
template <typename TIn, int NOUT>
__global__ __launch_bounds__(256) void hconv_kernel(const TIn *__restrict__ in, int ncols, int nrows,
                                                     float *__restrict__ outA, float *__restrict__ outB,
                                                     int out_cols, int xstride, int xoff, Taps ta, Taps tb)
{
    const int xs = blockIdx.x * 64 + threadIdx.x;
    const int y = blockIdx.y * 4 + threadIdx.y;
    if (xs >= out_cols || y >= nrows) return;
    const int x = xs * xstride + xoff;
    const TIn *row = in + (size_t)y * ncols;
    outA[(size_t)y * out_cols + xs] = correlate_at(row, 1, x, ncols, ta);
    if (NOUT == 2) outB[(size_t)y * out_cols + xs] = correlate_at(row, 1, x, ncols, tb);
}

// vertical pass: out[ys][x] = correlate(column x, at y = ys*ystride + yoff)
template <int NOUT>
__global__ __launch_bounds__(256) void vconv_kernel(const float *__restrict__ inA, const float *__restrict__ inB,
                                                     int ncols, int nrows, float *__restrict__ outA,
                                                     float *__restrict__ outB, int out_rows, int ystride, int yoff,
                                                     Taps ta, Taps tb, int ostride)
{
    const int x = blockIdx.x * 64 + threadIdx.x;
    const int ys = blockIdx.y * 4 + threadIdx.y;
    if (x >= ncols || ys >= out_rows) return;
    const int y = ys * ystride + yoff;
    const size_t o = ((size_t)ys * ncols + x) * ostride;          // ostride 2: outB == outA + 1, interleaved gradient planes
    outA[o] = correlate_at(inA + x, (size_t)ncols, y, nrows, ta);
    if (NOUT == 2) outB[o] = correlate_at(inB + x, (size_t)ncols, y, nrows, tb);
}

}  // namespace

template <typename TIn>
static void launch_hconv_t(hipStream_t s, const TIn *in, int ncols, int nrows, float *outA, float *outB,
                           int out_cols, int xstride, int xoff, const Taps &ta, const Taps *tb)
{
    dim3 block(64, 4), grid((out_cols + 63) / 64, (nrows + 3) / 4);
    if (tb)
        hipLaunchKernelGGL((hconv_kernel<TIn, 2>), grid, block, 0, s, in, ncols, nrows, outA, outB, out_cols, xstride, xoff, ta, *tb);
    else
        hipLaunchKernelGGL((hconv_kernel<TIn, 1>), grid, block, 0, s, in, ncols, nrows, outA, outB, out_cols, xstride, xoff, ta, ta);
}

void launch_hconv_u8(hipStream_t s, const uint8_t *in, int ncols, int nrows, float *outA, float *outB,
                     int out_cols, int xstride, int xoff, const Taps &ta, const Taps *tb)
{
    launch_hconv_t<uint8_t>(s, in, ncols, nrows, outA, outB, out_cols, xstride, xoff, ta, tb);
}

void launch_hconv_f32(hipStream_t s, const float *in, int ncols, int nrows, float *outA, float *outB,
                      int out_cols, int xstride, int xoff, const Taps &ta, const Taps *tb)
{
    launch_hconv_t<float>(s, in, ncols, nrows, outA, outB, out_cols, xstride, xoff, ta, tb);
}

// dst[i] = src[i * stride]: one of the two interleaved gradient planes as a plane of its own (the ABI's plane downloads)
__global__ __launch_bounds__(256) void take_strided_kernel(const float *__restrict__ src, float *__restrict__ dst, size_t n, int stride)
{
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) dst[i] = src[i * stride];
}

void launch_take_strided(hipStream_t s, const float *src, float *dst, size_t n, int stride)
{
    const size_t blocks = (n + 255) / 256;
    hipLaunchKernelGGL(take_strided_kernel, dim3((unsigned)(blocks < 4096 ? (blocks ? blocks : 1) : 4096)), dim3(256), 0, s, src, dst, n, stride);
}

void launch_vconv(hipStream_t s, const float *inA, const float *inB, int ncols, int nrows, float *outA, float *outB,
                  int out_rows, int ystride, int yoff, const Taps &ta, const Taps *tb, int ostride)
{
    dim3 block(64, 4), grid((ncols + 63) / 64, (out_rows + 3) / 4);
    if (tb)
        hipLaunchKernelGGL((vconv_kernel<2>), grid, block, 0, s, inA, inB, ncols, nrows, outA, outB, out_rows, ystride, yoff, ta, *tb, ostride);
    else
        hipLaunchKernelGGL((vconv_kernel<1>), grid, block, 0, s, inA, inB, ncols, nrows, outA, outB, out_rows, ystride, yoff, ta, ta, ostride);
}
