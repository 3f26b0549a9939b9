// Internal declarations shared by the HIP translation units of libkltgpu.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "klt_gpu.h"

// FP64 taps in the order scipy.ndimage.correlate1d consumes them (i.e. the convolve.py taps reversed,
// convolve1d does `weights[::-1]`), plus the symmetry class its inner loop switches on.
struct Taps {
    double k[KLT_MAX_KERNEL_WIDTH];
    int n;
    int sym;   // +1 symmetric, -1 antisymmetric, 0 neither
};

struct TrackLevel {
    const float *i1, *gx1, *gy1, *i2, *gx2, *gy2;
    int nc, nr;
};

struct TrackArgs {
    TrackLevel lv[KLT_MAX_LEVELS];
    const klt_feat *in;
    klt_feat *out;
    unsigned long long *stats;   // [0] features, [1..8] level visits, [9..16] iterations
    double half_window;          // window/2 as the Python float (3.5 for 7x7), trackFeatures.py:88-89
    double borderx, bordery;
    int n, nlevels, window, max_iterations, use_max_residue, retain, ncols, nrows;
    float small, th, step, max_residue, ss;
};

struct SelectArgs {
    const float *sat;        // 3 planes (gxx, gxy, gyy), each ncols*nrows
    float *valmap;           // [ny][nx]
    unsigned long long *keys;
    const uint8_t *seedmap;  // may be null
    double min_eig;
    int ncols, nrows, bx, by, step, nx, ny, hw, hh, npow2;
};

struct NmsArgs {
    const unsigned long long *keys;
    klt_feat *fl;
    uint32_t *grid_global;   // used when the cell grid does not fit in LDS
    int *placed_out;
    int nkeys, nfeat, overwrite_all, d /* mindist-1 */, cell, gw, gh, grid_in_lds;
};

// ---- launchers (each enqueues on `s`; no synchronisation) ----
void launch_hconv_u8(hipStream_t s, const uint8_t *in, int ncols, int nrows, float *outA, float *outB,
                     int out_cols, int xstride, int xoff, const Taps &ta, const Taps *tb);
void launch_hconv_f32(hipStream_t s, const float *in, int ncols, int nrows, float *outA, float *outB,
                      int out_cols, int xstride, int xoff, const Taps &ta, const Taps *tb);
void launch_vconv(hipStream_t s, const float *inA, const float *inB, int ncols, int nrows, float *outA, float *outB,
                  int out_rows, int ystride, int yoff, const Taps &ta, const Taps *tb);

void launch_sat_rows(hipStream_t s, const float *gx, const float *gy, float *sat, int ncols, int nrows);
void launch_sat_cols(hipStream_t s, float *sat, int ncols, int nrows);
void launch_seed_fill(hipStream_t s, const klt_feat *fl, int nfeat, uint8_t *seedmap, int ncols, int nrows, int d);
void launch_eigen(hipStream_t s, const SelectArgs &a);
void launch_sort_desc(hipStream_t s, unsigned long long *keys, int npow2);
int  launch_nms(hipStream_t s, const NmsArgs &a);   // returns 0 or a hipError_t
void launch_unpack_candidates(hipStream_t s, const unsigned long long *keys, int n, float *val, int *x, int *y);

int launch_track(hipStream_t s, const TrackArgs &a);   // returns 0, or -1 for an unsupported window
