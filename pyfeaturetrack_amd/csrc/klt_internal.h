// Internal declarations shared by the HIP translation units of libkltgpu.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <string>

#include "klt_gpu.h"

// FP64 taps in the order scipy.ndimage.correlate1d consumes them (i.e. the convolve.py taps reversed,
// convolve1d does `weights[::-1]`), plus the symmetry class its inner loop switches on.
struct Taps {
    double k[KLT_MAX_KERNEL_WIDTH];
    int n;
    int sym;   // +1 symmetric, -1 antisymmetric, 0 neither
};

constexpr int KLT_MAX_BATCH = 32;   // frames per fused pyramid launch (pointer tables travel in the kernarg segment)

struct SmoothGradArgs {
    const void *raw[KLT_MAX_BATCH];   // u8 or f32 frame (or, for gradients only, the level image)
    float *img[KLT_MAX_BATCH];        // smoothed image out (unused for gradients only)
    float *gx[KLT_MAX_BATCH], *gy[KLT_MAX_BATCH];   // gradient planes; gstride == 2: ONE interleaved plane per entry, gy[b] == gx[b] + 1
    int gstride;                      // element stride of a gradient plane: 2 for the planes of slots / the selection (KLT_GRAD_STRIDE), 1 for separate planes
    Taps smooth, ggauss, gderiv;
    int ncols, nrows, R;              // R = max gradient tap radius; ncols/nrows = largest entry (grid extent)
    short dim_c[KLT_MAX_BATCH], dim_r[KLT_MAX_BATCH];   // per-entry geometry when entries differ (0 = use ncols/nrows)
    // fused horizontal pass of the first pyramid reduction (smooth_grad_rb<..., HRED>): taps, H1 planes [nrows][h1_nc]
    Taps reduce;
    float *h1[KLT_MAX_BATCH];
    int h1_nc;
};

struct PyrReduceArgs {
    const float *src[KLT_MAX_BATCH];
    float *dst[KLT_MAX_BATCH];
    Taps taps;
    int src_nc, src_nr, dst_nc, dst_nr, ss, log2ss;
};

// The two gradient planes of a pyramid level are stored INTERLEAVED: pixel (y, x) holds gradx at element 2 (y nc + x) and grady right
// behind it, so gy == gx + 1 and a row of a tracking window's footprint is one contiguous piece (64 bytes for an 8-pixel row) instead of
// two 32-byte pieces in two planes -- the tracker is bound by the cache lines its footprints touch.  Producers
// (level-0 kernel, gradient kernels) write both values of a pixel together, consumers (tracker, affine check, summed-area row pass)
// read them together; only the plane download of the ABI separates them again.
constexpr int KLT_GRAD_STRIDE = 2;

struct TrackLevel {
    const float *i1, *gx1, *gy1, *i2, *gx2, *gy2;   // gy == gx + 1 (interleaved gradient planes)
    int nc, nr;
};

// one frame pair of a batched tracker launch (table in device memory, indexed by blockIdx.y)
struct TrackPairDesc {
    TrackLevel lv[KLT_MAX_LEVELS];
    const klt_feat *in;
    klt_feat *out;
};

struct TrackArgs {
    TrackLevel lv[KLT_MAX_LEVELS];      // single-pair launch: levels travel in the kernarg segment
    const klt_feat *in;
    klt_feat *out;
    const TrackPairDesc *pairs;         // batched launch: npairs descriptors (lv / in / out above unused)
    int npairs;
    uint32_t *order;                    // optional: scratch [npairs][n] for the XCD-aware feature order, one permutation per pair (see track_order_kernel)
    int order_chunk;                    // ceil(n / 8): features per XCD
    int order_refresh;                  // 1: sort before this launch; 0: reuse the order of an earlier launch on the same buffer (any
                                        // permutation of 0..n-1 is correct; an old one is only a little less local)
    double half_window;          // window/2 as the Python float (3.5 for 7x7), trackFeatures.py:88-89
    double borderx, bordery;
    int n, nlevels, window, max_iterations, use_max_residue, retain, ncols, nrows;
    float small, th, step, max_residue, ss, inv_ss;
    int tree_sums;               // KLT_OPT_TRACK_TREE_SUMS: the quad kernels add the five sums (and the residue) by a butterfly in registers
};

struct AffineArgs {
    const klt_feat *in;      // records before the translation tracker (frame-1 positions)
    klt_feat *out;           // records after it (updated in place)
    klt_affine_rec *rec;     // per-feature state
    float *tpl;              // [n][3][(height+2)*(width+2)] templates: image, gradx, grady
    const float *i1, *gx1, *gy1, *i2, *gx2, *gy2;     // level 0 of the two frames
    int n, ncols, nrows, mode, width, height, max_iterations;
    float step, small, th, th_aff, max_residue, max_differ;
};

struct SelectArgs {
    const float *sat;        // 3 planes (gxx, gxy, gyy), each ncols*nrows
    float *valmap;           // [ny][nx]
    unsigned long long *keys;
    const uint8_t *seedmap;  // may be null; a pixel is blocked when it holds seed_stamp
    uint8_t seed_stamp;
    const float *val_in;     // test hook: eigenvalues given instead of computed (may be null)
    unsigned *hist, *ticket, *info;   // eigen_hist_kernel: 8192 bins, workgroup ticket, threshold info (hist may be null)
    unsigned hist_target;
    const int *hist_slots;            // optional (device): number of free slots; the cut keeps max(hist_target, hist_per_slot * *hist_slots) keys
    unsigned hist_per_slot;           // (both in units of the sampled histogram: every 4th workgroup counts)
    double min_eig;
    int ncols, nrows, bx, by, step, nx, ny, hw, hh, npow2;
};

struct NmsArgs {
    const unsigned long long *keys;
    klt_feat *fl;
    uint32_t *grid_global;   // used when the cell grid does not fit in LDS
    int *placed_out;         // [0] features placed, [1] 1 if the candidates ran out before the list was full
    int *slots;              // scratch [nfeat]: fillable slot indices (REPLACING_SOME)
    klt_affine_rec *aff_rec; // optional: affine state reset for every slot filled (selectGoodFeatures.py:120-128)
    unsigned cell_magic;     // floor(2^32 / cell) + 1: x / cell == __umulhi(x, cell_magic) for x < 65536
    int nkeys, nfeat, overwrite_all, d /* mindist-1 */, cell, gw, gh, grid_in_lds;
};

// parallel minimum-distance passes (select_kernels.hip)
struct MisArgs {
    const unsigned long long *keys;   // [ny*nx] candidate keys of the eigenvalue pass (0 = no candidate)
    uint32_t *st;                     // [ny*nx] state per candidate cell
    uint32_t *list;                   // [tiles * 1024] undecided cells of every 32x32 tile
    unsigned *cnt;                    // [tiles] length of each tile's list
    unsigned *remaining;              // [rounds] undecided candidates left after pass r
    unsigned long long *acc_keys;     // [tiles * acc_cap] accepted candidates of every tile
    unsigned *acc_cnt;                // [tiles] how many of them
    int acc_cap;
    const unsigned *info;             // info[0] = lowest key bin that takes part (top-K prefilter)
    int nx, ny, R /* exclusion radius in cells */, stage /* 1: stage tile + halo in LDS */;
    int bx, by, step;                 // pixel position of cell (i, j) = (bx + i * step, by + j * step)
    const uint8_t *seed;              // optional: [nrows][ncols] squares of the live features (pixels holding seed_stamp); keys were scored WITHOUT it
    uint8_t seed_stamp;               // (klt_select_prepare_async)
    int ncols;
    int sparse;                       // 1: few candidates per tile are expected (a replacement behind the cut): later passes take several tiles per workgroup
};

// ---- timing by the dispatch's own timestamps --------------------------------------------------------------------------------
// klt_timing_enable(ctx, 2): the level-0 launch is enqueued with hipExtLaunchKernelGGL and a start / stop event pair that the runtime
// fills from the dispatch packet's begin and end timestamps -- the duration rocprofv3 reports for the kernel.  An event pair recorded
// around a launch (mode 1) also holds the boundary between two dependent launches (~2.6 us).  The API sets the two events, the next
// launch that goes through klt_launch takes them.
extern thread_local hipEvent_t g_klt_stamp_start, g_klt_stamp_stop;

#include <hip/hip_ext.h>
template <typename... Args, typename F = void (*)(Args...)>
inline void klt_launch(F kernel, const dim3 &grid, const dim3 &block, unsigned lds, hipStream_t s, Args... args)
{
    if (g_klt_stamp_start) {
        hipEvent_t a = g_klt_stamp_start, b = g_klt_stamp_stop;
        g_klt_stamp_start = g_klt_stamp_stop = nullptr;
        hipExtLaunchKernelGGL(kernel, grid, block, lds, s, a, b, 0, args...);
    } else {
        hipLaunchKernelGGL(kernel, grid, block, lds, s, args...);
    }
}

// ---- launchers (each enqueues on `s`; no synchronisation) ----
void launch_hconv_u8(hipStream_t s, const uint8_t *in, int ncols, int nrows, float *outA, float *outB,
                     int out_cols, int xstride, int xoff, const Taps &ta, const Taps *tb);
void launch_hconv_f32(hipStream_t s, const float *in, int ncols, int nrows, float *outA, float *outB,
                      int out_cols, int xstride, int xoff, const Taps &ta, const Taps *tb);
void launch_take_strided(hipStream_t s, const float *src, float *dst, size_t n, int stride);
// ostride: element stride of the output plane(s) -- KLT_GRAD_STRIDE when outA / outB are the two interleaved gradient planes of a
// level (outB == outA + 1), 1 for separate planes (stated by the caller, never inferred from the pointers)
void launch_vconv(hipStream_t s, const float *inA, const float *inB, int ncols, int nrows, float *outA, float *outB,
                  int out_rows, int ystride, int yoff, const Taps &ta, const Taps *tb, int ostride = 1);

// Plane accesses as raw buffer operations (device code): the plane pointer is workgroup-uniform (a descriptor in four
// SGPRs), the lane's 32-bit BYTE offset is the whole vector address.  With `plane + (size_t)y * nc + x` every access costs a
// 64-bit multiply-add (a quarter-rate instruction) and a 64-bit add; 32-bit offsets are a 24-bit multiply (full rate) and 32-bit adds.  (A plane is far below 2 GB; word 3 of the descriptor: data format 32 bits, raw dwords.)
typedef __amdgpu_buffer_rsrc_t plane_rsrc;
__device__ __forceinline__ plane_rsrc plane_of(const void *p)
{
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(p), 0, 0x7fffffff, 0x00020000);
}
__device__ __forceinline__ void plane_store(plane_rsrc r, unsigned byte_off, float v)
{
    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), r, byte_off, 0, 0);
}
__device__ __forceinline__ void plane_store2(plane_rsrc r, unsigned byte_off, float2 v)
{
    typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
    u32x2 w;
    w.x = __builtin_bit_cast(unsigned, v.x); w.y = __builtin_bit_cast(unsigned, v.y);
    __builtin_amdgcn_raw_buffer_store_b64(w, r, byte_off, 0, 0);
}

__device__ __forceinline__ void plane_store4(plane_rsrc r, unsigned byte_off, float2 a, float2 b)
{
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
    u32x4 w;
    w.x = __builtin_bit_cast(unsigned, a.x); w.y = __builtin_bit_cast(unsigned, a.y);
    w.z = __builtin_bit_cast(unsigned, b.x); w.w = __builtin_bit_cast(unsigned, b.y);
    __builtin_amdgcn_raw_buffer_store_b128(w, r, byte_off, 0, 0);
}
__device__ __forceinline__ float plane_load(plane_rsrc r, unsigned byte_off)
{
    return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, byte_off, 0, 0));
}

extern int g_track_variant;         // tracker kernels: 4 (default) quad-load kernels where they apply, 0 always track_kernel
size_t smooth_grad_lds_bytes(int smooth_radius /* -1: no smoothing stage */, int R);
size_t pyr_reduce_lds_bytes(int ss, int ntaps);
// kind: 0 = u8 frame + smoothing, 1 = f32 frame + smoothing, 2 = f32 image gradients only, 3 = u8 image gradients only
int launch_smooth_grad(hipStream_t s, const SmoothGradArgs &a, int batch, int kind, bool hred = false);
bool smooth_grad_hred_ok(const SmoothGradArgs &a, int batch, int kind, const Taps &reduce, int ss);
int launch_pyr_vreduce(hipStream_t s, const PyrReduceArgs &a, int batch);
int launch_pyr_reduce(hipStream_t s, const PyrReduceArgs &a, int batch);

// Eigenvalue and sort key of one candidate window from its three window sums (goodFeaturesUtils.pyx:17-19 as compiled: (gxx-gyy)^2 in f32,
// 4*gxy*gxy and the sum in f64, pow(., 0.5) -> f32, (gxx+gyy-s) in f32, /2 exact); key 0 = not a candidate (val < max(min_eigenvalue, 1)).
// One definition for every kernel that scores windows (select_kernels.hip, sat_pipeline.hip).
__device__ __forceinline__ float klt_window_value(float gxx, float gxy, float gyy)
{
    const float dif = gxx - gyy;
    const float sq = dif * dif;
    const double t = (double)sq + (4.0 * (double)gxy) * (double)gxy;
    const float s = (float)sqrt(t);
    const float sum = gxx + gyy;
    const float num = sum - s;
    return (float)((double)num / 2.0);
}

__device__ __forceinline__ unsigned long long klt_pack_key(float val, int x, int y)
{
    return ((unsigned long long)__float_as_uint(val) << 32) | ((unsigned long long)x << 16) | (unsigned long long)y;
}

__device__ __forceinline__ unsigned long long klt_window_key(float gxx, float gxy, float gyy, double min_eig, int x, int y, float *val_out)
{
    const float val = klt_window_value(gxx, gxy, gyy);
    *val_out = val;
    return (double)val >= min_eig ? klt_pack_key(val, x, y) : 0ull;          // val >= max(min_eigenvalue, 1) > 0
}

// the smallest f32 that is >= m: for an f32 v, (double)v >= m exactly when v >= klt_threshold_f32(m) -- the test without the conversion
// and the f64 compare (a kernel that scores windows waits for its CU's f64 pipes)
inline float klt_threshold_f32(double m)
{
    float t = (float)m;
    if ((double)t < m) t = nextafterf(t, INFINITY);
    return t;
}

void launch_sat_rows(hipStream_t s, const float *gx, const float *gy, float *sat, int ncols, int nrows);
void launch_sat_cols(hipStream_t s, float *sat, int ncols, int nrows);
// step-synchronous wavefront pipelines (sat_pipeline.hip); return 0 or a hipError_t
int  launch_sat_rows_pipe(hipStream_t s, const float *gx, const float *gy, float *sat, int ncols, int nrows);
int  launch_sat_cols_pipe(hipStream_t s, float *sat, int ncols, int nrows);
// column pass + eigenvalue keys in one launch (the column-summed tables never reach HBM); `sat` holds the ROW-summed planes, followed by
// a pad of KLT_SAT_PAD floats (the last strip reads past a row's end), and is only read; keys only (no seed map, value map, histogram or given values); -1 = not applicable (the caller runs the two separate kernels)
constexpr int KLT_SAT_PAD = 64;
bool sat_cols_eigen_ok(const SelectArgs &a);
int  launch_sat_cols_eigen_pipe(hipStream_t s, const float *sat, const SelectArgs &a);
void launch_seed_fill(hipStream_t s, const klt_feat *fl, int nfeat, uint8_t *seedmap, int ncols, int nrows, int d, uint8_t stamp);
void launch_eigen(hipStream_t s, const SelectArgs &a);
void launch_sort_desc(hipStream_t s, unsigned long long *keys, int npow2);
void launch_topk_prefilter(hipStream_t s, const unsigned long long *keys, int n, unsigned target, unsigned *hist,
                           unsigned *info /* [0] bin, [1] kept (histogram), [2] valid, [3] compaction counter */, unsigned long long *out);
int  launch_nms(hipStream_t s, const NmsArgs &a);   // returns 0 or a hipError_t
void launch_key_threshold(hipStream_t s, const unsigned long long *keys, int n, unsigned target, unsigned *hist, unsigned *info);
int  mis_tiles(int nx, int ny);
size_t mis_stage_bytes(int R);
void launch_mis_init(hipStream_t s, const MisArgs &a);
int  launch_mis_round(hipStream_t s, const MisArgs &a, int round);   // returns 0 or a hipError_t
void launch_mis_compact(hipStream_t s, const MisArgs &a, unsigned long long *out, unsigned *out_count);
int  mis_tile_capacity(int R);
void launch_zero_words(hipStream_t s, unsigned *p, size_t n);
void launch_mis_prepare(hipStream_t s, const klt_feat *fl, int nfeat, int overwrite_all, int *slots, int *nfill_out, klt_feat *snapshot,
                        unsigned *zero, size_t zero_n);
void launch_eigen_hist(hipStream_t s, const SelectArgs &a);
// histogram (every 4th block of 256 keys, as eigen_hist_kernel samples) of the keys outside the seed map, then the threshold
void launch_mask_hist(hipStream_t s, const SelectArgs &a);
void launch_mis_results(hipStream_t s, unsigned *host_out, const unsigned *rem, int look, const unsigned *info, const int *placed);
void launch_mis_place(hipStream_t s, const NmsArgs &a, const unsigned *count, unsigned *rank, const int *nfill, int bound,
                      unsigned *host_out, const unsigned *rem, int look, const unsigned *info);
void launch_unpack_candidates(hipStream_t s, const unsigned long long *keys, int n, float *val, int *x, int *y);

void launch_track_stats(hipStream_t s, const klt_feat *in, const klt_feat *out, int n, int nlevels, unsigned long long *stats);
int launch_track(hipStream_t s, const TrackArgs &a);
void launch_extract_patch(hipStream_t s, const float *img, int nc, int nr, float x, float y, int w, float *patch, int *bad);
void launch_track_iterate(hipStream_t s, const float *t_gx, const float *t_gy, const float *t_i, const float *i2, const float *gx2,
                          const float *gy2, int nc, int nr, int w, float x2, float y2, float step, float small, float th,
                          int max_iterations, float *res);
void launch_affine(hipStream_t s, const AffineArgs &a);
void launch_affine_reset(hipStream_t s, klt_affine_rec *rec, int n);   // returns 0, or -1 for an unsupported window

// ---- multi-GPU (comm.hip): RCCL communicator + side stream; every function returns 0 or a klt_status and fills `err` ----
struct KltComm;
int  comm_unique_id(void *out128, std::string &err);
int  comm_create(int device, int nranks, int rank, const void *unique_id, KltComm **out, std::string &err);
void comm_destroy(KltComm *k);
bool comm_poisoned(const KltComm *k);         // a host-side wait timed out: nothing may wait for this communicator's stream again
int  comm_nranks(const KltComm *k);
hipEvent_t comm_last_done(const KltComm *k);   // end of the most recent collective (an event of the communicator's ring)
int  comm_rank(const KltComm *k);
int  comm_allgather(KltComm *k, hipStream_t producer, const void *src, void *dst, size_t bytes, std::string &err);
int  comm_gather(KltComm *k, hipStream_t producer, const void *src, void *dst, size_t bytes, int root, std::string &err);
int  comm_gatherv(KltComm *k, hipStream_t producer, const void *src, void *dst, const size_t *counts, int root, std::string &err);
void comm_set_timeout(KltComm *k, double ms);
int  comm_sendrecv(KltComm *k, hipStream_t producer, const void *src, int to, void *dst, int from, size_t bytes, std::string &err);
int  comm_fence(KltComm *k, hipStream_t consumer, std::string &err);
int  comm_wait(KltComm *k, std::string &err);
int  comm_allreduce_max(KltComm *k, double *inout, int n, std::string &err);
