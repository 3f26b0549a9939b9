// Affine consistency check (BASELINE cfg-3) -- one wavefront per feature.
//
// PARITY UNPINNED: the reference calls _am_trackFeatureAffine / _am_getSubFloatImage
// (trackFeatures.py:347-399) but never defines them.  This kernel implements upstream KLT 1.3.4's
// behaviour behind the interface the reference's call site pins: after the
// translation tracker has succeeded,
//   * first success of a feature: cut (affine_window+2)^2 templates (image, gradx, grady) out of level 0
//     of frame 1 around (int)x, (int)y and remember the sub-pixel offset;
//   * later calls: Newton iterations on [dAxx dAyx dAxy dAyy dx dy] (6x6 normal equations; 4x4 for the
//     similarity model; 2x2 for translation only) of the template against level 0 of frame 2; the feature
//     is lost unless this returns KLT_TRACKED; on success the translation result is kept.
// Window sums: there is no reference summation order to reproduce here, so the order is part of the behaviour
// specification and the CPU checker under oracle/ restates it exactly (its am_fold): term k
// (row-major window index) is added, in increasing k, to the partial sum of lane k mod 64; the 64 partials are folded by
// the butterfly p[l] + p[l ^ m], m = 32 .. 1.  Statuses, positions and A matrices are bit-identical to the oracle's.
#include "klt_internal.h"

#pragma clang fp contract(off)

namespace {

// ST = element stride of the plane: 1 (image planes, the kernel's own template copies) or KLT_GRAD_STRIDE (one of the two interleaved
// gradient planes of a frame; `img` is then the plane's own first element)
template <int ST = 1>
__device__ __forceinline__ float bilinear_at(const float *__restrict__ img, int nc, float x, float y)
{
    const int ix = (int)x, iy = (int)y;
    const float ax = (float)((double)x - (double)ix), ay = (float)((double)y - (double)iy);
    const double w00 = (1. - (double)ax) * (1. - (double)ay), w01 = (double)ax * (1. - (double)ay), w10 = (1. - (double)ax) * (double)ay;
    const float w11 = ax * ay;
    // raw buffer loads from the (wave-uniform) plane pointer: the descriptor in SGPRs, one VGPR with a 32-bit byte offset per
    // address instead of a 64-bit pointer pair (the model-2 kernel has 48 of them in flight and was at 200+ VGPRs), the row offset a
    // 24-bit multiply (rows and row lengths are far below 2^24; the 32-bit and 64-bit integer multiplies are quarter-rate)
    const plane_rsrc r = plane_of(img);
    const unsigned o = 4u * ST * (__umul24((unsigned)iy, (unsigned)nc) + (unsigned)ix), down = 4u * ST * (unsigned)nc;
    const float t4 = w11 * plane_load(r, o + down + 4u * ST);
    double v = w00 * (double)plane_load(r, o);
    v = v + w01 * (double)plane_load(r, o + 4u * ST);
    v = v + w10 * (double)plane_load(r, o + down);
    v = v + (double)t4;
    return (float)v;
}

__device__ __forceinline__ float wave_sum(float v)
{
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v = v + __shfl_xor(v, m);
    return v;
}

// N (a power of two <= 32) sums over the wavefront at once.  Same additions as N calls of wave_sum -- value j goes through
// p[l] + p[l ^ 32], then ^ 16, ^ 8, ... -- but at every step a lane keeps only the half of the values its bit selects and hands the
// other half to its partner, so the step costs N/2, N/4, ... shuffles instead of N: 31 + 1 instead of 32 x 6 for N = 32.  On return
// v[0] of lane l holds the total of value (l >> s) & (N - 1), s = 6 - log2(N) (every total sits on 2^s adjacent lanes).
// (Every step is a template instantiation with its own constant H and M: written as one loop over run-time copies of them, the
// compiler indexed v[] dynamically -- 27-way compare / select chains, 860 of the 1620 vector instructions of an iteration.)
template <int N, int H, int M>
struct WaveSumScatter {
    static __device__ __forceinline__ void run(float (&v)[N], int lane)
    {
        typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
        if constexpr (H >= 1) {
            if constexpr (M == 32) {
                // gfx950's row swaps: v_permlane32_swap exchanges lanes 32-63 of its first operand with lanes 0-31 of its second, so
                // first + second is value i summed over (l, l ^ 32) in the lower half of the wavefront and value i + H in the upper
                // half -- one swap and one add per pair of values, no selects (v_permlane16_swap below: the same with rows of 16 lanes)
#pragma unroll
                for (int i = 0; i < H; i++) {
                    const u32x2 r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v[i]), __float_as_uint(v[i + H]), false, false);
                    v[i] = __uint_as_float(r.x) + __uint_as_float(r.y);
                }
            } else if constexpr (M == 16) {
#pragma unroll
                for (int i = 0; i < H; i++) {
                    const u32x2 r = __builtin_amdgcn_permlane16_swap(__float_as_uint(v[i]), __float_as_uint(v[i + H]), false, false);
                    v[i] = __uint_as_float(r.x) + __uint_as_float(r.y);
                }
            } else {
                const bool up = (lane & M) != 0;
#pragma unroll
                for (int i = 0; i < H; i++) {
                    const float send = up ? v[i] : v[i + H];
                    const float keep = up ? v[i + H] : v[i];
                    v[i] = keep + __shfl_xor(send, M);
                }
            }
            WaveSumScatter<N, H / 2, M / 2>::run(v, lane);
        } else if constexpr (M >= 1) {
            v[0] = v[0] + __shfl_xor(v[0], M);
            WaveSumScatter<N, 0, M / 2>::run(v, lane);
        }
    }
};

template <int N>
__device__ __forceinline__ void wave_sum_scatter(float (&v)[N], int lane)
{
    WaveSumScatter<N, N / 2, 32>::run(v, lane);
}

// Numerical Recipes' gaussj with full pivoting, as upstream's _am_gauss_jordan_elimination, spread over the wavefront:
// lane r * 6 + c holds a[r][c] (lanes 0..35), lane 36 + r holds b[r].  Every element goes through exactly the operations
// of the sequential routine (there are no sums), so the result is the sequential one; the pivot search reproduces the
// scan order of the original (`>=`: the last largest element in (row, column) order wins).  Returns the status; the
// solution is left in the b lanes.
__device__ __forceinline__ int gauss_jordan_wave(float &val, int n, int lane)
{
    const int r = lane < 36 ? lane / 6 : lane - 36, c = lane < 36 ? lane % 6 : -1;
    const bool is_a = lane < 36 && r < n && c < n, is_b = lane >= 36 && lane < 42 && r < n;
    unsigned used = 0u;                                     // bit k: column k has been a pivot (ipiv[k] == 1)
    for (int i = 0; i < n; i++) {
        const bool eligible = is_a && !((used >> r) & 1u) && !((used >> c) & 1u);
        const float mag = eligible ? fabsf(val) : -1.f;
        float big = mag;
#pragma unroll
        for (int m = 32; m >= 1; m >>= 1) big = fmaxf(big, __shfl_xor(big, m));
        const unsigned long long at_max = __ballot(eligible && mag == big);
        if (at_max == 0ull) return KLT_SMALL_DET;            // NaNs: nothing compares equal
        const int sel = 63 - __builtin_clzll(at_max);
        const int irow = sel / 6, icol = sel % 6;
        used |= 1u << icol;
        if (irow != icol) {                                  // swap rows irow and icol of a and b
            const int partner = is_a ? (r == irow ? icol * 6 + c : (r == icol ? irow * 6 + c : lane))
                                     : (is_b ? (r == irow ? 36 + icol : (r == icol ? 36 + irow : lane)) : lane);
            val = __shfl(val, partner);
        }
        const float piv = __shfl(val, icol * 6 + icol);
        if (piv == 0.f) return KLT_SMALL_DET;
        const float pivinv = 1.0f / piv;
        if (lane == icol * 6 + icol) val = 1.0f;
        if ((is_a || is_b) && r == icol) val = val * pivinv;
        const float pivot_row = __shfl(val, is_b ? 36 + icol : icol * 6 + (c < 0 ? 0 : c));      // a[icol][l] or b[icol]
        const float dum = __shfl(val, r * 6 + icol);                                                  // a[ll][icol]
        if ((is_a || is_b) && r != icol) {
            const float base = (is_a && c == icol) ? 0.f : val;
            val = base - pivot_row * dum;
        }
    }
    return KLT_TRACKED;
}

// MODE = tc.affineConsistencyCheck (0 translation, 1 similarity, 2 affine): one instantiation per model keeps the other
// models' accumulators out of the register file
template <int MODE>
__global__ __launch_bounds__(64) void affine_kernel(AffineArgs a)
{
    const int f = blockIdx.x, lane = threadIdx.x;
    if (f >= a.n) return;
    const klt_feat before = a.in[f];
    if (before.val < 0) return;
    klt_feat after = a.out[f];
    klt_affine_rec st = a.rec[f];
    if (after.val != KLT_TRACKED) {                       // lost by the translation tracker: template freed
        if (lane == 0 && (st.valid || st.pad)) { st.valid = 0; st.pad = 0; a.rec[f] = st; }
        return;
    }
    const int width = a.width, height = a.height, hw = width / 2, hh = height / 2, n = width * height;
    const int tw = width + 2, th = height + 2, tn = tw * th;
    float *tpl = a.tpl + (size_t)f * 3 * tn;
    const int nc = a.ncols, nr = a.nrows;

    if (!st.valid) {
        // _am_getSubFloatImage: integer-aligned copy around (int)x, (int)y of the feature's frame-1 position
        const int x0 = (int)before.x, y0 = (int)before.y, thw = tw / 2, thh = th / 2;
        if (x0 - thw < 0 || y0 - thh < 0 || x0 + thw >= nc || y0 + thh >= nr) return;     // upstream asserts; stay template-less
        for (int k = lane; k < tn; k += 64) {
            const size_t o = (size_t)(y0 - thh + k / tw) * nc + (x0 - thw + k % tw);
            tpl[k] = a.i1[o];
            tpl[tn + k] = a.gx1[KLT_GRAD_STRIDE * o];                 // interleaved gradient planes (klt_internal.h)
            tpl[2 * tn + k] = a.gy1[KLT_GRAD_STRIDE * o];
        }
        if (lane == 0) {
            st.aff_x = before.x - (float)x0 + (float)thw;
            st.aff_y = before.y - (float)y0 + (float)thh;
            st.Axx = 1.f; st.Ayx = 0.f; st.Axy = 0.f; st.Ayy = 1.f;
            st.valid = 1;
            st.pad = 0;
            a.rec[f] = st;
        }
        return;
    }

    const float one_plus_eps = 1.001f;
    const float x1 = st.aff_x, y1 = st.aff_y;
    float x2 = after.x, y2 = after.y;
    const float old_x2 = x2, old_y2 = y2;
    float Axx = st.Axx, Ayx = st.Ayx, Axy = st.Axy, Ayy = st.Ayy;
    const float *t_img = tpl, *t_gx = tpl + tn, *t_gy = tpl + 2 * tn;
    // The template is sampled at (x1 + i, y1 + j), the same positions in every iteration: sampled once into LDS (lane l owns window
    // pixels l, l + 64, ...).  The window loops below are NOT unrolled: four samples at a time with their 48 bilinear reads and 27
    // accumulators in flight needed 250+ VGPRs (one wavefront per SIMD); one sample at a time needs under 100 and five wavefronts
    // per SIMD hide the latency instead.
    extern __shared__ float tsamp[];                      // [n] image (+ [n] gradx, [n] grady for the translation model), then the offsets
    // window offsets (i, j) of every sample as floats, once per feature: `k % width` and `k / width` with a run-time width are ~25
    // integer instructions each, and the sample loops below ran them for every sample of every Newton iteration -- 28 % of this kernel's
    // 29.9 M wavefront-instructions were integer arithmetic (profiles/history/r02_g_cfg3_sq_counters.json), and the kernel sits on its issue roof
    float *const offx = tsamp + (MODE == 0 ? 3 * n : n), *const offy = offx + n;
    for (int k = lane; k < n; k += 64) {
        const float fi = (float)(k % width - hw), fj = (float)(k / width - hh);
        offx[k] = fi;
        offy[k] = fj;
        tsamp[k] = bilinear_at(t_img, tw, x1 + fi, y1 + fj);
        if (MODE == 0) {
            tsamp[n + k] = bilinear_at(t_gx, tw, x1 + fi, y1 + fj);
            tsamp[2 * n + k] = bilinear_at(t_gy, tw, x1 + fi, y1 + fj);
        }
    }
    // body(window offset x, y, template image, template gradx, template grady) for every window pixel of this lane, in increasing k
    auto for_samples = [&](auto body) {
#pragma unroll 1
        for (int k = lane; k < n; k += 64) body(offx[k], offy[k], tsamp[k], MODE == 0 ? tsamp[n + k] : 0.f, MODE == 0 ? tsamp[2 * n + k] : 0.f);
    };
    const float sxs[4] = {-(float)hw, -(float)hw, (float)hw, (float)hw};
    const float sys[4] = {(float)hh, -(float)hh, (float)hh, -(float)hh};
    int iteration = 0, status = KLT_TRACKED;
    bool convergence = false;
    do {
        float dx = 0.f, dy = 0.f;
        if (MODE == 0) {
            if (x1 - hw < 0.0f || tw - (x1 + hw) < one_plus_eps || x2 - hw < 0.0f || nc - (x2 + hw) < one_plus_eps ||
                y1 - hh < 0.0f || th - (y1 + hh) < one_plus_eps || y2 - hh < 0.0f || nr - (y2 + hh) < one_plus_eps) {
                status = KLT_OOB;
                break;
            }
            float gxx = 0.f, gxy = 0.f, gyy = 0.f, ex = 0.f, ey = 0.f;
            for_samples([&](float fi, float fj, float ti, float tgx, float tgy) {
                const float d = ti - bilinear_at(a.i2, nc, x2 + fi, y2 + fj);
                const float g1 = tgx + bilinear_at<KLT_GRAD_STRIDE>(a.gx2, nc, x2 + fi, y2 + fj);
                const float g2 = tgy + bilinear_at<KLT_GRAD_STRIDE>(a.gy2, nc, x2 + fi, y2 + fj);
                gxx = gxx + g1 * g1; gxy = gxy + g1 * g2; gyy = gyy + g2 * g2;
                ex = ex + d * g1; ey = ey + d * g2;
            });
            gxx = wave_sum(gxx); gxy = wave_sum(gxy); gyy = wave_sum(gyy);
            ex = wave_sum(ex) * a.step; ey = wave_sum(ey) * a.step;
            const float det = gxx * gyy - gxy * gxy;
            if (det < a.small) { status = KLT_SMALL_DET; }
            else { dx = (gyy * ex - gxy * ey) / det; dy = (gxx * ey - gxy * ex) / det; status = KLT_TRACKED; }
            convergence = fabsf(dx) < a.th && fabsf(dy) < a.th;
            x2 = x2 + dx; y2 = y2 + dy;
        } else {
            float cx[4], cy[4];
            bool oob = (x1 - hw < 0.0f || tw - (x1 + hw) < one_plus_eps || y1 - hh < 0.0f || th - (y1 + hh) < one_plus_eps);
#pragma unroll
            for (int k = 0; k < 4; k++) {                // ul, ll, ur, lr corners of the warped window
                cx[k] = Axx * sxs[k] + Axy * sys[k] + x2;
                cy[k] = Ayx * sxs[k] + Ayy * sys[k] + y2;
                if (cx[k] < 0.0f || nc - cx[k] < one_plus_eps || cy[k] < 0.0f || nr - cy[k] < one_plus_eps) oob = true;
            }
            if (oob) { status = KLT_OOB; break; }
            float T[6][6], e[6];
            for (int r = 0; r < 6; r++) { e[r] = 0.f; for (int q = 0; q < 6; q++) T[r][q] = 0.f; }
            const int nn = MODE == 1 ? 4 : 6;
            for_samples([&](float x, float y, float ti, float, float) {
                const float mi = Axx * x + Axy * y, mj = Ayx * x + Ayy * y;
                const float d = ti - bilinear_at(a.i2, nc, x2 + mi, y2 + mj);
                const float g1 = bilinear_at<KLT_GRAD_STRIDE>(a.gx2, nc, x2 + mi, y2 + mj);
                const float g2 = bilinear_at<KLT_GRAD_STRIDE>(a.gy2, nc, x2 + mi, y2 + mj);
                if (MODE == 1) {
                    const float u = x * g1 + y * g2, v = x * g2 - y * g1;
                    e[0] = e[0] + (d * g1 * x + d * g2 * y); e[1] = e[1] + (d * g2 * x - d * g1 * y);
                    e[2] = e[2] + d * g1; e[3] = e[3] + d * g2;
                    T[0][0] = T[0][0] + u * u; T[0][1] = T[0][1] + u * v; T[0][2] = T[0][2] + u * g1; T[0][3] = T[0][3] + u * g2;
                    T[1][1] = T[1][1] + v * v; T[1][2] = T[1][2] + v * g1; T[1][3] = T[1][3] + v * g2;
                    T[2][2] = T[2][2] + g1 * g1; T[2][3] = T[2][3] + g1 * g2; T[3][3] = T[3][3] + g2 * g2;
                } else {
                    const float dgx = d * g1, dgy = d * g2, gxx = g1 * g1, gxy = g1 * g2, gyy = g2 * g2;
                    const float xx = x * x, xy = x * y, yy = y * y;
                    e[0] = e[0] + dgx * x; e[1] = e[1] + dgy * x; e[2] = e[2] + dgx * y; e[3] = e[3] + dgy * y;
                    e[4] = e[4] + dgx; e[5] = e[5] + dgy;
                    T[0][0] = T[0][0] + xx * gxx; T[0][1] = T[0][1] + xx * gxy; T[0][2] = T[0][2] + xy * gxx;
                    T[0][3] = T[0][3] + xy * gxy; T[0][4] = T[0][4] + x * gxx; T[0][5] = T[0][5] + x * gxy;
                    T[1][1] = T[1][1] + xx * gyy; T[1][2] = T[1][2] + xy * gxy; T[1][3] = T[1][3] + xy * gyy;
                    T[1][4] = T[1][4] + x * gxy; T[1][5] = T[1][5] + x * gyy;
                    T[2][2] = T[2][2] + yy * gxx; T[2][3] = T[2][3] + yy * gxy; T[2][4] = T[2][4] + y * gxx; T[2][5] = T[2][5] + y * gxy;
                    T[3][3] = T[3][3] + yy * gyy; T[3][4] = T[3][4] + y * gxy; T[3][5] = T[3][5] + y * gyy;
                    T[4][4] = T[4][4] + gxx; T[4][5] = T[4][5] + gxy; T[5][5] = T[5][5] + gyy;
                }
            });
            // the wave-wide sums of e[] and of the upper triangle of T (27 values; 14 for the similarity model), then every lane
            // fetches its element of the normal equations: lane r * 6 + c holds T[r][c], lane 36 + r holds 0.5 e[r]
            const int lr = lane < 36 ? lane / 6 : lane - 36, lc = lane < 36 ? lane % 6 : 0;
            const int tr = lr < lc ? lr : lc, tc = lr < lc ? lc : lr;             // upper-triangle coordinates of my element
            float mine;
            if (MODE == 1) {
                float v[16];
                int k = 0;
#pragma unroll
                for (int r = 0; r < 4; r++) v[k++] = e[r];
#pragma unroll
                for (int r = 0; r < 4; r++)
#pragma unroll
                    for (int q = r; q < 4; q++) v[k++] = T[r][q];
                v[14] = v[15] = 0.f;
                wave_sum_scatter<16>(v, lane);
                // value index of (tr, tc) in the order above: 4 + tr * 4 - tr (tr - 1) / 2 + (tc - tr)
                const int idx = lane >= 36 ? lr : 4 + tr * 4 - (tr * (tr - 1)) / 2 + (tc - tr);
                mine = __shfl(v[0], (idx & 15) << 2);
            } else {
                float v[32];
                int k = 0;
#pragma unroll
                for (int r = 0; r < 6; r++) v[k++] = e[r];
#pragma unroll
                for (int r = 0; r < 6; r++)
#pragma unroll
                    for (int q = r; q < 6; q++) v[k++] = T[r][q];
#pragma unroll
                for (int z = 27; z < 32; z++) v[z] = 0.f;
                wave_sum_scatter<32>(v, lane);
                const int idx = lane >= 36 ? lr : 6 + tr * 6 - (tr * (tr - 1)) / 2 + (tc - tr);
                mine = __shfl(v[0], (idx & 31) << 1);
            }
            if (lane >= 36) mine = mine * 0.5f;
            if (lane >= 42 || (lane < 36 && (lr >= nn || lc >= nn)) || (lane >= 36 && lr >= nn)) mine = 0.f;
            status = gauss_jordan_wave(mine, nn, lane);
#pragma unroll
            for (int r = 0; r < 6; r++) e[r] = __shfl(mine, 36 + r);
            if (MODE == 1) { Axx = Axx + e[0]; Ayx = Ayx + e[1]; Ayy = Axx; Axy = -Ayx; dx = e[2]; dy = e[3]; }
            else { Axx = Axx + e[0]; Ayx = Ayx + e[1]; Axy = Axy + e[2]; Ayy = Ayy + e[3]; dx = e[4]; dy = e[5]; }
            x2 = x2 + dx; y2 = y2 + dy;
            convergence = fabsf(dx) < a.th && fabsf(dy) < a.th;
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const float ddx = cx[k] - (Axx * sxs[k] + Axy * sys[k] + x2), ddy = cy[k] - (Ayx * sxs[k] + Ayy * sys[k] + y2);
                if (!(fabsf(ddx) < a.th_aff && fabsf(ddy) < a.th_aff)) convergence = false;
            }
        }
        if (status == KLT_SMALL_DET) break;
        iteration++;
    } while (!convergence && iteration < a.max_iterations);

    if (x2 - hw < 0.0f || nc - (x2 + hw) < one_plus_eps || y2 - hh < 0.0f || nr - (y2 + hh) < one_plus_eps) status = KLT_OOB;
    if ((x2 - old_x2) > a.max_differ || (y2 - old_y2) > a.max_differ) status = KLT_OOB;
    if (status == KLT_TRACKED) {
        float s = 0.f;
        for_samples([&](float x, float y, float ti, float, float) {
            const float mi = MODE ? Axx * x + Axy * y : x, mj = MODE ? Ayx * x + Ayy * y : y;
            s = s + fabsf(ti - bilinear_at(a.i2, nc, x2 + mi, y2 + mj));
        });
        s = wave_sum(s);
        if (s / (float)n > a.max_residue) status = KLT_LARGE_RESIDUE;
    }
    if (lane == 0) {
        st.Axx = Axx; st.Ayx = Ayx; st.Axy = Axy; st.Ayy = Ayy;
        st.pad = iteration;                                  // iterations of this check (bench.py's algorithmic bytes)
        if (status != KLT_TRACKED) {
            after.x = -1.f; after.y = -1.f;
            st.aff_x = -1.f; st.aff_y = -1.f; st.valid = 0;
        }
        after.val = status;                                  // translation position kept on success (:396-399)
        a.out[f] = after;
        a.rec[f] = st;
    }
}

__global__ void affine_reset_kernel(klt_affine_rec *rec, int n)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    klt_affine_rec r;
    r.aff_x = -1.f; r.aff_y = -1.f; r.Axx = 1.f; r.Ayx = 0.f; r.Axy = 0.f; r.Ayy = 1.f; r.valid = 0; r.pad = 0;
    rec[i] = r;
}

}  // namespace

void launch_affine(hipStream_t s, const AffineArgs &a)
{
    if (a.n <= 0) return;
    const size_t n = (size_t)a.width * a.height;
    // template samples + the two offset arrays
    if (a.mode == 0) klt_launch(affine_kernel<0>, dim3(a.n), dim3(64), (unsigned)(5 * n * sizeof(float)), s, a);
    else if (a.mode == 1) klt_launch(affine_kernel<1>, dim3(a.n), dim3(64), (unsigned)(3 * n * sizeof(float)), s, a);
    else klt_launch(affine_kernel<2>, dim3(a.n), dim3(64), (unsigned)(3 * n * sizeof(float)), s, a);
}

void launch_affine_reset(hipStream_t s, klt_affine_rec *rec, int n)
{
    if (n <= 0) return;
    hipLaunchKernelGGL(affine_reset_kernel, dim3((n + 255) / 256), dim3(256), 0, s, rec, n);
}
