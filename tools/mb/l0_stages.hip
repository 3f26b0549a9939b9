// Stage timing of the level-0 kernel: includes the product source with KLT_STAGE_CLOCKS and prints, for 64 workgroups of one
// tile row, the wall-clock ticks (100 MHz) spent between the stage marks.  Build (from the repo root):
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off -DKLT_STAGE_CLOCKS -Iinclude -Ipyfeaturetrack_amd/csrc tools/mb/l0_stages.hip -o tools/mb/l0_stages
#include "../../pyfeaturetrack_amd/csrc/pyramid_kernels.hip"
#include <cstdio>
#include <cmath>
#include <vector>

static void gauss_taps(Taps &g, Taps &d, double sigma, int n)
{
    const int h = n / 2;
    double sg = 0, sd = 0;
    for (int i = -h; i <= h; i++) { g.k[i + h] = exp(-i * i / (2 * sigma * sigma)); d.k[i + h] = -i * g.k[i + h]; sg += g.k[i + h]; sd -= i * d.k[i + h]; }
    for (int i = 0; i < n; i++) { g.k[i] /= sg; d.k[i] /= sd; }
    g.n = d.n = n; g.sym = 1; d.sym = -1;
}

int main(int argc, char **argv)
{
    const bool reduce = argc > 1 && argv[1][0] == 'r';
    const bool level = false;
    const int nc = 1920, nr = 1080;
    SmoothGradArgs a = {};
    Taps dummy;
    gauss_taps(a.smooth, dummy, 0.7, 5);
    gauss_taps(a.ggauss, a.gderiv, 1.0, 7);
    a.ncols = nc; a.nrows = nr; a.R = 3;
    std::vector<uint8_t> h((size_t)nc * nr);
    for (size_t i = 0; i < h.size(); i++) h[i] = (uint8_t)((i * 2654435761u) >> 24);
    for (int b = 0; b < 2; b++) {
        uint8_t *raw; float *img, *gx, *gy;
        hipMalloc(&raw, h.size()); hipMalloc(&img, 4 * h.size()); hipMalloc(&gx, 4 * h.size()); hipMalloc(&gy, 4 * h.size());
        hipMemcpy(raw, h.data(), h.size(), hipMemcpyHostToDevice);
        a.raw[b] = raw; a.img[b] = img; a.gx[b] = gx; a.gy[b] = gy;
    }
    PyrReduceArgs pr = {};
    gauss_taps(pr.taps, dummy, 3.6, 21);
    pr.src_nc = nc; pr.src_nr = nr; pr.dst_nc = nc / 4; pr.dst_nr = nr / 4; pr.ss = 4; pr.log2ss = 2;
    for (int b = 0; b < 2; b++) { pr.src[b] = a.img[b]; pr.dst[b] = a.gx[b]; }
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const bool plain = argc > 1 && argv[1][0] == 'p';      // the kernel without the fused horizontal reduction
    a.reduce = pr.taps; a.h1_nc = nc / 4;
    for (int b = 0; b < 2; b++) { float *h1; hipMalloc(&h1, 4 * (size_t)nr * (nc / 4)); a.h1[b] = h1; }
    auto go = [&]() { if (reduce) launch_pyr_reduce(0, pr, 2); else launch_smooth_grad(0, a, 2, 0, !plain); };
    for (int rep = 0; rep < 3; rep++) go();
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int rep = 0; rep < 20; rep++) go();
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("kernel: %.2f us per launch\n", ms * 1000 / 20);
    long long clk[64 * 8];
    hipMemcpyFromSymbol(clk, HIP_SYMBOL(g_stage_clk), sizeof(clk));
    long long t0 = clk[0];
    const int nrow = level ? 8 : reduce ? 15 : 30;
    for (int b = 0; b < nrow; b++) t0 = clk[b * 8] < t0 ? clk[b * 8] : t0;
    printf("ticks of 10 ns; tile row 8 of frame 0\nblock  start | load  hsm  vsm  hgrad vgrad | total\n");
    for (int b = 0; b < nrow; b++) {
        const long long *c = clk + b * 8;
        if (reduce) printf("%4d %6lld | load %4lld  hpass %4lld  vpass %4lld | %5lld\n", b, c[0] - t0, c[1] - c[0], c[2] - c[1], c[5] - c[2], c[5] - c[0]);
        else printf("%4d %6lld | %4lld %4lld %4lld %4lld %4lld | %5lld\n", b, c[0] - t0, c[1] - c[0], c[2] - c[1], c[3] - c[2], c[4] - c[3], c[5] - c[4], c[5] - c[0]);
    }
    static long long bc[8192 * 2];
    static unsigned hw[8192];
    hipMemcpyFromSymbol(bc, HIP_SYMBOL(g_block_clk), sizeof(bc));
    hipMemcpyFromSymbol(hw, HIP_SYMBOL(g_block_hw), sizeof(hw));
    const int nb = level ? 8 * 17 * 2 : reduce ? 15 * 34 * 2 : 30 * 34 * 2;
    long long b0 = bc[0], b1 = 0;
    for (int i = 0; i < nb; i++) { if (bc[2 * i] < b0) b0 = bc[2 * i]; if (bc[2 * i + 1] > b1) b1 = bc[2 * i + 1]; }
    printf("first start -> last end: %lld ticks\n", b1 - b0);
    int hs[64] = {}, he[64] = {}; double dur[64] = {}; int nd[64] = {};
    for (int i = 0; i < nb; i++) {
        const int s = (int)((bc[2 * i] - b0) / 100), e = (int)((bc[2 * i + 1] - b0) / 100);
        hs[s < 63 ? s : 63]++; he[e < 63 ? e : 63]++; dur[s < 63 ? s : 63] += bc[2 * i + 1] - bc[2 * i]; nd[s < 63 ? s : 63]++;
    }
    printf("per microsecond: workgroups started / ended / mean duration (ticks) of those started\n");
    for (int t = 0; t < 40; t++) printf("%3d us: %5d %5d %7.0f\n", t, hs[t], he[t], nd[t] ? dur[t] / nd[t] : 0.0);
    // residency: which workgroups ran on each (xcc, se, sh, cu)
    static int cnt[8 * 256]; static long long lastend[8 * 256]; static long long firstend[8 * 256];
    for (int i = 0; i < 8 * 256; i++) { cnt[i] = 0; lastend[i] = 0; firstend[i] = 1 << 30; }
    for (int i = 0; i < nb; i++) {
        const unsigned h = hw[i] & 0xffff, x = (hw[i] >> 16) & 0x7;
        const int key = (int)x * 256 + (int)((h >> 8) & 0xff);       // cu_id[3:0] sh_id se_id[2:0]
        cnt[key]++;
        const long long e = bc[2 * i + 1] - b0;
        if (e > lastend[key]) lastend[key] = e;
        if (e < firstend[key]) firstend[key] = e;
    }
    int used = 0, hist[32] = {};
    for (int i = 0; i < 8 * 256; i++) if (cnt[i]) { used++; hist[cnt[i] < 31 ? cnt[i] : 31]++; }
    printf("CUs used %d; CUs by number of workgroups they ran:", used);
    for (int c = 0; c < 32; c++) if (hist[c]) printf("  %d wg: %d CUs", c, hist[c]);
    printf("\nper XCC: CUs, workgroups, mean first end, mean last end\n");
    for (int x = 0; x < 8; x++) {
        int c = 0, w = 0; double fe = 0, le = 0;
        for (int i = 0; i < 256; i++) if (cnt[x * 256 + i]) { c++; w += cnt[x * 256 + i]; fe += firstend[x * 256 + i]; le += lastend[x * 256 + i]; }
        if (c) printf("  xcc %d: %3d CUs %4d wgs  first end %.0f  last end %.0f\n", x, c, w, fe / c, le / c);
    }
    printf("xcc 0, per CU: id count firstend lastend\n");
    for (int i = 0; i < 256; i++) if (cnt[i]) printf("   %02x %2d %5lld %5lld\n", i, cnt[i], firstend[i], lastend[i]);
    // peak concurrency per CU: events
    return 0;
}
