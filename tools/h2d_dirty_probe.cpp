// Does a host-to-device copy run slower when the CPU has JUST written the pinned source buffer (dirty lines in the cores' caches that the
// DMA's reads must snoop) than from memory that was written long ago -- and do non-temporal stores avoid it?
// Build on the GPU box: hipcc -O2 -mavx2 tools/h2d_dirty_probe.cpp -o /tmp/h2d_dirty_probe   (prints one JSON line)
#include <hip/hip_runtime.h>
#include <immintrin.h>
#include <chrono>
#include <cstdio>
#include <cstring>
#include <vector>

static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

static void nt_copy(void *dst, const void *src, size_t n)
{
    auto *d = (__m256i *)dst;
    auto *s = (const __m256i *)src;
    for (size_t i = 0; i < n / 32; i++) _mm256_stream_si256(d + i, _mm256_loadu_si256(s + i));
    _mm_sfence();
}

int main()
{
    const size_t sizes[2] = {1920 * 1080, 3840 * 2160};
    printf("{");
    for (int si = 0; si < 2; si++) {
        const size_t n = sizes[si];
        void *pin = nullptr, *dev = nullptr;
        hipHostMalloc(&pin, n, hipHostMallocDefault);
        hipMalloc(&dev, n);
        std::vector<unsigned char> src(n, 7), evict(256u << 20, 1);
        hipStream_t st;
        hipStreamCreateWithFlags(&st, hipStreamNonBlocking);
        memset(pin, 3, n);
        double t[3] = {0, 0, 0};
        const int reps = 40;
        for (int mode = 0; mode < 3; mode++) {
            for (int r = 0; r < reps + 2; r++) {
                if (mode == 0) { volatile unsigned long long sum = 0; for (size_t i = 0; i < evict.size(); i += 64) sum += evict[i]; }   // source long out of the caches
                if (mode == 1) memcpy(pin, src.data(), n);
                if (mode == 2) nt_copy(pin, src.data(), n);
                const double a = now();
                hipMemcpyAsync(dev, pin, n, hipMemcpyHostToDevice, st);
                hipStreamSynchronize(st);
                if (r >= 2) t[mode] += now() - a;
            }
        }
        printf("%s\"h2d_%zu_bytes\": {\"source_cold_GBps\": %.2f, \"source_just_written_memcpy_GBps\": %.2f, \"source_just_written_nontemporal_GBps\": %.2f}",
               si ? ", " : "", n, n / (t[0] / reps) / 1e9, n / (t[1] / reps) / 1e9, n / (t[2] / reps) / 1e9);
        hipFree(dev);
        hipHostFree(pin);
    }
    printf("}\n");
    return 0;
}
