// Probe: host-to-device copy time of one 1920x1080 u8 frame, by source memory kind and copy call (HIP, gfx950 box).
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main()
{
    const size_t W = 1920, H = 1080, N = W * H;
    unsigned char *d, *pin, *pin_nc, *page = (unsigned char *)malloc(N);
    hipMalloc((void **)&d, N);
    hipHostMalloc((void **)&pin, N, hipHostMallocDefault);
    hipHostMalloc((void **)&pin_nc, N, hipHostMallocNonCoherent);
    memset(page, 1, N); memset(pin, 2, N); memset(pin_nc, 3, N);
    hipStream_t s;
    hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
    struct { const char *name; unsigned char *src; int two_d; } cases[] = {
        {"pageable 1D", page, 0}, {"pageable 2D", page, 1}, {"pinned(default) 1D", pin, 0}, {"pinned(default) 2D", pin, 1},
        {"pinned(noncoherent) 1D", pin_nc, 0}, {"pinned(noncoherent) 2D", pin_nc, 1}};
    for (auto &c : cases) {
        for (int rep = 0; rep < 3; rep++) {
            double t = now();
            for (int i = 0; i < 50; i++) {
                if (c.two_d) hipMemcpy2DAsync(d, W, c.src, W, W, H, hipMemcpyHostToDevice, s);
                else hipMemcpyAsync(d, c.src, N, hipMemcpyHostToDevice, s);
            }
            hipStreamSynchronize(s);
            double dt = (now() - t) / 50;
            if (rep == 2) printf("%-26s %.1f us per frame  %.1f GB/s\n", c.name, dt * 1e6, N / dt / 1e9);
        }
    }
    return 0;
}
