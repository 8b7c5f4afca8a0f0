// How long hipHostMalloc / hipHostFree take on the GPU box, by size and flag (the scheduler's output buffers are pinned).
//   hipcc -O2 -o /tmp/pinned_probe tools/probe/pinned_alloc_probe.cpp && /tmp/pinned_probe
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdio>

int main() {
    using Clock = std::chrono::steady_clock;
    auto ms = [](Clock::time_point a) { return std::chrono::duration<double, std::milli>(Clock::now() - a).count(); };
    (void)hipSetDevice(0);
    void *warm = nullptr;
    (void)hipMalloc(&warm, 1 << 20);
    const struct {
        const char *name;
        unsigned flag;
    } flags[] = {{"default", hipHostMallocDefault}, {"portable", hipHostMallocPortable}, {"numa-user", hipHostMallocNumaUser}};
    for (const auto &f : flags)
        for (size_t mb : {4, 16, 64, 256}) {
            for (int rep = 0; rep < 2; ++rep) {
                void *p = nullptr;
                const Clock::time_point t0 = Clock::now();
                const hipError_t rc = hipHostMalloc(&p, mb << 20, f.flag);
                const double t_alloc = ms(t0);
                if (rc != hipSuccess) {
                    std::printf("%-10s %4zu MB: %s\n", f.name, mb, hipGetErrorString(rc));
                    continue;
                }
                const Clock::time_point t1 = Clock::now();
                (void)hipHostFree(p);
                std::printf("%-10s %4zu MB: alloc %8.2f ms  free %8.2f ms\n", f.name, mb, t_alloc, ms(t1));
            }
        }
    return 0;
}
