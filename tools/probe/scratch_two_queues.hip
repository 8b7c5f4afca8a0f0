// Do two kernels that use private (scratch) memory corrupt each other when they run at the same time on two HIP streams?
// Every lane fills a 3 KB private array (dynamically indexed: it lives in scratch), does some global traffic in between so that the
// kernel lasts a while, reads the array back and counts what is not what it wrote.
//   hipcc -O2 --offload-arch=gfx950 -o scratch_two_queues tools/probe/scratch_two_queues.hip && ./scratch_two_queues [streams] [rounds]
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <thread>
#include <vector>

__global__ __launch_bounds__(512) void k_scratch(uint32_t *sink, uint32_t *errors, uint32_t salt, uint32_t spin) {
    uint32_t mine[776];  // 3104 bytes per lane, as k_aac_entropy_parse
    const uint32_t id = blockIdx.x * blockDim.x + threadIdx.x;
    for (uint32_t i = 0; i < 776; ++i) mine[(i * 7u + id) % 776u] = (id * 2654435761u) ^ (i * 40503u) ^ salt;
    uint32_t acc = 0;
    for (uint32_t k = 0; k < spin; ++k) {
        acc += sink[(id * 97u + k * 8191u) & 0xfffffu];
        acc ^= mine[(acc + k) % 776u];
    }
    uint32_t bad = 0;
    for (uint32_t i = 0; i < 776; ++i) bad += mine[(i * 7u + id) % 776u] != ((id * 2654435761u) ^ (i * 40503u) ^ salt);
    if (bad) atomicAdd(errors, bad);
    if (acc == 0x12345678u) sink[id & 0xfffffu] = acc;
}

int main(int argc, char **argv) {
    const int n_streams = argc > 1 ? std::atoi(argv[1]) : 2, rounds = argc > 2 ? std::atoi(argv[2]) : 200;
    (void)hipSetDevice(0);
    std::vector<uint32_t *> sink(n_streams), err(n_streams);
    std::vector<hipStream_t> st(n_streams);
    for (int s = 0; s < n_streams; ++s) {
        (void)hipMalloc(&sink[s], (1u << 20) * 4);
        (void)hipMemset(sink[s], 1, (1u << 20) * 4);
        (void)hipMalloc(&err[s], 4);
        (void)hipMemset(err[s], 0, 4);
        (void)hipStreamCreateWithFlags(&st[s], hipStreamNonBlocking);
    }
    std::vector<std::thread> th;
    for (int s = 0; s < n_streams; ++s)
        th.emplace_back([&, s] {
            (void)hipSetDevice(0);
            for (int r = 0; r < rounds; ++r) {
                hipLaunchKernelGGL(k_scratch, dim3(70 + 7 * s), dim3(512), 0, st[s], sink[s], err[s], (uint32_t)(r * 131 + s), 200u);
                if (r % 4 == 3) (void)hipStreamSynchronize(st[s]);
            }
            (void)hipStreamSynchronize(st[s]);
        });
    for (auto &t : th) t.join();
    for (int s = 0; s < n_streams; ++s) {
        uint32_t e = 0;
        (void)hipMemcpy(&e, err[s], 4, hipMemcpyDeviceToHost);
        std::printf("stream %d: %u private words read back wrong over %d launches\n", s, e, rounds);
    }
    return 0;
}
