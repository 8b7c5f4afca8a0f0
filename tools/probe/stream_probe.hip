// stream_probe.hip -- what shape of kernel streams fastest on this box?  y[i] = x[i] + 1 over 2.18 GB each way.
//   hipcc --offload-arch=gfx950 -O3 -o stream_probe stream_probe.hip && ./stream_probe
// Variants: tile-per-block (U float4 loads in flight per thread, no loop) with block sizes 256/512/1024, and a
// grid-stride loop sized to fill the chip k waves per SIMD deep.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <vector>

typedef float f4 __attribute__((ext_vector_type(4)));

template <int U>
__global__ void k_tile(const f4 *__restrict__ x, f4 *__restrict__ y, size_t n4) {
    const size_t base = (size_t)blockIdx.x * blockDim.x * U + threadIdx.x;
    f4 v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) v[u] = base + (size_t)u * blockDim.x < n4 ? x[base + (size_t)u * blockDim.x] : (f4){0, 0, 0, 0};
#pragma unroll
    for (int u = 0; u < U; ++u)
        if (base + (size_t)u * blockDim.x < n4) y[base + (size_t)u * blockDim.x] = v[u] + 1.0f;
}

template <int U>
__global__ void k_stride(const f4 *__restrict__ x, f4 *__restrict__ y, size_t n4) {
    const size_t step = (size_t)gridDim.x * blockDim.x * U;
    for (size_t base = (size_t)blockIdx.x * blockDim.x * U + threadIdx.x; base < n4; base += step) {
        f4 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) v[u] = base + (size_t)u * blockDim.x < n4 ? x[base + (size_t)u * blockDim.x] : (f4){0, 0, 0, 0};
#pragma unroll
        for (int u = 0; u < U; ++u)
            if (base + (size_t)u * blockDim.x < n4) y[base + (size_t)u * blockDim.x] = v[u] + 1.0f;
    }
}

// one wave walks a contiguous span of 4 KiB pieces in order (the synthesis kernel's shape): D pieces in flight
template <int D>
__global__ void k_walk(const f4 *__restrict__ x, f4 *__restrict__ y, size_t pieces_per_wave, size_t n_pieces, size_t piece_stride) {
    const size_t wave = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int lane = threadIdx.x & 63;
    // piece p of wave w sits at (p * n_waves + w): frame-major batch, every wave reads its own 4 KiB of each "frame"
    const size_t n_waves = ((size_t)gridDim.x * blockDim.x) >> 6;
    f4 v[D][4];
    for (size_t p0 = 0; p0 < pieces_per_wave; p0 += D) {
#pragma unroll
        for (int d = 0; d < D; ++d) {
            const size_t piece = (p0 + d) * n_waves + wave;
#pragma unroll
            for (int r = 0; r < 4; ++r) v[d][r] = x[piece * 256 + 64 * r + lane];
        }
#pragma unroll
        for (int d = 0; d < D; ++d) {
            const size_t piece = (p0 + d) * n_waves + wave;
#pragma unroll
            for (int r = 0; r < 4; ++r) y[piece * 256 + 64 * r + lane] = v[d][r] + 1.0f;
        }
    }
}

// the same walk with pieces of L KiB (L 16-byte loads per lane in flight)
template <int L>
__global__ void k_walk_small(const f4 *__restrict__ x, f4 *__restrict__ y, size_t pieces_per_wave) {
    const size_t wave = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int lane = threadIdx.x & 63;
    const size_t n_waves = ((size_t)gridDim.x * blockDim.x) >> 6;
    for (size_t p = 0; p < pieces_per_wave; ++p) {
        const size_t piece = p * n_waves + wave;
        f4 v[L];
#pragma unroll
        for (int r = 0; r < L; ++r) v[r] = x[piece * (64 * L) + 64 * r + lane];
#pragma unroll
        for (int r = 0; r < L; ++r) y[piece * (64 * L) + 64 * r + lane] = v[r] + 1.0f;
    }
}

// The shape proposed for the synthesis kernel: a workgroup = FR consecutive 4 KiB pieces ("frames") of one channel, one
// wave per piece, plus one wave that re-reads the piece before the group (the overlap it needs); every wave hands 4 KiB to
// its neighbour through LDS behind one barrier and stores its own piece -- single touch, blocks in address order.
// WORK: dependent FMAs per loaded value (stands in for the FFT); OUT_HALF: store 2 KiB instead of 4 (s16 output).
typedef float f2v __attribute__((ext_vector_type(2)));
template <int FR, int WORK, bool OUT_HALF>
__global__ __launch_bounds__((FR + 1) * 64) void k_group(const float *__restrict__ x, float *__restrict__ y, size_t n_pieces) {
    __shared__ f4 hand[FR + 1][256];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const size_t first = (size_t)blockIdx.x * FR;
    const long piece = (long)first + wave - 1;  // wave 0 reads the piece before the group
    f2v v[8];
    if (piece >= 0 && (size_t)piece < n_pieces) {
#pragma unroll
        for (int r = 0; r < 8; ++r) v[r] = *reinterpret_cast<const f2v *>(x + (size_t)piece * 1024 + 2 * lane + 128 * r);
    } else {
#pragma unroll
        for (int r = 0; r < 8; ++r) v[r] = (f2v){0.f, 0.f};
    }
#pragma unroll
    for (int k = 0; k < WORK; ++k)
#pragma unroll
        for (int r = 0; r < 8; ++r) v[r] = v[r] * 1.0001f + v[(r + 1) & 7];
#pragma unroll
    for (int r = 0; r < 4; ++r) hand[wave][64 * r + lane] = (f4){v[2 * r].x, v[2 * r].y, v[2 * r + 1].x, v[2 * r + 1].y};
    __syncthreads();
    if (wave == 0 || piece < 0 || (size_t)piece >= n_pieces) return;  // nothing is stored outside the arrays
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const f4 prev = hand[wave - 1][64 * r + lane];
        const f4 out = (f4){v[2 * r].x, v[2 * r].y, v[2 * r + 1].x, v[2 * r + 1].y} + prev;
        if (OUT_HALF) {
            if (r < 2) reinterpret_cast<f4 *>(y + (size_t)piece * 512)[64 * r + lane] = out;
        } else {
            reinterpret_cast<f4 *>(y + (size_t)piece * 1024)[64 * r + lane] = out;
        }
    }
}

#define CHECK(e) do { hipError_t err_ = (e); if (err_ != hipSuccess) { std::printf("%s: %s\n", #e, hipGetErrorString(err_)); return 1; } } while (0)

int main() {
    const size_t n = (size_t)4096 * 64 * 2 * 1024, n4 = n / 4;
    f4 *x, *y;
    CHECK(hipMalloc(&x, n * 4));
    CHECK(hipMalloc(&y, n * 4));
    CHECK(hipMemset(x, 0, n * 4));
    hipEvent_t a, b;
    CHECK(hipEventCreate(&a));
    CHECK(hipEventCreate(&b));
    auto time = [&](const char *name, auto launch) {
        for (int i = 0; i < 3; ++i) launch();
        hipEventRecord(a);
        for (int i = 0; i < 20; ++i) launch();
        hipEventRecord(b);
        hipEventSynchronize(b);
        float ms;
        hipEventElapsedTime(&ms, a, b);
        ms /= 20;
        std::printf("%-44s %.3f ms  %.2f TB/s\n", name, ms, 2.0 * n * 4 / ms / 1e9);
    };
    time("tile, 256 threads x 4 loads", [&] { hipLaunchKernelGGL(k_tile<4>, dim3((n4 + 1023) / 1024), dim3(256), 0, 0, x, y, n4); });
    time("tile, 256 threads x 8 loads", [&] { hipLaunchKernelGGL(k_tile<8>, dim3((n4 + 2047) / 2048), dim3(256), 0, 0, x, y, n4); });
    time("tile, 256 threads x 1 load", [&] { hipLaunchKernelGGL(k_tile<1>, dim3((n4 + 255) / 256), dim3(256), 0, 0, x, y, n4); });
    time("tile, 1024 threads x 4 loads", [&] { hipLaunchKernelGGL(k_tile<4>, dim3((n4 + 4095) / 4096), dim3(1024), 0, 0, x, y, n4); });
    time("grid-stride, 2048 blocks x 256 x 4 loads", [&] { hipLaunchKernelGGL(k_stride<4>, dim3(2048), dim3(256), 0, 0, x, y, n4); });
    time("grid-stride, 8192 blocks x 256 x 4 loads", [&] { hipLaunchKernelGGL(k_stride<4>, dim3(8192), dim3(256), 0, 0, x, y, n4); });
    time("grid-stride, 1024 blocks x 256 x 1 load", [&] { hipLaunchKernelGGL(k_stride<1>, dim3(1024), dim3(256), 0, 0, x, y, n4); });
    const size_t pieces = n4 / 256;
    time("walk, 8192 waves (4/block), 1 piece in flight", [&] { hipLaunchKernelGGL(k_walk<1>, dim3(2048), dim3(256), 0, 0, x, y, pieces / 8192, pieces, 0); });
    time("walk, 8192 waves, 2 pieces in flight", [&] { hipLaunchKernelGGL(k_walk<2>, dim3(2048), dim3(256), 0, 0, x, y, pieces / 8192, pieces, 0); });
    time("walk, 8192 waves, 4 pieces in flight", [&] { hipLaunchKernelGGL(k_walk<4>, dim3(2048), dim3(256), 0, 0, x, y, pieces / 8192, pieces, 0); });
    time("walk, 16384 waves, 1 piece in flight", [&] { hipLaunchKernelGGL(k_walk<1>, dim3(4096), dim3(256), 0, 0, x, y, pieces / 16384, pieces, 0); });
    time("walk, 4096 waves, 2 pieces in flight", [&] { hipLaunchKernelGGL(k_walk<2>, dim3(1024), dim3(256), 0, 0, x, y, pieces / 4096, pieces, 0); });
    {
        const size_t np = n / 1024;
        auto rate = [&](const char *name, double bytes, auto launch) {
            for (int i = 0; i < 3; ++i) launch();
            hipEventRecord(a);
            for (int i = 0; i < 20; ++i) launch();
            hipEventRecord(b);
            hipEventSynchronize(b);
            float ms;
            hipEventElapsedTime(&ms, a, b);
            ms /= 20;
            std::printf("%-60s %.3f ms  %.2f TB/s useful (%.2f incl. re-read)\n", name, ms, bytes / ms / 1e9, (bytes + n * 4.0 / 8) / ms / 1e9);
        };
        rate("group: 8 frames + 1 / block, no work, 4 KiB out", 2.0 * n * 4, [&] { hipLaunchKernelGGL((k_group<8, 0, false>), dim3(np / 8), dim3(576), 0, 0, (const float *)x, (float *)y, np); });
        rate("group: 8 frames + 1 / block, 40 FMA rounds, 4 KiB out", 2.0 * n * 4, [&] { hipLaunchKernelGGL((k_group<8, 40, false>), dim3(np / 8), dim3(576), 0, 0, (const float *)x, (float *)y, np); });
        rate("group: 8 frames + 1 / block, 40 FMA rounds, 2 KiB out", 1.5 * n * 4, [&] { hipLaunchKernelGGL((k_group<8, 40, true>), dim3(np / 8), dim3(576), 0, 0, (const float *)x, (float *)y, np); });
        rate("group: 15 frames + 1 / block, 40 FMA rounds, 4 KiB out", 2.0 * n * 4, [&] { hipLaunchKernelGGL((k_group<15, 40, false>), dim3((np + 14) / 15), dim3(1024), 0, 0, (const float *)x, (float *)y, np); });
        rate("group: 4 frames + 1 / block, 40 FMA rounds, 4 KiB out", 2.0 * n * 4, [&] { hipLaunchKernelGGL((k_group<4, 40, false>), dim3(np / 4), dim3(320), 0, 0, (const float *)x, (float *)y, np); });
        rate("group: 8 frames + 1 / block, 80 FMA rounds, 4 KiB out", 2.0 * n * 4, [&] { hipLaunchKernelGGL((k_group<8, 80, false>), dim3(np / 8), dim3(576), 0, 0, (const float *)x, (float *)y, np); });
    }
    for (int waves : {1024, 8192}) {
        char name[96];
        std::snprintf(name, sizeof name, "walk, %d waves, 4 KiB pieces", waves);
        time(name, [&] { hipLaunchKernelGGL(k_walk_small<4>, dim3(waves / 4), dim3(256), 0, 0, x, y, n4 / 256 / waves); });
        std::snprintf(name, sizeof name, "walk, %d waves, 1 KiB pieces", waves);
        time(name, [&] { hipLaunchKernelGGL(k_walk_small<1>, dim3(waves / 4), dim3(256), 0, 0, x, y, n4 / 64 / waves); });
        std::snprintf(name, sizeof name, "walk, %d waves, 2 KiB pieces", waves);
        time(name, [&] { hipLaunchKernelGGL(k_walk_small<2>, dim3(waves / 4), dim3(256), 0, 0, x, y, n4 / 128 / waves); });
    }
    return 0;
}
