// A foreign load for the co-residency experiments: nothing but matrix instructions, in small workgroups that leave room on every CU
// for other kernels' waves.  usage: mfma_aggressor SECONDS [kind]   kind 0: v_mfma_f32_16x16x32_bf16 (gfx950), 1: v_mfma_f32_32x32x8_f16,
// 2: no matrix instruction (v_fma loop of the same length)
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));

template <int KIND>
__global__ __launch_bounds__(64) void k_spin(float *out, int iters) {
    const int lane = threadIdx.x;
    if (KIND == 0) {
        bf16x8 a, b;
        for (int i = 0; i < 8; ++i) a[i] = (__bf16)(0.001f * (lane + i)), b[i] = (__bf16)(0.002f * (lane - i));
        f32x4 c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0;
        for (int k = 0; k < iters; ++k) {
            c0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c0, 0, 0, 0);
            c1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c1, 0, 0, 0);
            c2 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c2, 0, 0, 0);
            c3 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c3, 0, 0, 0);
        }
        out[blockIdx.x * 64 + lane] = c0[0] + c1[1] + c2[2] + c3[3];
    } else if (KIND == 1) {
        f16x4 a, b;
        for (int i = 0; i < 4; ++i) a[i] = (_Float16)(0.001f * (lane + i)), b[i] = (_Float16)(0.002f * (lane - i));
        f32x16 c0 = {}, c1 = {};
        for (int k = 0; k < iters; ++k) {
            c0 = __builtin_amdgcn_mfma_f32_32x32x8f16(a, b, c0, 0, 0, 0);
            c1 = __builtin_amdgcn_mfma_f32_32x32x8f16(a, b, c1, 0, 0, 0);
        }
        out[blockIdx.x * 64 + lane] = c0[0] + c1[5];
    } else {
        float x = lane * 0.5f, y = 1.0001f, z = 0.25f;
        for (int k = 0; k < iters * 16; ++k) x = __builtin_fmaf(x, y, z);
        out[blockIdx.x * 64 + lane] = x;
    }
}

int main(int argc, char **argv) {
    const double secs = argc > 1 ? std::atof(argv[1]) : 10.0;
    const int kind = argc > 2 ? std::atoi(argv[2]) : 0;
    float *out = nullptr;
    (void)hipMalloc(&out, 4096 * 64 * sizeof(float));
    const auto t0 = std::chrono::steady_clock::now();
    long launches = 0;
    while (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() < secs) {
        for (int r = 0; r < 8; ++r) {
            if (kind == 0) hipLaunchKernelGGL(k_spin<0>, dim3(2048), dim3(64), 0, 0, out, 4000);
            else if (kind == 1) hipLaunchKernelGGL(k_spin<1>, dim3(2048), dim3(64), 0, 0, out, 2000);
            else hipLaunchKernelGGL(k_spin<2>, dim3(2048), dim3(64), 0, 0, out, 4000);
        }
        (void)hipDeviceSynchronize();
        launches += 8;
    }
    std::printf("aggressor kind %d: %ld launches in %.1f s\n", kind, launches, secs);
    return 0;
}
