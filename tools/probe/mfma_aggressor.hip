// A foreign load for the co-residency experiments: nothing but matrix instructions, in small workgroups that leave room on every CU
// for other kernels' waves.  usage: mfma_aggressor SECONDS [kind]   kind 0: v_mfma_f32_16x16x32_bf16 (gfx950), 1: v_mfma_f32_32x32x8_f16,
// 2: no matrix instruction (v_fma loop of the same length), 3: v_mfma_f32_16x16x32_f16 (gfx950), 4: v_mfma_f32_16x16x16_f16 (older parts'),
// 5: v_mfma_f32_32x32x16_f16 (gfx950), 6: v_mfma_f32_16x16x32_bf16 as ONE dependent chain, 7: v_mfma_i32_16x16x64_i8 (gfx950),
// 8: v_mfma_f32_32x32x16_bf16 (gfx950), 9: v_mfma_f32_16x16x16_bf16 (older parts' "_1k" form)
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef short s16x4 __attribute__((ext_vector_type(4)));

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
    } else if (KIND == 3 || KIND == 5) {
        f16x8 a, b;
        for (int i = 0; i < 8; ++i) a[i] = (_Float16)(0.001f * (lane + i)), b[i] = (_Float16)(0.002f * (lane - i));
        if (KIND == 3) {
            f32x4 c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0;
            for (int k = 0; k < iters; ++k) {
                c0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c0, 0, 0, 0);
                c1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c1, 0, 0, 0);
                c2 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c2, 0, 0, 0);
                c3 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c3, 0, 0, 0);
            }
            out[blockIdx.x * 64 + lane] = c0[0] + c1[1] + c2[2] + c3[3];
        } else {
            f32x16 c0 = {}, c1 = {};
            for (int k = 0; k < iters; ++k) {
                c0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c0, 0, 0, 0);
                c1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c1, 0, 0, 0);
            }
            out[blockIdx.x * 64 + lane] = c0[0] + c1[5];
        }
    } else if (KIND == 4) {
        f16x4 a, b;
        for (int i = 0; i < 4; ++i) a[i] = (_Float16)(0.001f * (lane + i)), b[i] = (_Float16)(0.002f * (lane - i));
        f32x4 c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0;
        for (int k = 0; k < iters; ++k) {
            c0 = __builtin_amdgcn_mfma_f32_16x16x16f16(a, b, c0, 0, 0, 0);
            c1 = __builtin_amdgcn_mfma_f32_16x16x16f16(a, b, c1, 0, 0, 0);
            c2 = __builtin_amdgcn_mfma_f32_16x16x16f16(a, b, c2, 0, 0, 0);
            c3 = __builtin_amdgcn_mfma_f32_16x16x16f16(a, b, c3, 0, 0, 0);
        }
        out[blockIdx.x * 64 + lane] = c0[0] + c1[1] + c2[2] + c3[3];
    } else if (KIND == 6) {
        bf16x8 a, b;
        for (int i = 0; i < 8; ++i) a[i] = (__bf16)(0.001f * (lane + i)), b[i] = (__bf16)(0.002f * (lane - i));
        f32x4 c0 = {0, 0, 0, 0};
        for (int k = 0; k < iters * 2; ++k) c0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c0, 0, 0, 0);
        out[blockIdx.x * 64 + lane] = c0[0];
    } else if (KIND == 8) {
        bf16x8 a, b;
        for (int i = 0; i < 8; ++i) a[i] = (__bf16)(0.001f * (lane + i)), b[i] = (__bf16)(0.002f * (lane - i));
        f32x16 c0 = {}, c1 = {};
        for (int k = 0; k < iters; ++k) {
            c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c0, 0, 0, 0);
            c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c1, 0, 0, 0);
        }
        out[blockIdx.x * 64 + lane] = c0[0] + c1[5];
    } else if (KIND == 9) {
        s16x4 a, b;
        for (int i = 0; i < 4; ++i) a[i] = (short)(0x3f80 + lane + i), b[i] = (short)(0x3f00 + lane - i);
        f32x4 c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0;
        for (int k = 0; k < iters; ++k) {
            c0 = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(a, b, c0, 0, 0, 0);
            c1 = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(a, b, c1, 0, 0, 0);
            c2 = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(a, b, c2, 0, 0, 0);
            c3 = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(a, b, c3, 0, 0, 0);
        }
        out[blockIdx.x * 64 + lane] = c0[0] + c1[1] + c2[2] + c3[3];
    } else if (KIND == 7) {
        i32x4 a, b;
        for (int i = 0; i < 4; ++i) a[i] = 0x01020304 * (lane + i + 1), b[i] = 0x04030201 * (lane + 7 - i);
        i32x4 c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0;
        for (int k = 0; k < iters; ++k) {
            c0 = __builtin_amdgcn_mfma_i32_16x16x64_i8(a, b, c0, 0, 0, 0);
            c1 = __builtin_amdgcn_mfma_i32_16x16x64_i8(a, b, c1, 0, 0, 0);
            c2 = __builtin_amdgcn_mfma_i32_16x16x64_i8(a, b, c2, 0, 0, 0);
            c3 = __builtin_amdgcn_mfma_i32_16x16x64_i8(a, b, c3, 0, 0, 0);
        }
        out[blockIdx.x * 64 + lane] = (float)(c0[0] + c1[1] + c2[2] + c3[3]);
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
            else if (kind == 3) hipLaunchKernelGGL(k_spin<3>, dim3(2048), dim3(64), 0, 0, out, 4000);
            else if (kind == 4) hipLaunchKernelGGL(k_spin<4>, dim3(2048), dim3(64), 0, 0, out, 4000);
            else if (kind == 5) hipLaunchKernelGGL(k_spin<5>, dim3(2048), dim3(64), 0, 0, out, 2000);
            else if (kind == 6) hipLaunchKernelGGL(k_spin<6>, dim3(2048), dim3(64), 0, 0, out, 4000);
            else if (kind == 7) hipLaunchKernelGGL(k_spin<7>, dim3(2048), dim3(64), 0, 0, out, 4000);
            else if (kind == 8) hipLaunchKernelGGL(k_spin<8>, dim3(2048), dim3(64), 0, 0, out, 2000);
            else if (kind == 9) hipLaunchKernelGGL(k_spin<9>, dim3(2048), dim3(64), 0, 0, out, 4000);
            else hipLaunchKernelGGL(k_spin<2>, dim3(2048), dim3(64), 0, 0, out, 4000);
        }
        (void)hipDeviceSynchronize();
        launches += 8;
    }
    std::printf("aggressor kind %d: %ld launches in %.1f s\n", kind, launches, secs);
    return 0;
}
