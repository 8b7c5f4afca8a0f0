// Which instructions of a co-resident kernel are disturbed by another queue's v_mfma_f32_16x16x32_bf16 loop?
// One process, two streams: a host thread keeps the matrix-instruction kernel of mfma_aggressor.hip in flight on one stream while
// small single-purpose kernels run on the other; each is compared, bit for bit, with its own result from before the aggressor started.
// usage: mfma_victim [aggressor kind: 0 bf16 16x16x32 | 1 f16 32x32x8 | 2 v_fma | 3 f16 16x16x32 | 9 none in this process (run mfma_aggressor beside it)] [runs] [iterations inside a victim]
#include <hip/hip_runtime.h>

#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));

#define CK(x)                                                                                   \
    do {                                                                                        \
        hipError_t e_ = (x);                                                                    \
        if (e_ != hipSuccess) {                                                                 \
            std::fprintf(stderr, "%s: %s (line %d)\n", #x, hipGetErrorString(e_), __LINE__);   \
            std::exit(2);                                                                       \
        }                                                                                       \
    } while (0)

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
    } else if (KIND == 3) {
        f16x8 a, b;
        for (int i = 0; i < 8; ++i) a[i] = (_Float16)(0.001f * (lane + i)), b[i] = (_Float16)(0.002f * (lane - i));
        f32x4 c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0;
        for (int k = 0; k < iters; ++k) {
            c0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c0, 0, 0, 0);
            c1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c1, 0, 0, 0);
            c2 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c2, 0, 0, 0);
            c3 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c3, 0, 0, 0);
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

// ---- victims: 256 threads, each writes 4 floats ----
__global__ void k_fill(float *p, size_t n) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        p[i] = (float)((i * 2654435761u >> 8) & 0xffff) * (1.0f / 4096.0f) - 8.0f;
}

enum { V_FMA = 0, V_PK_FMA, V_PK_MUL_ADD, V_LDS, V_BPERMUTE, V_GLOBAL, V_PK_ILP, V_PK_OPSEL, V_PK_OPSEL_ILP, V_LDS_READ2, V_FFT_LIKE, V_SGPR_HOLD, V_VGPR_HOLD, V_LDS_HOLD, V_SLOAD, V_LOAD_PIPE, V_LOAD_PIPE_MATH, V_COUNT };
static const char *const kVictimName[V_COUNT] = {"v_fma_f32", "v_pk_fma_f32", "v_pk_mul_f32+v_pk_add_f32", "ds_write_b128/ds_read_b128",
                                                 "ds_bpermute_b32", "global_load/store dwordx4", "v_pk_fma_f32, 8 chains", "pk complex rotate (op_sel, neg)",
                                                 "pk complex rotate, 8 chains", "ds_read2/write2 b64, wave-private", "rotate + LDS exchange (FFT-like)",
                                                 "96 SGPRs held across a spin", "200 VGPRs held across a spin", "64 KB of LDS held across a spin", "s_load in a loop", "prefetched global loads, partial vmcnt waits", "prefetched loads + rotate + LDS exchange"};

__device__ float *big_in, *big_out;  // 2048 blocks x 4 waves x 32 rows x 1024 floats each (1 GiB each), set by main

template <int V>
__global__ __launch_bounds__(256) void k_victim(f32x4 *out, f32x4 *scratch, int iters) {
    __shared__ f32x4 lds[256];
    __shared__ f32x2 lds2[4 * 1024];
    const int t = threadIdx.x;
    const unsigned g = blockIdx.x * 256 + t;
    f32x4 x = {1.0f + (g & 1023) * 0.001f, 0.5f + (g & 255) * 0.002f, 0.25f + (g & 63) * 0.003f, 2.0f - (g & 511) * 0.001f};
    const f32x2 a = {0.99951171875f, 0.9990234375f}, b = {0.0009765625f, 0.001953125f};
    if (V == V_FMA) {
        for (int k = 0; k < iters; ++k) {
            asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x[0]) : "v"(a[0]), "v"(b[0]));
            asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x[1]) : "v"(a[1]), "v"(b[1]));
            asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x[2]) : "v"(a[0]), "v"(b[1]));
            asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x[3]) : "v"(a[1]), "v"(b[0]));
        }
    } else if (V == V_PK_FMA) {
        f32x2 p = {x[0], x[1]}, q = {x[2], x[3]};
        for (int k = 0; k < iters; ++k) {
            asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p) : "v"(a), "v"(b));
            asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(q) : "v"(a), "v"(b));
        }
        x = f32x4{p[0], p[1], q[0], q[1]};
    } else if (V == V_PK_MUL_ADD) {
        f32x2 p = {x[0], x[1]}, q = {x[2], x[3]};
        for (int k = 0; k < iters; ++k) {
            asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p) : "v"(a));
            asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p) : "v"(b));
            asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(q) : "v"(a));
            asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(q) : "v"(b));
        }
        x = f32x4{p[0], p[1], q[0], q[1]};
    } else if (V == V_LDS) {
        for (int k = 0; k < iters / 8; ++k) {
            lds[t] = x;
            __syncthreads();
            const f32x4 y = lds[(t * 37 + k) & 255];
            __syncthreads();
            x = f32x4{y[1], y[2], y[3], y[0]};
        }
    } else if (V == V_BPERMUTE) {
        for (int k = 0; k < iters / 4; ++k) {
            const int src = ((t * 5 + k) & 63) << 2;
            for (int c = 0; c < 4; ++c) x[c] = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(src, __builtin_bit_cast(int, x[c])));
            x = f32x4{x[3], x[0], x[1], x[2]};
        }
    } else if (V == V_GLOBAL) {
        f32x4 *mine = scratch + (size_t)blockIdx.x * 256;
        for (int k = 0; k < iters / 32; ++k) {
            __builtin_nontemporal_store(x, &mine[t]);
            __syncthreads();
            const f32x4 y = mine[(t * 29 + k) & 255];
            __syncthreads();
            x = f32x4{y[2], y[3], y[0], y[1]};
        }
    } else if (V == V_PK_ILP) {
        f32x2 p[8];
        for (int i = 0; i < 8; ++i) p[i] = f32x2{x[i & 3] + 0.01f * i, x[(i + 1) & 3] - 0.01f * i};
        for (int k = 0; k < iters / 2; ++k) {
#pragma unroll
            for (int i = 0; i < 8; ++i) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p[i]) : "v"(a), "v"(b));
        }
        x = f32x4{p[0][0] + p[1][1], p[2][0] + p[3][1], p[4][0] + p[5][1], p[6][0] + p[7][1]};
    } else if (V == V_PK_OPSEL) {
        const f32x2 w = {0.8f, 0.6f};  // a rotation: (x, y) -> (x c - y s, x s + y c)
        f32x2 p = {x[0], x[1]}, q = {x[2], x[3]}, tp, tq;
        for (int k = 0; k < iters; ++k) {
            asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel:[1,1] op_sel_hi:[1,0]" : "=&v"(tp) : "v"(p), "v"(w));
            asm volatile("v_pk_fma_f32 %0, %0, %1, %2 op_sel:[0,0,0] op_sel_hi:[0,1,1] neg_lo:[0,0,1]" : "+v"(p) : "v"(w), "v"(tp));
            asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel:[1,1] op_sel_hi:[1,0]" : "=&v"(tq) : "v"(q), "v"(w));
            asm volatile("v_pk_fma_f32 %0, %0, %1, %2 op_sel:[0,0,0] op_sel_hi:[0,1,1] neg_lo:[0,0,1]" : "+v"(q) : "v"(w), "v"(tq));
        }
        x = f32x4{p[0], p[1], q[0], q[1]};
    } else if (V == V_PK_OPSEL_ILP) {
        const f32x2 w = {0.8f, 0.6f};
        f32x2 p[8], tt[8];
        for (int i = 0; i < 8; ++i) p[i] = f32x2{x[i & 3] + 0.01f * i, x[(i + 1) & 3] - 0.01f * i};
        for (int k = 0; k < iters / 2; ++k) {
#pragma unroll
            for (int i = 0; i < 8; ++i) asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel:[1,1] op_sel_hi:[1,0]" : "=&v"(tt[i]) : "v"(p[i]), "v"(w));
#pragma unroll
            for (int i = 0; i < 8; ++i)
                asm volatile("v_pk_fma_f32 %0, %0, %1, %2 op_sel:[0,0,0] op_sel_hi:[0,1,1] neg_lo:[0,0,1]" : "+v"(p[i]) : "v"(w), "v"(tt[i]));
        }
        x = f32x4{p[0][0] + p[1][1], p[2][0] + p[3][1], p[4][0] + p[5][1], p[6][0] + p[7][1]};
    } else if (V == V_LDS_READ2) {
        // a wave's private 64 x 4 f32x2 exchange without a workgroup barrier, the way the synthesis kernels use LDS
        f32x2 *mine = lds2 + (t >> 6) * 1024;
        const int lane = t & 63;
        f32x2 p[4] = {f32x2{x[0], x[1]}, f32x2{x[2], x[3]}, f32x2{x[1], x[2]}, f32x2{x[3], x[0]}};
        for (int k = 0; k < iters / 8; ++k) {
#pragma unroll
            for (int i = 0; i < 4; ++i) mine[lane + 64 * i] = p[i];
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
            const int src = (lane * 4 + (k & 3)) & 255;
#pragma unroll
            for (int i = 0; i < 4; ++i) p[i] = mine[(src + i * 67) & 255];
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
        }
        x = f32x4{p[0][0] + p[1][1], p[2][0] + p[3][1], p[0][1], p[3][0]};
    } else if (V == V_FFT_LIKE) {
        f32x2 *mine = lds2 + (t >> 6) * 1024;
        const int lane = t & 63;
        const f32x2 w = {0.8f, 0.6f};
        f32x2 p[8], tt[8];
        for (int i = 0; i < 8; ++i) p[i] = f32x2{x[i & 3] + 0.01f * i, x[(i + 1) & 3] - 0.01f * i};
        for (int k = 0; k < iters / 16; ++k) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
#pragma unroll
                for (int i = 0; i < 8; ++i) asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel:[1,1] op_sel_hi:[1,0]" : "=&v"(tt[i]) : "v"(p[i]), "v"(w));
#pragma unroll
                for (int i = 0; i < 8; ++i)
                    asm volatile("v_pk_fma_f32 %0, %0, %1, %2 op_sel:[0,0,0] op_sel_hi:[0,1,1] neg_lo:[0,0,1]" : "+v"(p[i]) : "v"(w), "v"(tt[i]));
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const f32x2 s0 = p[i] + p[i + 4], s1 = p[i] - p[i + 4];
                    p[i] = s0 * 0.70703125f;
                    p[i + 4] = s1 * 0.70703125f;
                }
            }
#pragma unroll
            for (int i = 0; i < 8; ++i) mine[lane + 64 * i] = p[i];
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int i = 0; i < 8; ++i) p[i] = mine[((lane * 8 + i) * 9 + k) & 511];
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
        }
        x = f32x4{p[0][0] + p[1][1], p[2][0] + p[3][1], p[4][0] + p[5][1], p[6][0] + p[7][1]};
    } else if (V == V_SGPR_HOLD) {
        int sg[96];
#pragma unroll
        for (int i = 0; i < 96; ++i) asm volatile("s_mul_i32 %0, %1, %2" : "=s"(sg[i]) : "s"((int)blockIdx.x + 3), "s"(i * 2654435 + 17));
        for (int k = 0; k < iters; ++k) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x[0]) : "v"(a[0]), "v"(b[0]));
        int acc = 0;
#pragma unroll
        for (int i = 0; i < 96; ++i) asm volatile("s_xor_b32 %0, %0, %1\n\ts_lshl_b32 %0, %0, 1" : "+s"(acc) : "s"(sg[i]));
        x[1] = __builtin_bit_cast(float, acc & 0x3fffffff);
    } else if (V == V_VGPR_HOLD) {
        int vg[200];
#pragma unroll
        for (int i = 0; i < 200; ++i) asm volatile("v_mul_u32_u24 %0, %1, %2" : "=v"(vg[i]) : "v"((int)(g & 0xffff) + 3), "v"(i * 40503 + 17));
        for (int k = 0; k < iters; ++k) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x[0]) : "v"(a[0]), "v"(b[0]));
        int acc = 0;
#pragma unroll
        for (int i = 0; i < 200; ++i) asm volatile("v_xor_b32 %0, %0, %1\n\tv_lshlrev_b32 %0, 1, %0" : "+v"(acc) : "v"(vg[i]));
        x[1] = __builtin_bit_cast(float, acc & 0x3fffffff);
    } else if (V == V_LDS_HOLD) {
        for (int i = t; i < 4096; i += 256) lds2[i] = f32x2{(float)(i * 3 + (int)(blockIdx.x & 255)), (float)(i ^ 0x155)};
        __syncthreads();
        for (int k = 0; k < iters; ++k) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x[0]) : "v"(a[0]), "v"(b[0]));
        float sum = 0.f;
        for (int i = t; i < 4096; i += 256) sum += lds2[i][0] - lds2[i][1];
        x[1] = sum;
    } else if (V == V_SLOAD) {
        // wave-uniform loads through the constant address space (s_load), as the synthesis kernels read their schedule
        typedef const __attribute__((address_space(4))) int *cptr;
        cptr tab = reinterpret_cast<cptr>(reinterpret_cast<uintptr_t>(scratch));  // host fills scratch with a pattern before the launches
        int acc = 0;
        for (int k = 0; k < iters / 4; ++k) {
            const int idx = __builtin_amdgcn_readfirstlane((int)((blockIdx.x * 131 + k * 7 + (t >> 6) * 977) & 0xffff));
            const int v = tab[idx];
            acc = acc * 33 + v;
            asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x[0]) : "v"(a[0]), "v"(b[0]));
        }
        x[1] = __builtin_bit_cast(float, acc & 0x3fffffff);
    } else if (V == V_LOAD_PIPE || V == V_LOAD_PIPE_MATH) {
        // the synthesis kernels' shape: a wave walks rows of 1024 floats, the next row's eight loads are issued before the current row is
        // used (so the waits in front of the uses are partial: vmcnt(N) with N > 0), the registers are reused row after row
        const float *big = reinterpret_cast<const float *>(big_in);
        const int lane = t & 63;
        const size_t wave_id = (size_t)blockIdx.x * 4 + (t >> 6);
        const int rows = 32;
        const float *src = big + wave_id * rows * 1024 + 2 * lane;
        f32x2 *mine = lds2 + (t >> 6) * 1024;
        const f32x2 w = {0.8f, 0.6f};
        f32x2 cur[8], nxt[8], acc[8];
#pragma unroll
        for (int r = 0; r < 8; ++r) cur[r] = *reinterpret_cast<const f32x2 *>(src + 128 * r), acc[r] = f32x2{0.f, 0.f};
        for (int e = 0; e < rows; ++e) {
            const float *ahead = src + (size_t)(e + 1 < rows ? e + 1 : rows - 1) * 1024;
#pragma unroll
            for (int r = 0; r < 8; ++r) nxt[r] = *reinterpret_cast<const f32x2 *>(ahead + 128 * r);
            if (V == V_LOAD_PIPE_MATH) {
                f32x2 tt[8];
#pragma unroll
                for (int i = 0; i < 8; ++i) asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel:[1,1] op_sel_hi:[1,0]" : "=&v"(tt[i]) : "v"(cur[i]), "v"(w));
#pragma unroll
                for (int i = 0; i < 8; ++i)
                    asm volatile("v_pk_fma_f32 %0, %0, %1, %2 op_sel:[0,0,0] op_sel_hi:[0,1,1] neg_lo:[0,0,1]" : "+v"(cur[i]) : "v"(w), "v"(tt[i]));
#pragma unroll
                for (int i = 0; i < 8; ++i) mine[lane + 64 * i] = cur[i];
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                __builtin_amdgcn_wave_barrier();
#pragma unroll
                for (int i = 0; i < 8; ++i) cur[i] = mine[((lane * 8 + i) * 9 + e) & 511];
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                __builtin_amdgcn_wave_barrier();
            }
#pragma unroll
            for (int r = 0; r < 8; ++r) acc[r] = acc[r] * 0.5f + cur[r];
            float *dst = reinterpret_cast<float *>(big_out) + (wave_id * rows + e) * 1024 + 2 * lane;
#pragma unroll
            for (int r = 0; r < 8; ++r) *reinterpret_cast<f32x2 *>(dst + 128 * r) = acc[r];
#pragma unroll
            for (int r = 0; r < 8; ++r) cur[r] = nxt[r];
        }
        x = f32x4{acc[0][0] + acc[1][1], acc[2][0] + acc[3][1], acc[4][0] + acc[5][1], acc[6][0] + acc[7][1]};
    }
    out[g] = x;
}

static void launch_victim(int v, f32x4 *out, f32x4 *scratch, int blocks, int iters, hipStream_t s) {
    if (v == V_SLOAD) {
        static std::vector<int> pattern;
        if (pattern.empty()) {
            pattern.resize(65536);
            for (int i = 0; i < 65536; ++i) pattern[i] = i * 2654435 + 12345;
        }
        CK(hipMemcpyAsync(scratch, pattern.data(), pattern.size() * sizeof(int), hipMemcpyHostToDevice, s));
    }
    switch (v) {
        case V_FMA: hipLaunchKernelGGL(k_victim<V_FMA>, dim3(blocks), dim3(256), 0, s, out, scratch, iters); break;
        case V_PK_FMA: hipLaunchKernelGGL(k_victim<V_PK_FMA>, dim3(blocks), dim3(256), 0, s, out, scratch, iters); break;
        case V_PK_MUL_ADD: hipLaunchKernelGGL(k_victim<V_PK_MUL_ADD>, dim3(blocks), dim3(256), 0, s, out, scratch, iters); break;
        case V_LDS: hipLaunchKernelGGL(k_victim<V_LDS>, dim3(blocks), dim3(256), 0, s, out, scratch, iters); break;
        case V_BPERMUTE: hipLaunchKernelGGL(k_victim<V_BPERMUTE>, dim3(blocks), dim3(256), 0, s, out, scratch, iters); break;
        case V_GLOBAL: hipLaunchKernelGGL(k_victim<V_GLOBAL>, dim3(blocks), dim3(256), 0, s, out, scratch, iters); break;
        case V_PK_ILP: hipLaunchKernelGGL(k_victim<V_PK_ILP>, dim3(blocks), dim3(256), 0, s, out, scratch, iters); break;
        case V_PK_OPSEL: hipLaunchKernelGGL(k_victim<V_PK_OPSEL>, dim3(blocks), dim3(256), 0, s, out, scratch, iters); break;
        case V_PK_OPSEL_ILP: hipLaunchKernelGGL(k_victim<V_PK_OPSEL_ILP>, dim3(blocks), dim3(256), 0, s, out, scratch, iters); break;
        case V_LDS_READ2: hipLaunchKernelGGL(k_victim<V_LDS_READ2>, dim3(blocks), dim3(256), 0, s, out, scratch, iters); break;
        case V_FFT_LIKE: hipLaunchKernelGGL(k_victim<V_FFT_LIKE>, dim3(blocks), dim3(256), 0, s, out, scratch, iters); break;
        case V_SGPR_HOLD: hipLaunchKernelGGL(k_victim<V_SGPR_HOLD>, dim3(blocks), dim3(256), 0, s, out, scratch, iters); break;
        case V_VGPR_HOLD: hipLaunchKernelGGL(k_victim<V_VGPR_HOLD>, dim3(blocks), dim3(256), 0, s, out, scratch, iters); break;
        case V_LDS_HOLD: hipLaunchKernelGGL(k_victim<V_LDS_HOLD>, dim3(blocks), dim3(256), 0, s, out, scratch, iters); break;
        case V_SLOAD: hipLaunchKernelGGL(k_victim<V_SLOAD>, dim3(blocks), dim3(256), 0, s, out, scratch, iters); break;
        case V_LOAD_PIPE: hipLaunchKernelGGL(k_victim<V_LOAD_PIPE>, dim3(blocks), dim3(256), 0, s, out, scratch, iters); break;
        default: hipLaunchKernelGGL(k_victim<V_LOAD_PIPE_MATH>, dim3(blocks), dim3(256), 0, s, out, scratch, iters); break;
    }
}

int main(int argc, char **argv) {
    const int kind = argc > 1 ? std::atoi(argv[1]) : 0;
    const int runs = argc > 2 ? std::atoi(argv[2]) : 40;
    const int blocks = 2048, iters = argc > 3 ? std::atoi(argv[3]) : 4096;
    const size_t n = (size_t)blocks * 256;
    f32x4 *d_out, *d_scratch;
    float *d_spin;
    CK(hipMalloc(&d_out, n * sizeof(f32x4)));
    CK(hipMalloc(&d_scratch, n * sizeof(f32x4)));
    CK(hipMalloc(&d_spin, 4096 * 64 * sizeof(float)));
    {
        const size_t big_n = (size_t)blocks * 4 * 32 * 1024;
        float *in = nullptr, *outp = nullptr;
        CK(hipMalloc(&in, big_n * sizeof(float)));
        CK(hipMalloc(&outp, big_n * sizeof(float)));
        hipLaunchKernelGGL(k_fill, dim3(4096), dim3(256), 0, 0, in, big_n);
        CK(hipMemcpyToSymbol(HIP_SYMBOL(big_in), &in, sizeof(in)));
        CK(hipMemcpyToSymbol(HIP_SYMBOL(big_out), &outp, sizeof(outp)));
        CK(hipDeviceSynchronize());
    }
    hipStream_t sa, sv;
    CK(hipStreamCreateWithFlags(&sa, hipStreamNonBlocking));
    CK(hipStreamCreateWithFlags(&sv, hipStreamNonBlocking));
    std::vector<std::vector<uint32_t>> ref(V_COUNT, std::vector<uint32_t>(n * 4));
    std::vector<uint32_t> got(n * 4);
    for (int v = 0; v < V_COUNT; ++v) {  // alone: the reference, and that it repeats
        long unstable = 0;
        for (int r = 0; r < 4; ++r) {
            launch_victim(v, d_out, d_scratch, blocks, iters, sv);
            CK(hipStreamSynchronize(sv));
            CK(hipMemcpy(r ? got.data() : ref[v].data(), d_out, n * sizeof(f32x4), hipMemcpyDeviceToHost));
            if (r && std::memcmp(got.data(), ref[v].data(), n * sizeof(f32x4)) != 0) unstable += 1;
        }
        std::printf("alone   %-28s repeats itself: %s\n", kVictimName[v], unstable ? "NO" : "yes");
    }
    std::atomic<bool> stop{false};
    std::atomic<long> spins{0};
    std::thread aggressor([&] {
        (void)hipSetDevice(0);
        while (!stop.load()) {
            for (int r = 0; r < 16; ++r) {
                if (kind == 0) hipLaunchKernelGGL(k_spin<0>, dim3(2048), dim3(64), 0, sa, d_spin, 4000);
                else if (kind == 1) hipLaunchKernelGGL(k_spin<1>, dim3(2048), dim3(64), 0, sa, d_spin, 2000);
                else if (kind == 3) hipLaunchKernelGGL(k_spin<3>, dim3(2048), dim3(64), 0, sa, d_spin, 4000);
                else if (kind == 2) hipLaunchKernelGGL(k_spin<2>, dim3(2048), dim3(64), 0, sa, d_spin, 4000);
            }
            (void)hipStreamSynchronize(sa);
            spins += 16;
        }
    });
    while (spins.load() == 0) std::this_thread::yield();
    for (int v = 0; v < V_COUNT; ++v) {
        long bad_runs = 0, bad_words = 0;
        uint32_t ex_ref = 0, ex_got = 0;
        for (int r = 0; r < runs; ++r) {
            launch_victim(v, d_out, d_scratch, blocks, iters, sv);
            CK(hipStreamSynchronize(sv));
            CK(hipMemcpy(got.data(), d_out, n * sizeof(f32x4), hipMemcpyDeviceToHost));
            long w = 0;
            for (size_t i = 0; i < n * 4; ++i)
                if (got[i] != ref[v][i]) {
                    if (!w && !bad_words) ex_ref = ref[v][i], ex_got = got[i];
                    w += 1;
                }
            bad_runs += w != 0;
            bad_words += w;
        }
        std::printf("beside  %-28s %ld of %d runs differ, %ld words (first: %08x -> %08x)\n", kVictimName[v], bad_runs, runs, bad_words, ex_ref, ex_got);
    }
    stop = true;
    aggressor.join();
    std::printf("aggressor kind %d launched %ld times\n", kind, spins.load());
    return 0;
}
