// fir.hip -- 48 kHz -> 16 kHz sinc FIR decimator on the gfx950 f32 matrix cores.
//
// Replaces the inner loop of rubato 0.14.1 SincFixedIn<f32>::process as the reference
// configures it (soundkit/src/audio_pipeline.rs:474-491, soundkit-decoder/src/lib.rs:1939-1998):
// at ratio 1/3 the time step is exactly 3.0 and the fractional phase always 0, so
//     y[m] = sum_{p=0}^{255} h[p] * x[3m - 125 + p]
// with h = sub-filter 0 of the 256x256 windowed-sinc table.  This is the one dense
// sample x tap contraction on the path, so it runs on v_mfma_f32_16x16x4_f32
// (exact f32 fma chain):
//
//   D[i][j] += A[i][k] * B[k][j]     i = output sample within a block of 16
//                                    j = one of 16 independent channel signals (rows)
//                                    k = 4 input samples per MFMA step
//   A[i][k] = h[n - 3 i - 3]         (Toeplitz band of the taps; constant -> 76 VGPRs)
//   B[k][j] = x_j[n]                 (from an LDS ring, one ds_read_b128 per 4 steps)
//
// A block of 16 outputs spans 301 input samples = 19 groups of 16 = 76 MFMA steps, and
// consecutive blocks start 48 samples = 3 groups = 12 steps apart, so 7 blocks are live
// at once: every B fragment read from LDS feeds up to 7 MFMAs into 7 independent
// accumulators (which also covers the 40-cycle dependent-issue latency of this MFMA).
// Useful MACs / issued MACs = 256 / 304 = 84 %.
//
// Inside a group of 16 samples the k index is permuted (lane k-slot q, step t <-> sample
// 4q + t) so that one lane's four B operands are 4 contiguous floats in LDS.
#include "sk_device.h"

namespace sk {

namespace {

constexpr int kRing = 512;        // samples per row kept in LDS (two 256-sample chunks)
constexpr int kRowStride = 520;   // dwords; 520 mod 64 == 8 makes the ds_read_b128 B loads conflict-free
constexpr int kSteps = 76;

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void lds_void;
typedef const __attribute__((address_space(1))) void gbl_void;

template <bool ALIGNED>
__device__ __forceinline__ void stage_chunk(const FirArgs &a, float *ring, int lane, uint32_t row0, int64_t t0,
                                            int chunk) {
    // chunk c covers local samples n'' in [256 c, 256 c + 256); stream time = n'' + t0
    const int half = chunk & 1;
#pragma unroll
    for (int rr = 0; rr < 16; ++rr) {
        const uint32_t row = row0 + rr;
        const bool row_ok = row < a.rows;
        const float *row_ptr = a.in + (size_t)(row_ok ? row : 0) * a.in_stride;
        if (ALIGNED) {
            const int64_t idx = (int64_t)256 * chunk + 4 * lane + t0 - a.in_origin;
            const bool ok = row_ok && idx >= 0 && idx + 3 < (int64_t)a.in_frames;
            const float *src = ok ? row_ptr + idx : a.zeros + 4 * lane;
            __builtin_amdgcn_global_load_lds((gbl_void *)src, (lds_void *)(ring + rr * kRowStride + half * 256), 16, 0,
                                             0);
        } else {
#pragma unroll
            for (int qd = 0; qd < 4; ++qd) {
                const int64_t idx = (int64_t)256 * chunk + 64 * qd + lane + t0 - a.in_origin;
                const bool ok = row_ok && idx >= 0 && idx < (int64_t)a.in_frames;
                const float *src = ok ? row_ptr + idx : a.zeros + lane;
                __builtin_amdgcn_global_load_lds((gbl_void *)src,
                                                 (lds_void *)(ring + rr * kRowStride + half * 256 + 64 * qd), 4, 0, 0);
            }
        }
    }
}

template <bool ALIGNED>
__global__ __launch_bounds__(64) void k_fir_48k_16k(FirArgs a, uint32_t total_blocks, uint32_t blocks_per_seg,
                                                    uint32_t n_segs, int out_vec) {
    __shared__ float ring[16 * kRowStride];

    const int lane = threadIdx.x;
    const int j = lane & 15, kq = lane >> 4;
    const uint32_t group = blockIdx.x / n_segs, seg = blockIdx.x % n_segs;
    const uint32_t row0 = group * 16;
    const int32_t a_begin = (int32_t)(seg * blocks_per_seg);
    int32_t a_end = a_begin + (int32_t)blocks_per_seg;
    if (a_end > (int32_t)total_blocks) a_end = (int32_t)total_blocks;
    if (a_begin >= a_end) return;

    // stream time of local sample n'' = 0:  n = n'' + 3*out_first - 128
    const int64_t t0 = (int64_t)3 * a.out_first - 128;

    float af[kSteps];
#pragma unroll
    for (int s = 0; s < kSteps; ++s) af[s] = a.afrag[s * 64 + lane];

    f32x4 acc[7];
#pragma unroll
    for (int b = 0; b < 7; ++b) acc[b] = (f32x4){0.f, 0.f, 0.f, 0.f};

    const int g_first = 3 * a_begin;
    int chunk = g_first >> 4;
    stage_chunk<ALIGNED>(a, ring, lane, row0, t0, chunk);
    stage_chunk<ALIGNED>(a, ring, lane, row0, t0, chunk + 1);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");

    const uint32_t out_row = row0 + j;
    float *out_ptr = a.out + (size_t)(out_row < a.rows ? out_row : 0) * a.out_stride;

    for (int32_t A = a_begin; A < a_end + 6; ++A) {
#pragma unroll
        for (int gi = 0; gi < 3; ++gi) {
            const int G = 3 * A + gi;
            if ((G & 15) == 0 && (G >> 4) != chunk) {
                // entering chunk G>>4 (staged one chunk ago); its predecessor's half is free again
                chunk = G >> 4;
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                stage_chunk<ALIGNED>(a, ring, lane, row0, t0, chunk + 1);
            }
            const f32x4 xb = *reinterpret_cast<const f32x4 *>(&ring[j * kRowStride + ((16 * G + 4 * kq) & (kRing - 1))]);
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const int u = 4 * gi + t;
#pragma unroll
                for (int b = 0; b < 7; ++b) {
                    if (12 * b + u < kSteps) acc[b] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[12 * b + u], xb[t], acc[b], 0, 0, 0);
                }
            }
            if (gi == 0) {
                // block A-6 is complete after its step 75 (= u 3 of this period)
                const int32_t blk = A - 6;
                if (blk >= a_begin && blk < a_end && out_row < a.rows) {
                    const uint32_t m = (uint32_t)blk * 16 + 4 * kq;  // relative to out_first
                    if (out_vec && m + 3 < a.out_count) {
                        *reinterpret_cast<f32x4 *>(out_ptr + m) = acc[6];
                    } else {
#pragma unroll
                        for (int r = 0; r < 4; ++r)
                            if (m + r < a.out_count) out_ptr[m + r] = acc[6][r];
                    }
                }
            }
        }
#pragma unroll
        for (int b = 6; b > 0; --b) acc[b] = acc[b - 1];
        acc[0] = (f32x4){0.f, 0.f, 0.f, 0.f};
    }
}

}  // namespace

hipError_t launch_fir_48k_16k(const FirArgs &a, hipStream_t s) {
    if (a.rows == 0 || a.out_count == 0) return hipSuccess;
    const uint32_t total_blocks = (a.out_count + 15) / 16;
    const uint32_t groups = (a.rows + 15) / 16;
    // one wave per SIMD across the chip (256 CUs x 4) is the residency this kernel's LDS allows;
    // split the time axis until there are about that many waves, keeping segments >= 32 blocks
    uint32_t n_segs = (1024 + groups - 1) / groups;
    const uint32_t max_segs = total_blocks / 32 ? total_blocks / 32 : 1;
    if (n_segs > max_segs) n_segs = max_segs;
    if (n_segs < 1) n_segs = 1;
    const uint32_t blocks_per_seg = (total_blocks + n_segs - 1) / n_segs;
    n_segs = (total_blocks + blocks_per_seg - 1) / blocks_per_seg;

    const int64_t t0 = (int64_t)3 * a.out_first - 128;
    const bool aligned = (((t0 - a.in_origin) & 3) == 0) && (a.in_stride % 4 == 0) && (a.in_frames % 4 == 0) &&
                         (((uintptr_t)a.in & 15) == 0);
    const int out_vec = (a.out_stride % 4 == 0) && (((uintptr_t)a.out & 15) == 0);
    const dim3 grid(groups * n_segs), block(64);
    if (aligned)
        hipLaunchKernelGGL(k_fir_48k_16k<true>, grid, block, 0, s, a, total_blocks, blocks_per_seg, n_segs, out_vec);
    else
        hipLaunchKernelGGL(k_fir_48k_16k<false>, grid, block, 0, s, a, total_blocks, blocks_per_seg, n_segs, out_vec);
    return hipGetLastError();
}

}  // namespace sk
