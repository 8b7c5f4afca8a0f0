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

#include <cstdlib>
#include <type_traits>

namespace sk {

namespace {

constexpr int kChunk = 128;       // samples per row per staged chunk
constexpr int kBlockStride = 264; // dwords per (slot, row-pair) block: 2 x 128 samples + 8 pad
constexpr int kRingDwords = 16 * kBlockStride;  // 2 slots x 8 row pairs
constexpr int kSteps = 76;

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void lds_void;
typedef const __attribute__((address_space(1))) void gbl_void;

// LDS ring layout.  A chunk is 128 samples of each of the 16 rows; the ring holds two chunks.
// One LDS-DMA instruction moves 1 KiB = rows (p, p+8) x 128 samples into block (slot, p):
//     dword address = (slot*8 + (row & 7)) * 264 + (row >> 3) * 128 + (sample & 127)
// 264 = 4 * 66 and 66 mod 16 == 2, which makes the ds_read_b128 B-operand loads conflict-free:
// in every 16-lane read group the 16-byte slot index is 2*(row & 7) + (lane >> 4) mod 16, all distinct.
__device__ __forceinline__ int ring_addr(int row, int sample) {
    return (((sample >> 7) & 1) * 8 + (row & 7)) * kBlockStride + (row >> 3) * 128 + (sample & 127);
}

// element offset of sample idx of row `row`: plain rows, or the frame-packed planar layout the AAC
// synthesis writes (FirArgs::in_block == 1024, in_ch 1 or 2) -- shifts and masks only: this runs per
// lane per DMA instruction inside an MFMA-bound loop
template <bool PACKED>
__device__ __forceinline__ size_t sample_offset(const FirArgs &a, uint32_t row, int64_t idx) {
    if (!PACKED) return (size_t)row * a.in_stride + (size_t)idx;
    const uint32_t sh = a.in_ch - 1;  // channels 1 -> 0, 2 -> 1
    const uint32_t g = row >> sh, c = row & sh;
    const uint32_t i = (uint32_t)idx;
    return (size_t)g * a.in_group_stride + (size_t)(c << 10) + (size_t)(i >> 10) * a.in_block_stride + (size_t)(i & 1023);
}

// Per-lane constants of the aligned staging path, computed once per wave: f32 MFMA issues through the same vector
// pipeline as ordinary VALU work on this part (equal f32 matrix and vector peaks; profiles/r01_pmc_pipeline.md: every
// VALU instruction added to the loop showed up 1:1 in the launch time), so the per-chunk address arithmetic is kept
// to one offset shared by the eight DMA instructions plus a pointer add and select each.
struct StageLane {
    const float *base[8];  // row p (lanes 0-31) or p + 8 (lanes 32-63): address of its sample 0
    uint32_t ok_mask;      // bit p: that row exists
};

template <bool PACKED>
__device__ __forceinline__ void stage_lane_init(const FirArgs &a, int lane, uint32_t row0, const uint32_t (&phys)[16], StageLane &st) {
    st.ok_mask = 0;
#pragma unroll
    for (int p = 0; p < 8; ++p) {
        const uint32_t row = (lane >> 5) ? phys[p + 8] : phys[p];
        if (row0 + p + 8 * (lane >> 5) < a.rows) st.ok_mask |= 1u << p;
        st.base[p] = a.in + sample_offset<PACKED>(a, row, 0);
    }
}

template <bool ALIGNED, bool PACKED>
__device__ __forceinline__ void stage_chunk(const FirArgs &a, float *ring, int lane, uint32_t row0, const uint32_t (&phys)[16],
                                            int64_t t0, int chunk, const StageLane &st) {
    // chunk c covers local samples n'' in [128 c, 128 c + 128); stream time = n'' + t0
    const int slot = chunk & 1;
    if (ALIGNED) {
        // lanes 0-31 -> row p, lanes 32-63 -> row p + 8, 4 samples per lane; the sample index is the same for every p
        const int64_t idx = (int64_t)kChunk * chunk + 4 * (lane & 31) + t0 - a.in_origin;
        const bool in_range = (uint64_t)idx + 3 < (uint64_t)a.in_frames;  // idx < 0 wraps far above
        const uint32_t i = (uint32_t)idx;
        size_t off = PACKED ? (size_t)(i >> 10) * a.in_block_stride + (size_t)(i & 1023) : (size_t)i;
        asm volatile("" : "+v"(off));  // one sum, added once per instruction below (not re-associated into each of them)
        const float *zero = a.zeros + 4 * lane;
        const uint32_t sel = in_range ? st.ok_mask : 0u;
#pragma unroll
        for (int p = 0; p < 8; ++p) {
            // a select, not a branch: an LDS-DMA instruction must be issued once for the whole wave (the asm keeps the
            // compiler from unswitching the loop on the lane's range test, which doubled every DMA instruction)
            uint32_t okp = (sel >> p) & 1u;
            asm volatile("" : "+v"(okp));
#ifdef SK_FIR_ABLATE_ZEROSRC
            const float *src = okp == 77u ? st.base[p] + off : zero;
#else
            const float *src = okp ? st.base[p] + off : zero;
#endif
            __builtin_amdgcn_global_load_lds((gbl_void *)src, (lds_void *)(ring + (slot * 8 + p) * kBlockStride), 16, 0, 0);
        }
    } else {
#pragma unroll
        for (int rr = 0; rr < 16; ++rr) {
            const bool row_ok = row0 + rr < a.rows;
            const uint32_t row = phys[rr];
#pragma unroll
            for (int h = 0; h < 2; ++h) {  // 64 samples of one row per instruction
                const int64_t idx = (int64_t)kChunk * chunk + 64 * h + lane + t0 - a.in_origin;
                const bool ok = row_ok && idx >= 0 && idx < (int64_t)a.in_frames;
                const float *src = ok ? a.in + sample_offset<PACKED>(a, row, idx) : a.zeros + lane;
                __builtin_amdgcn_global_load_lds(
                    (gbl_void *)src, (lds_void *)(ring + (slot * 8 + (rr & 7)) * kBlockStride + (rr >> 3) * 128 + 64 * h),
                    4, 0, 0);
            }
        }
    }
}

// s_waitcnt vmcnt(n): all but the n youngest vector-memory operations have completed.  The staged
// chunk's LDS-DMA loads are older than the output stores issued since, so waiting down to `younger`
// outstanding operations retires the loads without draining those stores.
__device__ __forceinline__ void wait_vm_older_than(int younger) {
    switch (younger) {
    case 0: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
    case 1: asm volatile("s_waitcnt vmcnt(1)" ::: "memory"); break;
    case 2: asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); break;
    case 3: asm volatile("s_waitcnt vmcnt(3)" ::: "memory"); break;
    default: asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); break;
    }
}

// OUT16: 0 = f32 rows out; 1 / 2 = the worker's 16-bit output stage fused (mono rows / stereo row pairs interleaved)
template <bool ALIGNED, bool PACKED, int OUT16>
__global__ __launch_bounds__(64, 2) void k_fir_48k_16k(FirArgs a, uint32_t total_blocks, uint32_t blocks_per_seg,
                                                       uint32_t n_segs, int out_vec) {
    // + 2 KiB: 18.5 KiB per wave caps residency at 8 waves per CU (160 KiB LDS) = 2 per SIMD whatever the register
    // count; the launcher sizes the grid to exactly that many waves, and a ninth wave on some CUs (which 160 VGPRs
    // would allow) leaves one SIMD with three waves and the launch waiting for it (measured: 0.97 -> 1.35 ms)
    __shared__ float ring[kRingDwords + 512];

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

    // physical input rows of this tile (wave-uniform, resolved once: no loads inside the MFMA loop)
    uint32_t phys[16];
#pragma unroll
    for (int rr = 0; rr < 16; ++rr) {
        const uint32_t r = row0 + rr < a.rows ? row0 + rr : 0;
        phys[rr] = a.row_map ? __builtin_amdgcn_readfirstlane(a.row_map[r]) : r;
    }

    f32x4 acc[7];
#pragma unroll
    for (int b = 0; b < 7; ++b) acc[b] = (f32x4){0.f, 0.f, 0.f, 0.f};

    const int g_first = 3 * a_begin;
    int chunk = g_first >> 3;  // 8 groups of 16 samples per chunk
    StageLane st;
    if (ALIGNED) stage_lane_init<PACKED>(a, lane, row0, phys, st);
    stage_chunk<ALIGNED, PACKED>(a, ring, lane, row0, phys, t0, chunk, st);
    stage_chunk<ALIGNED, PACKED>(a, ring, lane, row0, phys, t0, chunk + 1, st);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    int stores_since_stage = 0;  // wave-uniform

    const uint32_t out_row = row0 + j;
    constexpr bool to_s16 = OUT16 != 0;
    float *out_ptr = to_s16 ? nullptr
                            : a.out + (size_t)(out_row < a.rows ? out_row : 0) * a.out_stride +
                                  ((a.out_off && out_row < a.rows) ? a.out_off[out_row] : 0);
    // s16: the lane of a stream's first channel writes the interleaved frames of all its channels
    int16_t *out16_ptr = to_s16 ? a.out16 + (size_t)((out_row < a.rows ? out_row : 0) / (OUT16 ? OUT16 : 1)) * a.out16_stride * OUT16
                                : nullptr;
    const int lane_base = ring_addr(j, 4 * kq);  // + slot and group offsets per read
    auto read_group = [&](int G) {
#ifdef SK_FIR_ABLATE_LDSREAD
        return (f32x4){(float)G, 1.f, 2.f, 3.f};
#else
        return *reinterpret_cast<const f32x4 *>(&ring[lane_base + ((G >> 3) & 1) * 8 * kBlockStride + 16 * (G & 7)]);
#endif
    };

    // B operands are read one group ahead of the MFMAs that consume them.  Seven periods are
    // unrolled so that block `blk` owns accumulator blk mod 7 statically (no register shuffling
    // between periods, and no VALU reads of accumulators an MFMA has just written).
    f32x4 xb = read_group(g_first);
    // one period (3 groups of 16 samples) with A mod 7 == P known at compile time
    auto period = [&](auto ptag, int32_t A) __attribute__((always_inline)) {
        constexpr int p = decltype(ptag)::value;
        acc[p] = (f32x4){0.f, 0.f, 0.f, 0.f};  // block A starts in this period
        // s16 output: block A-6 completes in group 0; its four values are converted one per MFMA step of group 1
        // and packed / stored after group 2, so the VALU work sits between MFMAs instead of stalling the pipe
        f32x4 pending = (f32x4){0.f, 0.f, 0.f, 0.f};
        int q[4] = {0, 0, 0, 0};
#pragma unroll
        for (int gi = 0; gi < 3; ++gi) {
            const int Gn = 3 * A + gi + 1;  // the group after the one computed now
            if ((Gn & 7) == 0 && (Gn >> 3) != chunk) {
                // Gn opens chunk Gn>>3 (staged one chunk ago); every read of its predecessor has been
                // issued (the current group's operands are in xb), so that slot can be refilled
                chunk = Gn >> 3;
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#ifndef SK_FIR_ABLATE_VMWAIT
                wait_vm_older_than(ALIGNED ? __builtin_amdgcn_readfirstlane(stores_since_stage) : 0);
#endif
#ifndef SK_FIR_ABLATE_STAGE
                stage_chunk<ALIGNED, PACKED>(a, ring, lane, row0, phys, t0, chunk + 1, st);
#endif
                stores_since_stage = 0;
            }
            const f32x4 xb_next = read_group(Gn);
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const int u = 4 * gi + t;
#pragma unroll
                for (int b = 0; b < 7; ++b) {  // block A - b is at step 12 b + u
                    if (12 * b + u < kSteps)
                        acc[(p - b + 7) % 7] =
                            __builtin_amdgcn_mfma_f32_16x16x4f32(af[12 * b + u], xb[t], acc[(p - b + 7) % 7], 0, 0, 0);
                }
                if (to_s16 && gi == 1) q[t] = dev_float_sample_to_i16_f32(pending[t]);
            }
            const int32_t blk = A - 6;  // complete after its step 75 (= u 3 of this period)
            const uint32_t m = (uint32_t)blk * 16 + 4 * kq;  // relative to out_first
            if (gi == 0) {
                pending = acc[(p + 1) % 7];
#ifdef SK_FIR_ABLATE_STORE
                if (!to_s16 && blk >= a_begin && blk < a_end && pending[0] == 1.2345e30f) {
#else
                if (!to_s16 && blk >= a_begin && blk < a_end) {  // wave-uniform
#endif
                    if (out_vec) ++stores_since_stage;
                    if (out_row < a.rows) {
                        if (out_vec && m + 3 < a.out_count) {
                            *reinterpret_cast<f32x4 *>(out_ptr + m) = pending;
                        } else {
#pragma unroll
                            for (int r = 0; r < 4; ++r)
                                if (m + r < a.out_count) out_ptr[m + r] = pending[r];
                        }
                    }
                }
            }
            if (to_s16 && gi == 2) {
                const uint32_t mine01 = ((uint32_t)q[0] & 0xffffu) | ((uint32_t)q[1] << 16);
                const uint32_t mine23 = ((uint32_t)q[2] & 0xffffu) | ((uint32_t)q[3] << 16);
                if (OUT16 == 2) {
                    // the neighbouring row's lane (lane ^ 1) holds the other channel: two packed dwords cross by DPP
                    // quad_perm [1,0,3,2], no LDS round trip in the middle of the MFMA stream
                    const uint32_t other01 = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)mine01, 0xB1, 0xF, 0xF, true);
                    const uint32_t other23 = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)mine23, 0xB1, 0xF, 0xF, true);
                    uint32_t w[4];  // frame m + r: L in the low half, R in the high half
                    w[0] = (mine01 & 0xffffu) | (other01 << 16);
                    w[1] = (mine01 >> 16) | (other01 & 0xffff0000u);
                    w[2] = (mine23 & 0xffffu) | (other23 << 16);
                    w[3] = (mine23 >> 16) | (other23 & 0xffff0000u);
                    if (blk >= a_begin && blk < a_end) {  // wave-uniform
                        if (out_vec) ++stores_since_stage;
                        if ((j & 1) == 0 && out_row < a.rows) {
                            uint32_t *dst = reinterpret_cast<uint32_t *>(out16_ptr) + m;
                            if (out_vec && m + 3 < a.out_count) {
                                *reinterpret_cast<uint4 *>(dst) = make_uint4(w[0], w[1], w[2], w[3]);
                            } else {
#pragma unroll
                                for (int r = 0; r < 4; ++r)
                                    if (m + r < a.out_count) dst[r] = w[r];
                            }
                        }
                    }
                } else if (blk >= a_begin && blk < a_end) {  // wave-uniform
                    if (out_vec) ++stores_since_stage;
                    if (out_row < a.rows) {
                        int16_t *dst = out16_ptr + m;
                        if (out_vec && m + 3 < a.out_count) {
                            *reinterpret_cast<uint2 *>(dst) = make_uint2(mine01, mine23);
                        } else {
#pragma unroll
                            for (int r = 0; r < 4; ++r)
                                if (m + r < a.out_count) dst[r] = (int16_t)q[r];
                        }
                    }
                }
            }
            xb = xb_next;
        }
    };
    const int32_t a_last = a_end + 6;
    for (int32_t A0 = a_begin - a_begin % 7; A0 < a_last; A0 += 7) {
        // skipping a period is wave-uniform; periods before a_begin only exist in the first trip
        if (A0 + 0 >= a_begin && A0 + 0 < a_last) period(std::integral_constant<int, 0>{}, A0 + 0);
        if (A0 + 1 >= a_begin && A0 + 1 < a_last) period(std::integral_constant<int, 1>{}, A0 + 1);
        if (A0 + 2 >= a_begin && A0 + 2 < a_last) period(std::integral_constant<int, 2>{}, A0 + 2);
        if (A0 + 3 >= a_begin && A0 + 3 < a_last) period(std::integral_constant<int, 3>{}, A0 + 3);
        if (A0 + 4 >= a_begin && A0 + 4 < a_last) period(std::integral_constant<int, 4>{}, A0 + 4);
        if (A0 + 5 >= a_begin && A0 + 5 < a_last) period(std::integral_constant<int, 5>{}, A0 + 5);
        if (A0 + 6 >= a_begin && A0 + 6 < a_last) period(std::integral_constant<int, 6>{}, A0 + 6);
    }
}

}  // namespace

hipError_t launch_fir_48k_16k(const FirArgs &a, hipStream_t s) {
    if (a.rows == 0 || a.out_count == 0) return hipSuccess;
    // SK_FIR_F32=1 keeps the f32-MFMA kernel below (A/B runs, profiles/r01_ab_fir.md)
    static const bool force_f32 = [] { const char *v = std::getenv("SK_FIR_F32"); return v && v[0] == '1'; }();
    if (a.in16) return fir_bf16_supported(a) ? launch_fir_48k_16k_bf16(a, s) : hipErrorInvalidValue;  // s16 rows: bf16 kernel only
    if (!force_f32 && fir_bf16_supported(a)) return launch_fir_48k_16k_bf16(a, s);
    const uint32_t total_blocks = (a.out_count + 15) / 16;
    const uint32_t groups = (a.rows + 15) / 16;
    // two waves per SIMD across the chip (256 CUs x 4 x 2) is the residency this kernel's LDS allows;
    // split the time axis until there are about that many waves, keeping segments >= 32 blocks
    uint32_t n_segs = (2048 + groups - 1) / groups;
    const uint32_t max_segs = total_blocks / 32 ? total_blocks / 32 : 1;
    if (n_segs > max_segs) n_segs = max_segs;
    if (n_segs < 1) n_segs = 1;
    const uint32_t blocks_per_seg = (total_blocks + n_segs - 1) / n_segs;
    n_segs = (total_blocks + blocks_per_seg - 1) / blocks_per_seg;

    const int64_t t0 = (int64_t)3 * a.out_first - 128;
    const bool strides_ok = a.in_block ? (a.in_block % 4 == 0 && a.in_block_stride % 4 == 0 && a.in_group_stride % 4 == 0)
                                       : (a.in_stride % 4 == 0);
    const bool aligned = (((t0 - a.in_origin) & 3) == 0) && strides_ok && (a.in_frames % 4 == 0) &&
                         (((uintptr_t)a.in & 15) == 0);
    if (a.out16 && (a.out16_ch < 1 || a.out16_ch > 2 || a.out_off || a.rows % a.out16_ch)) return hipErrorInvalidValue;
    const int out_vec = a.out16 ? ((a.out16_stride * a.out16_ch) % 8 == 0 && (((uintptr_t)a.out16 & 15) == 0))
                                : ((a.out_stride % 4 == 0) && (((uintptr_t)a.out & 15) == 0) && a.out_off == nullptr);
    const dim3 grid(groups * n_segs), block(64);
    const bool packed = a.in_block != 0;
    if (packed && (a.in_block != 1024 || a.in_ch < 1 || a.in_ch > 2)) return hipErrorInvalidValue;
#define SK_FIR_LAUNCH(AL, PK, O16) \
    hipLaunchKernelGGL((k_fir_48k_16k<AL, PK, O16>), grid, block, 0, s, a, total_blocks, blocks_per_seg, n_segs, out_vec)
    if (a.out16) {  // the fused 16-bit output exists for the frame-packed input of the synthesis kernel
        if (!packed) return hipErrorInvalidValue;
        if (aligned && a.out16_ch == 2) SK_FIR_LAUNCH(true, true, 2);
        else if (aligned) SK_FIR_LAUNCH(true, true, 1);
        else if (a.out16_ch == 2) SK_FIR_LAUNCH(false, true, 2);
        else SK_FIR_LAUNCH(false, true, 1);
    } else if (aligned && packed) SK_FIR_LAUNCH(true, true, 0);
    else if (aligned) SK_FIR_LAUNCH(true, false, 0);
    else if (packed) SK_FIR_LAUNCH(false, true, 0);
    else SK_FIR_LAUNCH(false, false, 0);
#undef SK_FIR_LAUNCH
    return hipGetLastError();
}

}  // namespace sk
