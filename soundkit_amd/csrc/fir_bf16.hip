// fir_bf16.hip -- the 48 kHz -> 16 kHz sinc FIR on the gfx950 bf16 matrix cores, f32-accurate.
//
// Same filter as fir.hip (rubato 0.14.1 SincFixedIn<f32> as soundkit configures it, soundkit/src/audio_pipeline.rs:474-491,
// soundkit-decoder/src/lib.rs:1939-1998):   y[m] = sum_{p=0}^{255} h[p] * x[3m - 125 + p].
//
// The f32 MFMA of fir.hip issues through the vector ALUs (157 TFLOP/s); the bf16 matrix cores are 16x faster.  An f32
// value is *exactly* the sum of three bf16 values (8 + 8 + 8 significand bits, taken by truncation):
//     x = x1 + x2 + x3,  h = h1 + h2 + h3
// and the six products  x1h1, x1h2, x2h1, x2h2, x1h3, x3h1  accumulated in f32 leave out only terms below 2^-24 of
// |x||h| -- the size of one f32 rounding.  Measured against the filter evaluated in f64: 2.2e-7 relative RMS (the oracle's
// f32 chain: 1.15e-7), at full scale and at 1e-6 of it alike (tests/test_fir_gpu.py); the bound is 1e-6.
//
//   D[i][j] += A[i][k] * B[k][j]     v_mfma_f32_16x16x32_bf16
//       i = output within a tile of 16 (absolute outputs 16 T + i)
//       j = one of 16 independent rows (channel signals)
//       k = 32 consecutive input samples (one "window")
//   A_s[i][k] = h[32 s + k - 3 i - 3]   s = 0..9: a tile reads 10 windows starting at sample 48 T - 128
//   B  [k][j] = x_j[window start + k]
//
// Tiles 48 samples apart alternate between windows aligned at 0 and at 16 modulo 32, so even and odd tiles read the
// row through two window grids (16 samples apart) and share ONE set of A fragments: 10 steps x {h1,h2,h3} x 4 VGPRs =
// 120 registers at most (92 with the products that are left out, below).  A window of one grid feeds the 3-4 tiles of
// that parity that are in flight (steps s, s+3, s+6, s+9), up to six MFMAs each.  Useful MACs / issued = 256 / 320 = 80 %.
//
// Input rows are loaded to registers 64 samples at a time, split into the three bf16 planes there (v_and / v_sub /
// v_perm: the matrix instruction holds the vector issue port for only half of its 16 cycles, so this work sits in the
// gaps) and written to a 192-sample LDS ring per plane; B operands are one ds_read_b128 per plane.
#include "sk_device.h"

#include <cstdlib>
#include <type_traits>

namespace sk {

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));

constexpr int kRing = 192;                   // samples of each row held in LDS
// LDS image: the planes of a row lie side by side -- row j at j * row_bytes, its plane p at + p * kPlaneBytes (384 bytes: the ring)
// -- with a row pitch of planes * 384 + 32 bytes = 50 (two planes) or 74 (three) 16-byte slots, both 2 modulo 8.
// Why: a wave's ds_read_b128 is served in four groups of 16 lanes, and the groups are NOT the four quarters of the wave:
// {0-3, 12-15, 20-27}, {4-11, 16-19, 28-31} and the same again 32 lanes up (MI355X_MICROARCH.md, LDS).  With lane = 16 q + j
// reading row j at slot q + const, a group holds every row once, eight of them (j = 4..11) one slot further on than the other
// eight: the slots  j * pitch + [4 <= j < 12]  (and the complement) must be distinct modulo 16, which a pitch of 2 or 6 modulo 8
// gives.  The first layout (one plane after the other, row pitch 25 slots: distinct over 16 lanes of EQUAL q, which no group
// is) put seven of every group's sixteen lanes on a bank quad already in use: SQ_LDS_BANK_CONFLICT was a third of the LDS cycles
// (profiles/r03_pmc.json).  The image is no larger than before (12 800 bytes for two planes: 12 workgroups per CU still fit;
// a pitch of 26 slots per plane measured 17 % SLOWER because 13 312 bytes are 11 granules and only 11 workgroups fit: r04_ab_fir.md).
constexpr int kPlaneBytes = 2 * kRing;  // one row of one plane
__host__ __device__ constexpr int row_bytes(int planes) { return planes * kPlaneBytes + 32; }
constexpr int kWindows = 10;                 // windows (K steps) per tile
// Products kept per window, in the order x1h1 | x1h2, x2h1 | x2h2, x1h3, x3h1 (relative size 1 | 2^-8 | 2^-16).  The taps a
// window can meet are small at both ends of the filter -- largest |h| relative to the peak tap: 2^-18.7, 2^-11.1, 2^-6.6,
// 2^-2.4, 1, 1, 2^-4.5, 2^-8.1, 2^-13.9, 2^-24.6 for windows 0..9 -- and a product whose size times that factor is under
// 2^-24 of the peak cannot be told from f32 rounding: 41 MFMAs per tile instead of 60, the same error against an f64
// evaluation to three digits (tests/test_fir_gpu.py holds the bound, tests/fir_split_model.py shows the budget).
__device__ constexpr int kProducts[kWindows] = {1, 3, 6, 6, 6, 6, 6, 3, 3, 1};
// s16 input (the worker's path: the resampler is fed float_sample_to_i16(x) / 32768, soundkit-decoder lib.rs:1793-1813,
// 3563-3617): a 16-bit sample is exactly TWO bf16 values, x * 32768 = 256 a + b with a = s >> 8, b = s & 255, so the third
// input plane and the product x3h1 do not exist -- 36 MFMAs per tile, two LDS planes, half the input bytes.  The factor
// 2^-15 is applied to the accumulated sums (a power of two: it commutes with every rounding).
__host__ __device__ constexpr int products_of(int s, bool in16) { return in16 && kProducts[s] > 5 ? 5 : kProducts[s]; }
// s16 input on the f16 matrix instruction (the default for s16 rows): both operands as two f16 values,
//     x = x1 + x2 (x1 = x toward zero in f16, x2 = x - x1: exact, |x2| < 2^-10 |x|),
//     h * 2^16 = h1 + h2 + r (rounded to nearest: |h2| <= 2^-12 |h1|, |r| <= 2^-24 |h1|)
// so that x1h1 + x1h2 + x2h1 leaves out only x2h2 < 2^-22 |x||h| and r -- measured against an f64 evaluation 1.0e-7 relative
// RMS (tests/fir_split_model.py; the bf16 form with five products: 1.4e-7).  Eleven significand bits per value instead of eight: three products where bf16 needs five, and a
// window whose largest tap is below 2^-13 of the peak needs only x1h1.  24 MFMAs per tile instead of 36.
__device__ constexpr int kProductsF16[kWindows] = {kFirProductsF16[0], kFirProductsF16[1], kFirProductsF16[2], kFirProductsF16[3], kFirProductsF16[4],
                                                     kFirProductsF16[5], kFirProductsF16[6], kFirProductsF16[7], kFirProductsF16[8], kFirProductsF16[9]};
__host__ __device__ constexpr int products_of(int s, bool in16, bool f16) { return f16 ? kProductsF16[s] : products_of(s, in16); }
#ifndef SK_BF_AHEAD
#define SK_BF_AHEAD 3
#endif
// half-chunks of input in flight per wave (2, 3, 4 or 6: a divisor of the 12 steps of the unrolled body).  In round 1 three made
// no difference (profiles/r01_ab_fir.md); with this round's shorter steps the loads of two half-chunks no longer land in time:
// 2 -> 3 is 4 % of the launch (0.381-0.386 -> 0.364-0.370 ms on one box), 4 the same, 6 spills (profiles/r03_ab_fir.md).
constexpr int kAhead = SK_BF_AHEAD;

__host__ __device__ constexpr int ring_index(int rho) { return ((rho % kRing) + kRing) % kRing; }

template <bool PACKED>
__device__ __forceinline__ size_t row_base_offset(const FirArgs &a, uint32_t row) {
    if (!PACKED) return (size_t)row * a.in_stride;
    const uint32_t sh = a.in_ch - 1;  // channels 1 -> 0, 2 -> 1
    return (size_t)(row >> sh) * a.in_group_stride + (size_t)((row & sh) << 10);
}

template <bool PACKED>
__device__ __forceinline__ size_t time_offset(const FirArgs &a, uint32_t i) {
    return PACKED ? (size_t)(i >> 10) * a.in_block_stride + (size_t)(i & 1023) : (size_t)i;
}

// x = p1 + p2 + p3 exactly, each a bf16 (the top half of an f32 word); two samples per dword, the earlier one low
__device__ __forceinline__ void split_pair(float x0, float x1, uint32_t &p1, uint32_t &p2, uint32_t &p3) {
    const uint32_t u0 = __float_as_uint(x0), u1 = __float_as_uint(x1);
    p1 = __builtin_amdgcn_perm(u1, u0, 0x07060302u);
    const float r0 = x0 - __uint_as_float(u0 & 0xffff0000u), r1 = x1 - __uint_as_float(u1 & 0xffff0000u);
    const uint32_t v0 = __float_as_uint(r0), v1 = __float_as_uint(r1);
    p2 = __builtin_amdgcn_perm(v1, v0, 0x07060302u);
    const float s0 = r0 - __uint_as_float(v0 & 0xffff0000u), s1 = r1 - __uint_as_float(v1 & 0xffff0000u);
    p3 = __builtin_amdgcn_perm(__float_as_uint(s1), __float_as_uint(s0), 0x07060302u);
}

// two s16 samples in one dword (the earlier one low) -> their two bf16 planes: p1 = 256 * (s >> 8), p2 = s & 255
__device__ __forceinline__ void split_pair16(uint32_t u, uint32_t &p1, uint32_t &p2) {
    const uint32_t m = u & 0xff00ff00u;
    const float a0 = (float)(int)(int16_t)(m & 0xffffu), a1 = (float)((int)m >> 16);
    const float b0 = (float)(u & 0xffu), b1 = (float)((u >> 16) & 0xffu);
    p1 = __builtin_amdgcn_perm(__float_as_uint(a1), __float_as_uint(a0), 0x07060302u);
    p2 = __builtin_amdgcn_perm(__float_as_uint(b1), __float_as_uint(b0), 0x07060302u);
}

// (the f16 planes of s16 samples: dev_split_pair16_f16, sk_device.h -- shared with the fused decode-tail kernel)
__device__ __forceinline__ void split_pair16_f16(uint32_t u, uint32_t &p1, uint32_t &p2) { dev_split_pair16_f16(u, p1, p2); }

template <bool IN16>
struct Stage {
    // ALIGNED: load ld (0..3) of a chunk holds samples 4 (lane & 15) .. + 3 of row 4 ld + (lane >> 4)
    // otherwise: its word e holds sample `lane` of row 4 ld + e
    // a half-chunk = loads 2 half, 2 half + 1 (eight rows); it sits in slot pair v[2 slot], v[2 slot + 1]
    using Elem = std::conditional_t<IN16, int16_t, float>;
    using Vec = std::conditional_t<IN16, u32x2, f32x4>;  // four samples of a row
    const Elem *base[4];
    uint32_t ok_mask;
    Vec v[2 * kAhead];  // kAhead half-chunks (two loads each) in flight
};

template <bool ALIGNED, bool PACKED, bool IN16>
__device__ __forceinline__ void stage_init(const FirArgs &a, int lane, uint32_t row0, Stage<IN16> &st) {
    st.ok_mask = 0;
    if (ALIGNED) {
#pragma unroll
        for (int ld = 0; ld < 4; ++ld) {
            const uint32_t r = row0 + 4 * ld + (lane >> 4);
            const bool ok = r < a.rows;  // rows past the end alias row 0: their results are never stored
            const uint32_t phys = a.row_map ? a.row_map[ok ? r : 0] : (ok ? r : 0);
            if constexpr (IN16) st.base[ld] = a.in16 + row_base_offset<PACKED>(a, phys) + 4 * (lane & 15);
            else st.base[ld] = a.in + row_base_offset<PACKED>(a, phys) + 4 * (lane & 15);
            if (ok) st.ok_mask |= 1u << ld;
        }
    }
}

// issue the global loads of loads [ld0, ld1) of the chunk that starts at stream sample n (wave-uniform)
template <bool ALIGNED, bool PACKED, bool INTERIOR = false, bool IN16 = false>
__device__ __forceinline__ void stage_issue(const FirArgs &a, int lane, uint32_t row0, int64_t n, Stage<IN16> &st, int half, int slot) {
    using Vec = typename Stage<IN16>::Vec;
    using Elem = typename Stage<IN16>::Elem;
    if (ALIGNED) {
        const int64_t c0 = n - a.in_origin;  // the chunk's first sample in the row
        // the usual chunk lies inside the row (and inside one 1024-sample block of the frame-packed layout): one scalar
        // offset for the wave and one 64-bit add per load.  The loads sit between matrix instructions, where every
        // vector instruction beyond two per MFMA costs its full issue time.
        const bool inside = c0 >= 0 && c0 + 64 <= (int64_t)a.in_frames;
        const bool one_block = !PACKED || (((uint32_t)c0 & 1023u) + 64u <= 1024u);
        if (INTERIOR || __builtin_expect(inside && one_block, 1)) {  // INTERIOR: the caller has checked the whole body
            const size_t off = time_offset<PACKED>(a, (uint32_t)c0);
#pragma unroll
            for (int e2 = 0; e2 < 2; ++e2) {
                const int ld = 2 * half + e2, sl = 2 * slot + e2;
                st.v[sl] = *reinterpret_cast<const Vec *>(st.base[ld] + off);
            }
        } else {
            asm volatile("; chunk at an edge of the row" ::: "memory");  // a real branch: not if-converted into the loads above
            const int64_t idx = c0 + 4 * (lane & 15);
            const bool in_range = (uint64_t)idx + 3 < (uint64_t)a.in_frames;  // idx < 0 wraps far above
            const ptrdiff_t off = (ptrdiff_t)time_offset<PACKED>(a, (uint32_t)idx) - 4 * (lane & 15);
            const uint32_t sel = in_range ? st.ok_mask : 0u;
#pragma unroll
            for (int e2 = 0; e2 < 2; ++e2) {
                const int ld = 2 * half + e2, sl = 2 * slot + e2;
                const Elem *src = ((sel >> ld) & 1u) ? st.base[ld] + off : reinterpret_cast<const Elem *>(a.zeros) + 4 * lane;
                st.v[sl] = *reinterpret_cast<const Vec *>(src);
            }
        }
    } else if constexpr (!IN16) {  // s16 rows are only taken aligned (launch_fir_48k_16k checks)
        const int64_t idx = n + lane - a.in_origin;
        const bool in_range = idx >= 0 && idx < (int64_t)a.in_frames;
        const size_t off = time_offset<PACKED>(a, (uint32_t)idx);
#pragma unroll
        for (int e2 = 0; e2 < 2; ++e2) {
            const int ld = 2 * half + e2, sl = 2 * slot + e2;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const uint32_t r = row0 + 4 * ld + e;
                const bool ok = in_range && r < a.rows;
                const uint32_t phys = a.row_map ? a.row_map[r < a.rows ? r : 0] : r;
                st.v[sl][e] = ok ? a.in[row_base_offset<PACKED>(a, phys) + off] : 0.0f;
            }
        }
    }
}

// split loads [ld0, ld1) and write them to the ring at sample position `ring_at` (of the chunk's first sample)
template <bool ALIGNED, bool IN16, bool F16 = false>
__device__ __forceinline__ void stage_commit(unsigned char *lds, int lane, int ring_at, const Stage<IN16> &st, int half, int slot) {
#pragma unroll
    for (int e2 = 0; e2 < 2; ++e2) {
        const int ld = 2 * half + e2, sl = 2 * slot + e2;
        if constexpr (IN16) {
            uint32_t p1a, p2a, p1b, p2b;
            if constexpr (F16) {
                split_pair16_f16(st.v[sl][0], p1a, p2a);
                split_pair16_f16(st.v[sl][1], p1b, p2b);
            } else {
                split_pair16(st.v[sl][0], p1a, p2a);
                split_pair16(st.v[sl][1], p1b, p2b);
            }
            unsigned char *dst = lds + (4 * ld + (lane >> 4)) * row_bytes(2) + 2 * (ring_at + 4 * (lane & 15));
            *reinterpret_cast<u32x2 *>(dst) = (u32x2){p1a, p1b};
            *reinterpret_cast<u32x2 *>(dst + kPlaneBytes) = (u32x2){p2a, p2b};
        } else if (ALIGNED) {
            uint32_t p1a, p2a, p3a, p1b, p2b, p3b;
            split_pair(st.v[sl][0], st.v[sl][1], p1a, p2a, p3a);
            split_pair(st.v[sl][2], st.v[sl][3], p1b, p2b, p3b);
            unsigned char *dst = lds + (4 * ld + (lane >> 4)) * row_bytes(3) + 2 * (ring_at + 4 * (lane & 15));
            *reinterpret_cast<u32x2 *>(dst) = (u32x2){p1a, p1b};
            *reinterpret_cast<u32x2 *>(dst + kPlaneBytes) = (u32x2){p2a, p2b};
            *reinterpret_cast<u32x2 *>(dst + 2 * kPlaneBytes) = (u32x2){p3a, p3b};
        } else {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                uint32_t p1, p2, p3;
                split_pair(st.v[sl][e], 0.0f, p1, p2, p3);
                unsigned char *dst = lds + (4 * ld + e) * row_bytes(3) + 2 * (ring_at + lane);
                *reinterpret_cast<uint16_t *>(dst) = (uint16_t)p1;
                *reinterpret_cast<uint16_t *>(dst + kPlaneBytes) = (uint16_t)p2;
                *reinterpret_cast<uint16_t *>(dst + 2 * kPlaneBytes) = (uint16_t)p3;
            }
        }
    }
}

struct BFrag {
    u32x4 p[3];
};

__device__ __forceinline__ f32x4 mfma_bf16(const u32x4 &av, const u32x4 &bv, const f32x4 &c) {
    return dev_mfma_bf16(av, bv, c);
}

__device__ __forceinline__ f32x4 mfma_f16(const u32x4 &av, const u32x4 &bv, const f32x4 &c) { return dev_mfma_f16(av, bv, c); }

// OUT16: 0 = f32 rows out.  F16 (with IN16): f16 planes and the f16 matrix instruction instead of bf16.
template <bool ALIGNED, bool PACKED, int OUT16, bool IN16 = false, bool F16 = false>
// the f16 form holds two planes of tap fragments (80 registers instead of 120): three waves per SIMD fit (168 VGPRs, a dozen
// words spilled), measured 4 % faster than two with the grid cut into 3072 waves
#ifndef SK_FIR_F16_WAVES
#define SK_FIR_F16_WAVES 3
#endif
__global__ __launch_bounds__(64, F16 ? SK_FIR_F16_WAVES : 2) void k_fir_48k_16k_bf16(FirArgs a, int32_t pair0, int32_t pair_end, int32_t pairs_per_seg,
                                                            uint32_t n_segs, int out_vec) {
    static_assert(!IN16 || ALIGNED, "s16 rows are read four samples at a time");
    static_assert(!F16 || IN16, "the f16 planes exist for s16 rows");
    constexpr int kPlanes = IN16 ? 2 : 3;
    __shared__ __attribute__((aligned(16))) unsigned char lds[16 * row_bytes(kPlanes)];

    const int lane = threadIdx.x;
    const int j = lane & 15, q = lane >> 4;
    const uint32_t group = blockIdx.x / n_segs, seg = blockIdx.x % n_segs;
    const uint32_t row0 = group * 16;
    // tile pair u = absolute outputs 32 u .. 32 u + 31 (even tile 2u, odd tile 2u + 1); segments start at multiples of 4
    const int32_t p_begin = pair0 + (int32_t)seg * pairs_per_seg;
    const int32_t p_end = p_begin + pairs_per_seg < pair_end ? p_begin + pairs_per_seg : pair_end;
    if (p_begin >= p_end) return;
    const int32_t p_last = p_end + 3;  // a pair started in period P completes in period P + 3

    u32x4 af[kWindows][3];
    {
        const u32x4 *src = reinterpret_cast<const u32x4 *>(F16 ? a.afrag_f16 : a.afrag16);
        constexpr int kTapPlanes = F16 ? 2 : 3;
#pragma unroll
        for (int s = 0; s < kWindows; ++s)
#pragma unroll
            for (int k = 0; k < kTapPlanes; ++k) af[s][k] = src[(s * kTapPlanes + k) * 64 + lane];
    }

    Stage<IN16> st;
    stage_init<ALIGNED, PACKED, IN16>(a, lane, row0, st);

    // body-relative time rho = n - 96 P0 (P0 = first period of an unrolled body of four): the ring index of rho is
    // static because 4 periods = 384 samples = 2 ring revolutions
    const int64_t n_begin = (int64_t)96 * p_begin;
    // the ring starts with rho in [-128, 0) in place and the half-chunks that the first kAhead steps commit in flight
#pragma unroll
    for (int c = -2; c < 0; ++c) {
        stage_issue<ALIGNED, PACKED, false, IN16>(a, lane, row0, n_begin + 64 * c, st, 0, 0);
        stage_issue<ALIGNED, PACKED, false, IN16>(a, lane, row0, n_begin + 64 * c, st, 1, 1);
        stage_commit<ALIGNED, IN16, F16>(lds, lane, ring_index(64 * c), st, 0, 0);
        stage_commit<ALIGNED, IN16, F16>(lds, lane, ring_index(64 * c), st, 1, 1);
    }
#pragma unroll
    for (int i = 0; i < kAhead; ++i) stage_issue<ALIGNED, PACKED, false, IN16>(a, lane, row0, n_begin + 64 * (i >> 1), st, i & 1, i % kAhead);

    const uint32_t out_row = row0 + j;
    float *out_ptr = OUT16 ? nullptr
                           : a.out + (size_t)(out_row < a.rows ? out_row : 0) * a.out_stride +
                                 ((a.out_off && out_row < a.rows) ? a.out_off[out_row] : 0);

    // the odd grid's window at ring position 176 wraps: its last 16 samples (lanes q >= 2) sit at the start of the row
    const unsigned char *b_base = lds + j * row_bytes(kPlanes) + 16 * q;
    const unsigned char *b_base_wrap = b_base - (q >= 2 ? 2 * kRing : 0);
    auto read_b = [&](int rho) __attribute__((always_inline)) {
        BFrag f;
        const unsigned char *src = (ring_index(rho) + 32 > kRing ? b_base_wrap : b_base) + 2 * ring_index(rho);
#pragma unroll
        for (int p = 0; p < kPlanes; ++p) f.p[p] = *reinterpret_cast<const u32x4 *>(src + p * kPlaneBytes);
        return f;
    };

    const bool row_exists = out_row < a.rows;
    constexpr bool to_s16 = OUT16 != 0;
    float *out_lane = to_s16 ? nullptr : out_ptr + 4 * q;
    // s16: the lane of a stream's first channel writes the interleaved frames of all its channels
    int16_t *out16_ptr = to_s16 ? a.out16 + (size_t)((row_exists ? out_row : 0) / (OUT16 ? OUT16 : 1)) * a.out16_stride * OUT16 : nullptr;
    auto store_tile = [&](auto itag, const f32x4 &vraw, int32_t pair, int parity) __attribute__((always_inline)) {
        // the samples went in as integers (and the f16 taps times 2^16): powers of two, they commute with every rounding
        const f32x4 v = F16 ? vraw * (1.0f / 2147483648.0f) : (IN16 ? vraw * (1.0f / 32768.0f) : vraw);
        // INTERIOR: every tile of the body lies inside the segment and the output range, all 16 rows exist, stores are
        // vector stores -- no branch, so a whole body is one scheduling region
        constexpr bool INTERIOR = decltype(itag)::value;
        if (!INTERIOR && (pair < p_begin || pair >= p_end)) return;  // wave-uniform
        const int32_t rel_tile = 32 * pair + 16 * parity - (int32_t)a.out_first;  // wave-uniform
        const bool whole = INTERIOR || (out_vec && rel_tile >= 0 && rel_tile + 16 <= (int32_t)a.out_count);
        const bool row_ok = INTERIOR || row_exists;
        const int32_t rel = rel_tile + 4 * q;
        if (!to_s16) {
            if (INTERIOR || __builtin_expect(whole, 1)) {
                if (row_ok) *reinterpret_cast<f32x4 *>(out_lane + rel_tile) = v;
            } else {
                asm volatile("; tile at an edge of the output, or unaligned rows" ::: "memory");
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if (row_ok && rel + r >= 0 && rel + r < (int32_t)a.out_count) out_ptr[rel + r] = v[r];
            }
            return;
        }
        // the worker's 16-bit output stage (float_sample_to_i16, interleave) on the four results a lane holds; the
        // conversions are ordinary vector work in the shadow of the matrix instructions
        // shortest exact form (sk_device.h); for s16 rows the power of two rides in the conversion's constants
        uint32_t mine01, mine23;
        if constexpr (IN16) {
            mine01 = dev_pack2_s16_scaled<F16 ? 31 : 15>(vraw[0], vraw[1]);
            mine23 = dev_pack2_s16_scaled<F16 ? 31 : 15>(vraw[2], vraw[3]);
        } else {
            mine01 = dev_pack2_s16(v[0], v[1]);
            mine23 = dev_pack2_s16(v[2], v[3]);
        }
        if (OUT16 == 2) {
            // the neighbouring row's lane (lane ^ 1) holds the other channel: two packed dwords cross by DPP quad_perm [1,0,3,2]
            const uint32_t other01 = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)mine01, 0xB1, 0xF, 0xF, true);
            const uint32_t other23 = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)mine23, 0xB1, 0xF, 0xF, true);
            uint32_t w[4];  // frame rel + r: L in the low half, R in the high half
            w[0] = (mine01 & 0xffffu) | (other01 << 16);
            w[1] = (mine01 >> 16) | (other01 & 0xffff0000u);
            w[2] = (mine23 & 0xffffu) | (other23 << 16);
            w[3] = (mine23 >> 16) | (other23 & 0xffff0000u);
            if ((j & 1) == 0 && row_ok) {
                uint32_t *dst = reinterpret_cast<uint32_t *>(out16_ptr) + rel;
                if (INTERIOR || __builtin_expect(whole, 1)) {
                    *reinterpret_cast<u32x4 *>(dst) = (u32x4){w[0], w[1], w[2], w[3]};
                } else {
                    asm volatile("; tile at an edge of the output, or unaligned rows" ::: "memory");
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        if (rel + r >= 0 && rel + r < (int32_t)a.out_count) dst[r] = w[r];
                }
            }
        } else if (row_ok) {
            int16_t *dst = out16_ptr + rel;
            if (INTERIOR || __builtin_expect(whole, 1)) {
                *reinterpret_cast<u32x2 *>(dst) = (u32x2){mine01, mine23};
            } else {
                asm volatile("; tile at an edge of the output, or unaligned rows" ::: "memory");
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if (rel + r >= 0 && rel + r < (int32_t)a.out_count) dst[r] = (int16_t)((r < 2 ? mine01 : mine23) >> (16 * (r & 1)));
            }
        }
    };

    f32x4 acc[2][4];
#pragma unroll
    for (int par = 0; par < 2; ++par)
#pragma unroll
        for (int b = 0; b < 4; ++b) acc[par][b] = (f32x4){0.f, 0.f, 0.f, 0.f};

    // window of half-step (period K of the body, wi, parity): even tiles read [t - 128, t - 96), odd tiles [t - 80, t - 48)
    // with t = 96 K + 32 wi
    BFrag bcur = read_b(-128);

    auto period = [&](auto ktag, auto itag, int32_t P, int64_t n_body) __attribute__((always_inline)) {
        constexpr int K = decltype(ktag)::value;
        constexpr bool INTERIOR = decltype(itag)::value;
#pragma unroll
        for (int wi = 0; wi < 3; ++wi) {
            const int step = 3 * K + wi;  // 0..11 within the body, t = 32 step
            // staging: chunk c = step / 2 covers rho in [64 c, 64 c + 64); its first eight rows are split and written in
            // the even step, the other eight in the odd step.  Loads run kAhead steps ahead of that.
            {
                // the scheduler works on one step at a time: across a whole branch-free body it hoists loads and LDS reads
                // far enough to spill
                __builtin_amdgcn_sched_barrier(0);
                stage_commit<ALIGNED, IN16, F16>(lds, lane, ring_index(64 * (step >> 1)), st, step & 1, step % kAhead);
                stage_issue<ALIGNED, PACKED, INTERIOR, IN16>(a, lane, row0, n_body + 64 * ((step + kAhead) >> 1), st,
                                                             (step + kAhead) & 1, step % kAhead);
            }
#pragma unroll
            for (int par = 0; par < 2; ++par) {
                const int t = 32 * step;
                // next half-step's window
                const int rho_next = par == 0 ? t - 80 : t + 32 - 128;
                const BFrag bnext = read_b(rho_next);
                // products in an order that never puts two MFMAs on one accumulator back to back
#pragma unroll
                for (int prod = 0; prod < 6; ++prod) {
                    constexpr int hk[6] = {0, 1, 0, 1, 2, 0};  // with f16 planes only the first three: x1h1 | x1h2, x2h1
                    constexpr int xk[6] = {0, 0, 1, 1, 0, 2};
#pragma unroll
                    for (int d = 0; d < 4; ++d) {
                        const int s = 3 * d + wi;
                        if (s >= kWindows || prod >= products_of(s, IN16, F16)) continue;
                        f32x4 &c = acc[par][(K - d + 4) & 3];
                        const f32x4 c0 = (s == 0 && prod == 0) ? (f32x4){0.f, 0.f, 0.f, 0.f} : c;
                        if constexpr (F16) c = mfma_f16(af[s][hk[prod]], bcur.p[xk[prod]], c0);
                        else c = mfma_bf16(af[s][hk[prod]], bcur.p[xk[prod]], c0);
                    }
                }
                // the tile that took its last step (s = 9, wi = 0) one half-step ago is stored now, behind 20 MFMAs
                // both halves of each row's 128-byte line leave together
                if (wi == 1 && par == 0) {
                    store_tile(itag, acc[0][(K + 1) & 3], P - 3, 0);
                    store_tile(itag, acc[1][(K + 1) & 3], P - 3, 1);
                }
                bcur = bnext;
            }
            // a matrix instruction holds the vector issue port for half of its 16 cycles: two ordinary vector
            // instructions per MFMA ride along free, a longer run between two MFMAs stalls the matrix pipe
            if (INTERIOR) {
                int mfmas = 0;
#pragma unroll
                for (int d = 0; d < 4; ++d)
                    if (3 * d + wi < kWindows) mfmas += 2 * products_of(3 * d + wi, IN16, F16);
#pragma unroll
                for (int g = 0; g < mfmas; ++g) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);  // one MFMA
                    __builtin_amdgcn_sched_group_barrier(0x002, 2, 0);  // two VALU
                }
            }
        }
    };

    using K0 = std::integral_constant<int, 0>;
    using K1 = std::integral_constant<int, 1>;
    using K2 = std::integral_constant<int, 2>;
    using K3 = std::integral_constant<int, 3>;
    // chunks of the frame-packed layout never straddle a 1024-sample block when they start at multiples of 64
    const bool chunks_in_blocks = !PACKED || ((uint32_t)(n_begin - a.in_origin) & 63u) == 0;
    // interior body: the chunks it loads (rho in [64 (kAhead / 2) .. 448)) lie inside the row, its eight tiles (pairs
    // P0 - 3 .. P0) inside the segment and the output range, all sixteen rows exist
    auto is_interior = [&](int32_t P0) {
        const int64_t n_body = (int64_t)96 * P0;
        const int64_t c_first = n_body + 64 * (kAhead >> 1) - a.in_origin, c_end = n_body + 64 * ((11 + kAhead) >> 1) + 64 - a.in_origin;
        const int64_t rel_first = (int64_t)32 * (P0 - 3) - (int64_t)a.out_first, rel_end = (int64_t)32 * P0 + 32 - (int64_t)a.out_first;
        return ALIGNED && chunks_in_blocks && c_first >= 0 && c_end <= (int64_t)a.in_frames && P0 - 3 >= p_begin && P0 < p_end &&
               out_vec && rel_first >= 0 && rel_end <= (int64_t)a.out_count && row0 + 16 <= a.rows;
    };
    auto general_body = [&](int32_t P0) __attribute__((always_inline)) {
        const int64_t n_body = (int64_t)96 * P0;
        if (P0 + 0 < p_last) period(K0{}, std::false_type{}, P0 + 0, n_body);
        if (P0 + 1 < p_last) period(K1{}, std::false_type{}, P0 + 1, n_body);
        if (P0 + 2 < p_last) period(K2{}, std::false_type{}, P0 + 2, n_body);
        if (P0 + 3 < p_last) period(K3{}, std::false_type{}, P0 + 3, n_body);
    };
    // three loops rather than a test inside one: with both bodies in one loop the register allocator spills
    int32_t P0 = p_begin;
    for (; P0 < p_last && !is_interior(P0); P0 += 4) general_body(P0);
    for (; P0 < p_last && is_interior(P0); P0 += 4) {
        const int64_t n_body = (int64_t)96 * P0;
        period(K0{}, std::true_type{}, P0 + 0, n_body);
        period(K1{}, std::true_type{}, P0 + 1, n_body);
        period(K2{}, std::true_type{}, P0 + 2, n_body);
        period(K3{}, std::true_type{}, P0 + 3, n_body);
    }
    for (; P0 < p_last; P0 += 4) general_body(P0);
}

}  // namespace

static bool fir_bf16_supported(const FirArgs &a) {
    // tile and sample indices are 32-bit inside the kernel; longer rows (> ~2^31 / 3 outputs) stay on the f32 kernel
    const int64_t end_pair = ((int64_t)a.out_first + a.out_count + 31) / 32;
    return a.afrag16 != nullptr && end_pair <= 0x7ffffff0ll / 96 && a.out_count <= 0x7fffff00u;
}

hipError_t launch_fir_48k_16k(const FirArgs &a, hipStream_t s) {
    if (a.rows == 0 || a.out_count == 0) return hipSuccess;
    if (!fir_bf16_supported(a)) return hipErrorInvalidValue;
    const uint32_t groups = (a.rows + 15) / 16;
    const int64_t first_pair = ((int64_t)a.out_first / 32) & ~(int64_t)3;
    const int64_t end_pair = ((int64_t)a.out_first + a.out_count + 31) / 32;
    if (end_pair > 0x7ffffff0ll / 96 || a.out_count > 0x7fffff00u) return hipErrorInvalidValue;
    const uint32_t pairs = (uint32_t)(end_pair - first_pair);
    // two waves per SIMD across the chip is what the register count allows; split the time axis until there are about
    // that many waves, keeping segments >= 16 pairs (three periods of every segment only fill the pipeline)
    static const bool f16_rows = [] { const char *v = std::getenv("SK_FIR_S16_BF16"); return !(v && v[0] == '1'); }();
    const uint32_t waves_target = (a.in16 && a.afrag_f16 && f16_rows) ? 1024u * SK_FIR_F16_WAVES : 2048u;
    uint32_t n_segs = (waves_target + groups - 1) / groups;
    const uint32_t max_segs = pairs / 16 ? pairs / 16 : 1;
    if (n_segs > max_segs) n_segs = max_segs;
    uint32_t pps = (pairs + n_segs - 1) / n_segs;
    pps = (pps + 3) & ~3u;
    n_segs = (pairs + pps - 1) / pps;

    const bool strides_ok = a.in_block ? (a.in_block % 4 == 0 && a.in_block_stride % 4 == 0 && a.in_group_stride % 4 == 0)
                                       : (a.in_stride % 4 == 0);
    const bool aligned = ((a.in_origin & 3) == 0) && strides_ok && (a.in_frames % 4 == 0) &&
                         (a.in16 ? (((uintptr_t)a.in16 & 7) == 0) : (((uintptr_t)a.in & 15) == 0));
    if (a.out16 && (a.out16_ch < 1 || a.out16_ch > 2 || a.out_off || a.rows % a.out16_ch)) return hipErrorInvalidValue;
    const int out_vec = a.out16 ? ((a.out_first % 4 == 0) && (a.out16_stride * a.out16_ch) % 8 == 0 && (((uintptr_t)a.out16 & 15) == 0))
                                : ((a.out_first % 4 == 0) && (a.out_stride % 4 == 0) && (((uintptr_t)a.out & 15) == 0) &&
                                   a.out_off == nullptr);
    const bool packed = a.in_block != 0;
    if (packed && (a.in_block != 1024 || a.in_ch < 1 || a.in_ch > 2)) return hipErrorInvalidValue;
    const dim3 grid(groups * n_segs), block(64);
#define SK_FIR_LAUNCH(AL, PK, O16)                                                                                 \
    hipLaunchKernelGGL((k_fir_48k_16k_bf16<AL, PK, O16>), grid, block, 0, s, a, (int32_t)first_pair, (int32_t)end_pair, \
                       (int32_t)pps, n_segs, out_vec)
    if (a.in16) {  // planar s16 from the synthesis kernel: frame-packed, aligned
        if (!packed || !aligned) return hipErrorInvalidValue;
        // f16 planes (24 MFMAs per tile) unless SK_FIR_S16_BF16=1 asks for the bf16 form (36) of the same filter
        const bool use_f16 = f16_rows;
#define SK_FIR_LAUNCH16(O16, F)                                                                                                    \
    hipLaunchKernelGGL((k_fir_48k_16k_bf16<true, true, O16, true, F>), grid, block, 0, s, a, (int32_t)first_pair, (int32_t)end_pair, \
                       (int32_t)pps, n_segs, out_vec)
        if (use_f16 && a.afrag_f16) {
            if (!a.out16) SK_FIR_LAUNCH16(0, true);
            else if (a.out16_ch == 2) SK_FIR_LAUNCH16(2, true);
            else SK_FIR_LAUNCH16(1, true);
        } else {
            if (!a.out16) SK_FIR_LAUNCH16(0, false);
            else if (a.out16_ch == 2) SK_FIR_LAUNCH16(2, false);
            else SK_FIR_LAUNCH16(1, false);
        }
#undef SK_FIR_LAUNCH16
    } else if (a.out16) {  // the fused 16-bit output exists for the frame-packed input of the synthesis kernel
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
