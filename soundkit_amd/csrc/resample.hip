// resample.hip -- generic-ratio windowed-sinc resampler for gfx950: a scalar form that keeps rubato's order of operations (this
// comment and the first half of the file) and the matrix-core form that batches take by default (k_sinc_taps + k_sinc_mfma, below).
//
// Replaces rubato 0.14.1 SincFixedIn<f32>::process (Linear interpolation) as the reference drives it
// for every ratio other than 48k->16k (soundkit/src/audio_pipeline.rs:474-491,
// soundkit-decoder/src/lib.rs:1939-2058): for each output, two 256-tap dot products against the
// sub-filters floor(frac*256) and the next one, blended by the fractional sub-phase:
//     index = floor(idx), sub = floor((idx - index) * 256), frac = idx*256 - floor(idx*256)
//     out   = p0 + frac * (p1 - p0),  p_k = sum_i buf[index_k + i] * sincs[sub_k][i]
// The time index of every output comes from the host (the exact f64 accumulation rubato does), so
// output counts and sub-filter choices match the restated reference bit for bit; the sums keep rubato's
// order (eight running sums per dot product, separate multiplies and adds).
//
// Mapping (round 3; the first form gave every lane its own output of one row, so that the 64 lanes of a wave read 64
// different 1 KiB sub-filters from L2 for every tap: 4.6 G outputs/s, 6 % of the vector peak):
//   the rows of a launch that sit at the same point of the walk ask for the SAME sub-filters at the same output, so a
//   wave takes outputs of 64 rows at a time.  The taps are then wave-uniform: they arrive by scalar loads and enter the
//   multiplies as scalar operands -- no vector-memory or LDS traffic for 512 of the 770 operands of an output.  The
//   inputs of the block's 64 rows sit in LDS transposed ([sample][row], pitch 65: lane = row reads and lane = sample
//   writes are both conflict-free); the sixteen waves of a block share that tile (134 KB: one block per CU, so the block
//   itself has to bring the four waves per SIMD that hide the scalar loads' and the LDS reads' latency) and take its (up to
//   64) outputs two at a time; results leave through a small transposed tile so that every row is written in runs.
//   Three things on top (28.4 ms -> 6.8 -> 4.4 ms at 4096 x 2 rows x 1 s of 44.1 -> 16 kHz): the two dot
//   products of an output share the packed f32 instructions; a wave computes two consecutive outputs in one pass over the
//   samples (dot_two); a workgroup walks several consecutive output blocks and has the next block's input in registers
//   while it works on the current one.  rubato's order of operations survives all three: the results are the oracle's bits
//   (tests/test_fir_gpu.py::test_generic_ratios_are_bit_identical_to_the_restated_rubato).
#include "sk_device.h"

#include <algorithm>

namespace sk {

namespace {

constexpr int kRows = 64;        // rows per workgroup (one per lane)
constexpr int kMaxOuts = 64;     // outputs per workgroup at most (the index sets carry the index of every 32nd output: a block starts at one)
constexpr int kSpanMax = 448;    // input samples staged per row: 64 outputs down to ratio ~1/3 (44.1 -> 16 kHz needs 437), fewer outputs below
constexpr int kPitch = kRows + 1;
constexpr int kWaves = 16;       // waves per workgroup

typedef const __attribute__((address_space(4))) float *const_floats;

// one output of 64 rows: SHIFT = index1 - index0 (1 only when the second sub-filter wraps to the next input sample)
template <int SHIFT>
__device__ __forceinline__ float dot_pair(const float *x /* tile + (index0 - base) * kPitch + lane */, const_floats s0, const_floats s1,
                                          float frac) {
    float acc0[8] = {0, 0, 0, 0, 0, 0, 0, 0}, acc1[8] = {0, 0, 0, 0, 0, 0, 0, 0};  // rubato's 8 running sums
    float carry = x[0];
    for (int i = 0; i < 256; i += 8) {
        float xs[9];
        xs[0] = carry;
#pragma unroll
        for (int j = 1; j < 9; ++j) xs[j] = x[(i + j) * kPitch];
        carry = xs[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            acc0[j] += xs[j] * s0[i + j];
            acc1[j] += xs[j + SHIFT] * s1[i + j];
        }
    }
    const float p0 = acc0[0] + acc0[1] + acc0[2] + acc0[3] + acc0[4] + acc0[5] + acc0[6] + acc0[7];
    const float p1 = acc1[0] + acc1[1] + acc1[2] + acc1[3] + acc1[4] + acc1[5] + acc1[6] + acc1[7];
    return p0 + frac * (p1 - p0);
}

// The same sums with the two dot products of an output on the two halves of the packed f32 instructions: the sample is the
// shared operand (broadcast), the taps of sub-filter `sub` and of `sub + 1` come interleaved from a second copy of the table
// (sincs2[sub][i] = {sincs[sub][i], sincs[sub + 1][i]}), 512 vector instructions per output instead of 1024.  Separate
// multiplies and adds in rubato's order as before: the same bits.  (SHIFT = 1 -- the second sub-filter wrapping to the next
// input sample, one output in 256 -- keeps the scalar form.)
typedef float f2 __attribute__((ext_vector_type(2)));
typedef const __attribute__((address_space(4))) f2 *const_pairs;
__device__ __forceinline__ float dot_pair_packed(const float *x, const_pairs s01, float frac) {
    f2 acc[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[j] = (f2){0.0f, 0.0f};
    for (int i = 0; i < 256; i += 8) {
        float xs[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) xs[j] = x[(i + j) * kPitch];
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[j] += (f2){xs[j], xs[j]} * s01[i + j];
    }
    const float p0 = acc[0].x + acc[1].x + acc[2].x + acc[3].x + acc[4].x + acc[5].x + acc[6].x + acc[7].x;
    const float p1 = acc[0].y + acc[1].y + acc[2].y + acc[3].y + acc[4].y + acc[5].y + acc[6].y + acc[7].y;
    return p0 + frac * (p1 - p0);
}

// Two consecutive outputs of the 64 rows in one pass over the samples: output B's window starts d samples after output A's,
// so sample p is tap p of A and tap p - d of B -- every LDS read feeds four products instead of two (with sixteen waves per
// CU the LDS reads weigh as much as the arithmetic).  B's eight running sums are kept under A's numbering (register j holds
// B's sum (j - d) mod 8) and put back in rubato's order for the final additions; the first d samples belong to A alone, the
// last d to B alone.  sBm = B's tap pairs moved back by d, so that sBm[p] is tap p - d.
template <int D>
__device__ __forceinline__ void sums_in_order(const f2 (&acc)[8], float &p0, float &p1) {
    p0 = acc[D & 7].x + acc[(1 + D) & 7].x + acc[(2 + D) & 7].x + acc[(3 + D) & 7].x + acc[(4 + D) & 7].x + acc[(5 + D) & 7].x + acc[(6 + D) & 7].x +
         acc[(7 + D) & 7].x;
    p1 = acc[D & 7].y + acc[(1 + D) & 7].y + acc[(2 + D) & 7].y + acc[(3 + D) & 7].y + acc[(4 + D) & 7].y + acc[(5 + D) & 7].y + acc[(6 + D) & 7].y +
         acc[(7 + D) & 7].y;
}
__device__ __forceinline__ void dot_two(const float *x, const_pairs sA, const_pairs sBm, int d /* 0..7, wave-uniform */, float frac_a, float frac_b,
                                        float &out_a, float &out_b) {
    f2 acc_a[8], acc_b[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) acc_a[j] = acc_b[j] = (f2){0.0f, 0.0f};
    {
        float xs[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) xs[j] = x[j * kPitch];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            acc_a[j] += (f2){xs[j], xs[j]} * sA[j];
            if (j >= d) acc_b[j] += (f2){xs[j], xs[j]} * sBm[j];
        }
    }
    for (int i = 8; i < 256; i += 8) {
        float xs[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) xs[j] = x[(i + j) * kPitch];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            acc_a[j] += (f2){xs[j], xs[j]} * sA[i + j];
            acc_b[j] += (f2){xs[j], xs[j]} * sBm[i + j];
        }
    }
#pragma unroll
    for (int j = 0; j < 7; ++j)
        if (j < d) acc_b[j] += (f2){x[(256 + j) * kPitch], x[(256 + j) * kPitch]} * sBm[256 + j];
    float p0, p1;
    sums_in_order<0>(acc_a, p0, p1);
    out_a = p0 + frac_a * (p1 - p0);
    switch (d) {
    case 0: sums_in_order<0>(acc_b, p0, p1); break;
    case 1: sums_in_order<1>(acc_b, p0, p1); break;
    case 2: sums_in_order<2>(acc_b, p0, p1); break;
    case 3: sums_in_order<3>(acc_b, p0, p1); break;
    case 4: sums_in_order<4>(acc_b, p0, p1); break;
    case 5: sums_in_order<5>(acc_b, p0, p1); break;
    case 6: sums_in_order<6>(acc_b, p0, p1); break;
    default: sums_in_order<7>(acc_b, p0, p1); break;
    }
    out_b = p0 + frac_b * (p1 - p0);
}

// A workgroup walks `group` consecutive output blocks of its 64 rows.  While the waves work on one block's tile in LDS the
// threads already hold the next block's input window in registers (its global loads were issued before the dot products
// began), so the only time nothing is computed is the two barriers around the LDS refill.  With one 134 KB workgroup per CU
// there is nobody else to cover that latency.
constexpr int kStageRows = kRows / kWaves;               // rows a wave stages: 4
constexpr int kStageCols = (kSpanMax + 1 + 63) / 64;     // samples per lane and row: 7

struct OutputAt {  // where an output sits: first sample, sub-filter, blend weight (rubato's own expressions)
    long index0;
    int sub0, shift;
    float frac;
};
__device__ __forceinline__ OutputAt output_at(double idx) {
    OutputAt r;
    const double fl = floor(idx);
    r.index0 = (long)fl;
    r.sub0 = (int)floor((idx - fl) * 256.0);
    r.shift = r.sub0 + 1 >= 256 ? 1 : 0;
    const double scaled = idx * 256.0;
    r.frac = (float)(scaled - floor(scaled));
    return r;
}

__global__ __launch_bounds__(kWaves * 64) void k_sinc_resample(SincArgs a, uint32_t outs_per_block, uint32_t group) {
    // dynamic LDS (134 KB: above the static limit): [sample][row] input tile with one more sample row for the rolling read's
    // look-ahead, the [output][row] result tile, the outputs' time indices of this block and of the next
    extern __shared__ double lds_raw[];
    double *sidx_buf = lds_raw;  // [2][kMaxOuts]
    float *tile = reinterpret_cast<float *>(lds_raw + 2 * kMaxOuts);
    float *otile = tile + (kSpanMax + 1) * kPitch;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const uint32_t row0 = blockIdx.y * kRows;
    // the rows of a workgroup share their index set (the host groups them)
    const uint32_t set = a.row_set ? a.row_set[row0] : 0;
    const uint32_t count = a.set_count[set];
    const uint32_t m_first = blockIdx.x * group * outs_per_block;
    if (m_first >= count) return;  // block-uniform
    const uint32_t n_blocks = min(group, (count - m_first + outs_per_block - 1) / outs_per_block);

    // rubato's own sequence of additions from the nearest index the host sent (f64 addition is not associative: no shortcut)
    auto walk = [&](uint32_t m0, double *sidx) {
        const uint32_t n_here = min(outs_per_block, count - m0);
        if (threadIdx.x < n_here) {
            const uint32_t m = m0 + threadIdx.x;
            double idx = a.set_starts[(size_t)set * a.starts_stride + (m >> 5)];
            const uint32_t steps = m & 31u;
            for (uint32_t i = 0; i < steps; ++i) idx += a.step;
            sidx[threadIdx.x] = idx;
        }
    };
    // the rows this wave stages
    const float *src[kStageRows];
    bool live[kStageRows];
#pragma unroll
    for (int k = 0; k < kStageRows; ++k) {
        const uint32_t row = row0 + (uint32_t)(wave * kStageRows + k);
        uint32_t phys = 0xffffffffu;
        if (row < a.rows) phys = a.row_map ? a.row_map[row] : row;
        live[k] = phys != 0xffffffffu;
        src[k] = a.in + (size_t)(live[k] ? phys : 0) * a.in_stride;
    }
    // a block's input window, 64 rows x [floor(idx first), floor(idx last) + 257], into registers / from registers into LDS
    float pre[kStageRows][kStageCols];
    auto fetch = [&](const double *sidx, uint32_t n_here) {
        const long base = (long)floor(sidx[0]);
        const int need = (int)((long)floor(sidx[n_here - 1]) + 258 - base);  // <= kSpanMax by the host's choice of outs_per_block
#pragma unroll
        for (int k = 0; k < kStageRows; ++k)
#pragma unroll
            for (int cc = 0; cc < kStageCols; ++cc) {
                const int c = lane + 64 * cc;
                const long n = base + c - a.in_origin;  // element of the row
                pre[k][cc] = (live[k] && c <= need && n >= 0 && n < (long)a.in_frames) ? src[k][n] : 0.0f;
            }
    };
    auto refill = [&]() {
#pragma unroll
        for (int k = 0; k < kStageRows; ++k)
#pragma unroll
            for (int cc = 0; cc < kStageCols; ++cc) {
                const int c = lane + 64 * cc;
                if (c <= kSpanMax) tile[c * kPitch + wave * kStageRows + k] = pre[k][cc];
            }
    };

    walk(m_first, sidx_buf);
    __syncthreads();
    fetch(sidx_buf, min(outs_per_block, count - m_first));
    refill();
    if (n_blocks > 1) walk(m_first + outs_per_block, sidx_buf + kMaxOuts);
    __syncthreads();

    const const_floats sincs = reinterpret_cast<const_floats>(reinterpret_cast<uintptr_t>(a.sincs));
    const const_pairs pairs = reinterpret_cast<const_pairs>(reinterpret_cast<uintptr_t>(a.sincs + 65536));
    for (uint32_t blk = 0; blk < n_blocks; ++blk) {
        const uint32_t m0 = m_first + blk * outs_per_block;
        const uint32_t n_here = min(outs_per_block, count - m0);
        const double *sidx = sidx_buf + (blk & 1u) * kMaxOuts;
        const double *sidx_next = sidx_buf + ((blk + 1) & 1u) * kMaxOuts;
        const bool more = blk + 1 < n_blocks;
        if (more) fetch(sidx_next, min(outs_per_block, count - (m0 + outs_per_block)));  // in flight during the dot products
        const long base = (long)floor(sidx[0]);

        for (uint32_t o = 2u * (uint32_t)wave; o < n_here; o += 2 * kWaves) {  // wave-uniform: outputs o and o + 1
            const OutputAt at = output_at(sidx[o]);
            const int off = __builtin_amdgcn_readfirstlane((int)(at.index0 - base));
            const int sub_a = __builtin_amdgcn_readfirstlane(at.sub0);
            const float *x = tile + off * kPitch + lane;
            bool paired = false;
            if (o + 1 < n_here) {
                const OutputAt bt = output_at(sidx[o + 1]);
                const int d = __builtin_amdgcn_readfirstlane((int)(bt.index0 - at.index0));
                const int sub_b = __builtin_amdgcn_readfirstlane(bt.sub0);
                if (!__builtin_amdgcn_readfirstlane(at.shift | bt.shift) && d >= 0 && d <= 7) {
                    float va, vb;
                    dot_two(x, pairs + sub_a * 256, pairs + sub_b * 256 - d, d, at.frac, bt.frac, va, vb);
                    otile[o * kPitch + lane] = va;
                    otile[(o + 1) * kPitch + lane] = vb;
                    paired = true;
                } else {
                    const int off_b = __builtin_amdgcn_readfirstlane((int)(bt.index0 - base));
                    const float *xb = tile + off_b * kPitch + lane;
                    const int sub_b1 = (sub_b + 1) & 255;
                    otile[(o + 1) * kPitch + lane] = __builtin_amdgcn_readfirstlane(bt.shift)
                                                         ? dot_pair<1>(xb, sincs + sub_b * 256, sincs + sub_b1 * 256, bt.frac)
                                                         : dot_pair_packed(xb, pairs + sub_b * 256, bt.frac);
                }
            }
            if (!paired) {
                const int sub_a1 = (sub_a + 1) & 255;
                otile[o * kPitch + lane] = __builtin_amdgcn_readfirstlane(at.shift) ? dot_pair<1>(x, sincs + sub_a * 256, sincs + sub_a1 * 256, at.frac)
                                                                                    : dot_pair_packed(x, pairs + sub_a * 256, at.frac);
            }
        }
        __syncthreads();  // the tile has been read, the results are in otile

        // rows leave in runs of n_here consecutive outputs: lane = output, 16 rows per pass of the block
        for (int r = (int)(threadIdx.x >> 6); r < kRows; r += kWaves) {
            const uint32_t row = row0 + (uint32_t)r, o = threadIdx.x & 63u;
            if (row >= a.rows || o >= n_here) continue;
            if (a.row_map && a.row_map[row] == 0xffffffffu) continue;  // a padding row of the host's grouping
            float *dst = a.out + (size_t)row * a.out_stride + (a.out_off ? a.out_off[row] : 0);
            dst[m0 + o] = otile[o * kPitch + r];
        }
        if (more) {
            refill();
            if (blk + 2 < n_blocks) walk(m0 + 2 * outs_per_block, sidx_buf + (blk & 1u) * kMaxOuts);  // this block's slot is free now
        }
        __syncthreads();
    }
}

// ---- the same resampler on the matrix cores -------------------------------------------------------------------------------------
// An output is a 256-tap (257 where the second sub-filter wraps) dot product of its row's samples with ONE effective filter,
//     g[p] = (1 - frac) sincs[sub][p] + frac sincs[sub + 1][p - shift],
// and the filter depends on the output's place in the walk only -- not on the row.  So per 16 consecutive outputs (a tile)
// the sixteen filters, placed at their offsets from the tile's first sample, are the A operand of the FIR's matrix product
// (fir_bf16.hip):  D[i][j] += A_s[i][k] B_s[k][j],  i = output of the tile, j = one of 16 rows, k = 32 samples of window s,
//     A_s[i][k] = g_i[32 s + k - (index0_i - base)],   B_s[k][j] = x_j[base + 32 s + k],
// both split exactly into three bf16 planes and multiplied as the six products  x1h1, x1h2, x2h1, x2h2, x1h3, x3h1  (what is
// left out is below 2^-24 |x||h|: one f32 rounding; 2.2e-7 relative RMS against f64 on the 48 -> 16 kHz path that uses the same
// products).  k_sinc_taps builds the A fragments of every tile of every index set once per launch (they are shared by all
// rows of the set: thousands of them in a batch); k_sinc_mfma stages 32 rows x the span of eight tiles as bf16 planes in LDS,
// one wave per tile, two row groups per wave.  This form does NOT keep rubato's order of operations (the scalar form above
// does, bit for bit, and stays available: SincArgs::exact); it is held to the float tolerance like the 48 -> 16 kHz FIR.
#ifndef SK_MFMA_ROWS
#define SK_MFMA_ROWS 32
#define SK_MFMA_TILES 8
#define SK_MFMA_SPAN 704
#define SK_MFMA_BLOCKS 1
#endif
#ifndef SK_MFMA_PITCH_PAD
#define SK_MFMA_PITCH_PAD 32
#endif
#ifndef SK_MFMA_PREFETCH_B
#define SK_MFMA_PREFETCH_B 1
#endif
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

constexpr int kMfmaRows = SK_MFMA_ROWS;           // rows per workgroup: row groups of 16
constexpr int kMfmaTiles = SK_MFMA_TILES;         // tiles (of 16 outputs) per workgroup at most: one per wave
constexpr int kMfmaSpan = SK_MFMA_SPAN;           // samples staged per row
// bytes per row of a plane: whole 128-sample groups + 32 = 1568 = 16 * 98.  The B operand is one ds_read_b128 per plane, lane 16 q + j on
// row j at slot q + const; a wave's read is served in four groups of sixteen lanes that are not its quarters ({0-3, 12-15, 20-27},
// {4-11, 16-19, 28-31} and the same 32 lanes up: fir_bf16.hip, MI355X_MICROARCH.md), so the rows' slots  j * pitch + [4 <= j < 12]  must
// differ modulo 16: a pitch of 2 or 6 modulo 8 slots.  (97 slots, "an odd multiple", put 40 % of the LDS cycles into bank conflicts:
// profiles/r03_resample_pmc.json.)
constexpr int kMfmaPitch = 2 * (128 * ((kMfmaSpan + 127) / 128)) + SK_MFMA_PITCH_PAD;
constexpr int kMfmaPlane = kMfmaRows * kMfmaPitch;
static_assert(kMfmaPitch % 16 == 0 && kMfmaRows % 16 == 0 && kMfmaRows % kMfmaTiles == 0, "row pitch: whole 16-byte slots");
constexpr int kMaxWindows = 12;                  // windows of a tile held in registers: steps up to ~6.9 (96 -> 16 kHz: 6); beyond: the scalar form

struct TileMeta {
    int32_t base;      // first sample of window 0 (a multiple of 8), in the index time base
    int32_t windows;   // 32-sample windows of this tile; 0: no outputs
};

// x = p1 + p2 + p3 exactly, each a bf16 (the top half of an f32 word); two values per dword, the earlier one low
__device__ __forceinline__ void split3(float x0, float x1, uint32_t &p1, uint32_t &p2, uint32_t &p3) {
    const uint32_t u0 = __float_as_uint(x0), u1 = __float_as_uint(x1);
    p1 = __builtin_amdgcn_perm(u1, u0, 0x07060302u);
    const float r0 = x0 - __uint_as_float(u0 & 0xffff0000u), r1 = x1 - __uint_as_float(u1 & 0xffff0000u);
    const uint32_t v0 = __float_as_uint(r0), v1 = __float_as_uint(r1);
    p2 = __builtin_amdgcn_perm(v1, v0, 0x07060302u);
    const float s0 = r0 - __uint_as_float(v0 & 0xffff0000u), s1 = r1 - __uint_as_float(v1 & 0xffff0000u);
    p3 = __builtin_amdgcn_perm(__float_as_uint(s1), __float_as_uint(s0), 0x07060302u);
}

// grid (tiles, sets), one wave: the A fragments [window][plane][lane] of tile t of index set `set`, and its TileMeta
__global__ __launch_bounds__(64) void k_sinc_taps(SincArgs a, uint32_t n_tiles, uint32_t tile_first, uint32_t max_windows, u32x4 *frags, TileMeta *meta) {
    const int lane = threadIdx.x, i = lane & 15, q = lane >> 4;
    const uint32_t t = tile_first + blockIdx.x, set = blockIdx.y;  // t: tile of the row; storage is per pass (n_tiles of them)
    const uint32_t count = a.set_count[set];
    TileMeta *my_meta = meta + (size_t)set * n_tiles + blockIdx.x;
    if (16u * t >= count) {
        if (lane == 0) *my_meta = TileMeta{0, 0};
        return;
    }
    const uint32_t last = min(15u, count - 1u - 16u * t);  // the tile's last output
    const uint32_t m = 16u * t + min((uint32_t)i, last);
    double idx = a.set_starts[(size_t)set * a.starts_stride + (m >> 5)];
    for (uint32_t k = 0; k < (m & 31u); ++k) idx += a.step;  // rubato's own additions
    const OutputAt at = output_at(idx);
    const int index0 = (int)at.index0;
    const int base = (__shfl(index0, 0) >> 3) << 3;  // floor to a multiple of 8: 16-byte aligned reads of the bf16 planes
    const int windows = (__shfl(index0, (int)last) - base + 257 + 31) / 32;
    if (lane == 0) *my_meta = TileMeta{base, windows <= (int)max_windows ? windows : 0};
    if (windows > (int)max_windows) return;  // (the host sized max_windows for this step: not reached)
    const bool valid = (uint32_t)i <= last;
    const float *s0 = a.sincs + (size_t)at.sub0 * 256, *s1 = a.sincs + (size_t)((at.sub0 + 1) & 255) * 256;
    const int delta = index0 - base;
    u32x4 *out = frags + ((size_t)set * n_tiles + blockIdx.x) * max_windows * 3 * 64 + lane;
    for (int s = 0; s < windows; ++s) {
        float g[8];
#pragma unroll
        for (int kk = 0; kk < 8; ++kk) {
            const int p = 32 * s + 8 * q + kk - delta, pb = p - at.shift;
            const float ta = (valid && p >= 0 && p < 256) ? s0[p] : 0.0f;
            const float tb = (valid && pb >= 0 && pb < 256) ? s1[pb] : 0.0f;
            g[kk] = ta + at.frac * (tb - ta);
        }
        u32x4 pl[3];
#pragma unroll
        for (int h = 0; h < 4; ++h) {
            uint32_t p1, p2, p3;
            split3(g[2 * h], g[2 * h + 1], p1, p2, p3);
            pl[0][h] = p1, pl[1][h] = p2, pl[2][h] = p3;
        }
#pragma unroll
        for (int k = 0; k < 3; ++k) out[(size_t)(s * 3 + k) * 64] = pl[k];
    }
}

// A workgroup = eight tiles (one per wave) x a run of consecutive 32-row blocks.  The waves keep their tiles' fragments in
// registers for the whole run (rows of one index set share them; the host orders rows by set, so a reload is rare) and hold
// the NEXT row block's samples in registers while the matrix instructions work on the current one out of LDS -- with 149 KB of
// LDS there is one workgroup per CU and nobody else to cover a trip to memory (one block per launch of 12.7 us, 13 % of the
// matrix pipe: profiles/r03_resample_pmc.json, the form before this loop).
struct RowBlock {
    uint32_t set, count, nt;
    const TileMeta *tm;
    int base_first, span;
    bool live;  // this tile chunk has outputs for the block's index set
};

__global__ __launch_bounds__(kMfmaTiles * 64, SK_MFMA_BLOCKS) void k_sinc_mfma(SincArgs a, uint32_t n_tiles, uint32_t tile_first, uint32_t max_windows,
                                                               uint32_t tiles_per_block, uint32_t row_blocks_per_group, const u32x4 *frags,
                                                               const TileMeta *meta) {
    extern __shared__ __attribute__((aligned(16))) unsigned char planes[];  // [3][kMfmaRows][kMfmaPitch]
    const int lane = threadIdx.x & 63, j = lane & 15, q = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const uint32_t n_row_blocks = (a.rows + kMfmaRows - 1) / kMfmaRows;
    const uint32_t rb_begin = blockIdx.x * row_blocks_per_group;
    const uint32_t rb_end = min(n_row_blocks, rb_begin + row_blocks_per_group);
    const uint32_t l0 = blockIdx.y * tiles_per_block;  // first tile of the workgroup within this pass (storage index)
    const uint32_t t0 = tile_first + l0;               // ... and within the row
    const uint32_t t = t0 + (uint32_t)wave;

    // the index sets of the run's row blocks, read once up front (eight independent loads instead of one dependent chain per
    // block); a block of the set its predecessor had takes over its description without touching memory again
    constexpr uint32_t kMaxRun = 8;
    uint32_t set_of[kMaxRun];
#pragma unroll
    for (uint32_t k = 0; k < kMaxRun; ++k)
        set_of[k] = (a.row_set && rb_begin + k < rb_end) ? a.row_set[(rb_begin + k) * kMfmaRows] : 0;  // 64 consecutive rows share their set: so do 32
    auto describe = [&](uint32_t rb, uint32_t set) {
        RowBlock b;
        b.set = set;
        b.count = a.set_count[b.set];
        b.tm = meta + (size_t)b.set * n_tiles + l0;
        b.live = 16u * t0 < b.count;
        b.nt = 0, b.base_first = 0, b.span = 0;
        if (b.live) {
            b.nt = min(min(tiles_per_block, n_tiles - l0), (b.count - 16u * t0 + 15u) / 16u);
            b.base_first = b.tm[0].base;
            for (uint32_t k = 0; k < b.nt; ++k) b.span = max(b.span, b.tm[k].base + 32 * b.tm[k].windows - b.base_first);
            if (b.span > kMfmaSpan || b.tm[0].windows == 0) b.live = false;  // (the host sized the launch for this step: not reached)
        }
        return b;
    };

    // the wave's tap fragments: up to 12 windows x 3 planes in registers (two waves per SIMD: 256 registers each)
    u32x4 h[kMaxWindows][3];
    uint32_t loaded_set = 0xffffffffu;
    bool taps_pending = false, has_tile = false;
    int off = 0, windows = 0;
    auto load_taps = [&](const RowBlock &b) __attribute__((always_inline)) {
        has_tile = (uint32_t)wave < b.nt;
        off = has_tile ? __builtin_amdgcn_readfirstlane(b.tm[has_tile ? wave : 0].base - b.base_first) : 0;  // a multiple of 8
        windows = has_tile ? __builtin_amdgcn_readfirstlane(b.tm[has_tile ? wave : 0].windows) : 0;
        const u32x4 *fa = frags + ((size_t)b.set * n_tiles + l0 + (uint32_t)wave) * max_windows * 3 * 64 + lane;
#pragma unroll
        for (int s = 0; s < kMaxWindows; ++s)
            if (s < windows) {
#pragma unroll
                for (int k = 0; k < 3; ++k)  // issued HERE by hand: an ordinary load is sunk to its use behind the barrier, a volatile one bypasses L2
                    asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(h[s][k]) : "v"(fa + (size_t)(s * 3 + k) * 64) : "memory");
            }
        loaded_set = b.set;
        taps_pending = true;
    };

    // a row block's samples: 32 rows x span from base_first; a thread takes pairs, all of a wave's loads (4 rows x 6 pairs per
    // lane) issued before the first is used; then split into three bf16 planes and written to LDS
    constexpr int kStageR = kMfmaRows / kMfmaTiles, kStageC = (kMfmaSpan / 2 + 63) / 64;
    float pre[kStageR][kStageC][2];
    auto fetch = [&](uint32_t rb, const RowBlock &b) __attribute__((always_inline)) {
        const long n0 = (long)b.base_first - a.in_origin;  // element of the row at the block's first sample
        // the usual block lies inside its rows and all 32 rows exist: no per-sample tests (wave-uniform branch)
        const bool inside = n0 >= 0 && n0 + 2 * 64 * kStageC <= (long)a.in_frames && (rb + 1) * kMfmaRows <= a.rows;
#pragma unroll
        for (int k = 0; k < kStageR; ++k) {
            const uint32_t row = rb * kMfmaRows + (uint32_t)(wave + kMfmaTiles * k);
            uint32_t phys = 0xffffffffu;
            if (row < a.rows) phys = a.row_map ? a.row_map[row] : row;
            const float *src = a.in + (size_t)(phys == 0xffffffffu ? 0 : phys) * a.in_stride;
            if (inside && phys != 0xffffffffu) {
                const float *at = src + n0 + 2 * lane;
#pragma unroll
                for (int cc = 0; cc < kStageC; ++cc) __builtin_memcpy(pre[k][cc], at + 128 * cc, 8);  // (rows are 4-byte aligned only)
                continue;
            }
#pragma unroll
            for (int cc = 0; cc < kStageC; ++cc) {
                const int c2 = lane + 64 * cc;
                const long n = n0 + 2 * c2;
                const bool live = phys != 0xffffffffu && 2 * c2 < b.span;
                pre[k][cc][0] = (live && n >= 0 && n < (long)a.in_frames) ? src[n] : 0.0f;
                pre[k][cc][1] = (live && n + 1 >= 0 && n + 1 < (long)a.in_frames) ? src[n + 1] : 0.0f;
            }
        }
    };
    auto refill = [&](const RowBlock &b) __attribute__((always_inline)) {
#pragma unroll
        for (int k = 0; k < kStageR; ++k) {
            unsigned char *dst = planes + (wave + kMfmaTiles * k) * kMfmaPitch;
#pragma unroll
            for (int cc = 0; cc < kStageC; ++cc) {
                const int c2 = lane + 64 * cc;
                if (128 * cc >= b.span) continue;  // wave-uniform: the row's pitch has room for whole 128-sample groups
                uint32_t p1, p2, p3;
                split3(pre[k][cc][0], pre[k][cc][1], p1, p2, p3);
                *reinterpret_cast<uint32_t *>(dst + 4 * c2) = p1;
                *reinterpret_cast<uint32_t *>(dst + kMfmaPlane + 4 * c2) = p2;
                *reinterpret_cast<uint32_t *>(dst + 2 * kMfmaPlane + 4 * c2) = p3;
            }
        }
    };

    RowBlock cur = describe(rb_begin, set_of[0]);
    if (cur.live) {
        load_taps(cur);
        fetch(rb_begin, cur);
        refill(cur);
    }
    __syncthreads();
    for (uint32_t rb = rb_begin; rb < rb_end; ++rb) {
        const bool more = rb + 1 < rb_end;
        RowBlock nxt = cur;
        if (more) {
            uint32_t set_next = 0;
#pragma unroll
            for (uint32_t k = 1; k < kMaxRun; ++k)
                if (rb + 1 - rb_begin == k) set_next = set_of[k];
            if (set_next != cur.set) nxt = describe(rb + 1, set_next);
            if (nxt.live) fetch(rb + 1, nxt);  // in flight during the matrix instructions below
        }
        if (cur.live && has_tile) {
            if (taps_pending) {  // wave-uniform.  Vector memory returns in order: this also waits for the loads just issued, once per set
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                taps_pending = false;
            }
            // the registers pass through this statement so that nothing that reads them is scheduled in front of the wait
#pragma unroll
            for (int s = 0; s < kMaxWindows; ++s) asm volatile("" : "+v"(h[s][0]), "+v"(h[s][1]), "+v"(h[s][2]));
            constexpr int kGroups = kMfmaRows / 16;
            f32x4 acc[kGroups];
            // where the results go, asked for in front of the matrix instructions (two dependent trips to memory otherwise, after them)
            float *dst_of[kGroups];
#pragma unroll
            for (int rg = 0; rg < kGroups; ++rg) {
                const uint32_t row = rb * kMfmaRows + (uint32_t)(rg * 16 + j);
                const bool there = row < a.rows && !(a.row_map && a.row_map[row] == 0xffffffffu);  // (0xffffffff: a padding row of the host's grouping)
                dst_of[rg] = there ? a.out + (size_t)row * a.out_stride + (a.out_off ? a.out_off[row] : 0) : nullptr;
            }
#pragma unroll
            for (int rg = 0; rg < kGroups; ++rg) acc[rg] = (f32x4){0.f, 0.f, 0.f, 0.f};
            struct BSet {
                u32x4 x[3];
            };
            auto read_b = [&](int s, int rg) __attribute__((always_inline)) {
                BSet b;
                const unsigned char *bp = planes + (rg * 16 + j) * kMfmaPitch + 2 * (off + 32 * s + 8 * q);
#pragma unroll
                for (int k = 0; k < 3; ++k) b.x[k] = *reinterpret_cast<const u32x4 *>(bp + k * kMfmaPlane);
                return b;
            };
            auto products = [&](int s, const BSet &b, f32x4 c) __attribute__((always_inline)) {
                c = dev_mfma_bf16(h[s][0], b.x[0], c);
                c = dev_mfma_bf16(h[s][1], b.x[0], c);
                c = dev_mfma_bf16(h[s][0], b.x[1], c);
                c = dev_mfma_bf16(h[s][1], b.x[1], c);
                c = dev_mfma_bf16(h[s][2], b.x[0], c);
                c = dev_mfma_bf16(h[s][0], b.x[2], c);
                return c;
            };
            static_assert(!SK_MFMA_PREFETCH_B || kGroups == 2, "the read-ahead below alternates between two row groups");
            if constexpr (SK_MFMA_PREFETCH_B) {
                // a row group's samples are on their way from LDS while the matrix instructions work on the other group's: the
                // second group of window s during the first group's six products, the first group of window s + 1 during the
                // second's (past the last window the read repeats it: no branch, never beyond the tile's span)
                BSet b0 = read_b(0, 0);
#pragma unroll
                for (int s = 0; s < kMaxWindows; ++s) {
                    if (s >= windows) continue;  // (wave-uniform; no break: the loop must unroll for h[s] to stay in registers)
                    const BSet b1 = read_b(s, 1);
                    acc[0] = products(s, b0, acc[0]);
                    b0 = read_b(s + 1 < windows ? s + 1 : s, 0);
                    acc[1] = products(s, b1, acc[1]);
                }
            } else {
#pragma unroll
                for (int s = 0; s < kMaxWindows; ++s) {
                    if (s >= windows) continue;
#pragma unroll
                    for (int rg = 0; rg < kGroups; ++rg) acc[rg] = products(s, read_b(s, rg), acc[rg]);
                }
            }
            // D[i][j]: lane (j, q) holds outputs i = 4 q .. 4 q + 3 of row j of each row group
#pragma unroll
            for (int rg = 0; rg < kGroups; ++rg) {
                float *dst = dst_of[rg];
                if (!dst) continue;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const uint32_t mo = 16u * t + 4u * (uint32_t)q + (uint32_t)r;
                    if (mo < cur.count) dst[mo] = acc[rg][r];
                }
            }
        }
        __syncthreads();  // the planes have been read
        if (more && nxt.live) {
            refill(nxt);
            if (nxt.set != loaded_set || !cur.live) load_taps(nxt);
        }
        cur = nxt;
        __syncthreads();
    }
}

__global__ __launch_bounds__(256) void k_row_copies(const float *src_base, float *dst_base, const RowCopy *jobs,
                                                    uint32_t n_jobs) {
    const uint32_t j = blockIdx.y;
    if (j >= n_jobs) return;
    const RowCopy job = jobs[j];
    const float *src = src_base + job.src_off;
    float *dst = dst_base + job.dst_off;
    const uint32_t total = job.count * job.pieces;
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
        const uint32_t piece = job.pieces > 1 ? i / job.count : 0u;  // (block-uniform branch; one job, one `pieces`)
        const float x = src[(size_t)piece * job.src_stride + (i - piece * job.count)];
        // via_s16: float_sample_to_i16 (soundkit-decoder lib.rs:1815-1827), then audio_data_to_f32_channels' / 32768
        dst[i] = job.via_s16 ? (float)dev_float_sample_to_i16_f32(x) / 32768.0f : x;
    }
}

}  // namespace

uint32_t sinc_rows_per_block() { return kRows; }

// windows a tile can need at this step: 16 outputs spread over 15 steps, + up to 7 samples of alignment, + 257 taps
static uint32_t sinc_mfma_windows(double step) { return (uint32_t)((15.0 * step + 1.0 + 7.0 + 257.0 + 31.0) / 32.0); }
static uint32_t sinc_mfma_tiles_per_block(double step) {
    uint32_t tiles = kMfmaTiles;
    // span of `tiles` consecutive tiles: their bases spread over (tiles - 1) * 16 steps, the last one's windows behind it
    while (tiles > 1 && (tiles - 1) * 16.0 * step + 8.0 + 32.0 * sinc_mfma_windows(step) > (double)kMfmaSpan) tiles >>= 1;
    return tiles;
}

// The fragments are built and used in passes of at most kPassTiles tiles (64 K outputs) per index set -- fewer when many index
// sets share the launch, so that the scratch stays within kScratchBudget -- so the scratch grows neither with the length of
// the rows nor without bound with the number of sets, and the FORM never depends on either: it is chosen by the ratio alone.
constexpr uint32_t kPassTiles = 4096;
constexpr size_t kScratchBudget = (size_t)8 << 30;

static bool sinc_mfma_takes(double step) {
    const uint32_t windows = sinc_mfma_windows(step);
    return windows <= (uint32_t)kMaxWindows && 32.0 * windows + 8.0 <= (double)kMfmaSpan;
}
static uint32_t sinc_mfma_pass_tiles(uint32_t n_sets, uint32_t out_count, double step) {
    const size_t per_tile = (size_t)sinc_mfma_windows(step) * 3 * 1024 + sizeof(TileMeta);
    const size_t all = std::min<size_t>((out_count + 15) / 16, kPassTiles);
    const size_t fit = kScratchBudget / (per_tile * (size_t)n_sets);
    return (uint32_t)std::max<size_t>(1, std::min(all, fit));
}

size_t sinc_mfma_scratch_bytes(uint32_t n_sets, uint32_t out_count, double step) {
    if (n_sets == 0 || out_count == 0 || !sinc_mfma_takes(step)) return 0;
    const size_t per_tile = (size_t)sinc_mfma_windows(step) * 3 * 1024 + sizeof(TileMeta);
    return (size_t)n_sets * sinc_mfma_pass_tiles(n_sets, out_count, step) * per_tile + 512;
}

static hipError_t launch_sinc_mfma(const SincArgs &a, hipStream_t s) {
    const uint32_t windows = sinc_mfma_windows(a.step), all_tiles = (a.out_count + 15) / 16, tpb = sinc_mfma_tiles_per_block(a.step);
    const uint32_t pass_tiles = sinc_mfma_pass_tiles(a.n_sets, a.out_count, a.step);
    const size_t frag_bytes = (size_t)a.n_sets * pass_tiles * windows * 3 * 1024;
    u32x4 *frags = reinterpret_cast<u32x4 *>(a.scratch);
    TileMeta *meta = reinterpret_cast<TileMeta *>(reinterpret_cast<unsigned char *>(a.scratch) + ((frag_bytes + 255) & ~(size_t)255));
    const uint32_t row_blocks = (a.rows + kMfmaRows - 1) / kMfmaRows;
    if (a.n_sets > 65535) return hipErrorInvalidValue;
    constexpr size_t lds_bytes = 3 * (size_t)kMfmaPlane;
    static const hipError_t attr = hipFuncSetAttribute(reinterpret_cast<const void *>(&k_sinc_mfma), hipFuncAttributeMaxDynamicSharedMemorySize,
                                                       (int)lds_bytes);
    if (attr != hipSuccess) return attr;
    for (uint32_t tile_first = 0; tile_first < all_tiles; tile_first += pass_tiles) {
        const uint32_t tiles = std::min(pass_tiles, all_tiles - tile_first);
        hipLaunchKernelGGL(k_sinc_taps, dim3(tiles, a.n_sets), dim3(64), 0, s, a, pass_tiles, tile_first, windows, frags, meta);
        // runs of row blocks per workgroup: long enough to amortise the fragments and the pipeline's fill, short enough for >= 4
        // workgroups per CU in the launch
        const uint32_t tile_chunks = (tiles + tpb - 1) / tpb;
        uint32_t run = 8;
        while (run > 1 && (uint64_t)((row_blocks + run - 1) / run) * tile_chunks < 1024) run >>= 1;
        hipLaunchKernelGGL(k_sinc_mfma, dim3((row_blocks + run - 1) / run, tile_chunks), dim3(kMfmaTiles * 64), lds_bytes, s, a, pass_tiles, tile_first, windows,
                           tpb, run, frags, meta);
    }
    return hipGetLastError();
}

hipError_t launch_sinc_resample(const SincArgs &a, hipStream_t s) {
    if (a.rows == 0 || a.out_count == 0) return hipSuccess;
    // The matrix-core form whenever the caller lent scratch for it -- for ONE row as for thousands: a row's samples must not
    // depend on how many other rows shared its launch (tests/test_scheduler_gpu.py compares a stream in a batch of many with
    // the same stream alone, bit for bit), and the two forms differ in the last bits.
    if (!a.exact && a.n_sets && sinc_mfma_takes(a.step)) {
        // a caller that asked for this form and did not bring its scratch is wrong; computing another arithmetic instead would
        // hide it (and give the row other bits than it gets elsewhere)
        if (!a.scratch || sinc_mfma_scratch_bytes(a.n_sets, a.out_count, a.step) > a.scratch_bytes) return hipErrorInvalidValue;
        return launch_sinc_mfma(a, s);
    }
    const uint32_t row_blocks = (a.rows + kRows - 1) / kRows;
    if (row_blocks > 65535) return hipErrorInvalidValue;
    // outputs per workgroup: as many as the staged span allows at this step (a power of two, so that blocks never
    // straddle the 32-output grid of the index sets)
    uint32_t outs = kMaxOuts;
    while (outs > 1 && (double)outs * a.step + 260.0 > (double)kSpanMax) outs >>= 1;
    if ((double)outs * a.step + 260.0 > (double)kSpanMax) return hipErrorInvalidValue;  // step > ~120: no common rate pair
    // a workgroup takes `group` consecutive blocks of outputs, the next one's input prefetched while it works on the current
    const uint32_t n_out_blocks = (a.out_count + outs - 1) / outs;
    const uint32_t group = n_out_blocks >= 64 ? 8u : (n_out_blocks >= 8 ? 4u : 1u);
    const dim3 grid((n_out_blocks + group - 1) / group, row_blocks);
    constexpr size_t lds_bytes = 2 * kMaxOuts * sizeof(double) + ((size_t)(kSpanMax + 1) * kPitch + (size_t)kMaxOuts * kPitch) * sizeof(float);
    static const hipError_t attr = hipFuncSetAttribute(reinterpret_cast<const void *>(&k_sinc_resample), hipFuncAttributeMaxDynamicSharedMemorySize,
                                                       (int)lds_bytes);
    if (attr != hipSuccess) return attr;
    hipLaunchKernelGGL(k_sinc_resample, grid, dim3(kWaves * 64), lds_bytes, s, a, outs, group);
    return hipGetLastError();
}

hipError_t launch_row_copies(const float *src_base, float *dst_base, const RowCopy *jobs, uint32_t n_jobs, hipStream_t s) {
    if (n_jobs == 0) return hipSuccess;
    if (n_jobs > 65535) return hipErrorInvalidValue;
    hipLaunchKernelGGL(k_row_copies, dim3(4, n_jobs), dim3(256), 0, s, src_base, dst_base, jobs, n_jobs);
    return hipGetLastError();
}

}  // namespace sk
