// resample.hip -- generic-ratio windowed-sinc resampler for gfx950.
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

__global__ __launch_bounds__(256) void k_row_copies(const float *src_base, float *dst_base, const RowCopy *jobs,
                                                    uint32_t n_jobs) {
    const uint32_t j = blockIdx.y;
    if (j >= n_jobs) return;
    const RowCopy job = jobs[j];
    const float *src = src_base + job.src_off;
    float *dst = dst_base + job.dst_off;
    if (job.via_s16) {
        for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < job.count; i += gridDim.x * blockDim.x) {
            // float_sample_to_i16 (soundkit-decoder lib.rs:1815-1827), then audio_data_to_f32_channels' / 32768
            const int r = dev_float_sample_to_i16_f32(src[i]);
            dst[i] = (float)r / 32768.0f;
        }
        return;
    }
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < job.count; i += gridDim.x * blockDim.x) dst[i] = src[i];
}

}  // namespace

uint32_t sinc_rows_per_block() { return kRows; }

hipError_t launch_sinc_resample(const SincArgs &a, hipStream_t s) {
    if (a.rows == 0 || a.out_count == 0) return hipSuccess;
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
