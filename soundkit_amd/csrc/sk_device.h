// sk_device.h -- structures shared between the host engine and the gfx950 kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/soundkit_amd.h"  // sk_mp3_requant_channel

namespace sk {

// One schedule entry = one channel-frame.  off1024: index (in units of 1024 f32) of this
// channel's spectrum inside the packed coeffs buffer and of its PCM inside the planar output.
// win: window_sequence | (window_shape << 2).
struct SynthEntry {
    uint32_t off1024;
    uint32_t win;
};

// One task = one wavefront = one (stream, channel); its entries are applied in order.
struct SynthTask {
    uint32_t state;  // stream * 2 + channel: index into delay / prev_shape
    uint32_t begin;  // first entry
    uint32_t count;  // entries
    uint32_t pad;
};

struct SynthTables {
    const float2 *tw_long;   // [512]  dsp.rs:99-106 twiddle, input_len 1024
    const float2 *tw_short;  // [64]   same, input_len 128
    const float2 *w64;       // [64]   e^{-2 pi i m/64}
    const float2 *w512;      // [512]  e^{-2 pi i m/512}
    const float *win;        // long_sine[2048] long_kbd[2048] short_sine[256] short_kbd[256], then the transition halves:
                             // start2_sine[1024] start2_kbd[1024] (LongStart's second half) stop1_sine[1024] stop1_kbd[1024]
};

struct SynthArgs {
    const float *coeffs;
    float *pcm;
    int16_t *pcm16;       // non-null: planar s16 out instead of `pcm` (same off1024 packing, 1024 i16 per channel-frame)
    float *delay;         // [states][1024]
    uint8_t *prev_shape;  // [states]
    const SynthTask *tasks;
    const SynthEntry *entries;
    uint32_t n_tasks;
    uint32_t only_long;   // 1: the caller vouches that no entry of any task is EightShort (straight-line kernel)
    SynthTables t;
};

// one frame of the planar-f32 -> interleaved-s16 pass that follows synthesis on the s16 path
struct FrameSpan {
    uint32_t off1024;  // planar f32 offset (units of 1024 f32) = s16 offset (units of 1024 i16)
    uint32_t channels;
};

hipError_t launch_aac_synth(const SynthArgs &a, hipStream_t s);
// The decode tail in one kernel (aac_synth.hip k_aac_tail): s.tasks are pairs as for launch_aac_synth_pairs(.., false, ..) --
// no EightShort frame anywhere, tasks 2p / 2p + 1 of equal length -- every task's entries are ALL frames of its channel in
// this launch, the first one frame 0 (the FIR is one-shot over them: silence in front, as downsample_audio).
struct TailArgs {
    SynthArgs s;
    const uint32_t *afrag_f16;      // FirArgs::afrag_f16
    int16_t *out16;                 // [stream row][out_stride frames][channels] interleaved s16; every stream of the launch has the same channel count
    size_t out_stride;              // frames per stream row
    size_t stream_stride;           // samples between the first frames of consecutive streams in the spectra's packing
};
hipError_t launch_aac_tail(const TailArgs &ta, hipStream_t s);
// OnlyLong tasks two per wave: a.tasks[2p] and a.tasks[2p + 1] have the same count (a.n_tasks even)
// with_short: the tasks' EightShort frames coincide pairwise (both channels of a pair switch together): the kernel with the
// wave-uniform eight-short arm; without: the caller vouches that no entry is EightShort
hipError_t launch_aac_synth_pairs(const SynthArgs &a, bool with_short, hipStream_t s);
hipError_t launch_reset_streams(float *delay, uint8_t *shape, uint32_t *pns, const uint32_t *ids, uint32_t n, hipStream_t s);
hipError_t launch_zero_spans(float *base, const uint32_t *ids, uint32_t n, uint32_t span_floats, hipStream_t s);
hipError_t launch_frames_to_s16(const float *planar, int16_t *out, const FrameSpan *frames, uint32_t n, hipStream_t s);
hipError_t launch_dequantize(const int16_t *q, const int16_t *sf, float *out, size_t n, const float *pow43,
                             const float *sftab, hipStream_t s);

// pcm.hip
hipError_t launch_pcm_convert(int op, const void *in, void *out, size_t n, hipStream_t s);
hipError_t launch_interleave(const void *planar, void *out, size_t frames, uint32_t ch, int elem_bytes, hipStream_t s);
hipError_t launch_deinterleave(const void *in, void *planar, size_t frames, uint32_t ch, int elem_bytes, hipStream_t s);
hipError_t launch_deinterleave_s24(const uint8_t *in, int32_t *planar, size_t frames, uint32_t ch, hipStream_t s);
hipError_t launch_bytes_to_f32_planar(int variant, int fmt, const uint8_t *in, size_t frames, uint32_t ch, float *planar,
                                      hipStream_t s);
hipError_t launch_f32_planar_to_bytes(int fmt, const float *planar, size_t frames, uint32_t ch, uint8_t *out,
                                      hipStream_t s);
hipError_t launch_downmix_mono(const float *planar, size_t frames, uint32_t ch, float *mono, hipStream_t s);
hipError_t launch_exact_to_i16(int fmt, const uint8_t *in, size_t samples, uint8_t *out, hipStream_t s);
hipError_t launch_f32_planar_to_bytes_batch(int fmt, const float *planar, size_t batch, size_t plane_stride, size_t frames,
                                            uint32_t ch, uint8_t *out, hipStream_t s);

// fir.hip
#ifdef __HIPCC__
// float_sample_to_i16, soundkit-decoder lib.rs:1815-1827: non-finite -> 0, clamp, x32768 / x32767 in f64, round half away
__device__ __forceinline__ int dev_float_sample_to_i16(float x) {
    const float f = isfinite(x) ? fminf(fmaxf(x, -1.0f), 1.0f) : 0.0f;
    const double scaled = f < 0.0f ? (double)f * 32768.0 : (double)f * 32767.0;
    const int r = (int)round(scaled);
    return r < -32768 ? -32768 : (r > 32767 ? 32767 : r);
}

// The same function without f64, exact.  With m = |x| clamped: a = m * 32768 is exact (power of two), a = k + d with
// k = floor(a), d in [0, 1) exact.  Negative side: the value is -a, so round-half-away is -(k + (d >= 0.5)).
// Positive side: the value is m * 32767 = k + (d - m) with d - m in (-1, 1); it rounds (half away) to k + 1 when
// d - m >= 0.5, to k - 1 when d - m < -0.5, else k.  d - 0.5 and d + 0.5 are exact in f32 wherever the comparison
// is not already decided by magnitudes (a >= 0.5 gives d a granularity of at least 2^-24; below that m < 2^-15), so
// both comparisons are exact.  tests/test_pcm_gpu.py compares it with the f64 form over all f32 inputs of interest.
__device__ __forceinline__ int dev_float_sample_to_i16_f32(float x) {
    const float m = isfinite(x) ? fminf(fabsf(x), 1.0f) : 0.0f;
    const float a = m * 32768.0f;
    const float k = floorf(a);
    const float d = a - k;
    const int ki = (int)k;
    const int neg = ki + (d >= 0.5f ? 1 : 0);
    const int pos = ki + ((d - 0.5f >= m) ? 1 : 0) - ((d + 0.5f < m) ? 1 : 0);
    return x < 0.0f ? -neg : pos;
}
// The same again with fewer instructions (the synthesis kernel converts 16 samples per lane per frame and is bound by
// vector issue when it does): one formula for both signs -- on the negative side the comparison value is 0 instead of m,
// which turns  (d - 0.5 >= m) - (d + 0.5 < m)  into  (d >= 0.5)  -- fract for the fraction, the sign put back with a xor.
__device__ __forceinline__ int dev_float_sample_to_i16_v2(float x) {
    const float m = isfinite(x) ? fminf(fabsf(x), 1.0f) : 0.0f;
    const float a = m * 32768.0f;                 // exact
    const float d = __builtin_amdgcn_fractf(a);   // exact: a < 2^15 + 1
    const int ki = (int)a;                        // floor, a >= 0
    const float mm = x < 0.0f ? 0.0f : m;
    const int r = ki + ((d - 0.5f >= mm) ? 1 : 0) - ((d + 0.5f < mm) ? 1 : 0);
    return x < 0.0f ? -r : r;
}
// The shortest form: through f64, which this part issues at the rate of unpacked f32.  t = x * 32767.5 - |x| / 2 is
// x * 32767 for x >= 0 and x * 32768 for x < 0, exact in f64 (24 x 16 bits); half a unit away from zero is folded into
// the first fma; v_cvt_i32_f64 truncates, saturates and turns NaN into 0, and fma(x, 0, x) turns +-inf into NaN, so the
// clamp and the non-finite rule of the reference come for free -- PROVIDED the result is then packed with
// v_cvt_pk_i16_i32, which saturates to i16 (values beyond +-1 arrive here as up to +-2^31).  Returns that pre-saturation
// integer; use dev_pack2_s16.  tools/check_f32_rounding.c sweeps all 2^32 inputs against the reference's form.
__device__ __forceinline__ int dev_float_sample_to_i16_presat(float x) {
    const float y = __builtin_fmaf(x, 0.0f, x);
    const double X = (double)y;
    const double half = __builtin_copysign(0.5, X);
    const double h = __builtin_fma(__builtin_fabs(X), -0.5, half);
    const double t = __builtin_fma(X, 32767.5, h);
    int k;
    asm("v_cvt_i32_f64 %0, %1" : "=v"(k) : "v"(t));  // the instruction's own out-of-range rule, not C's undefined cast
    return k;
}
// the part behind the non-finite guard (y = x for finite x, NaN otherwise)
__device__ __forceinline__ int dev_guarded_sample_to_i16_presat(float y) {
    const double X = (double)y;
    const double half = __builtin_copysign(0.5, X);
    const double h = __builtin_fma(__builtin_fabs(X), -0.5, half);
    const double t = __builtin_fma(X, 32767.5, h);
    int k;
    asm("v_cvt_i32_f64 %0, %1" : "=v"(k) : "v"(t));
    return k;
}
__device__ __forceinline__ uint32_t dev_pack2_s16(float lo, float hi) {
    typedef short s16x2_t __attribute__((ext_vector_type(2)));
    typedef float f32x2_t __attribute__((ext_vector_type(2)));
    const f32x2_t x = {lo, hi};
    const f32x2_t y = __builtin_elementwise_fma(x, (f32x2_t){0.0f, 0.0f}, x);  // the guard of both samples in one v_pk_fma_f32
    const s16x2_t p = __builtin_amdgcn_cvt_pk_i16(dev_guarded_sample_to_i16_presat(y.x), dev_guarded_sample_to_i16_presat(y.y));
    return __builtin_bit_cast(uint32_t, p);
}

// The same conversion for an accumulator that still carries a power-of-two factor 2^EXP (the FIR's s16 rows go in as integers,
// its f16 taps times 2^16): x = acc * 2^-EXP is folded into the two f64 constants -- exact, a power of two -- which saves the
// f32 multiply, and a sum of finite products needs no non-finite guard, which saves the fma(x, 0, x).  Same integer as
// dev_float_sample_to_i16_presat(acc * 2^-EXP) for every finite acc (where that product is an f32 denormal both give 0).
template <int EXP>
__device__ __forceinline__ int dev_scaled_sample_to_i16_presat(float acc) {
    constexpr double kScale = 1.0 / (double)(1ull << EXP);
    const double X = (double)acc;
    const double half = __builtin_copysign(0.5, X);
    const double h = __builtin_fma(__builtin_fabs(X), -0.5 * kScale, half);
    const double t = __builtin_fma(X, 32767.5 * kScale, h);
    int k;
    asm("v_cvt_i32_f64 %0, %1" : "=v"(k) : "v"(t));
    return k;
}
template <int EXP>
__device__ __forceinline__ uint32_t dev_pack2_s16_scaled(float lo, float hi) {
    typedef short s16x2_t __attribute__((ext_vector_type(2)));
    const s16x2_t p = __builtin_amdgcn_cvt_pk_i16(dev_scaled_sample_to_i16_presat<EXP>(lo), dev_scaled_sample_to_i16_presat<EXP>(hi));
    return __builtin_bit_cast(uint32_t, p);
}

// ---- shared by fir_bf16.hip and the fused decode-tail kernel (aac_synth.hip): the f16 form of the 48 -> 16 kHz FIR on s16 samples
typedef uint32_t sk_u32x4 __attribute__((ext_vector_type(4)));
typedef float sk_f32x4 __attribute__((ext_vector_type(4)));
// two s16 samples in one dword (the earlier one low) -> their two f16 planes: p1 = the sample rounded TOWARD ZERO to f16 (eleven
// significand bits; no overflow at 32767), p2 = the rest (same sign, below 32: exact).  Three and a half vector instructions per sample.
__device__ __forceinline__ void dev_split_pair16_f16(uint32_t u, uint32_t &p1, uint32_t &p2) {
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    const f32x2 f = {(float)(short)(u & 0xffffu), (float)((int)u >> 16)};
    const auto a = __builtin_amdgcn_cvt_pkrtz(f.x, f.y);
    const f32x2 r = f - (f32x2){(float)a[0], (float)a[1]};  // exact; one packed subtraction
    p1 = __builtin_bit_cast(uint32_t, a);
    p2 = __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_pkrtz(r.x, r.y));
}
__device__ __forceinline__ sk_f32x4 dev_mfma_f16(const sk_u32x4 &av, const sk_u32x4 &bv, const sk_f32x4 &c) {
#if SK_MFMA_K16  // experiment (-DSK_MFMA_K16=1): the K = 16 form older parts have, twice -- a lane's eight k values as two groups of four,
                 // the same products, summed in another order.  Half the matrix rate; not an aggressor for other kernels' packed-f32
                 // instructions (profiles/r04_lanes_corruption.md).
    typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
    typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
    const sk_f32x4 t = __builtin_amdgcn_mfma_f32_16x16x16f16(__builtin_bit_cast(f16x4, (u32x2){av.x, av.y}), __builtin_bit_cast(f16x4, (u32x2){bv.x, bv.y}), c, 0, 0, 0);
    return __builtin_amdgcn_mfma_f32_16x16x16f16(__builtin_bit_cast(f16x4, (u32x2){av.z, av.w}), __builtin_bit_cast(f16x4, (u32x2){bv.z, bv.w}), t, 0, 0, 0);
#else
    typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, av), __builtin_bit_cast(f16x8, bv), c, 0, 0, 0);
#endif
}
__device__ __forceinline__ sk_f32x4 dev_mfma_bf16(const sk_u32x4 &av, const sk_u32x4 &bv, const sk_f32x4 &c) {
#if SK_MFMA_K16
    typedef short s16x4 __attribute__((ext_vector_type(4)));
    typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
    const sk_f32x4 t = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(__builtin_bit_cast(s16x4, (u32x2){av.x, av.y}), __builtin_bit_cast(s16x4, (u32x2){bv.x, bv.y}), c, 0, 0, 0);
    return __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(__builtin_bit_cast(s16x4, (u32x2){av.z, av.w}), __builtin_bit_cast(s16x4, (u32x2){bv.z, bv.w}), t, 0, 0, 0);
#else
    typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, av), __builtin_bit_cast(bf16x8, bv), c, 0, 0, 0);
#endif
}
// products kept per window of 32 samples, in the order x1h1 | x1h2, x2h1 (fir_bf16.hip explains the budget)
__device__ constexpr int kFirProductsF16[10] = {1, 3, 3, 3, 3, 3, 3, 3, 1, 1};
#endif

struct FirArgs {
    const float *in;      // [rows][in_stride]
    const int16_t *in16;  // non-null (fir_bf16.hip only): the rows are s16, sample value s / 32768 -- same indexing as `in`
    float *out;           // [rows][out_stride]
    // fused s16 output (out == nullptr): the worker's f32_channels_to_bytes for 16 bits, interleaved over out16_ch
    // adjacent rows (1: each row its own mono stream; 2: rows 2k / 2k+1 are L / R of stream k)
    int16_t *out16;       // [rows / out16_ch][out16_stride frames][out16_ch]
    size_t out16_stride;
    uint32_t out16_ch;
    const float *zeros;   // >= 1 KiB of zeros (source for out-of-range input)
    const uint32_t *afrag16;  // [10][3][64][4] bf16 A-operand fragments of the Toeplitz tap matrix
    const uint32_t *afrag_f16;  // [10][2][64][4] f16 A-operand fragments of the taps times 2^16 (s16 rows)
    const float *taps;    // [256]
    size_t in_stride, out_stride;
    // frame-packed input (in_block 0 = plain rows; otherwise 1024 with in_ch 1 or 2): sample n of row r lives at
    //   in + (r / in_ch) * in_group_stride + (r % in_ch) * 1024 + (n / 1024) * in_block_stride + n % 1024
    uint32_t in_block, in_ch;
    size_t in_block_stride, in_group_stride;
    const uint32_t *row_map;  // optional: logical row r reads physical input row row_map[r]
    const uint32_t *out_off;  // optional: logical row r writes at column out_off[r] of its output row
    uint32_t rows;
    uint32_t in_frames;   // valid input samples per row (n >= in_frames reads as 0)
    int32_t in_origin;    // sample index of in[r][0] relative to the stream's time 0 (history rows: negative)
    uint32_t out_first;   // first output index m to produce
    uint32_t out_count;   // outputs per row
};
// fir_bf16.hip: the 48 -> 16 kHz filter on the bf16 / f16 matrix cores (exact split of both operands).  Rows of more than
// ~7e8 outputs in one call are refused (32-bit tile indices inside the kernel): hipErrorInvalidValue.
hipError_t launch_fir_48k_16k(const FirArgs &a, hipStream_t s);

// resample.hip -- generic-ratio windowed-sinc resampling (rubato SincFixedIn<f32>, Linear interpolation)
struct SincArgs {
    const float *in;          // [phys rows][in_stride]
    float *out;               // [rows][out_stride]
    const float *sincs;       // [256][256] sub-filter table of this ratio, followed by [256][256][2]: taps of sub-filters s and s + 1 interleaved
    // Time indices: rubato advances an f64 index by `step` before every output.  Rows may sit at different points of
    // that walk (streams of different ages), so indices come as "sets": set s holds the index of every 32nd output
    // (starts[s * starts_stride + b] = index of output 32 b) and its output count; a lane reproduces the additions
    // in between.  row_set picks the set of a row (null: set 0 for every row).  A workgroup takes sinc_rows_per_block()
    // consecutive rows, which must share their set: the host orders the rows by set and pads each set's rows to a multiple
    // of that with rows whose row_map entry is 0xffffffff (nothing read, nothing written).
    const double *set_starts;
    const uint32_t *set_count;
    const uint32_t *row_set;
    uint32_t starts_stride;
    double step;
    const uint32_t *row_map;  // optional, as FirArgs
    const uint32_t *out_off;  // optional, as FirArgs
    size_t in_stride, out_stride;
    uint32_t rows, in_frames, out_count;  // out_count: the largest count among the sets in use
    int32_t in_origin;        // sample index of in[r][0] in the index time base (history rows: negative)
    // the matrix-core form (resample.hip, k_sinc_taps + k_sinc_mfma): taken when the caller lends scratch of
    // sinc_mfma_scratch_bytes(n_sets, out_count, step) bytes and does not ask for rubato's own order of operations
    uint32_t n_sets;          // index sets in set_starts / set_count (0: unknown -> the scalar form)
    void *scratch;
    size_t scratch_bytes;
    int exact;                // 1: the scalar form, whose sums are rubato's bit for bit
};
hipError_t launch_sinc_resample(const SincArgs &a, hipStream_t s);
uint32_t sinc_rows_per_block();
size_t sinc_mfma_scratch_bytes(uint32_t n_sets, uint32_t out_count, double step);  // 0: this shape stays on the scalar form

// batched row copies (streaming resampler bookkeeping): job j copies count floats
struct RowCopy {
    uint64_t src_off, dst_off;  // element offsets from the two bases
    uint32_t count;
    uint32_t via_s16;  // 1: each sample goes through float_sample_to_i16 then / 32768 (the worker hands the resampler
                       // the i16 AudioData of decode_aac_access_unit, soundkit-decoder lib.rs:1793-1813, 3563-3617)
    // `pieces` runs of `count` elements, src_stride elements apart at the source, back to back at the destination: the units of a
    // stream in a tick (one job per channel instead of one per unit and channel: 83 000 jobs a tick otherwise, 0.85 ms to write down)
    uint32_t pieces = 1;
    uint32_t src_stride = 0;
};
hipError_t launch_row_copies(const float *src_base, float *dst_base, const RowCopy *jobs, uint32_t n_jobs, hipStream_t s);

// mp3_hybrid.hip -- Layer III hybrid synthesis (ISO/IEC 11172-3 2.4.3.4), one wave per (stream, channel)
constexpr uint32_t kMp3StateFloats = 1664;  // overlap[576] | ring[1024] | ring position | padding
struct Mp3Args {
    const float *xr;       // [granule-channels][576] requantised, stereo-processed, reordered frequency lines
    float *pcm;            // interleaved f32 out ...
    int16_t *pcm16;        // ... or s16 (f32_to_i16, soundkit-mp3 lib.rs:376-385); one of the two
    uint32_t planar_stride;  // != 0 (the scheduler's tick, with pcm): channel row `off` goes to pcm + off * planar_stride, 576 samples,
                             // each as f32_to_i16(sample) / 32768 -- the i16 AudioData Mp3Decoder hands the worker, as the f32 that
                             // audio_data_to_f32_channels makes of it (soundkit-decoder lib.rs:3563-3617); exact in f32
    float *state;          // [states][kMp3StateFloats]
    const SynthTask *tasks;      // state = stream * 2 + channel
    const SynthEntry *entries;   // off1024 = index of the channel's 576 lines; win = block_type | mixed << 2 | (channels - 1) << 3 | channel << 4
    uint32_t n_tasks;
    const float *imdct;    // [4][36][20]: IMDCT x window matrices of block types 0..3 (rows padded to 20)
    const float *matrix;   // [64][32]: cos((16 + i)(2k + 1) pi / 64)
    const float *window;   // [512]: the synthesis window D (caller-supplied)
    const float *cs_ca;    // [16]: alias-reduction cs[8] | ca[8]
};
hipError_t launch_mp3_hybrid(const Mp3Args &a, hipStream_t s);

// mp3_requant.hip -- Layer III requantisation + joint stereo + short-block reorder, one wave per granule
constexpr uint32_t kMp3Rates = 9;      // 44.1 / 48 / 32 / 22.05 / 24 / 16 / 11.025 / 12 / 8 kHz: one band-table slot each
constexpr uint32_t kMp3BandRow = 40;   // u16 per slot: 23 long offsets | 14 short offsets | padding
constexpr uint32_t kMp3Pow43 = 8208;   // |is| <= 15 + 2^13 - 1 (the widest linbits escape)
struct Mp3RequantRecord {              // what the host makes of one sk_mp3_requant_granule
    uint32_t off;                      // index of the granule's first channel, in 576-line rows of is / xr
    uint8_t slot, channels, flags, reserved;  // slot >= kMp3Rates: rejected, the kernel writes silence; flags = ms | intensity << 1
    sk_mp3_requant_channel ch[2];
};
static_assert(sizeof(Mp3RequantRecord) % 4 == 0 && sizeof(Mp3RequantRecord) <= 256, "one word per lane");
struct Mp3RequantArgs {
    const Mp3RequantRecord *records;
    const int16_t *is;
    float *xr;
    uint32_t n;
    const float *pow43;     // [kMp3Pow43]
    const float *root4;     // 2^(0/4) .. 2^(3/4)
    const float *is_k;      // [8]: t / (1 + t), t = tan(i pi / 12), i < 7
    const uint16_t *bands;  // [kMp3Rates][kMp3BandRow]
    const uint8_t *pretab;  // [kMp3Rates][24]
    const uint32_t *line_map;  // [kMp3Rates][long | short | mixed][576]: dest | band << 10 | (window + 1) << 15 of every bitstream-order line
};
hipError_t launch_mp3_requant(const Mp3RequantArgs &a, hipStream_t s);

// aac_entropy.hip -- the AAC-LC front-end on the device, one stream per lane
}  // namespace sk
#include "aac_entropy_core.h"
namespace sk {
constexpr int EC_SKIPPED = -199;  // a unit after a failed one of the same stream in the same launch
struct EntropyUnit {
    uint32_t word_offset;  // of the access unit in `words` (each unit 4-byte aligned, followed by >= 8 zero bytes)
    uint32_t byte_len;
    uint32_t off1024;      // where its spectra go (units of 1024 f32)
    uint32_t entry[2];     // synthesis schedule entries of its channels (their `win` is filled in here)
    uint32_t task;         // its stream's EntropyTask
};
struct EntropyTask {       // one stream
    uint32_t stream, first, count;
    int32_t sf_index;
    uint32_t channels;
};
struct EntropyArgs {
    sk_ec::Tables t;       // device pointers
    // the part of the tables every codeword touches, as one blob the kernel copies into LDS: index block, Huffman tables,
    // tuple table, scale-factor multipliers, band offsets (byte offsets from lds_blob)
    const uint8_t *lds_blob;
    uint32_t lds_bytes;
    uint32_t lds_meta_off, lds_lut_off, lds_tuple_off, lds_sf_off, lds_swb_off, lds_pow_off;
    const uint32_t *words;
    const EntropyUnit *units;
    const EntropyTask *tasks;
    uint32_t n_tasks;
    uint32_t *pns_state;   // [max_streams]
    float *coeffs;
    SynthEntry *entries;
    int32_t *status;       // per unit
    // frame-parallel form: one lane per access unit in the first and third kernel, one per stream in between
    uint32_t n_units;
    sk_ec::Scratch *side;  // [n_units] side information handed from the first phase to the third
    uint32_t *pns_start;   // [n_units] generator state each unit starts from
    uint32_t lane_shift;   // the first 64 >> lane_shift lanes of a wave carry a unit each (0..4)
    const uint32_t *order; // [n_units] or null: slot -> unit.  A wave runs until its slowest lane is done: the host hands out the
                           // units sorted by size, so that a wave's lanes have about the same number of codewords
    // quantised hand-over (sk_tick_run_q): instead of parsing, phase one rebuilds the side information from the host's
    // record and dequantises the host's integers; phase three ends with the host's verdict on the rest of the unit
    const sk_ec::WireUnit *wire;  // [n_units], null in the other modes
    const int16_t *quant;         // packed like coeffs
};
hipError_t launch_aac_entropy_parallel(const EntropyArgs &a, hipStream_t s);  // parse | link | finish

// pcm.hip -- the output stage of apply_output_options (soundkit-decoder lib.rs:3324-3456) for a batch of
// planar f32 pieces: [s16 round trip] -> [mono downmix] -> interleaved little-endian bytes.
enum PackMode : uint8_t {
    kPackPlain = 0,   // src is already the f32 the reference holds (resampler output)
    kPackViaS16 = 1,  // src is synthesis output: q = float_sample_to_i16(x); the f32 is q / 32768
    kPackDirect = 2,  // fast path (lib.rs:3339-3345): the bytes are q itself, s16le
    kPackFromQ = 3,   // src is q / 32768 already (MP3 rows of a tick): the fast path's bytes are (int)(x * 32768), exact
};
struct PackJob {
    const float *src0, *src1;  // channel rows (src1 unused when ch_in == 1)
    uint8_t *dst;              // 4-byte aligned
    uint32_t frames;
    uint8_t ch_in, ch_out, bits, mode;  // ch_out < ch_in: downmix to mono (lib.rs:3492-3561); bits 16 / 24 / 32
};
hipError_t launch_pack_jobs(const PackJob *jobs, uint32_t n_jobs, uint32_t max_frames, hipStream_t s);

}  // namespace sk
