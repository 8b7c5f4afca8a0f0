// aac_synth.hip -- batched AAC-LC synthesis for gfx950 (MI355X).
//
// Replaces, for a whole batch of streams at once, the reference's
//   AacLcDecoder::synthesize_channel            soundkit-aac-lc/src/decoder.rs:336-374
//   DspChannel::synthesize_long_sequence        soundkit-aac-lc/src/dsp.rs:230-282
//   DspChannel::synthesize_eight_short          soundkit-aac-lc/src/dsp.rs:284-338
//   imdct_fast                                  soundkit-aac-lc/src/dsp.rs:476-535
//
// Mapping: one 64-lane wavefront owns one (stream, channel) and walks that channel's
// frames in order.  The 1024-sample overlap delay lives in 16 VGPRs per lane for the
// whole launch (HBM sees it once at the start and once at the end), the N/2 = 512-point
// complex FFT is 8 x 8 x 8 with one radix-8 butterfly per lane per stage and two
// conflict-free LDS exchanges, and every global access is a full-wave contiguous run
// (512 B float2 loads of the spectrum, 1 KiB float4 stores of the PCM).
//
// Lane l owns output samples  i = 4l+256r+e  and  i = 1020-4l-256r+e  (r = 0,1; e = 0..3):
// the IMDCT's odd symmetry means those 16 samples, and the 16 delay samples at the same
// indices, need exactly the FFT bins {256+2l+128r, +1} and {254-2l-128r, +1}, so the
// post-twiddled spectrum crosses LDS once (8 x ds_write_b64, 4 x ds_read_b128 per lane).
#include "sk_device.h"

// streaming accesses of the main path: spectra are read once, PCM written once.  Plain loads and stores: the non-temporal
// forms measured no better (profiles/r01_ab_aac_synth.md)
#define SK_SYNTH_LOAD(p) (*(p))
#define SK_SYNTH_STORE(v, p) (*(p) = (v))

namespace sk {

namespace {

typedef float f2 __attribute__((ext_vector_type(2)));
typedef float f4 __attribute__((ext_vector_type(4)));

#ifndef SK_SYNTH_BLOCK_WAVES
#define SK_SYNTH_BLOCK_WAVES 4
#endif
constexpr int kWavesPerBlock = SK_SYNTH_BLOCK_WAVES;
#ifndef SK_WAVES_PER_SIMD
#define SK_WAVES_PER_SIMD 3
#endif
#ifndef SK_PREFETCH_DEPTH
#define SK_PREFETCH_DEPTH 1
#endif
constexpr int kDepth = SK_PREFETCH_DEPTH;  // spectra in flight per wave
constexpr int kWavesPerSimd = SK_WAVES_PER_SIMD;  // occupancy the register budget is held to
constexpr int kExchange = 576;  // f2 per wave: max(8*68, 8*72)
constexpr int kStage = 1024;    // floats per wave: rare-path staging (transition windows, eight-short output)

__device__ __forceinline__ f2 cmul(f2 a, f2 b) {
    return (f2){fmaf(a.x, b.x, -a.y * b.y), fmaf(a.x, b.y, a.y * b.x)};
}
__device__ __forceinline__ f2 cadd(f2 a, f2 b) { return (f2){a.x + b.x, a.y + b.y}; }
__device__ __forceinline__ f2 csub(f2 a, f2 b) { return (f2){a.x - b.x, a.y - b.y}; }
// multiply by -i
__device__ __forceinline__ f2 mul_mi(f2 a) { return (f2){a.y, -a.x}; }

// forward 8-point DFT (e^{-2 pi i nk/8}), natural order in and out, all in registers
__device__ __forceinline__ void dft8(f2 (&x)[8]) {
    const float h = 0.70710678118654752440f;
    // even half: DFT4(x0,x2,x4,x6)
    f2 t0 = cadd(x[0], x[4]), t1 = csub(x[0], x[4]);
    f2 t2 = cadd(x[2], x[6]), t3 = mul_mi(csub(x[2], x[6]));
    f2 e0 = cadd(t0, t2), e2 = csub(t0, t2), e1 = cadd(t1, t3), e3 = csub(t1, t3);
    // odd half: DFT4(x1,x3,x5,x7)
    f2 u0 = cadd(x[1], x[5]), u1 = csub(x[1], x[5]);
    f2 u2 = cadd(x[3], x[7]), u3 = mul_mi(csub(x[3], x[7]));
    f2 o0 = cadd(u0, u2), o2 = csub(u0, u2), o1 = cadd(u1, u3), o3 = csub(u1, u3);
    // twiddles W8^k
    f2 w1 = (f2){(o1.x + o1.y) * h, (o1.y - o1.x) * h};   // o1 * (1-i)/sqrt2
    f2 w2 = mul_mi(o2);                                          // o2 * (-i)
    f2 w3 = (f2){(o3.y - o3.x) * h, -(o3.x + o3.y) * h};  // o3 * (-1-i)/sqrt2
    x[0] = cadd(e0, o0); x[4] = csub(e0, o0);
    x[1] = cadd(e1, w1); x[5] = csub(e1, w1);
    x[2] = cadd(e2, w2); x[6] = csub(e2, w2);
    x[3] = cadd(e3, w3); x[7] = csub(e3, w3);
}

__device__ __forceinline__ void wave_sync() {
    // LDS traffic of one wave is ordered by the hardware; this only stops the compiler
    // from moving LDS accesses across the exchange points.
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

// The schedule is read through the constant address space: a wave-uniform load from it is a SCALAR load (s_load, counted
// in lgkmcnt).  As a plain global pointer the compiler cannot rule out that the kernel's own stores alias it, so it
// issued a vector load plus s_waitcnt vmcnt(0) at the top of every frame -- which also waits for the PCM stores of the
// frame before to be acknowledged by memory: the wave drained its whole memory pipeline once (twice, with the look-up
// for the prefetch address) per frame, and that, not bandwidth, set the launch time (profiles/r02_ab_synth_groups.md).
typedef const __attribute__((address_space(4))) SynthEntry *const_entries;
__device__ __forceinline__ const_entries as_constant(const SynthEntry *p) {
    return reinterpret_cast<const_entries>(reinterpret_cast<uintptr_t>(p));
}

typedef uint32_t u2 __attribute__((ext_vector_type(2)));
// four PCM samples as the s16 the worker emits (float_sample_to_i16, soundkit-decoder lib.rs:1815-1827), packed
// float_sample_to_i16 in its shortest exact form (through f64, sk_device.h; tools/check_f32_rounding.c sweeps all 2^32
// inputs): the s16 variant of the kernel is bound by vector issue, and with the f32 form a third of it was these conversions
__device__ __forceinline__ u2 pack4_s16(const f4 &v) {
    return (u2){dev_pack2_s16(v.x, v.y), dev_pack2_s16(v.z, v.w)};
}

// first-half and second-half window of a long-transform frame (dsp.rs:353-387): the long window of the previous / current
// shape, or for LongStop / LongStart the piecewise half that engine.cpp tabulates behind the four windows
__device__ __forceinline__ const float *first_half_window(const float *win, int seq, int prev_shape) {
    return seq == 3 ? win + 6656 + 1024 * prev_shape : win + 2048 * prev_shape;
}
__device__ __forceinline__ const float *second_half_window(const float *win, int seq, int shape) {
    return seq == 1 ? win + 4608 + 1024 * shape : win + 2048 * shape + 1024;
}

typedef __attribute__((address_space(3))) f2 lds_f2;
typedef __attribute__((address_space(3))) f4 lds_f4;
typedef __attribute__((address_space(3))) float lds_f;

// this lane's 16 positions (4l+256r+e, 1020-4l-256r+e) of a 1024-float LDS array
__device__ __forceinline__ void read_positions(const lds_f *buf, int lane, float (&v)[16]) {
#pragma unroll
    for (int r = 0; r < 2; ++r) {
        const int j = 4 * lane + 256 * r;
        const f4 f = *reinterpret_cast<const lds_f4 *>(buf + j);
        const f4 m = *reinterpret_cast<const lds_f4 *>(buf + 1020 - j);
        v[8 * r + 0] = f.x; v[8 * r + 1] = f.y; v[8 * r + 2] = f.z; v[8 * r + 3] = f.w;
        v[8 * r + 4] = m.x; v[8 * r + 5] = m.y; v[8 * r + 6] = m.z; v[8 * r + 7] = m.w;
    }
}

__device__ __forceinline__ void write_positions(lds_f *buf, int lane, const float (&v)[16]) {
#pragma unroll
    for (int r = 0; r < 2; ++r) {
        const int j = 4 * lane + 256 * r;
        *reinterpret_cast<lds_f4 *>(buf + j) = (f4){v[8 * r + 0], v[8 * r + 1], v[8 * r + 2], v[8 * r + 3]};
        *reinterpret_cast<lds_f4 *>(buf + 1020 - j) = (f4){v[8 * r + 4], v[8 * r + 5], v[8 * r + 6], v[8 * r + 7]};
    }
}

// 512-point forward FFT of z (z[r] = element 64 r + lane) through two LDS exchanges:
// n = 64 n1 + 8 n2 + n3, k = k1 + 8 k2 + 64 k3; on return z[j] = Z[lane + 64 j].
// Stage 1 (lane = 8 n2 + n3) needs W64^{n2 k1}; stage 2 (lane = 8 n3 + k1) needs
// W512^{n3 (k1 + 8 k2)} = W512^{n3 k1} * W64^{n3 k2}: both W64 factors are W64^{(lane>>3) k}, so one
// 8 x 8 table (t64[k][lane>>3], in LDS) serves both stages and stage 2 adds one per-lane constant.
__device__ __forceinline__ void fft512(f2 (&z)[8], lds_f2 *ex, const lds_f2 *t64, f2 base2, int lane) {
    const int hi3 = lane >> 3, lo3 = lane & 7;
    dft8(z);  // over n1 -> k1
#pragma unroll
    for (int k = 1; k < 8; ++k) z[k] = cmul(z[k], t64[8 * k + hi3]);
#pragma unroll
    for (int k = 0; k < 8; ++k) ex[k * 68 + lane] = z[k];
    wave_sync();
#pragma unroll
    for (int n2 = 0; n2 < 8; ++n2) z[n2] = cmul(ex[lo3 * 68 + 8 * n2 + hi3], base2);
    wave_sync();
    dft8(z);  // over n2 -> k2
#pragma unroll
    for (int k = 1; k < 8; ++k) z[k] = cmul(z[k], t64[8 * k + hi3]);
#pragma unroll
    for (int k = 0; k < 8; ++k) ex[hi3 * 72 + k * 8 + lo3] = z[k];
    wave_sync();
#pragma unroll
    for (int n3 = 0; n3 < 8; ++n3) z[n3] = ex[n3 * 72 + lane];
    wave_sync();
    dft8(z);  // over n3 -> k3
}

// Everything after the pre-twiddle for an EightShort frame (a few percent of real frames; LongStart and LongStop
// run the long path with their piecewise windows as tables).  Kept out of line on purpose: inlined, its
// registers and hoisted table loads are charged to the long loop and halve the occupancy.
// Its vector inputs and outputs travel through the wave's LDS, so that nothing but scalars and
// pointers is live across the call:
//   in : ex[64 r + lane] = pre-twiddled FFT input z[r] (long: element 64 r + lane; short: element
//        `lane` of block r); stage = overlap delay at this lane's 16 positions (write_positions)
//   out: PCM written to out_ptr; stage = new delay at this lane's 16 positions
__device__ __attribute__((noinline)) void synth_rare_frame(lds_f2 *ex, lds_f *stage, const lds_f2 *tw_lds,
                                                            const lds_f2 *t64, float base2_re, float base2_im,
                                                            const lds_f *short_win, const lds_f2 *w64, const lds_f2 *tw_short,
                                                            float *out_ptr, int16_t *out16_ptr, int seq, int prev_shape,
                                                            int shape, int lane) {
    const int hi3 = lane >> 3, lo3 = lane & 7;
    (void)base2_re, (void)base2_im, (void)t64, (void)tw_lds, (void)seq;  // only the short transform is left here
    // short_win (the sine and the KBD short window), w64, tw_short: LDS copies made once per workgroup -- from global memory
    // they were three trips to memory in the middle of every EightShort frame
    const lds_f *prev_short = short_win + 256 * prev_shape;
    const lds_f *cur_short = short_win + 256 * shape;
    f2 z[8];
#pragma unroll
    for (int r = 0; r < 8; ++r) z[r] = ex[64 * r + lane];
    float dly[16];
    read_positions(stage, lane, dly);
    wave_sync();
    float o[16], d[16];  // windowed first half / second half at this lane's 16 positions

    {
        // eight short windows (dsp.rs:284-338): 8 independent 64-point FFTs, lane = 8 w + a,
        // n = a + 8 b, k = kb + 8 ka
#pragma unroll
        for (int w = 0; w < 8; ++w) ex[64 * w + lane] = z[w];
        wave_sync();
#pragma unroll
        for (int b2 = 0; b2 < 8; ++b2) z[b2] = ex[64 * hi3 + lo3 + 8 * b2];
        wave_sync();
        dft8(z);  // over b -> kb
#pragma unroll
        for (int kb = 1; kb < 8; ++kb) z[kb] = cmul(z[kb], w64[lo3 * kb]);
#pragma unroll
        for (int kb = 0; kb < 8; ++kb) ex[64 * hi3 + 8 * kb + lo3] = z[kb];
        wave_sync();
#pragma unroll
        for (int aa = 0; aa < 8; ++aa) z[aa] = ex[64 * hi3 + 8 * lo3 + aa];  // lane = 8 w + kb
        wave_sync();
        dft8(z);  // over a -> ka: z[ka] = Z_w[kb + 8 ka]
#pragma unroll
        for (int ka = 0; ka < 8; ++ka) {
            const int k = lo3 + 8 * ka;
            ex[64 * hi3 + k] = cmul(tw_short[k], (f2){z[ka].x, -z[ka].y});
        }
        wave_sync();
        // The eight-short overlap buffer (dsp.rs:303-330), one 1024-sample half at a time in `stage`: block w puts its 256 windowed
        // samples at 448 + 128 w; lane l takes samples t = l + 64 m (m = 0..3) of every block.  For those, the four-segment
        // pattern of the 128-input IMDCT (dsp.rs:511-532) needs just two bins, v[lo] and v[32 + lo], lo = l / 2 for even
        // lanes and (63 - l) / 2 for odd ones; which half a (w, m) falls into is the same for the whole wave and known at
        // compile time (448 + 128 w + 64 m is a multiple of 64).  Even blocks do not overlap each other, nor do odd ones: the
        // even blocks are added to the zeroed half first, the odd ones after them -- every sample is 0 + a + b as in the
        // reference (its a and b may come the other way round: the sum of two numbers does not care).
        const bool odd_lane = lane & 1;
        const int lo = odd_lane ? (63 - lane) >> 1 : lane >> 1;
        float cw[4], pw[2];  // this lane's window values: the current shape's short window at l + 64 m, the previous one's rising half
#pragma unroll
        for (int m = 0; m < 4; ++m) cw[m] = cur_short[lane + 64 * m];
        pw[0] = prev_short[lane];
        pw[1] = prev_short[lane + 64];
        float smp[8][4];  // windowed sample m of block w
#pragma unroll
        for (int w = 0; w < 8; ++w) {
            const f2 A = ex[64 * w + lo], B = ex[64 * w + 32 + lo];
            const float s0 = odd_lane ? -A.y : -B.x, s1 = odd_lane ? B.x : A.y, s2 = odd_lane ? A.x : B.y, s3 = odd_lane ? B.y : A.x;
            smp[w][0] = s0 * (w == 0 ? pw[0] : cw[0]);
            smp[w][1] = s1 * (w == 0 ? pw[1] : cw[1]);
            smp[w][2] = s2 * cw[2];
            smp[w][3] = s3 * cw[3];
        }
        wave_sync();
#pragma unroll
        for (int half = 0; half < 2; ++half) {
#pragma unroll
            for (int r = 0; r < 4; ++r) *reinterpret_cast<lds_f4 *>(stage + 4 * lane + 256 * r) = (f4){0.f, 0.f, 0.f, 0.f};
            wave_sync();
#pragma unroll
            for (int parity = 0; parity < 2; ++parity) {
#pragma unroll
                for (int w = parity; w < 8; w += 2)
#pragma unroll
                    for (int m = 0; m < 4; ++m) {
                        const int base = 448 + 128 * w + 64 * m;  // of lane 0; a multiple of 64
                        if ((base >> 10) != half) continue;
                        lds_f *at = stage + (base & 1023) + lane;
                        *at = *at + smp[w][m];
                    }
                wave_sync();
            }
            if (half == 0) read_positions(stage, lane, o);
            else read_positions(stage, lane, d);
            wave_sync();
        }
    }

    // overlap-add (dsp.rs:277-278, 333-334).  The PCM goes back through LDS too (the spectra in `ex` are used up): the caller
    // stores it behind its spectrum prefetch, so that both arms of its frame loop leave the same memory operations in flight
    // in the same order and the loop head can wait for the spectrum alone (see k_aac_synth)
    float pcm[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) pcm[i] = o[i] + dly[i];
    write_positions((lds_f *)ex, lane, pcm);
    write_positions(stage, lane, d);
    wave_sync();
    (void)out_ptr, (void)out16_ptr;
}

// OUT16: the PCM leaves as planar s16 (float_sample_to_i16 of every sample; same [off1024][1024] packing, two bytes
// per sample) -- what decode_aac_access_unit hands on (soundkit-decoder lib.rs:1793-1813) before interleaving
// ONLY_LONG: every frame of every task is OnlyLong (the host knows the windows and sorts the tasks): the loop is one
// straight line, so the compiler can count its memory operations exactly instead of draining them at a join.
template <bool OUT16, bool ONLY_LONG>
__global__ __launch_bounds__(kWavesPerBlock * 64, (ONLY_LONG && kDepth == 1) ? kWavesPerSimd + 1 : kWavesPerSimd) void k_aac_synth(SynthArgs a) {
    __shared__ f2 lds[kWavesPerBlock][kExchange];
    __shared__ float stage_lds[ONLY_LONG ? 1 : kWavesPerBlock][ONLY_LONG ? 4 : kStage];  // rare-path staging only
    __shared__ f2 tw_tab[512];  // pre/post twiddle (dsp.rs:99-106), shared by the block's waves
    __shared__ f2 t64_tab[64];  // t64[k][n] = W64^{n k}
    __shared__ float short_win_tab[ONLY_LONG ? 2 : 512];  // the sine and the KBD short window (eight-short frames only)
    __shared__ f2 short_tw_tab[ONLY_LONG ? 1 : 128];       // w64[64] | tw_short[64]

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    for (int i = threadIdx.x; i < 512; i += kWavesPerBlock * 64) tw_tab[i] = reinterpret_cast<const f2 *>(a.t.tw_long)[i];
    if (threadIdx.x < 64) t64_tab[threadIdx.x] = reinterpret_cast<const f2 *>(a.t.w64)[((threadIdx.x >> 3) * (threadIdx.x & 7)) & 63];
    if (!ONLY_LONG) {
        for (int i = threadIdx.x; i < 512; i += kWavesPerBlock * 64) short_win_tab[i] = a.t.win[4096 + i];
        if (threadIdx.x < 128)
            short_tw_tab[threadIdx.x] = threadIdx.x < 64 ? reinterpret_cast<const f2 *>(a.t.w64)[threadIdx.x] : reinterpret_cast<const f2 *>(a.t.tw_short)[threadIdx.x - 64];
    }
    __syncthreads();

    const uint32_t task_id = blockIdx.x * kWavesPerBlock + wave;
    if (task_id >= a.n_tasks) return;
    lds_f2 *ex = (lds_f2 *)lds[wave];
    lds_f *stage = (lds_f *)stage_lds[ONLY_LONG ? 0 : wave];
    const lds_f2 *tw_lds = (const lds_f2 *)tw_tab;
    const lds_f2 *t64 = (const lds_f2 *)t64_tab;

    // everything read from the schedule is wave-uniform: say so, so that addresses keep an SGPR base
    // and the per-frame branches are scalar
    const SynthTask task_v = a.tasks[task_id];
    const uint32_t count = __builtin_amdgcn_readfirstlane(task_v.count);
    const uint32_t state = __builtin_amdgcn_readfirstlane(task_v.state);
    if (count == 0) return;
    const const_entries entries = as_constant(a.entries + __builtin_amdgcn_readfirstlane(task_v.begin));

    const int hi3 = lane >> 3, lo3 = lane & 7;
    const f2 base2 = reinterpret_cast<const f2 *>(a.t.w512)[hi3 * lo3];

    // carried state ---------------------------------------------------------------
    float *delay_ptr = a.delay + (size_t)state * 1024;
    int prev_shape = __builtin_amdgcn_readfirstlane((int)a.prev_shape[state]);
    float dly[16];
    {
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            const int j = 4 * lane + 256 * r;
            const f4 f = *reinterpret_cast<const f4 *>(delay_ptr + j);
            const f4 m = *reinterpret_cast<const f4 *>(delay_ptr + 1020 - j);
            dly[8 * r + 0] = f.x; dly[8 * r + 1] = f.y; dly[8 * r + 2] = f.z; dly[8 * r + 3] = f.w;
            dly[8 * r + 4] = m.x; dly[8 * r + 5] = m.y; dly[8 * r + 6] = m.z; dly[8 * r + 7] = m.w;
        }
    }

    // ---- frame loop with a register ring of kDepth prefetched spectra -------------------------------
    // One wave has (kDepth x 4 KiB) of HBM reads in flight while it transforms a frame; with two
    // waves per SIMD that is what keeps enough bytes in flight to cover HBM latency (one spectrum
    // per wave measured ~2 TB/s of reads = Little's law at ~4 us).  The ring is named statically
    // (loop unrolled by kDepth), so it stays in VGPRs.
    auto load_spectrum = [&](f2 (&xin)[8], uint32_t e) {
        const float *src = a.coeffs + (size_t)__builtin_amdgcn_readfirstlane(entries[e].off1024) * 1024 + 2 * lane;
#pragma unroll
        for (int r = 0; r < 8; ++r) xin[r] = SK_SYNTH_LOAD(reinterpret_cast<const f2 *>(src + 128 * r));
    };
    auto frame = [&](f2 (&xin)[8], uint32_t e) __attribute__((always_inline)) {
        const uint32_t ent_off = entries[e].off1024, ent_win = entries[e].win;
        const uint32_t win = __builtin_amdgcn_readfirstlane(ent_win);
        const int seq = (int)(win & 3);  // ONLY_LONG: never 2
        const int shape = (win >> 2) & 1;
        float *out_ptr = OUT16 ? nullptr : a.pcm + (size_t)__builtin_amdgcn_readfirstlane(ent_off) * 1024;
        int16_t *out16_ptr = OUT16 ? a.pcm16 + (size_t)__builtin_amdgcn_readfirstlane(ent_off) * 1024 : nullptr;

        // ---- pre-twiddle (dsp.rs:495-503): consumes xin so the next spectrum can land in it ----
        f2 z[8];
        if (ONLY_LONG || seq != 2) {
#pragma unroll
            for (int r = 0; r < 8; ++r) {  // z[64 r + lane]
                const float even = xin[r].x;
                const float odd = -__shfl(xin[7 - r].y, 63 - lane);  // X[1023 - 2 i]
                const f2 t = tw_lds[lane + 64 * r];
                z[r] = (f2){odd * t.y - even * t.x, odd * t.x + even * t.y};
            }
        } else {
            const f2 tws = short_tw_tab[ONLY_LONG ? 0 : 64 + lane];
#pragma unroll
            for (int w = 0; w < 8; ++w) {  // z_w[lane] of short block w
                const float even = xin[w].x;
                const float odd = -__shfl(xin[w].y, 63 - lane);  // X_w[127 - 2 lane]
                z[w] = (f2){odd * tws.y - even * tws.x, odd * tws.x + even * tws.y};
            }
        }
        if (ONLY_LONG || seq != 2) {
            const float *w1 = first_half_window(a.t.win, seq, prev_shape);
            const float *w2 = second_half_window(a.t.win, seq, shape);
            // Window loads go out BEFORE the prefetch: vector-memory results return in issue order, so
            // windows queued behind the next spectrum would make the epilogue wait for that HBM fetch.
            f4 w1f[2], w1m[2], w2f[2], w2m[2];
#pragma unroll
            for (int r = 0; r < 2; ++r) {
                const int j = 4 * lane + 256 * r;
                w1f[r] = *reinterpret_cast<const f4 *>(w1 + j);
                w1m[r] = *reinterpret_cast<const f4 *>(w1 + 1020 - j);
                w2f[r] = *reinterpret_cast<const f4 *>(w2 + j);
                w2m[r] = *reinterpret_cast<const f4 *>(w2 + 1020 - j);
            }
            // next spectrum in flight while this one is transformed.  Unconditional (the last frames re-read the task's
            // last spectrum): with a branch around these loads the compiler can no longer count what is outstanding when
            // the epilogue needs its windows, and waits for the prefetch itself before every store.
            {
                const uint32_t ahead = e + kDepth < count ? e + kDepth : count - 1;
                const float *src = a.coeffs + (size_t)__builtin_amdgcn_readfirstlane(entries[ahead].off1024) * 1024 + 2 * lane;
#pragma unroll
                for (int r = 0; r < 8; ++r) xin[r] = SK_SYNTH_LOAD(reinterpret_cast<const f2 *>(src + 128 * r));
            }
            fft512(z, ex, t64, base2, lane);
            // ---- post-twiddle: value = twiddle * conj(fft) (dsp.rs:512, 523) --------
#pragma unroll
            for (int j = 0; j < 8; ++j) ex[lane + 64 * j] = cmul(tw_lds[lane + 64 * j], (f2){z[j].x, -z[j].y});
            wave_sync();
            // ---- OnlyLong: window (dsp.rs:267-279) + overlap-add, one 512-sample half at a time ----
#pragma unroll
            for (int r = 0; r < 2; ++r) {
                const int q = 2 * lane + 128 * r, j = 2 * q;
                const f4 F = *reinterpret_cast<const lds_f4 *>(&ex[256 + q]);
                const f4 M = *reinterpret_cast<const lds_f4 *>(&ex[254 - q]);
                const f4 W1f = w1f[r], W1m = w1m[r], W2f = w2f[r], W2m = w2m[r];
                f4 f, m;
                // out0[j..j+3] = -F0.re, -M1.im, -F1.re, -M0.im   (dsp.rs:516, 528)
                f.x = -F.x * W1f.x + dly[8 * r + 0]; f.y = -M.w * W1f.y + dly[8 * r + 1];
                f.z = -F.z * W1f.z + dly[8 * r + 2]; f.w = -M.y * W1f.w + dly[8 * r + 3];
                // out1[508-j..511-j] = M0.im, F1.re, M1.im, F0.re (dsp.rs:517, 529)
                m.x = M.y * W1m.x + dly[8 * r + 4]; m.y = F.z * W1m.y + dly[8 * r + 5];
                m.z = M.w * W1m.z + dly[8 * r + 6]; m.w = F.x * W1m.w + dly[8 * r + 7];
                if (OUT16) {
                    SK_SYNTH_STORE(pack4_s16(f), reinterpret_cast<u2 *>(out16_ptr + j));
                    SK_SYNTH_STORE(pack4_s16(m), reinterpret_cast<u2 *>(out16_ptr + 1020 - j));
                } else {
                    SK_SYNTH_STORE(f, reinterpret_cast<f4 *>(out_ptr + j));
                    SK_SYNTH_STORE(m, reinterpret_cast<f4 *>(out_ptr + 1020 - j));
                }
                // out2[j..j+3] = F0.im, M1.re, F1.im, M0.re       (dsp.rs:518, 530)
                dly[8 * r + 0] = F.y * W2f.x; dly[8 * r + 1] = M.z * W2f.y;
                dly[8 * r + 2] = F.w * W2f.z; dly[8 * r + 3] = M.x * W2f.w;
                // out3[508-j..511-j] = M0.re, F1.im, M1.re, F0.im (dsp.rs:519, 531)
                dly[8 * r + 4] = M.x * W2m.x; dly[8 * r + 5] = F.w * W2m.y;
                dly[8 * r + 6] = M.z * W2m.z; dly[8 * r + 7] = F.y * W2m.w;
            }
            wave_sync();
        } else if constexpr (!ONLY_LONG) {
            // hand the frame over through LDS (see synth_rare_frame), then restart the prefetch
#pragma unroll
            for (int r = 0; r < 8; ++r) ex[64 * r + lane] = z[r];
            write_positions(stage, lane, dly);
            wave_sync();
            synth_rare_frame(ex, stage, tw_lds, t64, base2.x, base2.y, (const lds_f *)short_win_tab, (const lds_f2 *)short_tw_tab,
                             (const lds_f2 *)short_tw_tab + 64, out_ptr, out16_ptr, seq, prev_shape, shape, lane);
            read_positions(stage, lane, dly);
            float pcm[16];
            read_positions((const lds_f *)ex, lane, pcm);
            wave_sync();
            // as in the long arm: the next spectrum first (unconditional), then this frame's four stores
            {
                const uint32_t ahead = e + kDepth < count ? e + kDepth : count - 1;
                const float *src = a.coeffs + (size_t)__builtin_amdgcn_readfirstlane(entries[ahead].off1024) * 1024 + 2 * lane;
#pragma unroll
                for (int r = 0; r < 8; ++r) xin[r] = SK_SYNTH_LOAD(reinterpret_cast<const f2 *>(src + 128 * r));
            }
#pragma unroll
            for (int r = 0; r < 2; ++r) {
                const int j = 4 * lane + 256 * r;
                const f4 f = (f4){pcm[8 * r + 0], pcm[8 * r + 1], pcm[8 * r + 2], pcm[8 * r + 3]};
                const f4 m = (f4){pcm[8 * r + 4], pcm[8 * r + 5], pcm[8 * r + 6], pcm[8 * r + 7]};
                if (OUT16) {
                    SK_SYNTH_STORE(pack4_s16(f), reinterpret_cast<u2 *>(out16_ptr + j));
                    SK_SYNTH_STORE(pack4_s16(m), reinterpret_cast<u2 *>(out16_ptr + 1020 - j));
                } else {
                    SK_SYNTH_STORE(f, reinterpret_cast<f4 *>(out_ptr + j));
                    SK_SYNTH_STORE(m, reinterpret_cast<f4 *>(out_ptr + 1020 - j));
                }
            }
        }
        prev_shape = shape;  // decoder.rs:371
    };

    f2 ring[kDepth][8];
#pragma unroll
    for (int d = 0; d < kDepth; ++d)
        if ((uint32_t)d < count) load_spectrum(ring[d], d);
    for (uint32_t e0 = 0; e0 < count; e0 += kDepth) {
#pragma unroll
        for (int d = 0; d < kDepth; ++d)
            if (e0 + d < count) frame(ring[d], e0 + d);
    }

#pragma unroll
    for (int r = 0; r < 2; ++r) {
        const int j = 4 * lane + 256 * r;
        *reinterpret_cast<f4 *>(delay_ptr + j) =
            (f4){dly[8 * r + 0], dly[8 * r + 1], dly[8 * r + 2], dly[8 * r + 3]};
        *reinterpret_cast<f4 *>(delay_ptr + 1020 - j) =
            (f4){dly[8 * r + 4], dly[8 * r + 5], dly[8 * r + 6], dly[8 * r + 7]};
    }
    if (lane == 0) a.prev_shape[state] = (uint8_t)prev_shape;
}

// ---- OnlyLong channels, two per wave ----------------------------------------------------------------------------------------
// The walking kernel is bound by vector issue once its output is s16 (profiles/r02_pmc_aac_synth.md), and a third of what it
// issues is not arithmetic: a complex value sits in one register pair, so the packed instructions work on (re, im) and every
// multiplication by i, every twiddle product and every (1 +- i)/sqrt2 rotation needs halves swapped, broadcast or negated
// around it (v_mov / v_pk_mov / v_xor, and scalar-width adds where no packing was found).  Here a wave owns TWO channels with
// the same number of frames (normally the L and R of a stream, whose spectra are adjacent in memory) and every value is the
// pair (channel A, channel B): all arithmetic is element-wise on such pairs, a twiddle is one broadcast operand, and nothing
// is ever swapped.  Per channel the operations and their order are exactly those of k_aac_synth<.., true>, so the two kernels
// agree bit for bit and a channel's samples do not depend on whether it found a partner.
struct c2 {
    f2 re, im;  // one complex value of channel A (.x) and of channel B (.y)
};
__device__ __forceinline__ f2 splat(float v) { return (f2){v, v}; }
__device__ __forceinline__ f2 fma2(f2 a, f2 b, f2 c) { return __builtin_elementwise_fma(a, b, c); }
__device__ __forceinline__ c2 cadd(c2 a, c2 b) { return {a.re + b.re, a.im + b.im}; }
__device__ __forceinline__ c2 csub(c2 a, c2 b) { return {a.re - b.re, a.im - b.im}; }
__device__ __forceinline__ c2 mul_mi(c2 a) { return {a.im, -a.re}; }
// a * w for a twiddle common to both channels: cmul(a, w) per channel
__device__ __forceinline__ c2 cmul_w(c2 a, f2 w) {
    const f2 wr = splat(w.x), wi = splat(w.y);
    return {fma2(a.re, wr, -(a.im * wi)), fma2(a.re, wi, a.im * wr)};
}
__device__ __forceinline__ void dft8(c2 (&x)[8]) {
    const f2 h = splat(0.70710678118654752440f);
    c2 t0 = cadd(x[0], x[4]), t1 = csub(x[0], x[4]);
    c2 t2 = cadd(x[2], x[6]), t3 = mul_mi(csub(x[2], x[6]));
    c2 e0 = cadd(t0, t2), e2 = csub(t0, t2), e1 = cadd(t1, t3), e3 = csub(t1, t3);
    c2 u0 = cadd(x[1], x[5]), u1 = csub(x[1], x[5]);
    c2 u2 = cadd(x[3], x[7]), u3 = mul_mi(csub(x[3], x[7]));
    c2 o0 = cadd(u0, u2), o2 = csub(u0, u2), o1 = cadd(u1, u3), o3 = csub(u1, u3);
    c2 w1 = {(o1.re + o1.im) * h, (o1.im - o1.re) * h};
    c2 w2 = mul_mi(o2);
    c2 w3 = {(o3.im - o3.re) * h, -((o3.re + o3.im) * h)};
    x[0] = cadd(e0, o0); x[4] = csub(e0, o0);
    x[1] = cadd(e1, w1); x[5] = csub(e1, w1);
    x[2] = cadd(e2, w2); x[6] = csub(e2, w2);
    x[3] = cadd(e3, w3); x[7] = csub(e3, w3);
}
// exchange slots hold (re A, re B, im A, im B); strides are in 16-byte slots: 66 and 72 keep both transposes free of bank
// conflicts (16 lanes of a ds_*_b128 land on 16 different slots modulo 16)
constexpr int kPairExchange = 576;
__device__ __forceinline__ f4 as_slot(c2 v) { return (f4){v.re.x, v.re.y, v.im.x, v.im.y}; }
__device__ __forceinline__ c2 from_slot(f4 v) { return {(f2){v.x, v.y}, (f2){v.z, v.w}}; }
__device__ __forceinline__ void fft512(c2 (&z)[8], lds_f4 *ex, const lds_f2 *t64, f2 base2, int lane) {
    const int hi3 = lane >> 3, lo3 = lane & 7;
    dft8(z);
#pragma unroll
    for (int k = 1; k < 8; ++k) z[k] = cmul_w(z[k], t64[8 * k + hi3]);
#pragma unroll
    for (int k = 0; k < 8; ++k) ex[k * 66 + lane] = as_slot(z[k]);
    wave_sync();
#pragma unroll
    for (int n2 = 0; n2 < 8; ++n2) z[n2] = cmul_w(from_slot(ex[lo3 * 66 + 8 * n2 + hi3]), base2);
    wave_sync();
    dft8(z);
#pragma unroll
    for (int k = 1; k < 8; ++k) z[k] = cmul_w(z[k], t64[8 * k + hi3]);
#pragma unroll
    for (int k = 0; k < 8; ++k) ex[hi3 * 72 + k * 8 + lo3] = as_slot(z[k]);
    wave_sync();
#pragma unroll
    for (int n3 = 0; n3 < 8; ++n3) z[n3] = from_slot(ex[n3 * 72 + lane]);
    wave_sync();
    dft8(z);
}

// this lane's 16 positions (4l+256r+e, 1020-4l-256r+e) of a 1024-element LDS array of (channel A, channel B) pairs
__device__ __forceinline__ void read_positions2(const lds_f2 *buf, int lane, f2 (&v)[16]) {
#pragma unroll
    for (int r = 0; r < 2; ++r) {
        const int j = 4 * lane + 256 * r;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const f4 f = *reinterpret_cast<const lds_f4 *>(buf + j + 2 * h);
            const f4 m = *reinterpret_cast<const lds_f4 *>(buf + 1020 - j + 2 * h);
            v[8 * r + 2 * h] = (f2){f.x, f.y}; v[8 * r + 2 * h + 1] = (f2){f.z, f.w};
            v[8 * r + 4 + 2 * h] = (f2){m.x, m.y}; v[8 * r + 4 + 2 * h + 1] = (f2){m.z, m.w};
        }
    }
}
__device__ __forceinline__ void write_positions2(lds_f2 *buf, int lane, const f2 (&v)[16]) {
#pragma unroll
    for (int r = 0; r < 2; ++r) {
        const int j = 4 * lane + 256 * r;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            *reinterpret_cast<lds_f4 *>(buf + j + 2 * h) = (f4){v[8 * r + 2 * h].x, v[8 * r + 2 * h].y, v[8 * r + 2 * h + 1].x, v[8 * r + 2 * h + 1].y};
            *reinterpret_cast<lds_f4 *>(buf + 1020 - j + 2 * h) =
                (f4){v[8 * r + 4 + 2 * h].x, v[8 * r + 4 + 2 * h].y, v[8 * r + 4 + 2 * h + 1].x, v[8 * r + 4 + 2 * h + 1].y};
        }
    }
}

// synth_rare_frame for the two channels of a pair at once (dsp.rs:284-338), every value the pair (channel A, channel B) as
// in the pair kernel's long arm: per channel the operations of synth_rare_frame in its order, so the results are its
// results bit for bit.  Out of line for the same reason; vector state through the wave's LDS:
//   in : ex[64 w + lane] = pre-twiddled element `lane` of short block w (slot = re A, re B, im A, im B);
//        stage = the two overlap delays as pairs at this lane's 16 positions (write_positions2)
//   out: ex, viewed as 1024 pairs, = PCM; stage = new delays (both at this lane's 16 positions)
// short_win, w64, tw_short: the two 256-entry short windows (sine, KBD), the 64 roots and the 64 short twiddles in LDS, copied
// there once per workgroup -- read from global memory here they were three trips to memory in the middle of every EightShort frame
__device__ __attribute__((noinline)) void synth_rare_pair(lds_f4 *ex, lds_f2 *stage, const lds_f *short_win, const lds_f2 *w64, const lds_f2 *tw_short,
                                                          int prev_a, int shape_a, int prev_b, int shape_b, int lane) {
    const int hi3 = lane >> 3, lo3 = lane & 7;
    const lds_f *prev_short_a = short_win + 256 * prev_a, *cur_short_a = short_win + 256 * shape_a;
    const lds_f *prev_short_b = short_win + 256 * prev_b, *cur_short_b = short_win + 256 * shape_b;
    c2 z[8];
    f2 dly[16];
    read_positions2(stage, lane, dly);
    // eight 64-point FFTs, lane = 8 w + a, n = a + 8 b, k = kb + 8 ka (the blocks lie in ex as the first pass wants them)
#pragma unroll
    for (int b2 = 0; b2 < 8; ++b2) z[b2] = from_slot(ex[64 * hi3 + lo3 + 8 * b2]);
    wave_sync();
    dft8(z);  // over b -> kb
#pragma unroll
    for (int kb = 1; kb < 8; ++kb) z[kb] = cmul_w(z[kb], w64[lo3 * kb]);
#pragma unroll
    for (int kb = 0; kb < 8; ++kb) ex[64 * hi3 + 8 * kb + lo3] = as_slot(z[kb]);
    wave_sync();
#pragma unroll
    for (int aa = 0; aa < 8; ++aa) z[aa] = from_slot(ex[64 * hi3 + 8 * lo3 + aa]);  // lane = 8 w + kb
    wave_sync();
    dft8(z);  // over a -> ka: z[ka] = Z_w[kb + 8 ka]
#pragma unroll
    for (int ka = 0; ka < 8; ++ka) {
        const int k = lo3 + 8 * ka;
        const f2 t = tw_short[k];
        const f2 tx = splat(t.x), ty = splat(t.y);
        const c2 v = {fma2(tx, z[ka].re, -(ty * -z[ka].im)), fma2(tx, -z[ka].im, ty * z[ka].re)};  // tw * conj(z), per channel
        ex[64 * hi3 + k] = as_slot(v);
    }
    wave_sync();
    // the eight-short overlap buffer (dsp.rs:303-330), one 1024-sample half at a time in `stage`: see synth_rare_frame
    const bool odd_lane = lane & 1;
    const int lo = odd_lane ? (63 - lane) >> 1 : lane >> 1;
    f2 cw[4], pw[2];
#pragma unroll
    for (int m = 0; m < 4; ++m) cw[m] = (f2){cur_short_a[lane + 64 * m], cur_short_b[lane + 64 * m]};
    pw[0] = (f2){prev_short_a[lane], prev_short_b[lane]};
    pw[1] = (f2){prev_short_a[lane + 64], prev_short_b[lane + 64]};
    f2 smp[8][4];
#pragma unroll
    for (int w = 0; w < 8; ++w) {
        const c2 A = from_slot(ex[64 * w + lo]), B = from_slot(ex[64 * w + 32 + lo]);
        const f2 s0 = odd_lane ? -A.im : -B.re, s1 = odd_lane ? B.re : A.im, s2 = odd_lane ? A.re : B.im, s3 = odd_lane ? B.im : A.re;
        smp[w][0] = s0 * (w == 0 ? pw[0] : cw[0]);
        smp[w][1] = s1 * (w == 0 ? pw[1] : cw[1]);
        smp[w][2] = s2 * cw[2];
        smp[w][3] = s3 * cw[3];
    }
    wave_sync();
    f2 o[16], d[16];
#pragma unroll
    for (int half = 0; half < 2; ++half) {
#pragma unroll
        for (int r = 0; r < 8; ++r) *reinterpret_cast<lds_f4 *>(stage + 2 * lane + 128 * r) = (f4){0.f, 0.f, 0.f, 0.f};
        wave_sync();
#pragma unroll
        for (int parity = 0; parity < 2; ++parity) {
#pragma unroll
            for (int w = parity; w < 8; w += 2)
#pragma unroll
                for (int m = 0; m < 4; ++m) {
                    const int base = 448 + 128 * w + 64 * m;  // of lane 0; a multiple of 64
                    if ((base >> 10) != half) continue;
                    lds_f2 *at = stage + (base & 1023) + lane;
                    *at = *at + smp[w][m];
                }
            wave_sync();
        }
        if (half == 0) read_positions2(stage, lane, o);
        else read_positions2(stage, lane, d);
        wave_sync();
    }
    f2 pcm[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) pcm[i] = o[i] + dly[i];
    write_positions2((lds_f2 *)ex, lane, pcm);
    write_positions2(stage, lane, d);
    wave_sync();
}

// WITH_SHORT: frames in which BOTH channels are EightShort take a wave-uniform arm (synth_rare_frame per channel, as the
// one-channel kernel calls it); the host pairs only channels whose EightShort frames coincide.  The long arm is the same
// code in both instantiations, and per channel both arms do what k_aac_synth does: bit-identical results.
template <bool OUT16, bool WITH_SHORT>
__global__ __launch_bounds__(kWavesPerBlock * 64, 2) void k_aac_synth_pair(SynthArgs a) {
    __shared__ f4 lds[kWavesPerBlock][kPairExchange];
    __shared__ f2 stage_lds[WITH_SHORT ? kWavesPerBlock : 1][WITH_SHORT ? kStage : 2];  // eight-short arm only: 1024 (A, B) pairs per wave
    __shared__ f2 tw_tab[512];
    __shared__ f2 t64_tab[64];
    __shared__ float short_win_tab[WITH_SHORT ? 512 : 2];  // the sine and the KBD short window (eight-short arm only)
    __shared__ f2 short_tw_tab[WITH_SHORT ? 128 : 1];       // w64[64] | tw_short[64] (eight-short arm only)

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    for (int i = threadIdx.x; i < 512; i += kWavesPerBlock * 64) tw_tab[i] = reinterpret_cast<const f2 *>(a.t.tw_long)[i];
    if (threadIdx.x < 64) t64_tab[threadIdx.x] = reinterpret_cast<const f2 *>(a.t.w64)[((threadIdx.x >> 3) * (threadIdx.x & 7)) & 63];
    if (WITH_SHORT) {
        for (int i = threadIdx.x; i < 512; i += kWavesPerBlock * 64) short_win_tab[i] = a.t.win[4096 + i];
        if (threadIdx.x < 128)
            short_tw_tab[threadIdx.x] = threadIdx.x < 64 ? reinterpret_cast<const f2 *>(a.t.w64)[threadIdx.x] : reinterpret_cast<const f2 *>(a.t.tw_short)[threadIdx.x - 64];
    }
    __syncthreads();

    const uint32_t pair_id = blockIdx.x * kWavesPerBlock + wave;
    if (2 * pair_id + 1 >= a.n_tasks) return;  // n_tasks is even: tasks 2p and 2p + 1 have the same count
    lds_f4 *ex = (lds_f4 *)lds[wave];
    const lds_f2 *tw_lds = (const lds_f2 *)tw_tab;
    const lds_f2 *t64 = (const lds_f2 *)t64_tab;

    const SynthTask task_a = a.tasks[2 * pair_id], task_b = a.tasks[2 * pair_id + 1];
    const uint32_t count = __builtin_amdgcn_readfirstlane(task_a.count);
    if (count == 0) return;
    const uint32_t state_a = __builtin_amdgcn_readfirstlane(task_a.state), state_b = __builtin_amdgcn_readfirstlane(task_b.state);
    const const_entries ent_a = as_constant(a.entries + __builtin_amdgcn_readfirstlane(task_a.begin));
    const const_entries ent_b = as_constant(a.entries + __builtin_amdgcn_readfirstlane(task_b.begin));

    const int hi3 = lane >> 3, lo3 = lane & 7;
    const f2 base2 = reinterpret_cast<const f2 *>(a.t.w512)[hi3 * lo3];

    float *delay_a = a.delay + (size_t)state_a * 1024, *delay_b = a.delay + (size_t)state_b * 1024;
    int prev_a = __builtin_amdgcn_readfirstlane((int)a.prev_shape[state_a]);
    int prev_b = __builtin_amdgcn_readfirstlane((int)a.prev_shape[state_b]);
    float dly_a[16], dly_b[16];
    auto load_state = [&](const float *p, float (&d)[16]) {
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            const int j = 4 * lane + 256 * r;
            const f4 f = *reinterpret_cast<const f4 *>(p + j);
            const f4 m = *reinterpret_cast<const f4 *>(p + 1020 - j);
            d[8 * r + 0] = f.x; d[8 * r + 1] = f.y; d[8 * r + 2] = f.z; d[8 * r + 3] = f.w;
            d[8 * r + 4] = m.x; d[8 * r + 5] = m.y; d[8 * r + 6] = m.z; d[8 * r + 7] = m.w;
        }
    };
    load_state(delay_a, dly_a);
    load_state(delay_b, dly_b);

    auto load_spectrum = [&](f2 (&xin)[8], const_entries entries, uint32_t e) {
        const float *src = a.coeffs + (size_t)__builtin_amdgcn_readfirstlane(entries[e].off1024) * 1024 + 2 * lane;
#pragma unroll
        for (int r = 0; r < 8; ++r) xin[r] = SK_SYNTH_LOAD(reinterpret_cast<const f2 *>(src + 128 * r));
    };
    struct Win {
        f4 w1f[2], w1m[2], w2f[2], w2m[2];
    };
    auto load_windows = [&](Win &w, int seq, int prev_shape, int shape) {
        const float *w1 = first_half_window(a.t.win, seq, prev_shape);
        const float *w2 = second_half_window(a.t.win, seq, shape);
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            const int j = 4 * lane + 256 * r;
            w.w1f[r] = *reinterpret_cast<const f4 *>(w1 + j);
            w.w1m[r] = *reinterpret_cast<const f4 *>(w1 + 1020 - j);
            w.w2f[r] = *reinterpret_cast<const f4 *>(w2 + j);
            w.w2m[r] = *reinterpret_cast<const f4 *>(w2 + 1020 - j);
        }
    };
    // window + overlap-add + store of one channel's half r from its four post-twiddled bins (dsp.rs:267-279, 516-531)
    auto emit = [&](int r, f2 F0, f2 F1, f2 M0, f2 M1, const Win &w, float (&dly)[16], float *out_ptr, int16_t *out16_ptr) {
        const int j = 4 * lane + 256 * r;  // = 2 q, q = 2 lane + 128 r the first of the lane's two bins
        const f4 W1f = w.w1f[r], W1m = w.w1m[r], W2f = w.w2f[r], W2m = w.w2m[r];
        f4 f, m;
        f.x = -F0.x * W1f.x + dly[8 * r + 0]; f.y = -M1.y * W1f.y + dly[8 * r + 1];
        f.z = -F1.x * W1f.z + dly[8 * r + 2]; f.w = -M0.y * W1f.w + dly[8 * r + 3];
        m.x = M0.y * W1m.x + dly[8 * r + 4]; m.y = F1.x * W1m.y + dly[8 * r + 5];
        m.z = M1.y * W1m.z + dly[8 * r + 6]; m.w = F0.x * W1m.w + dly[8 * r + 7];
        if (OUT16) {
            SK_SYNTH_STORE(pack4_s16(f), reinterpret_cast<u2 *>(out16_ptr + j));
            SK_SYNTH_STORE(pack4_s16(m), reinterpret_cast<u2 *>(out16_ptr + 1020 - j));
        } else {
            SK_SYNTH_STORE(f, reinterpret_cast<f4 *>(out_ptr + j));
            SK_SYNTH_STORE(m, reinterpret_cast<f4 *>(out_ptr + 1020 - j));
        }
        dly[8 * r + 0] = F0.y * W2f.x; dly[8 * r + 1] = M1.x * W2f.y;
        dly[8 * r + 2] = F1.y * W2f.z; dly[8 * r + 3] = M0.x * W2f.w;
        dly[8 * r + 4] = M0.x * W2m.x; dly[8 * r + 5] = F1.y * W2m.y;
        dly[8 * r + 6] = M1.x * W2m.z; dly[8 * r + 7] = F0.y * W2m.w;
    };

    lds_f2 *stage = (lds_f2 *)stage_lds[WITH_SHORT ? wave : 0];
    auto store_pcm = [&](const float (&pcm)[16], uint32_t off) __attribute__((always_inline)) {
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            const int j = 4 * lane + 256 * r;
            const f4 f = (f4){pcm[8 * r + 0], pcm[8 * r + 1], pcm[8 * r + 2], pcm[8 * r + 3]};
            const f4 m = (f4){pcm[8 * r + 4], pcm[8 * r + 5], pcm[8 * r + 6], pcm[8 * r + 7]};
            if (OUT16) {
                SK_SYNTH_STORE(pack4_s16(f), reinterpret_cast<u2 *>(a.pcm16 + (size_t)off * 1024 + j));
                SK_SYNTH_STORE(pack4_s16(m), reinterpret_cast<u2 *>(a.pcm16 + (size_t)off * 1024 + 1020 - j));
            } else {
                SK_SYNTH_STORE(f, reinterpret_cast<f4 *>(a.pcm + (size_t)off * 1024 + j));
                SK_SYNTH_STORE(m, reinterpret_cast<f4 *>(a.pcm + (size_t)off * 1024 + 1020 - j));
            }
        }
    };

    f2 xa[8], xb[8];
    load_spectrum(xa, ent_a, 0);
    load_spectrum(xb, ent_b, 0);
    for (uint32_t e = 0; e < count; ++e) {
        const uint32_t off_a = __builtin_amdgcn_readfirstlane(ent_a[e].off1024), off_b = __builtin_amdgcn_readfirstlane(ent_b[e].off1024);
        const uint32_t win_a = __builtin_amdgcn_readfirstlane(ent_a[e].win), win_b = __builtin_amdgcn_readfirstlane(ent_b[e].win);
        const int shape_a = (win_a >> 2) & 1, shape_b = (win_b >> 2) & 1;
        if constexpr (WITH_SHORT) {
            if ((win_a & 3u) == 2u) {  // wave-uniform; channel B's frame is EightShort too (the host pairs no others)
                // both channels' EightShort frame (dsp.rs:284-338) through synth_rare_pair: pre-twiddled blocks and the two
                // overlaps in through the wave's LDS, PCM and new overlaps back the same way
                {
                    const f2 tws = short_tw_tab[64 + lane];
                    const f2 tx = splat(tws.x), ty = splat(tws.y);
#pragma unroll
                    for (int w = 0; w < 8; ++w) {  // z_w[lane] of short block w (dsp.rs:495-503 with input_len 128)
                        const f2 even = (f2){xa[w].x, xb[w].x};
                        const f2 odd = (f2){-__shfl(xa[w].y, 63 - lane), -__shfl(xb[w].y, 63 - lane)};
                        const c2 zw = {odd * ty - even * tx, odd * tx + even * ty};
                        ex[64 * w + lane] = as_slot(zw);
                    }
                    f2 both[16];
#pragma unroll
                    for (int i = 0; i < 16; ++i) both[i] = (f2){dly_a[i], dly_b[i]};
                    write_positions2(stage, lane, both);
                    wave_sync();
                }
                synth_rare_pair(ex, stage, (const lds_f *)short_win_tab, (const lds_f2 *)short_tw_tab, (const lds_f2 *)short_tw_tab + 64, prev_a, shape_a, prev_b,
                                shape_b, lane);
                float pcm_a[16], pcm_b[16];
                {
                    f2 both[16];
                    read_positions2(stage, lane, both);
#pragma unroll
                    for (int i = 0; i < 16; ++i) {
                        dly_a[i] = both[i].x;
                        dly_b[i] = both[i].y;
                    }
                    read_positions2((const lds_f2 *)ex, lane, both);
#pragma unroll
                    for (int i = 0; i < 16; ++i) {
                        pcm_a[i] = both[i].x;
                        pcm_b[i] = both[i].y;
                    }
                    wave_sync();
                }
                // as in the long arm: the next spectra first (unconditional), then this frame's stores.  (Issued in front of
                // the call instead, they gain nothing: the compiler drains the memory pipeline at a call, so the wave would wait
                // for them there -- measured 0.877-0.886 ms against 0.870 on the mixed batch; inlining the routine spills in the
                // long arm: 0.946 against 0.896, profiles/r03_ab_mix.md.)
                const uint32_t ahead = e + 1 < count ? e + 1 : count - 1;
                load_spectrum(xa, ent_a, ahead);
                load_spectrum(xb, ent_b, ahead);
                store_pcm(pcm_a, off_a);
                store_pcm(pcm_b, off_b);
                prev_a = shape_a;
                prev_b = shape_b;
                continue;
            }
        }
        // ---- pre-twiddle (dsp.rs:495-503) of both channels ----
        c2 z[8];
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            const f2 even = (f2){xa[r].x, xb[r].x};
            const f2 odd = (f2){-__shfl(xa[7 - r].y, 63 - lane), -__shfl(xb[7 - r].y, 63 - lane)};
            const f2 t = tw_lds[lane + 64 * r];
            const f2 tx = splat(t.x), ty = splat(t.y);
            z[r] = {odd * ty - even * tx, odd * tx + even * ty};
        }
        Win wa, wb;  // before the prefetch: vector-memory results return in issue order
        load_windows(wa, (int)(win_a & 3), prev_a, shape_a);
        load_windows(wb, (int)(win_b & 3), prev_b, shape_b);
        {
            const uint32_t ahead = e + 1 < count ? e + 1 : count - 1;  // unconditional, as in the one-channel kernel
            load_spectrum(xa, ent_a, ahead);
            load_spectrum(xb, ent_b, ahead);
        }
        fft512(z, ex, t64, base2, lane);
        // ---- post-twiddle: value = twiddle * conj(fft) (dsp.rs:512, 523): cmul(tw, conj z) per channel ----
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const f2 t = tw_lds[lane + 64 * j];
            const f2 tx = splat(t.x), ty = splat(t.y);
            const c2 v = {fma2(tx, z[j].re, -(ty * -z[j].im)), fma2(tx, -z[j].im, ty * z[j].re)};
            ex[lane + 64 * j] = as_slot(v);
        }
        wave_sync();
        float *out_a = OUT16 ? nullptr : a.pcm + (size_t)off_a * 1024, *out_b = OUT16 ? nullptr : a.pcm + (size_t)off_b * 1024;
        int16_t *out16_a = OUT16 ? a.pcm16 + (size_t)off_a * 1024 : nullptr, *out16_b = OUT16 ? a.pcm16 + (size_t)off_b * 1024 : nullptr;
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            const int q = 2 * lane + 128 * r;
            const f4 F0 = ex[256 + q], F1 = ex[257 + q], M0 = ex[254 - q], M1 = ex[255 - q];
            emit(r, (f2){F0.x, F0.z}, (f2){F1.x, F1.z}, (f2){M0.x, M0.z}, (f2){M1.x, M1.z}, wa, dly_a, out_a, out16_a);
            emit(r, (f2){F0.y, F0.w}, (f2){F1.y, F1.w}, (f2){M0.y, M0.w}, (f2){M1.y, M1.w}, wb, dly_b, out_b, out16_b);
        }
        wave_sync();
        prev_a = shape_a;  // decoder.rs:371
        prev_b = shape_b;
    }
    auto store_state = [&](float *p, const float (&d)[16]) {
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            const int j = 4 * lane + 256 * r;
            *reinterpret_cast<f4 *>(p + j) = (f4){d[8 * r + 0], d[8 * r + 1], d[8 * r + 2], d[8 * r + 3]};
            *reinterpret_cast<f4 *>(p + 1020 - j) = (f4){d[8 * r + 4], d[8 * r + 5], d[8 * r + 6], d[8 * r + 7]};
        }
    };
    store_state(delay_a, dly_a);
    store_state(delay_b, dly_b);
    if (lane == 0) {
        a.prev_shape[state_a] = (uint8_t)prev_a;
        a.prev_shape[state_b] = (uint8_t)prev_b;
    }
}

// ---- the decode tail in one kernel: synthesis -> s16 -> 48 to 16 kHz FIR -> interleaved s16 ----------------------------------
// decode_aac_access_unit + apply_output_options (soundkit-decoder lib.rs:1793-1813, 3324-3456) for every frame of a launch
// without the s16 PCM ever crossing HBM: what k_aac_synth<true, true> stores (float_sample_to_i16 of every sample) goes,
// split into the FIR's two f16 planes, into a 2048-sample ring in LDS instead, and the wave runs the FIR on its own
// channel whenever 768 more samples (256 outputs) are complete.  Same arithmetic as the two kernels it replaces, in the
// same order -- the synthesis is k_aac_synth's code, the FIR fir_bf16.hip's f16 form (tap fragments, sample split and the
// order of the matrix instructions per output: windows ascending, x1h1 | x1h2, x2h1) -- so the result is theirs bit for bit.
//
//   D[i][j] += A[i][k] * B[k][j]   v_mfma_f32_16x16x32_f16:  i = output within a block of 16, k = 32 samples of a window,
//   j = sixteen blocks of 16 outputs, 48 samples apart, of the SAME channel (fir_bf16.hip: sixteen channel rows): a tile is
//   256 consecutive outputs of one channel, its B operand a gather from the ring (sample 768 T - 128 + 48 j + 32 s + 8 q).
//
// One workgroup = the two tasks of a pair (k_aac_synth_pair's pairing: L and R of a stream, or two mono streams of equal
// length), one wave each.  For a stereo stream the two waves swap their packed results through LDS behind a barrier and
// each stores one half of the tile's interleaved frames (512 contiguous bytes); mono streams store their own.
constexpr int kTailRing = 2048;  // samples per plane and wave: 255 of history + at most 767 waiting + one new frame
constexpr int kTailDepth = 2;  // spectra in flight per wave (1 and 3 measured the same within 2 %)

__global__ __launch_bounds__(128, 2) void k_aac_tail(TailArgs ta) {
    const SynthArgs &a = ta.s;
    __shared__ f2 lds[2][kExchange];
    __shared__ f2 tw_tab[512];
    __shared__ f2 t64_tab[64];
    __shared__ __attribute__((aligned(16))) uint16_t ring_lds[2][2][kTailRing];  // [wave][plane][sample & 2047]
    __shared__ __attribute__((aligned(16))) uint32_t xchg[2][2][128];            // [tile parity][wave][2 x lane]

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    for (int i = threadIdx.x; i < 512; i += 128) tw_tab[i] = reinterpret_cast<const f2 *>(a.t.tw_long)[i];
    if (threadIdx.x < 64) t64_tab[threadIdx.x] = reinterpret_cast<const f2 *>(a.t.w64)[((threadIdx.x >> 3) * (threadIdx.x & 7)) & 63];
    // the ring starts as silence: the filter's history in front of the launch's first sample
    for (int i = threadIdx.x; i < 2 * 2 * kTailRing / 8; i += 128) reinterpret_cast<sk_u32x4 *>(&ring_lds[0][0][0])[i] = (sk_u32x4){0u, 0u, 0u, 0u};
    __syncthreads();

    const uint32_t pair_id = blockIdx.x;  // a.n_tasks is even: tasks 2p and 2p + 1 have the same count
    lds_f2 *ex = (lds_f2 *)lds[wave];
    const lds_f2 *tw_lds = (const lds_f2 *)tw_tab;
    const lds_f2 *t64 = (const lds_f2 *)t64_tab;
    unsigned char *ring = reinterpret_cast<unsigned char *>(&ring_lds[wave][0][0]);  // plane 1 at + 2 * kTailRing bytes

    const SynthTask task_v = a.tasks[2 * pair_id + (uint32_t)wave];
    const SynthTask other_v = a.tasks[2 * pair_id + (uint32_t)(wave ^ 1)];
    const uint32_t count = __builtin_amdgcn_readfirstlane(task_v.count);
    const uint32_t state = __builtin_amdgcn_readfirstlane(task_v.state);
    const uint32_t other_state = __builtin_amdgcn_readfirstlane(other_v.state);
    if (count == 0) return;  // both waves of the pair: the counts are equal
    const const_entries entries = as_constant(a.entries + __builtin_amdgcn_readfirstlane(task_v.begin));
    const bool stereo = (state ^ other_state) == 1u;  // L and R of one stream: interleaved output
    // where this channel's output goes: its stream's row in the caller's layout, from the position of its first frame
    const uint64_t first_elem = (uint64_t)__builtin_amdgcn_readfirstlane(entries[0].off1024) * 1024u;
    const uint32_t out_row = (uint32_t)(first_elem / ta.stream_stride);
    const uint32_t n_out = (1024u * count - 130u) / 3u;  // sk_downsample_48k_16k_out_frames(1024 * count): one-shot rubato SincFixedIn
    uint32_t *dst32 = reinterpret_cast<uint32_t *>(ta.out16 + (size_t)out_row * ta.out_stride * 2);  // stereo rows
    int16_t *dst16 = ta.out16 + (size_t)out_row * ta.out_stride;                                      // mono rows

    const int hi3 = lane >> 3, lo3 = lane & 7;
    const f2 base2 = reinterpret_cast<const f2 *>(a.t.w512)[hi3 * lo3];

    // the FIR's tap fragments (fir_bf16.hip: window s, plane k, this lane), held in registers for the whole launch: 68 VGPRs
    // that cost the third wave per SIMD -- fetched from the CU's L1 for every tile instead (168 VGPRs, three waves) the launch
    // took 1.75 ms against 1.22, at two waves 1.38 (profiles/r03_ab_fused.md)
    sk_u32x4 af[10][2];
    {
        const sk_u32x4 *af_src = reinterpret_cast<const sk_u32x4 *>(ta.afrag_f16) + lane;
#pragma unroll
        for (int sidx = 0; sidx < 10; ++sidx)
#pragma unroll
            for (int k = 0; k < 2; ++k) af[sidx][k] = af_src[(sidx * 2 + k) * 64];
    }

    float *delay_ptr = a.delay + (size_t)state * 1024;
    int prev_shape = __builtin_amdgcn_readfirstlane((int)a.prev_shape[state]);
    float dly[16];
#pragma unroll
    for (int r = 0; r < 2; ++r) {
        const int j = 4 * lane + 256 * r;
        const f4 f = *reinterpret_cast<const f4 *>(delay_ptr + j);
        const f4 m = *reinterpret_cast<const f4 *>(delay_ptr + 1020 - j);
        dly[8 * r + 0] = f.x; dly[8 * r + 1] = f.y; dly[8 * r + 2] = f.z; dly[8 * r + 3] = f.w;
        dly[8 * r + 4] = m.x; dly[8 * r + 5] = m.y; dly[8 * r + 6] = m.z; dly[8 * r + 7] = m.w;
    }

    // four samples of the frame at position pos (a multiple of 4) -> s16 -> the two f16 planes -> ring
    auto to_ring = [&](const f4 &v, int ring_pos) __attribute__((always_inline)) {
        const u2 p = pack4_s16(v);
        uint32_t p1a, p2a, p1b, p2b;
        dev_split_pair16_f16(p.x, p1a, p2a);
        dev_split_pair16_f16(p.y, p1b, p2b);
        *reinterpret_cast<u2 *>(ring + 2 * ring_pos) = (u2){p1a, p1b};
        *reinterpret_cast<u2 *>(ring + 2 * kTailRing + 2 * ring_pos) = (u2){p2a, p2b};
    };

    // one tile: outputs 256 T .. 256 T + 255 of this channel
    const int jq_off = 48 * (lane & 15) + 8 * (lane >> 4);
    auto fir_tile = [&](uint32_t T) __attribute__((always_inline)) {
        const int base = (int)(768u * T) - 128 + jq_off;
        sk_f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int sidx = 0; sidx < 10; ++sidx) {
            const int at = 2 * ((base + 32 * sidx) & (kTailRing - 1));
            const sk_u32x4 b1 = *reinterpret_cast<const sk_u32x4 *>(ring + at);
            acc = dev_mfma_f16(af[sidx][0], b1, acc);                                  // x1 h1
            if (kFirProductsF16[sidx] == 3) {
                const sk_u32x4 b2 = *reinterpret_cast<const sk_u32x4 *>(ring + 2 * kTailRing + at);
                acc = dev_mfma_f16(af[sidx][1], b1, acc);                              // x1 h2
                acc = dev_mfma_f16(af[sidx][0], b2, acc);                              // x2 h1
            }
        }
        // the samples went in as integers and the taps times 2^16: powers of two, they commute with every rounding
        // fir_bf16.hip's epilogue: the factor 2^-31 (integer samples, taps times 2^16) rides in the conversion's constants
        const uint32_t lo = dev_pack2_s16_scaled<31>(acc[0], acc[1]), hi = dev_pack2_s16_scaled<31>(acc[2], acc[3]);  // outputs 256 T + 16 j + 4 q + (0..3)
        if (!stereo) {
            const uint32_t m0 = 256u * T + 16u * (uint32_t)(lane & 15) + 4u * (uint32_t)(lane >> 4);
            if (m0 + 3 < n_out) {
                *reinterpret_cast<u2 *>(dst16 + m0) = (u2){lo, hi};
            } else {
                if (m0 + 0 < n_out) dst16[m0 + 0] = (int16_t)(lo & 0xffffu);
                if (m0 + 1 < n_out) dst16[m0 + 1] = (int16_t)(lo >> 16);
                if (m0 + 2 < n_out) dst16[m0 + 2] = (int16_t)(hi & 0xffffu);
            }
            return;
        }
        // stereo: both waves leave their tile in LDS (lane (j, q): 8 bytes at 8 * (16 q + j)), then wave w stores frames
        // 128 w .. 128 w + 127 of the tile, two per lane: frame m of the tile sits at bytes 8 * (16 q + j) + 2 r of a wave's
        // block with j = m >> 4, q = (m >> 2) & 3, r = m & 3
        uint32_t *mine = &xchg[T & 1u][wave][0];
        *reinterpret_cast<u2 *>(mine + 2 * lane) = (u2){lo, hi};
        __syncthreads();
        const uint32_t fm = 128u * (uint32_t)wave + 2u * (uint32_t)lane;  // this lane's first frame within the tile
        const uint32_t word = 2u * (16u * ((fm >> 2) & 3u) + (fm >> 4)) + ((fm >> 1) & 1u);
        const uint32_t l2 = xchg[T & 1u][(state & 1u) ? (wave ^ 1) : wave][word];  // two frames of the left channel
        const uint32_t r2 = xchg[T & 1u][(state & 1u) ? wave : (wave ^ 1)][word];
        const uint32_t f0 = __builtin_amdgcn_perm(r2, l2, 0x05040100u), f1 = __builtin_amdgcn_perm(r2, l2, 0x07060302u);
        const uint32_t m0 = 256u * T + fm;
        if (m0 + 1 < n_out) *reinterpret_cast<u2 *>(dst32 + m0) = (u2){f0, f1};
        else if (m0 < n_out) dst32[m0] = f0;
    };

    auto load_spectrum = [&](f2 (&xin)[8], uint32_t e) {
        const float *src = a.coeffs + (size_t)__builtin_amdgcn_readfirstlane(entries[e].off1024) * 1024 + 2 * lane;
#pragma unroll
        for (int r = 0; r < 8; ++r) xin[r] = SK_SYNTH_LOAD(reinterpret_cast<const f2 *>(src + 128 * r));
    };

    uint32_t next_tile = 0;
    // one frame; `xin` holds its spectrum and is refilled with the spectrum kTailDepth frames on (two waves per SIMD: one
    // spectrum in flight per wave left the memory pipeline idle a third of the time)
    auto frame = [&](f2 (&xin)[8], uint32_t e) __attribute__((always_inline)) {
        const uint32_t win = __builtin_amdgcn_readfirstlane(entries[e].win);
        const int seq = (int)(win & 3);  // never 2: the host sends only tasks without EightShort frames here
        const int shape = (win >> 2) & 1;
        // ---- k_aac_synth<true, true>'s frame, its stores redirected into the ring ----
        f2 z[8];
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            const float even = xin[r].x;
            const float odd = -__shfl(xin[7 - r].y, 63 - lane);
            const f2 t = tw_lds[lane + 64 * r];
            z[r] = (f2){odd * t.y - even * t.x, odd * t.x + even * t.y};
        }
        const float *w1 = first_half_window(a.t.win, seq, prev_shape);
        const float *w2 = second_half_window(a.t.win, seq, shape);
        f4 w1f[2], w1m[2], w2f[2], w2m[2];
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            const int j = 4 * lane + 256 * r;
            w1f[r] = *reinterpret_cast<const f4 *>(w1 + j);
            w1m[r] = *reinterpret_cast<const f4 *>(w1 + 1020 - j);
            w2f[r] = *reinterpret_cast<const f4 *>(w2 + j);
            w2m[r] = *reinterpret_cast<const f4 *>(w2 + 1020 - j);
        }
        load_spectrum(xin, e + kTailDepth < count ? e + kTailDepth : count - 1);
        fft512(z, ex, t64, base2, lane);
#pragma unroll
        for (int j = 0; j < 8; ++j) ex[lane + 64 * j] = cmul(tw_lds[lane + 64 * j], (f2){z[j].x, -z[j].y});
        wave_sync();
        const int frame_at = (int)((e & 1u) * 1024u);  // the frame's first sample in the ring: 1024 e mod 2048
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            const int q = 2 * lane + 128 * r, j = 2 * q;
            const f4 F = *reinterpret_cast<const lds_f4 *>(&ex[256 + q]);
            const f4 M = *reinterpret_cast<const lds_f4 *>(&ex[254 - q]);
            const f4 W1f = w1f[r], W1m = w1m[r], W2f = w2f[r], W2m = w2m[r];
            f4 f, m;
            f.x = -F.x * W1f.x + dly[8 * r + 0]; f.y = -M.w * W1f.y + dly[8 * r + 1];
            f.z = -F.z * W1f.z + dly[8 * r + 2]; f.w = -M.y * W1f.w + dly[8 * r + 3];
            m.x = M.y * W1m.x + dly[8 * r + 4]; m.y = F.z * W1m.y + dly[8 * r + 5];
            m.z = M.w * W1m.z + dly[8 * r + 6]; m.w = F.x * W1m.w + dly[8 * r + 7];
            to_ring(f, frame_at + j);
            to_ring(m, frame_at + 1020 - j);
            dly[8 * r + 0] = F.y * W2f.x; dly[8 * r + 1] = M.z * W2f.y;
            dly[8 * r + 2] = F.w * W2f.z; dly[8 * r + 3] = M.x * W2f.w;
            dly[8 * r + 4] = M.x * W2m.x; dly[8 * r + 5] = F.w * W2m.y;
            dly[8 * r + 6] = M.z * W2m.z; dly[8 * r + 7] = F.y * W2m.w;
        }
        wave_sync();
        prev_shape = shape;
        // ---- every tile whose last sample (768 T + 895) has arrived; after the last frame, every tile that holds an output ----
        const uint32_t have = 1024u * (e + 1);
        while (256u * next_tile < n_out && (768u * next_tile + 896u <= have || e + 1 == count)) {
            fir_tile(next_tile);
            ++next_tile;
        }
    };
    f2 ring_x[kTailDepth][8];
#pragma unroll
    for (int d = 0; d < kTailDepth; ++d) load_spectrum(ring_x[d], (uint32_t)d < count ? (uint32_t)d : count - 1);
    for (uint32_t e0 = 0; e0 < count; e0 += kTailDepth) {
#pragma unroll
        for (int d = 0; d < kTailDepth; ++d)
            if (e0 + (uint32_t)d < count) frame(ring_x[d], e0 + (uint32_t)d);
    }

#pragma unroll
    for (int r = 0; r < 2; ++r) {
        const int j = 4 * lane + 256 * r;
        *reinterpret_cast<f4 *>(delay_ptr + j) = (f4){dly[8 * r + 0], dly[8 * r + 1], dly[8 * r + 2], dly[8 * r + 3]};
        *reinterpret_cast<f4 *>(delay_ptr + 1020 - j) = (f4){dly[8 * r + 4], dly[8 * r + 5], dly[8 * r + 6], dly[8 * r + 7]};
    }
    if (lane == 0) a.prev_shape[state] = (uint8_t)prev_shape;
}

// planar f32 [ch][1024] -> interleaved i16 [1024][ch] with float_sample_to_i16
// (soundkit-decoder/src/lib.rs:1793-1827): non-finite -> 0, clamp +-1, f64 scale by 32768
// (negative) or 32767, round half away from zero, clamp.
__device__ __forceinline__ int16_t float_sample_to_i16(float s) {  // f64-free exact form, sk_device.h
    return (int16_t)dev_float_sample_to_i16_f32(s);
}

__global__ __launch_bounds__(256) void k_frames_to_s16(const float *planar, int16_t *out, const FrameSpan *frames,
                                                       uint32_t n) {
    const uint32_t f = blockIdx.x;
    if (f >= n) return;
    const FrameSpan sp = frames[f];
    const float *src = planar + (size_t)sp.off1024 * 1024;
    int16_t *dst = out + (size_t)sp.off1024 * 1024;
    const int i = threadIdx.x * 4;  // 4 frames of audio per thread
    if (sp.channels == 2) {
        const float4 l = *reinterpret_cast<const float4 *>(src + i);
        const float4 r = *reinterpret_cast<const float4 *>(src + 1024 + i);
        union { int16_t h[8]; uint4 v; } u;
        u.h[0] = float_sample_to_i16(l.x); u.h[1] = float_sample_to_i16(r.x);
        u.h[2] = float_sample_to_i16(l.y); u.h[3] = float_sample_to_i16(r.y);
        u.h[4] = float_sample_to_i16(l.z); u.h[5] = float_sample_to_i16(r.z);
        u.h[6] = float_sample_to_i16(l.w); u.h[7] = float_sample_to_i16(r.w);
        *reinterpret_cast<uint4 *>(dst + 2 * i) = u.v;
    } else {
        const float4 l = *reinterpret_cast<const float4 *>(src + i);
        union { int16_t h[4]; uint2 v; } u;
        u.h[0] = float_sample_to_i16(l.x); u.h[1] = float_sample_to_i16(l.y);
        u.h[2] = float_sample_to_i16(l.z); u.h[3] = float_sample_to_i16(l.w);
        *reinterpret_cast<uint2 *>(dst + i) = u.v;
    }
}

// dsp.rs:397-405 dequantize_signed_scaled with the two tables of dsp.rs:420-450
__global__ __launch_bounds__(256) void k_dequantize(const int16_t *q, const int16_t *sf, float *out, size_t n,
                                                    const float *pow43, const float *sftab) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride) {
        const int v = q[i];
        const int s = sf[i];
        float r = 0.0f;
        if (v != 0) {
            const int mag = v < 0 ? -v : v;
            const float m = mag < 8192 ? pow43[mag] : powf((float)mag, 4.0f / 3.0f);
            const float sc = (s >= -256 && s <= 511) ? sftab[s + 256] : powf(2.0f, ((float)s - 100.0f) * 0.25f);
            r = (v < 0 ? -1.0f : 1.0f) * m * sc;
        }
        out[i] = r;
    }
}

// carried state back to "just opened" for a list of streams: zero overlap, Sine as the previous window shape, the reference's
// PNS seed (dsp.rs:162-163, spectral.rs:2459) -- one workgroup per stream, one launch for all streams opened in a row
__global__ __launch_bounds__(256) void k_reset_streams(float *delay, uint8_t *shape, uint32_t *pns, const uint32_t *ids) {
    const uint32_t id = ids[blockIdx.x];
    f4 *d = reinterpret_cast<f4 *>(delay + (size_t)id * 2048);
    d[threadIdx.x] = (f4){0.f, 0.f, 0.f, 0.f};
    d[threadIdx.x + 256] = (f4){0.f, 0.f, 0.f, 0.f};
    if (threadIdx.x < 2) shape[(size_t)id * 2 + threadIdx.x] = 0;
    if (threadIdx.x == 2) pns[id] = 0x1f2e3d4cu;
}

// base[ids[b] * span .. + span) = 0 for every listed id (span a multiple of 4 floats, rows 16-byte aligned): the resampler rows and
// the Layer III state of the streams opened in a row, one launch instead of one fill per stream
__global__ __launch_bounds__(256) void k_zero_spans(float *base, const uint32_t *ids, uint32_t span) {
    f4 *d = reinterpret_cast<f4 *>(base + (size_t)ids[blockIdx.x] * span);
    for (uint32_t i = threadIdx.x; i < span / 4; i += 256) d[i] = (f4){0.f, 0.f, 0.f, 0.f};
}

}  // namespace

hipError_t launch_zero_spans(float *base, const uint32_t *ids, uint32_t n, uint32_t span_floats, hipStream_t s) {
    if (n == 0) return hipSuccess;
    if (span_floats % 4) return hipErrorInvalidValue;
    hipLaunchKernelGGL(k_zero_spans, dim3(n), dim3(256), 0, s, base, ids, span_floats);
    return hipGetLastError();
}

hipError_t launch_reset_streams(float *delay, uint8_t *shape, uint32_t *pns, const uint32_t *ids, uint32_t n, hipStream_t s) {
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(k_reset_streams, dim3(n), dim3(256), 0, s, delay, shape, pns, ids);
    return hipGetLastError();
}

hipError_t launch_aac_synth(const SynthArgs &a, hipStream_t s) {
    if (a.n_tasks == 0) return hipSuccess;
    const uint32_t blocks = (a.n_tasks + kWavesPerBlock - 1) / kWavesPerBlock;
    if (a.only_long) {
        if (a.pcm16) hipLaunchKernelGGL((k_aac_synth<true, true>), dim3(blocks), dim3(kWavesPerBlock * 64), 0, s, a);
        else hipLaunchKernelGGL((k_aac_synth<false, true>), dim3(blocks), dim3(kWavesPerBlock * 64), 0, s, a);
    } else {
        if (a.pcm16) hipLaunchKernelGGL((k_aac_synth<true, false>), dim3(blocks), dim3(kWavesPerBlock * 64), 0, s, a);
        else hipLaunchKernelGGL((k_aac_synth<false, false>), dim3(blocks), dim3(kWavesPerBlock * 64), 0, s, a);
    }
    return hipGetLastError();
}

hipError_t launch_aac_synth_pairs(const SynthArgs &a, bool with_short, hipStream_t s) {
    if (a.n_tasks < 2) return hipSuccess;
    if (a.n_tasks & 1) return hipErrorInvalidValue;
    const uint32_t pairs = a.n_tasks / 2, blocks = (pairs + kWavesPerBlock - 1) / kWavesPerBlock;
    const dim3 grid(blocks), block(kWavesPerBlock * 64);
    if (with_short) {
        if (a.pcm16) hipLaunchKernelGGL((k_aac_synth_pair<true, true>), grid, block, 0, s, a);
        else hipLaunchKernelGGL((k_aac_synth_pair<false, true>), grid, block, 0, s, a);
    } else {
        if (a.pcm16) hipLaunchKernelGGL((k_aac_synth_pair<true, false>), grid, block, 0, s, a);
        else hipLaunchKernelGGL((k_aac_synth_pair<false, false>), grid, block, 0, s, a);
    }
    return hipGetLastError();
}

hipError_t launch_aac_tail(const TailArgs &ta, hipStream_t s) {
    if (ta.s.n_tasks < 2) return hipSuccess;
    if ((ta.s.n_tasks & 1) || !ta.afrag_f16 || !ta.out16 || ta.stream_stride == 0) return hipErrorInvalidValue;
    hipLaunchKernelGGL(k_aac_tail, dim3(ta.s.n_tasks / 2), dim3(128), 0, s, ta);
    return hipGetLastError();
}

hipError_t launch_frames_to_s16(const float *planar, int16_t *out, const FrameSpan *frames, uint32_t n, hipStream_t s) {
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(k_frames_to_s16, dim3(n), dim3(256), 0, s, planar, out, frames, n);
    return hipGetLastError();
}

hipError_t launch_dequantize(const int16_t *q, const int16_t *sf, float *out, size_t n, const float *pow43,
                             const float *sftab, hipStream_t s) {
    if (n == 0) return hipSuccess;
    size_t blocks = (n + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(k_dequantize, dim3((unsigned)blocks), dim3(256), 0, s, q, sf, out, n, pow43, sftab);
    return hipGetLastError();
}

}  // namespace sk
