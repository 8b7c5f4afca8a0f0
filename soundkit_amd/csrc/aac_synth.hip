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

namespace sk {

namespace {

constexpr int kWavesPerBlock = 4;
constexpr int kExchange = 576;  // float2 per wave: max(8*68, 8*72)
constexpr int kStage = 1024;    // floats per wave: rare-path staging (transition windows, eight-short output)

__device__ __forceinline__ float2 cmul(float2 a, float2 b) {
    return make_float2(fmaf(a.x, b.x, -a.y * b.y), fmaf(a.x, b.y, a.y * b.x));
}
__device__ __forceinline__ float2 cadd(float2 a, float2 b) { return make_float2(a.x + b.x, a.y + b.y); }
__device__ __forceinline__ float2 csub(float2 a, float2 b) { return make_float2(a.x - b.x, a.y - b.y); }
// multiply by -i
__device__ __forceinline__ float2 mul_mi(float2 a) { return make_float2(a.y, -a.x); }

// forward 8-point DFT (e^{-2 pi i nk/8}), natural order in and out, all in registers
__device__ __forceinline__ void dft8(float2 (&x)[8]) {
    const float h = 0.70710678118654752440f;
    // even half: DFT4(x0,x2,x4,x6)
    float2 t0 = cadd(x[0], x[4]), t1 = csub(x[0], x[4]);
    float2 t2 = cadd(x[2], x[6]), t3 = mul_mi(csub(x[2], x[6]));
    float2 e0 = cadd(t0, t2), e2 = csub(t0, t2), e1 = cadd(t1, t3), e3 = csub(t1, t3);
    // odd half: DFT4(x1,x3,x5,x7)
    float2 u0 = cadd(x[1], x[5]), u1 = csub(x[1], x[5]);
    float2 u2 = cadd(x[3], x[7]), u3 = mul_mi(csub(x[3], x[7]));
    float2 o0 = cadd(u0, u2), o2 = csub(u0, u2), o1 = cadd(u1, u3), o3 = csub(u1, u3);
    // twiddles W8^k
    float2 w1 = make_float2((o1.x + o1.y) * h, (o1.y - o1.x) * h);   // o1 * (1-i)/sqrt2
    float2 w2 = mul_mi(o2);                                          // o2 * (-i)
    float2 w3 = make_float2((o3.y - o3.x) * h, -(o3.x + o3.y) * h);  // o3 * (-1-i)/sqrt2
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

// this lane's 16 positions (4l+256r+e, 1020-4l-256r+e) of a 1024-float LDS array
__device__ __forceinline__ void read_positions(const float *buf, int lane, float (&v)[16]) {
#pragma unroll
    for (int r = 0; r < 2; ++r) {
        const int j = 4 * lane + 256 * r;
        const float4 f = *reinterpret_cast<const float4 *>(buf + j);
        const float4 m = *reinterpret_cast<const float4 *>(buf + 1020 - j);
        v[8 * r + 0] = f.x; v[8 * r + 1] = f.y; v[8 * r + 2] = f.z; v[8 * r + 3] = f.w;
        v[8 * r + 4] = m.x; v[8 * r + 5] = m.y; v[8 * r + 6] = m.z; v[8 * r + 7] = m.w;
    }
}

// dsp.rs:353-368 / 370-387 for the two transition sequences (rare path: per-element lookups)
__device__ __forceinline__ float first_window(int seq, const float *prev_long, const float *prev_short, int i) {
    if (seq == 3) {  // LongStop
        if (i < 448) return 0.0f;
        if (i < 576) return prev_short[i - 448];
        return 1.0f;
    }
    return prev_long[i];
}
__device__ __forceinline__ float second_window(int seq, const float *cur_long, const float *cur_short, int i) {
    if (seq == 1) {  // LongStart
        if (i < 448) return 1.0f;
        if (i < 576) return cur_short[128 + i - 448];
        return 0.0f;
    }
    return cur_long[i + 1024];
}

// sample t (0..255) of a 128-input IMDCT from its post-twiddled spectrum v[0..63] (dsp.rs:511-532)
__device__ __forceinline__ float short_sample(const float2 *v, int t) {
    const int seg = t >> 6, u = t & 63;
    const bool odd = u & 1;
    const int lo = odd ? (63 - u) >> 1 : u >> 1;
    switch (seg) {
    case 0: return odd ? -v[lo].y : -v[32 + lo].x;
    case 1: return odd ? v[32 + lo].x : v[lo].y;
    case 2: return odd ? v[lo].x : v[32 + lo].y;
    default: return odd ? v[32 + lo].y : v[lo].x;
    }
}

// value at position p (0..2047) of the eight-short overlap buffer (dsp.rs:303-330)
__device__ __forceinline__ float short_buffer_value(const float2 *v, int p, const float *prev_short,
                                                    const float *cur_short) {
    const int q = p - 448;
    if (q < 0 || q >= 1152) return 0.0f;
    const int hi = q >> 7;
    float acc = 0.0f;
#pragma unroll
    for (int d = 1; d >= 0; --d) {  // block hi-1 was accumulated before block hi
        const int w = hi - d;
        if (w < 0 || w > 7) continue;
        const int t = q - 128 * w;
        const float win = (w == 0 && t < 128) ? prev_short[t] : cur_short[t];
        acc += short_sample(v + 64 * w, t) * win;
    }
    return acc;
}

__global__ __launch_bounds__(kWavesPerBlock * 64, 2) void k_aac_synth(SynthArgs a) {
    __shared__ float2 lds[kWavesPerBlock][kExchange];
    __shared__ float stage_lds[kWavesPerBlock][kStage];

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint32_t task_id = blockIdx.x * kWavesPerBlock + wave;
    if (task_id >= a.n_tasks) return;
    float2 *ex = lds[wave];
    float *stage = stage_lds[wave];

    const SynthTask task = a.tasks[task_id];
    const uint32_t count = task.count;
    if (count == 0) return;
    const SynthEntry *entries = a.entries + task.begin;

    // per-lane constants --------------------------------------------------------
    const int hi3 = lane >> 3, lo3 = lane & 7;
    float2 tw[8];   // pre/post twiddle[lane + 64 j]
    float2 tw1[8];  // W64^{(lane>>3) k}
    float2 tw2[8];  // W512^{n3 (k1 + 8 k2)}, stage-2 lane = 8 n3 + k1
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        tw[j] = a.t.tw_long[lane + 64 * j];
        tw1[j] = a.t.w64[(hi3 * j) & 63];
        tw2[j] = a.t.w512[(hi3 * (lo3 + 8 * j)) & 511];
    }

    // carried state ---------------------------------------------------------------
    float *delay_ptr = a.delay + (size_t)task.state * 1024;
    int prev_shape = a.prev_shape[task.state];
    float dly[16];
#pragma unroll
    for (int r = 0; r < 2; ++r) {
        const int j = 4 * lane + 256 * r;
        const float4 f = *reinterpret_cast<const float4 *>(delay_ptr + j);
        const float4 m = *reinterpret_cast<const float4 *>(delay_ptr + 1020 - j);
        dly[8 * r + 0] = f.x; dly[8 * r + 1] = f.y; dly[8 * r + 2] = f.z; dly[8 * r + 3] = f.w;
        dly[8 * r + 4] = m.x; dly[8 * r + 5] = m.y; dly[8 * r + 6] = m.z; dly[8 * r + 7] = m.w;
    }

    // prefetch the first spectrum ---------------------------------------------------
    float2 xin[8];
    {
        const float *src = a.coeffs + (size_t)entries[0].off1024 * 1024 + 2 * lane;
#pragma unroll
        for (int r = 0; r < 8; ++r) xin[r] = *reinterpret_cast<const float2 *>(src + 128 * r);
    }

    for (uint32_t e = 0; e < count; ++e) {
        const SynthEntry ent = entries[e];
        const int seq = ent.win & 3;
        const int shape = (ent.win >> 2) & 1;
        float *out_ptr = a.pcm + (size_t)ent.off1024 * 1024;

        const float *prev_long = a.t.win + 2048 * prev_shape;
        const float *cur_long = a.t.win + 2048 * shape;
        const float *prev_short = a.t.win + 4096 + 256 * prev_shape;
        const float *cur_short = a.t.win + 4096 + 256 * shape;

        float2 x[8];
#pragma unroll
        for (int r = 0; r < 8; ++r) x[r] = xin[r];
        if (e + 1 < count) {  // next spectrum in flight while this one is transformed
            const float *src = a.coeffs + (size_t)entries[e + 1].off1024 * 1024 + 2 * lane;
#pragma unroll
            for (int r = 0; r < 8; ++r) xin[r] = *reinterpret_cast<const float2 *>(src + 128 * r);
        }

        float o[16], d[16];  // windowed first half / second half at this lane's 16 positions

        if (seq != 2) {
            // ---- 1024-input IMDCT: pre-twiddle (dsp.rs:495-503) ------------------
            float2 z[8];
#pragma unroll
            for (int r = 0; r < 8; ++r) {  // z[64 r + lane]
                const float even = x[r].x;
                const float odd = -__shfl(x[7 - r].y, 63 - lane);  // X[1023 - 2 i]
                z[r] = make_float2(odd * tw[r].y - even * tw[r].x, odd * tw[r].x + even * tw[r].y);
            }
            // ---- 512-point FFT, n = 64 n1 + 8 n2 + n3, k = k1 + 8 k2 + 64 k3 --------
            dft8(z);  // over n1 -> k1; lane = 8 n2 + n3
#pragma unroll
            for (int k = 1; k < 8; ++k) z[k] = cmul(z[k], tw1[k]);
#pragma unroll
            for (int k = 0; k < 8; ++k) ex[k * 68 + lane] = z[k];
            wave_sync();
#pragma unroll
            for (int n2 = 0; n2 < 8; ++n2) z[n2] = ex[lo3 * 68 + 8 * n2 + hi3];  // lane = 8 n3 + k1
            wave_sync();
            dft8(z);  // over n2 -> k2
#pragma unroll
            for (int k = 0; k < 8; ++k) z[k] = cmul(z[k], tw2[k]);
#pragma unroll
            for (int k = 0; k < 8; ++k) ex[hi3 * 72 + k * 8 + lo3] = z[k];
            wave_sync();
#pragma unroll
            for (int n3 = 0; n3 < 8; ++n3) z[n3] = ex[n3 * 72 + lane];  // lane = k1 + 8 k2
            wave_sync();
            dft8(z);  // over n3 -> k3: z[j] = Z[lane + 64 j]
            // ---- post-twiddle: value = twiddle * conj(fft) (dsp.rs:512, 523) --------
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float2 c = make_float2(z[j].x, -z[j].y);
                ex[lane + 64 * j] = cmul(tw[j], c);
            }
            wave_sync();
            float im1[16], im2[16];  // imdct[i] and imdct[1024 + i] at this lane's positions
#pragma unroll
            for (int r = 0; r < 2; ++r) {
                const int q = 2 * lane + 128 * r;
                const float4 F = *reinterpret_cast<const float4 *>(&ex[256 + q]);  // v[256+q], v[257+q]
                const float4 M = *reinterpret_cast<const float4 *>(&ex[254 - q]);  // v[254-q], v[255-q]
                // out0[j..j+3], j = 2q  (dsp.rs:516, 528)
                im1[8 * r + 0] = -F.x; im1[8 * r + 1] = -M.w; im1[8 * r + 2] = -F.z; im1[8 * r + 3] = -M.y;
                // out1[508-j .. 511-j]  (dsp.rs:517, 529)
                im1[8 * r + 4] = M.y; im1[8 * r + 5] = F.z; im1[8 * r + 6] = M.w; im1[8 * r + 7] = F.x;
                // out2[j..j+3]  (dsp.rs:518, 530)
                im2[8 * r + 0] = F.y; im2[8 * r + 1] = M.z; im2[8 * r + 2] = F.w; im2[8 * r + 3] = M.x;
                // out3[508-j .. 511-j]  (dsp.rs:519, 531)
                im2[8 * r + 4] = M.x; im2[8 * r + 5] = F.w; im2[8 * r + 6] = M.z; im2[8 * r + 7] = F.y;
            }
            wave_sync();
            // ---- window (dsp.rs:267-279) ---------------------------------------------
            if (seq == 0) {
#pragma unroll
                for (int r = 0; r < 2; ++r) {
                    const int j = 4 * lane + 256 * r;
                    const float4 w1f = *reinterpret_cast<const float4 *>(prev_long + j);
                    const float4 w1m = *reinterpret_cast<const float4 *>(prev_long + 1020 - j);
                    const float4 w2f = *reinterpret_cast<const float4 *>(cur_long + 1024 + j);
                    const float4 w2m = *reinterpret_cast<const float4 *>(cur_long + 2044 - j);
                    o[8 * r + 0] = im1[8 * r + 0] * w1f.x; o[8 * r + 1] = im1[8 * r + 1] * w1f.y;
                    o[8 * r + 2] = im1[8 * r + 2] * w1f.z; o[8 * r + 3] = im1[8 * r + 3] * w1f.w;
                    o[8 * r + 4] = im1[8 * r + 4] * w1m.x; o[8 * r + 5] = im1[8 * r + 5] * w1m.y;
                    o[8 * r + 6] = im1[8 * r + 6] * w1m.z; o[8 * r + 7] = im1[8 * r + 7] * w1m.w;
                    d[8 * r + 0] = im2[8 * r + 0] * w2f.x; d[8 * r + 1] = im2[8 * r + 1] * w2f.y;
                    d[8 * r + 2] = im2[8 * r + 2] * w2f.z; d[8 * r + 3] = im2[8 * r + 3] * w2f.w;
                    d[8 * r + 4] = im2[8 * r + 4] * w2m.x; d[8 * r + 5] = im2[8 * r + 5] * w2m.y;
                    d[8 * r + 6] = im2[8 * r + 6] * w2m.z; d[8 * r + 7] = im2[8 * r + 7] * w2m.w;
                }
            } else {
                // LongStart / LongStop (rare): build the piecewise window in LDS with a rolled loop,
                // then apply it from this lane's positions
#pragma unroll 1
                for (int k = 0; k < 16; ++k) {
                    const int i = 64 * k + lane;
                    stage[i] = first_window(seq, prev_long, prev_short, i);
                }
                wave_sync();
                read_positions(stage, lane, o);
                wave_sync();
#pragma unroll 1
                for (int k = 0; k < 16; ++k) {
                    const int i = 64 * k + lane;
                    stage[i] = second_window(seq, cur_long, cur_short, i);
                }
                wave_sync();
                read_positions(stage, lane, d);
                wave_sync();
#pragma unroll
                for (int t = 0; t < 16; ++t) {
                    o[t] *= im1[t];
                    d[t] *= im2[t];
                }
            }
        } else {
            // ---- eight short windows (dsp.rs:284-338): 8 independent 64-point FFTs ---------
            const float2 tws = a.t.tw_short[lane];
#pragma unroll
            for (int w = 0; w < 8; ++w) {  // z_w[lane]
                const float even = x[w].x;
                const float odd = -__shfl(x[w].y, 63 - lane);  // X_w[127 - 2 lane]
                ex[64 * w + lane] = make_float2(odd * tws.y - even * tws.x, odd * tws.x + even * tws.y);
            }
            wave_sync();
            float2 g[8];
            // lane = 8 w + a: n = a + 8 b, k = kb + 8 ka
#pragma unroll
            for (int b = 0; b < 8; ++b) g[b] = ex[64 * hi3 + lo3 + 8 * b];
            wave_sync();
            dft8(g);  // over b -> kb
#pragma unroll
            for (int kb = 1; kb < 8; ++kb) g[kb] = cmul(g[kb], a.t.w64[lo3 * kb]);
#pragma unroll
            for (int kb = 0; kb < 8; ++kb) ex[64 * hi3 + 8 * kb + lo3] = g[kb];
            wave_sync();
#pragma unroll
            for (int aa = 0; aa < 8; ++aa) g[aa] = ex[64 * hi3 + 8 * lo3 + aa];  // lane = 8 w + kb
            wave_sync();
            dft8(g);  // over a -> ka: g[ka] = Z_w[kb + 8 ka]
#pragma unroll
            for (int ka = 0; ka < 8; ++ka) {
                const int k = lo3 + 8 * ka;
                const float2 c = make_float2(g[ka].x, -g[ka].y);
                ex[64 * hi3 + k] = cmul(a.t.tw_short[k], c);
            }
            wave_sync();
#pragma unroll 1
            for (int k = 0; k < 16; ++k) {
                const int i = 64 * k + lane;
                stage[i] = short_buffer_value(ex, i, prev_short, cur_short);
            }
            wave_sync();
            read_positions(stage, lane, o);
            wave_sync();
#pragma unroll 1
            for (int k = 0; k < 16; ++k) {
                const int i = 64 * k + lane;
                stage[i] = short_buffer_value(ex, 1024 + i, prev_short, cur_short);
            }
            wave_sync();
            read_positions(stage, lane, d);
            wave_sync();
        }

        // ---- overlap-add, state update (dsp.rs:277-278, 333-334; decoder.rs:371) -------
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            const int j = 4 * lane + 256 * r;
            float4 f, m;
            f.x = o[8 * r + 0] + dly[8 * r + 0]; f.y = o[8 * r + 1] + dly[8 * r + 1];
            f.z = o[8 * r + 2] + dly[8 * r + 2]; f.w = o[8 * r + 3] + dly[8 * r + 3];
            m.x = o[8 * r + 4] + dly[8 * r + 4]; m.y = o[8 * r + 5] + dly[8 * r + 5];
            m.z = o[8 * r + 6] + dly[8 * r + 6]; m.w = o[8 * r + 7] + dly[8 * r + 7];
            *reinterpret_cast<float4 *>(out_ptr + j) = f;
            *reinterpret_cast<float4 *>(out_ptr + 1020 - j) = m;
        }
#pragma unroll
        for (int s = 0; s < 16; ++s) dly[s] = d[s];
        prev_shape = shape;
    }

#pragma unroll
    for (int r = 0; r < 2; ++r) {
        const int j = 4 * lane + 256 * r;
        *reinterpret_cast<float4 *>(delay_ptr + j) =
            make_float4(dly[8 * r + 0], dly[8 * r + 1], dly[8 * r + 2], dly[8 * r + 3]);
        *reinterpret_cast<float4 *>(delay_ptr + 1020 - j) =
            make_float4(dly[8 * r + 4], dly[8 * r + 5], dly[8 * r + 6], dly[8 * r + 7]);
    }
    if (lane == 0) a.prev_shape[task.state] = (uint8_t)prev_shape;
}

// planar f32 [ch][1024] -> interleaved i16 [1024][ch] with float_sample_to_i16
// (soundkit-decoder/src/lib.rs:1793-1827): non-finite -> 0, clamp +-1, f64 scale by 32768
// (negative) or 32767, round half away from zero, clamp.
__device__ __forceinline__ int16_t float_sample_to_i16(float s) {
    float f = (isfinite(s)) ? fminf(fmaxf(s, -1.0f), 1.0f) : 0.0f;
    double scaled = f < 0.0f ? (double)f * 32768.0 : (double)f * 32767.0;
    int r = (int)round(scaled);
    r = r < -32768 ? -32768 : (r > 32767 ? 32767 : r);
    return (int16_t)r;
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

}  // namespace

hipError_t launch_aac_synth(const SynthArgs &a, hipStream_t s) {
    if (a.n_tasks == 0) return hipSuccess;
    const uint32_t blocks = (a.n_tasks + kWavesPerBlock - 1) / kWavesPerBlock;
    hipLaunchKernelGGL(k_aac_synth, dim3(blocks), dim3(kWavesPerBlock * 64), 0, s, a);
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
