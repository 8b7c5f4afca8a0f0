/*
 * sk_oracle.c -- CPU restatement of soundkit's per-frame decode DSP hot path.
 * TEST INFRASTRUCTURE ONLY (see sk_oracle.h).  Build: `make -C oracle`
 * (gcc -O2 -ffp-contract=off: no FMA contraction, so f32 expressions round
 * exactly as the reference's Rust, which never contracts).
 */
#include "sk_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

#define PI_F 3.14159274101257324219f /* std::f32::consts::PI */
#define PI_D 3.14159265358979323846

/* ---------------------------------------------------------------------- */
/* Rust cast semantics                                                    */
/* ---------------------------------------------------------------------- */

static inline int16_t f32_as_i16(float x) { /* `x as i16`: truncate, saturate, NaN -> 0 */
    if (x != x) return 0;
    if (x <= -32768.0f) return INT16_MIN;
    if (x >= 32767.0f) return INT16_MAX;
    return (int16_t)x;
}
static inline int32_t f32_as_i32(float x) {
    if (x != x) return 0;
    if (x <= -2147483648.0f) return INT32_MIN;
    if (x >= 2147483648.0f) return INT32_MAX;
    return (int32_t)x;
}
static inline int32_t f64_as_i32(double x) {
    if (x != x) return 0;
    if (x <= -2147483648.0) return INT32_MIN;
    if (x >= 2147483647.0) return INT32_MAX;
    return (int32_t)x;
}
static inline float f32_clamp(float x, float lo, float hi) { /* f32::clamp: NaN stays NaN */
    if (x < lo) x = lo;
    if (x > hi) x = hi;
    return x;
}

/* ---------------------------------------------------------------------- */
/* IMDCT                                                                  */
/* ---------------------------------------------------------------------- */

/* dsp.rs:453-474 */
void sko_imdct_direct_f32(const float *in, float *out, int n) {
    float nf = (float)n;
    float half_n = nf * 0.5f;
    float output_scale = (1.0f / 32768.0f) / nf;
    for (int s = 0; s < 2 * n; ++s) {
        float sample_phase = (float)s + 0.5f + half_n;
        float acc = 0.0f;
        for (int k = 0; k < n; ++k) {
            float bin_phase = (float)k + 0.5f;
            float angle = PI_F / nf * sample_phase * bin_phase;
            acc += in[k] * cosf(angle);
        }
        out[s] = acc * output_scale;
    }
}

void sko_imdct_direct_f64(const float *in, double *out, int n) {
    double scale = (1.0 / 32768.0) / (double)n;
    for (int s = 0; s < 2 * n; ++s) {
        double acc = 0.0;
        for (int k = 0; k < n; ++k) {
            /* reduce the integer phase product exactly before multiplying by pi/(4n):
             * (2s+1+n)(2k+1) mod 8n keeps the cosine argument small and exact */
            long long p = ((long long)(2 * s + 1 + n) * (long long)(2 * k + 1)) % (8LL * n);
            acc += (double)in[k] * cos(PI_D * (double)p / (4.0 * (double)n));
        }
        out[s] = acc * scale;
    }
}

/* forward complex FFT, unnormalised, e^{-2 pi i nk/N}; in-place radix-2, f32 (below).
 * Stands in for rustfft 6.4.1 (dsp.rs:107-109, 505): the reference pins only the
 * mathematical result (dsp.rs:694-723), not rustfft's rounding. */
/* per-length tables, built once (the reference builds its ImdctTransform once per decoder) */
typedef struct {
    int n;                 /* IMDCT input length */
    float twr[512], twi[512];   /* dsp.rs:99-106 */
    float fwr[256], fwi[256];   /* FFT roots e^{-2 pi i k / (n/2)}, k < n/4 */
} imdct_tables;

static void fft_forward_f32(float *re, float *im, int n, const float *fwr, const float *fwi) {
    for (int i = 1, j = 0; i < n; ++i) {
        int bit = n >> 1;
        for (; j & bit; bit >>= 1) j ^= bit;
        j ^= bit;
        if (i < j) {
            float t = re[i]; re[i] = re[j]; re[j] = t;
            t = im[i]; im[i] = im[j]; im[j] = t;
        }
    }
    for (int len = 2; len <= n; len <<= 1) {
        int half = len >> 1, step = n / len;
        for (int i = 0; i < n; i += len) {
            for (int k = 0; k < half; ++k) {
                float wr = fwr[k * step], wi = fwi[k * step];
                int a = i + k, b = a + half;
                float xr = re[b] * wr - im[b] * wi;
                float xi = re[b] * wi + im[b] * wr;
                re[b] = re[a] - xr; im[b] = im[a] - xi;
                re[a] = re[a] + xr; im[a] = im[a] + xi;
            }
        }
    }
}

static const imdct_tables *get_tables(int n) {
    static imdct_tables cache[8];
    static int used = 0;
    for (int i = 0; i < used; ++i)
        if (cache[i].n == n) return &cache[i];
    if (used == 8) used = 0;
    imdct_tables *t = &cache[used];
    float nf = (float)n;
    float output_scale = (1.0f / 32768.0f) / nf;
    float twiddle_scale = sqrtf(output_scale);
    int half = n / 2;
    for (int b = 0; b < half; ++b) {
        float angle = PI_F / nf * ((float)b + 0.125f);
        t->twr[b] = cosf(angle) * twiddle_scale;
        t->twi[b] = sinf(angle) * twiddle_scale;
    }
    for (int k = 0; k < half / 2; ++k) {
        double a = -2.0 * PI_D * (double)k / (double)half;
        t->fwr[k] = (float)cos(a);
        t->fwi[k] = (float)sin(a);
    }
    t->n = n;
    ++used;
    return t;
}

/* dsp.rs:94-119 (twiddles) + 476-535 */
int sko_imdct_fast(const float *in, float *out, int n) {
    if (n < 4 || n > 1024 || (n & (n - 1))) return -1;
    float fr[512], fi[512];
    const imdct_tables *tb = get_tables(n);
    const float *twr = tb->twr, *twi = tb->twi;
    int half = n / 2, quarter = n / 4;
    for (int i = 0; i < half; ++i) {
        float even = in[i * 2];
        float odd = -in[n - 1 - i * 2];
        fr[i] = odd * twi[i] - even * twr[i];
        fi[i] = odd * twr[i] + even * twi[i];
    }
    fft_forward_f32(fr, fi, half, tb->fwr, tb->fwi);
    float *out0 = out, *out1 = out + half, *out2 = out + 2 * half, *out3 = out + 3 * half;
    for (int i = 0; i < quarter; ++i) {
        /* value = twiddle * conj(fft) */
        float cr = fr[i], ci = -fi[i];
        float vr = twr[i] * cr - twi[i] * ci;
        float vi = twr[i] * ci + twi[i] * cr;
        int forward = i * 2, reverse = half - 1 - i * 2;
        out0[reverse] = -vi;
        out1[forward] = vi;
        out2[reverse] = vr;
        out3[forward] = vr;
    }
    for (int i = quarter; i < half; ++i) {
        float cr = fr[i], ci = -fi[i];
        float vr = twr[i] * cr - twi[i] * ci;
        float vi = twr[i] * ci + twi[i] * cr;
        int local = i - quarter;
        int forward = local * 2, reverse = half - 1 - local * 2;
        out0[forward] = -vr;
        out1[reverse] = vr;
        out2[forward] = vi;
        out3[reverse] = vi;
    }
    return 0;
}

/* ---------------------------------------------------------------------- */
/* windows                                                                */
/* ---------------------------------------------------------------------- */

/* dsp.rs:542-547 */
void sko_sine_window(int len, float *out) {
    float scale = PI_F / (float)len;
    for (int i = 0; i < len; ++i) out[i] = sinf(((float)i + 0.5f) * scale);
}

/* dsp.rs:572-587 */
static double bessel_i0_f64(double x) {
    double half = x * 0.5, sum = 1.0, term = 1.0;
    for (int k = 1; k <= 64; ++k) {
        double ratio = half / (double)k;
        term *= ratio * ratio;
        sum += term;
        if (fabs(term) < 1.0e-14 * sum) break;
    }
    return sum;
}

/* dsp.rs:549-570 */
void sko_kbd_window(int len, float alpha, float *out) {
    int half = len / 2;
    double *kernel = (double *)malloc(sizeof(double) * (size_t)(half + 1));
    double denom_arg = PI_D * (double)alpha;
    for (int i = 0; i <= half; ++i) {
        double ratio = 2.0 * (double)i / (double)half - 1.0;
        double inner = 1.0 - ratio * ratio;
        if (inner < 0.0) inner = 0.0;
        kernel[i] = bessel_i0_f64(denom_arg * sqrt(inner));
    }
    double total = 0.0;
    for (int i = 0; i <= half; ++i) total += kernel[i];
    double cumulative = 0.0;
    for (int i = 0; i < len; ++i) out[i] = 0.0f;
    for (int i = 0; i < half; ++i) {
        cumulative += kernel[i];
        out[i] = (float)sqrt(cumulative / total);
        out[len - 1 - i] = out[i];
    }
    free(kernel);
}

static float g_long_sine[2048], g_long_kbd[2048], g_short_sine[256], g_short_kbd[256];
static int g_windows_ready = 0;
static void ensure_windows(void) { /* dsp.rs:31-41 AacDsp::new */
    if (g_windows_ready) return;
    sko_sine_window(2048, g_long_sine);
    sko_kbd_window(2048, 4.0f, g_long_kbd);
    sko_sine_window(256, g_short_sine);
    sko_kbd_window(256, 6.0f, g_short_kbd);
    g_windows_ready = 1;
}
static const float *long_window(int shape) { return shape == SKO_KBD ? g_long_kbd : g_long_sine; }
static const float *short_window(int shape) { return shape == SKO_KBD ? g_short_kbd : g_short_sine; }

/* ---------------------------------------------------------------------- */
/* channel synthesis                                                      */
/* ---------------------------------------------------------------------- */

void sko_channel_init(sko_channel *ch) { /* dsp.rs:155-171 */
    memset(ch->delay, 0, sizeof(ch->delay));
    ch->prev_shape = SKO_SINE;
}

/* dsp.rs:353-368 */
static float first_window(int seq, const float *prev_long, const float *prev_short, int i) {
    if (seq == SKO_ONLY_LONG || seq == SKO_LONG_START) return prev_long[i];
    if (i < 448) return 0.0f; /* LongStop */
    if (i < 576) return prev_short[i - 448];
    return 1.0f;
}
/* dsp.rs:370-387 */
static float second_window(int seq, const float *cur_long, const float *cur_short, int i) {
    if (seq == SKO_ONLY_LONG || seq == SKO_LONG_STOP) return cur_long[i + 1024];
    if (i < 448) return 1.0f; /* LongStart */
    if (i < 576) return cur_short[128 + i - 448];
    return 0.0f;
}

int sko_synthesize_channel(sko_channel *ch, const float *coeffs, int seq, int shape, float *out) {
    if (seq < 0 || seq > 3 || (shape != SKO_SINE && shape != SKO_KBD)) return -1;
    ensure_windows();
    const int n = 1024;
    int prev = ch->prev_shape; /* decoder.rs:337 */
    float imdct[2048];
    if (seq != SKO_EIGHT_SHORT) { /* dsp.rs:230-282 */
        const float *pl = long_window(prev), *cl = long_window(shape);
        const float *ps = short_window(prev), *cs = short_window(shape);
        sko_imdct_fast(coeffs, imdct, n);
        for (int i = 0; i < n; ++i) {
            float first = imdct[i] * first_window(seq, pl, ps, i);
            float second = imdct[i + n] * second_window(seq, cl, cs, i);
            out[i] = first + ch->delay[i];
            ch->delay[i] = second;
        }
    } else { /* dsp.rs:284-338 */
        const float *ps = short_window(prev), *cs = short_window(shape);
        float sh[256];
        memset(imdct, 0, sizeof(imdct));
        for (int w = 0; w < 8; ++w) {
            int out_start = 448 + w * 128;
            sko_imdct_fast(coeffs + w * 128, sh, 128);
            if (w == 0) {
                for (int s = 0; s < 128; ++s) imdct[out_start + s] += sh[s] * ps[s];
                for (int s = 128; s < 256; ++s) imdct[out_start + s] += sh[s] * cs[s];
            } else {
                for (int s = 0; s < 256; ++s) imdct[out_start + s] += sh[s] * cs[s];
            }
        }
        for (int i = 0; i < n; ++i) {
            out[i] = imdct[i] + ch->delay[i];
            ch->delay[i] = imdct[i + n];
        }
    }
    ch->prev_shape = shape; /* decoder.rs:371 */
    return 0;
}

/* ---------------------------------------------------------------------- */
/* dequant                                                                */
/* ---------------------------------------------------------------------- */

float sko_pow43(uint32_t v) { return powf((float)v, 4.0f / 3.0f); } /* dsp.rs:420-437 */
float sko_scalefactor_multiplier(int sf) { /* dsp.rs:407-413, 439-450 */
    return powf(2.0f, ((float)sf - 100.0f) * 0.25f);
}
float sko_dequantize_signed(int32_t q, int sf) { /* dsp.rs:389-405 */
    if (q == 0) return 0.0f;
    float sign = q < 0 ? -1.0f : 1.0f;
    uint32_t mag = q < 0 ? (uint32_t)(-(int64_t)q) : (uint32_t)q;
    return sign * sko_pow43(mag) * sko_scalefactor_multiplier(sf);
}

void sko_seeded_spectrum(int len, uint32_t seed, float *out) { /* dsp.rs:725-738 */
    uint32_t state = seed;
    for (int i = 0; i < len; ++i) {
        state = state * 1664525u + 1013904223u;
        if (i % 7 == 0) {
            out[i] = 0.0f;
        } else {
            float centered = (float)((state >> 8) & 0xffffu) / 32768.0f - 1.0f;
            out[i] = centered * 12.0f;
        }
    }
}

void sko_pcm_stats_from(const float *pcm, size_t n, sko_pcm_stats *st) { /* aac-wasm-bench lib.rs:73-100 */
    double sum_squares = 0.0, peak = 0.0;
    uint64_t checksum = 0xcbf29ce484222325ull;
    for (size_t i = 0; i < n; ++i) {
        double s = (double)pcm[i];
        sum_squares += s * s;
        double a = fabs(s);
        if (a > peak) peak = a; /* f64::max ignores NaN on the right: same for a NaN-free stream */
        uint32_t bits;
        memcpy(&bits, &pcm[i], 4);
        checksum ^= (uint64_t)bits;
        checksum *= 0x100000001b3ull;
    }
    st->sample_count = n;
    st->rms = n ? sqrt(sum_squares / (double)n) : 0.0;
    st->peak_abs = peak;
    st->checksum = checksum;
}

/* ---------------------------------------------------------------------- */
/* scalar sample conversions                                              */
/* ---------------------------------------------------------------------- */

int16_t sko_float_sample_to_i16(float s) { /* soundkit-decoder lib.rs:1815-1827 */
    float finite = isfinite(s) ? f32_clamp(s, -1.0f, 1.0f) : 0.0f;
    double scaled = finite < 0.0f ? (double)finite * 32768.0 : (double)finite * 32767.0;
    int32_t r = f64_as_i32(round(scaled));
    if (r < INT16_MIN) r = INT16_MIN;
    if (r > INT16_MAX) r = INT16_MAX;
    return (int16_t)r;
}

int16_t sko_mp3_f32_to_i16(float s) { /* soundkit-mp3 lib.rs:376-385 */
    float scaled = roundf(s * 32767.0f);
    if (scaled > 32767.0f) return INT16_MAX;
    if (scaled < -32768.0f) return INT16_MIN;
    return f32_as_i16(scaled);
}

int32_t sko_mp3_f32_to_i32(float s) { /* soundkit-mp3 lib.rs:387-396; i32::MAX as f32 == 2^31 */
    float scaled = roundf(s * 2147483648.0f);
    if (scaled > 2147483648.0f) return INT32_MAX;
    if (scaled < -2147483648.0f) return INT32_MIN;
    return f32_as_i32(scaled);
}

/* ---------------------------------------------------------------------- */
/* audio_bytes                                                            */
/* ---------------------------------------------------------------------- */

static inline int32_t s24le(const uint8_t *b) { /* audio_bytes.rs:40-45 / 312-315 */
    uint32_t u = (uint32_t)b[0] | ((uint32_t)b[1] << 8) | ((uint32_t)b[2] << 16);
    if (u & 0x800000u) u |= 0xFF000000u;
    return (int32_t)u;
}
static inline int32_t s24be(const uint8_t *b) {
    uint32_t u = (uint32_t)b[2] | ((uint32_t)b[1] << 8) | ((uint32_t)b[0] << 16);
    if (u & 0x800000u) u |= 0xFF000000u;
    return (int32_t)u;
}
static inline int32_t s32le(const uint8_t *b) {
    return (int32_t)((uint32_t)b[0] | ((uint32_t)b[1] << 8) | ((uint32_t)b[2] << 16) | ((uint32_t)b[3] << 24));
}
static inline int32_t s32be(const uint8_t *b) {
    return (int32_t)((uint32_t)b[3] | ((uint32_t)b[2] << 8) | ((uint32_t)b[1] << 16) | ((uint32_t)b[0] << 24));
}
static inline int16_t s16le(const uint8_t *b) { return (int16_t)((uint16_t)b[0] | ((uint16_t)b[1] << 8)); }
static inline int16_t s16be(const uint8_t *b) { return (int16_t)((uint16_t)b[1] | ((uint16_t)b[0] << 8)); }
static inline float f32_from_bits(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }
static inline float f32le(const uint8_t *b) { return f32_from_bits((uint32_t)s32le(b)); }
static inline float f32be(const uint8_t *b) { return f32_from_bits((uint32_t)s32be(b)); }

static inline int32_t f32_to_i32_pcm(float x) { /* audio_bytes.rs:194-199; i32::MAX as f32 == -(i32::MIN as f32) == 2^31 */
    float c = f32_clamp(x, -1.0f, 1.0f);
    return c >= 0.0f ? f32_as_i32(c * 2147483648.0f) : f32_as_i32(c * 2147483648.0f);
}
static inline int32_t f32_to_s24_pcm(float x) { /* audio_bytes.rs:210-216 */
    float c = f32_clamp(x, -1.0f, 1.0f);
    return c >= 0.0f ? f32_as_i32(c * 8388607.0f) : f32_as_i32(c * 8388608.0f);
}

static const int8_t k_in_bytes[SKO_OP_COUNT] = {2, 2, 2, 3, 3, 3, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 2, 2, 2, 4, 4, 4, 2, 4, 4, 4, 4};
static const int8_t k_out_bytes[SKO_OP_COUNT] = {4, 2, 2, 4, 2, 2, 4, 4, 4, 4, 4, 4, 2, 2, 2, 2, 4, 4, 2, 2, 4, 2, 2, 2, 4, 4, 2, 2, 4};

int sko_op_in_bytes(int op) { return (op < 0 || op >= SKO_OP_COUNT) ? -1 : k_in_bytes[op]; }
int sko_op_out_bytes(int op) { return (op < 0 || op >= SKO_OP_COUNT) ? -1 : k_out_bytes[op]; }

int sko_pcm_convert(int op, const void *in_, void *out_, size_t n) {
    if (op < 0 || op >= SKO_OP_COUNT) return -1;
    const uint8_t *in = (const uint8_t *)in_;
    float *of = (float *)out_;
    int16_t *o16 = (int16_t *)out_;
    int32_t *o32 = (int32_t *)out_;
    for (size_t i = 0; i < n; ++i) {
        const uint8_t *p = in + i * (size_t)k_in_bytes[op];
        switch (op) {
        case SKO_OP_I16LE_TO_F32: of[i] = (float)s16le(p) / 32768.0f; break;
        case SKO_OP_I16_TO_I16LE: /* native i16 -> LE bytes: identity on a little-endian host */
        case SKO_OP_I16LE_TO_I16:
        case SKO_OP_S16LE_TO_I16: o16[i] = s16le(p); break;
        case SKO_OP_S24LE_TO_I32: o32[i] = s24le(p); break;
        case SKO_OP_S24LE_TO_I16: o16[i] = (int16_t)(s24le(p) >> 8); break;
        case SKO_OP_S24BE_TO_I16: o16[i] = (int16_t)(s24be(p) >> 8); break;
        case SKO_OP_S32LE_TO_I32: o32[i] = s32le(p); break;
        case SKO_OP_S32BE_TO_I32: o32[i] = s32be(p); break;
        case SKO_OP_S32LE_TO_S24: o32[i] = s32le(p) & 0x00FFFFFF; break;
        case SKO_OP_S32BE_TO_S24: o32[i] = s32be(p) & 0x00FFFFFF; break;
        /* 2.0f32.powi(31) - 1.0 rounds to 2^31 in f32 (audio_bytes.rs:128) */
        case SKO_OP_S32LE_TO_F32: of[i] = (float)s32le(p) / 2147483648.0f; break;
        case SKO_OP_S32BE_TO_F32: of[i] = (float)s32be(p) / 2147483648.0f; break;
        case SKO_OP_S32LE_TO_I16: o16[i] = (int16_t)(s32le(p) >> 16); break;
        case SKO_OP_S32BE_TO_I16: o16[i] = (int16_t)(s32be(p) >> 16); break;
        case SKO_OP_F32LE_TO_I16:
        case SKO_OP_VEC_F32_TO_I16: o16[i] = f32_as_i16(f32_clamp(f32le(p), -1.0f, 1.0f) * 32767.0f); break;
        case SKO_OP_F32BE_TO_I16: o16[i] = f32_as_i16(f32_clamp(f32be(p), -1.0f, 1.0f) * 32767.0f); break;
        case SKO_OP_F32LE_TO_I32: o32[i] = f32_to_i32_pcm(f32le(p)); break;
        case SKO_OP_F32LE_TO_S24: o32[i] = f32_to_s24_pcm(f32le(p)); break;
        case SKO_OP_S16BE_TO_I16: o16[i] = s16be(p); break;
        case SKO_OP_S16LE_TO_I32: o32[i] = (int32_t)s16le(p); break;
        case SKO_OP_STEREO_TO_MONO_TAKE_LEFT: o16[i] = s16le(p); break;
        case SKO_OP_STEREO_TO_MONO_AVG: o16[i] = (int16_t)(((int32_t)s16le(p) + (int32_t)s16le(p + 2)) / 2); break;
        case SKO_OP_VEC_I16_TO_F32: of[i] = (float)s16le(p) / 32768.0f; break;
        /* const MAX_I32: f32 = 2147483647.0 rounds to 2^31 (audio_pipeline.rs:42) */
        case SKO_OP_VEC_I32_TO_F32: of[i] = (float)s32le(p) / 2147483648.0f; break;
        case SKO_OP_FLOAT_TO_I16_ROUND: o16[i] = sko_float_sample_to_i16(f32le(p)); break;
        case SKO_OP_MP3_F32_TO_I16: o16[i] = sko_mp3_f32_to_i16(f32le(p)); break;
        case SKO_OP_MP3_F32_TO_I32: o32[i] = sko_mp3_f32_to_i32(f32le(p)); break;
        default: return -1;
        }
    }
    return 0;
}

void sko_interleave_i16(const int16_t *planar, size_t frames, int ch, uint8_t *out) { /* :250 */
    for (size_t i = 0; i < frames; ++i)
        for (int c = 0; c < ch; ++c) {
            uint16_t v = (uint16_t)planar[(size_t)c * frames + i];
            *out++ = (uint8_t)(v & 0xff);
            *out++ = (uint8_t)(v >> 8);
        }
}
void sko_deinterleave_i16(const uint8_t *in, size_t frames, int ch, int16_t *planar) { /* :264 */
    for (size_t i = 0; i < frames; ++i)
        for (int c = 0; c < ch; ++c) planar[(size_t)c * frames + i] = s16le(in + (i * (size_t)ch + (size_t)c) * 2);
}
void sko_deinterleave_s24(const uint8_t *in, size_t frames, int ch, int32_t *planar) { /* :280 */
    for (size_t i = 0; i < frames; ++i)
        for (int c = 0; c < ch; ++c) planar[(size_t)c * frames + i] = s24le(in + (i * (size_t)ch + (size_t)c) * 3);
}
void sko_deinterleave_f32(const uint8_t *in, size_t frames, int ch, float *planar) { /* :296 */
    for (size_t i = 0; i < frames; ++i)
        for (int c = 0; c < ch; ++c) planar[(size_t)c * frames + i] = f32le(in + (i * (size_t)ch + (size_t)c) * 4);
}
void sko_interleave_f32(const float *planar, size_t frames, int ch, uint8_t *out) { /* decoder lib.rs:3685 */
    for (size_t i = 0; i < frames; ++i)
        for (int c = 0; c < ch; ++c) {
            memcpy(out, &planar[(size_t)c * frames + i], 4);
            out += 4;
        }
}

int sko_fmt_bytes(int fmt) {
    switch (fmt) {
    case SKO_FMT_S16LE: case SKO_FMT_S16BE: return 2;
    case SKO_FMT_S24LE: case SKO_FMT_S24BE: return 3;
    case SKO_FMT_S32LE: case SKO_FMT_S32BE: case SKO_FMT_F32LE: case SKO_FMT_F32BE: return 4;
    default: return -1;
    }
}

int sko_decoder_bytes_to_f32_planar(int fmt, const uint8_t *in, size_t frames, int ch, float *planar) {
    int bps = sko_fmt_bytes(fmt);
    if (bps < 0 || ch <= 0) return -1;
    for (size_t i = 0; i < frames; ++i)
        for (int c = 0; c < ch; ++c) {
            const uint8_t *p = in + (i * (size_t)ch + (size_t)c) * (size_t)bps;
            float s;
            switch (fmt) {
            case SKO_FMT_F32LE: s = f32le(p); break;
            case SKO_FMT_F32BE: s = f32be(p); break;
            case SKO_FMT_S16LE: s = (float)s16le(p) / 32768.0f; break;
            case SKO_FMT_S16BE: s = (float)s16be(p) / 32768.0f; break;
            case SKO_FMT_S24LE: s = (float)s24le(p) / 8388608.0f; break;
            case SKO_FMT_S24BE: s = (float)s24be(p) / 8388608.0f; break;
            case SKO_FMT_S32LE: s = (float)s32le(p) / 2147483648.0f; break;
            default: s = (float)s32be(p) / 2147483648.0f; break;
            }
            planar[(size_t)c * frames + i] = isfinite(s) ? s : 0.0f; /* lib.rs:3614 */
        }
    return 0;
}

int sko_core_bytes_to_f32_planar(int fmt, const uint8_t *in, size_t frames, int ch, float *planar) {
    if (ch <= 0) return -1;
    for (size_t i = 0; i < frames; ++i)
        for (int c = 0; c < ch; ++c) {
            size_t k = i * (size_t)ch + (size_t)c;
            float s;
            switch (fmt) {
            case SKO_FMT_S16LE: s = (float)s16le(in + k * 2) / 32768.0f; break;       /* :94, :33 */
            case SKO_FMT_S24LE: s = (float)s24le(in + k * 3) / 2147483648.0f; break;   /* :62-65, :95, :45 */
            case SKO_FMT_S32LE: s = (float)s32le(in + k * 4) / 2147483648.0f; break;   /* :80-87 */
            case SKO_FMT_F32LE: s = f32le(in + k * 4); break;                          /* :66-69, :96 */
            default: return -1;
            }
            planar[(size_t)c * frames + i] = s;
        }
    return 0;
}

int sko_f32_planar_to_bytes(int fmt, const float *planar, size_t frames, int ch, uint8_t *out) {
    for (size_t i = 0; i < frames; ++i)
        for (int c = 0; c < ch; ++c) {
            float x = planar[(size_t)c * frames + i];
            switch (fmt) {
            case SKO_FMT_F32LE: memcpy(out, &x, 4); out += 4; break;
            case SKO_FMT_S16LE: {
                uint16_t v = (uint16_t)sko_float_sample_to_i16(x);
                *out++ = (uint8_t)(v & 0xff); *out++ = (uint8_t)(v >> 8);
                break;
            }
            case SKO_FMT_S24LE: { /* lib.rs:3649-3661 */
                float c1 = f32_clamp(x, -1.0f, 1.0f);
                uint32_t v = (uint32_t)(c1 >= 0.0f ? f32_as_i32(c1 * 8388607.0f) : f32_as_i32(c1 * 8388608.0f));
                *out++ = (uint8_t)(v & 0xff); *out++ = (uint8_t)((v >> 8) & 0xff); *out++ = (uint8_t)((v >> 16) & 0xff);
                break;
            }
            case SKO_FMT_S32LE: { /* lib.rs:3664-3677 */
                uint32_t v = (uint32_t)f32_to_i32_pcm(x);
                *out++ = (uint8_t)(v & 0xff); *out++ = (uint8_t)((v >> 8) & 0xff);
                *out++ = (uint8_t)((v >> 16) & 0xff); *out++ = (uint8_t)(v >> 24);
                break;
            }
            default: return -1;
            }
        }
    return 0;
}

void sko_downmix_mono(const float *planar, size_t frames, int ch, float *mono) { /* lib.rs:3500-3509 */
    float scale = 1.0f / (float)ch;
    for (size_t i = 0; i < frames; ++i) mono[i] = 0.0f;
    for (int c = 0; c < ch; ++c)
        for (size_t i = 0; i < frames; ++i) mono[i] += planar[(size_t)c * frames + i] * scale;
}

int sko_exact_signed_pcm_to_i16(int fmt, const uint8_t *in, size_t samples, uint8_t *out) { /* lib.rs:3458-3489 */
    for (size_t i = 0; i < samples; ++i) {
        int32_t s;
        int shift;
        switch (fmt) {
        case SKO_FMT_S24LE: s = s24le(in + i * 3); shift = 8; break;
        case SKO_FMT_S24BE: s = s24be(in + i * 3); shift = 8; break;
        case SKO_FMT_S32LE: s = s32le(in + i * 4); shift = 16; break;
        case SKO_FMT_S32BE: s = s32be(in + i * 4); shift = 16; break;
        default: return -1;
        }
        uint16_t v = (uint16_t)(int16_t)(s >> shift);
        out[i * 2] = (uint8_t)(v & 0xff);
        out[i * 2 + 1] = (uint8_t)(v >> 8);
    }
    return 0;
}

void sko_planar_f32_to_s16_interleaved(const float *planar, size_t frames, int ch, int16_t *out) { /* lib.rs:1802-1806 */
    for (size_t i = 0; i < frames; ++i)
        for (int c = 0; c < ch; ++c) out[i * (size_t)ch + (size_t)c] = sko_float_sample_to_i16(planar[(size_t)c * frames + i]);
}

/* ---------------------------------------------------------------------- */
/* rubato 0.14.1 SincFixedIn<f32>, Linear interpolation, restated         */
/* ---------------------------------------------------------------------- */

#define SINC_LEN 256
#define OVERSAMPLING 256

struct sko_resampler {
    double ratio;         /* resample_ratio == target_ratio (never changed by the reference) */
    size_t chunk_size;
    int channels;
    double last_index;
    float *sincs;         /* [OVERSAMPLING][SINC_LEN] */
    float *buffer;        /* [channels][chunk_size + 2*SINC_LEN] */
    size_t buf_stride;
};

/* rubato windows.rs blackman_harris (T = f32), squared for BlackmanHarris2 */
static void make_window_bh2(size_t npoints, float *w) {
    float pi2 = 2.0f * PI_F, pi4 = 4.0f * PI_F, pi6 = 6.0f * PI_F;
    float np_f = (float)npoints;
    float a = 0.35875f, b = 0.48829f, c = 0.14128f, d = 0.01168f;
    for (size_t x = 0; x < npoints; ++x) {
        float xf = (float)x;
        float v = a - b * cosf(pi2 * xf / np_f) + c * cosf(pi4 * xf / np_f) - d * cosf(pi6 * xf / np_f);
        w[x] = v * v;
    }
}
static float sinc_f32(float v) { return v == 0.0f ? 1.0f : sinf(v * PI_F) / (v * PI_F); } /* rubato sinc.rs */

/* rubato sinc.rs make_sincs (T = f32) */
static void make_sincs(float f_cutoff, float *sincs) {
    size_t npoints = SINC_LEN, factor = OVERSAMPLING, tot = npoints * factor;
    float *y = (float *)malloc(sizeof(float) * tot);
    float *w = (float *)malloc(sizeof(float) * tot);
    make_window_bh2(tot, w);
    float sum = 0.0f;
    for (size_t x = 0; x < tot; ++x) {
        float val = w[x] * sinc_f32(((float)x - (float)(tot / 2)) * f_cutoff / (float)factor);
        sum += val;
        y[x] = val;
    }
    sum /= (float)factor;
    for (size_t p = 0; p < npoints; ++p)
        for (size_t n = 0; n < factor; ++n) sincs[(factor - n - 1) * npoints + p] = y[factor * p + n] / sum;
    free(y);
    free(w);
}

sko_resampler *sko_resampler_new(double ratio, size_t chunk_size, int channels) {
    if (!(ratio > 0.0) || channels <= 0) return NULL;
    sko_resampler *r = (sko_resampler *)calloc(1, sizeof(*r));
    r->ratio = ratio;
    r->chunk_size = chunk_size;
    r->channels = channels;
    r->last_index = -(double)(SINC_LEN / 2);
    r->sincs = (float *)malloc(sizeof(float) * SINC_LEN * OVERSAMPLING);
    float f_cutoff = ratio >= 1.0 ? 0.95f : 0.95f * (float)ratio;
    make_sincs(f_cutoff, r->sincs);
    r->buf_stride = chunk_size + 2 * SINC_LEN;
    r->buffer = (float *)calloc((size_t)channels * r->buf_stride, sizeof(float));
    return r;
}

void sko_resampler_free(sko_resampler *r) {
    if (!r) return;
    free(r->sincs);
    free(r->buffer);
    free(r);
}

size_t sko_resampler_output_frames_max(const sko_resampler *r) {
    return (size_t)((double)r->chunk_size * r->ratio * 2.0 + 10.0);
}

void sko_resampler_taps_phase0(const sko_resampler *r, float *taps) { memcpy(taps, r->sincs, sizeof(float) * SINC_LEN); }

/* rubato interpolator scalar get_sinc_interpolated: 8 running sums over the 256 taps */
static float sinc_dot(const float *wave, const float *sinc) {
    float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int i = 0; i < SINC_LEN; i += 8)
        for (int j = 0; j < 8; ++j) acc[j] += wave[i + j] * sinc[i + j];
    return acc[0] + acc[1] + acc[2] + acc[3] + acc[4] + acc[5] + acc[6] + acc[7];
}

size_t sko_resampler_process(sko_resampler *r, const float *in, size_t in_stride, float *out, size_t out_stride) {
    const size_t chunk = r->chunk_size;
    double t_ratio = 1.0 / r->ratio;
    long end_idx = (long)chunk - ((long)SINC_LEN + 1) - (long)ceil(t_ratio);
    for (int c = 0; c < r->channels; ++c) {
        float *buf = r->buffer + (size_t)c * r->buf_stride;
        memmove(buf, buf + chunk, sizeof(float) * 2 * SINC_LEN);
        memcpy(buf + 2 * SINC_LEN, in + (size_t)c * in_stride, sizeof(float) * chunk);
    }
    double idx = r->last_index;
    size_t n = 0;
    while (idx < (double)end_idx) {
        idx += t_ratio; /* t_ratio_increment is 0: resample_ratio == target_ratio */
        double fl = floor(idx);
        long index0 = (long)fl;
        long sub0 = (long)floor((idx - fl) * (double)OVERSAMPLING);
        long index1 = index0, sub1 = sub0 + 1;
        if (sub1 >= OVERSAMPLING) { sub1 -= OVERSAMPLING; index1 += 1; }
        double scaled = idx * (double)OVERSAMPLING;
        float frac = (float)(scaled - floor(scaled));
        for (int c = 0; c < r->channels; ++c) {
            const float *buf = r->buffer + (size_t)c * r->buf_stride;
            float p0 = sinc_dot(buf + (index0 + 2 * SINC_LEN), r->sincs + (size_t)sub0 * SINC_LEN);
            float p1 = sinc_dot(buf + (index1 + 2 * SINC_LEN), r->sincs + (size_t)sub1 * SINC_LEN);
            out[(size_t)c * out_stride + n] = p0 + frac * (p1 - p0); /* interp_lin */
        }
        ++n;
    }
    r->last_index = idx - (double)chunk;
    return n;
}

size_t sko_resampler_process_partial(sko_resampler *r, const float *in, size_t in_stride, size_t n_in, float *out,
                                     size_t out_stride) {
    size_t chunk = r->chunk_size;
    if (n_in > chunk) return 0;
    float *pad = (float *)calloc((size_t)r->channels * chunk + 1, sizeof(float));
    if (in)
        for (int c = 0; c < r->channels; ++c) memcpy(pad + (size_t)c * chunk, in + (size_t)c * in_stride, sizeof(float) * n_in);
    size_t n = sko_resampler_process(r, pad, chunk, out, out_stride);
    free(pad);
    return n;
}

size_t sko_downsample_out_max(size_t frames, uint32_t in_hz, uint32_t out_hz) {
    return (size_t)((double)frames * ((double)out_hz / (double)in_hz) * 2.0 + 10.0);
}

size_t sko_downsample_planar(const float *in, size_t frames, int ch, uint32_t in_hz, uint32_t out_hz, float *out,
                             size_t out_stride) {
    sko_resampler *r = sko_resampler_new((double)out_hz / (double)in_hz, frames, ch); /* audio_pipeline.rs:482-489 */
    if (!r) return 0;
    size_t n = sko_resampler_process(r, in, frames, out, out_stride); /* :491 */
    sko_resampler_free(r);
    return n;
}

/* soundkit-decoder lib.rs:1917-2060 */
struct sko_streaming_resampler {
    sko_resampler *rs;
    size_t chunk_size;
    int channels;
    uint32_t in_hz, out_hz;
    float *accum; /* [channels][cap] */
    size_t cap, len, start;
};

sko_streaming_resampler *sko_streaming_new(uint32_t in_hz, uint32_t out_hz, int channels) {
    sko_streaming_resampler *s = (sko_streaming_resampler *)calloc(1, sizeof(*s));
    s->chunk_size = 4096; /* RESAMPLE_CHUNK_SIZE lib.rs:79 */
    s->channels = channels;
    s->in_hz = in_hz;
    s->out_hz = out_hz;
    s->rs = sko_resampler_new((double)out_hz / (double)in_hz, s->chunk_size, channels);
    if (!s->rs) { free(s); return NULL; }
    return s;
}

void sko_streaming_free(sko_streaming_resampler *s) {
    if (!s) return;
    sko_resampler_free(s->rs);
    free(s->accum);
    free(s);
}

size_t sko_streaming_process(sko_streaming_resampler *s, const float *in, size_t in_stride, size_t n, float *out,
                             size_t out_stride, size_t out_off) {
    if (s->len + n > s->cap) {
        size_t ncap = (s->len + n) * 2 + 4096;
        float *na = (float *)calloc((size_t)s->channels * ncap, sizeof(float));
        for (int c = 0; c < s->channels; ++c)
            if (s->len) memcpy(na + (size_t)c * ncap, s->accum + (size_t)c * s->cap, sizeof(float) * s->len);
        free(s->accum);
        s->accum = na;
        s->cap = ncap;
    }
    for (int c = 0; c < s->channels; ++c) memcpy(s->accum + (size_t)c * s->cap + s->len, in + (size_t)c * in_stride, sizeof(float) * n);
    s->len += n;
    size_t produced = 0;
    while (s->len - s->start >= s->chunk_size) { /* lib.rs:1987-2003 */
        produced += sko_resampler_process(s->rs, s->accum + s->start, s->cap, out + out_off + produced, out_stride);
        s->start += s->chunk_size;
    }
    if (s->start >= s->chunk_size * 8 && s->start * 2 >= s->len) { /* lib.rs:2005-2012 */
        for (int c = 0; c < s->channels; ++c)
            memmove(s->accum + (size_t)c * s->cap, s->accum + (size_t)c * s->cap + s->start, sizeof(float) * (s->len - s->start));
        s->len -= s->start;
        s->start = 0;
    }
    return produced;
}

size_t sko_streaming_flush(sko_streaming_resampler *s, float *out, size_t out_stride, size_t out_off) {
    size_t remaining = s->len - s->start;
    size_t n;
    if (remaining > 0) { /* lib.rs:2020-2047 */
        size_t padded = s->chunk_size - remaining;
        n = sko_resampler_process_partial(s->rs, s->accum + s->start, s->cap, remaining, out + out_off, out_stride);
        if (padded > 0) {
            size_t trim = (size_t)round(((double)padded * (double)s->out_hz) / (double)s->in_hz);
            n = n > trim ? n - trim : 0;
        }
        s->len = 0;
        s->start = 0;
    } else { /* lib.rs:2048-2056 */
        n = sko_resampler_process_partial(s->rs, NULL, 0, 0, out + out_off, out_stride);
    }
    return n;
}
