/*
 * sk_oracle.h -- CPU restatement of soundkit's per-frame decode DSP hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the product:
 * only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * link or call it, and there only as the checker / reported CPU baseline.
 * The product path (soundkit_amd/csrc, libsoundkit_amd.so) never includes,
 * links or falls back to this code.
 *
 * Every function cites the reference file:line it restates (paths relative
 * to the upstream soundkit tree).  Arithmetic follows the reference's types:
 * f32 where the reference computes in f32, f64 where it computes in f64,
 * with Rust cast semantics (`as i16` / `as i32` from float saturate, NaN -> 0;
 * f32::round is half-away-from-zero; f32::clamp keeps NaN).
 *
 * Parity pins (see tests/test_oracle_pins.py): the reference's own in-file
 * known answers -- IMDCT fast-vs-direct on the 9-value pattern and the LCG
 * seeded spectra (dsp.rs:654-738), window Princen-Bradley (dsp.rs:594-614),
 * window sequencing (dsp.rs:772-795), dequant (dsp.rs:810-822), audio_bytes
 * vectors (audio_bytes.rs:380-468), f32<->PCM vectors (audio_pipeline.rs:698-763).
 * The sinc resampler restates the third-party crate rubato 0.14.1 (absent from
 * the reference tree): its sample values are "parity unpinned"; only the
 * reference's length/rate assertions pin it (soundkit-decoder lib.rs:5188-5238).
 */
#ifndef SK_ORACLE_H
#define SK_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- AAC-LC synthesis: soundkit-aac-lc/src/dsp.rs ---------------------- */

enum { SKO_ONLY_LONG = 0, SKO_LONG_START = 1, SKO_EIGHT_SHORT = 2, SKO_LONG_STOP = 3 }; /* ics.rs:7-12 */
enum { SKO_SINE = 0, SKO_KBD = 1 };                                                    /* ics.rs:32-35 */

/* dsp.rs:453-474 (test-only direct form, f32 arithmetic as written there) */
void sko_imdct_direct_f32(const float *in, float *out, int n);
/* the mathematical definition the direct form evaluates, in f64 */
void sko_imdct_direct_f64(const float *in, double *out, int n);
/* dsp.rs:94-119 + 476-535: pre-twiddle, N/2-point forward complex FFT, post-twiddle scatter */
int sko_imdct_fast(const float *in, float *out, int n);

/* dsp.rs:542-547 / 549-570 / 572-587 */
void sko_sine_window(int len, float *out);
void sko_kbd_window(int len, float alpha, float *out);

/* dsp.rs:143-171 carried per-channel state; decoder.rs:336-374 update order */
typedef struct {
    float delay[1024];
    int32_t prev_shape;
} sko_channel;

void sko_channel_init(sko_channel *ch);
/* dsp.rs:230-338 via decoder.rs:336-374.  coeffs[1024] -> out[1024]; updates delay and
 * prev_shape.  Returns 0, or -1 for an invalid sequence/shape. */
int sko_synthesize_channel(sko_channel *ch, const float *coeffs, int window_sequence,
                           int window_shape, float *out);

/* dsp.rs:389-450 */
float sko_pow43(uint32_t v);
float sko_scalefactor_multiplier(int sf);
float sko_dequantize_signed(int32_t q, int sf);

/* dsp.rs:725-738 (test LCG spectrum) */
void sko_seeded_spectrum(int len, uint32_t seed, float *out);

/* aac-wasm-bench/src/lib.rs:73-100 PcmStats */
typedef struct {
    uint64_t sample_count;
    double rms;
    double peak_abs;
    uint64_t checksum;
} sko_pcm_stats;
void sko_pcm_stats_from(const float *pcm, size_t n, sko_pcm_stats *out);

/* ---- scalar sample conversions ---------------------------------------- */

int16_t sko_float_sample_to_i16(float s);  /* soundkit-decoder/src/lib.rs:1815-1827 */
int16_t sko_mp3_f32_to_i16(float s);       /* soundkit-mp3/src/lib.rs:376-385 */
int32_t sko_mp3_f32_to_i32(float s);       /* soundkit-mp3/src/lib.rs:387-396 */

/* ---- soundkit::audio_bytes (soundkit/src/audio_bytes.rs) --------------- */
/* Elementwise ops share one entry point; `op` values are the SKO_OP_* below.
 * n = number of OUTPUT samples (for the stereo->mono ops: output frames). */
enum {
    SKO_OP_I16LE_TO_F32 = 0,      /* :3   */
    SKO_OP_I16_TO_I16LE = 1,      /* :17  */
    SKO_OP_I16LE_TO_I16 = 2,      /* :25  */
    SKO_OP_S24LE_TO_I32 = 3,      /* :36  */
    SKO_OP_S24LE_TO_I16 = 4,      /* :51  */
    SKO_OP_S24BE_TO_I16 = 5,      /* :66  */
    SKO_OP_S32LE_TO_I32 = 6,      /* :81  */
    SKO_OP_S32BE_TO_I32 = 7,      /* :91  */
    SKO_OP_S32LE_TO_S24 = 8,      /* :101 */
    SKO_OP_S32BE_TO_S24 = 9,      /* :112 */
    SKO_OP_S32LE_TO_F32 = 10,     /* :123 */
    SKO_OP_S32BE_TO_F32 = 11,     /* :134 */
    SKO_OP_S32LE_TO_I16 = 12,     /* :145 */
    SKO_OP_S32BE_TO_I16 = 13,     /* :156 */
    SKO_OP_F32LE_TO_I16 = 14,     /* :167 */
    SKO_OP_F32BE_TO_I16 = 15,     /* :178 */
    SKO_OP_F32LE_TO_I32 = 16,     /* :189 */
    SKO_OP_F32LE_TO_S24 = 17,     /* :205 */
    SKO_OP_S16BE_TO_I16 = 18,     /* :222 */
    SKO_OP_S16LE_TO_I16 = 19,     /* :231 */
    SKO_OP_S16LE_TO_I32 = 20,     /* :240 */
    SKO_OP_STEREO_TO_MONO_TAKE_LEFT = 21, /* :317 */
    SKO_OP_STEREO_TO_MONO_AVG = 22,       /* :344 */
    SKO_OP_VEC_F32_TO_I16 = 23,   /* audio_pipeline.rs:17 */
    SKO_OP_VEC_I16_TO_F32 = 24,   /* audio_pipeline.rs:29 */
    SKO_OP_VEC_I32_TO_F32 = 25,   /* audio_pipeline.rs:40 */
    SKO_OP_FLOAT_TO_I16_ROUND = 26, /* soundkit-decoder lib.rs:1815 */
    SKO_OP_MP3_F32_TO_I16 = 27,   /* soundkit-mp3 lib.rs:376 */
    SKO_OP_MP3_F32_TO_I32 = 28,   /* soundkit-mp3 lib.rs:387 */
    SKO_OP_COUNT = 29
};
/* bytes per input / output element of an op (input of the mono ops = one stereo frame) */
int sko_op_in_bytes(int op);
int sko_op_out_bytes(int op);
int sko_pcm_convert(int op, const void *in, void *out, size_t n);

/* audio_bytes.rs:250 / 264 / 280 / 296 and soundkit-decoder lib.rs:3685 */
void sko_interleave_i16(const int16_t *planar, size_t frames, int ch, uint8_t *out);
void sko_deinterleave_i16(const uint8_t *in, size_t frames, int ch, int16_t *planar);
void sko_deinterleave_s24(const uint8_t *in, size_t frames, int ch, int32_t *planar);
void sko_deinterleave_f32(const uint8_t *in, size_t frames, int ch, float *planar);
void sko_interleave_f32(const float *planar, size_t frames, int ch, uint8_t *out);

/* sample formats of interleaved PCM byte buffers */
enum {
    SKO_FMT_S16LE = 0, SKO_FMT_S16BE = 1, SKO_FMT_S24LE = 2, SKO_FMT_S24BE = 3,
    SKO_FMT_S32LE = 4, SKO_FMT_S32BE = 5, SKO_FMT_F32LE = 6, SKO_FMT_F32BE = 7
};
int sko_fmt_bytes(int fmt);

/* soundkit-decoder/src/lib.rs:3563-3617 audio_data_to_f32_channels (non-finite -> 0) */
int sko_decoder_bytes_to_f32_planar(int fmt, const uint8_t *in, size_t frames, int ch, float *planar);
/* soundkit/src/audio_pipeline.rs:74-98 audio_to_f32_channels (LE only; s24 and s32 both / i32::MAX as f32) */
int sko_core_bytes_to_f32_planar(int fmt, const uint8_t *in, size_t frames, int ch, float *planar);
/* soundkit-decoder/src/lib.rs:3619-3683 f32_channels_to_bytes; fmt in {S16LE,S24LE,S32LE,F32LE} */
int sko_f32_planar_to_bytes(int fmt, const float *planar, size_t frames, int ch, uint8_t *out);
/* soundkit-decoder/src/lib.rs:3492-3509 mono downmix */
void sko_downmix_mono(const float *planar, size_t frames, int ch, float *mono);
/* soundkit-decoder/src/lib.rs:3458-3489; fmt in {S24LE,S24BE,S32LE,S32BE} -> s16le */
int sko_exact_signed_pcm_to_i16(int fmt, const uint8_t *in, size_t samples, uint8_t *out);
/* soundkit-decoder/src/lib.rs:1793-1813: planar f32 frame -> interleaved s16 */
void sko_planar_f32_to_s16_interleaved(const float *planar, size_t frames, int ch, int16_t *out);

/* ---- sinc resampler: rubato 0.14.1 SincFixedIn<f32> restated ------------ */
/* call sites: soundkit/src/audio_pipeline.rs:474-491, soundkit-decoder/src/lib.rs:1939-2058.
 * Parameters fixed by the reference: sinc_len 256, f_cutoff 0.95, Linear interpolation,
 * oversampling 256, BlackmanHarris2, max_resample_ratio_relative 2.0. */
typedef struct sko_resampler sko_resampler;
sko_resampler *sko_resampler_new(double ratio, size_t chunk_size, int channels);
void sko_resampler_free(sko_resampler *r);
size_t sko_resampler_output_frames_max(const sko_resampler *r);
/* one full chunk: in[ch][chunk_size] planar (channel stride = in_stride floats) -> out planar
 * (channel stride out_stride); returns frames written per channel */
size_t sko_resampler_process(sko_resampler *r, const float *in, size_t in_stride, float *out,
                             size_t out_stride);
/* process_partial: n_in < chunk_size frames, zero padded (in may be NULL with n_in = 0) */
size_t sko_resampler_process_partial(sko_resampler *r, const float *in, size_t in_stride,
                                     size_t n_in, float *out, size_t out_stride);
/* the 256 taps of sub-filter 0 (what a ratio with zero fractional phase uses) */
void sko_resampler_taps_phase0(const sko_resampler *r, float *taps256);

/* soundkit/src/audio_pipeline.rs:438-493 downsample_audio on planar f32 input:
 * returns output frames per channel; out must hold sko_downsample_out_max(frames, in_hz, out_hz). */
size_t sko_downsample_out_max(size_t frames, uint32_t in_hz, uint32_t out_hz);
size_t sko_downsample_planar(const float *in, size_t frames, int ch, uint32_t in_hz, uint32_t out_hz,
                             float *out, size_t out_stride);

/* soundkit-decoder/src/lib.rs:1917-2060 StreamingResampler (fixed 4096-frame chunks) */
typedef struct sko_streaming_resampler sko_streaming_resampler;
sko_streaming_resampler *sko_streaming_new(uint32_t in_hz, uint32_t out_hz, int channels);
void sko_streaming_free(sko_streaming_resampler *s);
/* push n frames (planar, channel stride in_stride); appends produced frames to out (planar,
 * channel stride out_stride, starting at out_off); returns frames appended */
size_t sko_streaming_process(sko_streaming_resampler *s, const float *in, size_t in_stride, size_t n,
                             float *out, size_t out_stride, size_t out_off);
size_t sko_streaming_flush(sko_streaming_resampler *s, float *out, size_t out_stride, size_t out_off);

#ifdef __cplusplus
}
#endif
#endif
