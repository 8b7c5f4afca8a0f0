/*
 * soundkit_amd.h -- C ABI of the MI355X (gfx950) batched decode-DSP engine.
 *
 * This is the drop-in boundary for soundkit's per-frame decode DSP hot path
 * (SURVEY.md section 8).  The reference has no C FFI on this path: the three Rust
 * surfaces it would bind behind are cited per entry point below
 * (paths relative to the upstream soundkit tree).  INTEGRATION.md shows the Rust
 * `extern "C"` block and the `impl Decoder` / access-unit shims a maintainer adds.
 *
 * Conventions (mirrored from the reference):
 *  - the caller owns every buffer it passes; the engine owns per-stream carried
 *    state (overlap delay + previous window shape: dsp.rs:143-152; resampler
 *    history: soundkit-decoder lib.rs:1917-1927) in device memory;
 *  - no allocation per call once buffers have grown to the working size
 *    (soundkit-aac-lc/tests/no_alloc_decode.rs);
 *  - a bad frame fails that frame only (per-frame status word), never the batch
 *    (soundkit-decoder lib.rs:3131-3134 ends only that stream's worker);
 *  - frames of one stream inside one call are applied in array order; calls for
 *    one stream must not race (one worker per stream, lib.rs:2764); calls for
 *    different streams may come from different threads (the engine serialises
 *    submission internally).
 *  - every function returns SK_OK (0) or a negative sk_status; sk_strerror() names it.
 *
 * Pointers named d_* are DEVICE addresses on the engine's GPU; the *_dev entry
 * points enqueue on the engine's HIP stream (sk_engine_hip_stream) and return
 * without synchronising.  All other pointers are host memory and those entry
 * points return with results complete.
 *
 * One engine per GPU is the design.  A process MAY create several engines on one device (the scheduler's lanes are that), and
 * then they take turns with the device.  Why: on this platform a packed-f32 instruction delivers a wrong low half now and then while
 * another wave of its CU executes a 16x16x32 matrix instruction -- the FIR's and the resampler's (profiles/r04_lanes_corruption.md).
 * The default build contains no packed-f32 instructions, is immune (sk_kernels_use_packed_f32() == 0) and takes no turns: its
 * engines' work overlaps freely on the device.  The turns are the protection of a PACKED_F32=1 build inside a process: there, while
 * a second engine exists on the device, a tick holds the device from its first upload to its last wait, and every other compute entry point (sk_aac_plan_run_*, sk_downsample_*, sk_resampler_*,
 * sk_mp3_* synthesis) holds it for the call and WAITS for its own work before returning -- the *_dev entry points are synchronous
 * then.  Across processes nothing protects a PACKED_F32=1 build: it needs its GPU (or disjoint CUs, HSA_CU_MASK) to itself.
 */
#ifndef SOUNDKIT_AMD_H
#define SOUNDKIT_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SK_AAC_FRAME_LEN 1024u /* dsp.rs:9 LONG_SPECTRUM_LEN */
#define SK_MAX_CHANNELS 2u     /* AAC-LC SCE / CPE only: decoder.rs:116-133 */

typedef enum sk_status {
    SK_OK = 0,
    SK_ERR_INVALID_ARG = -1,
    SK_ERR_NO_DEVICE = -2,   /* no usable gfx950 device / HIP runtime error at create */
    SK_ERR_HIP = -3,         /* a HIP call failed (sk_engine_last_hip_error) */
    SK_ERR_OOM = -4,
    SK_ERR_BAD_STREAM = -5,  /* stream id not open */
    SK_ERR_UNSUPPORTED = -6, /* e.g. resample ratio other than 48k->16k on the MFMA path */
    SK_ERR_CAPACITY = -7,    /* max_streams exhausted */
    SK_ERR_TIMEOUT = -8,     /* the device did not finish a tick within the engine's wait bound (sk_engine_set_wait_bound) */
    SK_ERR_INTERNAL = -9     /* a C++ exception other than std::bad_alloc (that one: SK_ERR_OOM) was caught at the ABI; sk_last_exception() */
} sk_status;

/* per-frame status words written by the AAC entry points */
typedef enum sk_frame_status {
    SK_FRAME_OK = 0,
    SK_FRAME_BAD_STREAM = 1,   /* stream not open */
    SK_FRAME_BAD_CHANNELS = 2, /* desc.channels != the stream's channel count */
    SK_FRAME_BAD_WINDOW = 3    /* window_sequence > 3 or window_shape > 1 */
} sk_frame_status;

/* ics.rs:7-12 WindowSequence (bitstream coding) / ics.rs:32-35 WindowShape */
enum { SK_ONLY_LONG = 0, SK_LONG_START = 1, SK_EIGHT_SHORT = 2, SK_LONG_STOP = 3 };
enum { SK_SHAPE_SINE = 0, SK_SHAPE_KBD = 1 };

typedef struct sk_engine sk_engine; /* one per GPU */

/* One decoded-but-not-yet-synthesised AAC-LC frame: what decoder.rs:336-374
 * synthesize_channel reads from IcsInfo per channel. */
typedef struct sk_aac_frame_desc {
    uint32_t stream;
    uint8_t channels;           /* 1 or 2; must equal the stream's */
    uint8_t window_sequence[2]; /* per channel */
    uint8_t window_shape[2];    /* per channel */
    uint8_t reserved[3];
} sk_aac_frame_desc;

/* ---- engine / stream lifetime ------------------------------------------- */

int sk_engine_create(int device, uint32_t max_streams, sk_engine **out);
void sk_engine_destroy(sk_engine *);
int sk_engine_device(const sk_engine *);
uint32_t sk_engine_max_streams(const sk_engine *);
void *sk_engine_hip_stream(sk_engine *); /* hipStream_t */
int sk_engine_synchronize(sk_engine *);
const char *sk_engine_last_hip_error(const sk_engine *);
/* 1 when the kernels were built with the packed-f32 vector instructions (make PACKED_F32=1), 0 in the default build.  On this
 * platform a packed-f32 instruction can deliver a wrong low half while another wave of its CU executes a 16x16x32 matrix
 * instruction (profiles/r04_lanes_corruption.md): a flavour-1 library is 2-3 % faster on the decode tail and must have its GPU to
 * itself -- no second decoding process, no GEMM workload on the same CUs; the default is immune.  No reference counterpart. */
int sk_kernels_use_packed_f32(void);
/* Diagnostics (no reference counterpart).  sk_engine_where: the stage the engine's current tick is in, as static text,
 * readable from any thread without the engine's lock ("idle" outside a tick).  sk_engine_set_wait_bound: how long a
 * tick waits for the device before it gives up with SK_ERR_TIMEOUT (default 120 s) instead of blocking for good.
 * sk_engine_debug_fail_after: the n-th HIP call from now fails as a launch failure (error-path tests; 0 disarms). */
const char *sk_engine_where(const sk_engine *);
int sk_engine_set_wait_bound(sk_engine *, double seconds);
/* Generic-ratio resampling (every pair of soundkit's common rates other than 48 -> 16 kHz) runs on the matrix cores, held to
 * the float tolerance like the 48 -> 16 kHz FIR.  exact = 1 selects the scalar form instead, whose sums keep rubato's order
 * of operations: the restated reference bit for bit, about five times slower. */
int sk_engine_set_resampler_exact(sk_engine *, int exact);
int sk_engine_debug_fail_after(sk_engine *, int n_hip_calls);
const char *sk_strerror(int status);
/* Exception barrier: no C++ exception leaves the library -- every entry point catches what is thrown below it and returns
 * SK_ERR_OOM (std::bad_alloc) or SK_ERR_INTERNAL; worker threads of the scheduler turn an exception into the error of the
 * stream they were serving (soundkit-decoder/src/lib.rs:3131-3134: an error ends that stream only).  Text of the last
 * exception caught on the calling thread ("" if none): */
const char *sk_last_exception(void);
/* Test hook: the (n+1)-th entry into the library from now throws (kind 0 std::bad_alloc, 1 std::length_error, 2 a type that
 * is no std::exception) as if an allocation inside had failed; n < 0 switches it off.  Returns the previous countdown. */
int sk_debug_throw_after(int n, int kind);
/* Test hook for the scheduler's own threads: the (n+1)-th passage of point `where` throws std::bad_alloc -- 0: inside the
 * per-stream guard of an entropy thread (that stream ends with SK_ERR_OOM, the others go on); 1 entropy thread, 2 submission
 * thread, 3 delivery thread outside any per-stream guard (the lane stops: every open stream ends with the error, later
 * spawns return it, the process lives). */
int sk_debug_throw_in_thread(int n, int where);
const char *sk_version(void);

/* AacLcDecoder::new (decoder.rs:56-78): allocates the per-channel DspChannel state
 * (delay zeroed, previous shape = Sine, dsp.rs:162-163). */
int sk_stream_open(sk_engine *, uint32_t sample_rate, uint8_t channels, uint32_t *stream_out);
int sk_stream_close(sk_engine *, uint32_t stream);
int sk_stream_reset(sk_engine *, uint32_t stream);
/* checkpoint / inspection of the carried state: delay [channels][1024] f32, prev_shape [channels] */
int sk_stream_get_state(sk_engine *, uint32_t stream, float *delay_out, uint8_t *prev_shape_out);
int sk_stream_set_state(sk_engine *, uint32_t stream, const float *delay, const uint8_t *prev_shape);

/* ---- AAC-LC synthesis: IMDCT + window + overlap-add ---------------------- */
/* Replaces AacLcDecoder::synthesize_channel (decoder.rs:336-374) =
 * DspChannel::synthesize_long_sequence / synthesize_eight_short (dsp.rs:230-338) over
 * imdct_fast (dsp.rs:476-535), for n frames at once.
 * coeffs: frame i holds desc[i].channels x 1024 f32, planar, frames packed back to back
 *         (the dequantised, stereo- and TNS-processed spectra: decoder.rs:341, 358).
 * pcm_out (f32): same packing, planar f32 = PlanarF32 (decoder.rs:22-36, 385-390).
 * pcm_out (s16): frame i holds 1024 x channels interleaved i16 = decode_aac_access_unit
 *         (soundkit-decoder lib.rs:1793-1813) with float_sample_to_i16 (lib.rs:1815-1827).
 * status_per_frame may be NULL.  Output of a frame whose status != 0 is not written. */
int sk_aac_synthesize_f32(sk_engine *, const sk_aac_frame_desc *descs, const float *coeffs, float *pcm_out,
                          uint32_t n, int32_t *status_per_frame);
int sk_aac_synthesize_s16(sk_engine *, const sk_aac_frame_desc *descs, const float *coeffs, int16_t *pcm_out,
                          uint32_t n, int32_t *status_per_frame);

/* The same, split so that a schedule can be validated/uploaded once and run on
 * device-resident spectra (the throughput path: no PCIe inside the run). */
typedef struct sk_aac_plan sk_aac_plan;
int sk_aac_plan_create(sk_engine *, const sk_aac_frame_desc *descs, uint32_t n, int32_t *status_per_frame,
                       sk_aac_plan **out);
void sk_aac_plan_destroy(sk_aac_plan *);
uint64_t sk_aac_plan_elements(const sk_aac_plan *); /* total f32 elements of coeffs / pcm */
uint32_t sk_aac_plan_frames_ok(const sk_aac_plan *);
int sk_aac_plan_run_f32_dev(sk_engine *, const sk_aac_plan *, const float *d_coeffs, float *d_pcm);
int sk_aac_plan_run_s16_dev(sk_engine *, const sk_aac_plan *, const float *d_coeffs, int16_t *d_pcm);
/* The s16 of decode_aac_access_unit (soundkit-decoder lib.rs:1793-1813: float_sample_to_i16 of every sample) written by
 * the synthesis kernel itself, PLANAR: the packing of the f32 form with two bytes per sample (channel c of frame i at
 * element (off_i + c) * 1024).  This is the intermediate of the worker's decode -> resample path -- the resampler is fed
 * these integers / 32768 (lib.rs:3563-3617) -- kept on the device at half the bytes of the f32 PCM;
 * sk_downsample_48k_16k_frames_s16_to_s16_dev reads it in place.  d_pcm16 is 8-byte aligned. */
int sk_aac_plan_run_s16_planar_dev(sk_engine *, const sk_aac_plan *, const float *d_coeffs, int16_t *d_pcm16);

/* ---- AAC-LC access-unit front-end (host cores) ------------------------------------------------
 * The entropy / side-information half of AacLcDecoder::decode_access_unit (soundkit-aac-lc/src/
 * decoder.rs:104-334): element loop, ICS info, sections, scalefactors, pulse, Huffman spectral decode
 * with dequantisation, PNS, intensity + mid/side stereo, TNS.  It stops where the GPU takes over:
 * sk_aac_decoder_parse fills the dequantised spectra [channels][1024] and the window fields of one
 * sk_aac_frame_desc (the caller sets .stream), ready for sk_aac_synthesize_* / sk_aac_plan_create.
 * Errors mirror AacLcError (error.rs:5-18); sk_aac_decoder_last_error carries the reference's message. */
typedef enum sk_aac_status {
    SK_AAC_ERR_EOF = -101,                        /* UnexpectedEof */
    SK_AAC_ERR_INVALID_AOT = -102,                /* InvalidAudioObjectType */
    SK_AAC_ERR_UNSUPPORTED_AOT = -103,            /* UnsupportedAudioObjectType */
    SK_AAC_ERR_UNSUPPORTED_SF_INDEX = -104,       /* UnsupportedSamplingFrequencyIndex */
    SK_AAC_ERR_UNSUPPORTED_CHANNEL_CONFIG = -105, /* UnsupportedChannelConfig */
    SK_AAC_ERR_UNSUPPORTED_FEATURE = -106,        /* UnsupportedFeature: callers fall back to another decoder */
    SK_AAC_ERR_INVALID_CONFIG = -107,             /* InvalidConfig */
    SK_AAC_ERR_INVALID_BITSTREAM = -108           /* InvalidBitstream */
} sk_aac_status;
typedef struct sk_aac_decoder sk_aac_decoder;
/* AacLcDecoder::from_audio_specific_config (decoder.rs:80, config.rs:139-260) */
int sk_aac_decoder_create(const uint8_t *asc, size_t asc_len, sk_aac_decoder **out);
void sk_aac_decoder_destroy(sk_aac_decoder *);
int sk_aac_decoder_info(const sk_aac_decoder *, uint32_t *sample_rate, uint8_t *channels); /* frame_info, decoder.rs:88 */
const char *sk_aac_decoder_last_error(const sk_aac_decoder *);
/* counters since creation: frames, EightShort channel-frames, LongStart/Stop channel-frames, channel-frames with TNS,
 * PNS bands, intensity bands, mid/side bands, channel-frames with pulse data (aac-wasm-bench lib.rs:1955-1986) */
int sk_aac_decoder_tool_usage(const sk_aac_decoder *, uint32_t out[8]);
int sk_aac_decoder_parse(sk_aac_decoder *, const uint8_t *access_unit, size_t len, float *coeffs /*[ch][1024]*/,
                         sk_aac_frame_desc *desc);
/* parse_adts_access_unit (soundkit-decoder/src/lib.rs:1007-1027): the 2-byte ASC and the raw access unit of
 * one ADTS frame; frame_len = bytes to the next frame header. */
/* The Huffman half of the front-end alone ("GPU dequant + stereo tools + TNS fed by i16 quantized
 * values + scalefactor bytes"): quant receives channels x 1024 i16 quantised values (pulses applied, spectral.rs:327-423,
 * 2198-2247), side a SK_AAC_UNIT_SIDE_BYTES record (section codebooks, transmitted scale factors, grouping, mid/side mask,
 * TNS filters, noise-sample count, and the verdict on the rest of the unit), desc the window fields.  Dequantisation
 * (dsp.rs:397-405), PNS, intensity + mid/side and TNS then run on the device: sk_tick_run_q.  5.6 KiB per stereo
 * access unit cross PCIe instead of 8 KiB.  A quantised magnitude beyond i16 (illegal in ISO/IEC 14496-3, which stops at
 * 8191; the reference accepts escapes up to 2^17, spectral.rs:214-228) travels in the record's list of wide values, up
 * to 48 per access unit; the 49th is SK_AAC_ERR_UNSUPPORTED_FEATURE in this mode only. */
#define SK_AAC_UNIT_SIDE_BYTES 1600u
int sk_aac_decoder_parse_q(sk_aac_decoder *, const uint8_t *access_unit, size_t len, int16_t *quant /*[ch][1024]*/,
                           void *side /*SK_AAC_UNIT_SIDE_BYTES*/, sk_aac_frame_desc *desc);
int sk_adts_parse(const uint8_t *data, size_t len, size_t *frame_len, size_t *payload_off, size_t *payload_len,
                  uint8_t asc[2]);

/* ---- the audio_packet::Decoder surface for one ADTS AAC-LC stream ------------------------------------ */
/* soundkit::audio_packet::Decoder (soundkit/src/audio_packet.rs:22-26) as soundkit-aac's AacDecoder implements it
 * (soundkit-aac/src/lib.rs:108-266) and the worker drives it (decode_i16_with_drain, soundkit-decoder lib.rs:
 * 2150-2181): each call appends `input` (<= 4 MiB per call and buffered) and decodes every whole frame that is
 * buffered and fits in `output`; *written = interleaved samples; 0 = needs more input / drained.  Call once with
 * data, then with len 0 until it returns 0.  decode_i32 is "Not implemented." in the reference and absent here. */
typedef struct sk_adts_decoder sk_adts_decoder;
int sk_adts_decoder_create(sk_engine *, sk_adts_decoder **out);
void sk_adts_decoder_destroy(sk_adts_decoder *);
int sk_adts_decoder_decode_i16(sk_adts_decoder *, const uint8_t *input, size_t len, int16_t *output, size_t out_cap,
                               size_t *written);
int sk_adts_decoder_decode_f32(sk_adts_decoder *, const uint8_t *input, size_t len, float *output, size_t out_cap,
                               size_t *written);
/* sample_rate() / channels() (lib.rs:133-139): 0 until the first frame has been decoded */
int sk_adts_decoder_info(const sk_adts_decoder *, uint32_t *sample_rate, uint8_t *channels);
const char *sk_adts_decoder_last_error(const sk_adts_decoder *);

/* dsp.rs:389-450 dequantize_signed_scaled over a batch: out[i] = sign(q)*|q|^(4/3)*2^((sf-100)/4).
 * quant: n i16 quantised values; sf_per_band_of[i]: i16 scale factor applying to value i. */
int sk_aac_dequantize_dev(sk_engine *, const int16_t *d_quant, const int16_t *d_scalefactor, float *d_out, size_t n);
int sk_aac_dequantize(sk_engine *, const int16_t *quant, const int16_t *scalefactor, float *out, size_t n);

/* ---- MPEG-1/2 Layer III hybrid synthesis ----------------------------------------------------------------
 * The transform half of what Mp3Decoder gets from nanomp3::Decoder::decode (soundkit-mp3/src/lib.rs:279-305, :284): for
 * every granule and channel, 576 requantised / stereo-processed / reordered frequency lines -> alias reduction ->
 * IMDCT 36 or 3 x 12 with the block-type windows -> overlap-add -> frequency inversion -> 32-band polyphase synthesis
 * (ISO/IEC 11172-3 2.4.3.4), 576 PCM samples out, float in +-1.0 interleaved as nanomp3 emits them, or s16 through
 * f32_to_i16 (lib.rs:376-385).  Carried state (overlap, polyphase FIFO) is per engine stream, reset by sk_stream_open /
 * sk_stream_reset.  The synthesis window D (Table B.3) is set once per engine: sk_mp3_decoder_create does it (with the
 * standard's table, csrc/mp3_iso_tables.h, unless the caller brings its own); callers of the bare stage call
 * sk_mp3_set_synthesis_window -- until then the synthesis entry points return SK_ERR_UNSUPPORTED.  Sample parity with
 * nanomp3 is unpinned (its source is absent); what pins this row is the decode of the reference's two MP3 fixtures against
 * the source PCM the reference holds for them (tests/test_mp3_fixtures_gpu.py).
 * Lines of a short block (block_type 2) are window-interleaved: X_w[m] = xr[18 sb + 3 m + w]. */
typedef struct sk_mp3_granule_desc {
    uint32_t stream;
    uint8_t channels;            /* 1 or 2; must equal the stream's */
    uint8_t block_type[2];       /* per channel: 0 normal, 1 start, 2 short, 3 stop */
    uint8_t mixed_block_flag[2]; /* per channel, meaningful with block_type 2 */
    uint8_t reserved[3];
} sk_mp3_granule_desc;
int sk_mp3_set_synthesis_window(sk_engine *, const float *d512);
/* xr: granule i holds channels x 576 f32, granules packed back to back; pcm_out: granule i holds 576 x channels
 * interleaved samples at the same packing.  A granule whose status != 0 leaves silence. */
int sk_mp3_hybrid_synthesize_f32(sk_engine *, const sk_mp3_granule_desc *descs, const float *xr, float *pcm_out, uint32_t n,
                                 int32_t *status_per_granule);
int sk_mp3_hybrid_synthesize_s16(sk_engine *, const sk_mp3_granule_desc *descs, const float *xr, int16_t *pcm_out, uint32_t n,
                                 int32_t *status_per_granule);
int sk_mp3_hybrid_synthesize_f32_dev(sk_engine *, const sk_mp3_granule_desc *descs, const float *d_xr, float *d_pcm, uint32_t n,
                                     int32_t *status_per_granule);

/* ---- MPEG Layer III: the fixed-syntax front of the bitstream half, and requantisation / stereo / reorder -------------
 * What nanomp3::Decoder::decode (soundkit-mp3/src/lib.rs:284) does between the bytes and its Huffman stage, and between that
 * stage and the hybrid synthesis above -- the parts that are closed-form syntax and arithmetic of ISO/IEC 11172-3 /
 * 13818-3.  The band tables (Table B.8) and the pre-emphasis table are set per engine and sampling rate like the synthesis
 * window (sk_mp3_decoder_create does it).  The reference's two MP3 fixtures pin the framing (tests/test_mp3_bitstream.py). */
enum sk_mp3_status {
    SK_MP3_NEED_MORE = -301,   /* not enough bytes for the header / side information / frame / reservoir */
    SK_MP3_NO_SYNC = -302,     /* not a frame header */
    SK_MP3_UNSUPPORTED = -303, /* Layer I / II; a free-format header whose frame length is not known (sk_mp3_scan_free measures it) */
    SK_MP3_INVALID = -304      /* a field combination the syntax forbids */
};
typedef struct sk_mp3_frame_info { /* nanomp3::FrameInfo (lib.rs:188-215 reads sample_rate, channels, bitrate) + framing */
    uint32_t offset;             /* sk_mp3_scan: position of the frame in the scanned buffer */
    uint32_t frame_bytes;        /* header to the next header */
    uint32_t sample_rate;
    uint16_t bitrate_kbps, samples_per_channel; /* 1152 (MPEG-1) or 576 (MPEG-2 / 2.5) */
    uint8_t version;             /* 1, 2 or 25 (= MPEG-2.5) */
    uint8_t channels, mode, mode_ext, has_crc, padding, granules, side_info_bytes;
} sk_mp3_frame_info;
typedef struct sk_mp3_granule_side { /* 11172-3 2.4.1.7 / 13818-3 2.4.1.7, one granule of one channel */
    uint16_t part2_3_length, big_values, scalefac_compress;
    uint8_t global_gain, window_switching, block_type, mixed_block_flag, table_select[3], subblock_gain[3], region0_count,
        region1_count, preflag, scalefac_scale, count1table_select, reserved;
} sk_mp3_granule_side;
typedef struct sk_mp3_side_info {
    uint16_t main_data_begin;
    uint8_t granules, channels;
    uint8_t scfsi[2][4];
    sk_mp3_granule_side gr[2][2]; /* [granule][channel] */
} sk_mp3_side_info;
int sk_mp3_parse_header(const uint8_t *data, size_t len, sk_mp3_frame_info *out);
int sk_mp3_parse_side_info(const uint8_t *frame, size_t len, const sk_mp3_frame_info *header, sk_mp3_side_info *out);
/* every frame of a byte stream (an ID3v2 tag in front is stepped over, garbage skipped, a header counts only if the next
 * frame's header follows it); *consumed = bytes up to the first incomplete frame */
int sk_mp3_scan(const uint8_t *data, size_t len, sk_mp3_frame_info *frames, uint32_t cap, uint32_t *n_frames, size_t *consumed);
/* The same with free-format streams (bit-rate index 0): the frame length is what lies between a header and the next two of the same
 * stream, as minimp3's mp3d_find_frame measures it (nanomp3 is its port; soundkit-mp3/src/lib.rs:284).  *free_format_bytes carries
 * that length (without the padding slot) from call to call, 0 at the start of a stream; sk_mp3_parse_header_free reads a header with
 * it.  The decoder handles and the batch scheduler use these two. */
int sk_mp3_scan_free(const uint8_t *data, size_t len, sk_mp3_frame_info *frames, uint32_t cap, uint32_t *n_frames, size_t *consumed,
                     uint32_t *free_format_bytes);
int sk_mp3_parse_header_free(const uint8_t *data, size_t len, uint32_t free_format_bytes, sk_mp3_frame_info *out);
/* the bytes parts 2 + 3 of a frame are read from: main_data_begin bytes of the reservoir + the frame's own main data */
int sk_mp3_main_data(const uint8_t *frame, size_t frame_len, const sk_mp3_frame_info *header, const sk_mp3_side_info *side,
                     const uint8_t *prev_main_data, size_t prev_len, uint8_t *out, size_t out_cap, size_t *out_len);

/* Requantisation (2.4.3.4.7.1), mid/side and MPEG-1 intensity stereo (2.4.3.4.9-10) and the short-block reorder
 * (2.4.3.4.8) for a batch of granules on the GPU: is = the Huffman stage's integers, [granule][channel][576] in bitstream
 * order; xr = what sk_mp3_hybrid_synthesize_* takes.  Band tables (Table B.8: 23 long and 14 short offsets per sampling
 * rate) and the pre-emphasis table (22 entries) come from the caller, once per engine and sampling rate. */
typedef struct sk_mp3_requant_channel {
    uint8_t global_gain, scalefac_scale, preflag, block_type, mixed_block_flag, subblock_gain[3];
    uint8_t scalefac_l[22];    /* long bands (also the long part of a mixed block) */
    uint8_t scalefac_s[13][3]; /* short bands x windows */
    uint8_t reserved;
} sk_mp3_requant_channel;
typedef struct sk_mp3_requant_granule {
    uint32_t sample_rate;
    uint8_t channels, ms_stereo, intensity_stereo, lsf; /* mode_ext bits of a joint-stereo frame; lsf: MPEG-2 / 2.5.  intensity_stereo: bit 0
                                                         * = on; bit 1 (lsf only) = intensity_scale, the low bit of the right channel's
                                                         * scalefac_compress (13818-3 2.4.3.2: i0 = 2^-1/2 instead of 2^-1/4).  In an lsf
                                                         * intensity granule bit 7 of a RIGHT-channel scale factor marks its position as
                                                         * "not intensity coded" (the largest value its field holds); the factor is bits 0-6 */
    sk_mp3_requant_channel ch[2];
} sk_mp3_requant_granule;
int sk_mp3_set_band_tables(sk_engine *, uint32_t sample_rate, const uint16_t long_offsets[23], const uint16_t short_offsets[14],
                           const uint8_t pretab[22]);
int sk_mp3_requantize(sk_engine *, const sk_mp3_requant_granule *granules, const int16_t *is, float *xr, uint32_t n,
                      int32_t *status_per_granule);
/* The two GPU stages back to back -- sk_mp3_requantize then sk_mp3_hybrid_synthesize_* with descs[i] describing the same
 * granule as granules[i] -- the frequency lines staying on the device: integers in, interleaved PCM out, one
 * synchronisation.  What sk_mp3_decoder_decode_* runs per call. */
int sk_mp3_decode_granules_f32(sk_engine *, const sk_mp3_requant_granule *granules, const sk_mp3_granule_desc *descs, const int16_t *is,
                               float *pcm_out, uint32_t n, int32_t *status_per_granule);
int sk_mp3_decode_granules_s16(sk_engine *, const sk_mp3_requant_granule *granules, const sk_mp3_granule_desc *descs, const int16_t *is,
                               int16_t *pcm_out, uint32_t n, int32_t *status_per_granule);

/* ---- MPEG Layer III: scale factors, the Huffman stage and a decoder handle in Mp3Decoder's shape ----------------------
 * Parts 2 and 3 of the main data are syntax over data tables of the standard.  They enter in the standard's own
 * presentation -- for every code its length and its bits -- and the library builds its decoding structures from that.
 * sk_mp3_iso_tables hands out the standard's tables (csrc/mp3_iso_tables.h: normative constants of ISO/IEC 11172-3 B.3 /
 * B.6 / B.7 / B.8 and 13818-3 2.4.3.2, written by tools/transcribe_iso_mp3_tables.py, which records where they were read
 * and the checks that gate them); a caller may pass any other complete prefix code set instead (the syntax tests do). */
typedef struct sk_mp3_code_table { /* one table of ISO/IEC 11172-3 Table B.7 */
    uint8_t xlen;                  /* values 0 .. xlen-1 for x and for y; 0 = table carries no codes (tables 0, 4, 14) */
    uint8_t linbits;
    const uint8_t *hlen;           /* [xlen * xlen], index x * xlen + y: code length in bits, 1..32 */
    const uint32_t *hcod;          /* the code, right-aligned */
} sk_mp3_code_table;
typedef struct sk_mp3_tables {
    sk_mp3_code_table big_values[32];
    uint8_t count1_hlen[2][16];    /* tables A and B of the count1 region, index v << 3 | w << 2 | x << 1 | y */
    uint8_t count1_hcod[2][16];
    uint8_t slen[16][2];           /* 11172-3 2.4.2.7: scalefac_compress -> slen1, slen2 */
    uint8_t lsf_partitions[6][3][4]; /* 13818-3 2.4.3.2: scale factors per partition, [row][long | short | mixed][partition] */
    uint16_t long_offsets[9][23];  /* Table B.8 by sampling rate: 44100 48000 32000 22050 24000 16000 11025 12000 8000 */
    uint16_t short_offsets[9][14];
    uint8_t rates_present[9];      /* which rows of the two arrays above are filled in */
    uint8_t pretab[22];
    float window[512];             /* Table B.3 */
} sk_mp3_tables;
typedef struct sk_mp3_codebook sk_mp3_codebook; /* host-side: validated copies + prefix-decoding tables; no GPU involved */
int sk_mp3_codebook_create(const sk_mp3_tables *, sk_mp3_codebook **out); /* SK_MP3_INVALID: a code set that is no prefix code */
void sk_mp3_codebook_destroy(sk_mp3_codebook *);
int sk_mp3_iso_tables(sk_mp3_tables *out);            /* the standard's tables; hlen / hcod point to static storage */
int sk_mp3_codebook_create_iso(sk_mp3_codebook **out); /* = sk_mp3_iso_tables + sk_mp3_codebook_create */

/* One granule of one channel out of parts 2 + 3: the scale factors in sk_mp3_requantize's layout and the 576 integers. */
typedef struct sk_mp3_granule_data {
    int16_t is[576];
    uint8_t scalefac_l[22];
    uint8_t scalefac_s[13][3];
    uint8_t preflag;     /* MPEG-1: the side information's; LSF: implied by scalefac_compress */
    uint8_t intensity_scale; /* LSF intensity channel: scalefac_compress & 1 (-> sk_mp3_requant_granule::intensity_stereo bit 1); its
                              * scale factors carry bit 7 where the position is the field's largest value ("not intensity coded") */
    uint16_t part2_bits; /* what the scale factors took of part2_3_length */
    uint16_t nonzero_lines; /* lines up to and including the last decoded pair / quadruple */
    uint16_t part3_bits; /* what the accepted pairs / quadruples took; a well-formed granule: part2_bits + part3_bits == part2_3_length */
    int32_t status;      /* SK_OK | SK_MP3_INVALID (a bit pattern that is no code, values past line 576) | SK_MP3_UNSUPPORTED */
} sk_mp3_granule_data;
/* main = what sk_mp3_main_data assembled.  out[granule][channel]; previous = the same frame's granule-0 scale factors are
 * taken from out itself (scfsi). */
int sk_mp3_decode_main_data(const sk_mp3_codebook *, const sk_mp3_frame_info *header, const sk_mp3_side_info *side, const uint8_t *main,
                            size_t main_len, sk_mp3_granule_data out[2][2]);

/* Mp3Decoder (soundkit-mp3/src/lib.rs:147-374): bytes in at any chunking, interleaved PCM out.  One call decodes every
 * complete frame its input buffer holds -- framing, reservoir, scale factors and Huffman on the host, then ONE
 * requantisation launch and ONE hybrid-synthesis launch over all their granules -- subject to the reference's output rule:
 * it stops once fewer than SK_MP3_MAX_SAMPLES_PER_FRAME samples of room are left (lib.rs:300-302) and fails with
 * SK_ERR_CAPACITY if a frame does not fit (lib.rs:237-243).  Input beyond 4 MiB buffered: SK_PIPE_CHUNK_TOO_LARGE
 * (lib.rs:155, 219-227).  A frame whose main data reaches further back than the reservoir holds (a stream joined in the
 * middle) is consumed without output, as are frames the later stages reject. */
#define SK_MP3_MAX_SAMPLES_PER_FRAME 2304u
typedef struct sk_mp3_decoder sk_mp3_decoder;
/* Mp3Decoder::new, lib.rs:157.  codebook NULL = the standard's tables (what nanomp3 has built in); the decoder installs the
 * codebook's band tables and synthesis window on the engine. */
int sk_mp3_decoder_create(sk_engine *, const sk_mp3_codebook *, sk_mp3_decoder **out);
void sk_mp3_decoder_destroy(sk_mp3_decoder *);
int sk_mp3_decoder_reset(sk_mp3_decoder *);                                           /* lib.rs:180-185 */
/* sample_rate / channels are 0 until the first frame was decoded (Option::None, lib.rs:167-173); buffer_len: lib.rs:176 */
int sk_mp3_decoder_info(const sk_mp3_decoder *, uint32_t *sample_rate, uint8_t *channels, size_t *buffer_len, uint64_t *frames_decoded);
int sk_mp3_decoder_decode_i16(sk_mp3_decoder *, const uint8_t *input, size_t len, int16_t *out, size_t out_cap, size_t *written);
int sk_mp3_decoder_decode_i32(sk_mp3_decoder *, const uint8_t *input, size_t len, int32_t *out, size_t out_cap, size_t *written);
int sk_mp3_decoder_decode_f32(sk_mp3_decoder *, const uint8_t *input, size_t len, float *out, size_t out_cap, size_t *written);

/* ---- sample-width / interleave conversion: soundkit::audio_bytes -------- */
/* Elementwise ops; n = number of OUTPUT samples.  Citations: soundkit/src/audio_bytes.rs
 * unless noted. */
typedef enum sk_pcm_op {
    SK_PCM_I16LE_TO_F32 = 0,      /* :3   i16le_to_f32 */
    SK_PCM_I16_TO_I16LE = 1,      /* :17  i16_to_i16le */
    SK_PCM_I16LE_TO_I16 = 2,      /* :25  i16le_to_i16 */
    SK_PCM_S24LE_TO_I32 = 3,      /* :36  s24le_to_i32 */
    SK_PCM_S24LE_TO_I16 = 4,      /* :51  s24le_to_i16 */
    SK_PCM_S24BE_TO_I16 = 5,      /* :66  s24be_to_i16 */
    SK_PCM_S32LE_TO_I32 = 6,      /* :81  s32le_to_i32 */
    SK_PCM_S32BE_TO_I32 = 7,      /* :91  s32be_to_i32 */
    SK_PCM_S32LE_TO_S24 = 8,      /* :101 s32le_to_s24 */
    SK_PCM_S32BE_TO_S24 = 9,      /* :112 s32be_to_s24 */
    SK_PCM_S32LE_TO_F32 = 10,     /* :123 s32le_to_f32 */
    SK_PCM_S32BE_TO_F32 = 11,     /* :134 s32be_to_f32 */
    SK_PCM_S32LE_TO_I16 = 12,     /* :145 s32le_to_i16 */
    SK_PCM_S32BE_TO_I16 = 13,     /* :156 s32be_to_i16 */
    SK_PCM_F32LE_TO_I16 = 14,     /* :167 f32le_to_i16 */
    SK_PCM_F32BE_TO_I16 = 15,     /* :178 f32be_to_i16 */
    SK_PCM_F32LE_TO_I32 = 16,     /* :189 f32le_to_i32 */
    SK_PCM_F32LE_TO_S24 = 17,     /* :205 f32le_to_s24 */
    SK_PCM_S16BE_TO_I16 = 18,     /* :222 s16be_to_i16 */
    SK_PCM_S16LE_TO_I16 = 19,     /* :231 s16le_to_i16 */
    SK_PCM_S16LE_TO_I32 = 20,     /* :240 s16le_to_i32 */
    SK_PCM_STEREO_TO_MONO_TAKE_LEFT = 21, /* :317 / :331 */
    SK_PCM_STEREO_TO_MONO_AVG = 22,       /* :344 / :360 */
    SK_PCM_VEC_F32_TO_I16 = 23,   /* soundkit/src/audio_pipeline.rs:17 */
    SK_PCM_VEC_I16_TO_F32 = 24,   /* soundkit/src/audio_pipeline.rs:29 */
    SK_PCM_VEC_I32_TO_F32 = 25,   /* soundkit/src/audio_pipeline.rs:40 */
    SK_PCM_FLOAT_TO_I16_ROUND = 26, /* soundkit-decoder/src/lib.rs:1815 float_sample_to_i16 */
    SK_PCM_MP3_F32_TO_I16 = 27,   /* soundkit-mp3/src/lib.rs:376 f32_to_i16 */
    SK_PCM_MP3_F32_TO_I32 = 28,   /* soundkit-mp3/src/lib.rs:387 f32_to_i32 */
    SK_PCM_OP_COUNT = 29
} sk_pcm_op;
int sk_pcm_op_in_bytes(int op);  /* bytes consumed per output sample */
int sk_pcm_op_out_bytes(int op); /* bytes produced per output sample */
int sk_pcm_convert(sk_engine *, int op, const void *in, void *out, size_t n);
int sk_pcm_convert_dev(sk_engine *, int op, const void *d_in, void *d_out, size_t n);

typedef enum sk_pcm_fmt {
    SK_FMT_S16LE = 0, SK_FMT_S16BE = 1, SK_FMT_S24LE = 2, SK_FMT_S24BE = 3,
    SK_FMT_S32LE = 4, SK_FMT_S32BE = 5, SK_FMT_F32LE = 6, SK_FMT_F32BE = 7
} sk_pcm_fmt;
int sk_pcm_fmt_bytes(int fmt);

/* Layout-only ops: interleave_vecs_i16 :250, deinterleave_vecs_i16 :264,
 * deinterleave_vecs_s24 :280 (widens to i32), deinterleave_vecs_f32 :296,
 * interleave_vecs_f32 (soundkit-decoder lib.rs:3685).  planar = [ch][frames]. */
int sk_pcm_interleave_i16(sk_engine *, const int16_t *planar, size_t frames, uint32_t ch, uint8_t *out);
int sk_pcm_deinterleave_i16(sk_engine *, const uint8_t *in, size_t frames, uint32_t ch, int16_t *planar);
int sk_pcm_deinterleave_s24(sk_engine *, const uint8_t *in, size_t frames, uint32_t ch, int32_t *planar);
int sk_pcm_deinterleave_f32(sk_engine *, const uint8_t *in, size_t frames, uint32_t ch, float *planar);
int sk_pcm_interleave_f32(sk_engine *, const float *planar, size_t frames, uint32_t ch, uint8_t *out);
int sk_pcm_interleave_i16_dev(sk_engine *, const int16_t *d_planar, size_t frames, uint32_t ch, uint8_t *d_out);
int sk_pcm_deinterleave_i16_dev(sk_engine *, const uint8_t *d_in, size_t frames, uint32_t ch, int16_t *d_planar);
int sk_pcm_deinterleave_s24_dev(sk_engine *, const uint8_t *d_in, size_t frames, uint32_t ch, int32_t *d_planar);
int sk_pcm_deinterleave_f32_dev(sk_engine *, const uint8_t *d_in, size_t frames, uint32_t ch, float *d_planar);
int sk_pcm_interleave_f32_dev(sk_engine *, const float *d_planar, size_t frames, uint32_t ch, uint8_t *d_out);

/* interleaved bytes -> planar f32.  variant 0 = soundkit-decoder audio_data_to_f32_channels
 * (lib.rs:3563-3617; any fmt; non-finite -> 0); variant 1 = soundkit::audio_to_f32_channels
 * (audio_pipeline.rs:74-98; LE only; s24 is divided by 2^31 as the reference does). */
int sk_pcm_bytes_to_f32_planar(sk_engine *, int variant, int fmt, const uint8_t *in, size_t frames, uint32_t ch,
                               float *planar);
int sk_pcm_bytes_to_f32_planar_dev(sk_engine *, int variant, int fmt, const uint8_t *d_in, size_t frames,
                                   uint32_t ch, float *d_planar);
/* planar f32 -> interleaved bytes: f32_channels_to_bytes (soundkit-decoder lib.rs:3619-3683);
 * fmt in {S16LE, S24LE, S32LE, F32LE} */
int sk_pcm_f32_planar_to_bytes(sk_engine *, int fmt, const float *planar, size_t frames, uint32_t ch, uint8_t *out);
int sk_pcm_f32_planar_to_bytes_dev(sk_engine *, int fmt, const float *d_planar, size_t frames, uint32_t ch,
                                   uint8_t *d_out);
/* the same for `batch` independent signals: planar [batch][ch][plane_stride] -> [batch][frames][ch] */
int sk_pcm_f32_planar_to_bytes_batch_dev(sk_engine *, int fmt, const float *d_planar, size_t batch, size_t plane_stride,
                                         size_t frames, uint32_t ch, uint8_t *d_out);
/* downmix_channels target 1 (soundkit-decoder lib.rs:3500-3509) */
int sk_pcm_downmix_mono(sk_engine *, const float *planar, size_t frames, uint32_t ch, float *mono);
int sk_pcm_downmix_mono_dev(sk_engine *, const float *d_planar, size_t frames, uint32_t ch, float *d_mono);
/* exact_signed_pcm_to_i16 (soundkit-decoder lib.rs:3458-3489); fmt in {S24LE,S24BE,S32LE,S32BE} */
int sk_pcm_exact_to_i16(sk_engine *, int fmt, const uint8_t *in, size_t samples, uint8_t *out_s16le);
int sk_pcm_exact_to_i16_dev(sk_engine *, int fmt, const uint8_t *d_in, size_t samples, uint8_t *d_out_s16le);

/* ---- 48 kHz -> 16 kHz sinc FIR ------------------------------------------ */
/* Replaces soundkit::audio_pipeline::downsample_audio (audio_pipeline.rs:438-493) for the
 * 48000 -> 16000 case, where rubato's SincFixedIn<f32> (sinc_len 256, f_cutoff 0.95,
 * Linear, oversampling 256, BlackmanHarris2) has step exactly 3 and zero fractional phase:
 *   out[r][m] = sum_{p<256} h[p] * in[r][3m - 125 + p],  in[<0] = 0,
 *   m < sk_downsample_48k_16k_out_frames(frames) = ceil((frames - 132) / 3).
 * rows = independent channel signals (streams x channels), each `frames` long. */
uint32_t sk_downsample_48k_16k_out_frames(uint32_t frames);
int sk_downsample_48k_16k_taps(sk_engine *, float *taps256); /* the h[] the engine uses */
int sk_downsample_48k_16k_f32(sk_engine *, const float *in, uint32_t rows, uint32_t frames, float *out,
                              uint32_t *out_frames);
int sk_downsample_48k_16k_f32_dev(sk_engine *, const float *d_in, size_t in_stride, uint32_t rows, uint32_t frames,
                                  float *d_out, size_t out_stride, uint32_t *out_frames);

/* downsample_audio for any pair of the reference's COMMON_SAMPLE_RATES (audio_pipeline.rs:12-13): rubato
 * SincFixedIn<f32> with Linear interpolation between the two nearest of 256 sub-filters; one chunk = the
 * whole input, so out_frames = number of index steps below frames - 257 - ceil(in_hz/out_hz).
 * 48000 -> 16000 takes the MFMA path above; every other ratio the generic kernel (csrc/resample.hip).
 * out rows have capacity out_cap; *out_frames receives the frames produced per row. */
uint32_t sk_downsample_out_frames(uint32_t frames, uint32_t in_hz, uint32_t out_hz);
int sk_downsample_f32(sk_engine *, const float *in, uint32_t rows, uint32_t frames, uint32_t in_hz, uint32_t out_hz,
                      float *out, uint32_t out_cap, uint32_t *out_frames);
int sk_downsample_f32_dev(sk_engine *, const float *d_in, size_t in_stride, uint32_t rows, uint32_t frames,
                          uint32_t in_hz, uint32_t out_hz, float *d_out, size_t out_stride, uint32_t *out_frames);

/* downsample_audio over the frame-packed planar PCM that sk_aac_plan_run_f32_dev wrote, in place (no
 * repack): sample n of (stream s, channel c) is read at
 *   d_pcm[s * stream_stride + (n / 1024) * frame_stride + c * 1024 + n % 1024],   n < frames_per_stream * 1024
 * and row s * channels + c of d_out receives its sk_downsample_48k_16k_out_frames(..) outputs. */
int sk_downsample_48k_16k_frames_dev(sk_engine *, const float *d_pcm, size_t stream_stride, size_t frame_stride,
                                     uint32_t channels, uint32_t n_streams, uint32_t frames_per_stream, float *d_out,
                                     size_t out_stride, uint32_t *out_frames);

/* The same with the worker's 16-bit output stage (f32_channels_to_bytes, lib.rs:3619-3647: float_sample_to_i16) fused
 * into the FIR's epilogue: d_out[s][m][c] interleaved s16, out_stride frames per stream.  Bit-identical to
 * sk_downsample_48k_16k_frames_dev followed by sk_pcm_f32_planar_to_bytes_batch_dev(SK_FMT_S16LE). */
int sk_downsample_48k_16k_frames_s16_dev(sk_engine *, const float *d_pcm, size_t stream_stride, size_t frame_stride,
                                         uint32_t channels, uint32_t n_streams, uint32_t frames_per_stream,
                                         int16_t *d_out, size_t out_stride, uint32_t *out_frames);

/* apply_output_options' resample step on the worker's own data (lib.rs:3324-3456): audio_data_to_f32_channels
 * (s16 / 32768, lib.rs:3563-3617) -> 48k->16k sinc FIR -> f32_channels_to_bytes (float_sample_to_i16), over the planar s16
 * PCM of sk_aac_plan_run_s16_planar_dev, indexed like sk_downsample_48k_16k_frames_dev's input (strides in samples,
 * multiples of 4).  A 16-bit sample is exactly two bf16 values, so the matrix-core FIR needs two input planes and
 * 36 instead of 41 products per tile.  d_out[s][m][c] interleaved s16, out_stride frames per stream. */
int sk_downsample_48k_16k_frames_s16_to_s16_dev(sk_engine *, const int16_t *d_pcm16, size_t stream_stride, size_t frame_stride,
                                                uint32_t channels, uint32_t n_streams, uint32_t frames_per_stream,
                                                int16_t *d_out, size_t out_stride, uint32_t *out_frames);
/* the same filter output before the 16-bit stage: row s * channels + c of d_out (f32, out_stride per row) */
int sk_downsample_48k_16k_frames_s16_to_f32_dev(sk_engine *, const int16_t *d_pcm16, size_t stream_stride, size_t frame_stride,
                                                uint32_t channels, uint32_t n_streams, uint32_t frames_per_stream,
                                                float *d_out, size_t out_stride, uint32_t *out_frames);

/* In a library built with packed-f32 instructions (sk_kernels_use_packed_f32() == 1) this entry point is WITHDRAWN: it returns
 * SK_ERR_UNSUPPORTED for every plan -- use the two calls -- unless SK_AAC_TAIL_ONE_LAUNCH=1 is in the environment, which is for
 * reproducing the defect only: a kernel whose waves run the synthesis with packed-f32 instructions while others of them run
 * matrix instructions on the same SIMDs is what this platform computes wrongly (profiles/r04_lanes_corruption.md); at the
 * headline batch 3 % of its samples are wrong, differently in every run.  In the default build (no packed-f32 instructions) it
 * is exact at every size.
 * sk_aac_plan_run_s16_planar_dev + sk_downsample_48k_16k_frames_s16_to_s16_dev over all frames of a plan as ONE launch
 * (decode_aac_access_unit + apply_output_options, lib.rs:1793-1813, 3324-3456): the s16 PCM between the two never crosses
 * HBM -- each channel's wave keeps its last 2048 samples as the FIR's two f16 planes in LDS and runs the FIR on them as
 * they complete.  Same arithmetic in the same order as the two calls.  For plans whose channels are all free of
 * EightShort frames and pair up (two channels of equal length), every stream with `channels` channels and
 * `frames_per_stream` frames from its frame 0; anything else returns SK_ERR_UNSUPPORTED: use the two calls.
 * stream_stride: samples between the first frames of consecutive streams in the spectra's packing (as above);
 * d_out[s][m][c] interleaved s16, out_stride frames per stream (a multiple of 4), 8-byte aligned. */
int sk_aac_plan_run_tail_s16_dev(sk_engine *, const sk_aac_plan *, const float *d_coeffs, size_t stream_stride, uint32_t channels,
                                 uint32_t frames_per_stream, int16_t *d_out, size_t out_stride, uint32_t *out_frames);

/* StreamingResampler (soundkit-decoder lib.rs:1917-2060), any pair of COMMON_SAMPLE_RATES, fixed 4096-frame
 * chunks, history kept per stream on the device.  in: n_streams x channels x frames planar
 * (stream-major); out: capacity out_cap frames per channel row; out_frames[s] receives the
 * frames produced for stream s by this call (0 until 4096 input frames have accumulated). */
int sk_resampler_open(sk_engine *, uint32_t stream, uint32_t in_hz, uint32_t out_hz);
int sk_resampler_close(sk_engine *, uint32_t stream);
int sk_resampler_process_f32(sk_engine *, const uint32_t *streams, uint32_t n_streams, const float *in,
                             uint32_t frames, float *out, uint32_t out_cap, uint32_t *out_frames);
int sk_resampler_flush_f32(sk_engine *, const uint32_t *streams, uint32_t n_streams, float *out, uint32_t out_cap,
                           uint32_t *out_frames);

/* ---- one scheduler tick: a batch of access units through the worker's output stage ---------------- */
/* For every access unit of every listed stream: decode_aac_access_unit (soundkit-decoder lib.rs:1793-1813: synthesis,
 * then float_sample_to_i16) followed by apply_output_options (lib.rs:3324-3456: fast path, or i16 -> f32 ->
 * StreamingResampler -> downmix -> f32_channels_to_bytes), one launch sequence for the whole batch.  The outputs are
 * the AudioData values the reference's pipeline_worker would have sent (lib.rs:3238-3259), per stream in order:
 * one per access unit without resampling, one per completed 4096-frame chunk with it (lib.rs:1970-2003), plus
 * the flushed tail when `flush` is set (lib.rs:2017-2058, 3307-3322). */
typedef struct sk_tick_stream {
    uint32_t stream;      /* engine stream; its access units are contiguous in descs / coeffs, streams in this order */
    uint32_t n_frames;    /* access units of this stream in the batch (may be 0 with flush) */
    uint8_t out_bits;     /* DecodeOptions::output_bits_per_sample resolved: 16 / 24 / 32 (signed little-endian) */
    uint8_t out_channels; /* DecodeOptions::output_channels resolved (source channels when None) */
    uint8_t resample;     /* 1: route through the stream's resampler (sk_resampler_open) */
    uint8_t flush;        /* 1: end of stream: flush the resampler after these frames */
    uint8_t codec;        /* SK_TICK_AAC (0) | SK_TICK_MP3: an MP3 stream's units are GRANULES (576 PCM frames each), taken in order
                           * from sk_tick_input's mp3_* arrays; only sk_tick_run_mixed accepts them */
    uint8_t reserved[3];
} sk_tick_stream;
enum { SK_TICK_AAC = 0, SK_TICK_MP3 = 1 };

typedef struct sk_tick_output {
    uint32_t stream_index; /* index into the tick's stream table */
    uint32_t frames;       /* PCM frames in this AudioData */
    uint64_t byte_offset;  /* into out_bytes (16-byte aligned) */
    uint32_t bytes;        /* frames * channels * bits / 8 */
    int32_t status;        /* 0, or the sk_frame_status of an access unit the engine rejected: the stream ends there */
    uint8_t channels, bits;
    uint16_t reserved;
} sk_tick_output;

/* upper bounds for the two output arrays of a tick: from the table alone (a resampling stream is taken for the largest ratio there
 * is, 8 -> 48 kHz: 48 KB per chunk and output channel at 32 bits), or -- _on -- from what the engine knows of the streams (their
 * resamplers' own ratios): the scheduler sizes its pinned output buffers with the second (the first made them gigabytes) */
size_t sk_tick_out_bound(const sk_tick_stream *streams, uint32_t n_streams, uint32_t *max_outputs);
size_t sk_tick_out_bound_on(sk_engine *, const sk_tick_stream *streams, uint32_t n_streams, uint32_t *max_outputs);
/* coeffs: host memory (pinned for full overlap), packed like sk_aac_synthesize_f32's.  Blocks until out_bytes holds
 * the results.  *out_bytes_used receives the bytes written. */
int sk_tick_run(sk_engine *, const sk_tick_stream *streams, uint32_t n_streams, const sk_aac_frame_desc *descs,
                const float *coeffs, uint32_t n_frames, uint8_t *out_bytes, size_t out_cap, sk_tick_output *outputs,
                uint32_t outputs_cap, uint32_t *n_outputs, size_t *out_bytes_used);

/* One tick over streams of both codecs -- the worker's per-format dispatch (FormatDecoder::process,
 * soundkit-decoder/src/lib.rs:2222-2241) for a whole batch: the AAC streams' units in ONE of the three forms above (spectra,
 * access units, or quantised values: leave the others NULL), the MP3 streams' granules as sk_mp3_decode_granules_* takes them
 * (Huffman stage done on the host: sk_mp3_decode_main_data), listed stream by stream in the order of `streams`.  An MP3
 * granule becomes what Mp3Decoder hands the worker -- i16 through f32_to_i16 (soundkit-mp3/src/lib.rs:376-385) -- and then
 * takes the same apply_output_options path as an AAC unit (fast path, or / 32768 -> StreamingResampler -> downmix -> bytes).
 * Outputs: one AudioData per AAC access unit / MP3 granule without resampling, one per 4096-frame chunk with it.
 * The engine must hold the MP3 tables (sk_mp3_set_band_tables / sk_mp3_set_synthesis_window, or any sk_mp3_decoder_create). */
typedef struct sk_tick_input {
    const sk_aac_frame_desc *descs;          /* host front-end: descs + coeffs; quantised hand-over: descs + q_sides + q_quant */
    const float *coeffs;
    const struct sk_au_item *units;          /* GPU front-end: units + au_bytes */
    const uint8_t *au_bytes;
    size_t au_bytes_len;
    const void *q_sides;
    const int16_t *q_quant;
    uint32_t n_aac_units;
    uint32_t n_mp3_granules;
    const sk_mp3_requant_granule *mp3_granules;
    const sk_mp3_granule_desc *mp3_descs;
    const int16_t *mp3_is;                   /* [granule][channel][576], granules packed back to back */
} sk_tick_input;
int sk_tick_run_mixed(sk_engine *, const sk_tick_stream *streams, uint32_t n_streams, const sk_tick_input *in, uint8_t *out_bytes,
                      size_t out_cap, sk_tick_output *outputs, uint32_t outputs_cap, uint32_t *n_outputs, size_t *out_bytes_used);

/* The same tick with the entropy front-end on the GPU too (SURVEY 8f ranks 1 + 4): instead of spectra, the raw access
 * units (ADTS headers stripped) of every stream.  units[k] addresses unit k in au_bytes; units are listed stream by
 * stream in the order of `streams`, byte_offset is a multiple of 4 and every unit is followed by >= 8 zero bytes.
 * The streams must have been opened with the AudioSpecificConfig's sample rate and channel count.  A unit the
 * front-end rejects yields an output record with its sk_aac_status in `status` and ends that stream's tick, exactly
 * like a unit sk_aac_decoder_parse would have rejected on the host (same codes; the message text is host-only).
 * byte_len is at most 8192 (an ADTS frame cannot exceed 8191 bytes) and au_bytes_len at most 4 GiB - 1. */
typedef struct sk_au_item {
    uint32_t byte_offset;
    uint32_t byte_len;
} sk_au_item;
int sk_tick_run_au(sk_engine *, const sk_tick_stream *streams, uint32_t n_streams, const sk_au_item *units, uint32_t n_units,
                   const uint8_t *au_bytes, size_t au_bytes_len, uint8_t *out_bytes, size_t out_cap, sk_tick_output *outputs,
                   uint32_t outputs_cap, uint32_t *n_outputs, size_t *out_bytes_used);

/* The same tick fed by sk_aac_decoder_parse_q: per access unit the side record and the quantised i16 values (packed like
 * coeffs: unit k at the running channel count x 1024); descs as for sk_tick_run.  Dequantisation, PNS (the stream's
 * generator lives in the engine, as for sk_tick_run_au), stereo tools and TNS run on the device before the synthesis. */
int sk_tick_run_q(sk_engine *, const sk_tick_stream *streams, uint32_t n_streams, const sk_aac_frame_desc *descs,
                  const void *sides /*[n_units][SK_AAC_UNIT_SIDE_BYTES]*/, const int16_t *quant, uint32_t n_units, uint8_t *out_bytes,
                  size_t out_cap, sk_tick_output *outputs, uint32_t outputs_cap, uint32_t *n_outputs, size_t *out_bytes_used);
/* The device half of the quantised hand-over alone (what sk_aac_entropy_decode is for the full device front-end): the
 * records and integers of sk_aac_decoder_parse_q in, and back come the finished spectra [unit][channels][1024] (dequantised,
 * noise filled, intensity / mid-side, TNS: dsp.rs:397-405, spectral.rs:2408-2460, stereo.rs:114-448, tns.rs:103-276), the
 * window fields and one sk_aac_status per unit -- nothing is synthesised.  descs[k].stream names the unit's stream as
 * in sk_tick_run_q; the streams' PNS generators advance as in a tick. */
int sk_aac_expand_q_decode(sk_engine *, const uint32_t *streams, const uint32_t *units_per_stream, uint32_t n_streams,
                           const sk_aac_frame_desc *descs, const void *sides /*[n_units][SK_AAC_UNIT_SIDE_BYTES]*/, const int16_t *quant,
                           uint32_t n_units, float *coeffs_out, sk_aac_frame_desc *descs_out, int32_t *status_out);

/* The front-end alone, batched on the GPU: AacLcDecoder::decode_access_unit (soundkit-aac-lc/src/decoder.rs:104-164) up to
 * the hand-over to synthesis (decoder.rs:336) for every listed unit -- the device counterpart of sk_aac_decoder_parse.
 * Units are listed stream by stream (units_per_stream[i] of streams[i], in order), laid out as for sk_tick_run_au.
 * coeffs_out receives channels * 1024 f32 per unit, packed in unit order; descs_out[k] the stream, channel count and
 * the window_sequence / window_shape of unit k -- ready to be passed to sk_aac_synthesize_*; status_out[k] is 0 or the
 * unit's sk_aac_status (a failed unit has zero spectra; the units after it in its stream are not decoded: -199).
 * Advances each stream's PNS generator (spectral.rs:2416-2459) exactly as decoding those units does; the synthesis
 * state is untouched. */
int sk_aac_entropy_decode(sk_engine *, const uint32_t *streams, const uint32_t *units_per_stream, uint32_t n_streams,
                          const sk_au_item *units, uint32_t n_units, const uint8_t *au_bytes, size_t au_bytes_len,
                          float *coeffs_out, sk_aac_frame_desc *descs_out, int32_t *status_out);

/* ---- batch scheduler: N streams -> one submission loop per GPU ------------------------------------- */
/* Replaces one pipeline_worker thread per stream (soundkit-decoder lib.rs:2891-3038) for ADTS AAC-LC input and keeps
 * DecodePipelineHandle's contract (lib.rs:2788-2889): send never blocks and reports a full input queue (128 chunks /
 * 8 MiB), an empty chunk ends the stream (flush), at most 16 undelivered AudioData per stream (a stream whose
 * consumer does not drain it stops being scheduled: the blocking send of lib.rs:3238-3240), an error is delivered
 * after the outputs produced before it and ends that stream only (lib.rs:3131-3134).  Entropy decoding runs on
 * `entropy_threads` host threads, everything after it in sk_tick_run on the engine's GPU. */
typedef struct sk_pipeline sk_pipeline;
typedef struct sk_pipeline_config {
    uint32_t entropy_threads;            /* 0 = usable CPUs (affinity, cgroup quota) - 5, at most 64 */
    uint32_t max_streams;                /* handles open at once; 0 = 1024 (<= the engine's max_streams) */
    uint32_t max_frames_per_tick;        /* access units per GPU tick; 0 = 16384 (65536 with gpu_entropy) */
    uint32_t max_stream_frames_per_tick; /* of one stream; 0 = 8 (32 with gpu_entropy: this quota decides how full the ticks are) */
    uint32_t input_buffer;               /* chunks per input queue; 0 = DEFAULT_INPUT_BUFFER 128 (lib.rs:77) */
    uint32_t output_buffer;              /* AudioData per output queue; 0 = DEFAULT_OUTPUT_BUFFER 16 (lib.rs:78) */
    uint32_t tick_wait_us;               /* how long a non-empty batch may wait for more frames; 0 = 200 (2000 with gpu_entropy) */
    uint32_t gpu_entropy;                /* 1: the host threads only frame the ADTS stream; Huffman decode, stereo tools and TNS
                                          * run on the GPU too (sk_tick_run_au).  0 (default): host front-end (sk_tick_run).
                                          * 2: the host threads do the Huffman decode only and hand over i16 quantised values +
                                          * side records; dequantisation, PNS, stereo tools and TNS run on the GPU (sk_tick_run_q) */
    uint32_t lanes;                      /* engines the streams are spread over, each with its own batches and submission
                                          * thread, so that ticks overlap on the device; lane 0 is the caller's engine, the
                                          * others are created on the same device.  0 = 2 with gpu_entropy and max_streams >= two ticks' worth of
                                          * streams (4096 at the defaults), else 1; at most 8.  The engines of a device take turns with
                                          * their ticks' device work (ticks on the device at the same time were found to corrupt samples,
                                          * round 4): a lane plans, uploads and delivers while another's tick has the device.
                                          * entropy_threads and max_streams are totals, split over the lanes */
} sk_pipeline_config;

typedef struct sk_decode_options { /* DecodeOptions, lib.rs:147-151; 0 = None */
    uint32_t output_sample_rate;
    uint8_t output_bits_per_sample;
    uint8_t output_channels;
    uint16_t reserved;
} sk_decode_options;

typedef struct sk_audio_info { /* AudioData (soundkit/src/audio_types.rs:9-61) minus its bytes, or a DecodeError */
    uint32_t sampling_rate;
    uint32_t frames;
    uint32_t bytes;          /* PCM bytes (signed little-endian, interleaved), or the length of the error text */
    int32_t status;          /* error only: the sk_status / sk_aac_status behind DecodeError::DecodingFailed */
    uint8_t bits_per_sample, channel_count;
    uint8_t is_error;        /* 1: the data is the error's message and the stream has ended */
    uint8_t reserved;
} sk_audio_info;

typedef struct sk_pipeline_stats {
    uint64_t ticks, frames, outputs, errors;
    uint64_t parse_ns;  /* summed over entropy threads */
    uint64_t tick_ns;   /* submission thread inside sk_tick_run */
    uint64_t idle_ns;   /* submission thread waiting for a batch */
    uint64_t deliver_ns; /* delivery thread handing outputs to the streams' queues */
    uint32_t entropy_threads; /* summed over lanes */
    uint32_t lanes;
} sk_pipeline_stats;

enum sk_pipeline_status {
    SK_PIPE_INPUT_FULL = -201,      /* DecodeError::InputBufferFull */
    SK_PIPE_CLOSED = -202,          /* DecodeError::PipelineClosed; from recv: the stream has ended and is drained */
    SK_PIPE_CHUNK_TOO_LARGE = -203  /* DecodeError::InputChunkTooLarge (> 4 MiB) */
};

int sk_pipeline_create(sk_engine *, const sk_pipeline_config *cfg /* NULL = defaults */, sk_pipeline **out);
void sk_pipeline_destroy(sk_pipeline *);
/* DecodePipeline::spawn_with_options for an ADTS AAC-LC stream (lib.rs:2590-2700, 2750-2786) */
int sk_pipeline_spawn(sk_pipeline *, const sk_decode_options *opt /* NULL = defaults */, uint32_t *handle);
int sk_pipeline_send(sk_pipeline *, uint32_t handle, const uint8_t *data, size_t len);   /* lib.rs:2795-2835 */
int sk_pipeline_finish(sk_pipeline *, uint32_t handle);                                   /* lib.rs:2838-2840 */
/* 1 = one output copied to data / info; 0 = nothing ready; SK_PIPE_CLOSED = ended and drained; SK_ERR_CAPACITY =
 * cap is smaller than info->bytes (the output stays queued).  lib.rs:2845-2858 */
int sk_pipeline_try_recv(sk_pipeline *, uint32_t handle, uint8_t *data, size_t cap, sk_audio_info *info);
int sk_pipeline_recv(sk_pipeline *, uint32_t handle, uint8_t *data, size_t cap, sk_audio_info *info, uint32_t timeout_ms);
/* For a caller that serves many handles from few threads (the reference gives every stream its own thread and a
 * blocking recv): blocks up to timeout_ms for handles that have something to receive -- outputs, or the end of their
 * stream -- and returns how many were written to `handles`.  A handle is reported once per batch of news: drain it with
 * try_recv until 0 / SK_PIPE_CLOSED; a handle left with outputs is reported again. */
int sk_pipeline_wait_outputs(sk_pipeline *, uint32_t *handles, uint32_t cap, uint32_t timeout_ms);
int sk_pipeline_cancel(sk_pipeline *, uint32_t handle); /* cancel() / Drop, lib.rs:2860-2889: frees the handle */
size_t sk_pipeline_queued_input_bytes(sk_pipeline *, uint32_t handle); /* lib.rs:2863-2866 */
int sk_pipeline_get_stats(sk_pipeline *, sk_pipeline_stats *out);
/* Where every thread of the scheduler stands, the batches, the stream table in counts and the engine's stage, as text
 * (what SK_PIPELINE_WATCHDOG=<seconds> prints when ticks stop).  Takes no lock it could block on.  Returns the length. */
size_t sk_pipeline_debug_dump(sk_pipeline *, char *buf, size_t cap);

#ifdef __cplusplus
}
#endif
#endif /* SOUNDKIT_AMD_H */
