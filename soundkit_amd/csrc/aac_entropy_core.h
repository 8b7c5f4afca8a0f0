// aac_entropy_core.h -- the AAC-LC access-unit front-end as exception-free, allocation-free code that compiles for
// the host and for gfx950 from one source (SURVEY 8f ranks 1 + 4: entropy decode on the GPU, one stream per lane).
//
// Same syntax, same arithmetic and the same error *codes* as csrc/aac_frontend.cpp (which stays the host product path
// and carries the reference's error messages); restated in a shape a GPU lane can run: every function returns a
// status, every loop is bounded by a syntax constant (a damaged stream can cost time, never hang a wave), all state
// lives in fixed-size structures, and the tables come in through pointers so that the device build can place them
// in LDS / global memory.  tests/entropy_core_check.cpp builds it for the CPU under AddressSanitizer and proves it
// equal to aac_frontend.cpp (status, spectra bit for bit, window fields) on every fixture access unit and on
// hundreds of thousands of mutated ones before the device build is trusted with a GPU.
//
// Reference citations as in aac_frontend.cpp: bitreader.rs, syntax.rs, channel.rs, ics.rs, section.rs,
// scalefactor.rs, spectral.rs, pulse.rs, stereo.rs, tns.rs, sfb.rs, decoder.rs of soundkit-aac-lc/src.
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define SKE __host__ __device__ inline
#define SKE_NOINLINE __host__ __device__ __attribute__((noinline))
#else
#include <math.h>
#include <string.h>
#define SKE inline
#define SKE_NOINLINE inline
#endif

namespace sk_ec {

enum : int {
    EC_OK = 0,
    EC_EOF = -101,                  // SK_AAC_ERR_EOF
    EC_UNSUPPORTED_SF_INDEX = -104, // SK_AAC_ERR_UNSUPPORTED_SF_INDEX
    EC_UNSUPPORTED_FEATURE = -106,  // SK_AAC_ERR_UNSUPPORTED_FEATURE
    EC_INVALID_CONFIG = -107,       // SK_AAC_ERR_INVALID_CONFIG
    EC_INVALID_BITSTREAM = -108,    // SK_AAC_ERR_INVALID_BITSTREAM
};

#define EC_TRY(expr)            \
    do {                        \
        const int _st = (expr); \
        if (_st != EC_OK) return _st; \
    } while (0)

// Everything is reached through a handful of base pointers plus one small index block, so that a device build keeps the
// whole structure in scalar registers whatever book or sampling-frequency index a lane happens to use.
enum MetaIndex : int {  // offsets into Tables::meta (uint32 each)
    META_LUT_OFFSET = 0,         // [12] start of each two-level Huffman table in `lut` ([0] scalefactors, [1..11] spectral)
    META_TUPLE_OFFSET = 12,      // [12] start of each book's tuples in `tuples`
    META_SWB_LONG_OFFSET = 24,   // [13] start of each long band-offset table in `swb`, by sampling-frequency index
    META_SWB_SHORT_OFFSET = 37,  // [13]
    META_BANDS_LONG = 50,        // [13] bands in that table (the table has bands + 1 entries)
    META_BANDS_SHORT = 63,       // [13]
    META_TNS_MAX_LONG = 76,      // [13] tns.rs:284-285
    META_TNS_MAX_SHORT = 89,     // [13]
    META_WORDS = 102,
};
constexpr uint32_t kPrimaryBits = 9;  // of every two-level table (aac_frontend.cpp Lut::kPrimary; every book's longest code >= 9)

struct Tables {
    const uint32_t *meta;     // [META_WORDS]
    const uint32_t *lut;      // the twelve two-level Huffman tables as aac_frontend.cpp builds them, back to back
    const uint64_t *tuples;   // per spectral symbol: bytes 0-3 the values (int8), byte 4 sign-bit count, byte 5 escape flag
    const uint16_t *swb;      // every band-offset table back to back
    const float *pow43;       // [kPow43Len]: v^(4/3) from the host's powf for every magnitude the syntax can produce
    const float *sf_wide;     // [65536]: 2^((sf - 100) / 4) for every i16 scale factor, host powf (dsp.rs:407-413)
    const float *is_wide;     // [65536]: 2^(-position / 4) for every i16 intensity position, host powf
    const float *pow43_lo;    // its first kPow43Lo entries again (device: a copy in LDS; all but escape values end here)
    const float *sf_mult;     // [768]: scale factor -256..511
    const float *is_mult;     // [512]: 2^(-position / 4) for intensity positions -256..255 (scalefactor.rs:208-210)
    const float *tns_sin;     // [2][17]: sin(signed * pi / divisor) for coef_res 3 / 4 bits, signed -8..8 (tns.rs:208-235)
};

// ---- bit reader over 32-bit big-endian-packed words (the buffer is 4-byte aligned and zero-padded by >= 8 bytes) ----
constexpr uint32_t kPow43Lo = 256;  // 1 KiB of LDS: magnitudes beyond it (escape sequences of 4+ extra bits) are rare
// The largest quantised magnitude: an escape of 16 extra bits (spectral.rs:214-230) is 2^17 - 1, four pulses add at most
// 4 * 15 (pulse.rs:20-35).  The reference computes powf beyond its 8192-entry table; a device powf is not the host's,
// so every reachable value is tabulated (host libm, once) and the device never evaluates a transcendental.
constexpr uint32_t kPow43Len = (1u << 17) + 64;

struct Bits {
    const uint32_t *words;
    uint32_t total, pos;
    // the two words around `pos`, kept in registers: on a GPU every lane reads its own access unit, so a load is an
    // uncoalesced, dependent round trip -- one per 32 bits consumed instead of two per peek
    uint32_t cached;  // index of the first cached word (0xffffffff: nothing cached yet)
    uint64_t window;  // bswap(words[cached]) << 32 | bswap(words[cached + 1])
    // the word after the window, requested when the window moves and not needed before it moves again: the load's
    // latency overlaps the decoding of 32 bits instead of standing in front of it
    uint32_t ahead, last_word;  // last_word: last index inside the unit's 8 bytes of zero padding
};

SKE Bits make_bits(const uint32_t *words, uint32_t len_bytes) {
    return Bits{words, len_bytes * 8, 0, 0xffffffffu, 0, 0, (len_bytes + 8) / 4 - 1};
}

SKE uint32_t ec_bswap(uint32_t v) { return (v >> 24) | ((v >> 8) & 0xff00u) | ((v << 8) & 0xff0000u) | (v << 24); }

SKE uint32_t peek32(Bits &b) {  // next 32 bits, left-aligned; bits past the end read as the padding (zero)
    const uint32_t i = b.pos >> 5, s = b.pos & 31;
    if (i != b.cached) {
        if (b.cached != 0xffffffffu && i == b.cached + 1) b.window = (b.window << 32) | ec_bswap(b.ahead);
        else b.window = ((uint64_t)ec_bswap(b.words[i]) << 32) | ec_bswap(b.words[i + 1]);
        b.cached = i;
        b.ahead = b.words[i + 2 < b.last_word ? i + 2 : b.last_word];
    }
    return (uint32_t)((b.window << s) >> 32);
}

SKE int read_bits(Bits &b, uint32_t n, uint32_t *out) {  // n <= 32
    if (b.total - b.pos < n) return EC_EOF;
    *out = n ? peek32(b) >> (32 - n) : 0;
    b.pos += n;
    return EC_OK;
}

SKE int read_flag(Bits &b, bool *out) {
    uint32_t v;
    EC_TRY(read_bits(b, 1, &v));
    *out = v != 0;
    return EC_OK;
}

SKE int huffman_in(const uint32_t *lut, Bits &b, uint32_t *symbol) {  // scalefactor.rs:252-266, spectral tuple readers
    const uint32_t look = peek32(b);
    uint32_t e = lut[look >> (32 - kPrimaryBits)];
    if (e & 0x80000000u) {
        const uint32_t extra = (e >> 24) & 0x7f;
        e = lut[(e & 0xffffffu) + ((look << kPrimaryBits) >> (32 - extra))];
    }
    const uint32_t len = e >> 16;
    if (len == 0 || len > b.total - b.pos) return EC_INVALID_BITSTREAM;
    b.pos += len;
    *symbol = e & 0xffffu;
    return EC_OK;
}

SKE int huffman(const Tables &t, int book, Bits &b, uint32_t *symbol) {
    return huffman_in(t.lut + t.meta[META_LUT_OFFSET + book], b, symbol);
}

// ---- side information -----------------------------------------------------------------------------------
enum { SEQ_ONLY_LONG = 0, SEQ_LONG_START = 1, SEQ_EIGHT_SHORT = 2, SEQ_LONG_STOP = 3 };
enum { BOOK_ZERO = 0, BOOK_NOISE = 13, BOOK_INTENSITY = 14, BOOK_INTENSITY_NEG = 15 };

struct Ics {  // ics.rs:46-54
    uint8_t sequence, shape, max_sfb, num_windows, num_groups;
    uint8_t group_len[8];
};

struct TnsFilter {
    uint8_t length, order, direction, coef_bits;
    int8_t coef[20];
};
struct TnsWindow {
    uint8_t filter_count, coef_res;
    TnsFilter filter[4];
};

struct Channel {  // IndividualChannelStream, channel.rs:36-75
    Ics ics;
    uint8_t global_gain;
    uint8_t pulse_present, pulse_start, pulse_count, pulse_offset[4], pulse_amp[4];
    uint8_t tns_present;
    uint8_t book[128];  // [group * stride + sfb], stride 16 for eight-short (max_sfb <= 15), 64 for long (one group)
    float mult[128];
    TnsWindow tns[8];
};

SKE int band_stride(const Ics &ics) { return ics.sequence == SEQ_EIGHT_SHORT ? 16 : 64; }

SKE int read_ics(Bits &b, Ics &ics) {  // ics.rs:57-110
    bool reserved;
    EC_TRY(read_flag(b, &reserved));
    if (reserved) return EC_INVALID_CONFIG;
    uint32_t v;
    EC_TRY(read_bits(b, 2, &v));
    ics.sequence = (uint8_t)v;
    EC_TRY(read_bits(b, 1, &v));
    ics.shape = (uint8_t)v;
    for (int i = 0; i < 8; ++i) ics.group_len[i] = 0;
    ics.group_len[0] = 1;
    if (ics.sequence == SEQ_EIGHT_SHORT) {
        EC_TRY(read_bits(b, 4, &v));
        ics.max_sfb = (uint8_t)v;
        uint32_t grouping;
        EC_TRY(read_bits(b, 7, &grouping));
        int group = 0;
        for (int bit = 0; bit < 7; ++bit) {
            if ((grouping >> (6 - bit)) & 1) ics.group_len[group] += 1;
            else ics.group_len[++group] = 1;
        }
        ics.num_windows = 8;
        ics.num_groups = (uint8_t)(group + 1);
    } else {
        EC_TRY(read_bits(b, 6, &v));
        ics.max_sfb = (uint8_t)v;
        bool prediction;
        EC_TRY(read_flag(b, &prediction));
        if (prediction) return EC_UNSUPPORTED_FEATURE;
        ics.num_windows = 1;
        ics.num_groups = 1;
    }
    return EC_OK;
}

SKE int read_sections(Bits &b, Channel &ch) {  // section.rs:60-120
    const Ics &ics = ch.ics;
    const uint32_t width = ics.sequence == SEQ_EIGHT_SHORT ? 3 : 5, escape = (1u << width) - 1;
    const int stride = band_stride(ics);
    for (int i = 0; i < 128; ++i) ch.book[i] = 0;
    for (int g = 0; g < ics.num_groups; ++g) {
        int sfb = 0;
        while (sfb < ics.max_sfb) {  // every pass consumes >= 1 band or fails: at most 63 passes
            uint32_t book;
            EC_TRY(read_bits(b, 4, &book));
            if (book == 12) return EC_INVALID_BITSTREAM;
            int len = 0;
            for (;;) {  // bounded: len grows by `escape` per pass and is checked against max_sfb below... so cap it here
                uint32_t incr;
                EC_TRY(read_bits(b, width, &incr));
                len += (int)incr;
                if (incr != escape) break;
                if (len > 64 * 31) break;  // cannot happen before EOF (a 6144-byte access unit holds < 10000 escapes); belt and braces
            }
            if (len == 0) return EC_INVALID_BITSTREAM;
            if (sfb + len > ics.max_sfb) return EC_INVALID_BITSTREAM;
            for (int k = sfb; k < sfb + len; ++k) ch.book[g * stride + k] = (uint8_t)book;
            sfb += len;
        }
    }
    return EC_OK;
}

SKE bool i16_add(int a, int b, int *out) {
    const int s = a + b;
    *out = s;
    return s >= -32768 && s <= 32767;
}

#if defined(__HIP_DEVICE_COMPILE__)
SKE float ec_powf(float a, float b) { return __builtin_powf(a, b); }
SKE float ec_sinf(float a) { return __builtin_sinf(a); }
SKE float ec_sqrtf(float a) { return __builtin_sqrtf(a); }
#else
SKE float ec_powf(float a, float b) { return powf(a, b); }
SKE float ec_sinf(float a) { return sinf(a); }
SKE float ec_sqrtf(float a) { return sqrtf(a); }
#endif

SKE float sf_multiplier(const Tables &t, int sf) {  // dsp.rs:407-413; sf is an i16 (checked_add above)
    if (sf >= -256 && sf <= 511) return t.sf_mult[sf + 256];
    return t.sf_wide[sf + 32768];
}

// sf_out (optional, [128]): the transmitted values themselves -- spectral scale factor, noise energy or intensity
// position of each band -- for the quantised hand-over to the device (WireChannel below)
SKE int read_scalefactors(const Tables &t, Bits &b, Channel &ch, int16_t *sf_out = nullptr) {  // scalefactor.rs:80-153
    int spectral = ch.global_gain, noise = ch.global_gain - 90, intensity = 0;
    bool first_noise = true;
    const int stride = band_stride(ch.ics);
    for (int i = 0; i < 128; ++i) ch.mult[i] = 0.0f;
    if (sf_out)
        for (int i = 0; i < 128; ++i) sf_out[i] = 0;
    for (int g = 0; g < ch.ics.num_groups; ++g)
        for (int sfb = 0; sfb < ch.ics.max_sfb; ++sfb) {
            const int book = ch.book[g * stride + sfb];
            if (book == BOOK_ZERO) continue;
            uint32_t sym;
            if (book == BOOK_NOISE) {
                if (first_noise) {
                    uint32_t raw;
                    EC_TRY(read_bits(b, 9, &raw));
                    if (!i16_add(noise, (int)raw - 256, &noise)) return EC_INVALID_BITSTREAM;
                    first_noise = false;
                } else {
                    EC_TRY(huffman(t, 0, b, &sym));
                    if (!i16_add(noise, (int)sym - 60, &noise)) return EC_INVALID_BITSTREAM;
                }
                ch.mult[g * stride + sfb] = sf_multiplier(t, noise);
                if (sf_out) sf_out[g * stride + sfb] = (int16_t)noise;
            } else if (book == BOOK_INTENSITY || book == BOOK_INTENSITY_NEG) {
                EC_TRY(huffman(t, 0, b, &sym));
                if (!i16_add(intensity, (int)sym - 60, &intensity)) return EC_INVALID_BITSTREAM;
                // scalefactor.rs:208-210; tabulated with the host's powf so that the device build agrees to the bit
                ch.mult[g * stride + sfb] = (intensity >= -256 && intensity <= 255) ? t.is_mult[intensity + 256] : t.is_wide[intensity + 32768];
                if (sf_out) sf_out[g * stride + sfb] = (int16_t)intensity;
            } else {
                EC_TRY(huffman(t, 0, b, &sym));
                if (!i16_add(spectral, (int)sym - 60, &spectral)) return EC_INVALID_BITSTREAM;
                ch.mult[g * stride + sfb] = sf_multiplier(t, spectral);
                if (sf_out) sf_out[g * stride + sfb] = (int16_t)spectral;
            }
        }
    return EC_OK;
}

SKE int read_tns(Bits &b, Channel &ch) {  // tns.rs:34-83
    const bool is_short = ch.ics.sequence == SEQ_EIGHT_SHORT;
    const uint32_t n_bits = is_short ? 1 : 2, len_bits = is_short ? 4 : 6, order_bits = is_short ? 3 : 5;
    for (int w = 0; w < ch.ics.num_windows; ++w) {
        TnsWindow &win = ch.tns[w];
        uint32_t v;
        EC_TRY(read_bits(b, n_bits, &v));
        win.filter_count = (uint8_t)v;  // <= 3
        win.coef_res = 0;
        if (win.filter_count == 0) continue;
        bool res;
        EC_TRY(read_flag(b, &res));
        win.coef_res = res;
        for (int f = 0; f < win.filter_count; ++f) {
            TnsFilter &flt = win.filter[f];
            EC_TRY(read_bits(b, len_bits, &v));
            flt.length = (uint8_t)v;
            EC_TRY(read_bits(b, order_bits, &v));
            flt.order = (uint8_t)v;
            flt.direction = 0;
            flt.coef_bits = 0;
            if (flt.order > 20) return EC_UNSUPPORTED_FEATURE;
            if (flt.order == 0) continue;
            bool dir, compress;
            EC_TRY(read_flag(b, &dir));
            EC_TRY(read_flag(b, &compress));
            flt.direction = dir;
            flt.coef_bits = (uint8_t)((win.coef_res ? 4 : 3) - (compress ? 1 : 0));
            for (int i = 0; i < flt.order; ++i) {  // read_signed, tns.rs:278-282
                EC_TRY(read_bits(b, flt.coef_bits, &v));
                const int shift = 8 - flt.coef_bits;
                flt.coef[i] = (int8_t)((int8_t)(uint8_t)(v << shift) >> shift);
            }
        }
    }
    return EC_OK;
}

SKE int read_channel(const Tables &t, Bits &b, Channel &ch, const Ics *common, int16_t *sf_out = nullptr, int mark = -100) {  // channel.rs:19-75
    uint32_t v;
    EC_TRY(read_bits(b, 8, &v));
    ch.global_gain = (uint8_t)v;
    if (common) ch.ics = *common;
    else EC_TRY(read_ics(b, ch.ics));
    EC_TRY(read_sections(b, ch));
    EC_TRY(read_scalefactors(t, b, ch, sf_out));
    bool flag;
    EC_TRY(read_flag(b, &flag));
    ch.pulse_present = flag;
    if (flag) {  // pulse.rs:20-35
        EC_TRY(read_bits(b, 2, &v));
        ch.pulse_count = (uint8_t)(v + 1);
        EC_TRY(read_bits(b, 6, &v));
        ch.pulse_start = (uint8_t)v;
        for (int i = 0; i < ch.pulse_count; ++i) {
            EC_TRY(read_bits(b, 5, &v));
            ch.pulse_offset[i] = (uint8_t)v;
            EC_TRY(read_bits(b, 4, &v));
            ch.pulse_amp[i] = (uint8_t)v;
        }
    }
    EC_TRY(read_flag(b, &flag));
    ch.tns_present = flag;
    if (flag) EC_TRY(read_tns(b, ch));
    EC_TRY(read_flag(b, &flag));
    if (flag) return EC_UNSUPPORTED_FEATURE;  // gain control
    return EC_OK;
}

// ---- spectral data -------------------------------------------------------------------------------------------
SKE float dequantize(const Tables &t, int q, float scale) {  // dsp.rs:397-405
    if (q == 0) return 0.0f;
    const float sign = q < 0 ? -1.0f : 1.0f;
    const uint32_t mag = q < 0 ? (uint32_t)(-(int64_t)q) : (uint32_t)q;
    const float m = mag < kPow43Lo ? t.pow43_lo[mag] : (mag < kPow43Len ? t.pow43[mag] : ec_powf((float)mag, 4.0f / 3.0f));
    return sign * m * scale;
}

SKE uint32_t ec_leading_ones(uint32_t v) {  // of a 32-bit word
    const uint32_t inv = ~v;
    if (inv == 0) return 32;
#if defined(__HIP_DEVICE_COMPILE__) || defined(__GNUC__)
    return (uint32_t)__builtin_clz(inv);
#else
    uint32_t n = 0;
    while (!((inv << n) & 0x80000000u)) ++n;
    return n;
#endif
}

// An escape sequence is N ones, a zero and N + 4 value bits (spectral.rs:214-230; N <= 12, beyond that the reference gives up):
// at most 29 bits, so ONE look at the bit window replaces the reference's bit-by-bit loop -- on the device every lane of a
// wave waits for the lane that is in that loop.  Same results and the same errors in the same order: the loop fails with
// UnexpectedEof at the first bit that is not there, with UnsupportedFeature at the 13th one.
SKE int read_escape(Bits &b, int *value) {
    const uint32_t avail = b.total - b.pos;
    const uint32_t look = peek32(b);             // bits past the end read as zero: they end the count below
    const uint32_t ones = ec_leading_ones(look);
    if (ones >= 13) return EC_UNSUPPORTED_FEATURE;  // thirteen real ones (padding is zero)
    if (avail < ones + 1) return EC_EOF;            // the terminating zero is padding: the loop ran out of bits
    const uint32_t extra = 4 + ones;
    if (avail - (ones + 1) < extra) return EC_EOF;  // read_bits(extra)
    const uint32_t low = (look << (ones + 1)) >> (32 - extra);
    b.pos += ones + 1 + extra;
    *value = (int)((1u << extra) + low);
    return EC_OK;
}

// one codeword of spectral book `book` -> its 4 (books 1-4) or 2 quantised values (spectral.rs:117-212)
struct BookRef {  // a band's codebook, resolved once per band (the index block and both tables may sit in LDS)
    const uint32_t *lut;
    const uint64_t *tuples;
    int book;
};
SKE BookRef book_ref(const Tables &t, int book) {
    return BookRef{t.lut + t.meta[META_LUT_OFFSET + book], t.tuples + t.meta[META_TUPLE_OFFSET + book], book};
}

SKE int read_tuple(const BookRef &br, Bits &b, int *q) {
    const int book = br.book;
    // ONE look at the bit window serves the codeword (<= 16 bits in every spectral book) and the sign bits behind it (<= 4):
    // on the device every look costs the refill test of the whole wave
    const uint32_t look = peek32(b);
    uint32_t e = br.lut[look >> (32 - kPrimaryBits)];
    if (e & 0x80000000u) {
        const uint32_t extra = (e >> 24) & 0x7f;
        e = br.lut[(e & 0xffffffu) + ((look << kPrimaryBits) >> (32 - extra))];
    }
    const uint32_t len = e >> 16;
    if (len == 0 || len > b.total - b.pos) return EC_INVALID_BITSTREAM;  // huffman_in
    b.pos += len;
    const uint64_t tu = br.tuples[e & 0xffffu];  // bytes 0-3 values, byte 4 sign-bit count, byte 5 escape flag, bits 48.. sign positions
    const int dim = book <= 4 ? 4 : 2;
    for (int k = 0; k < dim; ++k) q[k] = (int8_t)(tu >> (8 * k));
    const uint32_t nsign = (uint32_t)(tu >> 32) & 0xffu;  // 0 for the signed books
    if (nsign == 0) return EC_OK;
    // the sign bits of the non-zero magnitudes follow the codeword, in order: taken in one read and handed out by
    // position (no per-value branch; lanes of a wave sit in different tuples)
    if (b.total - b.pos < nsign) return EC_EOF;  // read_bits
    const uint32_t signs = (look << len) >> (32 - nsign);
    b.pos += nsign;
    for (int k = 0; k < dim; ++k) {
        const uint32_t at = (uint32_t)(tu >> (48 + 4 * k)) & 15u;  // 15: this value has no sign bit (signs < 16 >> 15 == 0)
        const int neg = (int)((signs >> at) & 1u);
        q[k] = (q[k] ^ -neg) + neg;
    }
    if ((tu >> 40) & 1u) {  // book 11 escapes follow both sign bits (finish_unsigned_escape_pair, spectral.rs:191-212)
        for (int k = 0; k < 2; ++k) {
            const int mag = q[k] < 0 ? -q[k] : q[k];
            if (mag == 16) {
                int esc;
                EC_TRY(read_escape(b, &esc));
                q[k] = q[k] < 0 ? -esc : esc;
            }
        }
    }
    return EC_OK;
}

SKE int band_range(const uint16_t *off, int bands, int sfb, int *s, int *e) {  // spectral.rs:92-105
    if (sfb < 0 || sfb + 1 > bands) return EC_INVALID_CONFIG;
    *s = off[sfb];
    *e = off[sfb + 1];
    return EC_OK;
}

struct alignas(16) NoiseQuad {
    float v[4];
};
SKE int noise_band(float scale, uint32_t &state, float *__restrict__ out, int n) {  // spectral.rs:2416-2450
    if (n == 0) return EC_OK;
    // the reference writes the raw noise, sums its energy, then scales in place; here the generator runs twice from the
    // same state (energy first, then the scaled values) so that nothing is read back from memory: same values
    uint32_t probe = state;
    float energy = 0.0f;
    for (int i = 0; i < n; ++i) {
        probe = probe * 1664525u + 1013904223u;
        const float v = (float)(int16_t)((int32_t)probe >> 16);
        energy += v * v;
    }
    if (energy <= 1.1920929e-07f) {
        state = probe;
        return EC_INVALID_BITSTREAM;
    }
    const float normalizer = scale / ec_sqrtf(energy);
    int i = 0;
    if ((((uintptr_t)out) & 15u) == 0) {  // bands start on multiples of four lines of a 4 KiB-aligned spectrum: 16-byte stores
        for (; i + 4 <= n; i += 4) {
            NoiseQuad q4;
            for (int k = 0; k < 4; ++k) {
                state = state * 1664525u + 1013904223u;
                q4.v[k] = (float)(int16_t)((int32_t)state >> 16) * normalizer;
            }
            memcpy((NoiseQuad *)__builtin_assume_aligned(out + i, 16), &q4, 16);
        }
    }
    for (; i < n; ++i) {
        state = state * 1664525u + 1013904223u;
        out[i] = (float)(int16_t)((int32_t)state >> 16) * normalizer;
    }
    return EC_OK;
}

struct Stream {  // what persists across the access units of one stream
    int sf_index;  // -1: explicit sample rate (no band tables)
    int channels;
    uint32_t pns_state;  // spectral.rs:2459, decoder.rs:76: starts at 0x1f2e3d4c
    // device: the spectra of this launch's units were zeroed by the whole wave (coalesced) before the lanes started, so
    // the decode does not have to clear them one scattered word per lane at a time
    bool prefilled = false;
};

SKE int layout(const Tables &t, const Stream &st, const Ics &ics, const uint16_t **off, int *bands) {  // decoder.rs:376-383, sfb.rs:52-71
    if (st.sf_index < 0) return EC_UNSUPPORTED_FEATURE;
    if (st.sf_index > 12) return EC_UNSUPPORTED_SF_INDEX;
    if (ics.sequence == SEQ_EIGHT_SHORT) {
        *off = t.swb + t.meta[META_SWB_SHORT_OFFSET + st.sf_index];
        *bands = (int)t.meta[META_BANDS_SHORT + st.sf_index];
    } else {
        *off = t.swb + t.meta[META_SWB_LONG_OFFSET + st.sf_index];
        *bands = (int)t.meta[META_BANDS_LONG + st.sf_index];
    }
    return EC_OK;
}

// decode_channel_spectrum (decoder.rs:220-244) + decode_standard_with_pulse_and_pns (spectral.rs:1907-2294).
// `coef` is written, never read, and cannot overlap the side information: it is declared restrict and everything the
// inner loops need is copied into locals first, so that on the device (where `ch` lives in private memory and `coef` is a
// flat pointer that could in principle alias it) a coefficient store does not force the side information to be re-read.
enum PnsMode {
    PNS_GENERATE,  // as the reference: noise bands are synthesised where they occur
    PNS_COUNT,     // frame-parallel decode: only count the samples (the generator state is not known yet)
};

// QUANT: instead of the dequantised spectrum, the quantised values themselves (pulses applied) go to `quant` as i16 --
// what the host front-end hands to the device when dequantisation, noise, stereo tools and TNS run there (SURVEY 8f rank
// 1).  The reference accepts escape sequences of up to 16 extra bits (spectral.rs:214-228: magnitudes to 131071; ISO/IEC
// 14496-3 allows 8191), which an i16 cannot hold: a value beyond +-32767 leaves the marker -32768 in `quant` and goes,
// with its position, into the unit's short list of wide values (kWideMax per access unit; one more is reported as an
// unsupported feature -- no encoder produces even one).  mode must be PNS_COUNT; coef is not touched.
constexpr int kWideMax = 48;
constexpr int16_t kWideMarker = -32768;
struct WideList {
    uint32_t n;
    uint16_t pos[kWideMax];  // channel * 1024 + line
    int32_t val[kWideMax];
};
SKE int16_t quant_store(int q, int pos, WideList *wide, int *status) {
    if (q <= 32767 && q >= -32767) return (int16_t)q;
    if (!wide || wide->n >= (uint32_t)kWideMax) {
        *status = EC_UNSUPPORTED_FEATURE;
        return 0;
    }
    wide->pos[wide->n] = (uint16_t)pos;
    wide->val[wide->n] = q;
    wide->n += 1;
    return kWideMarker;
}
template <bool QUANT>
SKE int decode_spectrum_nested(const Tables &t, Stream &st, Bits &bits, const Channel &ch, bool allow_intensity, float *__restrict__ coef,
                               PnsMode mode, uint32_t *noise_samples, int16_t *__restrict__ quant, WideList *wide = nullptr, int wide_base = 0) {
    const Ics ics = ch.ics;
    const int stride = band_stride(ics);
    const int max_sfb = ics.max_sfb, num_groups = ics.num_groups;
    if (!allow_intensity)
        for (int g = 0; g < num_groups; ++g)
            for (int sfb = 0; sfb < max_sfb; ++sfb) {
                const int book = ch.book[g * stride + sfb];
                if (book == BOOK_INTENSITY || book == BOOK_INTENSITY_NEG) return EC_INVALID_BITSTREAM;
            }
    const uint16_t *off;
    int bands;
    EC_TRY(layout(t, st, ics, &off, &bands));
    if (QUANT) {
        for (int i = 0; i < 1024; ++i) quant[i] = 0;
    } else if (!st.prefilled) {
        for (int i = 0; i < 1024; ++i) coef[i] = 0.0f;
    }
    Bits b = bits;  // a local copy: its fields stay in registers across the stores below
    uint32_t pns = st.pns_state;
    int q[4];
    int status = EC_OK;
    if (ics.sequence == SEQ_EIGHT_SHORT) {
        if (ch.pulse_present) return EC_INVALID_BITSTREAM;
        int w0 = 0;
        for (int g = 0; g < num_groups && status == EC_OK; ++g) {
            const int glen = ics.group_len[g];
            if (glen == 0 || w0 + glen > 8) {
                status = EC_INVALID_BITSTREAM;
                break;
            }
            for (int sfb = 0; sfb < max_sfb && status == EC_OK; ++sfb) {
                int s, e;
                status = band_range(off, bands, sfb, &s, &e);
                if (status != EC_OK) break;
                if (e > 128) {
                    status = EC_INVALID_CONFIG;
                    break;
                }
                const int book = ch.book[g * stride + sfb];
                const float scale = ch.mult[g * stride + sfb];
                if (book >= 1 && book <= 11) {
                    const BookRef br = book_ref(t, book);
                    const int dim = book <= 4 ? 4 : 2;
                    for (int w = w0; w < w0 + glen && status == EC_OK; ++w)
                        for (int i = s; i + dim <= e; i += dim) {
                            status = read_tuple(br, b, q);
                            if (status != EC_OK) break;
                            for (int k = 0; k < dim; ++k) {
                                if (QUANT) {
                                    quant[w * 128 + i + k] = quant_store(q[k], wide_base + w * 128 + i + k, wide, &status);
                                } else {
                                    coef[w * 128 + i + k] = dequantize(t, q[k], scale);
                                }
                            }
                            if (QUANT && status != EC_OK) break;
                        }
                } else if (book == BOOK_NOISE) {
                    if (mode == PNS_COUNT) *noise_samples += (uint32_t)(glen * (e - s));
                    else
                        for (int w = w0; w < w0 + glen && status == EC_OK; ++w) status = noise_band(scale, pns, coef + w * 128 + s, e - s);
                }
            }
            w0 += glen;
        }
        if (status == EC_OK && w0 != 8) status = EC_INVALID_BITSTREAM;
        bits = b;
        st.pns_state = pns;
        return status;
    }
    // long windows.  With pulse data the reference reads every band, then validates and adds the pulses, then
    // dequantises; the pulses only touch <= 4 coefficients, so they are tracked by position instead of keeping all
    // 1024 quantised values: targets are known before the spectral data (offsets from the start band), the value
    // read at a target is adjusted before it is dequantised, and the validity checks run after the last band so that a
    // damaged spectral codeword is reported first, as the reference does.
    const bool pulse_present = ch.pulse_present != 0;
    const int pulse_count = pulse_present ? ch.pulse_count : 0;
    int target[4] = {-1, -1, -1, -1}, amp[4] = {0, 0, 0, 0};
    bool start_known = false;
    if (pulse_present && ch.pulse_start < max_sfb && ch.pulse_start + 1 <= bands) {
        int index = off[ch.pulse_start];
        start_known = true;
        for (int i = 0; i < pulse_count; ++i) {
            index += ch.pulse_offset[i];
            target[i] = index;  // may be >= 1024 or outside the coded bands: checked after the spectral data
            amp[i] = ch.pulse_amp[i];
        }
    }
    for (int sfb = 0; sfb < max_sfb && status == EC_OK; ++sfb) {
        int s, e;
        status = band_range(off, bands, sfb, &s, &e);
        if (status != EC_OK) break;
        if (e > 1024) {
            status = EC_INVALID_CONFIG;
            break;
        }
        const int book = ch.book[sfb];
        const float scale = ch.mult[sfb];
        if (book >= 1 && book <= 11) {
            const BookRef br = book_ref(t, book);
            const int dim = book <= 4 ? 4 : 2;
            for (int i = s; i + dim <= e; i += dim) {
                status = read_tuple(br, b, q);
                if (status != EC_OK) break;
                for (int p = 0; p < pulse_count; ++p) {  // pulses apply in order; two may hit the same coefficient
                    const int k = target[p] - i;
                    if (k >= 0 && k < dim) q[k] += q[k] > 0 ? amp[p] : -amp[p];
                }
                for (int k = 0; k < dim; ++k) {
                    if (QUANT) {
                        quant[i + k] = quant_store(q[k], wide_base + i + k, wide, &status);
                    } else {
                        coef[i + k] = dequantize(t, q[k], scale);
                    }
                }
                if (QUANT && status != EC_OK) break;
            }
        } else if (book == BOOK_NOISE && !pulse_present) {
            if (mode == PNS_COUNT) *noise_samples += (uint32_t)(e - s);
            else status = noise_band(scale, pns, coef + s, e - s);
        }
    }
    bits = b;
    st.pns_state = pns;
    if (status != EC_OK || !pulse_present) return status;
    // apply_pulse_data's checks (spectral.rs:2198-2247), in its order
    if (ch.pulse_start >= max_sfb) return EC_INVALID_BITSTREAM;
    if (!start_known) return EC_INVALID_CONFIG;  // band_range(start_sfb) failed
    for (int p = 0; p < pulse_count; ++p) {
        const int index = target[p];
        if (index >= 1024) return EC_INVALID_BITSTREAM;
        int band = -1;
        for (int sfb = 0; sfb < max_sfb; ++sfb) {
            int s, e;
            EC_TRY(band_range(off, bands, sfb, &s, &e));
            if (index >= s && index < e) {
                band = sfb;
                break;
            }
        }
        if (band < 0) return EC_INVALID_BITSTREAM;
        const int book = ch.book[band];
        if (!(book >= 1 && book <= 11)) return EC_INVALID_BITSTREAM;
    }
    for (int sfb = 0; sfb < max_sfb; ++sfb) {  // the noise bands of the pulse path come last
        if (ch.book[sfb] != BOOK_NOISE) continue;
        int s, e;
        EC_TRY(band_range(off, bands, sfb, &s, &e));
        if (mode == PNS_COUNT) *noise_samples += (uint32_t)(e - s);
        else EC_TRY(noise_band(ch.mult[sfb], st.pns_state, coef + s, e - s));
    }
    return EC_OK;
}

// The same function as ONE loop that takes one step per pass -- open a window group, open a band, or decode one codeword
// -- instead of four nested loops.  Written for a wave of 64 lanes that each decode their own access unit: nested, the
// wave runs every band for as many passes as its slowest lane needs (a lane in a four-value book next to one in a
// two-value book idles half the time, one in a zero band idles for the whole band, and long and short windows take turns),
// flat, every lane that still has codewords left decodes one per pass whatever band, book or window it is in.  Per lane
// the operations, their order and therefore every result and every error code are those of the nested form
// (tests/entropy_core_check.cpp runs both on every unit it sees; SK_EC_FLAT selects this form in a host build).
template <bool QUANT>
SKE int decode_spectrum_flat(const Tables &t, Stream &st, Bits &bits, const Channel &ch, bool allow_intensity, float *__restrict__ coef,
                             PnsMode mode, uint32_t *noise_samples, int16_t *__restrict__ quant, WideList *wide = nullptr, int wide_base = 0) {
    const Ics ics = ch.ics;
    const int stride = band_stride(ics);
    const int max_sfb = ics.max_sfb, num_groups = ics.num_groups;
    if (!allow_intensity)
        for (int g = 0; g < num_groups; ++g)
            for (int sfb = 0; sfb < max_sfb; ++sfb) {
                const int book = ch.book[g * stride + sfb];
                if (book == BOOK_INTENSITY || book == BOOK_INTENSITY_NEG) return EC_INVALID_BITSTREAM;
            }
    const uint16_t *off;
    int bands;
    EC_TRY(layout(t, st, ics, &off, &bands));
    if (QUANT) {
        for (int i = 0; i < 1024; ++i) quant[i] = 0;
    } else if (!st.prefilled) {
        for (int i = 0; i < 1024; ++i) coef[i] = 0.0f;
    }
    const bool is_short = ics.sequence == SEQ_EIGHT_SHORT;
    if (is_short && ch.pulse_present) return EC_INVALID_BITSTREAM;
    const bool pulse_present = ch.pulse_present != 0;  // long windows only from here on
    const int pulse_count = pulse_present ? ch.pulse_count : 0;
    int target[4] = {-1, -1, -1, -1}, amp[4] = {0, 0, 0, 0};
    bool start_known = false;
    if (pulse_present && ch.pulse_start < max_sfb && ch.pulse_start + 1 <= bands) {
        int index = off[ch.pulse_start];
        start_known = true;
        for (int i = 0; i < pulse_count; ++i) {
            index += ch.pulse_offset[i];
            target[i] = index;
            amp[i] = ch.pulse_amp[i];
        }
    }
    Bits b = bits;
    uint32_t pns = st.pns_state;
    int status = EC_OK;
    // a long window is one group of one window of 1024 lines
    const int limit = is_short ? 128 : 1024, groups = is_short ? num_groups : 1;
    enum { OPEN_GROUP, OPEN_BAND, IN_BAND, DONE };
    int phase = OPEN_GROUP;
    int g = 0, sfb = 0, w0 = 0, glen = 0, w = 0, i = 0, s = 0, e = 0, dim = 4;
    float scale = 0.0f;
    BookRef br = BookRef{t.lut, t.tuples, 1};
    int q[4];
    while (phase != DONE) {
#if defined(__HIP_DEVICE_COMPILE__)
        // keep this ONE loop: left alone, jump threading turns the state machine back into a loop per state, and a wave
        // whose lanes are in different states runs those loops one after the other
        asm volatile("" : "+v"(phase));
#endif
        if (phase == OPEN_GROUP) {
            if (g >= groups) {
                phase = DONE;
            } else {
                glen = is_short ? ics.group_len[g] : 1;
                if (is_short && (glen == 0 || w0 + glen > 8)) {
                    status = EC_INVALID_BITSTREAM;
                    phase = DONE;
                } else {
                    sfb = 0;
                    phase = OPEN_BAND;
                }
            }
        }
        if (phase == OPEN_BAND) {
            if (sfb >= max_sfb) {
                w0 += glen;
                ++g;
                phase = OPEN_GROUP;
            } else {
                status = band_range(off, bands, sfb, &s, &e);
                if (status == EC_OK && e > limit) status = EC_INVALID_CONFIG;
                if (status != EC_OK) {
                    phase = DONE;
                } else {
                    const int book = ch.book[g * stride + sfb];
                    scale = ch.mult[g * stride + sfb];
                    if (book >= 1 && book <= 11) {
                        br = book_ref(t, book);
                        dim = book <= 4 ? 4 : 2;
                        if (s + dim <= e) {
                            w = w0;
                            i = s;
                            phase = IN_BAND;
                        } else {
                            ++sfb;
                        }
                    } else {
                        if (book == BOOK_NOISE && (is_short || !pulse_present)) {
                            if (mode == PNS_COUNT) *noise_samples += (uint32_t)(glen * (e - s));
                            else
                                for (int ww = w0; ww < w0 + glen && status == EC_OK; ++ww)
                                    status = noise_band(scale, pns, coef + (is_short ? ww * 128 : 0) + s, e - s);
                            if (status != EC_OK) phase = DONE;
                        }
                        ++sfb;
                    }
                }
            }
        }
        if (phase == IN_BAND) {
            status = read_tuple(br, b, q);
            if (status != EC_OK) {
                phase = DONE;
            } else {
                for (int p = 0; p < pulse_count; ++p) {  // pulses apply in order; two may hit the same coefficient
                    const int k = target[p] - i;
                    if (k >= 0 && k < dim) q[k] += q[k] > 0 ? amp[p] : -amp[p];
                }
                const int at = (is_short ? w * 128 : 0) + i;
#if defined(__HIP_DEVICE_COMPILE__)
                // device: a codeword's values leave in ONE store (16 bytes for the four-value books, 8 for the others; bands
                // start on multiples of four lines): a scalar store is a separate partial cache line per lane and value
                if (!QUANT && (((uintptr_t)coef) & 15u) == 0 && (at & (dim - 1)) == 0) {
                    if (dim == 4) {
                        NoiseQuad v4;
                        for (int k = 0; k < 4; ++k) v4.v[k] = dequantize(t, q[k], scale);
                        memcpy((NoiseQuad *)__builtin_assume_aligned(coef + at, 16), &v4, 16);
                    } else {
                        struct alignas(8) Pair { float v[2]; } v2;
                        for (int k = 0; k < 2; ++k) v2.v[k] = dequantize(t, q[k], scale);
                        memcpy((Pair *)__builtin_assume_aligned(coef + at, 8), &v2, 8);
                    }
                } else
#endif
                for (int k = 0; k < 4; ++k) {
                    if (k >= dim) break;
                    if (QUANT) {
                        quant[at + k] = quant_store(q[k], wide_base + at + k, wide, &status);
                    } else {
                        coef[at + k] = dequantize(t, q[k], scale);
                    }
                }
                if (QUANT && status != EC_OK) {
                    phase = DONE;
                } else {
                    i += dim;
                    if (i + dim > e) {
                        i = s;
                        ++w;
                        if (w >= w0 + glen) {
                            ++sfb;
                            phase = OPEN_BAND;
                        }
                    }
                }
            }
        }
    }
    if (is_short && status == EC_OK && w0 != 8) status = EC_INVALID_BITSTREAM;
    bits = b;
    st.pns_state = pns;
    if (is_short || status != EC_OK || !pulse_present) return status;
    // apply_pulse_data's checks (spectral.rs:2198-2247), in its order
    if (ch.pulse_start >= max_sfb) return EC_INVALID_BITSTREAM;
    if (!start_known) return EC_INVALID_CONFIG;  // band_range(start_sfb) failed
    for (int p = 0; p < pulse_count; ++p) {
        const int index = target[p];
        if (index >= 1024) return EC_INVALID_BITSTREAM;
        int band = -1;
        for (int k = 0; k < max_sfb; ++k) {
            int bs, be;
            EC_TRY(band_range(off, bands, k, &bs, &be));
            if (index >= bs && index < be) {
                band = k;
                break;
            }
        }
        if (band < 0) return EC_INVALID_BITSTREAM;
        const int book = ch.book[band];
        if (!(book >= 1 && book <= 11)) return EC_INVALID_BITSTREAM;
    }
    for (int k = 0; k < max_sfb; ++k) {  // the noise bands of the pulse path come last
        if (ch.book[k] != BOOK_NOISE) continue;
        int bs, be;
        EC_TRY(band_range(off, bands, k, &bs, &be));
        if (mode == PNS_COUNT) *noise_samples += (uint32_t)(be - bs);
        else EC_TRY(noise_band(ch.mult[k], st.pns_state, coef + bs, be - bs));
    }
    return EC_OK;
}

template <bool QUANT>
SKE int decode_spectrum_t(const Tables &t, Stream &st, Bits &bits, const Channel &ch, bool allow_intensity, float *__restrict__ coef,
                          PnsMode mode, uint32_t *noise_samples, int16_t *__restrict__ quant, WideList *wide = nullptr, int wide_base = 0) {
#if defined(__HIP_DEVICE_COMPILE__) || defined(SK_EC_FLAT)
    return decode_spectrum_flat<QUANT>(t, st, bits, ch, allow_intensity, coef, mode, noise_samples, quant, wide, wide_base);
#else
    return decode_spectrum_nested<QUANT>(t, st, bits, ch, allow_intensity, coef, mode, noise_samples, quant, wide, wide_base);
#endif
}

SKE int decode_spectrum(const Tables &t, Stream &st, Bits &bits, const Channel &ch, bool allow_intensity, float *__restrict__ coef,
                        PnsMode mode, uint32_t *noise_samples) {
    return decode_spectrum_t<false>(t, st, bits, ch, allow_intensity, coef, mode, noise_samples, nullptr);
}

// the noise bands of one channel, in the order decode_spectrum generates them (PNS_COUNT left them open)
SKE int fill_noise(const Tables &t, Stream &st, const Channel &ch, float *__restrict__ coef) {
    const Ics ics = ch.ics;
    const int stride = band_stride(ics), max_sfb = ics.max_sfb;
    const uint16_t *off;
    int bands;
    EC_TRY(layout(t, st, ics, &off, &bands));
    uint32_t pns = st.pns_state;
    int status = EC_OK;
    if (ics.sequence == SEQ_EIGHT_SHORT) {
        int w0 = 0;
        for (int g = 0; g < ics.num_groups && status == EC_OK; ++g) {
            const int glen = ics.group_len[g];
            for (int sfb = 0; sfb < max_sfb && status == EC_OK; ++sfb) {
                if (ch.book[g * stride + sfb] != BOOK_NOISE) continue;
                int s, e;
                status = band_range(off, bands, sfb, &s, &e);
                for (int w = w0; w < w0 + glen && status == EC_OK; ++w)
                    status = noise_band(ch.mult[g * stride + sfb], pns, coef + w * 128 + s, e - s);
            }
            w0 += glen;
        }
    } else {
        for (int sfb = 0; sfb < max_sfb && status == EC_OK; ++sfb) {
            if (ch.book[sfb] != BOOK_NOISE) continue;
            int s, e;
            status = band_range(off, bands, sfb, &s, &e);
            if (status == EC_OK) status = noise_band(ch.mult[sfb], pns, coef + s, e - s);
        }
    }
    st.pns_state = pns;
    return status;
}

// ---- stereo tools (decoder.rs:268-334, stereo.rs) ----------------------------------------------------------------
struct MsMask {
    uint8_t mode;        // 0 none, 1 per band, 2 all
    uint8_t used[128];   // [group * stride + sfb]
};

SKE int read_ms_mask(Bits &b, const Ics &ics, MsMask &m) {  // channel.rs:222-251
    uint32_t v;
    EC_TRY(read_bits(b, 2, &v));
    m.mode = (uint8_t)v;
    if (v == 3) return EC_INVALID_BITSTREAM;
    if (v == 1) {
        const int stride = band_stride(ics);
        for (int i = 0; i < 128; ++i) m.used[i] = 0;
        for (int g = 0; g < ics.num_groups; ++g)
            for (int sfb = 0; sfb < ics.max_sfb; ++sfb) {
                bool f;
                EC_TRY(read_flag(b, &f));
                m.used[g * stride + sfb] = f;
            }
    }
    return EC_OK;
}

struct alignas(16) Quad {
    float v[4];
};

SKE int stereo_tools(const Tables &t, const Stream &st, const MsMask &mask, const Ics &ics_in, const Channel &lch, const Channel &rch,
                     float *left, float *right) {
    // Everything the band loops ask of the side record is fetched ONCE: the record sits in memory the float stores below may
    // alias as far as the compiler knows (its fields are bytes), so through references every loop condition and every band's
    // codebook / mask lookup is a reload -- on the device a global-memory round trip per band, twice over.  The per-band facts
    // become four 128-bit masks (index g * stride + sfb): M/S selected, right band intensity, its sign, band exempt from M/S.
    const Ics ics = ics_in;
    const uint8_t mask_mode = mask.mode;
    uint64_t sel[2] = {0, 0}, is_band[2] = {0, 0}, is_neg[2] = {0, 0}, exempt[2] = {0, 0};
    const int words_used = ics.sequence == SEQ_EIGHT_SHORT ? 4 * (int)ics.num_groups : ((int)ics.max_sfb + 3) / 4;
    for (int wd = 0; wd < 32 && wd < words_used; ++wd) {  // entries g * stride + sfb with sfb < max_sfb (<= 15 / <= 51)
        uint32_t lw, rw, uw;
        memcpy(&lw, lch.book + 4 * wd, 4);
        memcpy(&rw, rch.book + 4 * wd, 4);
        memcpy(&uw, mask.used + 4 * wd, 4);
        for (int b = 0; b < 4; ++b) {
            const uint32_t lb = (lw >> (8 * b)) & 0xffu, rb = (rw >> (8 * b)) & 0xffu, used = (uw >> (8 * b)) & 0xffu;
            const int at = 4 * wd + b;
            const uint64_t bit = (uint64_t)1 << (at & 63);
            const bool intensity = rb == BOOK_INTENSITY || rb == BOOK_INTENSITY_NEG;
            if (mask_mode == 2 || (mask_mode == 1 && used)) sel[at >> 6] |= bit;
            if (intensity) is_band[at >> 6] |= bit;
            if (rb == BOOK_INTENSITY_NEG) is_neg[at >> 6] |= bit;
            if (intensity || lb == BOOK_NOISE || rb == BOOK_NOISE) exempt[at >> 6] |= bit;
        }
    }
    const uint16_t *off;
    int bands;
    EC_TRY(layout(t, st, ics, &off, &bands));
    const bool is_short = ics.sequence == SEQ_EIGHT_SHORT;
    const int wlen = is_short ? 128 : 1024, stride = band_stride(ics);
    for (int pass = 0; pass < 2; ++pass) {  // intensity first, then mid/side
        int w0 = 0;
        for (int g = 0; g < ics.num_groups; ++g) {
            const int glen = is_short ? ics.group_len[g] : 1;
            if (is_short) {
                if (glen == 0) return EC_INVALID_BITSTREAM;
                if (w0 + glen > 8) return EC_INVALID_BITSTREAM;
            }
            for (int sfb = 0; sfb < ics.max_sfb; ++sfb) {
                int s, e;
                EC_TRY(band_range(off, bands, sfb, &s, &e));
                if (e > wlen) return EC_INVALID_CONFIG;
                const int at = g * stride + sfb;
                const bool selected = (sel[at >> 6] >> (at & 63)) & 1u;
                if (pass == 0) {
                    if (!((is_band[at >> 6] >> (at & 63)) & 1u)) continue;
                    float sign = ((is_neg[at >> 6] >> (at & 63)) & 1u) ? -1.0f : 1.0f;  // stereo.rs:431-437
                    if (selected) sign = -sign;                        // stereo.rs:145-149
                    const float scale = rch.mult[g * stride + sfb];
                    for (int w = w0; w < w0 + glen; ++w) {
                        int i = w * wlen + s;
                        for (; i + 4 <= w * wlen + e; i += 4) {  // four loads in flight (one lane per unit: each is a round trip)
                            const float l0 = left[i], l1 = left[i + 1], l2 = left[i + 2], l3 = left[i + 3];
                            right[i] = l0 * scale * sign;
                            right[i + 1] = l1 * scale * sign;
                            right[i + 2] = l2 * scale * sign;
                            right[i + 3] = l3 * scale * sign;
                        }
                        for (; i < w * wlen + e; ++i) right[i] = left[i] * scale * sign;
                    }
                } else {
                    if (!selected) continue;
                    if ((exempt[at >> 6] >> (at & 63)) & 1u) continue;
                    for (int w = w0; w < w0 + glen; ++w) {
                        int i = w * wlen + s;
                        // One lane per unit: a scalar load is 32-64 separate cache lines per wave instruction, and the rate at
                        // which those are looked up, not their latency, bounds this loop.  Bands start and end on multiples of
                        // four lines (checked), the spectra are 4 KiB-aligned: 16-byte accesses, four lines per look-up.
                        if ((((uintptr_t)left | (uintptr_t)right) & 15u) == 0 && ((i | (w * wlen + e)) & 3) == 0) {
                            for (; i + 4 <= w * wlen + e; i += 4) {
                                Quad mq, sq;
                                memcpy(&mq, (const Quad *)__builtin_assume_aligned(left + i, 16), 16);
                                memcpy(&sq, (const Quad *)__builtin_assume_aligned(right + i, 16), 16);
                                const Quad lo = {{mq.v[0] + sq.v[0], mq.v[1] + sq.v[1], mq.v[2] + sq.v[2], mq.v[3] + sq.v[3]}};
                                const Quad ro = {{mq.v[0] - sq.v[0], mq.v[1] - sq.v[1], mq.v[2] - sq.v[2], mq.v[3] - sq.v[3]}};
                                memcpy((Quad *)__builtin_assume_aligned(left + i, 16), &lo, 16);
                                memcpy((Quad *)__builtin_assume_aligned(right + i, 16), &ro, 16);
                            }
                        }
                        for (; i < w * wlen + e; ++i) {
                            const float mid = left[i], side = right[i];
                            left[i] = mid + side;
                            right[i] = mid - side;
                        }
                    }
                }
            }
            w0 += glen;
        }
        if (is_short && w0 != 8) return EC_INVALID_BITSTREAM;
    }
    return EC_OK;
}

// ---- TNS (tns.rs:103-276) ----------------------------------------------------------------------------------------
SKE int tns_coefficient(const Tables &t, int encoded, int coef_bits, int res_bits, float *out) {  // tns.rs:208-235
    if (coef_bits == 0 || coef_bits > 4 || res_bits < 3 || res_bits > 4) return EC_INVALID_BITSTREAM;
    const int raw = encoded & ((1 << coef_bits) - 1);
    const int boundary = 1 << (coef_bits - 1);
    const int sgn = raw < boundary ? -raw : (1 << coef_bits) - raw;
    if (sgn == 0) {
        *out = 0.0f;
        return EC_OK;
    }
    *out = t.tns_sin[(res_bits - 3) * 17 + sgn + 8];  // sgn in [-7, 8]: the host's sinf, tabulated
    return EC_OK;
}

// The all-pole recursion of apply_tns_filter (tns.rs:237-276) over n lines from `first` in direction `step`, order <= TAPS.
// The `order` outputs the recursion reads back stay in a register shift line (newest first), and the lines themselves move
// in blocks of eight -- eight loads in flight, eight results, eight stores: one lane per access unit means every load is a
// dependent, uncoalesced round trip, and one per line was most of the kernel's time.  Same operations in the same order as
// the reference's loop (TAPS only bounds the unrolled tap loop; taps beyond min(done, order) are skipped as there).
template <int TAPS>
SKE void tns_filter(float *c, int first, int n, int step, int order, const float *lpc20) {
    float hist[TAPS], lpc[TAPS];
    for (int i = 0; i < TAPS; ++i) {
        hist[i] = 0.0f;
        lpc[i] = lpc20[i];
    }
    int pos = first, done = 0;
    auto line = [&](float v, int at) {  // one step of the recursion; `at` = lines done before this one
        const int mo = at < order ? at : order;
        for (int o = 1; o <= TAPS; ++o)
            if (o <= mo) v -= hist[o - 1] * lpc[o - 1];
        for (int k = TAPS - 1; k > 0; --k) hist[k] = hist[k - 1];
        hist[0] = v;
        return v;
    };
    // whole blocks: the next block's eight loads are in flight while this one is filtered and stored
    float x[8], nx[8];
    if (n >= 8)
        for (int j = 0; j < 8; ++j) x[j] = c[pos + j * step];
    for (; done + 8 <= n; done += 8, pos += 8 * step) {
        const bool more = done + 16 <= n;
        if (more)
            for (int j = 0; j < 8; ++j) nx[j] = c[pos + (8 + j) * step];
        for (int j = 0; j < 8; ++j) x[j] = line(x[j], done + j);
        for (int j = 0; j < 8; ++j) c[pos + j * step] = x[j];
        if (more)
            for (int j = 0; j < 8; ++j) x[j] = nx[j];
    }
    for (; done < n; ++done, pos += step) c[pos] = line(c[pos], done);
}

SKE int apply_tns(const Tables &t, const Stream &st, const Channel &ch, float *coef) {
    const Ics ics = ch.ics;  // a copy: through the reference every use is a reload (see stereo_tools)
    if (st.sf_index < 0) return EC_UNSUPPORTED_FEATURE;
    if (st.sf_index > 12) return EC_UNSUPPORTED_SF_INDEX;
    const bool is_short = ics.sequence == SEQ_EIGHT_SHORT;
    const uint16_t *off;
    int bands;
    EC_TRY(layout(t, st, ics, &off, &bands));
    const int wlen = is_short ? 128 : 1024;
    int limit = (int)t.meta[(is_short ? META_TNS_MAX_SHORT : META_TNS_MAX_LONG) + st.sf_index];
    if (limit > ics.max_sfb) limit = ics.max_sfb;
    if (limit > bands) limit = bands;
    for (int w = 0; w < ics.num_windows; ++w) {
        const TnsWindow &win = ch.tns[w];
        const int res_bits = win.coef_res ? 4 : 3;
        int bottom = bands;
        for (int f = 0; f < win.filter_count; ++f) {
            const TnsFilter &flt = win.filter[f];
            const int top = bottom;
            bottom = top > flt.length ? top - flt.length : 0;
            if (flt.order == 0) continue;
            const int start = off[bottom < limit ? bottom : limit], end = off[top < limit ? top : limit];
            if (end <= start) continue;
            float lpc[20], prev[20];  // tns_lpc_coefficients, tns.rs:176-206
            for (int i = 0; i < 20; ++i) lpc[i] = prev[i] = 0.0f;
            for (int i = 0; i < flt.order; ++i) {
                float c;
                EC_TRY(tns_coefficient(t, flt.coef[i], flt.coef_bits, res_bits, &c));
                const float refl = -c;
                lpc[i] = refl;
                for (int k = 0; k < ((i + 1) >> 1); ++k) {
                    const float fwd = prev[k], bwd = prev[i - 1 - k];
                    lpc[k] = fwd + refl * bwd;
                    lpc[i - 1 - k] = bwd + refl * fwd;
                }
                for (int k = 0; k <= i; ++k) prev[k] = lpc[k];
            }
            // apply_tns_filter, tns.rs:237-276
            float *c = coef + w * wlen;
            const int n = end - start, step = flt.direction ? -1 : 1;
            const int first = flt.direction ? end - 1 : start;
            if (flt.order <= 4) tns_filter<4>(c, first, n, step, flt.order, lpc);
            else if (flt.order <= 8) tns_filter<8>(c, first, n, step, flt.order, lpc);
            else if (flt.order <= 12) tns_filter<12>(c, first, n, step, flt.order, lpc);
            else tns_filter<20>(c, first, n, step, flt.order, lpc);
        }
    }
    return EC_OK;
}

// ---- element loop (decoder.rs:104-218, 393-438) --------------------------------------------------------------------
SKE bool rest_is_zero(const Bits &b) {
    Bits p = b;
    while (p.total - p.pos >= 32) {
        if (peek32(p) != 0) return false;
        p.pos += 32;
    }
    const uint32_t rem = p.total - p.pos;
    return rem == 0 || (peek32(p) >> (32 - rem)) == 0;
}

struct Scratch {  // an access unit's side information: per-lane working storage, and what the first phase hands to the second
    Channel ch[2];
    MsMask mask;
    uint8_t is_pair, common_window;
    uint32_t resume_pos;     // bit position after the channel element
    uint32_t noise_samples;  // PNS samples the unit consumes (both channels)
};

// First phase: everything up to and including the spectral data of the channel element.  With PNS_GENERATE the noise
// bands are filled on the way (the reference's order); with PNS_COUNT they are left open and only counted, which makes
// the phase independent of every other access unit of the stream.
struct QuantCapture {  // host side of the quantised hand-over: where parse_unit leaves the integers
    int16_t *quant;    // [channels][1024] quantised spectral values, pulses applied
    int16_t *sf[2];    // [128] each: transmitted scale factor / noise energy / intensity position per band
    WideList *wide;    // values beyond i16 (kWideMarker in quant); may be null: such a value is then an unsupported feature
};

SKE int parse_unit(const Tables &t, Stream &st, const uint32_t *au, uint32_t len_bytes, float *coef, uint8_t *sequence, uint8_t *shape,
                   Scratch &s, PnsMode mode, const QuantCapture *qc = nullptr) {
    Bits b = make_bits(au, len_bytes);
    s.noise_samples = 0;
    // The element loop only steps over leading fill elements; the channel element is parsed BEHIND it.  Inside the loop a
    // wave whose lanes reach their channel element in different passes (one unit opens with a fill element -- FFmpeg's
    // encoder string in the first unit of a stream -- the other 31 do not) would run the whole parse once per pass, one
    // after the other: measured, half the waves of a tick took twice as long (profiles/r02_entropy_parse.md).
    uint32_t id = 7, tag = 0;
    bool found = false;
    while (b.total - b.pos >= 3) {  // each pass consumes >= 3 bits: bounded by the length of the access unit
        EC_TRY(read_bits(b, 3, &id));
        if (id <= 5) EC_TRY(read_bits(b, 4, &tag));  // syntax.rs:54-63
        if (id != 6) {
            found = true;
            break;
        }
        // fill element, decoder.rs:393-419
        uint32_t count;
        EC_TRY(read_bits(b, 4, &count));
        if (count == 15) {
            uint32_t ext;
            EC_TRY(read_bits(b, 8, &ext));
            if (ext == 0) return EC_INVALID_BITSTREAM;
            count += ext - 1;
        }
        if (count == 0) continue;
        if (b.total - b.pos < count * 8) return EC_EOF;
        const uint32_t ext_type = peek32(b) >> 28;
        if (ext_type == 13 || ext_type == 14) return EC_UNSUPPORTED_FEATURE;  // SBR
        b.pos += count * 8;
    }
    (void)tag;
    if (!found || id == 7) return EC_INVALID_BITSTREAM;  // END, or the unit ran out, before any channel element:
                                                         // "raw access unit does not contain an AAC-LC channel element"
    if (id >= 2) return EC_UNSUPPORTED_FEATURE;          // CCE / LFE / DSE / PCE
    if (id == 0) {  // single channel element, decoder.rs:165-183
        if (st.channels != 1) return EC_INVALID_BITSTREAM;
        Channel &ch = s.ch[0];
        EC_TRY(read_channel(t, b, ch, nullptr, qc ? qc->sf[0] : nullptr));
        if (qc) EC_TRY(decode_spectrum_t<true>(t, st, b, ch, false, nullptr, PNS_COUNT, &s.noise_samples, qc->quant, qc->wide, 0));
        else EC_TRY(decode_spectrum(t, st, b, ch, false, coef, mode, &s.noise_samples));
        s.is_pair = 0;
        s.common_window = 0;
        sequence[0] = ch.ics.sequence;
        shape[0] = ch.ics.shape;
        sequence[1] = shape[1] = 0;
        s.resume_pos = b.pos;
        return EC_OK;
    }
    // channel pair element, decoder.rs:185-218
    if (st.channels != 2) return EC_INVALID_BITSTREAM;
    bool common_window;
    EC_TRY(read_flag(b, &common_window));
    Ics common;
    s.mask.mode = 0;
    if (common_window) {
        EC_TRY(read_ics(b, common));
        EC_TRY(read_ms_mask(b, common, s.mask));
    }
    Channel &left = s.ch[0], &right = s.ch[1];
    EC_TRY(read_channel(t, b, left, common_window ? &common : nullptr, qc ? qc->sf[0] : nullptr, 9));
    if (qc) EC_TRY(decode_spectrum_t<true>(t, st, b, left, false, nullptr, PNS_COUNT, &s.noise_samples, qc->quant, qc->wide, 0));
    else EC_TRY(decode_spectrum(t, st, b, left, false, coef, mode, &s.noise_samples));
    EC_TRY(read_channel(t, b, right, common_window ? &common : nullptr, qc ? qc->sf[1] : nullptr));
    if (qc) EC_TRY(decode_spectrum_t<true>(t, st, b, right, true, nullptr, PNS_COUNT, &s.noise_samples, qc->quant + 1024, qc->wide, 1024));
    else EC_TRY(decode_spectrum(t, st, b, right, true, coef + 1024, mode, &s.noise_samples));
    s.is_pair = 1;
    s.common_window = common_window;
    sequence[0] = left.ics.sequence;
    shape[0] = left.ics.shape;
    sequence[1] = right.ics.sequence;
    shape[1] = right.ics.shape;
    s.resume_pos = b.pos;
    return EC_OK;
}

SKE int unit_tail(const uint32_t *au, uint32_t len_bytes, uint32_t resume_pos);

// Second phase: [the noise bands, if the first phase only counted them,] stereo tools, TNS, then the rest of the access
// unit (fill elements, END, the trailing-zero rule) from where the first phase stopped.
SKE int finish_unit(const Tables &t, Stream &st, const uint32_t *au, uint32_t len_bytes, float *coef, Scratch &s, bool fill) {
    Channel &left = s.ch[0], &right = s.ch[1];
    if (fill) {
        EC_TRY(fill_noise(t, st, left, coef));
        if (s.is_pair) EC_TRY(fill_noise(t, st, right, coef + 1024));
    }
    if (s.is_pair) {
        if (!s.common_window) {  // decoder.rs:275-285
            const int stride = band_stride(right.ics);
            for (int g = 0; g < right.ics.num_groups; ++g)
                for (int sfb = 0; sfb < right.ics.max_sfb; ++sfb) {
                    const int book = right.book[g * stride + sfb];
                    if (book == BOOK_INTENSITY || book == BOOK_INTENSITY_NEG) return EC_INVALID_BITSTREAM;
                }
        } else {
            EC_TRY(stereo_tools(t, st, s.mask, left.ics, left, right, coef, coef + 1024));
        }
    }
    if (left.tns_present) EC_TRY(apply_tns(t, st, left, coef));
    if (s.is_pair && right.tns_present) EC_TRY(apply_tns(t, st, right, coef + 1024));
    return unit_tail(au, len_bytes, s.resume_pos);
}

// the rest of the access unit behind the channel element: fill elements, END, the trailing-zero rule (decoder.rs:134-161)
SKE int unit_tail(const uint32_t *au, uint32_t len_bytes, uint32_t resume_pos) {
    Bits b = make_bits(au, len_bytes);
    b.pos = resume_pos;
    while (b.total - b.pos >= 3) {
        if (rest_is_zero(b)) break;
        uint32_t id, tag;
        EC_TRY(read_bits(b, 3, &id));
        if (id <= 5) EC_TRY(read_bits(b, 4, &tag));
        if (id <= 1) return EC_INVALID_BITSTREAM;  // "raw access unit contains multiple channel elements"
        if (id <= 5) return EC_UNSUPPORTED_FEATURE;
        if (id == 7) break;
        uint32_t count;  // fill element
        EC_TRY(read_bits(b, 4, &count));
        if (count == 15) {
            uint32_t ext;
            EC_TRY(read_bits(b, 8, &ext));
            if (ext == 0) return EC_INVALID_BITSTREAM;
            count += ext - 1;
        }
        if (count == 0) continue;
        if (b.total - b.pos < count * 8) return EC_EOF;
        const uint32_t ext_type = peek32(b) >> 28;
        if (ext_type == 13 || ext_type == 14) return EC_UNSUPPORTED_FEATURE;  // SBR
        b.pos += count * 8;
    }
    if (!rest_is_zero(b)) return EC_INVALID_BITSTREAM;
    return EC_OK;
}

// One access unit -> spectra [channels][1024] and the window fields, in the reference's order of operations.
SKE int decode_access_unit(const Tables &t, Stream &st, const uint32_t *au, uint32_t len_bytes, float *coef, uint8_t *sequence,
                           uint8_t *shape, Scratch &s) {
    EC_TRY(parse_unit(t, st, au, len_bytes, coef, sequence, shape, s, PNS_GENERATE));
    return finish_unit(t, st, au, len_bytes, coef, s, false);
}

// LCG jump-ahead: the generator state after `n` steps of state = state * 1664525 + 1013904223 (spectral.rs:2447-2450)
SKE uint32_t pns_advance(uint32_t state, uint32_t n) {
    uint32_t mul = 1664525u, add = 1013904223u, acc_mul = 1u, acc_add = 0u;
    for (int bit = 0; bit < 32; ++bit) {
        if (n & 1u) {
            acc_add = acc_add * mul + add;
            acc_mul *= mul;
        }
        add = add * mul + add;
        mul *= mul;
        n >>= 1;
    }
    return state * acc_mul + acc_add;
}

// ---- quantised hand-over: host Huffman decode -> device dequantisation, noise, stereo tools, TNS -------
// What crosses PCIe per access unit instead of channels x 1024 f32: channels x 1024 i16 quantised values plus this record
// (section codebooks, transmitted scale factors, window / grouping, mid/side mask, TNS filters): sk_aac_unit_side in the C ABI.
struct WireTnsFilter {
    uint8_t window, length, order, direction, coef_bits;
    int8_t coef[20];
    uint8_t pad[3];
};
struct WireChannel {
    uint8_t sequence, shape, max_sfb, num_windows, num_groups, tns_present, n_filters, reserved;
    uint8_t group_len[8];
    uint8_t tns_filter_count[8], tns_coef_res[8];
    uint8_t book[128];
    int16_t sf[128];
    WireTnsFilter filter[8];  // long: <= 3 in window 0; eight-short: <= 1 per window
};
struct WireUnit {
    uint8_t channels, is_pair, common_window, ms_mode;
    int32_t tail_status;     // unit_tail's verdict (the host has the bitstream); applies if nothing fails before it
    uint32_t noise_samples;  // PNS samples the unit consumes: the generator jump of the units behind it
    uint8_t ms_used[16];     // bit i = MsMask::used[i]
    WireChannel ch[2];
    uint16_t n_wide, wide_pos[kWideMax + 1];  // quantised values beyond i16 (kWideMarker in the i16 array): position = channel * 1024 + line
    int32_t wide_val[kWideMax];
};

SKE void pack_unit(const Scratch &s, int channels, const int16_t *const sf[2], int32_t tail_status, WireUnit &w, const WideList *wide = nullptr) {
    for (size_t i = 0; i < sizeof w; ++i) reinterpret_cast<uint8_t *>(&w)[i] = 0;
    if (wide)
        for (uint32_t i = 0; i < wide->n && i < (uint32_t)kWideMax; ++i) {
            w.wide_pos[i] = wide->pos[i];
            w.wide_val[i] = wide->val[i];
            w.n_wide = (uint16_t)(i + 1);
        }
    w.channels = (uint8_t)channels;
    w.is_pair = s.is_pair;
    w.common_window = s.common_window;
    w.ms_mode = s.is_pair && s.common_window ? s.mask.mode : 0;
    w.tail_status = tail_status;
    w.noise_samples = s.noise_samples;
    if (w.ms_mode == 1)
        for (int i = 0; i < 128; ++i)
            if (s.mask.used[i]) w.ms_used[i >> 3] |= (uint8_t)(1u << (i & 7));
    for (int c = 0; c < channels; ++c) {
        const Channel &ch = s.ch[c];
        WireChannel &wc = w.ch[c];
        wc.sequence = ch.ics.sequence;
        wc.shape = ch.ics.shape;
        wc.max_sfb = ch.ics.max_sfb;
        wc.num_windows = ch.ics.num_windows;
        wc.num_groups = ch.ics.num_groups;
        wc.tns_present = ch.tns_present;
        for (int i = 0; i < 8; ++i) wc.group_len[i] = ch.ics.group_len[i];
        for (int i = 0; i < 128; ++i) {
            wc.book[i] = ch.book[i];
            wc.sf[i] = sf[c][i];
        }
        if (ch.tns_present)
            for (int win = 0; win < ch.ics.num_windows; ++win) {
                wc.tns_filter_count[win] = ch.tns[win].filter_count;
                wc.tns_coef_res[win] = ch.tns[win].coef_res;
                for (int f = 0; f < ch.tns[win].filter_count && wc.n_filters < 8; ++f) {
                    const TnsFilter &src = ch.tns[win].filter[f];
                    WireTnsFilter &dst = wc.filter[wc.n_filters++];
                    dst.window = (uint8_t)win;
                    dst.length = src.length;
                    dst.order = src.order;
                    dst.direction = src.direction;
                    dst.coef_bits = src.coef_bits;
                    for (int i = 0; i < 20; ++i) dst.coef[i] = i < src.order ? src.coef[i] : 0;
                }
            }
    }
}

// the device's side of it: the record back into the structures finish_unit works on (multipliers from the tables, as
// read_scalefactors derives them)
// Returns EC_INVALID_CONFIG for a record no parse_unit could have produced: the counts below drive loops and index
// arrays in dequant_channel / finish_unit, and sk_tick_run_q is a public entry point.
SKE int unpack_unit(const Tables &t, const WireUnit &w, Scratch &s) {
    if (w.channels < 1 || w.channels > 2 || w.is_pair != (w.channels == 2 ? 1 : 0) || w.common_window > 1 || w.ms_mode > 3) return EC_INVALID_CONFIG;
    s.is_pair = w.is_pair;
    s.common_window = w.common_window;
    s.resume_pos = 0;
    s.noise_samples = w.noise_samples;
    s.mask.mode = w.ms_mode;
    for (int i = 0; i < 128; ++i) s.mask.used[i] = (uint8_t)((w.ms_used[i >> 3] >> (i & 7)) & 1u);
    for (int c = 0; c < (int)w.channels; ++c) {
        const WireChannel &wc = w.ch[c];
        Channel &ch = s.ch[c];
        const bool is_short = wc.sequence == SEQ_EIGHT_SHORT;
        if (wc.sequence > 3 || wc.shape > 1 || wc.num_windows != (is_short ? 8 : 1) || wc.num_groups < 1 || wc.num_groups > (is_short ? 8 : 1) ||
            wc.max_sfb > (is_short ? 15 : 51) || wc.tns_present > 1 || wc.n_filters > 8)
            return EC_INVALID_CONFIG;
        int windows = 0;
        for (int g = 0; g < (int)wc.num_groups; ++g) {
            if (wc.group_len[g] < 1 || wc.group_len[g] > 8) return EC_INVALID_CONFIG;
            windows += wc.group_len[g];
        }
        if (windows != (int)wc.num_windows) return EC_INVALID_CONFIG;
        ch.ics.sequence = wc.sequence;
        ch.ics.shape = wc.shape;
        ch.ics.max_sfb = wc.max_sfb;
        ch.ics.num_windows = wc.num_windows;
        ch.ics.num_groups = wc.num_groups;
        for (int i = 0; i < 8; ++i) ch.ics.group_len[i] = wc.group_len[i];
        ch.global_gain = 0;
        ch.pulse_present = 0;
        ch.pulse_start = ch.pulse_count = 0;
        for (int i = 0; i < 4; ++i) ch.pulse_offset[i] = ch.pulse_amp[i] = 0;
        ch.tns_present = wc.tns_present;
        for (int i = 0; i < 128; ++i) {
            const int book = wc.book[i], sf = wc.sf[i];
            if (book > 15 || book == 12) return EC_INVALID_CONFIG;
            ch.book[i] = (uint8_t)book;
            float m = 0.0f;
            if (book == BOOK_INTENSITY || book == BOOK_INTENSITY_NEG) m = (sf >= -256 && sf <= 255) ? t.is_mult[sf + 256] : t.is_wide[sf + 32768];
            else if (book != BOOK_ZERO) m = sf_multiplier(t, sf);
            ch.mult[i] = m;
        }
        for (int win = 0; win < 8; ++win) {
            if (wc.tns_filter_count[win] > 4 || (win >= (int)wc.num_windows && wc.tns_filter_count[win])) return EC_INVALID_CONFIG;
            ch.tns[win].filter_count = wc.tns_filter_count[win];
            ch.tns[win].coef_res = wc.tns_coef_res[win];
            for (int f = 0; f < 4; ++f) {
                TnsFilter &z = ch.tns[win].filter[f];
                z.length = z.order = z.direction = z.coef_bits = 0;
                for (int i = 0; i < 20; ++i) z.coef[i] = 0;
            }
        }
        int slot[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        for (int k = 0; k < (int)wc.n_filters; ++k) {
            const WireTnsFilter &src = wc.filter[k];
            const int win = src.window & 7;
            if (src.window > 7 || slot[win] >= (int)wc.tns_filter_count[win] || src.order > 20) return EC_INVALID_CONFIG;
            TnsFilter &dst = ch.tns[win].filter[slot[win]++];
            dst.length = src.length;
            dst.order = src.order;
            dst.direction = src.direction;
            dst.coef_bits = src.coef_bits;
            for (int i = 0; i < 20; ++i) dst.coef[i] = src.coef[i];
        }
        for (int win = 0; win < 8; ++win)
            if (wc.tns_present && slot[win] != (int)wc.tns_filter_count[win]) return EC_INVALID_CONFIG;  // every counted filter was sent
    }
    return EC_OK;
}

// dsp.rs:397-405 over one channel's coded bands (decode_spectrum's stores, from the integers): zero, noise and intensity
// bands stay zero (noise is filled by fill_noise, intensity by the stereo tools)
SKE int dequant_channel(const Tables &t, const Stream &st, const Channel &ch, const int16_t *__restrict__ quant, float *__restrict__ coef) {
    const Ics ics = ch.ics;
    const int stride = band_stride(ics), max_sfb = ics.max_sfb;
    const uint16_t *off;
    int bands;
    EC_TRY(layout(t, st, ics, &off, &bands));
    for (int i = 0; i < 1024; ++i) coef[i] = 0.0f;
    if (ics.sequence == SEQ_EIGHT_SHORT) {
        int w0 = 0;
        for (int g = 0; g < ics.num_groups; ++g) {
            const int glen = ics.group_len[g];
            if (glen == 0 || w0 + glen > 8) return EC_INVALID_BITSTREAM;
            for (int sfb = 0; sfb < max_sfb; ++sfb) {
                const int book = ch.book[g * stride + sfb];
                if (!(book >= 1 && book <= 11)) continue;
                int s, e;
                EC_TRY(band_range(off, bands, sfb, &s, &e));
                if (e > 128) return EC_INVALID_CONFIG;
                const float scale = ch.mult[g * stride + sfb];
                for (int w = w0; w < w0 + glen; ++w)
                    for (int i = s; i < e; ++i) coef[w * 128 + i] = dequantize(t, quant[w * 128 + i], scale);
            }
            w0 += glen;
        }
        return EC_OK;
    }
    for (int sfb = 0; sfb < max_sfb; ++sfb) {
        const int book = ch.book[sfb];
        if (!(book >= 1 && book <= 11)) continue;
        int s, e;
        EC_TRY(band_range(off, bands, sfb, &s, &e));
        if (e > 1024) return EC_INVALID_CONFIG;
        const float scale = ch.mult[sfb];
        for (int i = s; i < e; ++i) coef[i] = dequantize(t, quant[i], scale);
    }
    return EC_OK;
}

// The values the i16 array could not hold (WireUnit::wide_*), dequantised into their lines with their band's multiplier:
// what dequant_channel's loop would have stored had the integer fitted.  A position whose line is not the marker, or
// lies outside a band coded with books 1..11, is a record no parse produced.
SKE int apply_wide(const WireUnit &w, uint32_t channels, const Scratch &s, const Tables &t, const Stream &st, const int16_t *__restrict__ quant,
                   float *__restrict__ coef) {
    if (w.n_wide > (uint16_t)kWideMax) return EC_INVALID_CONFIG;
    for (uint32_t n = 0; n < w.n_wide; ++n) {
        const uint32_t pos = w.wide_pos[n], c = pos >> 10, line = pos & 1023u;
        if (c >= channels || quant[pos] != kWideMarker) return EC_INVALID_CONFIG;
        const Channel &ch = s.ch[c];
        const Ics ics = ch.ics;
        const uint16_t *off;
        int bands;
        EC_TRY(layout(t, st, ics, &off, &bands));
        const bool is_short = ics.sequence == SEQ_EIGHT_SHORT;
        const int window = is_short ? (int)(line >> 7) : 0, within = is_short ? (int)(line & 127u) : (int)line;
        int group = 0;
        if (is_short) {
            int w0 = 0;
            group = -1;
            for (int g = 0; g < ics.num_groups; ++g) {
                if (window >= w0 && window < w0 + ics.group_len[g]) group = g;
                w0 += ics.group_len[g];
            }
            if (group < 0) return EC_INVALID_CONFIG;
        }
        int band = -1;
        for (int sfb = 0; sfb < ics.max_sfb && sfb < bands; ++sfb)
            if (within >= off[sfb] && within < off[sfb + 1]) band = sfb;
        if (band < 0) return EC_INVALID_CONFIG;
        const int slot = group * band_stride(ics) + band;
        if (!(ch.book[slot] >= 1 && ch.book[slot] <= 11)) return EC_INVALID_CONFIG;
        coef[pos] = dequantize(t, w.wide_val[n], ch.mult[slot]);
    }
    return EC_OK;
}

}  // namespace sk_ec
