// aac_frontend.cpp -- AAC-LC raw-access-unit front-end (host side of the boundary).
//
// Restates the entropy/side-information part of soundkit-aac-lc's AacLcDecoder::decode_access_unit
// (decoder.rs:104-334) up to, but not including, synthesis: element loop, ICS info, section data,
// scalefactors, pulse, TNS, Huffman spectral decode with fused dequantisation, PNS, intensity and
// mid/side stereo, TNS filtering.  Its output -- dequantised spectra [channels][1024] and the
// per-channel window sequence/shape -- is exactly what the GPU synthesis (sk_aac_synthesize_*) consumes.
// Bit-serial work stays on host cores (SURVEY: "stays on host cores"); nothing here touches the GPU.
//
// Follows, function by function: bitreader.rs, syntax.rs:54-63, config.rs:121-319, ics.rs:57-110,
// section.rs:60-120, scalefactor.rs:80-205, pulse.rs:20-35, tns.rs:34-276, spectral.rs:191-230,
// 327-423, 1907-2109, 2198-2294, 2408-2460, stereo.rs:114-448, sfb.rs:52-152, decoder.rs:104-438.
// Quirks kept on purpose (parity with the reference, not with the ISO text): section codebook 14 is the
// in-phase intensity codebook and 15 the out-of-phase one (stereo.rs:431-437), a fill element with
// esc_count 0 is rejected (decoder.rs:396-399), PNS uses the LCG of spectral.rs:2447-2450 seeded once
// per decoder.
#include "../../include/soundkit_amd.h"
#include "sk_abi.h"

#include <cmath>
#include <cstring>
#include <mutex>
#include <new>
#include <string>
#include <vector>

#include "aac_entropy_tables.h"
#include "aac_entropy_core.h"
#include "aac_tables.h"

namespace {

using namespace sk_aac_tables;

struct AacError {
    int code;
    std::string msg;
};

[[noreturn]] void fail(int code, const std::string &msg) { throw AacError{code, msg}; }

// ---- bitreader.rs ---------------------------------------------------------------------------------
struct BitReader {
    // data must be readable (zero-padded) for 8 bytes past len: peeks are single unaligned 64-bit loads
    const uint8_t *data;
    size_t total_bits, pos = 0;
    BitReader(const uint8_t *padded, size_t len) : data(padded), total_bits(len * 8) {}
    size_t remaining() const { return total_bits - pos; }
    uint32_t peek32() const {  // next 32 bits, left-aligned (bits past the end read as the zero padding)
        uint64_t w;
        std::memcpy(&w, data + (pos >> 3), 8);
        w = __builtin_bswap64(w);
        return (uint32_t)((w << (pos & 7)) >> 32);
    }
    uint32_t peek(unsigned bits) const { return bits ? peek32() >> (32 - bits) : 0; }  // bits <= 32
    [[noreturn]] void eof(size_t bits) const {
        fail(SK_AAC_ERR_EOF, "unexpected end of AAC bitstream: requested " + std::to_string(bits > 255 ? 255 : bits) +
                                 " bits, " + std::to_string(remaining()) + " bits remain");
    }
    uint32_t read(unsigned bits) {
        if (__builtin_expect(remaining() < bits, 0)) eof(bits);
        const uint32_t v = peek(bits);
        pos += bits;
        return v;
    }
    bool read_bool() { return read(1) != 0; }
    void skip(size_t bits) {
        if (remaining() < bits) eof(bits);
        pos += bits;
    }
};

// ---- Huffman lookup tables (built once from the ISO codebooks) ---------------------------------------
struct Lut {
    // two-level: the first kPrimary bits index `table`; an entry is (len << 16) | symbol for codes that fit, or
    // 0x80000000 | (extra_bits << 24) | offset of a sub-table indexed by the following extra_bits bits; 0 = invalid
    static constexpr int kPrimary = 9;
    int max_bits = 0;
    std::vector<uint32_t> table;
    void build(const uint8_t *lens, const uint32_t *codes32, const uint16_t *codes16, int n, int maxb) {
        max_bits = maxb;
        const int pb = maxb < kPrimary ? maxb : kPrimary;
        primary_bits = pb;
        table.assign((size_t)1 << pb, 0);
        std::vector<int> deepest((size_t)1 << pb, 0);
        for (int i = 0; i < n; ++i)
            if (lens[i] > pb) {
                const uint32_t code = codes32 ? codes32[i] : codes16[i];
                int &d = deepest[code >> (lens[i] - pb)];
                if (lens[i] - pb > d) d = lens[i] - pb;
            }
        for (size_t p = 0; p < deepest.size(); ++p)
            if (deepest[p]) {
                table[p] = 0x80000000u | ((uint32_t)deepest[p] << 24) | (uint32_t)table.size();
                table.resize(table.size() + ((size_t)1 << deepest[p]), 0);
            }
        for (int i = 0; i < n; ++i) {
            const int len = lens[i];
            if (!len) continue;
            const uint32_t code = codes32 ? codes32[i] : codes16[i];
            const uint32_t e = ((uint32_t)len << 16) | (uint32_t)i;
            if (len <= pb) {
                const size_t prefix = (size_t)code << (pb - len), slots = (size_t)1 << (pb - len);
                for (size_t k = 0; k < slots; ++k) table[prefix + k] = e;
            } else {
                const uint32_t link = table[code >> (len - pb)];
                const int extra = (int)((link >> 24) & 0x7f), rest = len - pb;
                const size_t base = link & 0xffffff;
                const size_t prefix = (size_t)(code & ((1u << rest) - 1)) << (extra - rest), slots = (size_t)1 << (extra - rest);
                for (size_t k = 0; k < slots; ++k) table[base + prefix + k] = e;
            }
        }
    }
    // scalefactor.rs:252-266 / spectral.rs read_*_tuple: peek what is there, the entry must fit in it
    int read(BitReader &r, const char *what) const {
        const uint32_t look = r.peek32();
        uint32_t e = table[look >> (32 - primary_bits)];
        if (e & 0x80000000u) {
            const unsigned extra = (e >> 24) & 0x7f;
            e = table[(e & 0xffffff) + ((look << primary_bits) >> (32 - extra))];
        }
        const unsigned len = e >> 16;
        if (__builtin_expect(len == 0 || len > r.remaining(), 0)) fail(SK_AAC_ERR_INVALID_BITSTREAM, what);
        r.pos += len;
        return (int)(e & 0xffff);
    }
    int primary_bits = 0;
};

// the values a spectral codeword stands for, unpacked once (spectral.rs tuple readers: idx -> digits of base 3 / 9 / 8 / 13 / 17)
struct Tuple {
    float mag[4];        // |v|^(4/3) from the reference's table (0 for v = 0)
    uint32_t sign[4];    // signed books: 0x80000000 where v < 0
    uint8_t sshift[4];   // unsigned books: which of the sign bits that follow the codeword belongs to value k (31: none)
    int8_t v[4];         // signed books: the values; unsigned books: the magnitudes
    uint8_t nsign;       // unsigned books: how many magnitudes are non-zero (= sign bits that follow the codeword)
    uint8_t escape;      // book 11: a magnitude of 16 (an escape sequence follows the sign bits)
    uint8_t pad[2];
};

struct Tables {
    Lut sf, cb[12];
    Tuple tuples[12][289];
    float pow43[8192];      // dsp.rs:420-429
    float sf_mult[768];     // dsp.rs:439-450, scale factors -256..511
    Tables() {
        sf.build(kSfLen, kSfCode, nullptr, 121, 19);
        cb[1].build(kCb1Len, nullptr, kCb1Code, 81, 11);
        cb[2].build(kCb2Len, nullptr, kCb2Code, 81, 9);
        cb[3].build(kCb3Len, nullptr, kCb3Code, 81, 16);
        cb[4].build(kCb4Len, nullptr, kCb4Code, 81, 12);
        cb[5].build(kCb5Len, nullptr, kCb5Code, 81, 13);
        cb[6].build(kCb6Len, nullptr, kCb6Code, 81, 11);
        cb[7].build(kCb7Len, nullptr, kCb7Code, 64, 12);
        cb[8].build(kCb8Len, nullptr, kCb8Code, 64, 10);
        cb[9].build(kCb9Len, nullptr, kCb9Code, 169, 15);
        cb[10].build(kCb10Len, nullptr, kCb10Code, 169, 12);
        cb[11].build(kCb11Len, nullptr, kCb11Code, 289, 12);
        for (int v = 0; v < 8192; ++v) pow43[v] = std::pow((float)v, 4.0f / 3.0f);
        for (int s = -256; s <= 511; ++s) sf_mult[s + 256] = std::pow(2.0f, ((float)s - 100.0f) * 0.25f);
        static const int kSymbols[12] = {0, 81, 81, 81, 81, 81, 81, 64, 64, 169, 169, 289};
        for (int book = 1; book <= 11; ++book)
            for (int idx = 0; idx < kSymbols[book]; ++idx) {
                Tuple &t = tuples[book][idx];
                t = Tuple{};
                if (book <= 4) {
                    const int d[4] = {idx / 27, (idx / 9) % 3, (idx / 3) % 3, idx % 3};
                    for (int k = 0; k < 4; ++k) t.v[k] = (int8_t)(book <= 2 ? d[k] - 1 : d[k]);
                } else {
                    const int dim = book <= 6 ? 9 : (book <= 8 ? 8 : (book <= 10 ? 13 : 17));
                    const int d[2] = {idx / dim, idx % dim};
                    for (int k = 0; k < 2; ++k) t.v[k] = (int8_t)(book <= 6 ? d[k] - 4 : d[k]);
                }
                const bool is_unsigned = book == 3 || book == 4 || book >= 7;
                if (is_unsigned)
                    for (int k = 0; k < 4; ++k) t.nsign += t.v[k] != 0;
                int seen = 0;
                for (int k = 0; k < 4; ++k) {
                    const int v = t.v[k];
                    t.mag[k] = pow43[v < 0 ? -v : v];
                    t.sign[k] = (!is_unsigned && v < 0) ? 0x80000000u : 0u;
                    t.sshift[k] = (is_unsigned && v != 0) ? (uint8_t)(t.nsign - 1 - seen++) : 31;
                    t.escape |= book == 11 && v == 16;
                }
            }
    }
};
const Tables &tables() {
    static const Tables t;
    return t;
}

float scalefactor_multiplier(int sf) {  // dsp.rs:407-413
    if (sf >= -256 && sf <= 511) return tables().sf_mult[sf + 256];
    return std::pow(2.0f, ((float)sf - 100.0f) * 0.25f);
}
float dequantize(int q, float scale) {  // dsp.rs:397-405
    if (q == 0) return 0.0f;
    const float sign = q < 0 ? -1.0f : 1.0f;
    const unsigned mag = q < 0 ? (unsigned)(-(int64_t)q) : (unsigned)q;
    const float m = mag < 8192 ? tables().pow43[mag] : std::pow((float)mag, 4.0f / 3.0f);
    return sign * m * scale;
}

// ---- sfb.rs ---------------------------------------------------------------------------------------
struct Layout {
    const uint16_t *off;
    int bands;  // offsets has bands + 1 entries
};
Layout long_layout(int sf_index) {  // sfb.rs:52-62
    switch (sf_index) {
    case 0: case 1: return {kSwb1024_96, 41};
    case 2: return {kSwb1024_64, 47};
    case 3: case 4: return {kSwb1024_48, 49};
    case 5: return {kSwb1024_32, 51};
    case 6: case 7: return {kSwb1024_24, 47};
    case 8: case 9: case 10: return {kSwb1024_16, 43};
    case 11: case 12: return {kSwb1024_8, 40};
    default: fail(SK_AAC_ERR_UNSUPPORTED_SF_INDEX, "unsupported AAC sampling frequency index " + std::to_string(sf_index));
    }
}
Layout short_layout(int sf_index) {  // sfb.rs:64-71
    switch (sf_index) {
    case 0: case 1: case 2: return {kSwb128_96, 12};
    case 3: case 4: case 5: return {kSwb128_48, 14};
    case 6: case 7: return {kSwb128_24, 15};
    case 8: case 9: case 10: return {kSwb128_16, 15};
    case 11: case 12: return {kSwb128_8, 15};
    default: fail(SK_AAC_ERR_UNSUPPORTED_SF_INDEX, "unsupported AAC sampling frequency index " + std::to_string(sf_index));
    }
}
void band_range(const Layout &l, int sfb, int *start, int *end) {  // spectral.rs:92-105
    if (sfb < 0 || sfb + 1 > l.bands) fail(SK_AAC_ERR_INVALID_CONFIG, "missing scale-factor band offset");
    *start = l.off[sfb];
    *end = l.off[sfb + 1];
}

// ---- side information -----------------------------------------------------------------------------
enum { CB_ZERO = 0, CB_NOISE = 13, CB_INTENSITY = 14, CB_INTENSITY_NEG = 15 };

struct IcsInfo {  // ics.rs:46-54
    int window_sequence = 0, window_shape = 0, max_sfb = 0, num_windows = 1, num_window_groups = 1;
    int window_group_len[8] = {1, 0, 0, 0, 0, 0, 0, 0};
};

IcsInfo read_ics_info(BitReader &r) {  // ics.rs:57-110
    if (r.read_bool()) fail(SK_AAC_ERR_INVALID_CONFIG, "ICS reserved bit is set");
    IcsInfo info;
    info.window_sequence = (int)r.read(2);
    info.window_shape = (int)r.read(1);
    if (info.window_sequence == SK_EIGHT_SHORT) {
        info.max_sfb = (int)r.read(4);
        const uint32_t grouping = r.read(7);
        int group = 0;
        info.window_group_len[0] = 1;
        for (int bit = 0; bit < 7; ++bit) {
            if ((grouping >> (6 - bit)) & 1) {
                info.window_group_len[group] += 1;
            } else {
                group += 1;
                info.window_group_len[group] = 1;
            }
        }
        info.num_windows = 8;
        info.num_window_groups = group + 1;
    } else {
        info.max_sfb = (int)r.read(6);
        if (r.read_bool()) fail(SK_AAC_ERR_UNSUPPORTED_FEATURE, "AAC prediction");
    }
    return info;
}

struct Pulse {  // pulse.rs
    bool present = false;
    int start_sfb = 0, count = 0, offset[4] = {0, 0, 0, 0}, amp[4] = {0, 0, 0, 0};
};

struct TnsFilter {  // tns.rs:10-18
    int length = 0, order = 0, coef_bits = 0;
    bool direction = false;
    int8_t coeffs[20] = {0};
};
struct TnsWindow {
    int filter_count = 0;
    bool coef_res = false;
    TnsFilter filters[4];
};
struct Tns {
    bool present = false;
    int window_count = 0;
    TnsWindow windows[8];
};

struct Channel {  // IndividualChannelStream (channel.rs:36-79)
    int global_gain = 0;
    IcsInfo info;
    uint8_t cb[8][64];       // section.rs: codebook per (group, sfb)
    float mult[8][64];       // scalefactor.rs: multiplier per band
    Pulse pulse;
    Tns tns;
};

void read_sections(BitReader &r, Channel &ch) {  // section.rs:60-120
    const IcsInfo &info = ch.info;
    if (info.max_sfb > 64) fail(SK_AAC_ERR_INVALID_BITSTREAM, "max_sfb exceeds parser capacity");
    const unsigned len_bits = info.window_sequence == SK_EIGHT_SHORT ? 3 : 5;
    const unsigned esc = (1u << len_bits) - 1;
    std::memset(ch.cb, 0, sizeof(ch.cb));
    for (int g = 0; g < info.num_window_groups; ++g) {
        int sfb = 0;
        while (sfb < info.max_sfb) {
            const unsigned cb = r.read(4);
            if (cb == 12) fail(SK_AAC_ERR_INVALID_BITSTREAM, "reserved AAC section codebook");
            int len = 0;
            for (;;) {
                const unsigned incr = r.read(len_bits);
                len += (int)incr;
                if (incr != esc) break;
            }
            if (len == 0) fail(SK_AAC_ERR_INVALID_BITSTREAM, "zero-length section");
            if (sfb + len > info.max_sfb) fail(SK_AAC_ERR_INVALID_BITSTREAM, "section length exceeds max_sfb");
            for (int b = sfb; b < sfb + len; ++b) ch.cb[g][b] = (uint8_t)cb;
            sfb += len;
        }
    }
}

int checked_i16_add(int a, int b, const char *what) {
    const int s = a + b;
    if (s < -32768 || s > 32767) fail(SK_AAC_ERR_INVALID_BITSTREAM, what);
    return s;
}

void read_scalefactors(BitReader &r, Channel &ch) {  // scalefactor.rs:80-153
    const Tables &t = tables();
    int spectral = ch.global_gain, noise = ch.global_gain - 90, intensity = 0;
    bool first_noise = true;
    std::memset(ch.mult, 0, sizeof(ch.mult));
    for (int g = 0; g < ch.info.num_window_groups; ++g) {
        for (int sfb = 0; sfb < ch.info.max_sfb; ++sfb) {
            const int cb = ch.cb[g][sfb];
            if (cb == CB_ZERO) continue;
            if (cb == CB_NOISE) {
                if (first_noise) {
                    noise = checked_i16_add(noise, (int)r.read(9) - 256, "noise scalefactor overflow");  // :204-206
                    first_noise = false;
                } else {
                    noise = checked_i16_add(noise, t.sf.read(r, "invalid AAC scalefactor codeword") - 60,
                                            "noise scalefactor overflow");
                }
                ch.mult[g][sfb] = scalefactor_multiplier(noise);
            } else if (cb == CB_INTENSITY || cb == CB_INTENSITY_NEG) {
                intensity = checked_i16_add(intensity, t.sf.read(r, "invalid AAC scalefactor codeword") - 60,
                                            "intensity scalefactor overflow");
                ch.mult[g][sfb] = std::pow(2.0f, -0.25f * (float)intensity);  // :208-210
            } else {
                spectral = checked_i16_add(spectral, t.sf.read(r, "invalid AAC scalefactor codeword") - 60,
                                           "spectral scalefactor overflow");
                ch.mult[g][sfb] = scalefactor_multiplier(spectral);
            }
        }
    }
}

void read_pulse(BitReader &r, Pulse &p) {  // pulse.rs:20-35
    p.present = true;
    p.count = (int)r.read(2) + 1;
    p.start_sfb = (int)r.read(6);
    for (int i = 0; i < p.count; ++i) {
        p.offset[i] = (int)r.read(5);
        p.amp[i] = (int)r.read(4);
    }
}

void read_tns(BitReader &r, const IcsInfo &info, Tns &tns) {  // tns.rs:34-83
    const bool is_short = info.window_sequence == SK_EIGHT_SHORT;
    tns = Tns();
    tns.present = true;
    tns.window_count = info.num_windows;
    const unsigned n_filt_bits = is_short ? 1 : 2, length_bits = is_short ? 4 : 6, order_bits = is_short ? 3 : 5;
    for (int w = 0; w < tns.window_count; ++w) {
        TnsWindow &win = tns.windows[w];
        win.filter_count = (int)r.read(n_filt_bits);
        if (win.filter_count > 4) fail(SK_AAC_ERR_INVALID_BITSTREAM, "too many TNS filters");
        if (win.filter_count == 0) continue;
        win.coef_res = r.read_bool();
        for (int f = 0; f < win.filter_count; ++f) {
            TnsFilter &flt = win.filters[f];
            flt.length = (int)r.read(length_bits);
            flt.order = (int)r.read(order_bits);
            if (flt.order > 20) fail(SK_AAC_ERR_UNSUPPORTED_FEATURE, "TNS order above 20");
            if (flt.order == 0) continue;
            flt.direction = r.read_bool();
            const bool compress = r.read_bool();
            flt.coef_bits = (win.coef_res ? 4 : 3) - (compress ? 1 : 0);
            for (int i = 0; i < flt.order; ++i) {  // read_signed, tns.rs:278-282
                const unsigned raw = r.read((unsigned)flt.coef_bits);
                const int shift = 8 - flt.coef_bits;
                flt.coeffs[i] = (int8_t)((int8_t)(uint8_t)(raw << shift) >> shift);
            }
        }
    }
}

void read_channel(BitReader &r, Channel &ch, const IcsInfo *common) {  // channel.rs:19-75
    ch.global_gain = (int)r.read(8);
    ch.info = common ? *common : read_ics_info(r);
    read_sections(r, ch);
    read_scalefactors(r, ch);
    ch.pulse = Pulse();
    if (r.read_bool()) read_pulse(r, ch.pulse);
    ch.tns = Tns();
    if (r.read_bool()) read_tns(r, ch.info, ch.tns);
    if (r.read_bool()) fail(SK_AAC_ERR_UNSUPPORTED_FEATURE, "gain control");
}

// ---- spectral data --------------------------------------------------------------------------------
int read_escape(BitReader &r) {  // spectral.rs:214-230
    unsigned extra = 4;
    while (r.read_bool()) {
        extra += 1;
        if (extra > 16) fail(SK_AAC_ERR_UNSUPPORTED_FEATURE, "AAC escape value above 16 extra bits");
    }
    return (1 << extra) + (int)r.read(extra);
}

// decodes the quantised values of one band segment (spectral.rs:327-423 dispatch + tuple readers)
void read_band_quantized(BitReader &r, int cb, int *q, int n) {
    const Tables &t = tables();
    const char *bad = "invalid AAC spectral codeword";
    if (cb <= 4) {
        for (int i = 0; i + 4 <= n; i += 4) {
            const int idx = t.cb[cb].read(r, bad);
            int v[4] = {idx / 27, (idx / 9) % 3, (idx / 3) % 3, idx % 3};
            if (cb <= 2) {
                for (int k = 0; k < 4; ++k) v[k] -= 1;
            } else {
                for (int k = 0; k < 4; ++k)  // apply_unsigned_signs, spectral.rs:714-721
                    if (v[k] != 0 && r.read_bool()) v[k] = -v[k];
            }
            for (int k = 0; k < 4; ++k) q[i + k] = v[k];
        }
    } else {
        const int dim = cb <= 6 ? 9 : (cb <= 8 ? 8 : (cb <= 10 ? 13 : 17));
        for (int i = 0; i + 2 <= n; i += 2) {
            const int idx = t.cb[cb].read(r, bad);
            int v[2] = {idx / dim, idx % dim};
            if (cb <= 6) {
                v[0] -= 4;
                v[1] -= 4;
            } else if (cb <= 10) {
                for (int k = 0; k < 2; ++k)
                    if (v[k] != 0 && r.read_bool()) v[k] = -v[k];
            } else {  // finish_unsigned_escape_pair, spectral.rs:191-212: signs first, then escapes
                bool sign[2] = {false, false};
                for (int k = 0; k < 2; ++k)
                    if (v[k] != 0) sign[k] = r.read_bool();
                for (int k = 0; k < 2; ++k)
                    if (v[k] == 16) v[k] = read_escape(r);
                for (int k = 0; k < 2; ++k)
                    if (sign[k]) v[k] = -v[k];
            }
            q[i] = v[0];
            q[i + 1] = v[1];
        }
    }
}

// The same band, dequantised on the way (the reference fuses this too: write_scaled_tuple, spectral.rs:723-727).
// Codeword, sign bits and values come out of one 64-bit window per tuple, without data-dependent branches; anything
// unusual (fewer than 32 bits left, an invalid codeword, an escape, a non-finite scale) goes through the bit-exact
// slow readers above so that errors and their messages are the reference's.
template <int DIM, bool UNSIGNED>
void read_band_scaled_t(const Tables &t, BitReader &r, int cb, float scale, float *out, int n) {
    const Lut &lut = t.cb[cb];
    const Tuple *tuples = t.tuples[cb];
    const int pb = lut.primary_bits;
    const uint32_t *table = lut.table.data();
    const uint8_t *data = r.data;
    const size_t total = r.total_bits;
    size_t pos = r.pos;
    for (int i = 0; i + DIM <= n; i += DIM) {
        if (__builtin_expect(total - pos >= 32, 1)) {
            uint64_t w;
            std::memcpy(&w, data + (pos >> 3), 8);
            w = __builtin_bswap64(w) << (pos & 7);  // at least 57 valid bits at the top
            const uint32_t look = (uint32_t)(w >> 32);
            uint32_t e = table[look >> (32 - pb)];
            if (e & 0x80000000u) {
                const unsigned extra = (e >> 24) & 0x7f;
                e = table[(e & 0xffffff) + ((look << pb) >> (32 - extra))];
            }
            const unsigned len = e >> 16;
            if (__builtin_expect(len != 0, 1)) {
                const Tuple &tu = tuples[e & 0xffff];
                uint32_t signs = 0;
                unsigned used = len;
                if (UNSIGNED) {
                    signs = (uint32_t)((w << len) >> 60) >> (4 - tu.nsign);  // the nsign (<= 4) bits after the codeword
                    used += tu.nsign;
                }
                for (int k = 0; k < DIM; ++k) {
                    uint32_t bits;
                    std::memcpy(&bits, &tu.mag[k], 4);
                    bits ^= tu.sign[k] ^ ((signs >> tu.sshift[k]) << 31);
                    float m;
                    std::memcpy(&m, &bits, 4);
                    out[i + k] = m * scale;
                }
                pos += used;
                if (__builtin_expect(tu.escape, 0)) {  // finish_unsigned_escape_pair (spectral.rs:191-212): escapes follow the signs
                    r.pos = pos;
                    for (int k = 0; k < 2; ++k) {
                        if (tu.v[k] != 16) continue;
                        const bool neg = (signs >> tu.sshift[k]) & 1;
                        const int esc = read_escape(r);
                        out[i + k] = dequantize(neg ? -esc : esc, scale);
                    }
                    pos = r.pos;
                }
                continue;
            }
        }
        int q[4];
        r.pos = pos;
        read_band_quantized(r, cb, q, DIM);
        for (int k = 0; k < DIM; ++k) out[i + k] = dequantize(q[k], scale);
        pos = r.pos;
    }
    r.pos = pos;
}

void read_band_scaled(const Tables &t, BitReader &r, int cb, float scale, float *out, int n) {
    if (__builtin_expect(!std::isfinite(scale), 0)) {  // 0 * inf must stay 0 (dsp.rs:397-399): exact path
        for (int i = 0; i + (cb <= 4 ? 4 : 2) <= n; i += (cb <= 4 ? 4 : 2)) {
            int q[4];
            read_band_quantized(r, cb, q, cb <= 4 ? 4 : 2);
            for (int k = 0; k < (cb <= 4 ? 4 : 2); ++k) out[i + k] = dequantize(q[k], scale);
        }
        return;
    }
    switch (cb) {
    case 1: case 2: read_band_scaled_t<4, false>(t, r, cb, scale, out, n); break;
    case 3: case 4: read_band_scaled_t<4, true>(t, r, cb, scale, out, n); break;
    case 5: case 6: read_band_scaled_t<2, false>(t, r, cb, scale, out, n); break;
    default: read_band_scaled_t<2, true>(t, r, cb, scale, out, n); break;
    }
}

struct Decoder {
    uint32_t sample_rate = 0;
    int sf_index = -1;  // -1: explicit rate (no band tables)
    int channels = 0;
    uint32_t pns_state = 0x1f2e3d4cu;  // spectral.rs:2459, decoder.rs:76
    std::string last_error;
    std::vector<uint8_t> padded;  // the access unit + 8 zero bytes (BitReader's load window)
    int quantized[1024];
    // tool usage since creation (aac-wasm-bench/src/lib.rs:1955-1986 asserts this coverage on its fixture)
    uint32_t n_frames = 0, n_short = 0, n_tns = 0, n_pns_bands = 0, n_is_bands = 0, n_ms_bands = 0, n_pulse = 0, n_transition = 0;
};

float next_noise(uint32_t &state) {  // spectral.rs:2447-2450
    state = state * 1664525u + 1013904223u;
    return (float)(int16_t)((int32_t)state >> 16);
}

void synthesize_noise_band(float scale, uint32_t &state, float *out, int n) {  // spectral.rs:2416-2445
    if (n == 0) return;
    float energy = 0.0f;
    for (int i = 0; i < n; ++i) {
        const float v = next_noise(state);
        out[i] = v;
        energy += v * v;
    }
    if (energy <= 1.1920929e-07f) fail(SK_AAC_ERR_INVALID_BITSTREAM, "PNS noise band has zero energy");
    const float normalizer = scale / std::sqrt(energy);
    for (int i = 0; i < n; ++i) out[i] *= normalizer;
}

Layout layout_for(const Decoder &d, const IcsInfo &info) {  // decoder.rs:376-383
    if (d.sf_index < 0) fail(SK_AAC_ERR_UNSUPPORTED_FEATURE, "explicit sample-rate scalefactor bands");
    return info.window_sequence == SK_EIGHT_SHORT ? short_layout(d.sf_index) : long_layout(d.sf_index);
}

// decoder.rs:220-244 + spectral.rs:1907-2109, 2198-2294
void decode_spectrum(Decoder &d, BitReader &r, const Channel &ch, bool allow_intensity, float *coef) {
    const IcsInfo &info = ch.info;
    if (!allow_intensity)
        for (int g = 0; g < info.num_window_groups; ++g)
            for (int sfb = 0; sfb < info.max_sfb; ++sfb)
                if (ch.cb[g][sfb] == CB_INTENSITY || ch.cb[g][sfb] == CB_INTENSITY_NEG)
                    fail(SK_AAC_ERR_INVALID_BITSTREAM, "intensity stereo is only valid in the right channel of a channel pair");
    const Layout lay = layout_for(d, info);
    const Tables &t = tables();
    std::memset(coef, 0, sizeof(float) * 1024);
    if (info.window_sequence == SK_EIGHT_SHORT) {
        if (ch.pulse.present) fail(SK_AAC_ERR_INVALID_BITSTREAM, "pulse data is not allowed for short windows");
        int window_start = 0;
        for (int g = 0; g < info.num_window_groups; ++g) {
            const int glen = info.window_group_len[g];
            if (glen == 0) fail(SK_AAC_ERR_INVALID_BITSTREAM, "short-window group has zero length");
            if (window_start + glen > 8) fail(SK_AAC_ERR_INVALID_BITSTREAM, "short-window groups exceed eight windows");
            for (int sfb = 0; sfb < info.max_sfb; ++sfb) {
                int s, e;
                band_range(lay, sfb, &s, &e);
                if (e > 128) fail(SK_AAC_ERR_INVALID_CONFIG, "short scale-factor band exceeds window length");
                const int cb = ch.cb[g][sfb];
                if (cb >= 1 && cb <= 11) {
                    for (int w = window_start; w < window_start + glen; ++w)
                        read_band_scaled(t, r, cb, ch.mult[g][sfb], coef + w * 128 + s, e - s);
                } else if (cb == CB_NOISE) {
                    d.n_pns_bands += 1;
                    for (int w = window_start; w < window_start + glen; ++w)
                        synthesize_noise_band(ch.mult[g][sfb], d.pns_state, coef + w * 128 + s, e - s);
                }
            }
            window_start += glen;
        }
        if (window_start != 8) fail(SK_AAC_ERR_INVALID_BITSTREAM, "short-window groups do not cover eight windows");
        return;
    }
    if (!ch.pulse.present) {  // decode_long_standard
        for (int sfb = 0; sfb < info.max_sfb; ++sfb) {
            int s, e;
            band_range(lay, sfb, &s, &e);
            if (e > 1024) fail(SK_AAC_ERR_INVALID_CONFIG, "scale-factor band exceeds coefficient buffer");
            const int cb = ch.cb[0][sfb];
            if (cb >= 1 && cb <= 11) {
                read_band_scaled(t, r, cb, ch.mult[0][sfb], coef + s, e - s);
            } else if (cb == CB_NOISE) {
                d.n_pns_bands += 1;
                synthesize_noise_band(ch.mult[0][sfb], d.pns_state, coef + s, e - s);
            }
        }
        return;
    }
    // decode_long_standard_with_pulse: read everything quantised, add the pulses, then dequantise
    std::memset(d.quantized, 0, sizeof(d.quantized));
    for (int sfb = 0; sfb < info.max_sfb; ++sfb) {
        int s, e;
        band_range(lay, sfb, &s, &e);
        if (e > 1024) fail(SK_AAC_ERR_INVALID_CONFIG, "scale-factor band exceeds coefficient buffer");
        const int cb = ch.cb[0][sfb];
        if (cb >= 1 && cb <= 11) read_band_quantized(r, cb, d.quantized + s, e - s);
    }
    {  // apply_pulse_data, spectral.rs:2198-2247
        const Pulse &p = ch.pulse;
        if (p.start_sfb >= info.max_sfb) fail(SK_AAC_ERR_INVALID_BITSTREAM, "pulse start scale-factor band exceeds max_sfb");
        int s0, e0;
        band_range(lay, p.start_sfb, &s0, &e0);
        int index = s0;
        for (int i = 0; i < p.count; ++i) {
            index += p.offset[i];
            if (index >= 1024) fail(SK_AAC_ERR_INVALID_BITSTREAM, "pulse target exceeds spectral coefficient buffer");
            int band = -1;
            for (int sfb = 0; sfb < info.max_sfb; ++sfb) {
                int s, e;
                band_range(lay, sfb, &s, &e);
                if (index >= s && index < e) { band = sfb; break; }
            }
            if (band < 0) fail(SK_AAC_ERR_INVALID_BITSTREAM, "pulse target exceeds coded scale-factor bands");
            const int cb = ch.cb[0][band];
            if (!(cb >= 1 && cb <= 11)) fail(SK_AAC_ERR_INVALID_BITSTREAM, "pulse target is not in a spectral band");
            if (d.quantized[index] > 0) d.quantized[index] += p.amp[i];
            else d.quantized[index] -= p.amp[i];
        }
    }
    for (int sfb = 0; sfb < info.max_sfb; ++sfb) {  // dequantize_long
        int s, e;
        band_range(lay, sfb, &s, &e);
        const int cb = ch.cb[0][sfb];
        if (cb >= 1 && cb <= 11) {
            for (int i = s; i < e; ++i) coef[i] = dequantize(d.quantized[i], ch.mult[0][sfb]);
        } else if (cb == CB_NOISE) {
            d.n_pns_bands += 1;
            synthesize_noise_band(ch.mult[0][sfb], d.pns_state, coef + s, e - s);
        }
    }
}

// ---- stereo tools (stereo.rs) -----------------------------------------------------------------------
struct MsMask {
    int mode = 0;  // 0 none, 1 per band, 2 all
    bool used[8][64];
};

void read_ms_mask(BitReader &r, const IcsInfo &info, MsMask &m) {  // channel.rs:222-251
    m.mode = (int)r.read(2);
    if (m.mode == 3) fail(SK_AAC_ERR_INVALID_BITSTREAM, "reserved mid/side mask mode");
    if (m.mode == 1) {
        if (info.max_sfb > 64) fail(SK_AAC_ERR_INVALID_BITSTREAM, "max_sfb exceeds parser capacity");
        std::memset(m.used, 0, sizeof(m.used));
        for (int g = 0; g < info.num_window_groups; ++g)
            for (int sfb = 0; sfb < info.max_sfb; ++sfb) m.used[g][sfb] = r.read_bool();
    }
}

bool ms_selected(const MsMask &m, int g, int sfb) { return m.mode == 2 || (m.mode == 1 && m.used[g][sfb]); }

void apply_stereo_tools(Decoder &d, const MsMask &mask, const IcsInfo &info, const Channel &lch, const Channel &rch,
                        float *left, float *right) {  // decoder.rs:268-334
    const Layout lay = layout_for(d, info);
    const bool is_short = info.window_sequence == SK_EIGHT_SHORT;
    const int wlen = is_short ? 128 : 1024;
    // intensity first (stereo.rs:114-241), then mid/side excluding intensity-right and noise bands (:44-112, 410-417)
    for (int pass = 0; pass < 2; ++pass) {
        int window_start = 0;
        for (int g = 0; g < info.num_window_groups; ++g) {
            const int glen = is_short ? info.window_group_len[g] : 1;
            if (is_short) {
                if (glen == 0) fail(SK_AAC_ERR_INVALID_BITSTREAM, "short-window group has zero length");
                if (window_start + glen > 8) fail(SK_AAC_ERR_INVALID_BITSTREAM, "short-window groups exceed eight windows");
            }
            for (int sfb = 0; sfb < info.max_sfb; ++sfb) {
                int s, e;
                band_range(lay, sfb, &s, &e);
                if (e > wlen) fail(SK_AAC_ERR_INVALID_CONFIG, "scale-factor band exceeds window length");
                const int rcb = rch.cb[g][sfb], lcb = lch.cb[g][sfb];
                if (pass == 0) {
                    if (rcb != CB_INTENSITY && rcb != CB_INTENSITY_NEG) continue;
                    float sign = rcb == CB_INTENSITY ? 1.0f : -1.0f;  // stereo.rs:431-437
                    if (ms_selected(mask, g, sfb)) sign = -sign;      // stereo.rs:145-149
                    d.n_is_bands += 1;
                    const float scale = rch.mult[g][sfb];
                    for (int w = window_start; w < window_start + glen; ++w)
                        for (int i = w * wlen + s; i < w * wlen + e; ++i) right[i] = left[i] * scale * sign;
                } else {
                    if (!ms_selected(mask, g, sfb)) continue;
                    if (rcb == CB_INTENSITY || rcb == CB_INTENSITY_NEG || lcb == CB_NOISE || rcb == CB_NOISE) continue;
                    d.n_ms_bands += 1;
                    for (int w = window_start; w < window_start + glen; ++w)
                        for (int i = w * wlen + s; i < w * wlen + e; ++i) {
                            const float mid = left[i], side = right[i];
                            left[i] = mid + side;
                            right[i] = mid - side;
                        }
                }
            }
            window_start += glen;
        }
        if (is_short && window_start != 8) fail(SK_AAC_ERR_INVALID_BITSTREAM, "short-window groups do not cover eight windows");
    }
}

// ---- TNS (tns.rs:103-276) ----------------------------------------------------------------------------
float tns_inverse_quant(int encoded, int coef_bits, int coef_res_bits) {  // tns.rs:208-235
    if (coef_bits == 0 || coef_bits > 4 || coef_res_bits < 3 || coef_res_bits > 4)
        fail(SK_AAC_ERR_INVALID_BITSTREAM, "invalid TNS coefficient resolution");
    const int raw = encoded & ((1 << coef_bits) - 1);
    const int boundary = 1 << (coef_bits - 1);
    const int sgn = raw < boundary ? -raw : (1 << coef_bits) - raw;
    if (sgn == 0) return 0.0f;
    const float divisor = (float)(sgn < 0 ? (1 << coef_res_bits) - 1 : (1 << coef_res_bits) + 1);
    return std::sin((float)sgn * 3.14159274101257324219f / divisor);
}

void apply_tns(const Decoder &d, const Channel &ch, float *coef) {  // tns.rs:103-174
    const Tns &tns = ch.tns;
    const IcsInfo &info = ch.info;
    if (tns.window_count != info.num_windows) fail(SK_AAC_ERR_INVALID_BITSTREAM, "TNS window count does not match ICS");
    if (d.sf_index < 0) fail(SK_AAC_ERR_UNSUPPORTED_FEATURE, "explicit sample-rate TNS max bands");
    if (d.sf_index > 12) fail(SK_AAC_ERR_UNSUPPORTED_SF_INDEX, "unsupported AAC sampling frequency index");
    const bool is_short = info.window_sequence == SK_EIGHT_SHORT;
    const Layout lay = layout_for(d, info);
    const int wlen = is_short ? 128 : 1024;
    int max_bands = is_short ? kTnsMaxBands128[d.sf_index] : kTnsMaxBands1024[d.sf_index];
    if (max_bands > info.max_sfb) max_bands = info.max_sfb;
    if (max_bands > lay.bands) max_bands = lay.bands;
    for (int w = 0; w < tns.window_count; ++w) {
        const TnsWindow &win = tns.windows[w];
        const int res_bits = win.coef_res ? 4 : 3;
        int bottom = lay.bands;
        for (int f = 0; f < win.filter_count; ++f) {
            const TnsFilter &flt = win.filters[f];
            const int top = bottom;
            bottom = top > flt.length ? top - flt.length : 0;
            if (flt.order == 0) continue;
            const int start = lay.off[bottom < max_bands ? bottom : max_bands];
            const int end = lay.off[top < max_bands ? top : max_bands];
            if (end <= start) continue;
            float lpc[20] = {0}, prev[20] = {0};  // tns.rs:176-206
            for (int i = 0; i < flt.order; ++i) {
                const float refl = -tns_inverse_quant(flt.coeffs[i], flt.coef_bits, res_bits);
                lpc[i] = refl;
                for (int k = 0; k < ((i + 1) >> 1); ++k) {
                    const float fwd = prev[k], bwd = prev[i - 1 - k];
                    lpc[k] = fwd + refl * bwd;
                    lpc[i - 1 - k] = bwd + refl * fwd;
                }
                for (int k = 0; k <= i; ++k) prev[k] = lpc[k];
            }
            float *c = coef + w * wlen;  // tns.rs:237-276
            if (flt.direction) {
                for (int pos = end - 1; pos >= start; --pos) {
                    const int done = end - 1 - pos;
                    const int mo = done < flt.order ? done : flt.order;
                    float v = c[pos];
                    for (int o = 1; o <= mo; ++o) v -= c[pos + o] * lpc[o - 1];
                    c[pos] = v;
                }
            } else {
                for (int pos = start; pos < end; ++pos) {
                    const int done = pos - start;
                    const int mo = done < flt.order ? done : flt.order;
                    float v = c[pos];
                    for (int o = 1; o <= mo; ++o) v -= c[pos - o] * lpc[o - 1];
                    c[pos] = v;
                }
            }
        }
    }
}

// ---- element loop (decoder.rs:104-218, 393-438) -------------------------------------------------------
bool remaining_zero(const BitReader &r) {
    BitReader p = r;
    while (p.remaining() >= 32)
        if (p.read(32) != 0) return false;
    const size_t rem = p.remaining();
    return rem == 0 || p.read((unsigned)rem) == 0;
}

void skip_fill(BitReader &r) {
    size_t count = r.read(4);
    if (count == 15) {
        const uint32_t ext = r.read(8);
        if (ext == 0) fail(SK_AAC_ERR_INVALID_BITSTREAM, "invalid fill element length");
        count += ext - 1;
    }
    if (count == 0) return;
    if (r.remaining() < count * 8)
        fail(SK_AAC_ERR_EOF, "unexpected end of AAC bitstream: requested " + std::to_string(count * 8 > 255 ? 255 : count * 8) +
                                 " bits, " + std::to_string(r.remaining()) + " bits remain");
    const uint32_t ext_type = r.peek(4);
    if (ext_type == 13 || ext_type == 14) fail(SK_AAC_ERR_UNSUPPORTED_FEATURE, "SBR/HE-AAC extension payload");
    r.skip(count * 8);
}

void parse_access_unit(Decoder &d, const uint8_t *au, size_t len, float *coeffs, sk_aac_frame_desc *desc) {
    if (d.padded.size() < len + 8) d.padded.resize(len + 8 + 1024);
    if (len) std::memcpy(d.padded.data(), au, len);
    std::memset(d.padded.data() + len, 0, 8);
    BitReader r(d.padded.data(), len);
    bool decoded = false;
    Channel left, right;
    while (r.remaining() >= 3) {
        if (decoded && remaining_zero(r)) break;
        const unsigned id = r.read(3);
        if (id <= 5) (void)r.read(4);  // element_instance_tag (syntax.rs:54-63)
        if (id == 0) {                 // SCE, decoder.rs:166-184
            if (decoded) fail(SK_AAC_ERR_INVALID_BITSTREAM, "raw access unit contains multiple channel elements");
            if (d.channels != 1) fail(SK_AAC_ERR_INVALID_BITSTREAM, "single channel element does not match configured channel count");
            read_channel(r, left, nullptr);
            decode_spectrum(d, r, left, false, coeffs);
            if (left.tns.present) apply_tns(d, left, coeffs);
            desc->window_sequence[0] = (uint8_t)left.info.window_sequence;
            desc->window_shape[0] = (uint8_t)left.info.window_shape;
            decoded = true;
        } else if (id == 1) {  // CPE, decoder.rs:186-218
            if (decoded) fail(SK_AAC_ERR_INVALID_BITSTREAM, "raw access unit contains multiple channel elements");
            if (d.channels != 2) fail(SK_AAC_ERR_INVALID_BITSTREAM, "channel pair element does not match configured channel count");
            const bool common_window = r.read_bool();
            IcsInfo common;
            MsMask mask;
            if (common_window) {
                common = read_ics_info(r);
                read_ms_mask(r, common, mask);
            }
            read_channel(r, left, common_window ? &common : nullptr);
            decode_spectrum(d, r, left, false, coeffs);
            read_channel(r, right, common_window ? &common : nullptr);
            decode_spectrum(d, r, right, true, coeffs + 1024);
            if (!common_window) {  // decoder.rs:275-285
                bool has_is = false;
                for (int g = 0; g < right.info.num_window_groups; ++g)
                    for (int sfb = 0; sfb < right.info.max_sfb; ++sfb)
                        has_is |= right.cb[g][sfb] == CB_INTENSITY || right.cb[g][sfb] == CB_INTENSITY_NEG;
                if (has_is) fail(SK_AAC_ERR_INVALID_BITSTREAM, "common stereo tools require common window");
            } else {
                apply_stereo_tools(d, mask, left.info, left, right, coeffs, coeffs + 1024);
            }
            if (left.tns.present) apply_tns(d, left, coeffs);
            if (right.tns.present) apply_tns(d, right, coeffs + 1024);
            desc->window_sequence[0] = (uint8_t)left.info.window_sequence;
            desc->window_shape[0] = (uint8_t)left.info.window_shape;
            desc->window_sequence[1] = (uint8_t)right.info.window_sequence;
            desc->window_shape[1] = (uint8_t)right.info.window_shape;
            decoded = true;
        } else if (id == 2) {
            fail(SK_AAC_ERR_UNSUPPORTED_FEATURE, "channel coupling element");
        } else if (id == 3) {
            fail(SK_AAC_ERR_UNSUPPORTED_FEATURE, "low frequency element");
        } else if (id == 4) {
            fail(SK_AAC_ERR_UNSUPPORTED_FEATURE, "data stream element");
        } else if (id == 5) {
            fail(SK_AAC_ERR_UNSUPPORTED_FEATURE, "program config element");
        } else if (id == 6) {
            skip_fill(r);
        } else {
            break;  // END
        }
    }
    if (!decoded) fail(SK_AAC_ERR_INVALID_BITSTREAM, "raw access unit does not contain an AAC-LC channel element");
    if (!remaining_zero(r)) fail(SK_AAC_ERR_INVALID_BITSTREAM, "raw access unit has non-zero trailing bits");
    desc->channels = (uint8_t)d.channels;
    d.n_frames += 1;
    for (int c = 0; c < d.channels; ++c) {
        const Channel &ch = c ? right : left;
        d.n_short += ch.info.window_sequence == SK_EIGHT_SHORT;
        d.n_transition += ch.info.window_sequence == SK_LONG_START || ch.info.window_sequence == SK_LONG_STOP;
        d.n_tns += ch.tns.present;
        d.n_pulse += ch.pulse.present;
    }
}

// ---- AudioSpecificConfig (config.rs:121-319) ----------------------------------------------------------
uint32_t rate_of_index(unsigned idx) {
    static const uint32_t rates[13] = {96000, 88200, 64000, 48000, 44100, 32000, 24000, 22050, 16000, 12000, 11025, 8000, 7350};
    if (idx > 12) fail(SK_AAC_ERR_UNSUPPORTED_SF_INDEX, "unsupported AAC sampling frequency index " + std::to_string(idx));
    return rates[idx];
}

unsigned read_aot(BitReader &r) {  // config.rs:271-279, :43-78
    unsigned v = r.read(5);
    if (v == 31) v = 32 + r.read(6);
    if (v == 0) fail(SK_AAC_ERR_INVALID_AOT, "invalid AAC audio object type 0");
    return v;
}

void parse_asc(Decoder &d, const uint8_t *asc, size_t len) {
    std::vector<uint8_t> padded(len + 8, 0);
    if (len) std::memcpy(padded.data(), asc, len);
    BitReader r(padded.data(), len);
    unsigned aot = read_aot(r);
    auto read_rate = [&](int *index) -> uint32_t {
        const unsigned idx = r.read(4);
        if (idx == 15) { *index = -1; return r.read(24); }
        *index = (int)idx;
        return rate_of_index(idx);
    };
    int sf_index = -1;
    const uint32_t rate = read_rate(&sf_index);
    const unsigned channel_config = r.read(4);
    bool sbr = false, ps = false;
    if (aot == 5 || aot == 29) {
        sbr = true;
        ps = aot == 29;
        int ext_index;
        (void)read_rate(&ext_index);
        aot = read_aot(r);
    }
    int frame_length = 1024;
    switch (aot) {  // read_ga_specific_config, config.rs:290-319
    case 1: case 2: case 3: case 4: case 6: case 17: case 19: case 20: {
        const bool flag = r.read_bool();
        if (r.read_bool()) (void)r.read(14);
        (void)r.read_bool();
        frame_length = flag ? 960 : 1024;
        break;
    }
    default: fail(SK_AAC_ERR_UNSUPPORTED_AOT, "unsupported AAC audio object type " + std::to_string(aot));
    }
    // validate_aac_lc_packet_path, config.rs:233-260
    if (aot != 2) fail(SK_AAC_ERR_UNSUPPORTED_AOT, "unsupported AAC audio object type " + std::to_string(aot));
    if (ps) fail(SK_AAC_ERR_UNSUPPORTED_FEATURE, "parametric stereo");
    if (sbr) fail(SK_AAC_ERR_UNSUPPORTED_FEATURE, "SBR/HE-AAC");
    if (frame_length != 1024) fail(SK_AAC_ERR_UNSUPPORTED_FEATURE, "960-sample AAC frames");
    if (channel_config == 0) fail(SK_AAC_ERR_UNSUPPORTED_FEATURE, "program config element channels");
    if (channel_config > 2) fail(SK_AAC_ERR_UNSUPPORTED_CHANNEL_CONFIG, "unsupported AAC channel configuration " + std::to_string(channel_config));
    d.sample_rate = rate;
    d.sf_index = sf_index;
    d.channels = (int)channel_config;
}


}  // namespace

// the same tables in the flat form aac_entropy_core.h takes (aac_entropy_tables.h)
namespace sk_ec {
const HostTables &host_tables() {
    static const HostTables flat = [] {
        HostTables h;
        const auto &t = tables();  // the host parser's own tables (this file's), not sk_ec::Tables
        for (int book = 0; book < 12; ++book) {
            const Lut &lut = book == 0 ? t.sf : t.cb[book];
            h.lut_offset[book] = (uint32_t)h.lut.size();
            h.primary_bits[book] = (uint32_t)lut.primary_bits;
            h.lut.insert(h.lut.end(), lut.table.begin(), lut.table.end());
            h.tuple_offset[book] = (uint32_t)h.tuples.size();
            if (book == 0) continue;
            for (int i = 0; i < 289; ++i) {
                const Tuple &tu = t.tuples[book][i];
                uint64_t packed = 0;
                for (int k = 0; k < 4; ++k) packed |= (uint64_t)(uint8_t)tu.v[k] << (8 * k);
                packed |= (uint64_t)tu.nsign << 32;
                packed |= (uint64_t)tu.escape << 40;
                for (int k = 0; k < 4; ++k) {  // which of the sign bits after the codeword belongs to value k (15: none)
                    const unsigned at = tu.sshift[k] == 31 ? 15u : (unsigned)tu.sshift[k];
                    packed |= (uint64_t)at << (48 + 4 * k);
                }
                h.tuples.push_back(packed);
            }
        }
        h.pow43.assign(t.pow43, t.pow43 + 8192);
        for (uint32_t v = 8192; v < kHostPow43Len; ++v) h.pow43.push_back(std::pow((float)v, 4.0f / 3.0f));  // as dequantize() above
        for (int sf = -32768; sf <= 32767; ++sf) h.sf_wide.push_back(std::pow(2.0f, ((float)sf - 100.0f) * 0.25f));
        for (int pos = -32768; pos <= 32767; ++pos) h.is_wide.push_back(std::pow(2.0f, -0.25f * (float)pos));
        h.sf_mult.assign(t.sf_mult, t.sf_mult + 768);
        for (int pos = -256; pos <= 255; ++pos) h.is_mult.push_back(std::pow(2.0f, -0.25f * (float)pos));
        for (int res_bits = 3; res_bits <= 4; ++res_bits)
            for (int sgn = -8; sgn <= 8; ++sgn) {
                const float divisor = (float)(sgn < 0 ? (1 << res_bits) - 1 : (1 << res_bits) + 1);
                h.tns_sin.push_back(sgn == 0 ? 0.0f : std::sin((float)sgn * 3.14159274101257324219f / divisor));
            }
        for (int sf = 0; sf < 13; ++sf) {
            const Layout l = long_layout(sf), sh = short_layout(sf);
            h.swb_long_offset[sf] = (uint32_t)h.swb.size();
            h.swb.insert(h.swb.end(), l.off, l.off + l.bands + 1);
            h.bands_long[sf] = (uint8_t)l.bands;
            h.swb_short_offset[sf] = (uint32_t)h.swb.size();
            h.swb.insert(h.swb.end(), sh.off, sh.off + sh.bands + 1);
            h.bands_short[sf] = (uint8_t)sh.bands;
            h.tns_max_long[sf] = kTnsMaxBands1024[sf];
            h.tns_max_short[sf] = kTnsMaxBands128[sf];
        }
        h.meta.assign(102, 0);  // sk_ec::META_WORDS
        for (int b = 0; b < 12; ++b) {
            h.meta[0 + b] = h.lut_offset[b];
            h.meta[12 + b] = h.tuple_offset[b];
        }
        for (int sf = 0; sf < 13; ++sf) {
            h.meta[24 + sf] = h.swb_long_offset[sf];
            h.meta[37 + sf] = h.swb_short_offset[sf];
            h.meta[50 + sf] = h.bands_long[sf];
            h.meta[63 + sf] = h.bands_short[sf];
            h.meta[76 + sf] = h.tns_max_long[sf];
            h.meta[89 + sf] = h.tns_max_short[sf];
        }
        return h;
    }();
    return flat;
}
}  // namespace sk_ec

struct sk_aac_decoder {
    Decoder d;
};

extern "C" {

int sk_aac_decoder_create(const uint8_t *asc, size_t asc_len, sk_aac_decoder **out) try {
    sk::abi_enter();
    if (!asc || !out) return SK_ERR_INVALID_ARG;
    *out = nullptr;
    sk_aac_decoder *dec = new (std::nothrow) sk_aac_decoder();
    if (!dec) return SK_ERR_OOM;
    try {
        parse_asc(dec->d, asc, asc_len);
        (void)tables();
    } catch (const AacError &e) {
        delete dec;
        return e.code;
    }
    *out = dec;
    return SK_OK;
} catch (...) {
    return sk::abi_caught("sk_aac_decoder_create");
}

void sk_aac_decoder_destroy(sk_aac_decoder *dec) try {
    sk::abi_enter();
    delete dec;
} catch (...) {
    (void)sk::abi_caught("sk_aac_decoder_destroy");
}

int sk_aac_decoder_info(const sk_aac_decoder *dec, uint32_t *sample_rate, uint8_t *channels) try {
    sk::abi_enter();
    if (!dec) return SK_ERR_INVALID_ARG;
    if (sample_rate) *sample_rate = dec->d.sample_rate;
    if (channels) *channels = (uint8_t)dec->d.channels;
    return SK_OK;
} catch (...) {
    return sk::abi_caught("sk_aac_decoder_info");
}

int sk_aac_decoder_tool_usage(const sk_aac_decoder *dec, uint32_t out[8]) try {
    sk::abi_enter();
    if (!dec || !out) return SK_ERR_INVALID_ARG;
    const Decoder &d = dec->d;
    const uint32_t v[8] = {d.n_frames, d.n_short, d.n_transition, d.n_tns, d.n_pns_bands, d.n_is_bands, d.n_ms_bands, d.n_pulse};
    std::memcpy(out, v, sizeof(v));
    return SK_OK;
} catch (...) {
    return sk::abi_caught("sk_aac_decoder_tool_usage");
}

const char *sk_aac_decoder_last_error(const sk_aac_decoder *dec) try {
    sk::abi_enter();
    return dec ? dec->d.last_error.c_str() : "";
} catch (...) {
    (void)sk::abi_caught("sk_aac_decoder_last_error");
    return sk::abi_message();
}

int sk_aac_decoder_parse(sk_aac_decoder *dec, const uint8_t *au, size_t len, float *coeffs, sk_aac_frame_desc *desc) try {
    sk::abi_enter();
    if (!dec || (!au && len) || !coeffs || !desc) return SK_ERR_INVALID_ARG;
    try {
        parse_access_unit(dec->d, au, len, coeffs, desc);
    } catch (const AacError &e) {
        dec->d.last_error = e.msg;
        return e.code;
    }
    return SK_OK;
} catch (...) {
    return sk::abi_caught("sk_aac_decoder_parse");
}

// The Huffman half alone: quantised values as i16 + the side record the device needs to dequantise,
// fill noise, run the stereo tools and TNS (aac_entropy_core.h WireUnit).  The core's parser is used for it -- the same
// source the device runs -- so what the device rebuilds is what this call saw.
int sk_aac_decoder_parse_q(sk_aac_decoder *dec, const uint8_t *au, size_t len, int16_t *quant, void *side, sk_aac_frame_desc *desc) try {
    sk::abi_enter();
    if (!dec || (!au && len) || !quant || !side || !desc || len > 8192) return SK_ERR_INVALID_ARG;
    static_assert(sizeof(sk_ec::WireUnit) == SK_AAC_UNIT_SIDE_BYTES, "sk_aac_unit_side size");
    static const sk_ec::Tables view = [] {
        const sk_ec::HostTables &h = sk_ec::host_tables();
        sk_ec::Tables t{};
        t.meta = h.meta.data();
        t.lut = h.lut.data();
        t.tuples = h.tuples.data();
        t.swb = h.swb.data();
        t.pow43 = h.pow43.data();
        t.pow43_lo = h.pow43.data();
        t.sf_mult = h.sf_mult.data();
        t.is_mult = h.is_mult.data();
        t.tns_sin = h.tns_sin.data();
        t.sf_wide = h.sf_wide.data();
        t.is_wide = h.is_wide.data();
        return t;
    }();
    thread_local std::vector<uint32_t> words;
    words.assign((len + 3) / 4 + 2, 0);  // 4-byte aligned, >= 8 zero bytes behind the unit
    if (len) std::memcpy(words.data(), au, len);
    sk_ec::Stream st{dec->d.sf_index, dec->d.channels, 0u};
    int16_t sf0[128], sf1[128];
    sk_ec::WideList wide;
    wide.n = 0;
    const sk_ec::QuantCapture qc{quant, {sf0, sf1}, &wide};
    sk_ec::Scratch scratch;
    uint8_t seq[2] = {0, 0}, shape[2] = {0, 0};
    int rc = sk_ec::parse_unit(view, st, words.data(), (uint32_t)len, nullptr, seq, shape, scratch, sk_ec::PNS_COUNT, &qc);
    if (rc != sk_ec::EC_OK) {  // the reference's message: from the host parser, on the rare failing unit
        dec->d.last_error = "AAC-LC front-end error";
        try {
            std::vector<float> sink(2048);
            sk_aac_frame_desc d2;
            parse_access_unit(dec->d, au, len, sink.data(), &d2);
        } catch (const AacError &e) {
            if (e.code == rc) dec->d.last_error = e.msg;
        }
        return rc;
    }
    const int32_t tail = sk_ec::unit_tail(words.data(), (uint32_t)len, scratch.resume_pos);
    const int16_t *sfs[2] = {sf0, sf1};
    sk_ec::pack_unit(scratch, dec->d.channels, sfs, tail, *reinterpret_cast<sk_ec::WireUnit *>(side), &wide);
    desc->channels = (uint8_t)dec->d.channels;
    for (int c = 0; c < 2; ++c) {
        desc->window_sequence[c] = c < dec->d.channels ? seq[c] : 0;
        desc->window_shape[c] = c < dec->d.channels ? shape[c] : 0;
    }
    return SK_OK;
} catch (...) {
    return sk::abi_caught("sk_aac_decoder_parse_q");
}

// parse_adts_access_unit, soundkit-decoder/src/lib.rs:1007-1027 (+ the 13-bit frame length for framing a file)
int sk_adts_parse(const uint8_t *data, size_t len, size_t *frame_len, size_t *payload_off, size_t *payload_len,
                  uint8_t asc[2]) try {
    sk::abi_enter();
    if (!data || len < 7 || data[0] != 0xff || (data[1] & 0xf6) != 0xf0) return SK_AAC_ERR_INVALID_BITSTREAM;
    const unsigned aot = ((data[2] >> 6) & 3) + 1;
    const unsigned sr = (data[2] >> 2) & 0x0f;
    const unsigned ch = ((data[2] & 1) << 2) | (data[3] >> 6);
    const size_t header = (data[1] & 1) ? 7 : 9;
    if (len < header) return SK_AAC_ERR_INVALID_BITSTREAM;
    const size_t flen = ((size_t)(data[3] & 3) << 11) | ((size_t)data[4] << 3) | (data[5] >> 5);
    if (flen < header) return SK_AAC_ERR_INVALID_BITSTREAM;
    if (asc) {
        asc[0] = (uint8_t)((aot << 3) | (sr >> 1));
        asc[1] = (uint8_t)(((sr & 1) << 7) | (ch << 3));
    }
    if (frame_len) *frame_len = flen;
    if (payload_off) *payload_off = header;
    if (payload_len) *payload_len = flen - header;
    return SK_OK;
} catch (...) {
    return sk::abi_caught("sk_adts_parse");
}

}  // extern "C"
