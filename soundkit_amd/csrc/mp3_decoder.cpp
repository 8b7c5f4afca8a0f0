// mp3_decoder.cpp -- parts 2 and 3 of the Layer III main data (scale factors, the Huffman stage) over caller-supplied
// tables, and a decoder handle in the shape of soundkit-mp3's Mp3Decoder (soundkit-mp3/src/lib.rs:147-374).
//
// In the reference all of this is inside nanomp3::Decoder::decode (lib.rs:284), a crate that is not in the reference tree.
// The syntax is ISO/IEC 11172-3 2.4.1.7 / 2.4.2.7 / 2.4.3.4.5-6 and 13818-3 2.4.3.2; the data it runs on -- Table B.7's 32
// code tables, the two count1 tables, the scale-factor length and partition tables, Table B.8 -- is NOT in this file: it
// arrives through sk_mp3_tables in the standard's own presentation (a length and a bit pattern per code) and is turned into
// binary decoding tries here.  Host code throughout; the arithmetic behind it (requantisation, stereo, reorder, hybrid
// synthesis) runs on the GPU, one launch each per decode call (mp3_requant.hip, mp3_hybrid.hip).
#include "../../include/soundkit_amd.h"
#include "sk_abi.h"
#include "mp3_iso_tables.h"

#include <cmath>
#include <cstring>
#include <memory>
#include <vector>

struct sk_mp3_codebook {
    struct Trie {
        std::vector<int32_t> next;  // [node][bit]: > 0 child node, <= 0: -(symbol) - 1 ... 0 = empty
        uint8_t xlen = 0, linbits = 0;
        // the next kLutBits bits of the stream -> (length << 16 | symbol + 1) of the code they start with, 0 if that code is longer
        // (or the bits are no code): one look-up for the short codes, which are the frequent ones; the trie walk for the rest
        std::vector<uint32_t> lut;
    };
    static constexpr int kLutBits = 10;
    Trie big[32], count1[2];
    sk_mp3_tables t;  // hlen / hcod pointers inside are not kept (copied into the tries)
};

namespace {

constexpr int32_t kEmpty = 0;
inline int32_t leaf(int symbol) { return -symbol - 1; }

// adds one code; false if it collides with an earlier one (equal, a prefix of it, or prefixed by it)
bool trie_add(sk_mp3_codebook::Trie &t, uint32_t code, int len, int symbol) {
    if (len < 1 || len > 32) return false;
    if (len < 32 && (code >> len)) return false;
    if (t.next.empty()) t.next.assign(2, kEmpty);
    int32_t node = 0;
    for (int i = len - 1; i >= 0; --i) {
        const int bit = (code >> i) & 1;
        int32_t &slot = t.next[(size_t)node * 2 + bit];
        if (i == 0) {
            if (slot != kEmpty) return false;
            slot = leaf(symbol);
            return true;
        }
        if (slot < 0) return false;  // an earlier, shorter code ends here
        if (slot == kEmpty) {
            const int32_t fresh = (int32_t)(t.next.size() / 2);
            t.next[(size_t)node * 2 + bit] = fresh;  // (slot may dangle after the resize below)
            t.next.resize(t.next.size() + 2, kEmpty);
            node = fresh;
        } else {
            node = slot;
        }
    }
    return false;
}

struct Bits {  // bits past the end read as zero, and the position keeps counting (the callers compare it with the granule's end)
    const uint8_t *p;
    size_t len_bits;
    size_t pos = 0;
    int bit() {
        if (pos >= len_bits) {
            ++pos;
            return 0;
        }
        const int v = (p[pos >> 3] >> (7 - (pos & 7))) & 1;
        ++pos;
        return v;
    }
    // the next n <= 24 bits without consuming them
    uint32_t peek(int n) const {
        const size_t byte = pos >> 3, total = (len_bits + 7) >> 3;
        uint32_t w;
        if (byte + 4 <= total) {
            w = ((uint32_t)p[byte] << 24) | ((uint32_t)p[byte + 1] << 16) | ((uint32_t)p[byte + 2] << 8) | (uint32_t)p[byte + 3];
        } else {
            w = 0;
            for (size_t i = 0; i < 4; ++i) w = (w << 8) | (byte + i < total ? (uint32_t)p[byte + i] : 0u);
        }
        return n ? (uint32_t)(w << (pos & 7)) >> (32 - n) : 0u;
    }
    uint32_t get(int n) {
        if (n == 0) return 0;
        if (n <= 24 && pos + (size_t)n <= len_bits) {
            const uint32_t v = peek(n);
            pos += (size_t)n;
            return v;
        }
        uint32_t v = 0;
        for (int i = 0; i < n; ++i) v = (v << 1) | (uint32_t)bit();
        return v;
    }
};

void trie_build_lut(sk_mp3_codebook::Trie &t) {
    constexpr int kBits = sk_mp3_codebook::kLutBits;
    t.lut.assign((size_t)1 << kBits, 0u);
    if (t.next.empty()) return;
    for (uint32_t v = 0; v < (1u << kBits); ++v) {
        int32_t node = 0;
        for (int depth = 1; depth <= kBits; ++depth) {
            const int32_t slot = t.next[(size_t)node * 2 + ((v >> (kBits - depth)) & 1u)];
            if (slot < 0) {
                t.lut[v] = ((uint32_t)depth << 16) | (uint32_t)(-slot);  // -slot = symbol + 1
                break;
            }
            if (slot == kEmpty) break;
            node = slot;
        }
    }
}

// -1: the bits are no code of this table
int trie_read(const sk_mp3_codebook::Trie &t, Bits &b) {
    if (t.next.empty()) return -1;
    if (b.pos + (size_t)sk_mp3_codebook::kLutBits <= b.len_bits) {
        const uint32_t hit = t.lut[b.peek(sk_mp3_codebook::kLutBits)];
        if (hit) {
            b.pos += hit >> 16;
            return (int)(hit & 0xffffu) - 1;
        }
    }
    int32_t node = 0;
    for (int depth = 0; depth < 33; ++depth) {
        const int32_t slot = t.next[(size_t)node * 2 + b.bit()];
        if (slot < 0) return -slot - 1;
        if (slot == kEmpty) return -1;
        node = slot;
    }
    return -1;
}

int rate_row(uint32_t hz) {
    static const uint32_t rates[9] = {44100, 48000, 32000, 22050, 24000, 16000, 11025, 12000, 8000};
    for (int i = 0; i < 9; ++i)
        if (rates[i] == hz) return i;
    return -1;
}

// 11172-3 2.4.2.7: the scale factors of one granule / channel of an MPEG-1 frame
void scale_factors_v1(const sk_mp3_codebook &cb, const sk_mp3_side_info &side, int gr, int ch, Bits &b, sk_mp3_granule_data out[2][2]) {
    const sk_mp3_granule_side &s = side.gr[gr][ch];
    sk_mp3_granule_data &g = out[gr][ch];
    const int slen1 = cb.t.slen[s.scalefac_compress & 15][0], slen2 = cb.t.slen[s.scalefac_compress & 15][1];
    g.preflag = s.preflag;
    if (s.window_switching && s.block_type == 2) {
        int first_short = 0;
        if (s.mixed_block_flag) {
            for (int band = 0; band < 8; ++band) g.scalefac_l[band] = (uint8_t)b.get(slen1);
            first_short = 3;
        }
        for (int band = first_short; band < 12; ++band)
            for (int w = 0; w < 3; ++w) g.scalefac_s[band][w] = (uint8_t)b.get(band < 6 ? slen1 : slen2);
        return;
    }
    static const int group_begin[5] = {0, 6, 11, 16, 21};
    for (int group = 0; group < 4; ++group)
        for (int band = group_begin[group]; band < group_begin[group + 1]; ++band) {
            if (gr == 1 && side.scfsi[ch][group]) g.scalefac_l[band] = out[0][ch].scalefac_l[band];
            else g.scalefac_l[band] = (uint8_t)b.get(group < 2 ? slen1 : slen2);
        }
}

// 13818-3 2.4.3.2: scalefac_compress (9 bits) -> four lengths and a row of the partition table
int scale_factors_lsf(const sk_mp3_codebook &cb, const sk_mp3_frame_info &h, const sk_mp3_side_info &side, int ch, Bits &b, sk_mp3_granule_data &g) {
    const sk_mp3_granule_side &s = side.gr[0][ch];
    int slen[4] = {0, 0, 0, 0}, row;
    unsigned sfc = s.scalefac_compress;
    g.preflag = 0;
    const bool intensity_channel = h.mode == 1 && (h.mode_ext & 1) && ch == 1;
    if (!intensity_channel) {
        if (sfc < 400) {
            slen[0] = (int)(sfc >> 4) / 5, slen[1] = (int)(sfc >> 4) % 5, slen[2] = (int)(sfc & 15) >> 2, slen[3] = (int)sfc & 3;
            row = 0;
        } else if (sfc < 500) {
            sfc -= 400;
            slen[0] = (int)(sfc >> 2) / 5, slen[1] = (int)(sfc >> 2) % 5, slen[2] = (int)sfc & 3;
            row = 1;
        } else {
            sfc -= 500;
            slen[0] = (int)sfc / 3, slen[1] = (int)sfc % 3;
            g.preflag = 1;
            row = 2;
        }
    } else {
        g.intensity_scale = (uint8_t)(sfc & 1);
        sfc >>= 1;
        if (sfc < 180) {
            slen[0] = (int)sfc / 36, slen[1] = (int)(sfc % 36) / 6, slen[2] = (int)(sfc % 36) % 6;
            row = 3;
        } else if (sfc < 244) {
            sfc -= 180;
            slen[0] = (int)(sfc & 63) >> 4, slen[1] = (int)(sfc & 15) >> 2, slen[2] = (int)sfc & 3;
            row = 4;
        } else {
            sfc -= 244;
            slen[0] = (int)sfc / 3, slen[1] = (int)sfc % 3;
            row = 5;
        }
    }
    const int column = (s.window_switching && s.block_type == 2) ? (s.mixed_block_flag ? 2 : 1) : 0;
    const uint8_t *parts = cb.t.lsf_partitions[row][column];
    // the factors come as one list: long bands in order; short bands band by band, window by window; a mixed block's
    // list starts with its long bands (those below line 36) and goes on with short band 3
    int index = 0;
    const int long_bands = column == 0 ? 22 : (column == 2 ? 6 : 0);
    for (int part = 0; part < 4; ++part)
        for (int k = 0; k < parts[part]; ++k, ++index) {
            uint8_t v = (uint8_t)b.get(slen[part]);
            // 13818-3 2.4.3.2: in the intensity channel the largest value a field can hold says "this band is not intensity coded"
            if (intensity_channel && slen[part] > 0 && v == (1u << slen[part]) - 1u) v |= 0x80;
            if (index < long_bands) {
                if (index < 22) g.scalefac_l[index] = v;
            } else {
                const int rel = index - long_bands + (column == 2 ? 9 : 0);
                if (rel / 3 < 13) g.scalefac_s[rel / 3][rel % 3] = v;
            }
        }
    return SK_OK;
}

// 2.4.3.4.6: big_values pairs in up to three regions, then count1 quadruples up to the end of part2_3_length
int huffman(const sk_mp3_codebook &cb, const sk_mp3_frame_info &h, const sk_mp3_granule_side &s, Bits &b, size_t end_bit, sk_mp3_granule_data &g) {
    const size_t start_bit = b.pos;
    const int row = rate_row(h.sample_rate);
    if (row < 0 || !cb.t.rates_present[row]) return SK_MP3_UNSUPPORTED;
    const uint16_t *lo = cb.t.long_offsets[row], *so = cb.t.short_offsets[row];
    // region boundaries count scale-factor band partitions of the granule's own cut of the 576 lines (2.4.2.7)
    int widths[64], n_widths = 0;
    const bool is_short = s.window_switching && s.block_type == 2;
    if (!is_short) {
        for (int band = 0; band < 22; ++band) widths[n_widths++] = lo[band + 1] - lo[band];
    } else {
        int first_short = 0;
        if (s.mixed_block_flag) {
            for (int band = 0; band < 22 && lo[band + 1] <= 36; ++band) widths[n_widths++] = lo[band + 1] - lo[band];
            while (first_short < 13 && 3 * so[first_short] < 36) ++first_short;
        }
        for (int band = first_short; band < 13; ++band)
            for (int w = 0; w < 3; ++w) widths[n_widths++] = so[band + 1] - so[band];
    }
    auto boundary = [&](int count) {
        int at = 0;
        for (int i = 0; i < count && i < n_widths; ++i) at += widths[i];
        return at > 576 ? 576 : at;
    };
    const int big_end = 2 * (int)s.big_values;
    if (big_end > 576) return SK_MP3_INVALID;
    int region1 = boundary(s.region0_count + 1), region2 = boundary(s.region0_count + 1 + s.region1_count + 1);
    if (s.window_switching) region2 = 576;  // two regions only
    if (region1 > big_end) region1 = big_end;
    if (region2 > big_end) region2 = big_end;
    const int bounds[4] = {0, region1, region2, big_end};
    int line = 0;
    for (int region = 0; region < 3; ++region) {
        const sk_mp3_codebook::Trie &t = cb.big[s.table_select[region] & 31];
        for (; line < bounds[region + 1]; line += 2) {
            int x = 0, y = 0;
            if (t.xlen) {
                const int symbol = trie_read(t, b);
                if (symbol < 0) return SK_MP3_INVALID;
                x = symbol / t.xlen, y = symbol % t.xlen;
                if (t.linbits && x == t.xlen - 1) x += (int)b.get(t.linbits);
                if (x && b.bit()) x = -x;
                if (t.linbits && y == t.xlen - 1) y += (int)b.get(t.linbits);
                if (y && b.bit()) y = -y;
            } else if (bounds[region + 1] > bounds[region] && (s.table_select[region] & 31) != 0) {
                return SK_MP3_INVALID;  // a region that holds lines names a table without codes (4, 14)
            }
            g.is[line] = (int16_t)x, g.is[line + 1] = (int16_t)y;
        }
    }
    if (b.pos > end_bit) return SK_MP3_INVALID;  // the big values alone overran part2_3_length
    const size_t part3_begin = start_bit;
    size_t accepted = b.pos;
    const sk_mp3_codebook::Trie &q = cb.count1[s.count1table_select & 1];
    while (b.pos < end_bit && line + 4 <= 576) {
        const int symbol = trie_read(q, b);
        if (symbol < 0) return SK_MP3_INVALID;
        int v[4];
        for (int k = 0; k < 4; ++k) {
            v[k] = (symbol >> (3 - k)) & 1;
            if (v[k] && b.bit()) v[k] = -1;
        }
        if (b.pos > end_bit) break;  // a quadruple that reaches past the end is stuffing, not data
        for (int k = 0; k < 4; ++k) g.is[line + k] = (int16_t)v[k];
        line += 4;
        accepted = b.pos;
    }
    g.part3_bits = (uint16_t)(accepted - part3_begin);
    g.nonzero_lines = (uint16_t)line;
    return SK_OK;
}

}  // namespace

extern "C" {

int sk_mp3_codebook_create(const sk_mp3_tables *t, sk_mp3_codebook **out) try {
    sk::abi_enter();
    if (!t || !out) return SK_ERR_INVALID_ARG;
    *out = nullptr;
    std::unique_ptr<sk_mp3_codebook> cb(new (std::nothrow) sk_mp3_codebook);
    if (!cb) return SK_ERR_OOM;
    cb->t = *t;
    for (int i = 0; i < 32; ++i) {
        const sk_mp3_code_table &src = t->big_values[i];
        sk_mp3_codebook::Trie &dst = cb->big[i];
        cb->t.big_values[i].hlen = nullptr, cb->t.big_values[i].hcod = nullptr;
        if (src.xlen == 0) continue;
        if (src.xlen > 16 || !src.hlen || !src.hcod || src.linbits > 13) return SK_ERR_INVALID_ARG;
        dst.xlen = src.xlen, dst.linbits = src.linbits;
        for (int s = 0; s < src.xlen * src.xlen; ++s)
            if (!trie_add(dst, src.hcod[s], src.hlen[s], s)) return SK_MP3_INVALID;
        trie_build_lut(dst);
    }
    for (int k = 0; k < 2; ++k) {
        for (int s = 0; s < 16; ++s)
            if (!trie_add(cb->count1[k], t->count1_hcod[k][s], t->count1_hlen[k][s], s)) return SK_MP3_INVALID;
        trie_build_lut(cb->count1[k]);
    }
    for (int i = 0; i < 16; ++i)
        if (t->slen[i][0] > 4 || t->slen[i][1] > 4) return SK_ERR_INVALID_ARG;  // a scale factor has at most 4 bits
    for (int row = 0; row < 6; ++row)
        for (int col = 0; col < 3; ++col) {
            int sum = 0;
            for (int part = 0; part < 4; ++part) sum += t->lsf_partitions[row][col][part];
            if (sum > 39) return SK_ERR_INVALID_ARG;  // 13 short bands x 3 windows at most
        }
    for (int row = 0; row < 9; ++row) {
        if (!t->rates_present[row]) continue;
        const uint16_t *lo = t->long_offsets[row], *so = t->short_offsets[row];
        if (lo[0] != 0 || lo[22] != 576 || so[0] != 0 || so[13] != 192) return SK_ERR_INVALID_ARG;
        for (int i = 0; i < 22; ++i)
            if (lo[i] >= lo[i + 1]) return SK_ERR_INVALID_ARG;
        for (int i = 0; i < 13; ++i)
            if (so[i] >= so[i + 1]) return SK_ERR_INVALID_ARG;
    }
    *out = cb.release();
    return SK_OK;
} catch (...) {
    return sk::abi_caught("sk_mp3_codebook_create");
}

void sk_mp3_codebook_destroy(sk_mp3_codebook *cb) try {
    sk::abi_enter();
    delete cb;
} catch (...) {
    (void)sk::abi_caught("sk_mp3_codebook_destroy");
}

// The standard's own tables (csrc/mp3_iso_tables.h) in the caller's presentation; pointers are to static storage.
int sk_mp3_iso_tables(sk_mp3_tables *out) try {
    sk::abi_enter();
    if (!out) return SK_ERR_INVALID_ARG;
    std::memset(out, 0, sizeof *out);
    for (int t = 0; t < 32; ++t) {
        out->big_values[t].xlen = sk_mp3_iso::select[t].xlen;
        out->big_values[t].linbits = sk_mp3_iso::linbits[t];
        out->big_values[t].hlen = sk_mp3_iso::select[t].hlen;
        out->big_values[t].hcod = sk_mp3_iso::select[t].hcod;
    }
    std::memcpy(out->count1_hlen, sk_mp3_iso::count1_hlen, sizeof out->count1_hlen);
    std::memcpy(out->count1_hcod, sk_mp3_iso::count1_hcod, sizeof out->count1_hcod);
    std::memcpy(out->slen, sk_mp3_iso::slen, sizeof out->slen);
    std::memcpy(out->lsf_partitions, sk_mp3_iso::lsf_partitions, sizeof out->lsf_partitions);
    std::memcpy(out->long_offsets, sk_mp3_iso::long_offsets, sizeof out->long_offsets);
    std::memcpy(out->short_offsets, sk_mp3_iso::short_offsets, sizeof out->short_offsets);
    std::memset(out->rates_present, 1, sizeof out->rates_present);
    std::memcpy(out->pretab, sk_mp3_iso::pretab, sizeof out->pretab);
    for (int i = 0; i < 512; ++i) out->window[i] = (float)sk_mp3_iso::window_q16[i] * (1.0f / 65536.0f);  // exact: |q16| < 2^17
    return SK_OK;
} catch (...) {
    return sk::abi_caught("sk_mp3_iso_tables");
}

int sk_mp3_codebook_create_iso(sk_mp3_codebook **out) try {
    sk::abi_enter();
    sk_mp3_tables t;
    const int rc = sk_mp3_iso_tables(&t);
    return rc == SK_OK ? sk_mp3_codebook_create(&t, out) : rc;
} catch (...) {
    return sk::abi_caught("sk_mp3_codebook_create_iso");
}

int sk_mp3_decode_main_data(const sk_mp3_codebook *cb, const sk_mp3_frame_info *h, const sk_mp3_side_info *side, const uint8_t *main, size_t main_len,
                            sk_mp3_granule_data out[2][2]) try {
    sk::abi_enter();
    if (!cb || !h || !side || !out || (main_len && !main)) return SK_ERR_INVALID_ARG;
    if (side->granules < 1 || side->granules > 2 || side->channels < 1 || side->channels > 2) return SK_ERR_INVALID_ARG;
    for (int gr = 0; gr < side->granules; ++gr)  // only the cells this frame has (one of four for an LSF mono frame)
        for (int ch = 0; ch < side->channels; ++ch) std::memset(&out[gr][ch], 0, sizeof(sk_mp3_granule_data));
    size_t start = 0;
    int worst = SK_OK;
    for (int gr = 0; gr < side->granules; ++gr)
        for (int ch = 0; ch < side->channels; ++ch) {
            const sk_mp3_granule_side &s = side->gr[gr][ch];
            sk_mp3_granule_data &g = out[gr][ch];
            const size_t end = start + s.part2_3_length;
            Bits b{main, main_len * 8, start};
            int rc = SK_OK;
            if (end > main_len * 8) rc = SK_MP3_NEED_MORE;
            if (rc == SK_OK) {
                if (h->version == 1) scale_factors_v1(*cb, *side, gr, ch, b, out);
                else rc = scale_factors_lsf(*cb, *h, *side, ch, b, g);
                g.part2_bits = (uint16_t)(b.pos - start);
                if (rc == SK_OK && b.pos > end) rc = SK_MP3_INVALID;  // the scale factors alone overran part2_3_length
            }
            if (rc == SK_OK) rc = huffman(*cb, *h, s, b, end, g);
            if (rc != SK_OK) std::memset(g.is, 0, sizeof g.is);
            g.status = rc;
            if (rc != SK_OK && worst == SK_OK) worst = rc;
            start = end;
        }
    return worst;
} catch (...) {
    return sk::abi_caught("sk_mp3_decode_main_data");
}

}  // extern "C"

// ---- the decoder handle ---------------------------------------------------------------------------------------------------

struct sk_mp3_decoder {
    sk_engine *engine = nullptr;
    const sk_mp3_codebook *cb = nullptr;
    sk_mp3_codebook *own_cb = nullptr;  // the standard's tables, when the caller passed none
    std::vector<uint8_t> buffer, reservoir;
    uint32_t sample_rate = 0;  // of the first frame (Option::get_or_insert, lib.rs:203-204)
    uint8_t channels = 0;
    uint64_t frames = 0;
    uint32_t free_format_bytes = 0;  // a free-format stream's frame length once measured (sk_mp3_scan_free)
    bool stream_open = false;
    uint32_t stream = 0;
    uint8_t stream_channels = 0;
    // scratch of one call
    std::vector<sk_mp3_frame_info> found;
    std::vector<sk_mp3_requant_granule> granules;
    std::vector<sk_mp3_granule_desc> descs;
    std::vector<int16_t> is;
    std::vector<float> pcm;
    std::vector<int32_t> status;
    std::vector<uint8_t> staged_reservoir, main;
    struct Queued {
        uint32_t first_granule, granules;
        size_t first_sample, samples;
    };
    std::vector<Queued> queued;
};

namespace {

constexpr size_t kMaxBuffered = 4u * 1024 * 1024;  // MAX_MP3_STREAM_BUFFER_BYTES, lib.rs:155
constexpr size_t kReservoirKept = 2048;            // main_data_begin reaches back 511 bytes at most

// soundkit-mp3/src/lib.rs:387-396
int32_t mp3_f32_to_i32(float sample) {
    const float scaled = std::round(sample * 2147483648.0f);  // i32::MAX as f32
    if (scaled > 2147483648.0f) return INT32_MAX;
    if (scaled < -2147483648.0f) return INT32_MIN;
    if (scaled != scaled) return 0;
    if (scaled >= 2147483648.0f) return INT32_MAX;  // Rust's saturating `as`
    return (int32_t)scaled;
}

enum class Out { I16, I32, F32 };

int decode(sk_mp3_decoder *d, const uint8_t *input, size_t len, void *out, size_t out_cap, size_t *written, Out kind) {
    if (!d || !written || (len && !input) || (out_cap && !out)) return SK_ERR_INVALID_ARG;
    *written = 0;
    if (d->buffer.size() + len > kMaxBuffered) return SK_PIPE_CHUNK_TOO_LARGE;
    d->buffer.insert(d->buffer.end(), input, input + len);
    if (d->buffer.empty()) return SK_OK;

    d->found.resize(d->buffer.size() / 24 + 2);
    uint32_t n_found = 0;
    size_t scanned = 0;
    uint32_t free_format_bytes = d->free_format_bytes;  // staged like the rest: the scan may measure it anew
    uint32_t free_format_run = d->free_format_bytes;
    int rc = sk_mp3_scan_free(d->buffer.data(), d->buffer.size(), d->found.data(), (uint32_t)d->found.size(), &n_found, &scanned, &free_format_bytes);
    if (rc != SK_OK) {
        d->buffer.resize(d->buffer.size() - len);  // a failed call leaves the decoder as it was
        return rc;
    }
    if (n_found > d->found.size()) n_found = (uint32_t)d->found.size();

    // Everything the call changes is staged here and committed after the device work succeeded: a failed call can be
    // repeated (the same frames are parsed again against the same reservoir).
    d->staged_reservoir = d->reservoir;
    std::vector<uint8_t> &reservoir = d->staged_reservoir;
    uint32_t sample_rate = d->sample_rate;
    uint8_t channels = d->channels;
    uint64_t frames = d->frames;
    d->queued.clear();

    d->granules.clear(), d->descs.clear(), d->is.clear();
    size_t samples = 0, consumed = 0;
    int result = SK_OK;
    bool stopped = false;
    for (uint32_t k = 0; k < n_found && !stopped; ++k) {
        const sk_mp3_frame_info &h = d->found[k];
        const uint8_t *frame = d->buffer.data() + h.offset;
        const size_t frame_samples = (size_t)h.samples_per_channel * h.channels;
        sk_mp3_side_info side;
        const size_t head = 4u + (h.has_crc ? 2u : 0u) + h.side_info_bytes;
        bool decodable = sk_mp3_parse_side_info(frame, h.frame_bytes, &h, &side) == SK_OK;
        d->main.resize(reservoir.size() + h.frame_bytes);
        size_t main_len = 0;
        if (decodable) decodable = sk_mp3_main_data(frame, h.frame_bytes, &h, &side, reservoir.data(), reservoir.size(), d->main.data(), d->main.size(), &main_len) == SK_OK;
        sk_mp3_granule_data data[2][2];
        if (decodable) decodable = sk_mp3_decode_main_data(d->cb, &h, &side, d->main.data(), main_len, data) == SK_OK;
        const bool joint = h.mode == 1;
        if (decodable) {
            if (samples + frame_samples > out_cap) {  // write_frame_*: "Output buffer too small for decoded frame"
                result = SK_ERR_CAPACITY;
                break;
            }
            if (!d->stream_open || d->stream_channels != h.channels) {  // the carried synthesis state belongs to a channel count
                if (!d->granules.empty()) break;  // granules queued for the old stream go first; this frame waits for the next call
                if (d->stream_open) (void)sk_stream_close(d->engine, d->stream);
                d->stream_open = false;
                rc = sk_stream_open(d->engine, h.sample_rate, h.channels, &d->stream);
                if (rc != SK_OK) {
                    d->buffer.resize(d->buffer.size() - len);
                    return rc;
                }
                d->stream_open = true;
                d->stream_channels = h.channels;
            }
            if (!sample_rate) sample_rate = h.sample_rate;
            if (!channels) channels = h.channels;
            d->queued.push_back({(uint32_t)d->granules.size(), h.granules, samples, frame_samples});
            for (int gr = 0; gr < h.granules; ++gr) {
                sk_mp3_requant_granule g;
                std::memset(&g, 0, sizeof g);
                g.sample_rate = h.sample_rate;
                g.channels = h.channels;
                g.ms_stereo = joint && (h.mode_ext & 2);
                g.intensity_stereo = joint && (h.mode_ext & 1);
                g.lsf = h.version != 1;
                if (g.lsf && g.intensity_stereo && h.channels == 2 && data[gr][1].intensity_scale) g.intensity_stereo |= 2;
                sk_mp3_granule_desc desc;
                std::memset(&desc, 0, sizeof desc);
                desc.stream = d->stream;
                desc.channels = h.channels;
                for (int ch = 0; ch < h.channels; ++ch) {
                    const sk_mp3_granule_side &s = side.gr[gr][ch];
                    const sk_mp3_granule_data &src = data[gr][ch];
                    sk_mp3_requant_channel &c = g.ch[ch];
                    c.global_gain = s.global_gain, c.scalefac_scale = s.scalefac_scale, c.preflag = src.preflag;
                    c.block_type = s.block_type, c.mixed_block_flag = s.mixed_block_flag;
                    std::memcpy(c.subblock_gain, s.subblock_gain, 3);
                    std::memcpy(c.scalefac_l, src.scalefac_l, 22);
                    std::memcpy(c.scalefac_s, src.scalefac_s, 39);
                    desc.block_type[ch] = s.block_type, desc.mixed_block_flag[ch] = s.mixed_block_flag;
                    d->is.insert(d->is.end(), src.is, src.is + 576);
                }
                d->granules.push_back(g);
                d->descs.push_back(desc);
            }
            samples += frame_samples;
            frames += 1;
        }
        // whatever became of the frame, its own main data is what later frames reach back into
        if (h.frame_bytes > head) reservoir.insert(reservoir.end(), frame + head, frame + h.frame_bytes);
        if (reservoir.size() > 4 * kReservoirKept) reservoir.erase(reservoir.begin(), reservoir.end() - kReservoirKept);  // trimmed now and then, not per frame
        consumed = h.offset + h.frame_bytes;
        // the free-format length in force behind this frame: what the scan of the bytes that stay buffered has to start from
        if ((frame[2] >> 4) == 0) free_format_run = h.frame_bytes - h.padding;
        if (decodable && out_cap - samples < SK_MP3_MAX_SAMPLES_PER_FRAME) stopped = true;  // lib.rs:300-302
        if (k + 1 == n_found) consumed = scanned, free_format_run = free_format_bytes;  // every frame taken: garbage in front of an incomplete frame goes too
    }

    const uint32_t n = (uint32_t)d->granules.size();
    if (n) {
        d->status.assign(n, 0);
        if (kind == Out::I16) {
            rc = sk_mp3_decode_granules_s16(d->engine, d->granules.data(), d->descs.data(), d->is.data(), (int16_t *)out, n, d->status.data());
        } else if (kind == Out::F32) {
            rc = sk_mp3_decode_granules_f32(d->engine, d->granules.data(), d->descs.data(), d->is.data(), (float *)out, n, d->status.data());
        } else {
            d->pcm.resize(d->is.size());
            rc = sk_mp3_decode_granules_f32(d->engine, d->granules.data(), d->descs.data(), d->is.data(), d->pcm.data(), n, d->status.data());
            for (size_t i = 0; rc == SK_OK && i < samples; ++i) ((int32_t *)out)[i] = mp3_f32_to_i32(d->pcm[i]);
        }
        if (rc != SK_OK) {
            d->buffer.resize(d->buffer.size() - len);
            return rc;
        }
        // a frame one of whose granules a GPU stage rejected is consumed without output, like the frames the host stages
        // reject: its samples are taken out of what the call returns
        const size_t width = kind == Out::I16 ? 2 : 4;
        size_t kept = 0;
        for (const auto &q : d->queued) {
            bool ok = true;
            for (uint32_t g = 0; g < q.granules; ++g) ok = ok && d->status[q.first_granule + g] == 0;
            if (!ok) {
                frames -= 1;
                continue;
            }
            if (kept != q.first_sample) std::memmove((uint8_t *)out + kept * width, (uint8_t *)out + q.first_sample * width, q.samples * width);
            kept += q.samples;
        }
        samples = kept;
    } else if (n_found == 0) {
        consumed = scanned;
        free_format_run = free_format_bytes;
    }
    d->reservoir.swap(d->staged_reservoir);
    d->sample_rate = sample_rate, d->channels = channels, d->frames = frames;
    d->free_format_bytes = free_format_run;
    d->buffer.erase(d->buffer.begin(), d->buffer.begin() + (ptrdiff_t)consumed);
    *written = samples;
    return result;
}

}  // namespace

extern "C" {

int sk_mp3_decoder_create(sk_engine *e, const sk_mp3_codebook *cb, sk_mp3_decoder **out) try {
    sk::abi_enter();
    if (!e || !out) return SK_ERR_INVALID_ARG;
    *out = nullptr;
    sk_mp3_codebook *own = nullptr;
    if (!cb) {  // Mp3Decoder::new() takes no tables: the decoder it wraps has the standard's built in
        const int rc = sk_mp3_codebook_create_iso(&own);
        if (rc != SK_OK) return rc;
        cb = own;
    }
    struct Guard {
        sk_mp3_codebook *p;
        ~Guard() { delete p; }
    } guard{own};
    static const uint32_t rates[9] = {44100, 48000, 32000, 22050, 24000, 16000, 11025, 12000, 8000};
    for (int row = 0; row < 9; ++row)
        if (cb->t.rates_present[row]) {
            const int rc = sk_mp3_set_band_tables(e, rates[row], cb->t.long_offsets[row], cb->t.short_offsets[row], cb->t.pretab);
            if (rc != SK_OK) return rc;
        }
    const int rc = sk_mp3_set_synthesis_window(e, cb->t.window);
    if (rc != SK_OK) return rc;
    sk_mp3_decoder *d = new (std::nothrow) sk_mp3_decoder;
    if (!d) return SK_ERR_OOM;
    d->engine = e;
    d->cb = cb;
    d->own_cb = own;
    guard.p = nullptr;
    d->buffer.reserve(16 * 1024);  // lib.rs:160
    *out = d;
    return SK_OK;
} catch (...) {
    return sk::abi_caught("sk_mp3_decoder_create");
}

void sk_mp3_decoder_destroy(sk_mp3_decoder *d) try {
    sk::abi_enter();
    if (!d) return;
    if (d->stream_open) (void)sk_stream_close(d->engine, d->stream);
    delete d->own_cb;
    delete d;
} catch (...) {
    (void)sk::abi_caught("sk_mp3_decoder_destroy");
}

int sk_mp3_decoder_reset(sk_mp3_decoder *d) try {
    sk::abi_enter();
    if (!d) return SK_ERR_INVALID_ARG;
    if (d->stream_open) (void)sk_stream_close(d->engine, d->stream);
    d->stream_open = false;
    d->buffer.clear(), d->reservoir.clear();
    d->sample_rate = 0, d->channels = 0, d->free_format_bytes = 0;
    return SK_OK;
} catch (...) {
    return sk::abi_caught("sk_mp3_decoder_reset");
}

int sk_mp3_decoder_info(const sk_mp3_decoder *d, uint32_t *sample_rate, uint8_t *channels, size_t *buffer_len, uint64_t *frames) try {
    sk::abi_enter();
    if (!d) return SK_ERR_INVALID_ARG;
    if (sample_rate) *sample_rate = d->sample_rate;
    if (channels) *channels = d->channels;
    if (buffer_len) *buffer_len = d->buffer.size();
    if (frames) *frames = d->frames;
    return SK_OK;
} catch (...) {
    return sk::abi_caught("sk_mp3_decoder_info");
}

int sk_mp3_decoder_decode_i16(sk_mp3_decoder *d, const uint8_t *input, size_t len, int16_t *out, size_t out_cap, size_t *written) try {
    sk::abi_enter();
    return decode(d, input, len, out, out_cap, written, Out::I16);
} catch (...) {
    return sk::abi_caught("sk_mp3_decoder_decode_i16");
}
int sk_mp3_decoder_decode_i32(sk_mp3_decoder *d, const uint8_t *input, size_t len, int32_t *out, size_t out_cap, size_t *written) try {
    sk::abi_enter();
    return decode(d, input, len, out, out_cap, written, Out::I32);
} catch (...) {
    return sk::abi_caught("sk_mp3_decoder_decode_i32");
}
int sk_mp3_decoder_decode_f32(sk_mp3_decoder *d, const uint8_t *input, size_t len, float *out, size_t out_cap, size_t *written) try {
    sk::abi_enter();
    return decode(d, input, len, out, out_cap, written, Out::F32);
} catch (...) {
    return sk::abi_caught("sk_mp3_decoder_decode_f32");
}

}  // extern "C"
