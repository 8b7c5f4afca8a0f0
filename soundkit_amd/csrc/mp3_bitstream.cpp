// mp3_bitstream.cpp -- the fixed-syntax front of an MPEG-1 / MPEG-2 (LSF) / MPEG-2.5 Layer III decoder: frame sync and
// header, side information, the main-data bit reservoir.  Host code: what nanomp3::Decoder::decode does before its Huffman
// stage (soundkit-mp3/src/lib.rs:284 calls it; the crate is not in the reference tree).  Everything here is closed-form
// syntax of ISO/IEC 11172-3 2.4.1-2.4.2 and 13818-3 2.4.1-2.4.2 -- fixed bit fields and two short index tables of the header
// (bit rate, sampling rate).  The Huffman tables (B.7), the scale-factor band tables (B.8) and the synthesis window (B.3)
// are not part of this file or this tree: DESIGN.md section 7.  Parity of the MP3 row is unpinned; what pins this file is
// the reference's two MP3 fixtures, whose frames must chain exactly and whose side information must add up
// (tests/test_mp3_bitstream.py).
#include "../../include/soundkit_amd.h"
#include "sk_abi.h"

#include <cstring>

namespace {

// header index tables (11172-3 2.4.2.3, 13818-3 2.4.2.3); kbit/s, index 0 = free format, 15 = forbidden
const uint16_t kBitrateV1L3[16] = {0, 32, 40, 48, 56, 64, 80, 96, 112, 128, 160, 192, 224, 256, 320, 0};
const uint16_t kBitrateV2L3[16] = {0, 8, 16, 24, 32, 40, 48, 56, 64, 80, 96, 112, 128, 144, 160, 0};
const uint32_t kSampleRate[3][3] = {{44100, 48000, 32000}, {22050, 24000, 16000}, {11025, 12000, 8000}};  // MPEG-1, -2, -2.5

struct Mp3HeaderBits {
    const uint8_t *p;
    size_t len;
    size_t pos = 0;  // in bits
    uint32_t get(int n) {
        uint32_t v = 0;
        for (int i = 0; i < n; ++i) {
            const size_t byte = pos >> 3;
            const uint32_t bit = byte < len ? (p[byte] >> (7 - (pos & 7))) & 1u : 0u;
            v = (v << 1) | bit;
            ++pos;
        }
        return v;
    }
};


// free_format_bytes: the length (without the padding slot) of the stream's free-format frames once sk_mp3_scan_free has found
// it, 0 = not known: a free-format header (bit-rate index 0: the length is not in the header) is then SK_MP3_UNSUPPORTED
int parse_header(const uint8_t *d, size_t len, uint32_t free_format_bytes, sk_mp3_frame_info *out) {
    if (!d || !out) return SK_ERR_INVALID_ARG;
    if (len < 4) return SK_MP3_NEED_MORE;
    if (d[0] != 0xff || (d[1] & 0xe0) != 0xe0) return SK_MP3_NO_SYNC;
    const unsigned version_bits = (d[1] >> 3) & 3;  // 00 MPEG-2.5, 01 reserved, 10 MPEG-2, 11 MPEG-1
    const unsigned layer_bits = (d[1] >> 1) & 3;    // 01 = Layer III
    if (version_bits == 1) return SK_MP3_NO_SYNC;
    if (layer_bits != 1) return layer_bits == 0 ? SK_MP3_NO_SYNC : SK_MP3_UNSUPPORTED;  // Layers I / II are not this path
    const unsigned bitrate_index = d[2] >> 4, sr_index = (d[2] >> 2) & 3;
    if (bitrate_index == 15 || sr_index == 3) return SK_MP3_NO_SYNC;
    if (bitrate_index == 0 && free_format_bytes == 0) return SK_MP3_UNSUPPORTED;  // free format, length not known (yet)
    const int v = version_bits == 3 ? 0 : (version_bits == 2 ? 1 : 2);
    std::memset(out, 0, sizeof *out);
    out->version = version_bits == 3 ? 1 : (version_bits == 2 ? 2 : 25);
    out->has_crc = (d[1] & 1) ? 0 : 1;
    out->sample_rate = kSampleRate[v][sr_index];
    out->padding = (d[2] >> 1) & 1;
    out->mode = d[3] >> 6;  // 0 stereo, 1 joint stereo, 2 dual channel, 3 single channel
    out->mode_ext = (d[3] >> 4) & 3;  // joint stereo: bit 1 = mid/side, bit 0 = intensity
    out->channels = out->mode == 3 ? 1 : 2;
    out->granules = v == 0 ? 2 : 1;
    out->samples_per_channel = v == 0 ? 1152 : 576;
    if (bitrate_index == 0) {
        // free format: every frame of the stream has the length the scan measured between two headers (plus its own padding slot);
        // the bit rate follows from it
        out->frame_bytes = free_format_bytes + out->padding;
        out->bitrate_kbps = (uint16_t)((uint64_t)free_format_bytes * out->sample_rate / ((v == 0 ? 144u : 72u) * 1000u));
    } else {
        out->bitrate_kbps = v == 0 ? kBitrateV1L3[bitrate_index] : kBitrateV2L3[bitrate_index];
        // Layer III: 144 * bitrate / fs bytes for MPEG-1, 72 * bitrate / fs for the lower sampling frequencies, plus the padding slot
        out->frame_bytes = (uint32_t)((v == 0 ? 144u : 72u) * (uint32_t)out->bitrate_kbps * 1000u / out->sample_rate + out->padding);
    }
    out->side_info_bytes = v == 0 ? (out->channels == 1 ? 17 : 32) : (out->channels == 1 ? 9 : 17);
    if (out->frame_bytes < 4u + (out->has_crc ? 2u : 0u) + out->side_info_bytes) return SK_MP3_NO_SYNC;
    return SK_OK;
}

bool is_free_format(const uint8_t *d) { return (d[2] >> 4) == 0; }
// two headers of one stream (minimp3's hdr_compare): version, layer and sampling rate agree and both are free format or neither is
bool same_stream(const uint8_t *a, const uint8_t *b) {
    return b[0] == 0xff && ((a[1] ^ b[1]) & 0xfe) == 0 && ((a[2] ^ b[2]) & 0x0c) == 0 && is_free_format(a) == is_free_format(b);
}
constexpr size_t kMaxFreeFormatFrame = 2304;  // minimp3's MAX_FREE_FORMAT_FRAME_SIZE

int scan(const uint8_t *d, size_t len, sk_mp3_frame_info *frames, uint32_t cap, uint32_t *n_frames, size_t *consumed, uint32_t *free_format_bytes) {
    if (!d || !n_frames || (cap && !frames)) return SK_ERR_INVALID_ARG;
    *n_frames = 0;
    uint32_t ffb = free_format_bytes ? *free_format_bytes : 0;
    size_t pos = 0;
    if (len >= 10 && d[0] == 'I' && d[1] == 'D' && d[2] == '3' && !((d[6] | d[7] | d[8] | d[9]) & 0x80)) {
        const size_t tag = 10u + (((size_t)d[6] << 21) | ((size_t)d[7] << 14) | ((size_t)d[8] << 7) | d[9]) + ((d[5] & 0x10) ? 10u : 0u);
        if (tag <= len) pos = tag;
    }
    while (pos + 4 <= len) {
        sk_mp3_frame_info h;
        int rc = parse_header(d + pos, len - pos, ffb, &h);
        if (rc == SK_MP3_UNSUPPORTED && free_format_bytes && is_free_format(d + pos) && parse_header(d + pos, len - pos, 4096, &h) == SK_OK) {
            // A free-format header and no length yet: the frame ends where a header of the same stream stands, and the frame after it
            // ends with such a header too (mp3d_find_frame).  Until both are in the buffer the answer waits for more input.
            const uint8_t *here = d + pos;
            bool found = false, wait = false;
            for (size_t k = 4; k < kMaxFreeFormatFrame; ++k) {
                if (pos + k + 4 > len) {
                    wait = true;
                    break;
                }
                if (!same_stream(here, here + k)) continue;
                const size_t fb = k - ((here[2] >> 1) & 1), next_fb = fb + ((here[k + 2] >> 1) & 1);
                if (pos + k + next_fb + 4 > len) {
                    wait = true;
                    break;
                }
                if (!same_stream(here, here + k + next_fb)) continue;
                ffb = (uint32_t)fb;
                found = true;
                break;
            }
            if (wait) break;
            if (found) rc = parse_header(d + pos, len - pos, ffb, &h);
        }
        if (rc != SK_OK) {
            ++pos;
            continue;
        }
        const size_t next = pos + h.frame_bytes;
        if (next > len) break;  // an incomplete frame at the end: needs more input
        if (next + 4 <= len) {
            sk_mp3_frame_info follow;
            const int frc = parse_header(d + next, len - next, ffb, &follow);
            if (frc != SK_OK || follow.version != h.version || follow.sample_rate != h.sample_rate) {
                if (is_free_format(d + pos)) ffb = 0;  // not the stream's length after all: measure again at the next candidate
                ++pos;
                continue;
            }
        }
        h.offset = (uint32_t)pos;
        if (*n_frames < cap) frames[*n_frames] = h;
        *n_frames += 1;
        pos = next;
    }
    if (consumed) *consumed = pos;
    if (free_format_bytes) *free_format_bytes = ffb;
    return SK_OK;
}

}  // namespace

extern "C" {

int sk_mp3_parse_header(const uint8_t *d, size_t len, sk_mp3_frame_info *out) try {
    sk::abi_enter();
    return parse_header(d, len, 0, out);
} catch (...) {
    return sk::abi_caught("sk_mp3_parse_header");
}

int sk_mp3_parse_header_free(const uint8_t *d, size_t len, uint32_t free_format_bytes, sk_mp3_frame_info *out) try {
    sk::abi_enter();
    return parse_header(d, len, free_format_bytes, out);
} catch (...) {
    return sk::abi_caught("sk_mp3_parse_header_free");
}

int sk_mp3_parse_side_info(const uint8_t *frame, size_t len, const sk_mp3_frame_info *h, sk_mp3_side_info *out) try {
    sk::abi_enter();
    if (!frame || !h || !out) return SK_ERR_INVALID_ARG;
    const size_t at = 4u + (h->has_crc ? 2u : 0u);
    if (len < at + h->side_info_bytes) return SK_MP3_NEED_MORE;
    std::memset(out, 0, sizeof *out);
    Mp3HeaderBits b{frame + at, h->side_info_bytes};
    const bool v1 = h->version == 1;
    const int ch = h->channels;
    out->granules = h->granules;
    out->channels = h->channels;
    if (v1) {
        out->main_data_begin = (uint16_t)b.get(9);
        (void)b.get(ch == 1 ? 5 : 3);  // private bits
        for (int c = 0; c < ch; ++c)
            for (int band = 0; band < 4; ++band) out->scfsi[c][band] = (uint8_t)b.get(1);
    } else {
        out->main_data_begin = (uint16_t)b.get(8);
        (void)b.get(ch == 1 ? 1 : 2);
    }
    for (int g = 0; g < h->granules; ++g)
        for (int c = 0; c < ch; ++c) {
            sk_mp3_granule_side &s = out->gr[g][c];
            s.part2_3_length = (uint16_t)b.get(12);
            s.big_values = (uint16_t)b.get(9);
            s.global_gain = (uint8_t)b.get(8);
            s.scalefac_compress = (uint16_t)b.get(v1 ? 4 : 9);
            s.window_switching = (uint8_t)b.get(1);
            if (s.window_switching) {
                s.block_type = (uint8_t)b.get(2);
                s.mixed_block_flag = (uint8_t)b.get(1);
                for (int r = 0; r < 2; ++r) s.table_select[r] = (uint8_t)b.get(5);
                for (int w = 0; w < 3; ++w) s.subblock_gain[w] = (uint8_t)b.get(3);
                // implicit region split (2.4.2.7): region0 ends after 8 long bands (block types 1, 3, mixed) or 9 short ones
                s.region0_count = (s.block_type == 2 && !s.mixed_block_flag) ? 8 : 7;
                s.region1_count = 36;  // "the rest": regions 1 and 2 are not told apart by a count here
                if (s.block_type == 0) return SK_MP3_INVALID;  // window switching with block type 0 is forbidden
            } else {
                for (int r = 0; r < 3; ++r) s.table_select[r] = (uint8_t)b.get(5);
                s.region0_count = (uint8_t)b.get(4);
                s.region1_count = (uint8_t)b.get(3);
            }
            if (v1) s.preflag = (uint8_t)b.get(1);
            s.scalefac_scale = (uint8_t)b.get(1);
            s.count1table_select = (uint8_t)b.get(1);
            if (s.big_values > 288) return SK_MP3_INVALID;  // 2 x big_values lines must fit the 576 of a granule
        }
    return SK_OK;
} catch (...) {
    return sk::abi_caught("sk_mp3_parse_side_info");
}

// Frames of a byte stream: an ID3v2 tag in front is stepped over (its length is in the tag), bytes that are no header are
// skipped one at a time, and a candidate header counts as a frame only if the next frame's header follows it where its
// length says (or the data ends there) -- the usual guard against sync words inside audio data.
int sk_mp3_scan(const uint8_t *d, size_t len, sk_mp3_frame_info *frames, uint32_t cap, uint32_t *n_frames, size_t *consumed) try {
    sk::abi_enter();
    return scan(d, len, frames, cap, n_frames, consumed, nullptr);
} catch (...) {
    return sk::abi_caught("sk_mp3_scan");
}

// The same with free-format streams (bit-rate index 0: the frame length is measured between headers, as minimp3's mp3d_find_frame
// does).  *free_format_bytes carries the measured length from call to call (0 at the start of a stream) -- the last frames of a
// stream have no two headers behind them to measure it again.
int sk_mp3_scan_free(const uint8_t *d, size_t len, sk_mp3_frame_info *frames, uint32_t cap, uint32_t *n_frames, size_t *consumed,
                     uint32_t *free_format_bytes) try {
    sk::abi_enter();
    if (!free_format_bytes) return SK_ERR_INVALID_ARG;
    return scan(d, len, frames, cap, n_frames, consumed, free_format_bytes);
} catch (...) {
    return sk::abi_caught("sk_mp3_scan_free");
}

// The main data of frame k starts main_data_begin bytes BEFORE its own main-data area, in what earlier frames left unused
// (the bit reservoir, 2.4.2.7): assembles the bytes parts 2 + 3 of frame k are read from.  prev / prev_len: the main-data
// bytes of the frames before it, oldest first (at least main_data_begin of them, else SK_MP3_NEED_MORE: a stream joined in
// the middle).  Returns the number of bytes written to out.
int sk_mp3_main_data(const uint8_t *frame, size_t frame_len, const sk_mp3_frame_info *h, const sk_mp3_side_info *side, const uint8_t *prev,
                     size_t prev_len, uint8_t *out, size_t out_cap, size_t *out_len) try {
    sk::abi_enter();
    if (!frame || !h || !side || !out || !out_len || (prev_len && !prev)) return SK_ERR_INVALID_ARG;
    const size_t head = 4u + (h->has_crc ? 2u : 0u) + h->side_info_bytes;
    if (frame_len < h->frame_bytes || h->frame_bytes < head) return SK_MP3_NEED_MORE;
    const size_t own = h->frame_bytes - head, back = side->main_data_begin;
    if (back > prev_len) return SK_MP3_NEED_MORE;
    if (back + own > out_cap) return SK_ERR_CAPACITY;
    if (back) std::memcpy(out, prev + prev_len - back, back);
    std::memcpy(out + back, frame + head, own);
    *out_len = back + own;
    // parts 2 + 3 of all granules and channels must fit in what is there
    size_t bits = 0;
    for (int g = 0; g < side->granules; ++g)
        for (int c = 0; c < side->channels; ++c) bits += side->gr[g][c].part2_3_length;
    if (bits > 8 * (back + own)) return SK_MP3_INVALID;
    return SK_OK;
} catch (...) {
    return sk::abi_caught("sk_mp3_main_data");
}

}  // extern "C"
