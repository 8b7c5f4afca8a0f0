"""oracle/mp3_bitstream.py -- CPU restatement (pure Python / numpy f64) of the fixed-syntax front of a Layer III decoder
and of requantisation, joint stereo and the short-block reorder.

TEST INFRASTRUCTURE ONLY; nothing under soundkit_amd/ imports it.

PARITY UNPINNED for decoded samples (see oracle/mp3_hybrid.py: the decoder is the third-party crate `nanomp3`, reached at
soundkit-mp3/src/lib.rs:284, its source is not in the reference tree).  What pins the FRAMING half of this file is the
reference's own MP3 fixtures (testdata/mp3 and golden/mp3, copied as data to tests/golden/mp3/): the reference's tests
decode the mono one as 16 kHz, 1 channel (soundkit-mp3/src/lib.rs:551-552) and write the stereo one from its 16 kHz
stereo WAV (lib.rs:482-518), so their frames must chain from the
first header to the end of the file with exactly those parameters, and the side information of every frame must add up
to what the frame and the bit reservoir hold.

Restated from ISO/IEC 11172-3:1993 and 13818-3:
  header            2.4.1.3 / 2.4.2.3: 11(+1) sync bits, version, layer, protection, bitrate index, sampling-rate index,
                    padding, mode, mode extension; Layer III frame length 144 (MPEG-1) or 72 (LSF) x bitrate / fs + padding
  side information  2.4.1.7: main_data_begin 9 (8) bits, private bits, scfsi, and per granule / channel part2_3_length 12,
                    big_values 9, global_gain 8, scalefac_compress 4 (9), window_switching_flag 1, then either block_type 2,
                    mixed_block_flag 1, 2 x table_select 5, 3 x subblock_gain 3 or 3 x table_select 5, region0_count 4,
                    region1_count 3; preflag 1 (MPEG-1 only), scalefac_scale 1, count1table_select 1
  bit reservoir     2.4.2.7: main data starts main_data_begin bytes before the frame's own main-data area
  requantisation    2.4.3.4.7.1; stereo 2.4.3.4.9; reorder 2.4.3.4.8 (formulas at the functions)
"""
import math

import numpy as np

BITRATE_V1 = [0, 32, 40, 48, 56, 64, 80, 96, 112, 128, 160, 192, 224, 256, 320]
BITRATE_V2 = [0, 8, 16, 24, 32, 40, 48, 56, 64, 80, 96, 112, 128, 144, 160]
SAMPLE_RATES = {3: [44100, 48000, 32000], 2: [22050, 24000, 16000], 0: [11025, 12000, 8000]}  # by the two version bits


class Bits:
    def __init__(self, data):
        self.data, self.pos = data, 0

    def take(self, n):
        v = 0
        for _ in range(n):
            byte = self.data[self.pos >> 3]
            v = (v << 1) | ((byte >> (7 - (self.pos & 7))) & 1)
            self.pos += 1
        return v


def parse_header(b, free_format_bytes=0):
    """4 bytes -> dict, or None if they are no Layer III frame header.  A free-format header (bit-rate index 0) is one only when the
    stream's frame length (without the padding slot) is given"""
    if len(b) < 4 or b[0] != 0xFF or (b[1] & 0xE0) != 0xE0:
        return None
    version_bits, layer_bits = (b[1] >> 3) & 3, (b[1] >> 1) & 3
    if version_bits == 1 or layer_bits != 1:
        return None
    bitrate_index, rate_index = b[2] >> 4, (b[2] >> 2) & 3
    if bitrate_index == 15 or rate_index == 3 or (bitrate_index == 0 and not free_format_bytes):
        return None
    mpeg1 = version_bits == 3
    rate = SAMPLE_RATES[version_bits][rate_index]
    padding = (b[2] >> 1) & 1
    if bitrate_index:
        kbps = (BITRATE_V1 if mpeg1 else BITRATE_V2)[bitrate_index]
        frame_bytes = (144 if mpeg1 else 72) * kbps * 1000 // rate + padding
    else:
        kbps = free_format_bytes * rate // ((144 if mpeg1 else 72) * 1000)
        frame_bytes = free_format_bytes + padding
    mode = b[3] >> 6
    channels = 1 if mode == 3 else 2
    side_info_bytes = (17 if channels == 1 else 32) if mpeg1 else (9 if channels == 1 else 17)
    if frame_bytes < 4 + (0 if b[1] & 1 else 2) + side_info_bytes:
        return None
    return {
        "version": 1 if mpeg1 else (2 if version_bits == 2 else 25),
        "has_crc": 0 if b[1] & 1 else 1,
        "bitrate_kbps": kbps,
        "sample_rate": rate,
        "padding": padding,
        "mode": mode,
        "mode_ext": (b[3] >> 4) & 3,
        "channels": channels,
        "granules": 2 if mpeg1 else 1,
        "samples_per_channel": 1152 if mpeg1 else 576,
        "frame_bytes": frame_bytes,
        "side_info_bytes": side_info_bytes,
    }


def parse_side_info(frame, h):
    at = 4 + (2 if h["has_crc"] else 0)
    bits = Bits(frame[at:at + h["side_info_bytes"]])
    mpeg1, ch = h["version"] == 1, h["channels"]
    out = {"scfsi": [[0] * 4 for _ in range(2)]}
    if mpeg1:
        out["main_data_begin"] = bits.take(9)
        bits.take(5 if ch == 1 else 3)
        for c in range(ch):
            out["scfsi"][c] = [bits.take(1) for _ in range(4)]
    else:
        out["main_data_begin"] = bits.take(8)
        bits.take(1 if ch == 1 else 2)
    out["gr"] = []
    for _ in range(h["granules"]):
        row = []
        for _ in range(ch):
            s = {"part2_3_length": bits.take(12), "big_values": bits.take(9), "global_gain": bits.take(8),
                 "scalefac_compress": bits.take(4 if mpeg1 else 9), "window_switching": bits.take(1),
                 "block_type": 0, "mixed_block_flag": 0, "subblock_gain": [0, 0, 0], "table_select": [0, 0, 0], "preflag": 0}
            if s["window_switching"]:
                s["block_type"] = bits.take(2)
                s["mixed_block_flag"] = bits.take(1)
                s["table_select"][:2] = [bits.take(5), bits.take(5)]
                s["subblock_gain"] = [bits.take(3) for _ in range(3)]
                s["region0_count"] = 8 if (s["block_type"] == 2 and not s["mixed_block_flag"]) else 7
                s["region1_count"] = 36
            else:
                s["table_select"] = [bits.take(5) for _ in range(3)]
                s["region0_count"] = bits.take(4)
                s["region1_count"] = bits.take(3)
            if mpeg1:
                s["preflag"] = bits.take(1)
            s["scalefac_scale"] = bits.take(1)
            s["count1table_select"] = bits.take(1)
            row.append(s)
        out["gr"].append(row)
    assert bits.pos == 8 * h["side_info_bytes"], "the side information fills its bytes exactly"
    return out


def _same_stream(a, b):
    """minimp3 hdr_compare: version, layer, sampling rate agree; both free format or neither"""
    return b[0] == 0xFF and (a[1] ^ b[1]) & 0xFE == 0 and (a[2] ^ b[2]) & 0x0C == 0 and ((a[2] >> 4) == 0) == ((b[2] >> 4) == 0)


def scan(data, free_format_bytes=None):
    """every frame of a byte string: ([(offset, header)], bytes consumed).  With free_format_bytes (a one-element list: the state
    carried between calls, [0] at the start of a stream) free-format frames are found too: their length is what lies between a header
    and the next two of the same stream (minimp3 mp3d_find_frame; MAX_FREE_FORMAT_FRAME_SIZE 2304), measured once per stream."""
    pos = 0
    if len(data) >= 10 and data[:3] == b"ID3":
        pos = 10 + ((data[6] & 0x7F) << 21 | (data[7] & 0x7F) << 14 | (data[8] & 0x7F) << 7 | (data[9] & 0x7F))
        if data[5] & 0x10:
            pos += 10
    frames = []
    ffb = free_format_bytes[0] if free_format_bytes else 0
    while pos + 4 <= len(data):
        here = data[pos:pos + 4]
        h = parse_header(here, ffb)
        if h is None and free_format_bytes is not None and not ffb and (here[2] >> 4) == 0 and parse_header(here, 4096) is not None:
            wait = False
            for k in range(4, 2304):
                if pos + k + 4 > len(data):
                    wait = True
                    break
                if not _same_stream(here, data[pos + k:pos + k + 4]):
                    continue
                fb = k - ((here[2] >> 1) & 1)
                nxt_fb = fb + ((data[pos + k + 2] >> 1) & 1)
                if pos + k + nxt_fb + 4 > len(data):
                    wait = True
                    break
                if _same_stream(here, data[pos + k + nxt_fb:pos + k + nxt_fb + 4]):
                    ffb = fb
                    h = parse_header(here, ffb)
                    break
            if wait:
                break
        if h is None:
            pos += 1
            continue
        nxt = pos + h["frame_bytes"]
        if nxt > len(data):
            break
        if nxt + 4 <= len(data):
            follow = parse_header(data[nxt:nxt + 4], ffb)
            if follow is None or follow["version"] != h["version"] or follow["sample_rate"] != h["sample_rate"]:
                if (here[2] >> 4) == 0:
                    ffb = 0
                pos += 1
                continue
        frames.append((pos, h))
        pos = nxt
    if free_format_bytes is not None:
        free_format_bytes[0] = ffb
    return frames, pos


def main_data(frames, data):
    """the reservoir walk: per frame, the bytes its parts 2 + 3 are read from (None when the reservoir does not reach back)"""
    kept, out = b"", []
    for off, h in frames:
        frame = data[off:off + h["frame_bytes"]]
        side = parse_side_info(frame, h)
        head = 4 + (2 if h["has_crc"] else 0) + h["side_info_bytes"]
        back = side["main_data_begin"]
        own = frame[head:]
        out.append((kept[len(kept) - back:] if back else b"") + own if back <= len(kept) else None)
        kept = (kept + own)[-1024:]
    return out


# ---- requantisation / stereo / reorder ---------------------------------------------------------------------------------

def locate(i, short_lines, long_offsets, short_offsets):
    """bitstream-order line i -> (band, window or -1, position after the reorder)"""
    if not short_lines:
        band = max(l for l in range(22) if long_offsets[l] <= i)
        return band, -1, i
    band = max(s for s in range(13) if 3 * short_offsets[s] <= i)
    begin, width = short_offsets[band], short_offsets[band + 1] - short_offsets[band]
    w, j = divmod(i - 3 * begin, width)
    return band, w, 3 * (begin + j) + w


def requantize_granule(g, quant, long_offsets, short_offsets, pretab):
    """g: dict(channels, ms_stereo, intensity_stereo, ch=[dict(global_gain, scalefac_scale, preflag, block_type,
    mixed_block_flag, subblock_gain[3], scalefac_l[22], scalefac_s[13][3])]); quant [channels][576] integers in bitstream
    order -> xr [channels][576] f64 in the order the hybrid synthesis reads"""
    channels = g["channels"]
    ms_stereo, intensity_stereo = g.get("ms_stereo", 0), g.get("intensity_stereo", 0)
    vals = np.zeros((channels, 576))
    where = [[None] * 576 for _ in range(channels)]
    for c in range(channels):
        ch = g["ch"][c]
        mult = 1.0 if ch["scalefac_scale"] else 0.5
        for i in range(576):
            short_lines = ch["block_type"] == 2 and not (ch["mixed_block_flag"] and i < 36)
            band, w, dest = locate(i, short_lines, long_offsets, short_offsets)
            where[c][i] = (band, w, dest)
            q = int(quant[c][i])
            if w < 0:
                exponent = (ch["global_gain"] - 210) / 4.0 - mult * ((ch["scalefac_l"][band] & 0x7F) + (int(pretab[band]) if ch["preflag"] else 0))
            else:
                exponent = (ch["global_gain"] - 210 - 8 * ch["subblock_gain"][w]) / 4.0 - mult * (ch["scalefac_s"][band][w] & 0x7F)
            vals[c][i] = math.copysign(abs(q) ** (4.0 / 3.0) * 2.0 ** exponent, q) if q else 0.0
    if channels == 2 and (ms_stereo or intensity_stereo):
        # Which bands are intensity coded, as minimp3 decides it (nanomp3, the reference's decoder, is its port; minimp3.h
        # L3_stereo_top_band, L3_intensity_stereo, L3_stereo_process): number the bands in the order of their scale factors --
        # long bands, then short bands band by band and window by window (a mixed granule: its long bands, then its short bands;
        # numbered from 64 here, only their order matters) -- and let bound[r % 3] be the highest number whose band holds a
        # non-zero line of the right channel.  Long and
        # mixed granules use max(bound) for all bands, short granules bound[window].  Band r is intensity coded iff
        # r > bound and its position is legal.  The last band has no factor: it takes the position of the band below in its
        # window if THAT one is above the bound, else 3 (MPEG-1) / 0 (13818-3) -- both channels alike.
        lsf = bool(g.get("lsf"))
        left_ch = g["ch"][0]
        kind = (2 if left_ch["mixed_block_flag"] else 1) if left_ch["block_type"] == 2 else 0

        def number(band, w):
            band, w = int(band), int(w)
            if w < 0:
                return band
            return (64 if kind == 2 else 0) + 3 * band + w
        bound = [-1, -1, -1]
        for i in range(576):
            band, w, _ = where[1][i]
            if quant[1][i] != 0:
                r = number(band, w)
                bound[r % 3] = max(bound[r % 3], r)
        if kind != 1:
            bound = [max(bound)] * 3
        right = g["ch"][1]
        for i in range(576):
            band, w, _ = where[0][i]
            done = False
            r = number(band, w)
            if intensity_stereo and r > bound[r % 3]:
                last = band >= (21 if w < 0 else 12)
                below = r - (1 if w < 0 else 3)
                if last and bound[r % 3] >= below:
                    pos = 0 if lsf else 3
                else:
                    pos = right["scalefac_l"][min(band, 20)] if w < 0 else right["scalefac_s"][min(band, 11)][w]
                if lsf:
                    # ISO/IEC 13818-3 2.4.3.2: is_pos 0: both channels take the line; odd: left scaled by i0^((is_pos + 1) / 2),
                    # even: right by i0^(is_pos / 2); i0 = 2^-1/4, or 2^-1/2 when intensity_scale (intensity_stereo bit 1) is set
                    if not pos & 0x80:
                        steps = ((pos + 1) >> 1) << (1 if intensity_stereo & 2 else 0)
                        f = 2.0 ** (-steps / 4.0)
                        x = vals[0][i]
                        vals[0][i], vals[1][i] = (x * f, x) if pos & 1 else (x, x * f)
                        done = True
                elif pos < 7:
                    if pos == 6:
                        kl = 1.0
                    else:
                        t = math.tan(pos * math.pi / 12)
                        kl = t / (1 + t)
                    x = vals[0][i]
                    vals[0][i], vals[1][i] = x * kl, x * (1 - kl)
                    done = True
            if not done and ms_stereo:
                m, s = vals[0][i], vals[1][i]
                vals[0][i], vals[1][i] = (m + s) / math.sqrt(2), (m - s) / math.sqrt(2)
    out = np.zeros((channels, 576))
    for c in range(channels):
        for i in range(576):
            out[c][where[c][i][2]] = vals[c][i]
    return out


# ---- parts 2 and 3 of the main data: scale factors and the Huffman stage, over caller-supplied tables --------------------
# `tables` is a dict in the shape of sk_mp3_tables: big_values[32] = None | dict(xlen, linbits, hlen[], hcod[]),
# count1[2] = dict(hlen[16], hcod[16]), slen[16][2], lsf_partitions[6][3][4], bands{rate: (long23, short14)}, pretab[22],
# window[512].  Codes are matched by reading bits until (length, value) is a code of the table.

def _code_map(hlen, hcod):
    return {(int(n), int(c)): s for s, (n, c) in enumerate(zip(hlen, hcod))}


def _read_code(bits, code_map):
    value = 0
    for n in range(1, 33):
        value = (value << 1) | bits.take1()
        if (n, value) in code_map:
            return code_map[(n, value)]
    return None


class MainBits:
    """bit reader that returns zeros past the end and keeps counting (like the product's)"""

    def __init__(self, data, pos=0):
        self.data, self.pos = data, pos

    def take1(self):
        byte = self.pos >> 3
        v = (self.data[byte] >> (7 - (self.pos & 7))) & 1 if byte < len(self.data) else 0
        self.pos += 1
        return v

    def take(self, n):
        v = 0
        for _ in range(n):
            v = (v << 1) | self.take1()
        return v


def scale_factors(tables, h, side, gr, ch, bits, first_granule):
    """-> (scalefac_l[22], scalefac_s[13][3], preflag); first_granule: granule 0's scalefac_l of this channel (scfsi)"""
    s = side["gr"][gr][ch]
    sl, ss = [0] * 22, [[0, 0, 0] for _ in range(13)]
    short = s["window_switching"] and s["block_type"] == 2
    if h["version"] == 1:
        slen1, slen2 = tables["slen"][s["scalefac_compress"]]
        if short:
            first = 0
            if s["mixed_block_flag"]:
                for band in range(8):
                    sl[band] = bits.take(slen1)
                first = 3
            for band in range(first, 12):
                for w in range(3):
                    ss[band][w] = bits.take(slen1 if band < 6 else slen2)
        else:
            for group, (a, b) in enumerate(((0, 6), (6, 11), (11, 16), (16, 21))):
                for band in range(a, b):
                    if gr == 1 and side["scfsi"][ch][group]:
                        sl[band] = first_granule[band]
                    else:
                        sl[band] = bits.take(slen1 if group < 2 else slen2)
        return sl, ss, s["preflag"]
    sfc, preflag = s["scalefac_compress"], 0
    if not (h["mode"] == 1 and (h["mode_ext"] & 1) and ch == 1):
        if sfc < 400:
            lens, row = [(sfc >> 4) // 5, (sfc >> 4) % 5, (sfc % 16) >> 2, sfc % 4], 0
        elif sfc < 500:
            sfc -= 400
            lens, row = [(sfc >> 2) // 5, (sfc >> 2) % 5, sfc % 4, 0], 1
        else:
            sfc -= 500
            lens, row, preflag = [sfc // 3, sfc % 3, 0, 0], 2, 1
    else:
        sfc >>= 1
        if sfc < 180:
            lens, row = [sfc // 36, (sfc % 36) // 6, (sfc % 36) % 6, 0], 3
        elif sfc < 244:
            sfc -= 180
            lens, row = [(sfc % 64) >> 4, (sfc % 16) >> 2, sfc % 4, 0], 4
        else:
            sfc -= 244
            lens, row = [sfc // 3, sfc % 3, 0, 0], 5
    column = (2 if s["mixed_block_flag"] else 1) if short else 0
    values = []
    for part, count in enumerate(tables["lsf_partitions"][row][column]):
        for _ in range(count):
            v = bits.take(lens[part])
            # 13818-3 2.4.3.2: in the intensity channel the largest value of a field means "not intensity coded" (bit 7 here)
            values.append(v | 0x80 if row >= 3 and lens[part] > 0 and v == (1 << lens[part]) - 1 else v)
    if column == 0:
        sl[:len(values)] = values
    else:
        n_long = 6 if column == 2 else 0
        sl[:n_long] = values[:n_long]
        for k, v in enumerate(values[n_long:]):
            band, w = divmod(k + (9 if column == 2 else 0), 3)
            ss[band][w] = v
    return sl, ss, preflag


def region_bounds(s, long_offsets, short_offsets):
    """line numbers where regions 1 and 2 of the big values begin"""
    short = s["window_switching"] and s["block_type"] == 2
    if not short:
        widths = [long_offsets[b + 1] - long_offsets[b] for b in range(22)]
    else:
        widths, first = [], 0
        if s["mixed_block_flag"]:
            widths = [long_offsets[b + 1] - long_offsets[b] for b in range(22) if long_offsets[b + 1] <= 36]
            first = min(b for b in range(14) if 3 * short_offsets[b] >= 36)
        for band in range(first, 13):
            widths += [short_offsets[band + 1] - short_offsets[band]] * 3
    r1 = min(576, sum(widths[:s["region0_count"] + 1]))
    r2 = 576 if s["window_switching"] else min(576, sum(widths[:s["region0_count"] + s["region1_count"] + 2]))
    return r1, r2


def huffman_granule(tables, h, s, bits, end):
    """-> 576 integers or None (a bit pattern that is no code, or an overrun)"""
    long_offsets, short_offsets = tables["bands"][h["sample_rate"]]
    out = [0] * 576
    big_end = 2 * s["big_values"]
    if big_end > 576:
        return None
    r1, r2 = region_bounds(s, long_offsets, short_offsets)
    bounds = [0, min(r1, big_end), min(r2, big_end), big_end]
    line = 0
    for region in range(3):
        select = s["table_select"][region]
        table = tables["big_values"][select]
        code_map = _code_map(table["hlen"], table["hcod"]) if table else None
        while line < bounds[region + 1]:
            x = y = 0
            if table:
                symbol = _read_code(bits, code_map)
                if symbol is None:
                    return None
                x, y = divmod(symbol, table["xlen"])
                if table["linbits"] and x == table["xlen"] - 1:
                    x += bits.take(table["linbits"])
                if x and bits.take1():
                    x = -x
                if table["linbits"] and y == table["xlen"] - 1:
                    y += bits.take(table["linbits"])
                if y and bits.take1():
                    y = -y
            elif select != 0:
                return None
            out[line], out[line + 1] = x, y
            line += 2
    if bits.pos > end:
        return None
    quad = tables["count1"][s["count1table_select"]]
    code_map = _code_map(quad["hlen"], quad["hcod"])
    while bits.pos < end and line + 4 <= 576:
        symbol = _read_code(bits, code_map)
        if symbol is None:
            return None
        v = [(symbol >> (3 - k)) & 1 for k in range(4)]
        for k in range(4):
            if v[k] and bits.take1():
                v[k] = -1
        if bits.pos > end:
            break
        out[line:line + 4] = v
        line += 4
    return out


def decode_main_data(tables, h, side, main):
    """-> [granule][channel] dict(is, scalefac_l, scalefac_s, preflag, part2_bits) or None where the granule is undecodable"""
    out, start = [], 0
    first_granule = [[0] * 22, [0] * 22]  # granule 0's long scale factors stand for scfsi even if its Huffman data is damaged
    for gr in range(h["granules"]):
        row = []
        for ch in range(h["channels"]):
            s = side["gr"][gr][ch]
            end = start + s["part2_3_length"]
            bits = MainBits(main, start)
            g = None
            if end <= 8 * len(main):
                sl, ss, preflag = scale_factors(tables, h, side, gr, ch, bits, first_granule[ch])
                if gr == 0:
                    first_granule[ch] = sl
                part2 = bits.pos - start
                if bits.pos <= end:
                    values = huffman_granule(tables, h, s, bits, end)
                    if values is not None:
                        g = {"is": values, "scalefac_l": sl, "scalefac_s": ss, "preflag": preflag, "part2_bits": part2}
            row.append(g)
            start = end
        out.append(row)
    return out


class Decoder:
    """the whole decode in f64: frames -> interleaved PCM; mirrors what sk_mp3_decoder_* does with a frame it cannot decode
    (consumed without output).  hybrid: oracle/mp3_hybrid.py's Channel class."""

    def __init__(self, tables):
        from . import mp3_hybrid
        self.tables, self.hybrid = tables, mp3_hybrid
        self.kept, self.channels, self.state = b"", 0, []
        self.window = np.asarray(tables["window"], np.float32).astype(np.float64)

    def frame(self, data, off, h):
        """-> [samples_per_channel][channels] f64 or None"""
        frame = data[off:off + h["frame_bytes"]]
        side = parse_side_info(frame, h)
        head = 4 + (2 if h["has_crc"] else 0) + h["side_info_bytes"]
        back, own = side["main_data_begin"], frame[head:]
        main = (self.kept[len(self.kept) - back:] if back else b"") + own if back <= len(self.kept) else None
        self.kept = (self.kept + own)[-2048:]
        if main is None:
            return None
        joint = h["mode"] == 1
        grs = decode_main_data(self.tables, h, side, main)
        if any(g is None for row in grs for g in row):
            return None
        if self.channels != h["channels"]:
            self.channels, self.state = h["channels"], [self.hybrid.Channel() for _ in range(h["channels"])]
        long_offsets, short_offsets = self.tables["bands"][h["sample_rate"]]
        out = []
        for gr in range(h["granules"]):
            lsf = h["version"] != 1
            intensity = int(joint and bool(h["mode_ext"] & 1))
            if intensity and lsf and h["channels"] == 2 and side["gr"][gr][1]["scalefac_compress"] & 1:
                intensity |= 2  # intensity_scale
            g = {"channels": h["channels"], "ms_stereo": int(joint and bool(h["mode_ext"] & 2)), "intensity_stereo": intensity, "lsf": int(lsf),
                 "ch": []}
            for ch in range(h["channels"]):
                s, d = side["gr"][gr][ch], grs[gr][ch]
                g["ch"].append({"global_gain": s["global_gain"], "scalefac_scale": s["scalefac_scale"], "preflag": d["preflag"],
                                "block_type": s["block_type"], "mixed_block_flag": s["mixed_block_flag"],
                                "subblock_gain": s["subblock_gain"], "scalefac_l": d["scalefac_l"], "scalefac_s": d["scalefac_s"]})
            xr = requantize_granule(g, [grs[gr][ch]["is"] for ch in range(h["channels"])], long_offsets, short_offsets, self.tables["pretab"])
            # the product hands f32 lines from one GPU stage to the next
            xr = xr.astype(np.float32).astype(np.float64)
            pcm = [self.state[ch].granule(xr[ch], side["gr"][gr][ch]["block_type"], side["gr"][gr][ch]["mixed_block_flag"], self.window)
                   for ch in range(h["channels"])]
            out.append(np.stack(pcm, axis=1))
        return np.concatenate(out)
