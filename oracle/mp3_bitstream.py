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


def parse_header(b):
    """4 bytes -> dict, or None if they are no Layer III frame header (free format counts as none)"""
    if len(b) < 4 or b[0] != 0xFF or (b[1] & 0xE0) != 0xE0:
        return None
    version_bits, layer_bits = (b[1] >> 3) & 3, (b[1] >> 1) & 3
    if version_bits == 1 or layer_bits != 1:
        return None
    bitrate_index, rate_index = b[2] >> 4, (b[2] >> 2) & 3
    if bitrate_index in (0, 15) or rate_index == 3:
        return None
    mpeg1 = version_bits == 3
    kbps = (BITRATE_V1 if mpeg1 else BITRATE_V2)[bitrate_index]
    rate = SAMPLE_RATES[version_bits][rate_index]
    padding = (b[2] >> 1) & 1
    mode = b[3] >> 6
    channels = 1 if mode == 3 else 2
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
        "frame_bytes": (144 if mpeg1 else 72) * kbps * 1000 // rate + padding,
        "side_info_bytes": (17 if channels == 1 else 32) if mpeg1 else (9 if channels == 1 else 17),
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


def scan(data):
    """every frame of a byte string: [(offset, header)]"""
    pos = 0
    if len(data) >= 10 and data[:3] == b"ID3":
        pos = 10 + ((data[6] & 0x7F) << 21 | (data[7] & 0x7F) << 14 | (data[8] & 0x7F) << 7 | (data[9] & 0x7F))
        if data[5] & 0x10:
            pos += 10
    frames = []
    while pos + 4 <= len(data):
        h = parse_header(data[pos:pos + 4])
        if h is None:
            pos += 1
            continue
        nxt = pos + h["frame_bytes"]
        if nxt > len(data):
            break
        if nxt + 4 <= len(data):
            follow = parse_header(data[nxt:nxt + 4])
            if follow is None or follow["version"] != h["version"] or follow["sample_rate"] != h["sample_rate"]:
                pos += 1
                continue
        frames.append((pos, h))
        pos = nxt
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
                exponent = (ch["global_gain"] - 210) / 4.0 - mult * (ch["scalefac_l"][band] + (int(pretab[band]) if ch["preflag"] else 0))
            else:
                exponent = (ch["global_gain"] - 210 - 8 * ch["subblock_gain"][w]) / 4.0 - mult * ch["scalefac_s"][band][w]
            vals[c][i] = math.copysign(abs(q) ** (4.0 / 3.0) * 2.0 ** exponent, q) if q else 0.0
    if channels == 2 and (ms_stereo or intensity_stereo):
        top = {}
        for i in range(576):
            band, w, _ = where[1][i]
            if quant[1][i] != 0:
                top[w] = max(top.get(w, -1), band)
        right = g["ch"][1]
        for i in range(576):
            band, w, _ = where[0][i]
            done = False
            if intensity_stereo and band > top.get(w, -1):
                pos = right["scalefac_l"][min(band, 20)] if w < 0 else right["scalefac_s"][min(band, 11)][w]
                if pos < 7:
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
