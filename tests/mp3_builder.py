"""Test-side MP3 encoder: synthetic code books in the shape of sk_mp3_tables and Layer III streams written with them.

Nothing here is the standard's data.  The code books are random prefix codes with the STRUCTURE Table B.7 has (which
tables exist, their sizes, which carry linbits); the band tables are random partitions; the scale-factor length and
partition tables below are small syntax tables of 11172-3 2.4.2.7 / 13818-3 2.4.3.2 typed in as TEST DATA -- the product
takes them from its caller like everything else, and no parity claim rests on them.  What the tests check is that the
product's bitstream syntax (framing, reservoir, scale factors, regions, escapes, signs, count1) inverts this writer and
agrees with oracle/mp3_bitstream.py, and that the PCM behind it matches the f64 chain."""
import heapq

import numpy as np

from oracle import mp3_bitstream as ref
from oracle import mp3_hybrid

RATES = [44100, 48000, 32000, 22050, 24000, 16000, 11025, 12000, 8000]
# which big-value tables exist and how large they are (structure only); linbits of 16..23 and 24..31
XLEN = [0, 2, 3, 3, 0, 4, 4, 6, 6, 6, 8, 8, 8, 16, 0, 16] + [16] * 16
LINBITS = [0] * 16 + [1, 2, 3, 4, 6, 8, 10, 13, 4, 5, 6, 7, 8, 9, 11, 13]
# test data, see the module docstring
SLEN = [[0, 0], [0, 1], [0, 2], [0, 3], [3, 0], [1, 1], [1, 2], [1, 3], [2, 1], [2, 2], [2, 3], [3, 1], [3, 2], [3, 3], [4, 2], [4, 3]]
LSF_PARTITIONS = [[[6, 5, 5, 5], [9, 9, 9, 9], [6, 9, 9, 9]], [[6, 5, 7, 3], [9, 9, 12, 6], [6, 9, 12, 6]],
                  [[11, 10, 0, 0], [18, 18, 0, 0], [15, 18, 0, 0]], [[7, 7, 7, 0], [12, 12, 12, 0], [6, 15, 12, 0]],
                  [[6, 6, 6, 3], [12, 9, 9, 6], [6, 12, 9, 6]], [[8, 8, 5, 0], [15, 12, 9, 0], [6, 18, 9, 0]]]


def random_prefix_code(rng, n, max_len=19):
    """a Huffman tree over random weights with random 0/1 at every merge -> (hlen[n], hcod[n]); not canonical"""
    while True:
        weights = rng.random(n) ** 2 + 0.02
        heap = [(float(w), i, (i,)) for i, w in enumerate(weights)]
        heapq.heapify(heap)
        codes = {i: "" for i in range(n)}
        tie = n
        while len(heap) > 1:
            a, b = heapq.heappop(heap), heapq.heappop(heap)
            flip = int(rng.integers(0, 2))
            for i in a[2]:
                codes[i] = str(flip) + codes[i]
            for i in b[2]:
                codes[i] = str(1 - flip) + codes[i]
            heapq.heappush(heap, (a[0] + b[0], tie, a[2] + b[2]))
            tie += 1
        if max(len(c) for c in codes.values()) <= max_len:
            return [len(codes[i]) for i in range(n)], [int(codes[i], 2) for i in range(n)]


def partition(rng, parts, total, must_have):
    while True:
        inner = sorted(set(int(v) for v in rng.choice(np.arange(1, total), parts - 1, replace=False)) | {must_have})
        if len(inner) == parts:
            inner.pop(int(rng.integers(0, parts)))
        if len(inner) == parts - 1 and must_have in inner:
            return [0] + inner + [total]


def make_tables(seed=0, rates=(44100, 48000, 32000, 22050, 24000, 16000, 11025, 12000, 8000)):
    rng = np.random.default_rng(seed)
    big = []
    shared = {}
    for t in range(32):
        if XLEN[t] == 0:
            big.append(None)
            continue
        # like in the standard, 16..23 and 24..31 are one code set each with different linbits
        key = 16 if 16 <= t < 24 else (24 if t >= 24 else t)
        if key not in shared:
            shared[key] = random_prefix_code(rng, XLEN[t] * XLEN[t])
        hlen, hcod = shared[key]
        big.append({"xlen": XLEN[t], "linbits": LINBITS[t], "hlen": hlen, "hcod": hcod})
    hlen_a, hcod_a = random_prefix_code(rng, 16, 8)
    perm = rng.permutation(16)
    count1 = [{"hlen": hlen_a, "hcod": hcod_a}, {"hlen": [4] * 16, "hcod": [int(v) for v in perm]}]
    bands = {r: (partition(rng, 22, 576, 36), partition(rng, 13, 192, 12)) for r in rates}
    return {"big_values": big, "count1": count1, "slen": SLEN, "lsf_partitions": LSF_PARTITIONS, "bands": bands,
            "pretab": [int(v) for v in rng.integers(0, 4, 22)], "window": mp3_hybrid.synthetic_window(seed + 1)}


# ---- the writer ------------------------------------------------------------------------------------------------------------

class BitWriter:
    def __init__(self):
        self.bits = []

    def put(self, v, n):
        assert 0 <= v < (1 << n) or n == 0
        self.bits.extend((v >> (n - 1 - i)) & 1 for i in range(n))

    def __len__(self):
        return len(self.bits)

    def tobytes(self):
        b = self.bits + [0] * (-len(self.bits) % 8)
        return bytes(int("".join(map(str, b[i:i + 8])), 2) for i in range(0, len(b), 8))


def header_bytes(version, rate, bitrate_index, channels, mode, mode_ext, crc, padding=0):
    version_bits = {1: 3, 2: 2, 25: 0}[version]
    rate_index = ref.SAMPLE_RATES[version_bits].index(rate)
    b1 = 0xE0 | (version_bits << 3) | (1 << 1) | (0 if crc else 1)
    b2 = (bitrate_index << 4) | (rate_index << 2) | (padding << 1)
    b3 = (mode << 6) | (mode_ext << 4)
    return bytes([0xFF, b1, b2, b3])


def put_pair(w, table, x, y):
    ax, ay = min(abs(x), table["xlen"] - 1), min(abs(y), table["xlen"] - 1)
    if not table["linbits"]:
        assert abs(x) < table["xlen"] and abs(y) < table["xlen"]
    symbol = ax * table["xlen"] + ay
    w.put(table["hcod"][symbol], table["hlen"][symbol])
    for v, a in ((x, ax), (y, ay)):
        if table["linbits"] and a == table["xlen"] - 1:
            w.put(abs(v) - a, table["linbits"])
        if v:
            w.put(1 if v < 0 else 0, 1)


def random_granule(rng, tables, h, gr, first_granule, budget_bits, shape=None, allow_mixed=True, intensity_channel=False):
    """-> (side dict, scalefac_l, scalefac_s, is[576], BitWriter) within budget_bits (and 4095); shape = (window_switching,
    block_type, mixed_block_flag) to copy (the second channel of a joint-stereo pair)"""
    long_o, short_o = tables["bands"][h["sample_rate"]]
    mpeg1 = h["version"] == 1
    budget_bits = min(budget_bits, 4095)
    for attempt in range(60):
        w = BitWriter()
        ws = int(rng.random() < 0.4)
        bt = int(rng.integers(1, 4)) if ws else 0
        mixed = int(bt == 2 and allow_mixed and rng.random() < 0.3)
        if shape is not None:
            ws, bt, mixed = shape
        s = {"window_switching": ws, "block_type": bt, "mixed_block_flag": mixed, "global_gain": int(rng.integers(90, 160)),
             "subblock_gain": [int(v) for v in rng.integers(0, 4, 3)] if ws else [0, 0, 0], "scalefac_scale": int(rng.integers(0, 2)),
             "count1table_select": int(rng.integers(0, 2)), "preflag": int(mpeg1 and rng.random() < 0.3), "scfsi": [0, 0, 0, 0]}
        sl, ss = [0] * 22, [[0, 0, 0] for _ in range(13)]
        short = bt == 2
        # part 2
        if mpeg1:
            s["scalefac_compress"] = int(rng.integers(0, 16))
            slen1, slen2 = SLEN[s["scalefac_compress"]]
            if short:
                first = 0
                if mixed:
                    for band in range(8):
                        sl[band] = int(rng.integers(0, 1 << slen1))
                        w.put(sl[band], slen1)
                    first = 3
                for band in range(first, 12):
                    n = slen1 if band < 6 else slen2
                    for win in range(3):
                        ss[band][win] = int(rng.integers(0, 1 << n))
                        w.put(ss[band][win], n)
            else:
                for group, (a, b) in enumerate(((0, 6), (6, 11), (11, 16), (16, 21))):
                    reuse = gr == 1 and first_granule is not None and rng.random() < 0.4
                    s["scfsi"][group] = int(reuse)
                    n = slen1 if group < 2 else slen2
                    for band in range(a, b):
                        if reuse:
                            sl[band] = first_granule[band]
                        else:
                            sl[band] = int(rng.integers(0, 1 << n))
                            w.put(sl[band], n)
        elif intensity_channel:
            # 13818-3 2.4.3.2, the right channel of an intensity-stereo frame: scalefac_compress >> 1 selects three (four) field
            # widths and a row 3..5 of the partition table; its low bit is intensity_scale.  Fields at their largest value
            # ("not intensity coded") come up by themselves: widths are small
            kind = 3 + int(rng.integers(0, 3))
            if kind == 3:
                lens = [int(rng.integers(0, 5)), int(rng.integers(0, 6)), int(rng.integers(0, 6)), 0]
                sfc = lens[0] * 36 + lens[1] * 6 + lens[2]
            elif kind == 4:
                lens = [int(rng.integers(0, 4)), int(rng.integers(0, 4)), int(rng.integers(0, 4)), 0]
                sfc = 180 + ((lens[0] << 4) | (lens[1] << 2) | lens[2])
            else:
                lens = [int(rng.integers(0, 4)), int(rng.integers(0, 3)), 0, 0]
                sfc = 244 + lens[0] * 3 + lens[1]
            assert sfc < 256
            s["scalefac_compress"] = (sfc << 1) | int(rng.integers(0, 2))
            column = (2 if mixed else 1) if short else 0
            values = []
            for part, count in enumerate(LSF_PARTITIONS[kind][column]):
                for _ in range(count):
                    v = int(rng.integers(0, 1 << lens[part]))
                    w.put(v, lens[part])
                    values.append(v | 0x80 if lens[part] > 0 and v == (1 << lens[part]) - 1 else v)  # as the decoders mark it
            if column == 0:
                sl[:len(values)] = values
            else:
                n_long = 6 if column == 2 else 0  # a mixed granule: six long bands, then short band 3 on
                sl[:n_long] = values[:n_long]
                for k, v in enumerate(values[n_long:]):
                    band, win = divmod(k + (9 if column == 2 else 0), 3)
                    ss[band][win] = v
        else:
            kind = int(rng.integers(0, 3))
            if kind == 0:
                lens = [int(rng.integers(0, 5)), int(rng.integers(0, 5)), int(rng.integers(0, 4)), int(rng.integers(0, 4))]
                s["scalefac_compress"] = ((lens[0] * 5 + lens[1]) << 4) | (lens[2] << 2) | lens[3]
            elif kind == 1:
                lens = [int(rng.integers(0, 5)), int(rng.integers(0, 5)), int(rng.integers(0, 4)), 0]
                s["scalefac_compress"] = 400 + (((lens[0] * 5 + lens[1]) << 2) | lens[2])
            else:
                lens = [int(rng.integers(0, 4)), int(rng.integers(0, 3)), 0, 0]
                s["scalefac_compress"] = 500 + lens[0] * 3 + lens[1]
            assert s["scalefac_compress"] < 512
            column = (2 if mixed else 1) if short else 0
            values = []
            for part, count in enumerate(LSF_PARTITIONS[kind][column]):
                for _ in range(count):
                    values.append(int(rng.integers(0, 1 << lens[part])))
                    w.put(values[-1], lens[part])
            if column == 0:
                sl[:len(values)] = values
            else:
                n_long = 6 if column == 2 else 0
                sl[:n_long] = values[:n_long]
                for k, v in enumerate(values[n_long:]):
                    band, win = divmod(k + (9 if column == 2 else 0), 3)
                    ss[band][win] = v
        # part 3
        room = budget_bits - len(w)
        if room < 0:
            continue
        scale = 0.5 ** attempt
        big_values = int(rng.integers(0, 289) * min(1.0, room / 3000.0) * scale)
        if ws:
            s["region0_count"], s["region1_count"] = (8 if (short and not mixed) else 7), 36
        else:
            s["region0_count"], s["region1_count"] = int(rng.integers(0, 16)), int(rng.integers(0, 8))
        r1, r2 = ref.region_bounds(s, long_o, short_o)
        big_end = 2 * big_values
        bounds = [0, min(r1, big_end), min(r2, big_end), big_end]
        usable = [t for t in range(32) if XLEN[t]]
        s["table_select"] = [int(rng.choice(usable + [0])) for _ in range(3)]
        if ws:
            s["table_select"][2] = 0
        values = [0] * 576
        line = 0
        for region in range(3):
            t = s["table_select"][region]
            table = tables["big_values"][t]
            while line < bounds[region + 1]:
                x = y = 0
                if table:
                    top = table["xlen"] - 1 + ((1 << table["linbits"]) - 1 if table["linbits"] else 0)
                    top = min(top, 8206)
                    x, y = (int(min(top, rng.geometric(0.35) - 1)) * int(rng.choice([-1, 1])) for _ in range(2))
                    if table["linbits"] and rng.random() < 0.02:
                        x = int(rng.integers(-top, top + 1))
                    put_pair(w, table, x, y)
                values[line], values[line + 1] = x, y
                line += 2
        quads = int(rng.integers(0, (576 - line) // 4 + 1) * scale)
        quad = tables["count1"][s["count1table_select"]]
        for _ in range(quads):
            v = [int(rng.random() < 0.4) * int(rng.choice([-1, 1])) for _ in range(4)]
            if len(w) + 12 > budget_bits:
                break
            symbol = sum((1 if v[k] else 0) << (3 - k) for k in range(4))
            w.put(quad["hcod"][symbol], quad["hlen"][symbol])
            for k in range(4):
                if v[k]:
                    w.put(1 if v[k] < 0 else 0, 1)
            values[line:line + 4] = v
            line += 4
        if len(w) > budget_bits:
            continue
        s["big_values"] = big_values
        s["part2_3_length"] = len(w)
        return s, sl, ss, values, w
    raise AssertionError("no granule fits %d bits" % budget_bits)


def pack_side_info(h, side):
    w = BitWriter()
    mpeg1, ch = h["version"] == 1, h["channels"]
    w.put(side["main_data_begin"], 9 if mpeg1 else 8)
    w.put(0, (5 if ch == 1 else 3) if mpeg1 else (1 if ch == 1 else 2))
    if mpeg1:
        for c in range(ch):
            for v in side["scfsi"][c]:
                w.put(v, 1)
    for g in range(h["granules"]):
        for c in range(ch):
            s = side["gr"][g][c]
            w.put(s["part2_3_length"], 12), w.put(s["big_values"], 9), w.put(s["global_gain"], 8)
            w.put(s["scalefac_compress"], 4 if mpeg1 else 9), w.put(s["window_switching"], 1)
            if s["window_switching"]:
                w.put(s["block_type"], 2), w.put(s["mixed_block_flag"], 1)
                w.put(s["table_select"][0], 5), w.put(s["table_select"][1], 5)
                for k in range(3):
                    w.put(s["subblock_gain"][k], 3)
            else:
                for r in range(3):
                    w.put(s["table_select"][r], 5)
                w.put(s["region0_count"], 4), w.put(s["region1_count"], 3)
            if mpeg1:
                w.put(s["preflag"], 1)
            w.put(s["scalefac_scale"], 1), w.put(s["count1table_select"], 1)
    assert len(w) == 8 * h["side_info_bytes"]
    return w.tobytes()


def build_stream(tables, seed, version=1, rate=44100, channels=2, mode=None, n_frames=12, crc=False, bitrate_indices=(5, 9, 12),
                 joint_modes=(0, 2), free_format_bytes=0):
    """-> (bytes, [per frame: dict(header, side, granules=[gr][ch] dict(is, scalefac_l, scalefac_s, preflag))])
    mode: 0 stereo, 1 joint stereo (mode_ext from joint_modes: bit 1 = mid/side, bit 0 = intensity), 2 dual, 3 mono
    free_format_bytes: a free-format stream (bit-rate index 0) whose frames are that long, plus a padding slot in some"""
    rng = np.random.default_rng(seed)
    if mode is None:
        mode = 3 if channels == 1 else 1
    maxback = 511 if version == 1 else 255
    frames, blobs = [], []
    slot_start, prev_end = 0, 0
    positions = []
    for k in range(n_frames):
        mode_ext = int(rng.choice(joint_modes)) if mode == 1 else 0
        if free_format_bytes:
            hb = header_bytes(version, rate, 0, channels, mode, mode_ext, crc, padding=int(rng.integers(0, 2)))
        else:
            hb = header_bytes(version, rate, int(rng.choice(bitrate_indices)), channels, mode, mode_ext, crc)
        h = ref.parse_header(hb, free_format_bytes)
        slot = h["frame_bytes"] - 4 - (2 if crc else 0) - h["side_info_bytes"]
        assert slot > 0
        start = max(prev_end, slot_start - maxback)
        if k == 0:
            start = 0
        budget = 8 * (slot_start + slot - start)
        side = {"main_data_begin": slot_start - start, "scfsi": [[0] * 4, [0] * 4], "gr": []}
        grans, w_all = [], BitWriter()
        n_units = h["granules"] * channels
        first = [None, None]
        unit = 0
        for gr in range(h["granules"]):
            row_side, row = [], []
            shape = None
            for ch in range(channels):
                share = (budget - len(w_all)) // (n_units - unit)
                s, sl, ss, values, w = random_granule(rng, tables, h, gr, first[ch] if gr == 1 else None, int(share * rng.uniform(0.3, 1.0)),
                                                      shape if mode == 1 else None,
                                                      intensity_channel=version != 1 and mode == 1 and bool(mode_ext & 1) and ch == 1)
                shape = (s["window_switching"], s["block_type"], s["mixed_block_flag"])  # a joint pair is cut up the same way
                if gr == 1:
                    side["scfsi"][ch] = s["scfsi"]
                if gr == 0:
                    first[ch] = sl if s["block_type"] != 2 else None
                if gr == 1 and first[ch] is None:
                    assert not any(s["scfsi"])
                row_side.append(s)
                lsf_is = version != 1 and mode == 1 and bool(mode_ext & 1) and ch == 1
                row.append({"is": values, "scalefac_l": sl, "scalefac_s": ss,
                            "preflag": s["preflag"] if version == 1 else int(not lsf_is and s["scalefac_compress"] >= 500)})
                w_all.bits.extend(w.bits)
                unit += 1
            side["gr"].append(row_side)
            grans.append(row)
        blob = w_all.tobytes()
        extra = int(rng.integers(0, max(1, (budget // 8 - len(blob)) // 2 + 1)))  # stuffing: lets the reservoir grow
        blob += bytes(extra)
        assert len(blob) * 8 <= budget
        positions.append(start)
        blobs.append(blob)
        frames.append({"header": h, "header_bytes": hb, "side": side, "granules": grans, "slot": slot, "slot_start": slot_start})
        prev_end = start + len(blob)
        slot_start += slot
    main = bytearray(slot_start)
    for start, blob in zip(positions, blobs):
        main[start:start + len(blob)] = blob
    out = b""
    for f in frames:
        out += f["header_bytes"] + (b"\x5a\xa5" if crc else b"") + pack_side_info(f["header"], f["side"])
        out += bytes(main[f["slot_start"]:f["slot_start"] + f["slot"]])
    return out, frames


# ---- ctypes form of the tables ---------------------------------------------------------------------------------------------

def to_ctypes(tables):
    """-> (Mp3Tables, keepalive list) for sk_mp3_codebook_create"""
    import ctypes as C

    from soundkit_amd._lib import Mp3Tables
    t = Mp3Tables()
    keep = []
    for i, table in enumerate(tables["big_values"]):
        if not table:
            continue
        hlen = (C.c_uint8 * len(table["hlen"]))(*table["hlen"])
        hcod = (C.c_uint32 * len(table["hcod"]))(*table["hcod"])
        keep += [hlen, hcod]
        t.big_values[i].xlen, t.big_values[i].linbits = table["xlen"], table["linbits"]
        t.big_values[i].hlen = C.cast(hlen, C.POINTER(C.c_uint8))
        t.big_values[i].hcod = C.cast(hcod, C.POINTER(C.c_uint32))
    for k in range(2):
        for s in range(16):
            t.count1_hlen[k][s], t.count1_hcod[k][s] = tables["count1"][k]["hlen"][s], tables["count1"][k]["hcod"][s]
    for i in range(16):
        t.slen[i][0], t.slen[i][1] = tables["slen"][i]
    for r in range(6):
        for c in range(3):
            for p in range(4):
                t.lsf_partitions[r][c][p] = tables["lsf_partitions"][r][c][p]
    for row, rate in enumerate(RATES):
        if rate in tables["bands"]:
            t.rates_present[row] = 1
            for i in range(23):
                t.long_offsets[row][i] = tables["bands"][rate][0][i]
            for i in range(14):
                t.short_offsets[row][i] = tables["bands"][rate][1][i]
    for i in range(22):
        t.pretab[i] = tables["pretab"][i]
    for i in range(512):
        t.window[i] = float(tables["window"][i])
    return t, keep
