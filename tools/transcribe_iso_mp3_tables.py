#!/usr/bin/env python3
"""Writes the data tables of ISO/IEC 11172-3 / 13818-3 Layer III that the MP3 path runs on:

    soundkit_amd/csrc/mp3_iso_tables.h   (product: what sk_mp3_iso_tables() hands out)
    oracle/mp3_iso_tables.json           (checker: the same numbers for oracle/mp3_bitstream.py)

WHAT THE NUMBERS ARE.  Normative constants of the standard: Table B.7 (the 15 distinct big-value Huffman code sets with
the 32-entry table_select -> (code set, linbits) map, and count1 tables A / B), Table B.8 (scale-factor band widths for
the nine sampling rates), the scale-factor length table of 11172-3 2.4.2.7, the partition table of 13818-3 2.4.3.2, the
pre-emphasis table B.6 and the synthesis window D (Table B.3; every D[i] is an integer multiple of 2^-16).

WHERE THEY WERE READ.  The reference decodes MP3 through the crate nanomp3, whose source is not in the reference tree, and
the annexes are not in this image as text.  The image does hold the constants as DATA: the read-only data segment of a
third-party binary that has nothing to do with the reference (the Chromium inside the `kaleido` Python package embeds an
MPEG audio decoder, and its constant arrays lie there in plain little-endian form).  This script finds them by anchors
(the 44.1 kHz long band widths, the head of D, code set 5) and reads them out.  No code is taken: arrays of numbers only.

WHAT MAKES A WRONG ENTRY IMPOSSIBLE TO SHIP (checked here before anything is written; the tests re-check what is written):
  * every big-value code set and count1 table A is a COMPLETE prefix code: prefix-free and Kraft sum exactly 1, every
    code below 2^length, maximum length 19 -- a misplaced byte breaks one of them;
  * the values this author knows by heart agree: code sets 1, 2, 3, 5, count1 A / B, the slen table, the partition
    table, the 44.1 kHz widths, linbits, the first 54 window values, D[64] = 213 / 65536 and the peak D[256] = 75038 / 65536;
  * band widths of every rate sum to 576 (long) and 192 (short);
  * D has the standard's symmetry D[512 - i] = -D[i] (i not a multiple of 64), D[512 - i] = D[i] otherwise.
The tests add: exact part2_3_length consumption on all 168 frames of the reference's two MP3 fixtures, a filterbank
reconstruction test on D, and the decoded fixtures against the reference-held source PCM (tests/test_mp3_iso_tables.py,
tests/test_mp3_fixtures_gpu.py).

usage: tools/transcribe_iso_mp3_tables.py [path-to-binary]      (default: the kaleido executable of the image)
"""
import json
import mmap
import os
import struct
import sys
from fractions import Fraction

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DEFAULT = "/usr/local/lib/python3.10/dist-packages/kaleido/executable/bin/kaleido"

RATES = [44100, 48000, 32000, 22050, 24000, 16000, 11025, 12000, 8000]
CODE_SETS = [(1, 2), (2, 3), (3, 3), (5, 4), (6, 4), (7, 6), (8, 6), (9, 6), (10, 8), (11, 8), (12, 8), (13, 16), (15, 16), (16, 16), (24, 16)]
# table_select -> code set, linbits (11172-3 Table B.7's headers)
SELECT = [0, 1, 2, 3, 0, 5, 6, 7, 8, 9, 10, 11, 12, 13, 0, 15] + [16] * 8 + [24] * 8
LINBITS = [0] * 16 + [1, 2, 3, 4, 6, 8, 10, 13, 4, 5, 6, 7, 8, 9, 11, 13]

KNOWN = {  # from memory of the annex; (lengths, codes) indexed x * xlen + y
    1: ([1, 3, 2, 3], [1, 1, 1, 0]),
    2: ([1, 3, 6, 3, 3, 5, 5, 5, 6], [1, 2, 1, 3, 1, 1, 3, 2, 0]),
    3: ([2, 2, 6, 3, 2, 5, 5, 5, 6], [3, 2, 1, 1, 1, 1, 3, 2, 0]),
    5: ([1, 3, 6, 7, 3, 3, 6, 7, 6, 6, 7, 8, 7, 6, 7, 8], [1, 2, 6, 5, 3, 1, 4, 4, 7, 5, 7, 1, 6, 1, 1, 0]),
}
KNOWN_QUAD_LEN = [[1, 4, 4, 5, 4, 6, 5, 6, 4, 5, 5, 6, 5, 6, 6, 6], [4] * 16]
KNOWN_QUAD_COD = [[1, 5, 4, 5, 6, 5, 4, 4, 7, 3, 6, 0, 7, 2, 3, 1], list(range(15, -1, -1))]
KNOWN_SLEN = [[0, 0], [0, 1], [0, 2], [0, 3], [3, 0], [1, 1], [1, 2], [1, 3], [2, 1], [2, 2], [2, 3], [3, 1], [3, 2], [3, 3], [4, 2], [4, 3]]
KNOWN_PARTITIONS = [[[6, 5, 5, 5], [9, 9, 9, 9], [6, 9, 9, 9]], [[6, 5, 7, 3], [9, 9, 12, 6], [6, 9, 12, 6]],
                    [[11, 10, 0, 0], [18, 18, 0, 0], [15, 18, 0, 0]], [[7, 7, 7, 0], [12, 12, 12, 0], [6, 15, 12, 0]],
                    [[6, 6, 6, 3], [12, 9, 9, 6], [6, 12, 9, 6]], [[8, 8, 5, 0], [15, 12, 9, 0], [6, 18, 9, 0]]]
KNOWN_LONG_44 = [4, 4, 4, 4, 4, 4, 6, 6, 8, 8, 10, 12, 16, 20, 24, 28, 34, 42, 50, 54, 76, 158]
KNOWN_SHORT_44 = [4, 4, 4, 4, 6, 8, 10, 12, 14, 18, 22, 30, 56]
KNOWN_PRETAB = [0] * 11 + [1, 1, 1, 1, 2, 2, 3, 3, 3, 2, 0]
KNOWN_D_HEAD = [0, -1, -1, -1, -1, -1, -1, -2, -2, -2, -2, -3, -3, -4, -4, -5, -5, -6, -7, -7, -8, -9, -10, -11, -13, -14, -16, -17, -19,
                -21, -24, -26, -29, -31, -35, -38, -41, -45, -49, -53, -58, -63, -68, -73, -79, -85, -91, -97, -104, -111, -117, -125,
                -132, -139]


def complete_prefix_code(hlen, hcod, max_len=19):
    if any(n < 1 or n > max_len or c >> n for n, c in zip(hlen, hcod)):
        return False
    if sum(Fraction(1, 1 << n) for n in hlen) != 1:
        return False
    words = sorted(format(c, "0%db" % n) for n, c in zip(hlen, hcod))
    return all(not b.startswith(a) for a, b in zip(words, words[1:]))


def find_one(mm, pattern, what):
    at = mm.find(pattern)
    if at < 0 or mm.find(pattern, at + 1) >= 0:
        raise SystemExit("anchor %s: not found exactly once" % what)
    return at


def read_code_set(mm, cursor, n):
    """the lengths (n*n bytes) and the codes (n*n u16) of one code set lie next to each other after `cursor`, each at some
    small alignment; the first placement that is a complete prefix code is the only one that can be"""
    cells = n * n
    for a in range(cursor, cursor + 64):
        hlen = list(mm[a:a + cells])
        if min(hlen) < 1 or max(hlen) > 19 or sum(Fraction(1, 1 << v) for v in hlen) != 1:
            continue
        for b in range(a + cells, a + cells + 32):
            hcod = list(struct.unpack("<%dH" % cells, mm[b:b + 2 * cells]))
            if complete_prefix_code(hlen, hcod):
                return hlen, hcod, b + 2 * cells
    raise SystemExit("no complete prefix code of %d x %d after 0x%x" % (n, n, cursor))


def main():
    path = sys.argv[1] if len(sys.argv) > 1 else DEFAULT
    with open(path, "rb") as fh:
        mm = mmap.mmap(fh.fileno(), 0, access=mmap.ACCESS_READ)

    # ---- Table B.8 -------------------------------------------------------------------------------------------------------
    at_long = find_one(mm, bytes(KNOWN_LONG_44), "44.1 kHz long band widths")
    long_w = [list(mm[at_long + 22 * r:at_long + 22 * r + 22]) for r in range(9)]
    at_short = find_one(mm, bytes(KNOWN_SHORT_44 + [4, 4, 4, 4, 6, 6]), "44.1 kHz short band widths")
    short_w = [list(mm[at_short + 13 * r:at_short + 13 * r + 13]) for r in range(9)]
    assert all(sum(w) == 576 for w in long_w) and all(sum(w) == 192 for w in short_w)
    assert all(min(w) >= 2 for w in long_w + short_w)
    long_off = [[sum(w[:i]) for i in range(23)] for w in long_w]
    short_off = [[sum(w[:i]) for i in range(14)] for w in short_w]

    # ---- Table B.7 -------------------------------------------------------------------------------------------------------
    sets = {}
    cursor = at_long + 9 * 22  # the code sets follow the long widths in this binary; each placement is validated anyway
    for number, n in CODE_SETS:
        hlen, hcod, cursor = read_code_set(mm, cursor, n)
        sets[number] = (n, hlen, hcod)
        if number in KNOWN:
            assert (hlen, hcod) == KNOWN[number], "code set %d differs from the annex as remembered" % number
    # sanity on the symbol assignment that a prefix check cannot see: (0, 0) has the shortest code of sets 1..12 except 3,
    # and lengths grow with x + y on the whole
    for number, (n, hlen, _) in sets.items():
        assert hlen[0] == min(hlen) or number in (3, 6, 8, 9, 11, 12, 16, 24), number
        assert hlen[n * n - 1] >= hlen[0]
    at_ql = find_one(mm, bytes(KNOWN_QUAD_LEN[0] + KNOWN_QUAD_LEN[1]), "count1 lengths")
    at_qc = find_one(mm, bytes(KNOWN_QUAD_COD[0] + KNOWN_QUAD_COD[1]), "count1 codes")
    quad_len = [list(mm[at_ql + 16 * k:at_ql + 16 * k + 16]) for k in range(2)]
    quad_cod = [list(mm[at_qc + 16 * k:at_qc + 16 * k + 16]) for k in range(2)]
    assert all(complete_prefix_code(quad_len[k], quad_cod[k]) for k in range(2))
    # the (code set, linbits) map as the binary holds it: 32 pairs
    pairs = bytes(v for t in range(32) for v in ([0, 1, 2, 3, 0, 4, 5, 6, 7, 8, 9, 10, 11, 12, 0, 13] + [14] * 8 + [15] * 8)[t:t + 1] + [LINBITS[t]])
    find_one(mm, pairs, "table_select -> (code set, linbits)")

    # ---- the small syntax tables ---------------------------------------------------------------------------------------------
    flat = bytes(v for row in KNOWN_PARTITIONS for col in row for v in col)
    find_one(mm, flat, "13818-3 partition table")
    if mm.find(bytes(v for pair in zip(*KNOWN_SLEN) for v in pair)) < 0 and mm.find(bytes([r[0] for r in KNOWN_SLEN] + [r[1] for r in KNOWN_SLEN])) < 0:
        raise SystemExit("slen table not found")
    find_one(mm, bytes([0] * 22 + KNOWN_PRETAB), "pre-emphasis table")

    # ---- Table B.3 -------------------------------------------------------------------------------------------------------
    at_d = find_one(mm, struct.pack("<54i", *KNOWN_D_HEAD), "head of the synthesis window")
    half = list(struct.unpack("<257i", mm[at_d:at_d + 4 * 257]))
    assert half[64] == 213 and half[256] == 75038 and max(abs(v) for v in half) == 75038
    d = half + [0] * 255
    for i in range(1, 256):
        d[512 - i] = -half[i] if i & 63 else half[i]
    for i in range(1, 512):  # the prototype behind D is smooth: no entry jumps away from its neighbours after the sign flips
        proto = [d[j] * (-1 if (j >> 6) & 1 else 1) for j in (i - 1, i)]
        assert abs(proto[1] - proto[0]) <= 1900, i

    write_header(sets, quad_len, quad_cod, long_off, short_off, d)
    write_json(sets, quad_len, quad_cod, long_off, short_off, d)
    print("wrote the tables: %d code sets, window peak %d/65536" % (len(sets), max(d)))


def c_array(values, per_line=24):
    lines = [", ".join(str(v) for v in values[i:i + per_line]) for i in range(0, len(values), per_line)]
    return "{\n    " + ",\n    ".join(lines) + "}"


def write_header(sets, quad_len, quad_cod, long_off, short_off, d):
    out = ["// mp3_iso_tables.h -- data tables of ISO/IEC 11172-3 (Tables B.3, B.6, B.7, B.8, the slen table of 2.4.2.7) and",
           "// ISO/IEC 13818-3 (the partition table of 2.4.3.2).  GENERATED by tools/transcribe_iso_mp3_tables.py, which says where",
           "// the numbers were read and which checks they passed before this file was written; tests/test_mp3_iso_tables.py",
           "// repeats the checks on what is compiled in.  Numbers only -- normative constants of the standard, no code.",
           "// Huffman code sets: length and right-aligned bits per (x, y), index x * xlen + y, as Table B.7 prints them.",
           "#pragma once", "#include <cstdint>", "", "namespace sk_mp3_iso {", ""]
    for number, (n, hlen, hcod) in sets.items():
        out.append("static const uint8_t hlen_%d[%d] = %s;" % (number, n * n, c_array(hlen, 32)))
        out.append("static const uint32_t hcod_%d[%d] = %s;" % (number, n * n, c_array(hcod, 24)))
    out.append("")
    out.append("struct CodeSet { uint8_t xlen; const uint8_t *hlen; const uint32_t *hcod; };")
    out.append("// table_select 0..31 -> code set (0, 4, 14: no codes) and linbits")
    rows = []
    for t in range(32):
        s = SELECT[t]
        rows.append("{%d, %s, %s}" % ((sets[s][0], "hlen_%d" % s, "hcod_%d" % s) if s else (0, "nullptr", "nullptr")))
    out.append("static const CodeSet select[32] = {\n    " + ",\n    ".join(", ".join(rows[i:i + 4]) for i in range(0, 32, 4)) + "};")
    out.append("static const uint8_t linbits[32] = %s;" % c_array(LINBITS, 32))
    out.append("static const uint8_t count1_hlen[2][16] = {%s, %s};" % (c_array(quad_len[0]), c_array(quad_len[1])))
    out.append("static const uint8_t count1_hcod[2][16] = {%s, %s};" % (c_array(quad_cod[0]), c_array(quad_cod[1])))
    out.append("static const uint8_t slen[16][2] = {%s};" % ", ".join("{%d, %d}" % tuple(r) for r in KNOWN_SLEN))
    out.append("static const uint8_t lsf_partitions[6][3][4] = {%s};" % ", ".join(
        "{" + ", ".join("{%d, %d, %d, %d}" % tuple(c) for c in row) + "}" for row in KNOWN_PARTITIONS))
    out.append("// rows: 44100 48000 32000 22050 24000 16000 11025 12000 8000")
    out.append("static const uint16_t long_offsets[9][23] = {%s};" % ",\n    ".join(c_array(r, 23) for r in long_off))
    out.append("static const uint16_t short_offsets[9][14] = {%s};" % ",\n    ".join(c_array(r, 14) for r in short_off))
    out.append("static const uint8_t pretab[22] = %s;" % c_array(KNOWN_PRETAB, 22))
    out.append("// D[i] * 65536 (every entry of Table B.3 is a multiple of 2^-16)")
    out.append("static const int32_t window_q16[512] = %s;" % c_array(d, 16))
    out += ["", "}  // namespace sk_mp3_iso", ""]
    with open(os.path.join(ROOT, "soundkit_amd", "csrc", "mp3_iso_tables.h"), "w") as fh:
        fh.write("\n".join(out))


def write_json(sets, quad_len, quad_cod, long_off, short_off, d):
    big = []
    for t in range(32):
        s = SELECT[t]
        big.append({"xlen": sets[s][0], "linbits": LINBITS[t], "hlen": sets[s][1], "hcod": sets[s][2]} if s else None)
    doc = {"_comment": "GENERATED by tools/transcribe_iso_mp3_tables.py -- ISO/IEC 11172-3 / 13818-3 Layer III tables for the oracle",
           "big_values": big, "count1": [{"hlen": quad_len[k], "hcod": quad_cod[k]} for k in range(2)], "slen": KNOWN_SLEN,
           "lsf_partitions": KNOWN_PARTITIONS, "bands": {str(r): [long_off[i], short_off[i]] for i, r in enumerate(RATES)},
           "pretab": KNOWN_PRETAB, "window_q16": d}
    with open(os.path.join(ROOT, "oracle", "mp3_iso_tables.json"), "w") as fh:
        json.dump(doc, fh, separators=(",", ":"))
        fh.write("\n")


if __name__ == "__main__":
    main()
