#!/usr/bin/env python3
"""One-off transcription of the ISO/IEC 14496-3 (13818-7) AAC constant tables into
soundkit_amd/csrc/aac_tables.h.

These are the standard's normative tables (Huffman codebooks 1-11 and the scalefactor
codebook: Tables 4.A.1-4.A.12; scalefactor-band offsets: Tables 4.129-4.147; TNS_MAX_BANDS),
identical in every conforming decoder.  They cannot be derived (the codes are not canonical),
so they are copied as DATA from where the reference carries them
(soundkit-aac-lc/src/spectral.rs:1027-1845, scalefactor.rs:222-250, sfb.rs:73-152,
tns.rs:284-285) and re-laid-out flat.  No code is taken.  tests/test_aac_frontend.py checks the
result independently (Kraft equality and prefix-freeness of every codebook).
Needs the reference tree; the build never runs this script.
"""
import re
import sys

REF = sys.argv[1] if len(sys.argv) > 1 else "/root/reference/soundkit-aac-lc/src"


def array(text, name):
    m = re.search(r"const %s:[^=]*=\s*(?:&)?\[" % name, text)
    assert m, name
    i, depth, j = m.end() - 1, 0, m.end() - 1
    while True:
        c = text[j]
        if c == "[":
            depth += 1
        elif c == "]":
            depth -= 1
            if depth == 0:
                break
        j += 1
    return [int(t, 0) for t in re.findall(r"0x[0-9a-fA-F]+|\d+", text[i:j])]


def emit(f, ctype, name, values, per_line=16):
    f.write("static const %s %s[%d] = {\n" % (ctype, name, len(values)))
    for k in range(0, len(values), per_line):
        f.write("    " + ", ".join(str(v) for v in values[k:k + per_line]) + ",\n")
    f.write("};\n\n")


spectral = open(REF + "/spectral.rs").read()
scalefactor = open(REF + "/scalefactor.rs").read()
sfb = open(REF + "/sfb.rs").read()
tns = open(REF + "/tns.rs").read()

books = {}
for hi, lo, lens in ((1, 2, "1_2"), (3, 4, "3_4"), (5, 6, "5_6"), (7, 8, "7_8"), (9, 10, "9_10")):
    packed = array(spectral, "STANDARD_CODEBOOK_%s_LENGTHS" % lens)
    books[hi] = ([p >> 16 for p in packed], array(spectral, "STANDARD_CODEBOOK_%d_CODES" % hi))
    books[lo] = ([p & 0xFFFF for p in packed], array(spectral, "STANDARD_CODEBOOK_%d_CODES" % lo))
books[11] = (array(spectral, "STANDARD_CODEBOOK_11_LENGTHS"), array(spectral, "STANDARD_CODEBOOK_11_CODES"))

with open("soundkit_amd/csrc/aac_tables.h", "w") as f:
    f.write("// aac_tables.h -- ISO/IEC 14496-3 AAC constant tables (normative data, see tools/transcribe_iso_tables.py).\n")
    f.write("// Huffman codebooks: entry i of codebook b is the codeword of index i in the standard's ordering\n")
    f.write("// (quads: 27w+9x+3y+z, pairs: dim*y+z; signed books offset by their largest absolute value).\n")
    f.write("#pragma once\n#include <stdint.h>\n\nnamespace sk_aac_tables {\n\n")
    emit(f, "uint8_t", "kSfLen", array(scalefactor, "STANDARD_SCALE_FACTOR_CODE_LENGTHS"))
    emit(f, "uint32_t", "kSfCode", array(scalefactor, "STANDARD_SCALE_FACTOR_CODES"), 8)
    for b in range(1, 12):
        lens, codes = books[b]
        assert len(lens) == len(codes)
        emit(f, "uint8_t", "kCb%dLen" % b, lens)
        emit(f, "uint16_t", "kCb%dCode" % b, codes)
    for name in ("1024_96", "1024_64", "1024_48", "1024_32", "1024_24", "1024_16", "1024_8", "128_96", "128_48", "128_24",
                 "128_16", "128_8"):
        emit(f, "uint16_t", "kSwb" + name, array(sfb, "SWB_OFFSET_" + name))
    emit(f, "uint8_t", "kTnsMaxBands1024", array(tns, "TNS_MAX_BANDS_1024"))
    emit(f, "uint8_t", "kTnsMaxBands128", array(tns, "TNS_MAX_BANDS_128"))
    f.write("}  // namespace sk_aac_tables\n")
print("wrote soundkit_amd/csrc/aac_tables.h")
