"""A tiny AAC-LC access-unit *encoder* for the parity tests: random, syntactically valid raw_data_blocks that exercise
every tool of the front-end -- all four window sequences and groupings, every section codebook (0, 1..11, noise 13,
intensity 14 / 15), escapes, pulse data, PNS, intensity + mid/side masks, multi-filter TNS, fill elements -- far more
densely than the reference's fixture streams do (none of which carries pulse data, for one).

It only *writes* the syntax of ISO/IEC 14496-3 4.4.2 (the element order the reference reads in decoder.rs:104-218,
channel.rs:36-75); it computes no audio.  What the units decode to is decided by oracle/aac_frontend.py, and the tests
ask the product's front-ends (host C++ and gfx950) for bit-identical spectra.  Test infrastructure, like the oracle.
"""
import numpy as np

from oracle import aac_frontend as OF

ONLY_LONG, LONG_START, EIGHT_SHORT, LONG_STOP = 0, 1, 2, 3


class Writer:
    def __init__(self):
        self.bits = []

    def put(self, value, width):
        assert 0 <= value < (1 << width), (value, width)
        self.bits += [(value >> b) & 1 for b in range(width - 1, -1, -1)]

    def code(self, book, index):
        lens, codes = OF._BOOKS[book]
        assert lens[index] > 0
        self.put(codes[index], lens[index])

    def bytes(self):
        bits = self.bits + [0] * ((-len(self.bits)) % 8)
        out = bytearray(len(bits) // 8)
        for i, b in enumerate(bits):
            if b:
                out[i // 8] |= 0x80 >> (i % 8)
        return bytes(out)


def _ics(rng, w, n_long, n_short):
    seq = int(rng.choice([ONLY_LONG, LONG_START, EIGHT_SHORT, LONG_STOP], p=[0.55, 0.1, 0.25, 0.1]))
    shape = int(rng.integers(2))
    w.put(0, 1)
    w.put(seq, 2)
    w.put(shape, 1)
    if seq == EIGHT_SHORT:
        max_sfb = int(rng.integers(0, min(n_short, 15) + 1))
        grouping = int(rng.integers(128))
        w.put(max_sfb, 4)
        w.put(grouping, 7)
        lens = [1]
        for bit in range(7):
            if (grouping >> (6 - bit)) & 1:
                lens[-1] += 1
            else:
                lens.append(1)
    else:
        max_sfb = int(rng.integers(0, min(n_long, 63) + 1))
        w.put(max_sfb, 6)
        w.put(0, 1)  # predictor_data_present
        lens = [1]
    return {"seq": seq, "shape": shape, "max_sfb": max_sfb, "groups": lens}


def _sections(rng, w, ics, allow_intensity, dense):
    short = ics["seq"] == EIGHT_SHORT
    width = 3 if short else 5
    esc = (1 << width) - 1
    pool = [0] * 3 + list(range(1, 12)) * (3 if dense else 1) + [13] * 2 + ([14, 15] * 2 if allow_intensity else [])
    books = []
    for _ in ics["groups"]:
        row = []
        while len(row) < ics["max_sfb"]:
            book = int(rng.choice(pool))
            length = int(min(ics["max_sfb"] - len(row), 1 + rng.geometric(0.35)))
            w.put(book, 4)
            rest = length
            while rest >= esc:
                w.put(esc, width)
                rest -= esc
            w.put(rest, width)
            row += [book] * length
        books.append(row)
    return books


def _scalefactors(rng, w, gain, books):
    spectral, noise, intensity, first_noise = gain, gain - 90, 0, True

    def delta(current, lo, hi):
        d = int(np.clip(rng.integers(-6, 7) if rng.random() < 0.8 else rng.integers(-40, 41), lo - current, hi - current))
        d = int(np.clip(d, -60, 60))
        w.code("sf", d + 60)
        return current + d
    for row in books:
        for book in row:
            if book == 0:
                continue
            if book == 13:
                if first_noise:
                    target = int(np.clip(noise + rng.integers(-20, 21), noise - 256, noise + 255))
                    w.put(target - noise + 256, 9)
                    noise, first_noise = target, False
                else:
                    noise = delta(noise, -100, 200)
            elif book in (14, 15):
                intensity = delta(intensity, -60, 60)
            else:
                spectral = delta(spectral, 40, 200)


def _pulse(rng, w, ics, books, off):
    """pulse data whose targets land in spectral bands (else no pulse); returns True when written"""
    if ics["seq"] == EIGHT_SHORT or ics["max_sfb"] == 0 or rng.random() > 0.35:
        w.put(0, 1)
        return
    for _ in range(8):
        count = int(rng.integers(1, 5))
        start = int(rng.integers(ics["max_sfb"]))
        pulses, index, ok = [], off[start], True
        for _ in range(count):
            offset, amp = int(rng.integers(32)), int(rng.integers(16))
            index += offset
            band = next((b for b in range(ics["max_sfb"]) if off[b] <= index < off[b + 1]), None)
            if band is None or not 1 <= books[0][band] <= 11:
                ok = False
                break
            pulses.append((offset, amp))
        if ok:
            w.put(1, 1)
            w.put(count - 1, 2)
            w.put(start, 6)
            for offset, amp in pulses:
                w.put(offset, 5)
                w.put(amp, 4)
            return
    w.put(0, 1)


def _tns(rng, w, ics):
    if rng.random() > 0.4:
        w.put(0, 1)
        return
    w.put(1, 1)
    short = ics["seq"] == EIGHT_SHORT
    n_bits, len_bits, order_bits, max_order = (1, 4, 3, 7) if short else (2, 6, 5, 12)
    for _ in range(8 if short else 1):
        n = int(rng.integers(0, (1 << n_bits)))
        w.put(n, n_bits)
        if not n:
            continue
        res = int(rng.integers(2))
        w.put(res, 1)
        for _ in range(n):
            w.put(int(rng.integers(1 << len_bits)), len_bits)
            order = int(rng.integers(0, max_order + 1))
            w.put(order, order_bits)
            if order:
                w.put(int(rng.integers(2)), 1)
                compress = int(rng.integers(2))
                w.put(compress, 1)
                bits = 3 + res - compress
                for _ in range(order):
                    w.put(int(rng.integers(1 << bits)), bits)


# extra bits of an escape sequence beyond the first four (N ones): the reference takes up to 12 (magnitudes to 131071,
# spectral.rs:214-228).  The default stays below i16; tests of the quantised hand-over's wide values widen it.
ESCAPE_SIZES = [0, 0, 0, 1, 2, 4, 8, 9]


def _band(rng, w, book, count, loud):
    if book <= 4:
        for _ in range(count // 4):
            if book <= 2:
                v = [int(x) for x in rng.integers(-1, 2, 4)] if rng.random() < loud else [0, 0, 0, 0]
                w.code(book, (v[0] + 1) * 27 + (v[1] + 1) * 9 + (v[2] + 1) * 3 + v[3] + 1)
            else:
                v = [int(x) for x in rng.integers(0, 3, 4)] if rng.random() < loud else [0, 0, 0, 0]
                w.code(book, v[0] * 27 + v[1] * 9 + v[2] * 3 + v[3])
                for x in v:
                    if x:
                        w.put(int(rng.integers(2)), 1)
        return
    dim, lo, hi = {5: (9, -4, 5), 6: (9, -4, 5), 7: (8, 0, 8), 8: (8, 0, 8), 9: (13, 0, 13), 10: (13, 0, 13), 11: (17, 0, 17)}[book]
    for _ in range(count // 2):
        v = [int(x) for x in rng.integers(lo, hi, 2)] if rng.random() < loud else [0, 0]
        if book <= 6:
            w.code(book, (v[0] + 4) * dim + v[1] + 4)
            continue
        w.code(book, v[0] * dim + v[1])
        for x in v:
            if x:
                w.put(int(rng.integers(2)), 1)
        if book == 11:
            for x in v:
                if x == 16:  # escape: N ones, a zero, N + 4 bits  ->  2^(N+4) + bits
                    n = int(rng.choice(ESCAPE_SIZES))
                    for _ in range(n):
                        w.put(1, 1)
                    w.put(0, 1)
                    w.put(int(rng.integers(1 << (n + 4))), n + 4)


def _spectral(rng, w, ics, books, off):
    loud = float(rng.choice([0.15, 0.5, 0.9]))
    short = ics["seq"] == EIGHT_SHORT
    for g, glen in enumerate(ics["groups"]):
        for sfb in range(ics["max_sfb"]):
            book = books[g][sfb]
            if 1 <= book <= 11:
                for _ in range(glen if short else 1):
                    _band(rng, w, book, off[sfb + 1] - off[sfb], loud)


def _channel(rng, w, ics_common, allow_intensity, n_long, n_short, sf_index, dense):
    gain = int(rng.integers(90, 180))
    w.put(gain, 8)
    ics = ics_common if ics_common is not None else _ics(rng, w, n_long, n_short)
    off = OF.short_offsets(sf_index) if ics["seq"] == EIGHT_SHORT else OF.long_offsets(sf_index)
    books = _sections(rng, w, ics, allow_intensity, dense)
    _scalefactors(rng, w, gain, books)
    _pulse(rng, w, ics, books, off)
    _tns(rng, w, ics)
    w.put(0, 1)  # gain_control_data_present
    _spectral(rng, w, ics, books, off)


def random_access_unit(rng, sf_index, channels, dense=True):
    """one raw_data_block: optional fill element, SCE or CPE, optional END, zero padding to the byte"""
    n_long, n_short = len(OF.long_offsets(sf_index)) - 1, len(OF.short_offsets(sf_index)) - 1
    w = Writer()
    if rng.random() < 0.2:     # FIL before the channel element (decoder.rs:393-419)
        n = int(rng.integers(0, 5))
        w.put(6, 3)
        w.put(n, 4)
        for i in range(n):
            w.put(0 if i == 0 else int(rng.integers(256)), 8)   # extension type 0 in the first nibble: not SBR
    if channels == 1:
        w.put(0, 3)
        w.put(int(rng.integers(16)), 4)
        _channel(rng, w, None, False, n_long, n_short, sf_index, dense)
    else:
        w.put(1, 3)
        w.put(int(rng.integers(16)), 4)
        common = rng.random() < 0.7
        w.put(int(common), 1)
        ics = None
        if common:
            ics = _ics(rng, w, n_long, n_short)
            mode = int(rng.choice([0, 1, 2]))
            w.put(mode, 2)
            if mode == 1:
                for _ in ics["groups"]:
                    for _ in range(ics["max_sfb"]):
                        w.put(int(rng.integers(2)), 1)
        _channel(rng, w, ics, False, n_long, n_short, sf_index, dense)
        _channel(rng, w, ics, common, n_long, n_short, sf_index, dense)
    if rng.random() < 0.7:
        w.put(7, 3)
    return w.bytes()


def asc_for(sf_index, channels):
    return bytes([(2 << 3) | (sf_index >> 1), ((sf_index & 1) << 7) | (channels << 3)])


def adts_frame(au, sf_index, channels):
    """the access unit behind a 7-byte ADTS header (protection absent), as parse_adts_access_unit expects it
    (soundkit-decoder lib.rs:1007-1027)"""
    n = len(au) + 7
    assert n < 8192
    return bytes([0xFF, 0xF1, (1 << 6) | (sf_index << 2) | (channels >> 2), ((channels & 3) << 6) | (n >> 11),
                  (n >> 3) & 0xFF, ((n & 7) << 5) | 0x1F, 0xFC]) + au


def extreme_scalefactor_units():
    """Scale factors and intensity positions outside the reference's tables (dsp.rs:407-413 falls back to powf beyond
    -256..511; scalefactor.rs:208-210 always computes): [(sf_index, channels, access unit)] with small quantised
    values so that nothing overflows.  44.1 kHz: the first bands are 4 wide, one codebook-1 quad each."""
    out = []
    for gain, step in ((255, 60), (0, -60)):           # spectral scale factor walks to 255 + 300 / 0 - 300
        w = Writer()
        w.put(0, 3), w.put(0, 4), w.put(gain, 8)
        w.put(0, 1), w.put(0, 2), w.put(0, 1), w.put(8, 6), w.put(0, 1)      # OnlyLong, sine, max_sfb 8, no prediction
        w.put(1, 4), w.put(8, 5)                                             # one section: codebook 1, 8 bands
        for k in range(8):
            w.code("sf", 60 + (step if k < 5 else 0))
        w.put(0, 1), w.put(0, 1), w.put(0, 1)
        for k in range(8):
            v = [(k + j) % 3 - 1 for j in range(4)]
            w.code(1, (v[0] + 1) * 27 + (v[1] + 1) * 9 + (v[2] + 1) * 3 + v[3] + 1)
        w.put(7, 3)
        out.append((4, 1, w.bytes()))
    for step in (60, -60):                             # intensity position walks to +-300 on the right channel
        w = Writer()
        w.put(1, 3), w.put(0, 4), w.put(1, 1)
        w.put(0, 1), w.put(0, 2), w.put(1, 1), w.put(8, 6), w.put(0, 1)
        w.put(0, 2)                                                          # no mid/side
        w.put(100, 8), w.put(1, 4), w.put(8, 5)
        for _ in range(8):
            w.code("sf", 60)
        w.put(0, 1), w.put(0, 1), w.put(0, 1)
        for k in range(8):
            v = [(k + j) % 3 - 1 for j in range(4)]
            w.code(1, (v[0] + 1) * 27 + (v[1] + 1) * 9 + (v[2] + 1) * 3 + v[3] + 1)
        w.put(100, 8), w.put(15 if step > 0 else 14, 4), w.put(8, 5)
        for k in range(8):
            w.code("sf", 60 + (step if k < 5 else 0))
        w.put(0, 1), w.put(0, 1), w.put(0, 1)
        w.put(7, 3)
        out.append((4, 2, w.bytes()))
    return out
