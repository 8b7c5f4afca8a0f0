"""The gates behind the Layer III data tables (csrc/mp3_iso_tables.h, oracle/mp3_iso_tables.json), CPU side.

The tables are normative constants of ISO/IEC 11172-3 / 13818-3 entered as data (tools/transcribe_iso_mp3_tables.py says
from where).  A wrong entry must not be able to ship, so:
  (i)   every Huffman code set is a COMPLETE prefix code (prefix-free, Kraft sum exactly 1);
  (ii)  on all 82 + 86 frames of the reference's two MP3 fixtures every granule-channel's scale factors + Huffman stage
        consume exactly part2_3_length bits, no bit pattern is left without a code, no value runs past line 576 --
        for the oracle's reader and for the product's (host code, no GPU needed);
  (iii) the synthesis window D reconstructs: the standard's analysis filterbank with C = D / 32 followed by the synthesis
        filterbank returns its input 481 samples late to better than -80 dB;
  (iv)  the oracle's decode of the two fixtures meets the source PCM the reference holds for them
        (soundkit-mp3/src/lib.rs:478-520 writes golden/mp3 from testdata/wav_stereo; testdata/mp3 has testdata/linear16
        beside it) at the measured SNR, after an offset search as soundkit-codec-fate does;
  (v)   the reference's own assertions on the fixtures: 16 kHz, 2 / 1 channels (soundkit-mp3/src/lib.rs:551-552).
The same (ii), (iv), (v) plus chunk invariance (lib.rs:745-760) for the GPU path: tests/test_mp3_fixtures_gpu.py."""
import ctypes as C
import os
import wave
from fractions import Fraction

import numpy as np
import pytest

from oracle import mp3_bitstream as ref
from oracle import mp3_hybrid, mp3_iso
from soundkit_amd import mp3

HERE = os.path.dirname(os.path.abspath(__file__))
STEREO = os.path.join(HERE, "golden", "mp3", "stereo16k_A_Tusk_encoded.mp3")
MONO = os.path.join(HERE, "golden", "mp3", "mono16k_A_Tusk.mp3")
SOURCE_STEREO = os.path.join(HERE, "golden", "wav_stereo_A_Tusk.wav")
SOURCE_MONO = os.path.join(HERE, "golden", "linear16_A_Tusk.s16le")

# measured with the oracle (f64) on the committed fixtures, see test_oracle_decode_meets_the_source_pcm
SNR_FLOOR_STEREO_BELOW_6K5 = 45.0   # measured 45.86 dB: 128 kbit/s; the encoder scales by 0.95 and low-passes at 7 kHz
SNR_FLOOR_STEREO_FULL_BAND = 22.0   # measured 22.50 dB (the 7-8 kHz band the encoder removed is in the source)
SNR_FLOOR_MONO_FULL_BAND = 20.0     # measured 20.54 dB: variable bit rate, 40 kbit/s on average
SNR_FLOOR_MONO_BELOW_2K = 30.0      # measured 30.39 dB
ENCODER_DELAY = 1681                # both files: 1105 + 576 samples between source and decoded output


def complete_prefix_code(hlen, hcod):
    if any(n < 1 or n > 19 or c >> n for n, c in zip(hlen, hcod)):
        return False
    if sum(Fraction(1, 1 << n) for n in hlen) != 1:
        return False
    words = sorted(format(c, "0%db" % n) for n, c in zip(hlen, hcod))
    return all(not b.startswith(a) for a, b in zip(words, words[1:]))


# ---- (i) + the two copies agree ------------------------------------------------------------------------------------------

def test_every_code_set_is_a_complete_prefix_code():
    t = mp3.iso_tables()
    seen = 0
    for i in range(32):
        table = t.big_values[i]
        if table.xlen == 0:
            assert i in (0, 4, 14)
            continue
        n = table.xlen * table.xlen
        assert complete_prefix_code([table.hlen[k] for k in range(n)], [table.hcod[k] for k in range(n)]), i
        seen += 1
    assert seen == 29
    for k in range(2):
        assert complete_prefix_code(list(t.count1_hlen[k]), list(t.count1_hcod[k]))
    assert [t.big_values[i].linbits for i in range(16, 32)] == [1, 2, 3, 4, 6, 8, 10, 13, 4, 5, 6, 7, 8, 9, 11, 13]
    assert [t.big_values[i].xlen for i in range(16)] == [0, 2, 3, 3, 0, 4, 4, 6, 6, 6, 8, 8, 8, 16, 0, 16]


def test_the_library_and_the_oracle_hold_the_same_numbers():
    t, o = mp3.iso_tables(), mp3_iso.tables()
    for i in range(32):
        table, want = t.big_values[i], o["big_values"][i]
        if want is None:
            assert table.xlen == 0
            continue
        n = want["xlen"] ** 2
        assert (table.xlen, table.linbits) == (want["xlen"], want["linbits"])
        assert [table.hlen[k] for k in range(n)] == want["hlen"] and [table.hcod[k] for k in range(n)] == want["hcod"]
    for k in range(2):
        assert list(t.count1_hlen[k]) == o["count1"][k]["hlen"] and list(t.count1_hcod[k]) == o["count1"][k]["hcod"]
    assert [list(r) for r in t.slen] == o["slen"]
    assert [[list(c) for c in row] for row in t.lsf_partitions] == o["lsf_partitions"]
    rates = [44100, 48000, 32000, 22050, 24000, 16000, 11025, 12000, 8000]
    for row, rate in enumerate(rates):
        assert t.rates_present[row] == 1
        assert list(t.long_offsets[row]) == o["bands"][rate][0] and list(t.short_offsets[row]) == o["bands"][rate][1]
        assert t.long_offsets[row][22] == 576 and t.short_offsets[row][13] == 192
    assert list(t.pretab) == o["pretab"]
    assert np.array_equal(np.asarray(list(t.window), np.float64), o["window"])  # multiples of 2^-16: exact in f32


def test_window_symmetry_and_scale():
    d = np.round(mp3_iso.tables()["window"] * 65536).astype(np.int64)
    assert d[0] == 0 and d[256] == 75038 and np.abs(d).max() == 75038
    for i in range(1, 256):
        assert d[512 - i] == (-d[i] if i & 63 else d[i]), i
    # the prototype low-pass behind D (D with the sign of every other 64-block undone) has DC gain 64
    proto = d * np.where((np.arange(512) // 64) % 2 == 1, -1, 1)
    assert abs(proto.sum() / 65536.0 - 64.0) < 0.01


# ---- (ii) exact consumption --------------------------------------------------------------------------------------------

def frames_of(path):
    data = open(path, "rb").read()
    frames, used = ref.scan(data)
    assert used == len(data)
    return data, frames


@pytest.mark.parametrize("path, n_frames, channels", [(STEREO, 82, 2), (MONO, 86, 1)])
def test_oracle_reader_consumes_exactly_part2_3_length(path, n_frames, channels):
    tables = mp3_iso.tables()
    data, frames = frames_of(path)
    assert len(frames) == n_frames
    mains = ref.main_data(frames, data)
    checked = used_tables = 0
    selected = set()
    for (off, h), main in zip(frames, mains):
        assert (h["sample_rate"], h["channels"]) == (16000, channels)  # (v): soundkit-mp3/src/lib.rs:551-552
        assert main is not None
        side = ref.parse_side_info(data[off:off + h["frame_bytes"]], h)
        start, first = 0, [[0] * 22, [0] * 22]
        for gr in range(h["granules"]):
            for ch in range(h["channels"]):
                s = side["gr"][gr][ch]
                end = start + s["part2_3_length"]
                bits = ref.MainBits(main, start)
                sl, _, _ = ref.scale_factors(tables, h, side, gr, ch, bits, first[ch])
                if gr == 0:
                    first[ch] = sl
                values = ref.huffman_granule(tables, h, s, bits, end)
                assert values is not None, (off, gr, ch)
                assert bits.pos == end, (off, gr, ch, bits.pos - end)  # not one stuffing bit, not one bit short
                selected.update(s["table_select"][:2 if s["window_switching"] else 3])
                start, checked = end, checked + 1
    assert checked == n_frames * channels
    assert len(selected - {0}) >= 8  # the fixtures reach a good part of Table B.7 (which ones: see DESIGN.md)


@pytest.mark.parametrize("path, n_frames, channels", [(STEREO, 82, 2), (MONO, 86, 1)])
def test_product_reader_consumes_exactly_part2_3_length_and_equals_the_oracle(path, n_frames, channels):
    tables = mp3_iso.tables()
    book = mp3.Codebook()
    data = open(path, "rb").read()
    found, used = mp3.scan(data)
    assert used == len(data) and len(found) == n_frames
    _, oracle_frames = frames_of(path)
    oracle_mains = ref.main_data(oracle_frames, data)
    reservoir = b""
    for k, info in enumerate(found):
        assert (info.sample_rate, info.channels) == (16000, channels)
        frame = data[info.offset:info.offset + info.frame_bytes]
        rc, side = mp3.parse_side_info(frame, info)
        assert rc == 0
        rc, main = mp3.main_data(frame, info, side, reservoir)
        assert rc == 0 and main == oracle_mains[k]
        rc, out = mp3.decode_main_data(book, info, side, main)
        assert rc == 0
        h = oracle_frames[k][1]
        want = ref.decode_main_data(tables, h, ref.parse_side_info(frame, h), main)
        for gr in range(info.granules):
            for ch in range(info.channels):
                g = out[gr][ch]
                assert g.status == 0
                assert g.part2_bits + g.part3_bits == side.gr[gr][ch].part2_3_length, (k, gr, ch)
                assert list(g.is_) == want[gr][ch]["is"]
                assert list(g.scalefac_l) == want[gr][ch]["scalefac_l"]
                assert [list(r) for r in g.scalefac_s] == want[gr][ch]["scalefac_s"]
        head = 4 + (2 if info.has_crc else 0) + info.side_info_bytes
        reservoir = (reservoir + frame[head:])[-2048:]
    book.close()


# ---- (iii) the window reconstructs -----------------------------------------------------------------------------------------

def test_analysis_with_d_over_32_then_synthesis_reconstructs_below_minus_80_db():
    """ISO/IEC 11172-3 C.1.3 (analysis: X shifted by 32, Z = C X, Y_k = sum_j Z[k + 64 j], S_i = sum_k cos((2i+1)(k-16) pi/64) Y_k)
    with C = D / 32, then the synthesis of oracle/mp3_hybrid.py with D.  Test code only."""
    d = mp3_iso.tables()["window"]
    c = d / 32.0
    m = np.cos((2 * np.arange(32)[:, None] + 1) * (np.arange(64)[None, :] - 16) * np.pi / 64)
    x = np.random.default_rng(7).uniform(-1, 1, 32 * 160)
    fifo, channel, y = np.zeros(512), mp3_hybrid.Channel(), []
    for t in range(len(x) // 32):
        fifo[32:] = fifo[:-32].copy()
        fifo[:32] = x[32 * t:32 * t + 32][::-1]
        y.append(channel.polyphase(m @ (c * fifo).reshape(8, 64).sum(axis=0), d))
    y = np.concatenate(y)
    delay = 481
    a, b = x[:len(x) - delay], y[delay:]
    snr = 10 * np.log10((a * a).sum() / ((a - b) ** 2).sum())
    assert snr > 80.0, snr  # measured 84.5 dB
    # and a damaged window does not: one entry off by 1 % of the peak is enough to fail the gate
    bad = d.copy()
    bad[200] += 0.01
    channel, y = mp3_hybrid.Channel(), []
    fifo[:] = 0
    for t in range(len(x) // 32):
        fifo[32:] = fifo[:-32].copy()
        fifo[:32] = x[32 * t:32 * t + 32][::-1]
        y.append(channel.polyphase(m @ (c * fifo).reshape(8, 64).sum(axis=0), bad))
    b = np.concatenate(y)[delay:]
    assert 10 * np.log10((a * a).sum() / ((a - b) ** 2).sum()) < 60.0


# ---- (iv) decoded PCM against the source the reference holds -------------------------------------------------------------------

def oracle_decode(path):
    data, frames = frames_of(path)
    dec = ref.Decoder(mp3_iso.tables())
    out = [dec.frame(data, off, h) for off, h in frames]
    assert all(p is not None for p in out)
    return np.concatenate(out)


def source_stereo():
    w = wave.open(SOURCE_STEREO, "rb")
    return np.frombuffer(w.readframes(w.getnframes()), "<i2").reshape(-1, 2).astype(np.float64) / 32768.0


def source_mono():
    return np.fromfile(SOURCE_MONO, "<i2").astype(np.float64)[:, None] / 32768.0


def snr_against_source(decoded, source, offset, f_max=None, rate=16000):
    """least-squares gain, then 10 log10(signal / error), optionally below f_max only (an encoder low-pass is not the
    decoder's error); -> (dB, gain)"""
    n = min(len(source), len(decoded) - offset)
    s, d = source[:n], decoded[offset:offset + n]
    if f_max is not None:
        keep = np.fft.rfftfreq(n, 1.0 / rate) < f_max
        s, d = np.fft.rfft(s, axis=0)[keep], np.fft.rfft(d, axis=0)[keep]
    gain = (np.conj(d) * s).sum().real / (np.abs(d) ** 2).sum()
    return 10 * np.log10((np.abs(s) ** 2).sum() / (np.abs(gain * d - s) ** 2).sum()), gain


def best_offset(decoded, source, span=3000):
    best = (-1e9, 0)
    for off in range(span):
        n = min(len(source), len(decoded) - off)
        if n < len(source) * 3 // 4:
            break
        s, d = source[:n, 0], decoded[off:off + n, 0]
        c = (s * d).sum() / np.sqrt((s * s).sum() * (d * d).sum())
        best = max(best, (c, off))
    return best[1]


def test_oracle_decode_meets_the_source_pcm():
    pcm = oracle_decode(STEREO)
    assert pcm.shape == (82 * 576, 2)
    src = source_stereo()
    off = best_offset(pcm, src)
    assert off == ENCODER_DELAY
    full, gain = snr_against_source(pcm, src, off)
    low, gain_low = snr_against_source(pcm, src, off, 6500)
    assert abs(gain_low - 1 / 0.95) < 0.002  # the encoder's input scale
    assert full > SNR_FLOOR_STEREO_FULL_BAND and low > SNR_FLOOR_STEREO_BELOW_6K5, (full, low)

    pcm = oracle_decode(MONO)
    assert pcm.shape == (86 * 576, 1)
    src = source_mono()
    off = best_offset(pcm, src)
    assert off == ENCODER_DELAY
    full, gain = snr_against_source(pcm, src, off)
    low, _ = snr_against_source(pcm, src, off, 2000)
    assert abs(gain - 1.0) < 0.005
    assert full > SNR_FLOOR_MONO_FULL_BAND and low > SNR_FLOOR_MONO_BELOW_2K, (full, low)


def test_a_wrong_table_entry_fails_the_gates():
    """the gates have teeth: swap two symbols of equal code length in a code set the stereo fixture uses -- still a
    complete prefix code, (ii) may still pass, (iv) does not"""
    import copy
    tables = copy.deepcopy(mp3_iso.tables())
    data, frames = frames_of(STEREO)
    counts = {}
    for off, h in frames:
        side = ref.parse_side_info(data[off:off + h["frame_bytes"]], h)
        for gr in range(h["granules"]):
            for ch in range(h["channels"]):
                for sel in side["gr"][gr][ch]["table_select"][:2]:
                    counts[sel] = counts.get(sel, 0) + 1
    select = max((s for s in counts if s and tables["big_values"][s]), key=lambda s: counts[s])
    table = tables["big_values"][select]
    n = table["xlen"]
    a, b = 1 * n + 1, 0 * n + 2 if n > 2 else 1  # (1, 1) <-> another cell of the same length
    same = [k for k in range(n * n) if k != a and table["hlen"][k] == table["hlen"][a]]
    b = same[0] if same else b
    for shared in tables["big_values"]:
        if shared is not None and shared["hcod"] is table["hcod"] and shared is not table:
            pass
    table["hcod"] = list(table["hcod"])
    table["hlen"] = list(table["hlen"])
    table["hcod"][a], table["hcod"][b] = table["hcod"][b], table["hcod"][a]
    table["hlen"][a], table["hlen"][b] = table["hlen"][b], table["hlen"][a]
    assert complete_prefix_code(table["hlen"], table["hcod"])
    dec = ref.Decoder(tables)
    out = [dec.frame(data, off, h) for off, h in frames]
    if any(p is None for p in out):
        return  # gate (ii) caught it
    pcm = np.concatenate(out)
    low, _ = snr_against_source(pcm, source_stereo(), ENCODER_DELAY, 6500)
    assert low < SNR_FLOOR_STEREO_BELOW_6K5 - 10
