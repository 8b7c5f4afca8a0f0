"""The reference's per-tool known answers, reproduced against oracle/aac_frontend.py.

Every test below is one `#[test]` of soundkit-aac-lc/src/{bitreader,ics,section,scalefactor,pulse,spectral,stereo,tns,
sfb,channel}.rs with the reference's literal inputs and expected outputs (name and line in the docstring).  They pin
the restatement tool by tool -- codebook decode per book, escape order, pulse, PNS (generator + energy normalisation),
the 13 stereo cases, TNS inverse quantisation and filtering -- where test_oracle_frontend.py pins whole access units.

Where the reference injects a toy decoder (MiniSpectralDecoder, MINI_SF, SequenceScaleFactorDecoder,
FillSpectralDecoder) the oracle's band_reader / read_delta hooks take the same toy.  Not reproduced, because they test
Rust scaffolding that has no counterpart outside the reference (placeholder decoders returning NotImplemented, VlcTable
construction from custom entries, the zero-spectral compatibility reader): spectral.rs:2875-2906 (toy VLC tuple
tables), :3182-3227, scalefactor.rs:402-411, channel.rs:279-301, :413-438, :524-543.  The prefix-freeness tests
(spectral.rs:2575-2743, scalefactor.rs:377-400) are the Kraft/prefix tests of tests/test_aac_frontend.py.

The last part runs the same scenarios as whole access units through the product's front-end (host build), so the
tool-level pins reach the product through oracle == product.
"""
import numpy as np
import pytest

from oracle import aac_frontend as OF
from test_aac_frontend import build_bits

F = np.float32
ONLY_LONG, EIGHT_SHORT = OF.ONLY_LONG, OF.EIGHT_SHORT


def long_ics(max_sfb):
    return OF.Ics.make(ONLY_LONG, 0, max_sfb)


def short_ics(max_sfb, group_len):
    return OF.Ics.make(EIGHT_SHORT, 0, max_sfb, group_len)


def long_prefix(gain, max_sfb, fields):
    return OF.Channel.prefix(gain, long_ics(max_sfb), build_bits(fields))


def short_prefix(gain, max_sfb, group_len, fields):
    return OF.Channel.prefix(gain, short_ics(max_sfb, group_len), build_bits(fields))


def mini_sf(r):
    """scalefactor.rs:306-322 MINI_SF: 0 -> 0, 10 -> +1, 11 -> -1 (spectral.rs:2473-2477 is its first entry alone)"""
    if not r.flag():
        return 0
    return -1 if r.flag() else 1


def feed(values):
    """MiniSpectralDecoder (spectral.rs:2511-2528): hands out the given quantised values, codebook must be 1"""
    state = {"pos": 0}

    def read(_r, book, count):
        assert book == 1
        out = list(values[state["pos"]:state["pos"] + count])
        state["pos"] += count
        return out
    read.state = state
    return read


def decoder(sf_index=4):
    d = OF.Decoder(bytes([0x12, 0x10]))
    d.sf_index = sf_index
    return d


def raises(kind, message, fn, *args, **kw):
    with pytest.raises(OF.AacError) as exc:
        fn(*args, **kw)
    assert exc.value.kind == kind and str(exc.value) == message, (exc.value.kind, str(exc.value))


# ---- bitreader.rs:190-254 ---------------------------------------------------------------------------------------
def test_reads_msb_first_across_byte_boundary():
    r = OF.Bits(bytes([0b10101100, 0b01100001]))
    assert [r.read(3), r.read(5), r.read(4), r.read(4)] == [0b101, 0b01100, 0b0110, 0b0001]
    assert r.remaining() == 0


def test_reports_eof_without_advancing():
    r = OF.Bits(bytes([0xFF]))
    assert r.read(7) == 0x7F
    raises("UnexpectedEof", "unexpected end of AAC bitstream: requested 2 bits, 1 bits remain", r.read, 2)
    assert r.pos == 7 and r.read(1) == 1


def test_peeks_without_advancing_logical_position():
    r = OF.Bits(bytes([0b10101100, 0b01100001, 0b11110000]))
    assert r.read(3) == 0b101
    assert r.peek(9) == 0b011000110 and r.pos == 3
    assert r.read(9) == 0b011000110 and r.pos == 12


def test_skips_across_prefetched_bytes():
    r = OF.Bits(bytes([0b11110000, 0b10100101, 0b00111100, 0b01011010, 0b00001111]))
    assert r.peek(16) == 0b1111000010100101
    r.read(20)
    assert r.pos == 20 and r.read(8) == 0b11000101


# ---- ics.rs:118-180 ------------------------------------------------------------------------------------------------
def test_parses_only_long_ics_info():
    ics = OF.Ics(OF.Bits(build_bits([(0, 1), (0, 2), (1, 1), (42, 6), (0, 1)])))
    assert (ics.sequence, ics.shape, ics.max_sfb, ics.num_windows, ics.groups, ics.group_len[0]) == (0, 1, 42, 1, 1, 1)


def test_parses_eight_short_window_groups():
    ics = OF.Ics(OF.Bits(build_bits([(0, 1), (2, 2), (0, 1), (12, 4), (0b1100100, 7)])))
    assert (ics.sequence, ics.shape, ics.max_sfb, ics.num_windows, ics.groups) == (2, 0, 12, 8, 5)
    assert ics.group_len == [3, 1, 2, 1, 1]


def test_ics_rejections():
    raises("InvalidConfig", "ICS reserved bit is set", OF.Ics, OF.Bits(build_bits([(1, 1)])))
    raises("UnsupportedFeature", "AAC prediction", OF.Ics, OF.Bits(build_bits([(0, 1), (0, 2), (0, 1), (20, 6), (1, 1)])))


# ---- section.rs:161-262 ----------------------------------------------------------------------------------------------
def test_parses_single_long_window_section():
    ch = long_prefix(0, 4, [(1, 4), (4, 5)])
    assert ch.books == [[1, 1, 1, 1]]


def test_detects_all_zero_sections():
    assert long_prefix(0, 3, [(0, 4), (3, 5)]).books == [[0, 0, 0]]


def test_parses_long_window_escaped_section_length():
    ch = long_prefix(0, 33, [(2, 4), (31, 5), (2, 5)])
    assert len(ch.books[0]) == 33 and ch.books[0][0] == 2 and ch.books[0][32] == 2


def test_parses_short_window_sections_per_group():
    ch = short_prefix(0, 3, [3, 5], [(1, 4), (1, 3), (0, 4), (2, 3), (5, 4), (3, 3)])
    assert ch.books == [[1, 0, 0], [5, 5, 5]]


def test_section_rejections():
    raises("InvalidBitstream", "reserved AAC section codebook", long_prefix, 0, 1, [(12, 4)])
    raises("InvalidBitstream", "section length exceeds max_sfb", long_prefix, 0, 2, [(1, 4), (3, 5)])


# ---- scalefactor.rs:327-375 ----------------------------------------------------------------------------------------
def test_parses_zero_sections_without_reading_scalefactor_huffman():
    ch = long_prefix(100, 2, [(0, 4), (2, 5)])
    ch.read_scalefactors(OF.Bits(b""))
    assert ch.values == [[("Zero", 0), ("Zero", 0)]]


def test_parses_spectral_scalefactors_with_vlc_decoder():
    ch = long_prefix(100, 2, [(1, 4), (2, 5)])
    ch.read_scalefactors(OF.Bits(bytes([0b10110000])), read_delta=mini_sf)
    assert ch.values == [[("Spectral", 101), ("Spectral", 100)]]


def test_parses_first_noise_scalefactor_from_pcm_bits():
    ch = long_prefix(100, 1, [(13, 4), (1, 5)])
    ch.read_scalefactors(OF.Bits(build_bits([(5, 9)])))
    assert ch.values == [[("Noise", -241)]]


def test_parses_spectral_scalefactors_with_standard_aac_codebook():
    ch = long_prefix(100, 3, [(1, 4), (3, 5)])
    ch.read_scalefactors(OF.Bits(bytes([0b10001010])))
    assert ch.values == [[("Spectral", 99), ("Spectral", 99), ("Spectral", 100)]]


# ---- pulse.rs:43-77 ------------------------------------------------------------------------------------------------
def test_parses_one_pulse():
    assert OF.read_pulse(OF.Bits(build_bits([(0, 2), (12, 6), (7, 5), (9, 4)]))) == (12, [(7, 9)])


def test_parses_four_pulses():
    start, pulses = OF.read_pulse(OF.Bits(build_bits([(3, 2), (4, 6), (1, 5), (2, 4), (3, 5), (4, 4), (5, 5), (6, 4), (7, 5), (8, 4)])))
    assert start == 4 and len(pulses) == 4 and pulses[3] == (7, 8)


# ---- spectral.rs:2745-2872: one codeword of every standard codebook --------------------------------------------------
@pytest.mark.parametrize("book,fields,count,want", [
    (1, [(0b0, 1), (0b10100, 5), (0b0, 1)], 12, [0, 0, 0, 0, 0, 0, 0, 1, 0, 0, 0, 0]),       # :2745-2755
    (2, [(0b000, 3), (0b00111, 5)], 8, [0, 0, 0, 0, 0, 0, 0, 1]),                            # :2757-2763
    (3, [(0b0, 1), (0b1000, 4), (1, 1)], 8, [0, 0, 0, 0, -1, 0, 0, 0]),                      # :2765-2773
    (4, [(0b0111, 4), (0b0101, 4), (0, 1)], 8, [0, 0, 0, 0, 1, 0, 0, 0]),                    # :2775-2783
    (5, [(0b0, 1), (0b1010, 4)], 4, [0, 0, 0, 1]),                                           # :2785-2793
    (6, [(0b0000, 4), (0b0001, 4)], 4, [0, 0, 1, 0]),                                        # :2795-2803
    (7, [(0b0, 1), (0b100, 3), (1, 1)], 4, [0, 0, -1, 0]),                                   # :2805-2813
    (8, [(0b000, 3), (1, 1), (0, 1)], 2, [-1, 1]),                                           # :2815-2821
    (9, [(0b0, 1), (0b100, 3), (0, 1)], 4, [0, 0, 1, 0]),                                    # :2823-2829
    (10, [(0b0000, 4), (1, 1), (0, 1)], 2, [-1, 1]),                                         # :2831-2839
    (11, [(0b0000, 4), (0b0001, 4), (0, 1), (1, 1)], 4, [0, 0, 1, -1]),                      # :2841-2849
    (11, [(0b111000010, 9), (1, 1), (0b00101, 5)], 2, [-21, 0]),                             # :2851-2858 (escape)
])
def test_standard_spectral_decoder_per_codebook(book, fields, count, want):
    assert OF.read_band(OF.Bits(build_bits(fields)), book, count) == want


def test_standard_spectral_decoder_rejects_reserved_codebook():            # spectral.rs:2861-2872
    raises("InvalidBitstream", "reserved AAC spectral codebook", OF.read_band, OF.Bits(b""), 12, 4)


# ---- spectral.rs:2908-3180: band decode, pulse, PNS ----------------------------------------------------------------
def test_decodes_and_dequantizes_long_spectral_band():                     # spectral.rs:2908-2943
    ch = long_prefix(100, 2, [(1, 4), (1, 5), (0, 4), (1, 5)])
    ch.scale = ch.read_scalefactors(OF.Bits(bytes([0])), read_delta=mini_sf)
    assert ch.values[0][0] == ("Spectral", 100)
    out = decoder(4).spectrum(OF.Bits(b""), ch, False, band_reader=feed([1, -1, 8, 0]), length=8)
    assert abs(out[0] - 1.0) < 1e-6 and abs(out[1] + 1.0) < 1e-6 and abs(out[2] - 16.0) < 1e-5
    assert out[3] == 0.0 and not out[4:8].any()


def test_applies_pulse_data_before_long_dequantization():                  # spectral.rs:2945-2983
    ch = long_prefix(100, 2, [(1, 4), (2, 5)])
    ch.scale = ch.read_scalefactors(OF.Bits(bytes([0])), read_delta=mini_sf)
    ch.pulse = (0, [(0, 3), (2, 2)])
    d = decoder()
    out = d.spectrum(OF.Bits(b""), ch, False, off=[0, 4, 8], band_reader=feed([1, -1, 0, 4, 5, 6, 7, 8]), length=8)
    assert d.quant == [4, -1, -2, 4, 5, 6, 7, 8]
    assert abs(out[0] - F(4.0) ** F(4.0 / 3.0)) < 1e-5 and abs(out[2] + F(2.0) ** F(4.0 / 3.0)) < 1e-5


def energy(x):
    e = F(0.0)
    for v in x:
        e = e + F(v) * F(v)
    return e


def test_reconstructs_long_pns_band_from_noise_scalefactor():              # spectral.rs:2985-3026
    ch = long_prefix(100, 1, [(13, 4), (1, 5)])
    ch.scale = ch.read_scalefactors(OF.Bits(build_bits([(346, 9)])))
    assert ch.values[0][0] == ("Noise", 100)
    d = decoder()
    first = d.spectrum(OF.Bits(b""), ch, False, off=[0, 4], length=4)
    assert first.any() and abs(energy(first) - 1.0) < 1e-6
    # SpectralCoefficients::new seeds its own generator (spectral.rs:2459), so a second, fresh decode repeats the band
    again = decoder().spectrum(OF.Bits(b""), ch, False, off=[0, 4], length=4)
    assert np.array_equal(first, again)
    # the generator itself: s = s*1664525 + 1013904223 from 0x1f2e3d4c, sample = high half as i16 (spectral.rs:2416-2445)
    s, vals = 0x1F2E3D4C, []
    for _ in range(4):
        s = (s * 1664525 + 1013904223) & 0xFFFFFFFF
        vals.append(float(np.int16(np.uint16(s >> 16))))
    norm = 1.0 / np.sqrt(sum(v * v for v in vals))
    assert np.allclose(first, [v * norm for v in vals], rtol=3e-7, atol=0)


def test_reconstructs_grouped_short_pns_bands():                           # spectral.rs:3028-3066
    ch = short_prefix(100, 1, [2, 6], [(13, 4), (1, 3), (13, 4), (1, 3)])
    ch.scale = ch.read_scalefactors(OF.Bits(build_bits([(346, 9), (0, 1)])), read_delta=mini_sf)
    out = decoder().spectrum(OF.Bits(b""), ch, False, off=[0, 4])
    for start in (0, 128, 256):
        assert abs(energy(out[start:start + 4]) - 1.0) < 1e-6
    assert not out[4:128].any()


def test_rejects_pulse_target_in_zero_codebook_band():                     # spectral.rs:3068-3105
    ch = long_prefix(100, 2, [(1, 4), (1, 5), (0, 4), (1, 5)])
    ch.scale = ch.read_scalefactors(OF.Bits(bytes([0])), read_delta=mini_sf)
    ch.pulse = (0, [(4, 1)])
    raises("InvalidBitstream", "pulse target is not in a spectral band", decoder().spectrum, OF.Bits(b""), ch, False,
           off=[0, 4, 8], band_reader=feed([1, -1, 0, 4]), length=8)


def test_decodes_long_spectral_band_with_standard_codebook_1():            # spectral.rs:3107-3130
    ch = long_prefix(100, 1, [(1, 4), (1, 5)])
    ch.scale = ch.read_scalefactors(OF.Bits(bytes([0])), read_delta=mini_sf)
    d = decoder()
    out = d.spectrum(OF.Bits(bytes([0])), ch, False, off=[0, 8], length=8)
    assert not out.any() and d.quant == [0] * 8


def test_decodes_grouped_short_window_spectral_bands():                    # spectral.rs:3132-3180
    ch = short_prefix(100, 2, [2, 6], [(1, 4), (1, 3), (0, 4), (1, 3), (0, 4), (1, 3), (1, 4), (1, 3)])
    ch.scale = ch.read_scalefactors(OF.Bits(bytes([0])), read_delta=mini_sf)
    reader = feed(list(range(1, 33)))
    d = decoder()
    d.spectrum(OF.Bits(b""), ch, False, off=[0, 4, 8], band_reader=reader)
    q = d.quant
    assert q[0:8] == [1, 2, 3, 4, 0, 0, 0, 0] and q[128:136] == [5, 6, 7, 8, 0, 0, 0, 0]
    for w in range(2, 8):
        first = 9 + 4 * (w - 2)
        assert q[128 * w:128 * w + 8] == [0, 0, 0, 0, first, first + 1, first + 2, first + 3]
    assert reader.state["pos"] == 32


# ---- channel.rs:303-522 --------------------------------------------------------------------------------------------
def test_parses_common_window_mid_side_some_mask():                        # channel.rs:303-336
    r = OF.Bits(build_bits([(1, 1), (0, 1), (0, 2), (0, 1), (3, 6), (0, 1), (1, 2), (1, 1), (0, 1), (1, 1)]))
    assert r.flag()
    ics = OF.Ics(r)
    mode, used = OF.read_ms_mask(r, ics)
    assert ics.max_sfb == 3 and mode == 1 and len(used) == 1 and used[0] == [True, False, True]


def test_rejects_reserved_mid_side_mode():                                 # channel.rs:338-355
    r = OF.Bits(build_bits([(1, 1), (0, 1), (0, 2), (0, 1), (1, 6), (0, 1), (3, 2)]))
    r.flag()
    raises("InvalidBitstream", "reserved mid/side mask mode", OF.read_ms_mask, r, OF.Ics(r))


def fill(_r, book, count):
    """FillSpectralDecoder (channel.rs:578-596): +1, -1, +1, ..."""
    assert book == 1
    return [1 if i % 2 == 0 else -1 for i in range(count)]


NONZERO = [(100, 8), (0, 1), (0, 2), (0, 1), (1, 6), (0, 1), (1, 4), (1, 5), (0, 1), (0, 1), (0, 1), (0, 1)]


def test_channel_stream_long_decode_and_pulse():                           # channel.rs:382-411, 440-483
    one_bit = lambda r: int(r.flag())                       # OneBitScaleFactorDecoder, channel.rs:566-576
    ch = OF.Channel(OF.Bits(build_bits(NONZERO[:8] + [(1, 1)] + NONZERO[9:])), None, read_delta=one_bit)
    assert ch.values == [[("Spectral", 101)]] and ch.pulse is None and ch.tns is None
    ch = OF.Channel(OF.Bits(build_bits(NONZERO)), None, read_delta=one_bit)
    assert ch.values == [[("Spectral", 100)]]
    d = decoder()
    out = d.spectrum(OF.Bits(b""), ch, False, off=[0, 2], band_reader=fill, length=2)
    assert out[0] > 0 and out[1] < 0 and d.quant == [1, -1]
    ch.pulse = (0, [(0, 2)])
    d.spectrum(OF.Bits(b""), ch, False, off=[0, 2], band_reader=fill, length=2)
    assert d.quant == [3, -1]


def test_channel_stream_dispatches_short_spectral_decode():                # channel.rs:485-522
    fields = [(100, 8), (0, 1), (2, 2), (0, 1), (1, 4), (0b1111111, 7), (1, 4), (1, 3), (0, 1), (0, 1), (0, 1), (0, 1)]
    ch = OF.Channel(OF.Bits(build_bits(fields)), None)
    assert ch.ics.group_len == [8]
    calls = []

    def counted(r, book, count):
        calls.append(book)
        return fill(r, book, count)
    d = decoder()
    d.spectrum(OF.Bits(b""), ch, False, off=[0, 4], band_reader=counted)
    assert calls == [1] * 8 and d.quant[0:4] == [1, -1, 1, -1] and d.quant[128:132] == [1, -1, 1, -1]


# ---- stereo.rs:463-731 -----------------------------------------------------------------------------------------------
def f32(*v):
    return np.array(v, np.float32)


NONE, ALL = (0, None), (2, None)


def some(used):
    return (1, used)


def test_long_mid_side_masks():                                            # stereo.rs:463-510
    off = [0, 2, 4]
    left, right = f32(10, 20, 30, 40), f32(1, 2, 3, 4)
    OF.apply_mid_side(NONE, long_ics(2), off, left, right)
    assert left.tolist() == [10, 20, 30, 40] and right.tolist() == [1, 2, 3, 4]
    OF.apply_mid_side(ALL, long_ics(2), off, left, right)
    assert left.tolist() == [11, 22, 33, 44] and right.tolist() == [9, 18, 27, 36]
    left, right = f32(10, 20, 30, 40), f32(1, 2, 3, 4)
    OF.apply_mid_side(some([[False, True]]), long_ics(2), off, left, right)
    assert left.tolist() == [10, 20, 33, 44] and right.tolist() == [1, 2, 27, 36]


def test_intensity_long_reconstructs_right_and_skips_mid_side_transform():  # stereo.rs:512-544
    off = [0, 2, 4]
    lch = long_prefix(100, 2, [(1, 4), (2, 5)])
    rch = long_prefix(100, 2, [(14, 4), (1, 5), (1, 4), (1, 5)])
    deltas = iter([4, 0])
    rch.scale = rch.read_scalefactors(OF.Bits(b""), read_delta=lambda r: next(deltas))
    left, right = f32(8, 16, 10, 20), f32(0, 0, 1, 2)
    OF.apply_intensity(ALL, long_ics(2), off, rch.books, rch.scale, left, right)
    OF.apply_mid_side(ALL, long_ics(2), off, left, right, lch.books, rch.books)
    assert left.tolist() == [8, 16, 11, 22] and right.tolist() == [-4, -8, 9, 18]


def test_mid_side_long_skips_noise_bands_on_either_channel():              # stereo.rs:546-567
    lch = long_prefix(100, 3, [(1, 4), (1, 5), (13, 4), (1, 5), (1, 4), (1, 5)])
    rch = long_prefix(100, 3, [(1, 4), (2, 5), (13, 4), (1, 5)])
    left, right = f32(10, 20, 30, 40, 50, 60), f32(1, 2, 3, 4, 5, 6)
    OF.apply_mid_side(ALL, long_ics(3), [0, 2, 4, 6], left, right, lch.books, rch.books)
    assert left.tolist() == [11, 22, 30, 40, 50, 60] and right.tolist() == [9, 18, 3, 4, 5, 6]


def test_long_mid_side_rejections():                                       # stereo.rs:569-615
    used = [[True, True], [True, True]]
    raises("NotImplemented", "grouped mid/side stereo reconstruction", OF.apply_mid_side, some(used), long_ics(2),
           [0, 2, 4], f32(10, 20, 30, 40), f32(1, 2, 3, 4))
    raises("InvalidConfig", "mid/side channel buffers have different lengths", OF.apply_mid_side, ALL, long_ics(2),
           [0, 2, 4], f32(10, 20, 30, 40), f32(1, 2, 3))
    raises("InvalidConfig", "mid/side scale-factor band exceeds channel buffer", OF.apply_mid_side, ALL, long_ics(2),
           [0, 2, 5], f32(10, 20, 30, 40), f32(1, 2, 3, 4))


def sample_channel():
    return np.arange(1024, dtype=np.float32)


def test_short_mid_side_masks():                                           # stereo.rs:617-680
    off = [0, 2, 4]
    left, right = sample_channel(), np.ones(1024, np.float32)
    OF.apply_mid_side(NONE, short_ics(2, [8]), off, left, right)
    assert np.array_equal(left, sample_channel()) and np.array_equal(right, np.ones(1024, np.float32))
    OF.apply_mid_side(ALL, short_ics(2, [2, 6]), off, left, right)
    assert left[0:4].tolist() == [1, 2, 3, 4] and right[0:4].tolist() == [-1, 0, 1, 2]
    assert left[128:132].tolist() == [129, 130, 131, 132] and right[128:132].tolist() == [127, 128, 129, 130]
    assert left[896:900].tolist() == [897, 898, 899, 900] and right[896:900].tolist() == [895, 896, 897, 898]
    left, right = sample_channel(), np.ones(1024, np.float32)
    OF.apply_mid_side(some([[True, False], [False, True]]), short_ics(2, [2, 6]), off, left, right)
    assert left[0:4].tolist() == [1, 2, 2, 3] and right[0:4].tolist() == [-1, 0, 1, 1]
    assert left[128:132].tolist() == [129, 130, 130, 131] and right[128:132].tolist() == [127, 128, 1, 1]
    assert left[256:260].tolist() == [256, 257, 259, 260] and right[256:260].tolist() == [1, 1, 257, 258]
    assert left[896:900].tolist() == [896, 897, 899, 900] and right[896:900].tolist() == [1, 1, 897, 898]


def test_intensity_short_reconstructs_grouped_windows():                   # stereo.rs:682-705
    ics = short_ics(1, [8])
    rch = OF.Channel.prefix(100, ics, build_bits([(15, 4), (1, 3)]))
    rch.scale = rch.read_scalefactors(OF.Bits(b""), read_delta=lambda r: 0)
    left, right = sample_channel(), np.zeros(1024, np.float32)
    OF.apply_intensity(NONE, ics, [0, 2], rch.books, rch.scale, left, right)
    assert right[0:4].tolist() == [-0.0, -1.0, 0.0, 0.0]
    assert right[128:132].tolist() == [-128.0, -129.0, 0.0, 0.0] and right[896:900].tolist() == [-896.0, -897.0, 0.0, 0.0]


def test_short_some_mask_rejects_missing_group_coverage():                 # stereo.rs:707-731
    raises("InvalidConfig", "mid/side mask does not cover requested short-window groups/bands", OF.apply_mid_side,
           some([[False, False]]), short_ics(2, [2, 6]), [0, 2, 4], sample_channel(), np.ones(1024, np.float32))


# ---- tns.rs:293-420 --------------------------------------------------------------------------------------------------
def test_parses_tns_data():                                                # tns.rs:293-344
    assert OF.read_tns(OF.Bits(build_bits([(0, 2)])), long_ics(1)) == [(False, [])]
    fields = [(1, 2), (1, 1), (12, 6), (2, 5), (1, 1), (0, 1), (0b0111, 4), (0b1111, 4)]
    (res, filters), = OF.read_tns(OF.Bits(build_bits(fields)), long_ics(1))
    assert res and filters == [(12, 2, True, 4, [7, -1])]
    windows = OF.read_tns(OF.Bits(build_bits([(0, 1)] * 8)), short_ics(4, [8]))
    assert len(windows) == 8 and all(not f for _, f in windows)


def test_maps_lc_tns_max_bands_for_common_sample_rates():                  # tns.rs:346-364
    assert OF.tns_max_bands(4, False) == 42 and OF.tns_max_bands(4, True) == 14


def test_inverse_quantizes_tns_coefficients_from_transmitted_bits():       # tns.rs:366-384
    for encoded, bits, res_bits, want in [(1, 4, 4, -0.2079117), (-8, 4, 4, 0.99573416), (-1, 4, 4, 0.18374951),
                                          (-4, 3, 4, 0.67369562)]:
        assert abs(OF.Decoder.tns_coefficient(encoded, bits, res_bits) - F(want)) < 1e-6


def test_applies_forward_tns_filter_to_long_coefficients():                # tns.rs:386-420
    coef = np.zeros(1024, np.float32)
    coef[0], coef[1] = 1.0, 2.0
    OF.apply_tns([(True, [(2, 1, False, 4, [1])])], long_ics(1), [0, 4, 8], 2, coef)
    assert abs(coef[0] - 1.0) < 1e-6 and abs(coef[1] - F(2.0 - 0.2079117)) < 1e-6


# ---- sfb.rs:159-203 --------------------------------------------------------------------------------------------------
def test_band_layouts():
    for index in (3, 4):
        off = OF.long_offsets(index)
        assert len(off) - 1 == 49 and off[:4] == [0, 4, 8, 12] and off[-1] == 1024 and (off[48], off[49]) == (928, 1024)
    off = OF.long_offsets(8)
    assert len(off) - 1 == 43 and (off[0], off[1]) == (0, 8) and off[-1] == 1024
    off = OF.long_offsets(12)
    assert len(off) - 1 == 40 and (off[39], off[40]) == (944, 1024)
    off = OF.short_offsets(4)
    assert len(off) - 1 == 14 and (off[6], off[7]) == (28, 36) and off[-1] == 128
    d = decoder(-1)
    raises("UnsupportedFeature", "explicit sample-rate scalefactor bands", d.offsets, long_ics(1))


# ---- the same tools as whole access units: product front-end == oracle -----------------------------------------------
def _cases():
    """Access units built field by field (as decoder.rs:576-736 builds its own) that put the per-tool vectors above
    into a real element: codebook-1 tuple + pulse, long PNS band, intensity + mid/side pair, a TNS filter."""
    sce = [(0, 3), (0, 4)]
    ics_long = lambda max_sfb: [(0, 1), (0, 2), (0, 1), (max_sfb, 6), (0, 1)]
    yield "pulse", 1, build_bits(sce + [(100, 8)] + ics_long(2) + [(1, 4), (2, 5), (0, 1), (0, 1)]
                                + [(1, 1), (1, 2), (0, 6), (3, 5), (3, 4), (2, 5), (2, 4)] + [(0, 1), (0, 1)]
                                + [(0b10100, 5), (0b0, 1)])       # quads {0,0,0,1}{0,0,0,0}; pulses at bins 3 and 5
    yield "pns", 1, build_bits(sce + [(100, 8)] + ics_long(1) + [(13, 4), (1, 5), (346, 9), (0, 1), (0, 1), (0, 1)])
    yield "tns", 1, build_bits(sce + [(100, 8)] + ics_long(2) + [(1, 4), (2, 5), (0, 1), (0, 1), (0, 1)]
                               + [(1, 1), (1, 2), (1, 1), (49, 6), (1, 5), (0, 1), (0, 1), (1, 4)] + [(0, 1)]
                               + [(0b10100, 5), (0b10100, 5)])
    cpe = [(1, 3), (0, 4), (1, 1)] + ics_long(2) + [(2, 2)]        # common window, mid/side on every band
    left = [(100, 8), (1, 4), (2, 5), (0, 1), (0, 1), (0, 1), (0, 1), (0, 1), (0b10100, 5), (0b10100, 5)]
    lens, codes = OF._BOOKS["sf"]
    plus4 = (codes[64], lens[64])                                  # scalefactor delta +4 = index 60 + 4
    right = [(100, 8), (14, 4), (1, 5), (1, 4), (1, 5), plus4, (0, 1), (0, 1), (0, 1), (0, 1), (0b10100, 5)]
    yield "intensity+ms", 2, build_bits(cpe + left + right)


@pytest.mark.parametrize("name,channels,au", list(_cases()), ids=[c[0] for c in _cases()])
def test_tool_access_units_product_equals_oracle(name, channels, au):
    from soundkit_amd import aac_lc
    asc = bytes([0x12, 0x08 if channels == 1 else 0x10])
    want, wseq, wshape = OF.Decoder(asc).decode_access_unit(au)
    got, gseq, gshape = aac_lc.AacLcFrontEnd(asc).parse(au)
    assert (wseq, wshape) == (gseq, gshape)
    assert np.array_equal(want.view(np.uint32), got.view(np.uint32))
    assert np.count_nonzero(want) >= (4 if name == "pns" else 2)
    if name == "pulse":   # bins 3 and 5: |1|+3 and 0 -> -2 (a zero takes the negative amplitude, spectral.rs:2239-2243)
        q = [0, 0, 0, 4, 0, -2, 0, 0]
        assert np.allclose(want[0, :8], [np.sign(v) * abs(v) ** (4.0 / 3.0) for v in q], rtol=1e-6)
    if name == "intensity+ms":   # right band 0 = left * 2^(-4/4) * (+1 flipped by the mask)
        assert np.array_equal(want[1, :4], want[0, :4] * F(0.5) * F(-1.0))
