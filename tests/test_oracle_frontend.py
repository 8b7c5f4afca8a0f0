"""oracle/aac_frontend.py (the CPU restatement of the reference's AAC-LC front-end) pinned by the reference's own
bitstream vectors, then used as the checker for the product's C++ front-end (csrc/aac_frontend.cpp): bit-identical
spectra and window fields on every access unit of the fixtures, and the same error (kind and message) on mutated
access units.  Two implementations written independently from the same reference files: a slip in either shows up
as a mismatch here rather than as slightly wrong audio."""
import os

import numpy as np
import pytest

from oracle import aac_frontend as OF
from soundkit_amd import aac_lc
from test_aac_frontend import FIXTURES, SILENT_SCE, build_bits

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "aac")


# ---- the oracle against the reference's vectors (decoder.rs:445-736, config.rs) ---------------------------------
def test_oracle_asc_cases():
    assert OF.parse_asc(bytes([0x12, 0x10])) == (4, 44100, 2)
    assert OF.parse_asc(bytes([0x12, 0x08])) == (4, 44100, 1)
    assert OF.parse_asc(bytes([0x11, 0x90])) == (3, 48000, 2)
    for asc, kind in [(bytes([0x2B, 0x92, 0x08, 0x00]), "UnsupportedFeature"), (bytes([0x0A, 0x10]), "UnsupportedAudioObjectType"),
                      (bytes([0x12, 0x00]), "UnsupportedFeature"), (bytes([0x12, 0x18]), "UnsupportedChannelConfig"),
                      (bytes([0x16, 0x90]), "UnsupportedSamplingFrequencyIndex"), (bytes([0x12, 0x14]), "UnsupportedFeature"),
                      (bytes([0x00, 0x00]), "InvalidAudioObjectType"), (bytes([0x12]), "UnexpectedEof")]:
        with pytest.raises(OF.AacError) as exc:
            OF.parse_asc(asc)
        assert exc.value.kind == kind, asc.hex()


def test_oracle_access_unit_vectors():
    mono, stereo = OF.Decoder(bytes([0x12, 0x08])), OF.Decoder(bytes([0x12, 0x10]))
    with pytest.raises(OF.AacError) as exc:                       # decoder.rs:481-492
        stereo.decode_access_unit(bytes([0b00100000]))
    assert exc.value.kind == "UnexpectedEof" and "requested 8 bits, 0 bits remain" in str(exc.value)
    for extra in ([], [(7, 3)], [(6, 3), (1, 4), (0, 8), (7, 3)]):  # decoder.rs:494-538
        coeffs, seq, shape = mono.decode_access_unit(build_bits(SILENT_SCE + extra))
        assert coeffs.shape == (1, 1024) and not coeffs.any() and seq == [0] and shape == [0]
    with pytest.raises(OF.AacError) as exc:                       # decoder.rs:540-557
        mono.decode_access_unit(build_bits(SILENT_SCE + [(6, 3), (1, 4), (13, 4), (0, 4)]))
    assert exc.value.kind == "UnsupportedFeature" and "SBR/HE-AAC extension payload" in str(exc.value)
    with pytest.raises(OF.AacError) as exc:                       # decoder.rs:559-574
        mono.decode_access_unit(build_bits(SILENT_SCE + [(0, 3), (1, 4)]))
    assert "multiple channel elements" in str(exc.value)
    au = build_bits([(0, 3), (0, 4), (100, 8), (0, 1), (0, 2), (0, 1), (1, 6), (0, 1), (1, 4), (1, 5), (0, 1), (0, 1), (0, 1),
                     (0, 1), (0b10100, 5)])                        # decoder.rs:576-604
    coeffs, _, _ = mono.decode_access_unit(au)
    want = np.zeros(1024, np.float32)
    want[3] = 1.0
    assert np.array_equal(coeffs[0], want)
    cpe = [(1, 3), (0, 4), (1, 1), (0, 1), (0, 2), (0, 1), (1, 6), (0, 1), (0, 2)]
    chan = [(100, 8), (0, 4), (1, 5), (0, 1), (0, 1), (0, 1)]
    coeffs, seq, _ = stereo.decode_access_unit(build_bits(cpe + chan + chan))  # decoder.rs:606-640
    assert coeffs.shape == (2, 1024) and not coeffs.any() and seq == [0, 0]
    fields = [(1, 3), (0, 4), (0, 1), (100, 8), (0, 1), (2, 2), (0, 1), (1, 4), (0, 7)] + [(0, 4), (1, 3)] * 8
    fields += [(0, 1), (0, 1), (0, 1), (100, 8), (0, 1), (0, 2), (0, 1), (1, 6), (0, 1), (0, 4), (1, 5), (0, 1), (0, 1), (0, 1)]
    coeffs, seq, _ = OF.Decoder(bytes([0x11, 0x90])).decode_access_unit(build_bits(fields))  # decoder.rs:642-692
    assert not coeffs.any() and seq == [2, 0]
    fields = [(1, 3), (0, 4), (1, 1), (0, 1), (0, 2), (0, 1), (1, 6), (0, 1), (0, 2),
              (100, 8), (1, 4), (1, 5), (0, 1), (0, 1), (0, 1), (0, 1), (0b10100, 5),
              (100, 8), (0, 4), (1, 5), (0, 1), (0, 1), (0, 1)]
    coeffs, _, _ = stereo.decode_access_unit(build_bits(fields))               # decoder.rs:694-736
    assert coeffs[0, 3] == 1.0 and np.count_nonzero(coeffs) == 1


# ---- the product against the oracle -------------------------------------------------------------------------------
@pytest.mark.parametrize("name", FIXTURES)
def test_product_front_end_equals_oracle_bit_for_bit(name):
    data = open(os.path.join(GOLD, name), "rb").read()
    frames = OF.split_adts(data)
    assert frames == aac_lc.split_adts(data)
    oracle, product = OF.Decoder(frames[0][0]), aac_lc.AacLcFrontEnd(frames[0][0])
    assert (oracle.sample_rate, oracle.channels) == (product.sample_rate, product.channels)
    for k, (_, au) in enumerate(frames):
        want, wseq, wshape = oracle.decode_access_unit(au)
        got, gseq, gshape = product.parse(au)
        assert (wseq, wshape) == (gseq, gshape), k
        assert np.array_equal(want.view(np.uint32), got.view(np.uint32)), (k, np.abs(want - got).max())


def test_product_and_oracle_fail_alike_on_mutated_access_units():
    """Same verdict on damaged input: both accept with identical spectra, or both reject with the same error kind
    and message (the messages are the reference's)."""
    state = 0x2545F4914F6CDD1D
    accepted = rejected = 0
    for name in ("aac-stereo-48k.adts", "mono16k_A_Tusk.aac", "stereo-music-44100-192k.aac"):
        frames = OF.split_adts(open(os.path.join(GOLD, name), "rb").read())
        for trial in range(400):
            state ^= (state << 13) & 0xFFFFFFFFFFFFFFFF
            state ^= state >> 7
            state ^= (state << 17) & 0xFFFFFFFFFFFFFFFF
            au = bytearray(frames[(state >> 8) % len(frames)][1])
            for k in range(1 + (state >> 20) % 3):
                r = (state >> (24 + 9 * k)) & 0xFFFFFF
                if r % 3 == 0:
                    au[(r >> 4) % len(au)] ^= 1 << (r & 7)
                elif r % 3 == 1:
                    au[(r >> 4) % len(au)] = (r >> 12) & 0xFF
                else:
                    del au[(r >> 4) % len(au):]
                    if not au:
                        au = bytearray(b"\\0")
            # fresh decoders: the PNS generator's state must be the same on both sides
            oracle, product = OF.Decoder(frames[0][0]), aac_lc.AacLcFrontEnd(frames[0][0])
            try:
                want = oracle.decode_access_unit(bytes(au))
            except OF.AacError as e:
                with pytest.raises(aac_lc.AacLcError) as exc:
                    product.parse(bytes(au))
                assert exc.value.kind == e.kind and str(e) in str(exc.value), (name, trial, str(e), str(exc.value))
                rejected += 1
                continue
            got = product.parse(bytes(au))
            assert want[1:] == got[1:] and np.array_equal(want[0].view(np.uint32), got[0].view(np.uint32)), (name, trial)
            accepted += 1
    assert accepted > 50 and rejected > 300
