"""The audio_packet::Decoder surface for ADTS AAC-LC (csrc/adts_decoder.cpp, soundkit_amd/aac.py): chunk-size
invariance through decode_i16_with_drain as the worker drives it (the reference tests this for its stream decoders:
soundkit-mp3 lib.rs:678-761, soundkit-decoder lib.rs:5339-5378), equality with the access-unit decoder, buffer limits
and the getters' None-until-first-frame behaviour (soundkit-aac lib.rs:133-139, 213-215)."""
import os

import numpy as np
import pytest

from soundkit_amd import aac, aac_lc

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "aac")
SCRATCH = 262144  # the worker's scratch: soundkit-decoder lib.rs:82


def whole_decode(engine, data):
    frames = aac_lc.split_adts(data)
    dec = aac_lc.AacLcDecoder(frames[0][0], engine)
    out = np.concatenate([dec.decode_access_unit_s16(au).ravel() for _, au in frames])
    dec.close()
    return out


@pytest.mark.parametrize("name", ["aac-stereo-48k.adts", "mono16k_A_Tusk.aac", "stereo-music-44100-192k.aac"])
@pytest.mark.parametrize("chunk", [1, 333, 4096, 65536, 10 ** 7])
def test_chunk_size_invariance(engine, name, chunk):
    data = open(os.path.join(GOLD, name), "rb").read()
    if chunk == 1:  # one byte per call: a dozen frames are enough
        data = data[:sum(len(au) + 7 for _, au in aac_lc.split_adts(data)[:12])]
    want = whole_decode(engine, data)
    dec = aac.AacDecoder.new(engine)
    dec.init()
    assert dec.sample_rate() is None and dec.channels() is None
    out = np.zeros(SCRATCH, np.int16)
    got = []
    for pos in range(0, len(data), chunk):
        got += aac.decode_i16_with_drain(dec, data[pos:pos + chunk], out)
    got = np.concatenate(got)
    assert np.array_equal(got, want)
    fe = aac_lc.AacLcFrontEnd(aac_lc.split_adts(data)[0][0])
    assert (dec.sample_rate(), dec.channels()) == (fe.sample_rate, fe.channels)
    dec.close()


def test_output_limits_and_errors(engine):
    data = open(os.path.join(GOLD, "aac-stereo-48k.adts"), "rb").read()
    dec = aac.AacDecoder(engine)
    # the output holds three frames: the rest stays buffered and comes out of the empty-input calls
    out = np.zeros(3 * 2048, np.int16)
    n = dec.decode_i16(data, out)
    assert n == 3 * 2048
    total = n
    while True:
        n = dec.decode_i16(b"", out)
        if n == 0:
            break
        total += n
    assert total == 48 * 2048
    with pytest.raises(ValueError) as exc:  # soundkit-aac lib.rs:202-207
        dec.decode_i16(data[:1000], np.zeros(100, np.int16))
    assert "Output buffer too small for decoded frame (needed 2048, had 100)" in str(exc.value)
    with pytest.raises(ValueError) as exc:
        dec.decode_i16(b"\0" * (4 * 1024 * 1024 + 1), out)
    assert "streaming budget" in str(exc.value)
    with pytest.raises(ValueError) as exc:
        dec.decode_i32(b"", np.zeros(4, np.int32))
    assert str(exc.value) == "Not implemented."
    # decode_f32 = decode_i16 / 32768
    d2, d3 = aac.AacDecoder(engine), aac.AacDecoder(engine)
    a, b = np.zeros(SCRATCH, np.int16), np.zeros(SCRATCH, np.float32)
    na, nb = d2.decode_i16(data, a), d3.decode_f32(data, b)
    assert na == nb == 48 * 2048 and np.array_equal(b[:nb], a[:na].astype(np.float32) / np.float32(32768.0))
    for d in (dec, d2, d3):
        d.close()


def test_crc_protected_adts_decodes_the_same(engine):
    from test_aac_frontend import with_crc
    data = open(os.path.join(GOLD, "aac-stereo-48k.adts"), "rb").read()
    out = np.zeros(SCRATCH, np.int16)
    a, b = aac.AacDecoder(engine), aac.AacDecoder(engine)
    na = a.decode_i16(data, out)
    want = out[:na].copy()
    nb = b.decode_i16(with_crc(data), out)
    assert na == nb == 48 * 2048 and np.array_equal(out[:nb], want)
    a.close(), b.close()


def test_frames_before_an_error_in_the_same_call_advance_the_state(engine):
    """soundkit-aac lib.rs:150-245: when a call returns Err, the frames decoded earlier in that call are lost but the decoder's
    overlap state has moved past them.  Here: frames 0-4 and a damaged frame 5 in ONE call (error), then frames 6-9; the PCM
    of 6-9 must equal what a decoder gives that was fed 0-4 in their own call, then the damaged frame alone (error), then 6-9."""
    data = open(os.path.join(GOLD, "aac-stereo-48k.adts"), "rb").read()
    frames = aac_lc.split_adts(data)
    sizes = [len(au) + 7 for _, au in frames]
    starts = np.concatenate([[0], np.cumsum(sizes)])
    k = 20                                  # frames 20-24, a damaged frame 25, frames 26-29: the clip is loud there
    bad = bytearray(data[starts[k + 5]:starts[k + 6]])
    bad[7] = (bad[7] & 0x1f) | (2 << 5)   # first element becomes a CCE: UnsupportedFeature, the frame is consumed
    head, tail = data[starts[k]:starts[k + 5]], data[starts[k + 6]:starts[k + 10]]
    out = np.zeros(SCRATCH, np.int16)
    a, b, c = aac.AacDecoder(engine), aac.AacDecoder(engine), aac.AacDecoder(engine)
    with pytest.raises(ValueError):
        a.decode_i16(head + bytes(bad), out)          # one call: five good frames, then the error
    na = a.decode_i16(tail, out)
    got = out[:na].copy()
    assert b.decode_i16(head, out) == 5 * 2048         # the same frames, the error in a call of its own
    with pytest.raises(ValueError):
        b.decode_i16(bytes(bad), out)
    nb = b.decode_i16(tail, out)
    assert na == nb == 4 * 2048 and np.array_equal(got, out[:nb])
    nc = c.decode_i16(tail, out)                       # a decoder that never saw frames 20-24: another first frame
    assert nc == na and not np.array_equal(got[:2048], out[:2048]) and np.array_equal(got[2048:], out[2048:nc])
    for d in (a, b, c):
        d.close()
