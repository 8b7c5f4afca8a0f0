"""The reference's two MP3 fixtures through the product path -- Mp3Decoder() with the standard's tables
(csrc/mp3_iso_tables.h): framing + scale factors + Huffman on the host, requantisation / stereo / reorder and the hybrid
synthesis on the GPU (mp3_requant.hip, mp3_hybrid.hip) -- GPU side of the gates of tests/test_mp3_iso_tables.py:

  (ii)  every frame decodes (no granule rejected), 82 x 576 x 2 and 86 x 576 samples;
  (iv)  the PCM meets the source PCM the reference holds for the fixtures at the floors measured with the oracle, and the
        oracle's f64 decode within 1e-6 relative RMS (north_star's float tolerance); s16 bit-exact through f32_to_i16 on the
        decoder's own floats (soundkit-mp3/src/lib.rs:376-385);
  (v)   the reference's own assertions: 16 kHz, 1 channel on testdata/mp3 decoded in 4096-byte chunks into a
        2 x MAX_SAMPLES_PER_FRAME scratch with an empty-input drain (lib.rs:523-559); two chunks vs 1200-byte chunks
        bit-identical (lib.rs:678-761).
Sample parity with nanomp3 itself stays unpinned (its source is absent); these pins are reference-held data."""
import numpy as np
import pytest

from oracle import oracle as O
from soundkit_amd import mp3
from test_mp3_iso_tables import (ENCODER_DELAY, MONO, SNR_FLOOR_MONO_BELOW_2K, SNR_FLOOR_MONO_FULL_BAND, SNR_FLOOR_STEREO_BELOW_6K5,
                                 SNR_FLOOR_STEREO_FULL_BAND, STEREO, best_offset, oracle_decode, snr_against_source, source_mono,
                                 source_stereo)

pytestmark = pytest.mark.gpu
MAX = mp3.MAX_SAMPLES_PER_FRAME


def decode_chunks(dec, chunks, kind="i16", room=2 * MAX):
    """the reference tests' loop: one decode call per chunk, then empty input until nothing comes"""
    dtype = {"f32": np.float32, "i16": np.int16, "i32": np.int32}[kind]
    fn = getattr(dec, "decode_" + kind)
    out, scratch = [], np.zeros(room, dtype)
    for chunk in chunks:
        n = fn(chunk, scratch)
        out.append(scratch[:n].copy())
    while True:
        n = fn(b"", scratch)
        if n == 0:
            break
        out.append(scratch[:n].copy())
    return np.concatenate(out)


def pieces(data, size):
    return [data[i:i + size] for i in range(0, len(data), size)]


def test_streaming_decode_of_the_mono_fixture(engine):
    """soundkit-mp3/src/lib.rs:523-559"""
    data = open(MONO, "rb").read()
    dec = mp3.Mp3Decoder(engine=engine)
    try:
        decoded = decode_chunks(dec, pieces(data, 4096))
        assert decoded.size == 86 * 576
        assert dec.sample_rate() == 16000 and dec.channels() == 1
        assert dec.frames_decoded() == 86 and dec.buffer_len() == 0
    finally:
        dec.close()


@pytest.mark.parametrize("path", [MONO, STEREO])
def test_chunk_size_invariance(engine, path):
    """soundkit-mp3/src/lib.rs:678-761: two large chunks vs 1200-byte chunks, bit-identical i16"""
    data = open(path, "rb").read()
    outs = []
    for chunks in ([data[:len(data) // 2], data[len(data) // 2:]], pieces(data, 1200), pieces(data, 61), [data]):
        dec = mp3.Mp3Decoder(engine=engine)
        try:
            outs.append(decode_chunks(dec, chunks))
        finally:
            dec.close()
    assert outs[0].size == (86 * 576 if path == MONO else 82 * 576 * 2)
    for other in outs[1:]:
        assert np.array_equal(outs[0], other)


@pytest.mark.parametrize("path, channels, frames", [(STEREO, 2, 82), (MONO, 1, 86)])
def test_decoded_pcm_against_the_oracle_and_the_source(engine, path, channels, frames):
    data = open(path, "rb").read()
    dec = mp3.Mp3Decoder(engine=engine)
    try:
        got = decode_chunks(dec, [data], "f32", room=1 << 17).reshape(-1, channels)
        assert dec.sample_rate() == 16000 and dec.channels() == channels and dec.frames_decoded() == frames
        dec.reset()
        got16 = decode_chunks(dec, [data], "i16", room=1 << 17).reshape(-1, channels)
        dec.reset()
        got32 = decode_chunks(dec, [data], "i32", room=1 << 17).reshape(-1, channels)
    finally:
        dec.close()
    want = oracle_decode(path)
    assert got.shape == want.shape == (frames * 576, channels)
    err = np.sqrt(np.mean((got.astype(np.float64) - want) ** 2)) / np.sqrt(np.mean(want ** 2))
    assert err < 1e-6, err
    assert np.abs(got - want).max() < 4e-6 * np.abs(want).max()
    # the integer tails on the decoder's own floats: bit-exact (lib.rs:376-396)
    assert np.array_equal(got16, O.pcm_convert("MP3_F32_TO_I16", got.reshape(-1)).reshape(-1, channels))
    assert np.array_equal(got32, O.pcm_convert("MP3_F32_TO_I32", got.reshape(-1)).reshape(-1, channels))

    src = source_stereo() if channels == 2 else source_mono()
    for pcm in (got.astype(np.float64), got16.astype(np.float64) / 32767.0):
        assert best_offset(pcm, src) == ENCODER_DELAY
        full, gain = snr_against_source(pcm, src, ENCODER_DELAY)
        if channels == 2:
            low, _ = snr_against_source(pcm, src, ENCODER_DELAY, 6500)
            assert full > SNR_FLOOR_STEREO_FULL_BAND and low > SNR_FLOOR_STEREO_BELOW_6K5, (full, low)
        else:
            low, _ = snr_against_source(pcm, src, ENCODER_DELAY, 2000)
            assert full > SNR_FLOOR_MONO_FULL_BAND and low > SNR_FLOOR_MONO_BELOW_2K, (full, low)


def test_a_caller_supplied_codebook_of_the_same_tables_gives_the_same_bits(engine):
    """sk_mp3_iso_tables -> sk_mp3_codebook_create -> decoder == the default decoder"""
    data = open(STEREO, "rb").read()
    book = mp3.Codebook(mp3.iso_tables())
    a, b = mp3.Mp3Decoder(book, engine), mp3.Mp3Decoder(engine=engine)
    try:
        assert np.array_equal(decode_chunks(a, pieces(data, 4096)), decode_chunks(b, pieces(data, 4096)))
    finally:
        a.close(), b.close(), book.close()


def test_a_mono_to_stereo_splice_never_mixes_the_two_in_one_launch(engine):
    """Two streams back to back in one byte stream (channel count 1 -> 2): the decode call stops its launch in front of the first
    stereo frame (granules queued for the mono engine stream must not be joined by ones for the stereo stream), the next call
    goes on; nothing is lost, nothing comes out as silence, and every sample is what the two files give alone."""
    mono, stereo = open(MONO, "rb").read(), open(STEREO, "rb").read()
    alone = []
    for data in (mono, stereo):
        dec = mp3.Mp3Decoder(engine=engine)
        try:
            alone.append(decode_chunks(dec, [data], "i16", room=1 << 17))
        finally:
            dec.close()
    for chunks in ([mono + stereo], pieces(mono + stereo, 5000), [mono[:9000], mono[9000:] + stereo[:700], stereo[700:]]):
        dec = mp3.Mp3Decoder(engine=engine)
        try:
            got = decode_chunks(dec, chunks, "i16", room=1 << 17)
            assert dec.frames_decoded() == 86 + 82
            assert dec.sample_rate() == 16000 and dec.channels() == 1  # those of the first frame (soundkit-mp3/src/lib.rs:203-204)
        finally:
            dec.close()
        assert got.size == alone[0].size + alone[1].size
        assert np.array_equal(got[:alone[0].size], alone[0]) and np.array_equal(got[alone[0].size:], alone[1])
