"""sk_mp3_decoder_* (csrc/mp3_decoder.cpp -> mp3_requant.hip -> mp3_hybrid.hip) in the shape of soundkit-mp3's Mp3Decoder
(soundkit-mp3/src/lib.rs:147-374), on streams written by tests/mp3_builder.py with synthetic code books, against
oracle/mp3_bitstream.py's f64 Decoder: PCM within 1e-6 relative RMS (north_star's float tolerance), the s16 / s32 tails
bit-exact on the decoder's own floats (lib.rs:376-396), and the handle's contract (chunked input, the output-room rule,
the 4 MiB budget, reset, sample_rate / channels / buffer_len).  Parity of the MP3 row stays unpinned."""
import numpy as np
import pytest

import mp3_builder as B
from oracle import mp3_bitstream as ref
from soundkit_amd import mp3
from soundkit_amd._lib import SoundkitError

pytestmark = pytest.mark.gpu
TABLES = B.make_tables(5)
CTABLES, _KEEP = B.to_ctypes(TABLES)


@pytest.fixture(scope="module")
def codebook():
    cb = mp3.Codebook(CTABLES)
    yield cb
    cb.close()


def oracle_pcm(data, skip_reservoir_misses=True):
    frames, _ = ref.scan(data, [0])
    dec = ref.Decoder(TABLES)
    out = [dec.frame(data, off, h) for off, h in frames]
    return [o for o in out if o is not None], frames


def decode_all(dec, data, chunk, kind="f32", room=1 << 16):
    dtype = {"f32": np.float32, "i16": np.int16, "i32": np.int32}[kind]
    fn = getattr(dec, "decode_" + kind)
    out, scratch = [], np.zeros(room, dtype)
    for at in range(0, len(data), chunk):
        n = fn(data[at:at + chunk], scratch)
        out.append(scratch[:n].copy())
    while True:  # the reference's tests drain with empty input (lib.rs:541-547)
        n = fn(b"", scratch)
        if n == 0:
            break
        out.append(scratch[:n].copy())
    return np.concatenate(out) if out else np.zeros(0, dtype)


def rel_rms(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return np.sqrt(np.mean((a - b) ** 2)) / np.sqrt(np.mean(b ** 2))


STREAMS = [dict(version=1, rate=44100, channels=2, mode=1, joint_modes=(0, 2)), dict(version=1, rate=48000, channels=1),
           dict(version=1, rate=44100, channels=2, mode=1, joint_modes=(1, 3)), dict(version=1, rate=32000, channels=2, mode=0, crc=True),
           dict(version=2, rate=16000, channels=2, mode=1, joint_modes=(0, 2), bitrate_indices=(8, 10, 13)),
           dict(version=25, rate=11025, channels=1, bitrate_indices=(8, 11)),
           # 13818-3 intensity stereo (with and without mid/side, both intensity scales, "not intensity coded" positions)
           dict(version=2, rate=24000, channels=2, mode=1, joint_modes=(1, 3), bitrate_indices=(8, 10, 13)),
           dict(version=25, rate=8000, channels=2, mode=1, joint_modes=(1, 2, 3), bitrate_indices=(6, 8)),
           # free format (bit-rate index 0): the frame length is measured between headers; with and without padding slots, up to
           # the largest frame the scan looks for
           dict(version=1, rate=48000, channels=2, mode=1, joint_modes=(0, 1, 2, 3), free_format_bytes=1000),
           dict(version=2, rate=22050, channels=1, free_format_bytes=417), dict(version=1, rate=44100, channels=2, mode=0, crc=True, free_format_bytes=2200)]


@pytest.mark.parametrize("k", range(len(STREAMS)))
def test_streams_decode_to_the_f64_chain(engine, codebook, k):
    data, frames = B.build_stream(TABLES, 200 + k, n_frames=20, **STREAMS[k])
    want, found = oracle_pcm(data)
    assert len(want) == len(frames) == len(found)
    want = np.concatenate(want).reshape(-1)
    channels = STREAMS[k]["channels"]
    dec = mp3.Mp3Decoder(codebook, engine)
    try:
        assert dec.sample_rate() is None and dec.channels() is None and dec.buffer_len() == 0
        got = decode_all(dec, data, 4096)
        assert dec.sample_rate() == STREAMS[k]["rate"] and dec.channels() == channels and dec.buffer_len() == 0
        assert dec.frames_decoded() == len(frames)
        assert got.shape == want.shape
        assert rel_rms(got, want) < 1e-6
        assert np.abs(got - want).max() < 4e-6 * np.abs(want).max()
        assert 1e-4 < np.abs(want).max()
        # any chunking gives the same samples; so does a fresh decoder
        dec.reset()
        assert dec.sample_rate() is None and dec.buffer_len() == 0
        again = decode_all(dec, data, 97)
        assert np.array_equal(again, got)
        # the integer tails, on the decoder's own floats
        dec.reset()
        s16 = decode_all(dec, data, 1000, "i16")
        scaled = (got.astype(np.float32) * np.float32(32767.0)).astype(np.float64)
        scaled = np.sign(scaled) * np.floor(np.abs(scaled) + 0.5)   # f32::round: half away from zero
        assert np.array_equal(s16, np.clip(scaled, -32768, 32767).astype(np.int16))
        dec.reset()
        s32 = decode_all(dec, data, 5000, "i32")
        scaled = (got.astype(np.float32) * np.float32(2147483648.0)).astype(np.float64)
        scaled = np.sign(scaled) * np.floor(np.abs(scaled) + 0.5)
        assert np.array_equal(s32, np.clip(scaled, -2147483648, 2147483647).astype(np.int64).astype(np.int32))
    finally:
        dec.close()


def test_output_room_rule_tag_garbage_and_budget(engine, codebook):
    data, frames = B.build_stream(TABLES, 300, version=1, rate=44100, channels=2, mode=0, n_frames=9)
    want, _ = oracle_pcm(data)
    want = np.concatenate(want).reshape(-1)
    tag = b"ID3\x04\x00\x00\x00\x00\x00\x15" + bytes(21)
    dec = mp3.Mp3Decoder(codebook, engine)
    try:
        # room for one frame and a bit: every call hands back exactly one frame (lib.rs:300-302), the rest stays buffered
        room = np.zeros(2304 + 100, np.float32)
        n = dec.decode_f32(tag + b"\x00\xff\x12junk" + data, room)
        assert n == 2304 and dec.buffer_len() == len(data) - frames[0]["header"]["frame_bytes"]
        got = [room[:n].copy()]
        while True:
            n = dec.decode_f32(b"", room)
            if n == 0:
                break
            assert n == 2304
            got.append(room[:n].copy())
        got = np.concatenate(got)
        assert got.shape == want.shape and rel_rms(got, want) < 1e-6 and dec.buffer_len() == 0
        # a buffer that cannot take a frame: the reference's "Output buffer too small for decoded frame"
        dec.reset()
        with pytest.raises(SoundkitError) as exc:
            dec.decode_f32(data, np.zeros(1000, np.float32))
        assert exc.value.status == -7
        assert dec.buffer_len() == len(data)  # nothing was lost: a larger buffer gets it all
        assert dec.decode_f32(b"", np.zeros(1 << 16, np.float32)) == want.size
        # half a frame: nothing comes out until the rest arrives
        dec.reset()
        cut = frames[0]["header"]["frame_bytes"] // 2
        assert dec.decode_f32(data[:cut], room) == 0 and dec.buffer_len() == cut
        assert dec.decode_f32(data[cut:frames[0]["header"]["frame_bytes"] + 4], room) == 2304
        # MAX_MP3_STREAM_BUFFER_BYTES (lib.rs:155, 219-227)
        dec.reset()
        assert dec.decode_f32(bytes(4 * 1024 * 1024), room) == 0   # no frame in it: scanned and dropped
        assert dec.buffer_len() < 4
        with pytest.raises(SoundkitError) as exc:
            dec.decode_f32(bytes(4 * 1024 * 1024 + 1), room)
        assert exc.value.status == -203
    finally:
        dec.close()


def test_a_stream_joined_in_the_middle(engine, codebook):
    """frames whose main data begins in frames that were never seen are consumed without output; decoding starts with the
    first frame whose reservoir is complete -- the oracle's Decoder does the same, and from there on the samples agree"""
    data, frames = B.build_stream(TABLES, 400, version=2, rate=24000, channels=2, mode=1, joint_modes=(0, 2), n_frames=30, bitrate_indices=(8, 9, 11))
    first = next(i for i in range(3, len(frames)) if frames[i]["side"]["main_data_begin"] > 0)
    at = sum(f["header"]["frame_bytes"] for f in frames[:first])
    tail = data[at:]
    want, found = oracle_pcm(tail)
    assert 0 < len(want) < len(found), "the joined stream starts with frames that cannot be decoded"
    want = np.concatenate(want).reshape(-1)
    dec = mp3.Mp3Decoder(codebook, engine)
    try:
        got = decode_all(dec, tail, 333)
        assert got.shape == want.shape and rel_rms(got, want) < 1e-6
        assert dec.frames_decoded() == want.size // (576 * 2)
    finally:
        dec.close()


def test_two_decoders_share_an_engine(engine, codebook):
    a_data, _ = B.build_stream(TABLES, 500, version=1, rate=44100, channels=2, mode=1, n_frames=8)
    b_data, _ = B.build_stream(TABLES, 501, version=2, rate=22050, channels=1, n_frames=8, bitrate_indices=(8, 12))
    wa, _ = oracle_pcm(a_data)
    wb, _ = oracle_pcm(b_data)
    a, b = mp3.Mp3Decoder(codebook, engine), mp3.Mp3Decoder(codebook, engine)
    try:
        ga, gb = [], []
        room = np.zeros(1 << 16, np.float32)
        for at in range(0, max(len(a_data), len(b_data)), 700):  # interleaved calls: each handle owns its synthesis state
            n = a.decode_f32(a_data[at:at + 700], room)
            ga.append(room[:n].copy())
            n = b.decode_f32(b_data[at:at + 700], room)
            gb.append(room[:n].copy())
        assert rel_rms(np.concatenate(ga), np.concatenate(wa).reshape(-1)) < 1e-6
        assert rel_rms(np.concatenate(gb), np.concatenate(wb).reshape(-1)) < 1e-6
        assert (a.sample_rate(), a.channels(), b.sample_rate(), b.channels()) == (44100, 2, 22050, 1)
    finally:
        a.close(), b.close()
