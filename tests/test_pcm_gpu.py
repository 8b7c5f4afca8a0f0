"""HIP sample-width / interleave kernels vs the oracle: bit-exact (integer, byte and exact-f32 work)."""
import struct

import numpy as np
import pytest

import soundkit_amd
from soundkit_amd import audio_bytes, audio_pipeline, decoder
from soundkit_amd.audio_types import AudioData, EncodingFlag, Endianness

pytestmark = pytest.mark.gpu

EDGE_F32 = np.array([0.0, -0.0, 1.0, -1.0, 0.5, -0.5, 0.99999994, -0.99999994, 1.0000001, -1.0000001, 2.0, -2.0,
                     np.nan, np.inf, -np.inf, 1e-8, -1e-8, 3.0517578e-05, -3.0517578e-05, 0.25, -0.25,
                     1.5259022e-05, 4.5777066e-05, 0.999969482421875, 1e30, -1e30], np.float32)


def raw_input_for(op_name, n, rng):
    ib = soundkit_amd._lib.lib.sk_pcm_op_in_bytes(soundkit_amd.engine.PCM_OP[op_name])
    if op_name.startswith(("F32", "VEC_F32", "FLOAT_", "MP3_")):
        x = rng.uniform(-1.3, 1.3, n).astype(np.float32)
        x[:EDGE_F32.size] = EDGE_F32[:min(n, EDGE_F32.size)]
        ties = x[EDGE_F32.size:EDGE_F32.size + 64]
        ties[:] = ((rng.integers(-32768, 32768, ties.size) + 0.5).astype(np.float32) / 32767.0)  # rounding ties
        if "BE" in op_name:
            x = x.byteswap()
        return x.view(np.uint8)
    raw = rng.integers(0, 256, n * ib, dtype=np.uint8)
    # extremes for integer formats
    raw[:ib * 4] = np.frombuffer(b"\x00" * ib + b"\xff" * ib + b"\x00" * (ib - 1) + b"\x80" + b"\xff" * (ib - 1) + b"\x7f",
                                 np.uint8)
    return raw


@pytest.mark.parametrize("op", soundkit_amd.engine.PCM_OPS)
@pytest.mark.parametrize("n", [0, 1, 3, 4, 5, 1023, 4099])
def test_elementwise_op_bit_exact(engine, oracle, op, n):
    rng = np.random.default_rng(hash(op) % 2**32 + n)
    raw = raw_input_for(op, max(n, 128), rng)
    ib = soundkit_amd._lib.lib.sk_pcm_op_in_bytes(soundkit_amd.engine.PCM_OP[op])
    raw = raw[:n * ib]
    got = engine.pcm_convert(op, raw, n)
    want = oracle.pcm_convert(op, raw, n)
    assert got.dtype == want.dtype
    assert np.array_equal(got.view(np.uint8), want.view(np.uint8)), op


@pytest.mark.parametrize("op", ["F32LE_TO_I16", "S24LE_TO_I32", "STEREO_TO_MONO_AVG", "FLOAT_TO_I16_ROUND", "S16BE_TO_I16"])
def test_elementwise_op_large_and_unaligned(engine, oracle, op):
    import torch
    n = 1 << 20
    rng = np.random.default_rng(9)
    ib = soundkit_amd._lib.lib.sk_pcm_op_in_bytes(soundkit_amd.engine.PCM_OP[op])
    ob = soundkit_amd._lib.lib.sk_pcm_op_out_bytes(soundkit_amd.engine.PCM_OP[op])
    raw = raw_input_for(op, n + 8, rng)
    want = oracle.pcm_convert(op, raw[:n * ib], n)
    # device path, 16-byte aligned
    d_in = torch.from_numpy(raw[:n * ib].copy()).cuda()
    d_out = torch.zeros(n * ob, dtype=torch.uint8, device="cuda")
    torch.cuda.synchronize()  # inputs are produced on torch's stream, the engine runs on its own
    engine.pcm_convert_dev(op, d_in, d_out, n)
    engine.synchronize()
    assert np.array_equal(d_out.cpu().numpy(), want.view(np.uint8))
    # device path, misaligned by one element on both sides (scalar kernel)
    d_in2 = torch.zeros(n * ib + 64, dtype=torch.uint8, device="cuda")
    d_in2[ib:ib + n * ib] = d_in
    d_out2 = torch.zeros(n * ob + 64, dtype=torch.uint8, device="cuda")
    torch.cuda.synchronize()
    engine.pcm_convert_dev(op, d_in2.data_ptr() + ib, d_out2.data_ptr() + ob, n)
    engine.synchronize()
    assert np.array_equal(d_out2[ob:ob + n * ob].cpu().numpy(), want.view(np.uint8))
    assert d_out2[:ob].sum().item() == 0 and d_out2[ob + n * ob:].sum().item() == 0  # no overrun


# ---- the reference's own vectors (audio_bytes.rs:380-468) through the mirror API ----------------

def test_reference_audio_bytes_vectors(engine):
    assert audio_bytes.deinterleave_vecs_i16(bytes([1, 0, 2, 0, 3, 0, 4, 0, 5, 0, 6, 0]), 2).tolist() == [[1, 3, 5], [2, 4, 6]]
    assert audio_bytes.interleave_vecs_i16([[1, 3, 5], [2, 4, 6]]).tolist() == [1, 0, 2, 0, 3, 0, 4, 0, 5, 0, 6, 0]
    s24 = bytes([1, 0, 0, 2, 0, 0, 3, 0, 0, 4, 0, 0, 5, 0, 0, 6, 0, 0])
    assert audio_bytes.deinterleave_vecs_s24(s24, 2).tolist() == [[1, 3, 5], [2, 4, 6]]
    f32 = bytes([0, 0, 128, 63, 0, 0, 0, 64, 0, 0, 64, 64, 0, 0, 128, 64, 0, 0, 160, 64, 0, 0, 192, 64])
    assert audio_bytes.deinterleave_vecs_f32(f32, 2).tolist() == [[1.0, 3.0, 5.0], [2.0, 4.0, 6.0]]
    got = audio_bytes.i16le_to_f32(bytes([0, 0, 0, 64, 255, 127, 0, 192, 0, 128]))
    assert np.abs(got - np.array([0.0, 0.5, 0.9999694, -0.5, -1.0])).max() < 1e-4
    assert audio_bytes.stereo_to_mono_take_left(np.array([10, 20, -30, -40, 50, 60], np.int16)).tolist() == [10, -30, 50]
    assert audio_bytes.stereo_to_mono_avg(np.array([100, -100, 50, 150, -200, 200], np.int16)).tolist() == [0, 100, 0]
    with pytest.raises(AssertionError):
        audio_bytes.i16le_to_f32(bytes([1, 2, 3]))
    with pytest.raises(AssertionError):
        audio_bytes.stereo_to_mono_avg(np.array([1, 2, 3], np.int16))


@pytest.mark.parametrize("kind,ch,frames", [("i16", 1, 7), ("i16", 2, 1001), ("i16", 6, 333), ("s24", 2, 515),
                                            ("s24", 3, 64), ("f32", 2, 4097), ("f32", 5, 10), ("i16", 2, 0)])
def test_layout_ops_bit_exact(engine, oracle, kind, ch, frames):
    rng = np.random.default_rng(frames * 8 + ch)
    bps = {"i16": 2, "s24": 3, "f32": 4}[kind]
    raw = rng.integers(0, 256, frames * ch * bps, dtype=np.uint8)
    got = engine.deinterleave(kind, raw, ch)
    want = oracle.deinterleave(kind, raw, ch)
    assert np.array_equal(got.view(np.uint8), want.view(np.uint8))
    if kind == "i16":
        assert np.array_equal(engine.interleave_i16(want), oracle.interleave_i16(want))
    if kind == "f32":
        assert np.array_equal(engine.interleave_f32(want), oracle.interleave_f32(want))


@pytest.mark.parametrize("fmt", range(8))
@pytest.mark.parametrize("ch,frames", [(1, 5), (2, 2048), (2, 777), (6, 100)])
def test_bytes_to_f32_planar_decoder_variant(engine, oracle, fmt, ch, frames):
    rng = np.random.default_rng(fmt * 100 + frames)
    bps = oracle.fmt_bytes(fmt)
    raw = rng.integers(0, 256, frames * ch * bps, dtype=np.uint8)
    got = engine.bytes_to_f32_planar(0, fmt, raw, ch)
    want = oracle.decoder_bytes_to_f32_planar(fmt, raw, ch)
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32))


@pytest.mark.parametrize("fmt", [0, 2, 4, 6])
def test_bytes_to_f32_planar_core_variant(engine, oracle, fmt):
    rng = np.random.default_rng(fmt)
    raw = rng.integers(0, 256, 999 * 2 * oracle.fmt_bytes(fmt), dtype=np.uint8)
    if fmt == 6:  # keep NaN payloads out of the bitwise compare: pass-through is exact for any bits anyway
        raw = rng.uniform(-2, 2, 999 * 2).astype(np.float32).view(np.uint8)
    got = engine.bytes_to_f32_planar(1, fmt, raw, 2)
    want = oracle.core_bytes_to_f32_planar(fmt, raw, 2)
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32))


@pytest.mark.parametrize("fmt", [0, 2, 4, 6])
@pytest.mark.parametrize("ch,frames", [(1, 9), (2, 4096), (2, 1001), (3, 50)])
def test_f32_planar_to_bytes(engine, oracle, fmt, ch, frames):
    rng = np.random.default_rng(fmt * 7 + frames)
    x = rng.uniform(-1.2, 1.2, (ch, frames)).astype(np.float32)
    flat = x.ravel()
    flat[:min(flat.size, EDGE_F32.size)] = EDGE_F32[:min(flat.size, EDGE_F32.size)]
    if fmt == 6:
        flat[np.isnan(flat)] = 0
    got = engine.f32_planar_to_bytes(fmt, x)
    want = oracle.f32_planar_to_bytes(fmt, x)
    assert np.array_equal(got, want)


def test_downmix_and_exact(engine, oracle):
    rng = np.random.default_rng(2)
    x = rng.uniform(-1, 1, (2, 3001)).astype(np.float32)
    assert np.array_equal(engine.downmix_mono(x), oracle.downmix_mono(x))
    x6 = rng.uniform(-1, 1, (6, 100)).astype(np.float32)
    assert np.array_equal(engine.downmix_mono(x6), oracle.downmix_mono(x6))
    for fmt in (2, 3, 4, 5):
        raw = rng.integers(0, 256, 1000 * oracle.fmt_bytes(fmt), dtype=np.uint8)
        assert np.array_equal(engine.exact_to_i16(fmt, raw), oracle.exact_signed_pcm_to_i16(fmt, raw))


def read_wav(path):
    """Walk the RIFF chunks as WavStreamProcessor::add does (soundkit/src/wav.rs:95-262)."""
    data = open(path, "rb").read()
    assert data[:4] == b"RIFF" and data[8:12] == b"WAVE"
    pos, fmt, pcm = 12, None, None
    while pos + 8 <= len(data):
        cid, size = data[pos:pos + 4], struct.unpack("<I", data[pos + 4:pos + 8])[0]
        body = data[pos + 8:pos + 8 + size]
        if cid == b"fmt ":
            fmt = struct.unpack("<HHIIHH", body[:16])
        elif cid == b"data":
            pcm = body
        pos += 8 + size + (size & 1)
    return fmt, pcm


def test_config1_wav_stereo_plumbing(engine, oracle):
    """BASELINE configs[0]: s16le <-> f32 + deinterleave on testdata/wav_stereo (fixture copy)."""
    import os
    fmt, pcm = read_wav(os.path.join(os.path.dirname(__file__), "golden", "wav_stereo_A_Tusk.wav"))
    tag, ch, rate, _, align, bits = fmt
    assert (tag, ch, rate, align, bits) == (1, 2, 16000, 4, 16) and len(pcm) == 189440
    i16 = audio_bytes.s16le_to_i16(pcm)
    assert np.array_equal(i16, np.frombuffer(pcm, "<i2"))
    planes = audio_bytes.deinterleave_vecs_i16(pcm, 2)
    assert planes.shape == (2, 47360) and np.array_equal(planes[0], i16[0::2]) and np.array_equal(planes[1], i16[1::2])
    f = np.stack([audio_pipeline.vec_i16_to_f32(planes[c]) for c in range(2)])
    assert np.array_equal(f, planes.astype(np.float32) / np.float32(32768.0))
    back = np.stack([audio_pipeline.vec_f32_to_i16(f[c]) for c in range(2)])
    want_back = np.stack([oracle.pcm_convert("VEC_F32_TO_I16", f[c]) for c in range(2)])
    assert np.array_equal(back, want_back)
    # x/32768*32767 truncated: never larger in magnitude than the source, off by at most 1
    assert np.all(np.abs(back.astype(np.int32)) <= np.abs(planes.astype(np.int32)))
    assert np.abs(back.astype(np.int32) - planes.astype(np.int32)).max() <= 1
    inter = audio_bytes.interleave_vecs_i16(back)
    assert np.array_equal(inter, oracle.interleave_i16(back))
    # the decoder-side path: AudioData -> f32 channels -> s16 bytes is the identity for s16 input
    audio = AudioData(16, 2, 16000, np.frombuffer(pcm, np.uint8), EncodingFlag.PCMSigned, Endianness.LittleEndian)
    chans = decoder.audio_data_to_f32_channels(audio)
    assert np.array_equal(chans, f)
    out = decoder.f32_channels_to_bytes(chans, 16, EncodingFlag.PCMSigned)
    assert np.array_equal(out, oracle.f32_planar_to_bytes(oracle.FMT_S16LE, chans))


def test_apply_output_options_routes(engine, oracle):
    rng = np.random.default_rng(4)
    raw = rng.integers(0, 256, 480 * 2 * 3, dtype=np.uint8)
    audio = AudioData(24, 2, 48000, raw)
    out, _ = decoder.apply_output_options(audio)  # fast path: untouched
    assert out[0] is audio
    out, _ = decoder.apply_output_options(audio, output_bits_per_sample=16)  # exact integer narrowing
    assert np.array_equal(out[0].data, oracle.exact_signed_pcm_to_i16(oracle.FMT_S24LE, raw))
    out, _ = decoder.apply_output_options(audio, output_bits_per_sample=16, output_channels=1)
    chans = oracle.decoder_bytes_to_f32_planar(oracle.FMT_S24LE, raw, 2)
    want = oracle.f32_planar_to_bytes(oracle.FMT_S16LE, oracle.downmix_mono(chans)[None])
    assert out[0].channel_count == 1 and np.array_equal(out[0].data, want)
