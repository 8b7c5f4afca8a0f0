"""BASELINE config 4's shape -- a mixed AAC-LC + MP3 batch through decode -> normalise -> resample to 16 kHz -- on one
engine: the reference's three AAC fixtures through AacDecoder, three MP3 streams through Mp3Decoder, their decode calls
interleaved chunk by chunk as a worker would drive them (soundkit-decoder/src/lib.rs:2150-2181), then
decoder_bytes_to_f32_planar (lib.rs:1793-1827's inverse on the way in) and downsample_audio
(soundkit/src/audio_pipeline.rs:438-493) on every stream.

test_config4_on_the_reference_fixtures is the config on its real inputs: golden/aac (three files) + golden/mp3 and
testdata/mp3 (the reference's two MP3 files) with the standard's tables, every stream against the whole CPU chain (oracle
decoder -> the decoder's 16-bit tail -> / 32768 -> oracle resampler).  test_mixed_aac_and_mp3_batch keeps the synthetic
MP3 streams (other rates, MPEG-1, CRC): every stream's result must be what the same stream gives alone on a fresh
engine, bit for bit, and the stages behind the decoders must match the oracle on the decoders' own samples."""
import os

import numpy as np
import pytest

import mp3_builder as B
import soundkit_amd
from soundkit_amd import aac, mp3

pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(os.path.dirname(__file__), "golden")
AAC_FILES = ["aac/stereo-music-44100-192k.aac", "aac/A_Tusk_is_used_to_make_costly_gifts_encoded.aac", "aac/mono16k_A_Tusk.aac"]
TABLES = B.make_tables(11)
CTABLES, _KEEP = B.to_ctypes(TABLES)
MP3_STREAMS = [dict(version=1, rate=44100, channels=2, mode=1, joint_modes=(0, 2)), dict(version=1, rate=48000, channels=1),
               dict(version=2, rate=22050, channels=2, mode=0, bitrate_indices=(9, 12))]


def run(engine, codebook, jobs, interleaved):
    """jobs: [(kind, bytes)] -> per job (rate, channels, s16 interleaved PCM, 16 kHz planar f32)"""
    decoders = [aac.AacDecoder(engine) if kind == "aac" else mp3.Mp3Decoder(codebook, engine) for kind, _ in jobs]  # codebook None: ISO
    pcm = [[] for _ in jobs]
    room = np.zeros(1 << 16, np.int16)
    try:
        chunk = 1500
        longest = max(len(data) for _, data in jobs)
        order = [(at, j) for at in range(0, longest, chunk) for j in range(len(jobs))] if interleaved else \
                [(at, j) for j in range(len(jobs)) for at in range(0, longest, chunk)]
        for at, j in order:
            piece = jobs[j][1][at:at + chunk]
            if piece:
                pcm[j] += aac.decode_i16_with_drain(decoders[j], piece, room)
        out = []
        for j, dec in enumerate(decoders):
            rate, channels = dec.sample_rate(), dec.channels()
            samples = np.concatenate(pcm[j])
            planar = engine.bytes_to_f32_planar(0, 0, np.frombuffer(samples.tobytes(), np.uint8), channels)  # decoder variant, s16le
            low = engine.downsample(planar, rate, 16000) if rate != 16000 else planar
            out.append((rate, channels, samples, low))
        return out
    finally:
        for dec in decoders:
            dec.close()


def test_mixed_aac_and_mp3_batch(engine, oracle):
    jobs = []
    for name in AAC_FILES:
        with open(os.path.join(GOLDEN, name), "rb") as f:
            jobs.append(("aac", f.read()))
    for k, params in enumerate(MP3_STREAMS):
        jobs.append(("mp3", B.build_stream(TABLES, 700 + k, n_frames=16, **params)[0]))
    jobs = [jobs[i] for i in (0, 3, 1, 4, 2, 5)]  # the two codecs alternate in the batch
    codebook = mp3.Codebook(CTABLES)
    try:
        mixed = run(engine, codebook, jobs, interleaved=True)
        alone = soundkit_amd.Engine(0, 16)
        try:
            for j, job in enumerate(jobs):
                (rate, channels, samples, low), = run(alone, codebook, [job], interleaved=False)
                assert (rate, channels) == mixed[j][:2]
                assert np.array_equal(samples, mixed[j][2]), "a stream's samples do not depend on its neighbours in the batch"
                assert np.array_equal(low, mixed[j][3])
        finally:
            alone.close()
        assert [m[:2] for m in mixed] == [(44100, 2), (44100, 2), (16000, 2), (48000, 1), (16000, 1), (22050, 2)]
        for rate, channels, samples, low in mixed:
            assert samples.size > 9000 and np.abs(samples.astype(np.int32)).max() > 50
            planar = oracle.decoder_bytes_to_f32_planar(oracle.FMT_S16LE, np.frombuffer(samples.tobytes(), np.uint8), channels)
            if rate == 16000:
                assert np.array_equal(low, planar)
                continue
            want = oracle.downsample_planar(planar, rate, 16000)
            assert low.shape == want.shape
            err = np.sqrt(np.mean((low.astype(np.float64) - want) ** 2)) / np.sqrt(np.mean(want.astype(np.float64) ** 2))
            assert err < 1e-6, (rate, err)
    finally:
        codebook.close()


MP3_FILES = ["mp3/stereo16k_A_Tusk_encoded.mp3", "mp3/mono16k_A_Tusk.mp3"]


def oracle_chain(oracle, kind, data):
    """the CPU chain of one file: (rate, channels, s16 interleaved) as the reference's decoders hand it to the worker"""
    if kind == "aac":
        from oracle import aac_frontend as OF
        frames = OF.split_adts(data)
        dec = OF.Decoder(frames[0][0])
        chans = [oracle.Channel() for _ in range(dec.channels)]
        out = []
        for _, au in frames:
            coeffs, seqs, shapes = dec.decode_access_unit(au)
            pcm, _ = oracle.synthesize_stream(coeffs[None], [seqs], [shapes], chans)
            out.append(oracle.planar_f32_to_s16_interleaved(pcm[0]))  # decode_aac_access_unit, lib.rs:1793-1827
        return dec.sample_rate, dec.channels, np.concatenate(out)
    from oracle import mp3_bitstream, mp3_iso
    frames, _ = mp3_bitstream.scan(data)
    dec = mp3_bitstream.Decoder(mp3_iso.tables())
    pcm = np.concatenate([dec.frame(data, off, h) for off, h in frames])
    h = frames[0][1]
    return h["sample_rate"], h["channels"], oracle.pcm_convert("MP3_F32_TO_I16", pcm.astype(np.float32).reshape(-1))  # lib.rs:376-385


def test_config4_on_the_reference_fixtures(engine, oracle):
    jobs = []
    for name in AAC_FILES + MP3_FILES:
        with open(os.path.join(GOLDEN, name), "rb") as f:
            jobs.append(("aac" if name.startswith("aac") else "mp3", f.read()))
    jobs = [jobs[i] for i in (0, 3, 1, 4, 2)]  # the two codecs alternate in the batch
    mixed = run(engine, None, jobs, interleaved=True)
    assert [m[:2] for m in mixed] == [(44100, 2), (16000, 2), (16000, 2), (16000, 1), (16000, 1)]
    alone = soundkit_amd.Engine(0, 16)
    try:
        for j, job in enumerate(jobs):
            (rate, channels, samples, low), = run(alone, None, [job], interleaved=False)
            assert np.array_equal(samples, mixed[j][2]) and np.array_equal(low, mixed[j][3])
    finally:
        alone.close()
    for (kind, data), (rate, channels, samples, low) in zip(jobs, mixed):
        want_rate, want_channels, want = oracle_chain(oracle, kind, data)
        assert (rate, channels) == (want_rate, want_channels)
        # the decoders: the GPU's f32 synthesis against the CPU's, after the exact 16-bit tail -> at most one LSB, rarely
        assert samples.shape == want.shape
        d = np.abs(samples.astype(np.int32) - want.astype(np.int32))
        assert d.max() <= 1 and (d > 0).mean() < 0.01, (kind, rate, int(d.max()), float((d > 0).mean()))
        assert np.abs(want.astype(np.int32)).max() > 500
        # normalise + resample: against the CPU chain continued from the CPU decoder's samples
        planar = oracle.decoder_bytes_to_f32_planar(oracle.FMT_S16LE, np.frombuffer(want.tobytes(), np.uint8), channels)
        chain = oracle.downsample_planar(planar, rate, 16000) if rate != 16000 else planar
        assert low.shape == chain.shape
        err = np.sqrt(np.mean((low.astype(np.float64) - chain) ** 2)) / np.sqrt(np.mean(chain.astype(np.float64) ** 2))
        assert err < (1e-3 if rate != 16000 else 2e-3), (kind, rate, err)  # one-LSB differences of the s16 stage in front
        # ... and exactly / within the float tolerance from the product's own samples
        planar = oracle.decoder_bytes_to_f32_planar(oracle.FMT_S16LE, np.frombuffer(samples.tobytes(), np.uint8), channels)
        if rate == 16000:
            assert np.array_equal(low, planar)
        else:
            own = oracle.downsample_planar(planar, rate, 16000)
            assert np.sqrt(np.mean((low.astype(np.float64) - own) ** 2)) / np.sqrt(np.mean(own.astype(np.float64) ** 2)) < 1e-6
