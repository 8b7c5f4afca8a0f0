"""BASELINE config 4's shape -- a mixed AAC-LC + MP3 batch through decode -> normalise -> resample to 16 kHz -- on one
engine: the reference's three AAC fixtures through AacDecoder, three MP3 streams through Mp3Decoder, their decode calls
interleaved chunk by chunk as a worker would drive them (soundkit-decoder/src/lib.rs:2150-2181), then
decoder_bytes_to_f32_planar (lib.rs:1793-1827's inverse on the way in) and downsample_audio
(soundkit/src/audio_pipeline.rs:438-493) on every stream.

The MP3 side runs on tests/mp3_builder.py's synthetic code books (the standard's Table B.7 is not in this tree), so this
test is about the plumbing of a mixed batch: every stream's result must be what the same stream gives alone on a fresh
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
    decoders = [aac.AacDecoder(engine) if kind == "aac" else mp3.Mp3Decoder(codebook, engine) for kind, _ in jobs]
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
