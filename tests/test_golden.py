"""Committed golden vectors (tests/golden/*.npz, made by tests/golden/make_golden.py).

CPU: the oracle still reproduces them bit for bit (guards the checker against drift).
GPU: the HIP path matches them (bit-exact for integer/byte work, 1e-6 RMS for float)."""
import os

import numpy as np
import pytest

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load(name):
    return np.load(os.path.join(GOLD, name))


def rel_rms(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return np.sqrt(np.mean((a - b) ** 2)) / (np.sqrt(np.mean(b * b)) or 1.0)


def test_oracle_reproduces_aac_golden(oracle):
    g = load("aac_synth.npz")
    pcm, chans = oracle.synthesize_stream(g["coeffs"], g["seqs"], g["shapes"])
    assert np.array_equal(pcm, g["pcm"])
    assert np.array_equal(np.stack([c.delay for c in chans]), g["delay"])
    assert [c.prev_shape for c in chans] == g["prev_shape"].tolist()
    s16 = np.stack([oracle.planar_f32_to_s16_interleaved(pcm[f]).reshape(1024, 2) for f in range(pcm.shape[0])])
    assert np.array_equal(s16, g["s16"])


def test_oracle_reproduces_fir_golden(oracle):
    g = load("fir_48k_16k.npz")
    assert np.array_equal(oracle.downsample_planar(g["x"], 48000, 16000), g["y"])
    assert np.array_equal(oracle.resampler_taps(16000 / 48000), g["taps"])


def test_oracle_reproduces_pcm_golden(oracle):
    g = load("pcm_ops.npz")
    for name in oracle.OPS:
        want = g["out_" + name]
        got = oracle.pcm_convert(name, g["in_" + name], want.size)
        assert np.array_equal(got.view(np.uint8), want.view(np.uint8)), name


@pytest.mark.gpu
def test_gpu_matches_aac_golden(engine):
    from soundkit_amd import aac_lc
    g = load("aac_synth.npz")
    n = g["coeffs"].shape[0]
    a, b = engine.open_stream(48000, 2), engine.open_stream(48000, 2)
    pcm, st = aac_lc.synthesize_batch(engine, [a] * n, 2, g["coeffs"], g["seqs"], g["shapes"])
    s16, _ = aac_lc.synthesize_batch(engine, [b] * n, 2, g["coeffs"], g["seqs"], g["shapes"], out="s16")
    delay, shape = engine.get_state(a, 2)
    engine.close_stream(a), engine.close_stream(b)
    assert not st.any()
    for f in range(n):
        assert rel_rms(pcm[f], g["pcm"][f]) < 1e-6, f
    assert rel_rms(delay, g["delay"]) < 1e-6 and shape.tolist() == g["prev_shape"].tolist()
    assert np.abs(s16.astype(np.int32) - g["s16"].astype(np.int32)).max() <= 1


@pytest.mark.gpu
def test_gpu_matches_fir_golden(engine):
    g = load("fir_48k_16k.npz")
    assert np.array_equal(engine.taps(), g["taps"])
    y = engine.downsample_48k_16k(g["x"])
    assert y.shape == g["y"].shape and rel_rms(y, g["y"]) < 1e-6 and np.abs(y - g["y"]).max() < 2e-6


@pytest.mark.gpu
def test_gpu_matches_pcm_golden(engine):
    import soundkit_amd
    g = load("pcm_ops.npz")
    for name in soundkit_amd.engine.PCM_OPS:
        want = g["out_" + name]
        got = engine.pcm_convert(name, g["in_" + name], want.size)
        assert np.array_equal(got.view(np.uint8), want.view(np.uint8)), name
