"""HIP AAC-LC synthesis vs the oracle, through the C ABI.  Tolerance: north_star's 1e-6 RMS for
float IMDCT (relative to the signal's own RMS), plus a max-abs bound; integer PCM bit-exact."""
import numpy as np
import pytest

import soundkit_amd
from soundkit_amd import aac_lc

pytestmark = pytest.mark.gpu

RMS_TOL = 1.0e-6


def spectra(oracle, n_frames, ch, seed0, amp=1.0):
    out = np.empty((n_frames, ch, 1024), np.float32)
    for f in range(n_frames):
        for c in range(ch):
            out[f, c] = oracle.seeded_spectrum(1024, (seed0 + 0x9E3779B9 * (2 * f + c)) & 0xFFFFFFFF) * np.float32(amp)
    return out


def rel_rms(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    den = np.sqrt(np.mean(b * b)) or 1.0
    return np.sqrt(np.mean((a - b) ** 2)) / den


def run_stream(engine, oracle, coeffs, seqs, shapes, ch):
    n = coeffs.shape[0]
    sid = engine.open_stream(48000, ch)
    pcm, status = aac_lc.synthesize_batch(engine, [sid] * n, ch, coeffs, seqs, shapes)
    assert np.all(status == 0)
    want, chans = oracle.synthesize_stream(coeffs, seqs, shapes)
    delay, shape = engine.get_state(sid, ch)
    engine.close_stream(sid)
    return pcm, want, delay, shape, chans


def test_only_long_stream_matches_oracle(engine, oracle):
    n, ch = 12, 2
    coeffs = spectra(oracle, n, ch, 0x12345678, 2000.0)
    seqs = np.zeros((n, 2), np.uint8)
    shapes = np.array([[f & 1, (f >> 1) & 1] for f in range(n)], np.uint8)  # alternate Sine / KBD
    pcm, want, delay, shape, chans = run_stream(engine, oracle, coeffs, seqs, shapes, ch)
    assert rel_rms(pcm, want) < RMS_TOL
    assert np.abs(pcm - want).max() < 2e-6 * np.abs(want).max()
    for c in range(ch):
        assert rel_rms(delay[c], chans[c].delay) < RMS_TOL
        assert shape[c] == chans[c].prev_shape


def test_mixed_window_sequences_match_oracle(engine, oracle):
    # long -> start -> short -> short -> stop -> long, both shapes, per-channel independent
    seq_l = [0, 1, 2, 2, 3, 0, 0, 1, 2, 3, 0, 0]
    seq_r = [0, 0, 1, 2, 3, 0, 1, 2, 2, 2, 3, 0]
    n, ch = len(seq_l), 2
    coeffs = spectra(oracle, n, ch, 0xA5A50101, 3000.0)
    seqs = np.array(list(zip(seq_l, seq_r)), np.uint8)
    shapes = np.array([[(f // 2) & 1, (f // 3) & 1] for f in range(n)], np.uint8)
    pcm, want, delay, shape, chans = run_stream(engine, oracle, coeffs, seqs, shapes, ch)
    for f in range(n):
        for c in range(ch):
            assert rel_rms(pcm[f, c], want[f, c]) < RMS_TOL, (f, c, seqs[f, c])
    assert np.abs(pcm - want).max() < 2e-6 * np.abs(want).max()
    for c in range(ch):
        assert rel_rms(delay[c], chans[c].delay) < RMS_TOL


@pytest.mark.parametrize("seq", [0, 1, 2, 3])
@pytest.mark.parametrize("prev,cur", [(0, 0), (0, 1), (1, 0), (1, 1)])
def test_every_sequence_and_shape_pair(engine, oracle, seq, prev, cur):
    rng = np.random.default_rng(seq * 16 + prev * 2 + cur)
    coeffs = spectra(oracle, 1, 1, 0xDEADBEEF + seq, 1500.0)
    delay0 = rng.uniform(-0.5, 0.5, 1024).astype(np.float32)
    sid = engine.open_stream(44100, 1)
    engine.set_state(sid, delay0[None], [prev])
    pcm, status = aac_lc.synthesize_batch(engine, [sid], 1, coeffs, [[seq, 0]], [[cur, 0]])
    delay, shape = engine.get_state(sid, 1)
    engine.close_stream(sid)
    ch = oracle.Channel()
    ch.set_state(delay0, prev)
    want = ch.synthesize(coeffs[0, 0], seq, cur)
    assert status[0] == 0 and shape[0] == cur
    assert np.abs(pcm[0, 0] - want).max() < 3e-7 * max(1.0, np.abs(want).max())
    assert np.abs(delay[0] - ch.delay).max() < 3e-7 * max(1.0, np.abs(ch.delay).max())


def test_reference_fixed_pattern_and_seeds(engine, oracle):
    """dsp.rs:653-723 inputs through the whole GPU synthesis: with zero delay and a Sine window
    the output is imdct[:1024] * window -- compare with the direct-form IMDCT (tol 4e-8 abs there)."""
    pat = [0.0, 1.0, -2.0, 0.5, -0.25, 4.0, -8.0, 0.125, -0.75]
    inputs = [np.array([pat[i % 9] for i in range(1024)], np.float32)]
    inputs += [oracle.seeded_spectrum(1024, s) for s in (0x12345678, 0xA5A50101, 0xDEADBEEF)]
    win = oracle.sine_window(2048)
    for x in inputs:
        sid = engine.open_stream(48000, 1)
        pcm, _ = aac_lc.synthesize_batch(engine, [sid], 1, x[None, None], [[0, 0]], [[0, 0]])
        delay, _ = engine.get_state(sid, 1)
        engine.close_stream(sid)
        direct = oracle.imdct_direct_f64(x)
        assert np.abs(pcm[0, 0] - direct[:1024] * win[:1024]).max() < 4.0e-8
        assert np.abs(delay[0] - direct[1024:] * win[1024:]).max() < 4.0e-8


def test_batch_of_streams_interleaved_order_and_mono_mix(engine, oracle):
    """Frames of many streams interleaved in one call; mono and stereo mixed; array order kept per stream."""
    rng = np.random.default_rng(11)
    n_streams, n_frames = 37, 5
    chans = [1 + (s % 3 != 0) for s in range(n_streams)]
    sids = [engine.open_stream(48000, chans[s]) for s in range(n_streams)]
    frames = []  # (stream index, frame index)
    for f in range(n_frames):
        order = rng.permutation(n_streams)
        frames += [(int(s), f) for s in order]
    per_stream = {s: spectra(oracle, n_frames, chans[s], 1000 + s, 800.0) for s in range(n_streams)}
    seq_of = {s: rng.integers(0, 2, (n_frames, 2)) * 0 for s in range(n_streams)}
    # legal sequence chains per channel
    chain = [0, 1, 2, 3, 0]
    for s in range(n_streams):
        seq_of[s] = np.array([[chain[(f + s) % 5] if s % 2 else 0, chain[(f + 2 * s) % 5] if s % 4 == 1 else 0]
                              for f in range(n_frames)], np.uint8)
    shape_of = {s: rng.integers(0, 2, (n_frames, 2)).astype(np.uint8) for s in range(n_streams)}
    descs = [(sids[s], chans[s], seq_of[s][f], shape_of[s][f]) for s, f in frames]
    arr, n = soundkit_amd.make_descs(descs)
    coeffs = np.concatenate([per_stream[s][f].ravel() for s, f in frames])
    pcm, status = engine.aac_synthesize(arr, n, coeffs)
    assert np.all(status == 0)
    want = {s: oracle.synthesize_stream(per_stream[s], seq_of[s], shape_of[s])[0] for s in range(n_streams)}
    off = 0
    for s, f in frames:
        size = chans[s] * 1024
        got = pcm[off:off + size].reshape(chans[s], 1024)
        assert rel_rms(got, want[s][f]) < RMS_TOL, (s, f)
        off += size
    for sid in sids:
        engine.close_stream(sid)


def test_state_carries_across_calls(engine, oracle):
    n, ch = 6, 2
    coeffs = spectra(oracle, n, ch, 0x0BADF00D, 500.0)
    seqs = np.zeros((n, 2), np.uint8)
    shapes = np.ones((n, 2), np.uint8)
    sid = engine.open_stream(48000, ch)
    parts = [aac_lc.synthesize_batch(engine, [sid] * 2, ch, coeffs[i:i + 2], seqs[i:i + 2], shapes[i:i + 2])[0]
             for i in range(0, n, 2)]
    engine.close_stream(sid)
    want, _ = oracle.synthesize_stream(coeffs, seqs, shapes)
    assert rel_rms(np.concatenate(parts), want) < RMS_TOL


def test_bad_frames_fail_alone(engine, oracle):
    sid = engine.open_stream(48000, 2)
    coeffs = spectra(oracle, 4, 2, 42, 100.0)
    descs, n = soundkit_amd.make_descs([
        (sid, 2, (0, 0), (0, 0)),
        (0xFFFF0000, 2, (0, 0), (0, 0)),   # not open
        (sid, 1, (0, 0), (0, 0)),          # wrong channel count (packs 1 x 1024)
        (sid, 2, (4, 0), (0, 0)),          # bad window sequence
        (sid, 2, (0, 0), (1, 1)),
    ])
    packed = np.concatenate([coeffs[0].ravel(), coeffs[1].ravel(), coeffs[2, 0], coeffs[3].ravel(), coeffs[3].ravel()])
    sentinel = np.full(packed.size, 7.0, np.float32)
    pcm, status = engine.aac_synthesize(descs, n, packed, pcm=sentinel.copy())
    assert status.tolist() == [0, 1, 2, 3, 0]
    want, _ = oracle.synthesize_stream(np.stack([coeffs[0], coeffs[3]]), [[0, 0], [0, 0]], [[0, 0], [1, 1]])
    assert rel_rms(pcm[:2048].reshape(2, 1024), want[0]) < RMS_TOL
    assert np.all(pcm[2048:2048 + 2048 + 1024 + 2048] == 7.0)  # failed frames: output untouched
    assert rel_rms(pcm[-2048:].reshape(2, 1024), want[1]) < RMS_TOL
    engine.close_stream(sid)


def test_s16_output_bit_exact(engine, oracle):
    n, ch = 5, 2
    coeffs = spectra(oracle, n, ch, 77, 1.2e6)  # loud enough to exercise clamping
    seqs = np.zeros((n, 2), np.uint8)
    shapes = np.zeros((n, 2), np.uint8)
    a, b = engine.open_stream(48000, ch), engine.open_stream(48000, ch)
    f32, _ = aac_lc.synthesize_batch(engine, [a] * n, ch, coeffs, seqs, shapes)
    s16, _ = aac_lc.synthesize_batch(engine, [b] * n, ch, coeffs, seqs, shapes, out="s16")
    engine.close_stream(a), engine.close_stream(b)
    # integer stage bit-exact on identical float input (decode_aac_access_unit, lib.rs:1793-1813)
    for f in range(n):
        assert np.array_equal(s16[f].ravel(), oracle.planar_f32_to_s16_interleaved(f32[f]))
    want, _ = oracle.synthesize_stream(coeffs, seqs, shapes)
    full = np.stack([oracle.planar_f32_to_s16_interleaved(want[f]).reshape(1024, ch) for f in range(n)])
    assert np.abs(s16.astype(np.int32) - full.astype(np.int32)).max() <= 1
    assert (s16 == 32767).any() or (s16 == -32768).any()


def test_handle_mirror_of_access_unit_decoder(engine, oracle):
    dec = aac_lc.AacLcSynth(44100, 2, engine)
    info = dec.frame_info()
    assert (info.sample_rate, info.channels, info.frames) == (44100, 2, 1024)
    x = spectra(oracle, 1, 2, 5, 100.0)[0]
    pcm = dec.synthesize(x, (0, 0), (1, 1))
    assert pcm.frames() == 1024 and pcm.channels().shape == (2, 1024)
    with pytest.raises(ValueError):
        dec.synthesize(x, (7, 0), (0, 0))
    dec.close()


def test_dequantize_matches_oracle(engine, oracle):
    rng = np.random.default_rng(5)
    q = rng.integers(-8191, 8192, 50000).astype(np.int16)
    q[:100] = 0
    sf = rng.integers(-60, 260, 50000).astype(np.int16)
    got = engine.dequantize(q, sf)
    L = oracle.lib()
    want = np.array([L.sko_dequantize_signed(int(a), int(b)) for a, b in zip(q[:4000], sf[:4000])], np.float32)
    # table entries come from powf on both sides; allow 1 ulp-ish
    assert np.allclose(got[:4000], want, rtol=3e-7, atol=0)
    assert np.all(got[:100] == 0)


def test_full_size_linearity_property(engine, oracle):
    """Config-2 sized batch (1024 streams x 8 frames, stereo): synthesis is linear in the spectrum,
    so synth(a) + synth(b) == synth(a + b) to rounding, and a spot-checked stream equals the oracle."""
    import torch
    n_streams, n_frames, ch = 1024, 8, 2
    g = torch.Generator(device="cuda").manual_seed(1)
    shape = (n_streams * n_frames, ch, 1024)
    a = (torch.rand(shape, generator=g, device="cuda") - 0.5) * 2000
    b = (torch.rand(shape, generator=g, device="cuda") - 0.5) * 2000
    ab = a + b
    torch.cuda.synchronize()  # inputs are produced on torch's stream, the engine runs on its own
    seqs = np.zeros((n_streams * n_frames, 2), np.uint8)
    shapes = np.tile(np.array([[0, 1]], np.uint8), (n_streams * n_frames, 1))
    outs = []
    for x in (a, b, ab):
        sids = [engine.open_stream(48000, ch) for _ in range(n_streams)]
        ids = np.repeat(np.array(sids, np.uint32), n_frames)
        descs, n = soundkit_amd.descs_from_arrays(ids, ch, seqs, shapes)
        plan = engine.plan(descs, n)
        assert plan.frames_ok == n
        y = torch.empty_like(x)
        torch.cuda.synchronize()
        plan.run_f32(x, y)
        engine.synchronize()
        outs.append(y)
        plan.destroy()
        for s in sids:
            engine.close_stream(s)
    err = (outs[0] + outs[1] - outs[2]).abs().max().item()
    assert err < 5e-7 * outs[2].abs().max().item() + 1e-9
    s = 517
    got = outs[0][s * n_frames:(s + 1) * n_frames].cpu().numpy()
    want, _ = oracle.synthesize_stream(a[s * n_frames:(s + 1) * n_frames].cpu().numpy(), seqs[:n_frames], shapes[:n_frames])
    assert rel_rms(got, want) < RMS_TOL
