"""The worker's decode -> resample data flow kept in 16 bits on the device, as the reference has it: the AAC decoder's
output is s16 (decode_aac_access_unit, soundkit-decoder lib.rs:1793-1813), apply_output_options turns it back into
f32 = s / 32768 (audio_data_to_f32_channels, lib.rs:3563-3617), resamples and narrows again (lib.rs:3619-3647).

  sk_aac_plan_run_s16_planar_dev           synthesis kernel writing float_sample_to_i16(x), planar
  sk_downsample_48k_16k_frames_s16_to_*    the bf16-matrix FIR reading those integers (two bf16 planes, 36 products)
"""
import numpy as np
import pytest

import soundkit_amd

pytestmark = pytest.mark.gpu


def make_batch(engine, oracle, layout, ch, n_streams, n_frames, gain=2500.0):
    coeffs = np.empty((n_streams, n_frames, ch, 1024), np.float32)
    for s in range(n_streams):
        for f in range(n_frames):
            for c in range(ch):
                coeffs[s, f, c] = oracle.seeded_spectrum(1024, 0x12345678 + 977 * s + 2 * f + c) * np.float32(gain)
    seq_chain = [0, 1, 2, 3, 0, 0]
    seqs = np.zeros((n_streams, n_frames, 2), np.uint8)
    shapes = np.zeros((n_streams, n_frames, 2), np.uint8)
    for s in range(n_streams):
        for f in range(n_frames):
            seqs[s, f] = seq_chain[(f + s) % 6] if s % 2 else 0
            shapes[s, f] = (f + s) & 1
    sids = np.array([engine.open_stream(48000, ch) for _ in range(n_streams)], np.uint32)
    if layout == "frame":
        order = [(s, f) for f in range(n_frames) for s in range(n_streams)]
        strides = (ch * 1024, n_streams * ch * 1024)
    else:
        order = [(s, f) for s in range(n_streams) for f in range(n_frames)]
        strides = (n_frames * ch * 1024, ch * 1024)
    packed = np.stack([coeffs[s, f] for s, f in order])
    descs, n = soundkit_amd.descs_from_arrays([sids[s] for s, f in order], ch, [seqs[s, f] for s, f in order],
                                              [shapes[s, f] for s, f in order])
    return coeffs, seqs, shapes, sids, order, strides, packed, descs, n


@pytest.mark.parametrize("layout", ["frame", "stream"])
@pytest.mark.parametrize("ch", [1, 2])
def test_planar_s16_synthesis_is_the_rounded_f32_synthesis(engine, oracle, layout, ch):
    """every window sequence and shape pair; the s16 kernel output equals float_sample_to_i16 applied (by the oracle) to
    the f32 kernel's output of the same frames, bit for bit, and the carried state is the same afterwards"""
    import torch
    n_streams, n_frames = 6, 6
    coeffs, seqs, shapes, sids, order, strides, packed, descs, n = make_batch(engine, oracle, layout, ch, n_streams, n_frames, 9000.0)
    plan = engine.plan(descs, n)
    d_coeffs = torch.from_numpy(packed).cuda()
    d_f32 = torch.empty_like(d_coeffs)
    d_s16 = torch.zeros(d_coeffs.shape, dtype=torch.int16, device="cuda")
    torch.cuda.synchronize()
    plan.run_f32(d_coeffs, d_f32)
    engine.synchronize()
    state_f32 = [engine.get_state(int(sid), ch) for sid in sids]
    for sid in sids:
        engine.reset_stream(int(sid))
    plan.run_s16_planar(d_coeffs, d_s16)
    engine.synchronize()
    f32 = d_f32.cpu().numpy()
    want = oracle.pcm_convert("FLOAT_TO_I16_ROUND", f32.ravel()).reshape(f32.shape)
    got = d_s16.cpu().numpy()
    assert np.array_equal(got, want)
    assert np.abs(got.astype(np.int32)).max() > 5000   # loud enough to mean something
    for sid, (delay, shape) in zip(sids, state_f32):
        d2, s2 = engine.get_state(int(sid), ch)
        assert np.array_equal(d2, delay) and np.array_equal(s2, shape)
    plan.destroy()
    for sid in sids:
        engine.close_stream(int(sid))


@pytest.mark.parametrize("gain,poison", [(3.0e5, None), (1.0e12, None), (1.0e36, None), (9000.0, "nan"), (9000.0, "inf")])
def test_planar_s16_synthesis_out_of_range_and_non_finite(engine, oracle, gain, poison):
    """float_sample_to_i16's clamp and its non-finite -> 0 rule (soundkit-decoder lib.rs:1815-1827) on the kernel's own
    conversion: PCM far beyond +-1 (also beyond the i32 range, and overflowing to inf inside the transform), NaN and inf
    coefficients; every window sequence.  Expected = the oracle's conversion of the f32 kernel's output of the same frames."""
    import torch
    ch, n_streams, n_frames = 2, 4, 6
    coeffs, seqs, shapes, sids, order, strides, packed, descs, n = make_batch(engine, oracle, "stream", ch, n_streams, n_frames, gain)
    if poison:
        packed = packed.copy()
        packed[1::3, :, 5::97] = np.float32(np.nan if poison == "nan" else np.inf)
        packed[2::3, 0, 11::131] = np.float32(np.nan if poison == "nan" else -np.inf)
    plan = engine.plan(descs, n)
    d_coeffs = torch.from_numpy(packed).cuda()
    d_f32 = torch.empty_like(d_coeffs)
    d_s16 = torch.zeros(d_coeffs.shape, dtype=torch.int16, device="cuda")
    torch.cuda.synchronize()
    plan.run_f32(d_coeffs, d_f32)
    engine.synchronize()
    for sid in sids:
        engine.reset_stream(int(sid))
    plan.run_s16_planar(d_coeffs, d_s16)
    engine.synchronize()
    f32 = d_f32.cpu().numpy()
    want = oracle.pcm_convert("FLOAT_TO_I16_ROUND", f32.ravel()).reshape(f32.shape)
    got = d_s16.cpu().numpy()
    assert np.array_equal(got, want)
    if poison:
        assert not np.isfinite(f32).all() and (want[~np.isfinite(f32)] == 0).all()
    else:
        assert (got == 32767).any() and (got == -32768).any()
        assert gain < 1e30 or (np.abs(f32) > 2.0 ** 40).any()   # beyond what the integer conversion itself can hold
    plan.destroy()
    for sid in sids:
        engine.close_stream(int(sid))


@pytest.mark.parametrize("ch,n_streams,n_frames,layout", [(2, 37, 5, "stream"), (1, 21, 3, "frame"), (2, 16, 7, "frame"), (1, 33, 4, "stream")])
def test_fir_on_s16_rows(engine, oracle, ch, n_streams, n_frames, layout):
    """random full-range s16 rows (extremes included): the f32 result against the filter evaluated in f64 on s / 32768
    (<= 1e-6 relative RMS, north_star), against the oracle's f32 chain, and the s16 result = float_sample_to_i16 of it"""
    import torch
    rng = np.random.default_rng(11 + ch + n_streams)
    x = rng.integers(-32768, 32768, (n_streams, n_frames, ch, 1024), dtype=np.int64).astype(np.int16)
    x[0, 0, 0, :6] = [32767, -32768, 0, 1, -1, 255]
    if layout == "frame":
        packed = np.ascontiguousarray(x.transpose(1, 0, 2, 3))
        strides = (ch * 1024, n_streams * ch * 1024)
    else:
        packed = x
        strides = (n_frames * ch * 1024, ch * 1024)
    d_in = torch.from_numpy(packed).cuda()
    n_out = engine.downsample_out_frames(n_frames * 1024)
    f_stride = (n_out + 3) // 4 * 4
    o_stride = (n_out + 7) // 8 * 8
    d_f32 = torch.zeros((n_streams * ch, f_stride), device="cuda")
    d_s16 = torch.zeros((n_streams, o_stride, ch), dtype=torch.int16, device="cuda")
    torch.cuda.synchronize()
    assert engine.downsample_48k_16k_frames_s16_to_f32_dev(d_in, strides[0], strides[1], ch, n_streams, n_frames, d_f32, f_stride) == n_out
    assert engine.downsample_48k_16k_frames_s16_to_s16_dev(d_in, strides[0], strides[1], ch, n_streams, n_frames, d_s16, o_stride) == n_out
    engine.synchronize()
    taps = engine.taps().astype(np.float64)
    got = d_f32.cpu().numpy()[:, :n_out].reshape(n_streams, ch, n_out)
    s16 = d_s16.cpu().numpy()
    assert not s16[:, n_out:].any()
    worst64 = worst32 = 0.0
    for s in range(n_streams):
        rows = x[s].transpose(1, 0, 2).reshape(ch, n_frames * 1024).astype(np.float32) / np.float32(32768.0)
        want32 = oracle.downsample_planar(rows, 48000, 16000)
        padded = np.concatenate([np.zeros((ch, 125)), rows.astype(np.float64), np.zeros((ch, 256))], axis=1)
        want64 = np.stack([[np.dot(taps, padded[c, 3 * m:3 * m + 256]) for m in range(n_out)] for c in range(ch)])
        rms = np.sqrt(np.mean(want64 ** 2))
        worst64 = max(worst64, np.sqrt(np.mean((got[s] - want64) ** 2)) / rms)
        worst32 = max(worst32, np.sqrt(np.mean((got[s] - want32.astype(np.float64)) ** 2)) / rms)
        assert np.array_equal(s16[s, :n_out], oracle.planar_f32_to_s16_interleaved(got[s]).reshape(n_out, ch))
        d = np.abs(s16[s, :n_out].astype(np.int32) - oracle.planar_f32_to_s16_interleaved(want32).reshape(n_out, ch))
        assert d.max() <= 1 and (d > 0).mean() < 0.01
    assert worst64 < 1e-6 and worst32 < 1e-6, (worst64, worst32)


@pytest.mark.parametrize("layout", ["frame", "stream"])
@pytest.mark.parametrize("ch", [1, 2])
def test_s16_chain_matches_the_oracle_chain(engine, oracle, layout, ch):
    import torch
    n_streams, n_frames = 5, 6
    coeffs, seqs, shapes, sids, order, strides, packed, descs, n = make_batch(engine, oracle, layout, ch, n_streams, n_frames)
    plan = engine.plan(descs, n)
    d_coeffs = torch.from_numpy(packed).cuda()
    d_pcm16 = torch.zeros(d_coeffs.shape, dtype=torch.int16, device="cuda")
    n_out = engine.downsample_out_frames(n_frames * 1024)
    o_stride = (n_out + 7) // 8 * 8
    d_out = torch.zeros((n_streams, o_stride, ch), dtype=torch.int16, device="cuda")
    torch.cuda.synchronize()
    plan.run_s16_planar(d_coeffs, d_pcm16)
    assert engine.downsample_48k_16k_frames_s16_to_s16_dev(d_pcm16, strides[0], strides[1], ch, n_streams, n_frames, d_out, o_stride) == n_out
    engine.synchronize()
    out = d_out.cpu().numpy()
    differing = total = 0
    for s in range(n_streams):
        pcm, _ = oracle.synthesize_stream(coeffs[s], seqs[s], shapes[s])
        planar = np.ascontiguousarray(pcm.transpose(1, 0, 2).reshape(ch, n_frames * 1024))
        q = oracle.pcm_convert("FLOAT_TO_I16_ROUND", planar.ravel()).reshape(planar.shape).astype(np.float32) / np.float32(32768.0)
        want = oracle.planar_f32_to_s16_interleaved(oracle.downsample_planar(q, 48000, 16000)).reshape(n_out, ch)
        d = np.abs(out[s, :n_out].astype(np.int32) - want.astype(np.int32))
        assert d.max() <= 1, (s, int(d.max()))
        differing, total = differing + int((d > 0).sum()), total + d.size
        assert np.abs(want).max() > 100
    assert differing <= 0.01 * total
    plan.destroy()
    for sid in sids:
        engine.close_stream(int(sid))


def test_full_size_s16_chain_spot_checked_against_oracle(oracle):
    """bench.py's default workload at full size (4096 streams x 64 stereo frames, frame-major): three streams restated on the CPU"""
    import os
    import sys
    import torch
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
    import bench
    n_streams, n_frames, ch = 4096, 64, 2
    eng = soundkit_amd.Engine(0, n_streams)
    try:
        coeffs = bench.seeded_spectra(torch, torch.device("cuda:0"), n_streams, n_frames, ch) * bench.SPECTRUM_GAIN
        picked = [0, 2049, 4095]
        host = {s: coeffs.view(n_streams, n_frames, ch, 1024)[s].cpu().numpy() for s in picked}
        packed = coeffs.view(n_streams, n_frames, ch, 1024).transpose(0, 1).contiguous().view(-1, ch, 1024)
        del coeffs
        sids = np.array([eng.open_stream(48000, ch) for _ in range(n_streams)], np.uint32)
        shape_of_frame = (np.arange(n_frames) & 1).astype(np.uint8)
        descs, n = soundkit_amd.descs_from_arrays(np.tile(sids, n_frames), ch, np.zeros((n_streams * n_frames, 2), np.uint8),
                                                  np.repeat(shape_of_frame, n_streams)[:, None].repeat(2, 1))
        plan = eng.plan(descs, n)
        pcm16 = torch.zeros(packed.shape, dtype=torch.int16, device="cuda")
        n_out = eng.downsample_out_frames(n_frames * 1024)
        stride = (n_out + 7) // 8 * 8
        out = torch.zeros((n_streams, stride, ch), dtype=torch.int16, device="cuda")
        torch.cuda.synchronize()
        plan.run_s16_planar(packed, pcm16)
        got = eng.downsample_48k_16k_frames_s16_to_s16_dev(pcm16, ch * 1024, n_streams * ch * 1024, ch, n_streams, n_frames, out, stride)
        eng.synchronize()
        assert got == n_out
        seqs = np.zeros((n_frames, 2), np.uint8)
        shapes = np.repeat(shape_of_frame[:, None], 2, 1)
        for s in picked:
            ref_pcm, _ = oracle.synthesize_stream(host[s], seqs, shapes)
            planar = np.ascontiguousarray(ref_pcm.transpose(1, 0, 2).reshape(ch, n_frames * 1024))
            q = oracle.pcm_convert("FLOAT_TO_I16_ROUND", planar.ravel()).reshape(planar.shape).astype(np.float32) / np.float32(32768.0)
            want = oracle.planar_f32_to_s16_interleaved(oracle.downsample_planar(q, 48000, 16000)).reshape(n_out, ch)
            mine = out[s, :n_out].cpu().numpy()
            d = np.abs(mine.astype(np.int32) - want.astype(np.int32))
            assert d.max() <= 1 and (d > 0).mean() < 0.01, (s, int(d.max()), float((d > 0).mean()))
            assert np.abs(mine).max() > 100
        assert not out[:, n_out:].any()
        plan.destroy()
    finally:
        eng.close()


@pytest.mark.parametrize("layout", ["frame", "stream"])
@pytest.mark.parametrize("ch,n_frames", [(2, 7), (1, 7), (2, 3), (2, 12), (1, 1)])
def test_fused_tail_equals_the_two_calls(engine, oracle, layout, ch, n_frames, monkeypatch):
    """(In a packed-f32 build the entry point is withdrawn and this runs the kernel behind its diagnostic switch, at a size where a
    launch is a few workgroups and the platform's co-residency defect does not show: include/soundkit_amd.h.)
    sk_aac_plan_run_tail_s16_dev (k_aac_tail: synthesis, s16 narrowing, the FIR's f16 planes in an LDS ring, the MFMA FIR on
    the wave's own channel, interleaved s16 out -- one launch, no PCM in HBM) against sk_aac_plan_run_s16_planar_dev +
    sk_downsample_48k_16k_frames_s16_to_s16_dev: three calls back to back on the same streams (the overlap state carries;
    OnlyLong, LongStart and LongStop frames with both window shapes; frame counts that end in partial tiles), bit for bit,
    and the carried state afterwards."""
    import torch
    from soundkit_amd._lib import SoundkitError
    monkeypatch.setenv("SK_AAC_TAIL_ONE_LAUNCH", "1")
    n_streams = 10
    coeffs = np.empty((n_streams, n_frames, ch, 1024), np.float32)
    for s in range(n_streams):
        for f in range(n_frames):
            for c in range(ch):
                coeffs[s, f, c] = oracle.seeded_spectrum(1024, 0x2468ACE + 977 * s + 2 * f + c) * np.float32(9000.0)
    seq_chain = [0, 1, 3, 0, 0, 1, 3]
    seqs = np.zeros((n_streams, n_frames, 2), np.uint8)
    shapes = np.zeros((n_streams, n_frames, 2), np.uint8)
    for s in range(n_streams):
        for f in range(n_frames):
            seqs[s, f] = seq_chain[(f + s) % 7] if s % 2 else 0
            shapes[s, f] = [(f + s) & 1, (f // 2 + s) & 1]
    sids = np.array([engine.open_stream(48000, ch) for _ in range(n_streams)], np.uint32)
    if layout == "frame":
        order = [(s, f) for f in range(n_frames) for s in range(n_streams)]
        strides = (ch * 1024, n_streams * ch * 1024)
    else:
        order = [(s, f) for s in range(n_streams) for f in range(n_frames)]
        strides = (n_frames * ch * 1024, ch * 1024)
    packed = np.stack([coeffs[s, f] for s, f in order])
    descs, n = soundkit_amd.descs_from_arrays([sids[s] for s, f in order], ch, [seqs[s, f] for s, f in order], [shapes[s, f] for s, f in order])
    plan = engine.plan(descs, n)
    inputs = [torch.from_numpy(packed * np.float32(g)).cuda() for g in (1.0, 0.4, 2.5)]
    n_out = engine.downsample_out_frames(n_frames * 1024)
    stride = (n_out + 7) // 8 * 8
    torch.cuda.synchronize()
    want = []
    pcm16 = torch.zeros(inputs[0].shape, dtype=torch.int16, device="cuda")
    for x in inputs:
        out = torch.zeros((n_streams, stride, ch), dtype=torch.int16, device="cuda")
        torch.cuda.synchronize()
        plan.run_s16_planar(x, pcm16)
        assert engine.downsample_48k_16k_frames_s16_to_s16_dev(pcm16, strides[0], strides[1], ch, n_streams, n_frames, out, stride) == n_out
        engine.synchronize()
        want.append(out.cpu().numpy())
    state_want = [engine.get_state(int(sid), ch) for sid in sids]
    for sid in sids:
        engine.reset_stream(int(sid))
    got = []
    for x in inputs:
        out = torch.zeros((n_streams, stride, ch), dtype=torch.int16, device="cuda")
        torch.cuda.synchronize()
        if ch == 1 and n_streams % 2:   # an odd number of mono channels leaves one without a partner: not the fused kernel's case
            with pytest.raises(SoundkitError):
                plan.run_tail_s16(x, strides[0], ch, n_frames, out, stride)
            return
        assert plan.run_tail_s16(x, strides[0], ch, n_frames, out, stride) == n_out
        engine.synchronize()
        got.append(out.cpu().numpy())
    for k in range(len(inputs)):
        assert np.array_equal(got[k], want[k]), (k, int((got[k] != want[k]).sum()), np.argwhere(got[k] != want[k])[:5].tolist())
    assert np.abs(want[0].astype(np.int32)).max() > 1000
    for sid, (d, sh) in zip(sids, state_want):
        d2, sh2 = engine.get_state(int(sid), ch)
        assert np.array_equal(d, d2) and np.array_equal(sh, sh2)
    plan.destroy()
    for sid in sids:
        engine.close_stream(int(sid))


def test_fused_tail_is_withdrawn_in_a_packed_f32_build(engine, monkeypatch):
    """A library built with packed-f32 instructions (make PACKED_F32=1) refuses the one-launch tail for every plan unless
    SK_AAC_TAIL_ONE_LAUNCH=1 (SK_ERR_UNSUPPORTED: use the two calls) and writes nothing: synthesis waves using those instructions and
    matrix-instruction waves sharing SIMDs inside one launch is what the platform computes wrongly (profiles/r04_lanes_corruption.md).
    The default build has no such instructions: the entry point works (the other tests here)."""
    import torch
    from soundkit_amd._lib import SoundkitError, lib
    if lib.sk_kernels_use_packed_f32() == 0:
        pytest.skip("default build: no packed-f32 instructions, the one-launch tail is exact")
    monkeypatch.delenv("SK_AAC_TAIL_ONE_LAUNCH", raising=False)
    n_streams, n_frames, ch = 4, 3, 2
    sids = np.array([engine.open_stream(48000, ch) for _ in range(n_streams)], np.uint32)
    ids = np.repeat(sids, n_frames)
    descs, n = soundkit_amd.descs_from_arrays(ids, ch, np.zeros((n_streams * n_frames, 2), np.uint8), np.zeros((n_streams * n_frames, 2), np.uint8))
    plan = engine.plan(descs, n)
    x = torch.ones((n_streams * n_frames, ch, 1024), dtype=torch.float32, device="cuda")
    n_out = engine.downsample_out_frames(n_frames * 1024)
    stride = (n_out + 7) // 8 * 8
    out = torch.zeros((n_streams, stride, ch), dtype=torch.int16, device="cuda")
    with pytest.raises(SoundkitError) as err:
        plan.run_tail_s16(x, n_frames * ch * 1024, ch, n_frames, out, stride)
    assert err.value.status == -6
    engine.synchronize()
    assert not out.any()
    plan.destroy()
    for sid in sids:
        engine.close_stream(int(sid))


def test_fused_tail_at_a_full_device_equals_the_two_calls():
    """The one-launch tail with every CU holding several workgroups (4096 stereo streams x 8 frames): synthesis waves and the FIR's
    matrix instructions share SIMDs there.  Built with packed-f32 instructions the kernel came out wrong in half of the streams at this
    size, differently in every run (round 4, profiles/r04_lanes_corruption.md) -- the tests of round 3 ran ten streams and never saw
    it.  Three runs against the two-launch chain, bit for bit."""
    import torch
    from soundkit_amd._lib import lib
    if lib.sk_kernels_use_packed_f32() != 0:
        pytest.skip("packed-f32 build: the one-launch tail is withdrawn")
    dev = torch.device("cuda")
    streams, frames, ch = 4096, 8, 2
    eng = soundkit_amd.Engine(0, streams + 8)
    try:
        g = torch.Generator(device="cpu").manual_seed(5)
        coeffs = ((torch.rand((streams * frames, ch, 1024), generator=g) * 2 - 1) * 2.5e5).to(dev)
        sids = np.array([eng.open_stream(48000, ch) for _ in range(streams)], np.uint32)
        shapes = np.tile((np.arange(frames) & 1).astype(np.uint8), streams)[:, None].repeat(2, 1)
        descs, n = soundkit_amd.descs_from_arrays(np.repeat(sids, frames), ch, np.zeros((streams * frames, 2), np.uint8), shapes)
        plan = eng.plan(descs, n)
        stream_stride, frame_stride = frames * ch * 1024, ch * 1024
        n_out = eng.downsample_out_frames(frames * 1024)
        stride = (n_out + 7) // 8 * 8
        torch.cuda.synchronize()
        pcm16 = torch.zeros(coeffs.shape, dtype=torch.int16, device=dev)
        want = torch.zeros((streams, stride, ch), dtype=torch.int16, device=dev)
        plan.run_s16_planar(coeffs, pcm16)
        assert eng.downsample_48k_16k_frames_s16_to_s16_dev(pcm16, stream_stride, frame_stride, ch, streams, frames, want, stride) == n_out
        eng.synchronize()
        assert float(want.float().pow(2).mean().sqrt()) > 500
        for _ in range(3):
            for sid in sids:
                eng.reset_stream(int(sid))
            got = torch.zeros_like(want)
            torch.cuda.synchronize()
            assert plan.run_tail_s16(coeffs, stream_stride, ch, frames, got, stride) == n_out
            eng.synchronize()
            assert torch.equal(got, want), int((got != want).sum())
        plan.destroy()
    finally:
        eng.close()
