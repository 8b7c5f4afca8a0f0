"""The worker's whole device-side tail on the GPU, chained without leaving HBM:
AAC synthesis -> 48k->16k FIR over the frame-packed planar PCM in place -> interleaved s16."""
import numpy as np
import pytest

import soundkit_amd

pytestmark = pytest.mark.gpu


def rel_rms(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return np.sqrt(np.mean((a - b) ** 2)) / (np.sqrt(np.mean(b * b)) or 1.0)


@pytest.mark.parametrize("layout", ["frame", "stream"])
@pytest.mark.parametrize("ch", [1, 2])
def test_synth_fir_s16_chain_matches_oracle(engine, oracle, layout, ch):
    import torch
    n_streams, n_frames = 5, 6
    coeffs = np.empty((n_streams, n_frames, ch, 1024), np.float32)
    for s in range(n_streams):
        for f in range(n_frames):
            for c in range(ch):
                coeffs[s, f, c] = oracle.seeded_spectrum(1024, 0x12345678 + 977 * s + 2 * f + c) * np.float32(2500.0)
    seq_chain = [0, 1, 2, 3, 0, 0]
    seqs = np.zeros((n_streams, n_frames, 2), np.uint8)
    shapes = np.zeros((n_streams, n_frames, 2), np.uint8)
    for s in range(n_streams):
        for f in range(n_frames):
            seqs[s, f] = seq_chain[(f + s) % 6] if s % 2 else 0
            shapes[s, f] = (f + s) & 1
    sids = np.array([engine.open_stream(48000, ch) for _ in range(n_streams)], np.uint32)
    if layout == "frame":   # one tick of every stream after another
        order = [(s, f) for f in range(n_frames) for s in range(n_streams)]
        stream_stride, frame_stride = ch * 1024, n_streams * ch * 1024
    else:
        order = [(s, f) for s in range(n_streams) for f in range(n_frames)]
        stream_stride, frame_stride = n_frames * ch * 1024, ch * 1024
    packed = np.stack([coeffs[s, f] for s, f in order])
    descs, n = soundkit_amd.descs_from_arrays([sids[s] for s, f in order], ch, [seqs[s, f] for s, f in order],
                                              [shapes[s, f] for s, f in order])
    plan = engine.plan(descs, n)
    assert plan.frames_ok == n
    d_coeffs = torch.from_numpy(packed).cuda()
    d_pcm = torch.empty_like(d_coeffs)
    n_out = engine.downsample_out_frames(n_frames * 1024)
    out_stride = (n_out + 3) // 4 * 4
    d_fir = torch.zeros((n_streams * ch, out_stride), device="cuda")
    d_s16 = torch.zeros((n_streams, n_out, ch), dtype=torch.int16, device="cuda")
    torch.cuda.synchronize()
    plan.run_f32(d_coeffs, d_pcm)
    got_n = engine.downsample_48k_16k_frames_dev(d_pcm, stream_stride, frame_stride, ch, n_streams, n_frames, d_fir, out_stride)
    engine.f32_planar_to_bytes_batch_dev(soundkit_amd.engine.FMT_S16LE, d_fir, n_streams, out_stride, n_out, ch, d_s16)
    engine.synchronize()
    assert got_n == n_out
    fir = d_fir.cpu().numpy()[:, :n_out].reshape(n_streams, ch, n_out)
    s16 = d_s16.cpu().numpy()
    for s in range(n_streams):
        pcm, _ = oracle.synthesize_stream(coeffs[s], seqs[s], shapes[s])
        planar = np.ascontiguousarray(pcm.transpose(1, 0, 2).reshape(ch, n_frames * 1024))
        want = oracle.downsample_planar(planar, 48000, 16000)
        assert want.shape == (ch, n_out)
        assert rel_rms(fir[s], want) < 1e-6, s
        # integer stage: bit-exact on the GPU's own float input, within 1 LSB of the all-CPU chain
        assert np.array_equal(s16[s].ravel(), oracle.planar_f32_to_s16_interleaved(fir[s]))
        assert np.abs(s16[s].astype(np.int32) - oracle.planar_f32_to_s16_interleaved(want).reshape(n_out, ch)).max() <= 1
    plan.destroy()
    for sid in sids:
        engine.close_stream(int(sid))


@pytest.mark.parametrize("ch,n_streams,n_frames,stride_pad", [(2, 37, 5, 0), (1, 21, 3, 0), (2, 16, 2, 3), (1, 33, 4, 5)])
def test_fused_s16_epilogue_equals_fir_then_convert(engine, ch, n_streams, n_frames, stride_pad):
    """sk_downsample_48k_16k_frames_s16_dev = the FIR followed by f32_channels_to_bytes(16): same bytes, one kernel
    (aligned strides take the vector stores, odd ones the scalar tail; amplitudes go past +-1 to hit the clamp)."""
    import torch
    g = torch.Generator(device="cuda").manual_seed(7 + ch)
    pcm = (torch.rand((n_streams, n_frames, ch, 1024), generator=g, device="cuda") * 2 - 1) * 1.3
    pcm[0, 0, 0, :8] = torch.tensor([float("nan"), float("inf"), -float("inf"), 1.0, -1.0, 0.5 / 32767, -0.5 / 32768, 0.0])
    n_out = engine.downsample_out_frames(n_frames * 1024)
    f_stride = (n_out + 3) // 4 * 4
    d_fir = torch.zeros((n_streams * ch, f_stride), device="cuda")
    sep = torch.zeros((n_streams, n_out, ch), dtype=torch.int16, device="cuda")
    o_stride = (n_out + 7) // 8 * 8 + stride_pad
    fused = torch.zeros((n_streams, o_stride, ch), dtype=torch.int16, device="cuda")
    torch.cuda.synchronize()
    stream_stride, frame_stride = n_frames * ch * 1024, ch * 1024
    engine.downsample_48k_16k_frames_dev(pcm, stream_stride, frame_stride, ch, n_streams, n_frames, d_fir, f_stride)
    engine.f32_planar_to_bytes_batch_dev(soundkit_amd.engine.FMT_S16LE, d_fir, n_streams, f_stride, n_out, ch, sep)
    got = engine.downsample_48k_16k_frames_s16_dev(pcm, stream_stride, frame_stride, ch, n_streams, n_frames, fused, o_stride)
    engine.synchronize()
    assert got == n_out
    assert torch.equal(fused[:, :n_out], sep)
    assert not fused[:, n_out:].any()  # nothing written past the end


def test_full_size_default_chain_spot_checked_against_oracle(oracle):
    """The bench's default workload at its full size (4096 streams x 64 stereo frames, frame-major, the reference's seeded
    spectra scaled to audible level, s16 in the FIR epilogue): three streams out of the batch are restated on the CPU.  The FIR sums in another
    order than the oracle, so an s16 value may differ by one LSB on a small fraction of samples."""
    import os
    import sys
    import torch
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
    import bench
    n_streams, n_frames, ch = 4096, 64, 2
    eng = soundkit_amd.Engine(0, n_streams)
    try:
        # the reference's test spectrum is in +-12, which synthesises to well under one 16-bit step: scaled up as in the
        # small-batch test above so that the s16 comparison has something to compare
        coeffs = bench.seeded_spectra(torch, torch.device("cuda:0"), n_streams, n_frames, ch) * 2500.0  # [stream][frame]
        picked = [0, 1777, 4095]
        host = {s: coeffs.view(n_streams, n_frames, ch, 1024)[s].cpu().numpy() for s in picked}
        packed = coeffs.view(n_streams, n_frames, ch, 1024).transpose(0, 1).contiguous().view(-1, ch, 1024)
        del coeffs
        sids = np.array([eng.open_stream(48000, ch) for _ in range(n_streams)], np.uint32)
        shape_of_frame = (np.arange(n_frames) & 1).astype(np.uint8)
        descs, n = soundkit_amd.descs_from_arrays(np.tile(sids, n_frames), ch, np.zeros((n_streams * n_frames, 2), np.uint8),
                                                  np.repeat(shape_of_frame, n_streams)[:, None].repeat(2, 1))
        plan = eng.plan(descs, n)
        assert plan.frames_ok == n
        pcm = torch.empty_like(packed)
        n_out = eng.downsample_out_frames(n_frames * 1024)
        stride = (n_out + 7) // 8 * 8
        out = torch.zeros((n_streams, stride, ch), dtype=torch.int16, device="cuda")
        torch.cuda.synchronize()
        plan.run_f32(packed, pcm)
        got = eng.downsample_48k_16k_frames_s16_dev(pcm, ch * 1024, n_streams * ch * 1024, ch, n_streams, n_frames, out, stride)
        eng.synchronize()
        assert got == n_out
        seqs = np.zeros((n_frames, 2), np.uint8)
        shapes = np.repeat(shape_of_frame[:, None], 2, 1)
        for s in picked:
            ref_pcm, _ = oracle.synthesize_stream(host[s], seqs, shapes)
            planar = np.ascontiguousarray(ref_pcm.transpose(1, 0, 2).reshape(ch, n_frames * 1024))
            want = oracle.planar_f32_to_s16_interleaved(oracle.downsample_planar(planar, 48000, 16000)).reshape(n_out, ch)
            mine = out[s, :n_out].cpu().numpy()
            d = np.abs(mine.astype(np.int32) - want.astype(np.int32))
            assert d.max() <= 1 and (d > 0).mean() < 0.01, (s, int(d.max()), float((d > 0).mean()))
            assert np.abs(mine).max() > 100  # not silence
        assert not out[:, n_out:].any()
        plan.destroy()
    finally:
        eng.close()
