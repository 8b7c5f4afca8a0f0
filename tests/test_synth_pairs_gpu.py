"""k_aac_synth_pair (two OnlyLong channels of equal length per wave) against k_aac_synth<.., true> (one channel per wave):
per channel the two kernels perform the same operations in the same order, so a channel's samples and carried state must
not depend on whether it found a partner, nor on who the partner is.  (Parity of both with the oracle:
tests/test_aac_synth_gpu.py, tests/test_s16_chain_gpu.py -- their all-long stereo streams run as pairs.)"""
import numpy as np
import pytest

import soundkit_amd

pytestmark = pytest.mark.gpu


def spectra(oracle, n, seed, gain):
    return np.stack([oracle.seeded_spectrum(1024, seed + 31 * f) * np.float32(gain) for f in range(n)])


@pytest.mark.parametrize("out", ["f32", "s16"])
def test_a_channel_does_not_depend_on_its_partner(engine, oracle, out):
    import torch
    left, right = spectra(oracle, 6, 0x1234, 7000.0), spectra(oracle, 6, 0x9876, 11000.0)
    # stream 0: stereo, 6 frames (L, R pair up); 1: mono = L, 6 frames (pairs with 3); 2: mono = R, 5 frames (stays alone: the
    # one-channel kernel); 3: mono = R, 6 frames
    sids = [engine.open_stream(48000, 2)] + [engine.open_stream(48000, 1) for _ in range(3)]
    streams, chans, shapes, chunks = [], [], [], []
    for f in range(6):
        for k, sid in enumerate(sids):
            if k == 2 and f == 5:
                continue
            streams.append(sid)
            chans.append(2 if k == 0 else 1)
            shapes.append([(f + 1) & 1, (f + 1) & 1] if k in (0, 1) else [f & 1, f & 1])
            chunks.append(np.stack([left[f], right[f]]) if k == 0 else (left[f] if k == 1 else right[f])[None])
    shapes = np.array(shapes, np.uint8)
    if True:  # the stereo stream's R channel uses the shape sequence of the mono R streams
        for i, (sid, s) in enumerate(zip(streams, shapes)):
            if sid == sids[0]:
                f = sum(1 for j in range(i) if streams[j] == sid)
                shapes[i] = [(f + 1) & 1, f & 1]
    packed = np.concatenate([c.reshape(-1) for c in chunks]).astype(np.float32)
    descs, n = soundkit_amd.descs_from_arrays(streams, np.array(chans, np.uint8), np.zeros((len(streams), 2), np.uint8), shapes)
    plan = engine.plan(descs, n)
    d_in = torch.from_numpy(packed).cuda()
    if out == "f32":
        d_out = torch.empty_like(d_in)
        torch.cuda.synchronize()
        plan.run_f32(d_in, d_out)
    else:
        d_out = torch.zeros(d_in.shape, dtype=torch.int16, device="cuda")
        torch.cuda.synchronize()
        plan.run_s16_planar(d_in, d_out)
    engine.synchronize()
    got = d_out.cpu().numpy()
    per = {k: [] for k in range(4)}
    at = 0
    for sid, ch in zip(streams, chans):
        per[sids.index(sid)].append(got[at:at + ch * 1024].reshape(ch, 1024))
        at += ch * 1024
    stereo = np.stack(per[0])          # [6][2][1024]
    m1, m2, m3 = (np.stack(per[k])[:, 0] for k in (1, 2, 3))
    assert np.array_equal(m1, stereo[:, 0])        # paired with R / paired with another stream's channel
    assert np.array_equal(m3, stereo[:, 1])
    assert np.array_equal(m2, stereo[:5, 1])       # alone (one-channel kernel) / paired
    assert np.abs(stereo.astype(np.float64)).max() > (0.05 if out == "f32" else 1500)
    # carried state: the same overlap whatever kernel produced it
    d0, s0 = engine.get_state(sids[0], 2)
    d1, s1 = engine.get_state(sids[1], 1)
    d3, s3 = engine.get_state(sids[3], 1)
    assert np.array_equal(d0[0], d1[0]) and np.array_equal(d0[1], d3[0]) and s0[0] == s1[0] and s0[1] == s3[0]
    plan.destroy()
    for sid in sids:
        engine.close_stream(sid)


def test_transition_frames_take_the_long_kernels(engine, oracle):
    """LongStart / LongStop frames run the long code path with their piecewise windows as tables (engine.cpp build_tables,
    dsp.rs:353-387): a stereo stream with transitions but no EightShort frame goes to the two-channel kernel, a mono stream of
    another length to the one-channel long kernel; both against the oracle, and the shared channel bit for bit."""
    from soundkit_amd import aac_lc
    seq_l = [0, 1, 3, 0, 1, 3, 3, 1, 0, 0]          # not a sequence an encoder would emit; the arithmetic does not care
    seq_r = [1, 3, 0, 0, 0, 1, 1, 3, 3, 0]
    n = len(seq_l)
    coeffs = np.stack([np.stack([oracle.seeded_spectrum(1024, 0x777 + 5 * f + c) * np.float32(6000.0) for c in range(2)]) for f in range(n)])
    seqs = np.array(list(zip(seq_l, seq_r)), np.uint8)
    shapes = np.array([[(f // 2) & 1, (f // 3) & 1] for f in range(n)], np.uint8)
    sid = engine.open_stream(48000, 2)
    pcm, status = aac_lc.synthesize_batch(engine, [sid] * n, 2, coeffs, seqs, shapes)
    assert np.all(status == 0)
    want, chans = oracle.synthesize_stream(coeffs, seqs, shapes)
    den = np.sqrt(np.mean(want.astype(np.float64) ** 2))
    assert np.sqrt(np.mean((pcm.astype(np.float64) - want) ** 2)) / den < 1.0e-6
    assert np.abs(pcm - want).max() < 2e-6 * np.abs(want).max()
    delay, shape = engine.get_state(sid, 2)
    for c in range(2):
        assert np.abs(delay[c] - chans[c].delay).max() < 2e-6 * max(1e-9, np.abs(chans[c].delay).max()) and shape[c] == chans[c].prev_shape
    engine.close_stream(sid)
    # the left channel again as a mono stream of 9 frames (no partner of that length: one-channel kernel)
    mono = engine.open_stream(48000, 1)
    pcm1, status = aac_lc.synthesize_batch(engine, [mono] * (n - 1), 1, coeffs[:n - 1, :1], seqs[:n - 1], shapes[:n - 1])
    assert np.all(status == 0) and np.array_equal(pcm1[:, 0], pcm[:n - 1, 0])
    engine.close_stream(mono)
