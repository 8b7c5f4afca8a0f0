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


@pytest.mark.parametrize("seed", [1, 2, 3])
def test_random_plans_against_the_oracle(engine, oracle, seed):
    """Random batches for the schedule builder: mono and stereo streams with 1-7 frames each in an arbitrary interleaving,
    random window sequences (no EightShort / with transitions / with EightShort) and shapes -- so that one plan holds
    two-channel-kernel pairs (equal counts), leftover one-channel long tasks and general tasks -- in two consecutive calls
    (carried state), f32 against the oracle and s16 = float_sample_to_i16 of the f32 result."""
    rng = np.random.default_rng(4000 + seed)
    n_streams = 29
    chans = [int(rng.integers(1, 3)) for _ in range(n_streams)]
    sids = [engine.open_stream(48000, c) for c in chans]
    kind = [int(rng.integers(0, 3)) for _ in range(n_streams)]   # 0 OnlyLong, 1 long + transitions, 2 anything
    history = {s: ([], [], []) for s in range(n_streams)}          # coeffs, seqs, shapes over both calls
    got_f32 = {s: [] for s in range(n_streams)}
    for call in range(2):
        counts = [int(rng.integers(1, 8)) if rng.random() > 0.15 else 0 for _ in range(n_streams)]
        if seed == 1:
            counts = [c if c == 0 else 4 + (s % 2) for s, c in enumerate(counts)]   # many equal counts: many pairs
        frames = [(s, f) for s in range(n_streams) for f in range(counts[s])]
        order = rng.permutation(len(frames))
        per = {}
        for s in range(n_streams):
            c = chans[s]
            x = np.stack([np.stack([oracle.seeded_spectrum(1024, 77 * seed + 1000 * call + 31 * s + 7 * f + ch) * np.float32(5000.0)
                                    for ch in range(c)]) for f in range(counts[s])]) if counts[s] else np.zeros((0, c, 1024), np.float32)
            choices = {0: [0], 1: [0, 1, 3], 2: [0, 1, 2, 3]}[kind[s]]
            seqs = rng.choice(choices, (counts[s], 2)).astype(np.uint8)
            shapes = rng.integers(0, 2, (counts[s], 2)).astype(np.uint8)
            per[s] = (x, seqs, shapes)
            for k, v in zip(range(3), (x, seqs, shapes)):
                history[s][k].append(v)
        # frames of a stream keep their order inside the interleaving
        seen = {s: 0 for s in range(n_streams)}
        sequence = []
        for i in order:
            s = frames[i][0]
            sequence.append((s, seen[s]))
            seen[s] += 1
        descs = [(sids[s], chans[s], per[s][1][f], per[s][2][f]) for s, f in sequence]
        arr, n = soundkit_amd.make_descs(descs)
        coeffs = np.concatenate([per[s][0][f].ravel() for s, f in sequence]) if sequence else np.zeros(0, np.float32)
        states = [engine.get_state(sids[s], chans[s]) for s in range(n_streams)]
        pcm, status = engine.aac_synthesize(arr, n, coeffs)
        assert np.all(status == 0)
        for s in range(n_streams):   # the same call again from the same state, s16 out: the rounded f32
            engine.set_state(sids[s], *states[s])
        s16, status = engine.aac_synthesize(arr, n, coeffs, out="s16")
        assert np.all(status == 0)
        off = 0
        for s, f in sequence:
            size = chans[s] * 1024
            got = pcm[off:off + size].reshape(chans[s], 1024)
            got_f32[s].append(got)
            assert np.array_equal(s16[off:off + size].reshape(1024, chans[s]), oracle.planar_f32_to_s16_interleaved(got).reshape(1024, chans[s]))
            off += size
    for s in range(n_streams):
        x, seqs, shapes = (np.concatenate(history[s][k]) for k in range(3))
        if len(x) == 0:
            continue
        want, _ = oracle.synthesize_stream(x, seqs, shapes)
        got = np.stack(got_f32[s])
        den = np.sqrt(np.mean(want.astype(np.float64) ** 2))
        assert np.sqrt(np.mean((got.astype(np.float64) - want) ** 2)) / den < 1.0e-6, (s, kind[s], chans[s])
    for sid in sids:
        engine.close_stream(sid)


@pytest.mark.parametrize("out", ["f32", "s16"])
def test_pairs_with_coinciding_eight_short_frames(engine, oracle, out):
    """Two channels whose EightShort frames fall on the same frame numbers share a wave (k_aac_synth_pair<.., true>: the
    long arm plus a wave-uniform eight-short arm, dsp.rs:284-338), whatever the rest of their sequences; a channel whose
    short frames sit elsewhere, or that finds no partner, runs the one-channel kernel.  Per channel both do the same
    operations in the same order: the stereo stream's L and R must equal, bit for bit, the same channels decoded as mono
    streams that cannot pair -- and both match the oracle."""
    import torch
    n = 14
    # L and R: EightShort at frames 3, 4, 9 in both, different transitions / shapes around them
    seq_l = [0, 0, 1, 2, 2, 3, 0, 0, 1, 2, 3, 0, 1, 3]
    seq_r = [0, 1, 3, 2, 2, 3, 1, 3, 1, 2, 3, 0, 0, 0]
    shapes_l = [(f // 2) & 1 for f in range(n)]
    shapes_r = [(f // 3) & 1 for f in range(n)]
    left, right = spectra(oracle, n, 0x2468, 6000.0), spectra(oracle, n, 0x1357, 9000.0)
    stereo = engine.open_stream(48000, 2)
    mono_l = engine.open_stream(48000, 1)     # n frames: its only possible partner (mono_r) has its shorts elsewhere
    mono_r = engine.open_stream(48000, 1)     # n - 1 frames: no partner of that length
    streams, chans, seqs, shapes, chunks = [], [], [], [], []
    for f in range(n):
        streams.append(stereo); chans.append(2); seqs.append([seq_l[f], seq_r[f]]); shapes.append([shapes_l[f], shapes_r[f]])
        chunks.append(np.stack([left[f], right[f]]))
        streams.append(mono_l); chans.append(1); seqs.append([seq_l[f], 0]); shapes.append([shapes_l[f], 0]); chunks.append(left[f][None])
        if f < n - 1:
            streams.append(mono_r); chans.append(1); seqs.append([seq_r[f], 0]); shapes.append([shapes_r[f], 0]); chunks.append(right[f][None])
    packed = np.concatenate([c.reshape(-1) for c in chunks]).astype(np.float32)
    descs, cnt = soundkit_amd.descs_from_arrays(streams, np.array(chans, np.uint8), np.array(seqs, np.uint8), np.array(shapes, np.uint8))
    plan = engine.plan(descs, cnt)
    d_in = torch.from_numpy(packed).cuda()
    d_out = torch.empty_like(d_in) if out == "f32" else torch.zeros(d_in.shape, dtype=torch.int16, device="cuda")
    torch.cuda.synchronize()
    (plan.run_f32 if out == "f32" else plan.run_s16_planar)(d_in, d_out)
    engine.synchronize()
    got = d_out.cpu().numpy()
    per = {stereo: [], mono_l: [], mono_r: []}
    at = 0
    for sid, ch in zip(streams, chans):
        per[sid].append(got[at:at + ch * 1024].reshape(ch, 1024))
        at += ch * 1024
    st = np.stack(per[stereo])
    assert np.array_equal(np.stack(per[mono_l])[:, 0], st[:, 0])
    assert np.array_equal(np.stack(per[mono_r])[:, 0], st[:n - 1, 1])
    ds, ss = engine.get_state(stereo, 2)
    dl, sl = engine.get_state(mono_l, 1)
    assert np.array_equal(ds[0], dl[0]) and ss[0] == sl[0]
    want, _ = oracle.synthesize_stream(np.stack([left, right], 1), np.array(list(zip(seq_l, seq_r)), np.uint8),
                                       np.array(list(zip(shapes_l, shapes_r)), np.uint8))
    if out == "f32":
        den = np.sqrt(np.mean(want.astype(np.float64) ** 2))
        assert np.sqrt(np.mean((st.astype(np.float64) - want) ** 2)) / den < 1.0e-6
        assert np.abs(st - want).max() < 2e-6 * np.abs(want).max()
    else:
        exp = np.stack([oracle.planar_f32_to_s16_interleaved(w).reshape(1024, 2).T for w in want])
        d = np.abs(st.astype(np.int32) - exp.astype(np.int32))
        assert d.max() <= 1 and (d > 0).mean() < 1e-3
    plan.destroy()
    for sid in (stereo, mono_l, mono_r):
        engine.close_stream(sid)
