"""sk_tick_run: a batch of access units through decode_aac_access_unit + apply_output_options in one launch
sequence (soundkit-decoder lib.rs:1793-1813, 3324-3456), checked against the same steps taken one call at a time
through the mirrors (bit-exact: same kernels, different batching) and against the oracle."""
import os

import numpy as np
import pytest

from soundkit_amd import aac_lc, decoder
from soundkit_amd.audio_types import AudioData
from soundkit_amd.engine import make_descs

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "aac")


def parsed(name):
    frames = aac_lc.split_adts(open(os.path.join(GOLD, name), "rb").read())
    fe = aac_lc.AacLcFrontEnd(frames[0][0])
    return fe.sample_rate, fe.channels, [fe.parse(au) for _, au in frames]


def one_at_a_time(engine, rate, ch, frames, bits, out_rate, out_ch):
    """The reference worker's order of operations, one access unit per call."""
    dec = aac_lc.AacLcSynth(rate, ch, engine)
    rs, out = None, []
    for coeffs, seqs, shapes in frames:
        s16 = dec.synthesize_s16(coeffs, seqs, shapes)
        audio = AudioData(16, ch, rate, s16.view(np.uint8))
        got, rs = decoder.apply_output_options(audio, bits, out_rate, out_ch, rs)
        out += [(a.bits_per_sample, a.channel_count, a.data.tobytes()) for a in got]
    if rs is not None:
        out += [(a.bits_per_sample, a.channel_count, a.data.tobytes())
                for a in decoder.flush_resampler_frames(rs, bits or 16, out_ch or ch)]
        rs.close()
    dec.close()
    return out


def run_ticks(engine, specs, frames_per_tick):
    """specs: list of (rate, ch, frames, bits, out_rate, out_ch).  Streams advance together, frames_per_tick[i]
    access units of stream i per tick; returns the outputs per stream."""
    sids, outs, pos = [], [[] for _ in specs], [0] * len(specs)
    for rate, ch, frames, bits, out_rate, out_ch in specs:
        sid = engine.open_stream(rate, ch)
        if out_rate and out_rate != rate:
            engine.resampler_open(sid, rate, out_rate)
        sids.append(sid)
    done = [False] * len(specs)
    while not all(done):
        table, descs_in, coeffs = [], [], []
        for i, (rate, ch, frames, bits, out_rate, out_ch) in enumerate(specs):
            if done[i]:
                continue
            take = frames[pos[i]:pos[i] + frames_per_tick[i]]
            pos[i] += len(take)
            last = pos[i] >= len(frames)
            resample = bool(out_rate and out_rate != rate)
            table.append({"stream": sids[i], "n_frames": len(take), "out_bits": bits or 16, "out_channels": out_ch or ch,
                          "resample": resample, "flush": last and resample, "index": i})
            for c, seqs, shapes in take:
                descs_in.append((sids[i], ch, list(seqs) + [0] * (2 - ch), list(shapes) + [0] * (2 - ch)))
                coeffs.append(c.ravel())
            done[i] = last
        descs, n = make_descs(descs_in)
        flat = np.concatenate(coeffs) if coeffs else np.zeros(0, np.float32)
        for idx, status, nframes, ch_o, bits_o, data in engine.tick_run(table, descs, n, flat):
            assert status == 0
            outs[table[idx]["index"]].append((bits_o, ch_o, data))
    for sid in sids:
        engine.close_stream(sid)
    return outs


OPTIONS = [(None, None, None), (None, None, 1), (24, None, None), (32, None, 1), (16, 16000, None), (16, 16000, 1),
           (24, 16000, 1), (None, 8000, None), (32, 44100, None)]


@pytest.mark.parametrize("bits,out_rate,out_ch", OPTIONS)
def test_tick_matches_one_call_at_a_time(engine, bits, out_rate, out_ch):
    """48 kHz stereo fixture (the TS sample) and the 16 kHz mono one side by side in the same ticks, uneven frame
    counts per tick; every AudioData must equal the one-access-unit-at-a-time result byte for byte."""
    a = parsed("aac-stereo-48k.adts")
    b = parsed("mono16k_A_Tusk.aac")
    specs = []
    for rate, ch, frames in (a, b):
        o_rate = out_rate if out_rate != rate else None
        specs.append((rate, ch, frames, bits, o_rate, out_ch))
    got = run_ticks(engine, specs, [5, 3])
    for spec, mine in zip(specs, got):
        want = one_at_a_time(engine, *spec)
        assert len(mine) == len(want)
        for (b1, c1, d1), (b2, c2, d2) in zip(mine, want):
            assert (b1, c1) == (b2, c2) and d1 == d2


def test_tick_against_oracle_with_resampling(engine, oracle):
    """Same chain restated on the CPU: oracle synthesis -> float_sample_to_i16 -> /32768 -> oracle streaming
    resampler -> mono downmix -> s16.  The FIR sums in another order than the oracle, so s16 may differ by one LSB
    on a small fraction of samples."""
    rate, ch, frames = parsed("aac-stereo-48k.adts")
    got = run_ticks(engine, [(rate, ch, frames, 16, 16000, 1)], [7])[0]
    chans = [oracle.Channel() for _ in range(ch)]
    rs = oracle.StreamingResampler(rate, 16000, ch)
    want = []
    for coeffs, seqs, shapes in frames:
        pcm, _ = oracle.synthesize_stream(coeffs[None], [seqs], [shapes], chans)
        q = oracle.planar_f32_to_s16_interleaved(pcm[0]).reshape(1024, ch).T.astype(np.float32) / np.float32(32768.0)
        res = rs.process(q)
        if res.shape[1]:
            want.append(res)
    tail = rs.flush()
    if tail.shape[1]:
        want.append(tail)
    assert len(got) == len(want)
    worst, differing, total = 0, 0, 0
    for (bits_o, ch_o, data), w in zip(got, want):
        mono = oracle.downmix_mono(w)
        exp = oracle.planar_f32_to_s16_interleaved(mono[None])
        mine = np.frombuffer(data, "<i2")
        assert (bits_o, ch_o) == (16, 1) and mine.size == exp.size
        d = np.abs(mine.astype(np.int32) - exp.astype(np.int32))
        worst, differing, total = max(worst, int(d.max())), differing + int((d > 0).sum()), total + d.size
    assert worst <= 1 and differing <= 0.01 * total, (worst, differing, total)


def test_tick_rejects_bad_tables(engine):
    from soundkit_amd._lib import SoundkitError
    rate, ch, frames = parsed("aac-stereo-48k.adts")
    sid = engine.open_stream(rate, ch)
    c, seqs, shapes = frames[0]
    descs, n = make_descs([(sid, ch, list(seqs), list(shapes))])
    ok = {"stream": sid, "n_frames": 1, "out_bits": 16, "out_channels": 2}
    for bad in ({"out_bits": 20}, {"out_channels": 0}, {"n_frames": 2}, {"resample": 1}, {"stream": 8191}):
        with pytest.raises(SoundkitError):
            engine.tick_run([dict(ok, **bad)], descs, n, c.ravel())
    with pytest.raises(SoundkitError):
        engine.tick_run([ok, ok], descs, n, c.ravel())
    # a frame the engine rejects ends that stream only: status record, other stream untouched
    other = engine.open_stream(rate, ch)
    descs, n = make_descs([(sid, ch, [7, 0], list(shapes)), (other, ch, list(seqs), list(shapes))])
    res = engine.tick_run([ok, dict(ok, stream=other)], descs, n, np.concatenate([c.ravel(), c.ravel()]))
    assert [(r[0], r[1] != 0, r[2]) for r in res] == [(0, True, 0), (1, False, 1024)]
    engine.close_stream(sid), engine.close_stream(other)


def test_a_failed_launch_does_not_run_the_bookkeeping_ahead(engine):
    """A HIP call that fails part-way through a tick (injected: sk_engine_debug_fail_after makes the n-th call from now
    report a launch failure, n = 1, 2, ... until a tick gets through): the tick returns SK_ERR_HIP with the stage in the
    engine's error text, the engine is idle again, and the streaming resampler's fill -- advanced on the host while the
    appends are queued -- is put back, so the stream is exactly two access units into its first 4096-frame chunk after the
    first tick that succeeds however many failed before it: no output then, one chunk's output after two more units."""
    from soundkit_amd._lib import SoundkitError
    rate, ch, frames = parsed("aac-stereo-48k.adts")
    sid = engine.open_stream(rate, ch)
    engine.resampler_open(sid, rate, 16000)
    table = [{"stream": sid, "n_frames": 2, "out_bits": 16, "out_channels": 1, "resample": True}]

    def tick(first):
        take = frames[first:first + 2]
        descs, n = make_descs([(sid, ch, list(seqs), list(shapes)) for _, seqs, shapes in take])
        return engine.tick_run(table, descs, n, np.concatenate([c.ravel() for c, _, _ in take]))
    try:
        failures, res = 0, None
        for n in range(1, 200):
            engine.debug_fail_after(n)
            try:
                res = tick(0)
            except SoundkitError as e:
                assert e.status == -3, str(e)
                assert engine.where() == "idle"
                failures += 1
                continue
            break
        engine.debug_fail_after(0)
        assert failures >= 8 and res is not None, failures      # the tick is at least that many HIP calls long
        assert res == []                                         # 2048 frames into the chunk: nothing to send yet
        res = tick(2)
        assert len(res) == 1 and res[0][1] == 0 and 1300 <= res[0][2] <= 1400, [(r[1], r[2]) for r in res]
    finally:
        engine.debug_fail_after(0)
        engine.close_stream(sid)


def test_a_tick_that_outlasts_the_wait_bound_is_an_error_not_a_parked_thread(engine):
    """sk_engine_set_wait_bound: with a bound of 100 ns a tick of 256 streams x 8 access units cannot finish in time -- the
    call returns SK_ERR_TIMEOUT (-8) with the stage in the engine's error text instead of blocking, the engine reports idle,
    and with the bound back at its default the engine works: fresh streams decode to what fresh streams decoded before."""
    from soundkit_amd._lib import SoundkitError
    rate, ch, frames = parsed("aac-stereo-48k.adts")
    n_streams, per = 256, 8

    def run(sids):
        table = [{"stream": sid, "n_frames": per, "out_bits": 16, "out_channels": 1, "resample": True} for sid in sids]
        descs, n = make_descs([(sid, ch, list(seqs), list(shapes)) for sid in sids for _, seqs, shapes in frames[:per]])
        coeffs = np.concatenate([c.ravel() for _ in sids for c, _, _ in frames[:per]])
        return engine.tick_run(table, descs, n, coeffs)

    def fresh():
        sids = [engine.open_stream(rate, ch) for _ in range(n_streams)]
        for sid in sids:
            engine.resampler_open(sid, rate, 16000)
        return sids
    first = fresh()
    want = run(first)
    assert len(want) == 2 * n_streams and all(r[1] == 0 for r in want)
    for sid in first:
        engine.close_stream(sid)
    second = fresh()
    engine.set_wait_bound(1e-7)
    try:
        with pytest.raises(SoundkitError) as exc:
            run(second)
        assert exc.value.status == -8 and "did not finish" in str(exc.value), str(exc.value)
        assert engine.where() == "idle"
    finally:
        engine.set_wait_bound(120.0)
    engine.synchronize()            # whatever that tick had queued is done now; its streams are in no defined state: drop them
    for sid in second:
        engine.close_stream(sid)
    third = fresh()
    got = run(third)
    assert [(r[1], r[2], r[5]) for r in got] == [(r[1], r[2], r[5]) for r in want]
    for sid in third:
        engine.close_stream(sid)
