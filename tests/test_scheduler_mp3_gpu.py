"""MP3 streams behind the batch scheduler, beside AAC streams: the worker's per-format dispatch (FormatDecoder::process,
soundkit-decoder/src/lib.rs:2222-2241; detect_and_init_decoder :3041-3053; decode_i16_with_drain :2150-2181) for a whole batch.
A stream's first bytes choose its decoder; the MP3 streams' granules (framing, reservoir, scale factors and Huffman codes on
the entropy threads) ride in the same tick as the AAC units (sk_tick_run_mixed) and take the same apply_output_options path.

* a 50 / 50 mix of the reference's AAC and MP3 fixtures in random chunkings, in every front-end mode: each stream delivers
  exactly what its single decoder (AacDecoder / Mp3Decoder) gives for the same bytes -- rate, channels, every sample;
* an MP3 stream resampled 16 -> 8 kHz, mono, against the CPU chain (oracle MP3 decode -> f32_to_i16 -> / 32768 -> oracle
  StreamingResampler -> downmix -> s16);
* a damaged MP3 stream loses frames or ends alone; its neighbours are untouched."""
import os
import threading

import numpy as np
import pytest

from soundkit_amd import aac, mp3, pipeline
from test_scheduler_gpu import drain, feed_all

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
AAC = ["aac/aac-stereo-48k.adts", "aac/A_Tusk_is_used_to_make_costly_gifts_encoded.aac", "aac/mono16k_A_Tusk.aac", "aac/stereo-music-44100-192k.aac"]
MP3 = ["mp3/stereo16k_A_Tusk_encoded.mp3", "mp3/mono16k_A_Tusk.mp3"]


def read(name):
    with open(os.path.join(GOLD, name), "rb") as f:
        return f.read()


def single_decoder(engine, name, data):
    """(rate, channels, all s16 samples) from the per-stream decoder handle"""
    dec = aac.AacDecoder(engine) if name.startswith("aac") else mp3.Mp3Decoder(engine=engine)
    try:
        room = np.zeros(1 << 18, np.int16)
        pcm = aac.decode_i16_with_drain(dec, data, room)
        return dec.sample_rate(), dec.channels(), np.concatenate(pcm)
    finally:
        dec.close()


@pytest.mark.parametrize("front_end", [0, 1, 2], ids=["host_front_end", "gpu_front_end", "host_huffman_gpu_rest"])
def test_mixed_aac_and_mp3_streams_equal_their_single_decoders(engine, front_end):
    names = [n for pair in zip(AAC * 2, (MP3 * 4)) for n in pair]  # AAC, MP3, AAC, MP3, ... : 16 streams, half of each codec
    datas = [read(n) for n in names]
    want = {n: single_decoder(engine, n, read(n)) for n in set(names)}
    rng = np.random.default_rng(front_end)
    chunks = [int(c) for c in rng.integers(200, 6000, len(names))]
    chunks[1], chunks[3] = 61, 100000
    sched = pipeline.BatchScheduler(engine, entropy_threads=4, max_streams=32, max_frames_per_tick=96, max_stream_frames_per_tick=5,
                                    gpu_entropy=front_end)
    try:
        handles = [sched.spawn() for _ in names]
        feeder = threading.Thread(target=feed_all, args=(handles, datas, chunks))
        feeder.start()
        outs = drain(handles, 120)
        feeder.join()
        for h in handles:
            h.cancel()
    finally:
        sched.close()
    for name, got in zip(names, outs):
        rate, channels, samples = want[name]
        assert got and not any(isinstance(a, Exception) for a in got), (name, [a for a in got if isinstance(a, Exception)][:1])
        assert all(a.sampling_rate == rate and a.channel_count == channels and a.bits_per_sample == 16 for a in got)
        unit = 576 if name.startswith("mp3") else 1024  # one AudioData per MP3 granule / AAC access unit
        assert all(a.data.size == unit * channels * 2 for a in got), name
        mine = np.concatenate([np.frombuffer(a.data.tobytes(), "<i2") for a in got])
        assert mine.size == samples.size, (name, mine.size, samples.size)
        assert np.array_equal(mine, samples), name
        assert np.abs(samples.astype(np.int32)).max() > 500


def test_an_mp3_stream_resampled_and_downmixed_meets_the_cpu_chain(engine, oracle):
    from oracle import mp3_bitstream, mp3_iso
    data = read(MP3[0])
    frames, _ = mp3_bitstream.scan(data)
    dec = mp3_bitstream.Decoder(mp3_iso.tables())
    pcm = np.concatenate([dec.frame(data, off, h) for off, h in frames])                      # [n][2] f64
    q = oracle.pcm_convert("MP3_F32_TO_I16", pcm.astype(np.float32).reshape(-1)).reshape(-1, 2)  # Mp3Decoder's i16 AudioData
    rs = oracle.StreamingResampler(16000, 8000, 2)
    outs = []
    for g in range(0, q.shape[0], 576):  # the worker resamples AudioData by AudioData
        res = rs.process(np.ascontiguousarray(q[g:g + 576].T.astype(np.float32) / np.float32(32768.0)))
        if res.shape[1]:
            outs.append(oracle.planar_f32_to_s16_interleaved(oracle.downmix_mono(res)[None]))
    tail = rs.flush()
    if tail.shape[1]:
        outs.append(oracle.planar_f32_to_s16_interleaved(oracle.downmix_mono(tail)[None]))
    want = np.concatenate([o.reshape(-1) for o in outs]).astype(np.int32)

    sched = pipeline.BatchScheduler(engine, entropy_threads=2, max_streams=8)
    try:
        handles = [sched.spawn(pipeline.DecodeOptions(16, 8000, 1)) for _ in range(3)]
        feeder = threading.Thread(target=feed_all, args=(handles, [data] * 3, [777, 4096, 100000]))
        feeder.start()
        got = drain(handles, 120)
        feeder.join()
        for h in handles:
            h.cancel()
    finally:
        sched.close()
    for outs in got:
        assert outs and not any(isinstance(a, Exception) for a in outs)
        assert all(a.sampling_rate == 8000 and a.channel_count == 1 and a.bits_per_sample == 16 for a in outs)
        mine = np.concatenate([np.frombuffer(a.data.tobytes(), "<i2") for a in outs]).astype(np.int32)
        assert mine.size == want.size, (mine.size, want.size)
        d = np.abs(mine - want)
        assert d.max() <= 1 and (d > 0).mean() < 0.01, (int(d.max()), float((d > 0).mean()))
    assert np.abs(want).max() > 300


def test_a_damaged_mp3_stream_is_alone_with_its_damage(engine):
    good, aac_data = read(MP3[0]), read(AAC[0])
    bad = bytearray(good)
    for at in range(20000, 20000 + 576 * 3, 7):  # three frames' worth of side information and main data overwritten
        bad[at] = (at * 131) & 0xff
    want_mp3 = single_decoder(engine, MP3[0], good)[2]
    want_bad = single_decoder(engine, MP3[0], bytes(bad))[2]
    want_aac = single_decoder(engine, AAC[0], aac_data)[2]
    sched = pipeline.BatchScheduler(engine, entropy_threads=3, max_streams=8, max_stream_frames_per_tick=4)
    try:
        handles = [sched.spawn() for _ in range(4)]
        datas = [good, bytes(bad), aac_data, good]
        feeder = threading.Thread(target=feed_all, args=(handles, datas, [1500, 1500, 1500, 333]))
        feeder.start()
        got = drain(handles, 120)
        feeder.join()
        for h in handles:
            h.cancel()
    finally:
        sched.close()
    joined = [np.concatenate([np.frombuffer(a.data.tobytes(), "<i2") for a in outs if not isinstance(a, Exception)]) for outs in got]
    assert np.array_equal(joined[0], want_mp3) and np.array_equal(joined[3], want_mp3) and np.array_equal(joined[2], want_aac)
    # the damaged stream: what the single decoder makes of the same bytes (frames it cannot decode are consumed without output)
    assert np.array_equal(joined[1], want_bad) and want_bad.size < want_mp3.size


def test_free_format_and_mixed_block_intensity_streams_behind_the_scheduler(engine):
    """Streams written by tests/mp3_builder.py with the standard's code books: free format (the frame length measured between headers,
    carried per stream) and joint stereo with intensity coding in every kind of granule, mixed ones included.  Each stream out of the
    scheduler equals its single decoder, and that equals the f64 chain (oracle/mp3_bitstream.py Decoder)."""
    import mp3_builder as B
    from oracle import mp3_bitstream, mp3_iso
    tables = mp3_iso.tables()
    specs = [dict(version=1, rate=44100, channels=2, mode=1, joint_modes=(0, 1, 2, 3), free_format_bytes=700),
             dict(version=2, rate=24000, channels=2, mode=1, joint_modes=(1, 3), free_format_bytes=431),
             dict(version=1, rate=48000, channels=2, mode=1, joint_modes=(1, 3), bitrate_indices=(9, 12)),
             dict(version=1, rate=32000, channels=1, free_format_bytes=2000)]
    datas = [B.build_stream(tables, 940 + k, n_frames=30, **spec)[0] for k, spec in enumerate(specs)]
    mixed_is = 0
    for data, spec in zip(datas, specs):
        state = [0]
        frames, used = mp3_bitstream.scan(data, state)
        assert len(frames) == 30 and used == len(data) and state[0] == spec.get("free_format_bytes", 0)
        for off, h in frames:
            side = mp3_bitstream.parse_side_info(data[off:off + h["frame_bytes"]], h)
            mixed_is += sum(1 for gr in side["gr"] if h["mode"] == 1 and h["mode_ext"] & 1 and gr[0]["mixed_block_flag"])
    assert mixed_is > 3, "the streams hold intensity-coded mixed granules"
    want = []
    for data, spec in zip(datas, specs):
        rate, channels, samples = single_decoder(engine, "mp3", data)
        assert (rate, channels) == (spec["rate"], spec["channels"])
        dec = mp3_bitstream.Decoder(tables)
        frames, _ = mp3_bitstream.scan(data, [0])
        pcm = np.concatenate([dec.frame(data, off, h) for off, h in frames]).reshape(-1)
        assert samples.size == pcm.size
        q = np.clip(np.sign(pcm) * np.floor(np.abs(pcm) * 32767.0 + 0.5), -32768, 32767)
        d = np.abs(samples.astype(np.int64) - q.astype(np.int64))
        assert d.max() <= 1 and (d > 0).mean() < 0.01, (int(d.max()), float((d > 0).mean()))
        assert np.abs(samples.astype(np.int32)).max() > 500
        want.append(samples)
    sched = pipeline.BatchScheduler(engine, entropy_threads=3, max_streams=8, max_stream_frames_per_tick=6)
    try:
        handles = [sched.spawn() for _ in datas]
        feeder = threading.Thread(target=feed_all, args=(handles, datas, [700, 333, 4096, 1999]))
        feeder.start()
        got = drain(handles, 120)
        feeder.join()
        for h in handles:
            h.cancel()
    finally:
        sched.close()
    for outs, samples, spec in zip(got, want, specs):
        assert outs and not any(isinstance(a, Exception) for a in outs)
        assert all(a.sampling_rate == spec["rate"] and a.channel_count == spec["channels"] for a in outs)
        mine = np.concatenate([np.frombuffer(a.data.tobytes(), "<i2") for a in outs])
        assert np.array_equal(mine, samples)


@pytest.mark.parametrize("front_end", [0, 1, 2], ids=["host_front_end", "gpu_front_end", "host_huffman_gpu_rest"])
def test_two_lanes_and_the_default_quota_give_every_stream_its_single_decoder_output(engine, front_end):
    """The configuration a GPU-front-end pipeline of 4096 streams gets by default since round 4: the streams dealt out over TWO
    lanes (engines on the same device, each with its own batches, submission and delivery threads) behind one handle space, 32
    units per stream and tick.  AAC and MP3 streams, resampled and not, in random chunkings: every stream delivers exactly what
    its single decoder gives -- which lane it lands on, and what shares its ticks, must not show."""
    names = [n for pair in zip(AAC * 2, (MP3 * 4)) for n in pair] + AAC  # 20 streams: handles alternate between the lanes
    datas = [read(n) for n in names]
    want = {n: single_decoder(engine, n, read(n)) for n in set(names)}
    rng = np.random.default_rng(40 + front_end)
    chunks = [int(c) for c in rng.integers(150, 9000, len(names))]
    sched = pipeline.BatchScheduler(engine, entropy_threads=4, max_streams=32, max_frames_per_tick=256, max_stream_frames_per_tick=32,
                                    gpu_entropy=front_end, lanes=2)
    try:
        handles = [sched.spawn() for _ in names]
        assert len({h.id % 2 for h in handles}) == 2, "handles of both lanes"
        feeder = threading.Thread(target=feed_all, args=(handles, datas, chunks))
        feeder.start()
        outs = drain(handles, 120)
        feeder.join()
        # a second generation on the same handles' slots: a lane's streams are closed and opened again while the other lane runs
        for h in handles[:6]:
            h.cancel()
        again = [sched.spawn(pipeline.DecodeOptions(16, 8000, 1)) for _ in range(6)]
        feeder = threading.Thread(target=feed_all, args=(again, datas[:6], chunks[:6]))
        feeder.start()
        outs2 = drain(again, 120)
        feeder.join()
        for h in handles[6:] + again:
            h.cancel()
    finally:
        sched.close()
    for name, got in zip(names, outs):
        rate, channels, samples = want[name]
        assert got and not any(isinstance(a, Exception) for a in got), (name, [a for a in got if isinstance(a, Exception)][:1])
        assert all(a.sampling_rate == rate and a.channel_count == channels and a.bits_per_sample == 16 for a in got)
        mine = np.concatenate([np.frombuffer(a.data.tobytes(), "<i2") for a in got])
        assert mine.size == samples.size and np.array_equal(mine, samples), name
    for name, got in zip(names[:6], outs2):
        assert got and not any(isinstance(a, Exception) for a in got), name
        assert all(a.sampling_rate == 8000 and a.channel_count == 1 and a.bits_per_sample == 16 for a in got)
        rate, _, samples = want[name]
        total = sum(a.data.size // 2 for a in got)
        expect = samples.size // want[name][1] * 8000 / rate
        assert abs(total - expect) < 2200, (name, total, expect)  # the resampler's delay line and the flush's trim
    by_name = {}
    for name, got in zip(names[:6], outs2):  # streams of one file, whichever lane: the same bytes
        blob = b"".join(a.data.tobytes() for a in got)
        assert by_name.setdefault(name, blob) == blob, name
