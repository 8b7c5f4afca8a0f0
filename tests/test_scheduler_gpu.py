"""The batch scheduler (csrc/pipeline.cpp) behind the reference's pipeline handle (soundkit-decoder lib.rs:2788-2889):
chunk-size invariance (as soundkit-decoder lib.rs:5339-5378 tests for MP3), equality with the one-stream-at-a-time
path, end-of-stream flush, backpressure and per-stream error isolation."""
import os
import time

import numpy as np
import pytest

from soundkit_amd import aac_lc, pipeline
from test_tick_gpu import one_at_a_time, parsed

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "aac")


@pytest.fixture(scope="module", params=["host_front_end", "gpu_front_end", "gpu_front_end_two_lanes", "host_huffman_gpu_rest"])
def sched(engine, request):
    """Every scenario runs four times: entropy decode on host threads (sk_tick_run), on the GPU (sk_tick_run_au), on the GPU
    with the streams spread over two engines behind the one handle space (sk_pipeline_config.lanes), and split -- Huffman
    decode on host threads, dequantisation / PNS / stereo tools / TNS on the GPU (sk_tick_run_q, gpu_entropy = 2)."""
    s = pipeline.BatchScheduler(engine, entropy_threads=4, max_streams=64, max_frames_per_tick=256, max_stream_frames_per_tick=4,
                                gpu_entropy={"host_front_end": 0, "host_huffman_gpu_rest": 2}.get(request.param, 1),
                                lanes=2 if request.param.endswith("two_lanes") else 1)
    yield s
    s.close()


def drain(handles, deadline_s=60):
    outs = [[] for _ in handles]
    t0 = time.time()
    live = set(range(len(handles)))
    while live:
        assert time.time() - t0 < deadline_s, "scheduler stalled"
        for i in list(live):
            got = handles[i].try_recv()
            if got is not None:
                outs[i].append(got)
            elif handles[i].ended():
                live.discard(i)
        time.sleep(0.0005)
    return outs


def feed_all(handles, datas, chunk_sizes):
    pos = [0] * len(handles)
    pending = set(range(len(handles)))
    while pending:
        for i in list(pending):
            if pos[i] >= len(datas[i]):
                try:
                    handles[i].finish()
                    pending.discard(i)
                except pipeline.DecodeError as e:
                    assert e.kind == "InputBufferFull"
                continue
            piece = datas[i][pos[i]:pos[i] + chunk_sizes[i]]
            try:
                handles[i].send(piece)
                pos[i] += len(piece)
            except pipeline.DecodeError as e:
                assert e.kind == "InputBufferFull"
        time.sleep(0.0002)


def as_tuples(outs):
    return [(a.bits_per_sample, a.channel_count, a.sampling_rate, a.data.tobytes()) for a in outs]


@pytest.mark.parametrize("bits,rate,ch", [(None, None, None), (16, 16000, 1), (24, None, 1), (32, 8000, None)])
def test_many_streams_match_the_single_stream_path(engine, sched, bits, rate, ch):
    names = ["aac-stereo-48k.adts", "mono16k_A_Tusk.aac", "stereo-music-44100-192k.aac", "A_Tusk_is_used_to_make_costly_gifts_encoded.aac"]
    datas = [open(os.path.join(GOLD, n), "rb").read() for n in names] * 3
    chunk_sizes = [4096, 333, 1, 100000, 777, 64, 5000, 17, 2048, 1500, 9, 40000]
    chunk_sizes[2] = 211  # one-byte chunks of a 90 KB file would only test the input bound
    opts = pipeline.DecodeOptions(bits, rate, ch)
    handles = [sched.spawn(opts) for _ in datas]
    import threading
    feeder = threading.Thread(target=feed_all, args=(handles, datas, chunk_sizes))
    feeder.start()
    outs = drain(handles)
    feeder.join()
    want = {}
    for i, name in enumerate(names):
        srate, sch, frames = parsed(name)
        o_rate = rate if rate != srate else None
        exp = one_at_a_time(engine, srate, sch, frames, bits, o_rate, ch)
        want[i] = [(b, c, o_rate or srate, d) for b, c, d in exp]
    for i, got in enumerate(outs):
        assert all(not isinstance(g, pipeline.DecodeError) for g in got), [str(g) for g in got if isinstance(g, Exception)]
        assert as_tuples(got) == want[i % len(names)], (i, len(got), len(want[i % len(names)]))
    for h in handles:
        h.cancel()


def test_backpressure_and_limits(engine, sched):
    data = open(os.path.join(GOLD, "aac-stereo-48k.adts"), "rb").read()
    h = sched.spawn()
    with pytest.raises(pipeline.DecodeError) as exc:
        h.send(b"\0" * (4 * 1024 * 1024 + 1))
    assert exc.value.kind == "InputChunkTooLarge"
    # nobody takes the outputs: at most 16 AudioData pile up, then the stream stops being scheduled and the input
    # queue fills (128 chunks): send reports InputBufferFull instead of blocking
    frames = aac_lc.split_adts(data)
    sent = full = 0
    for rep in range(40):
        for k in range(len(frames)):
            piece = data[sum(len(a) + 7 for _, a in frames[:k]):sum(len(a) + 7 for _, a in frames[:k + 1])]
            try:
                h.send(piece)
                sent += 1
            except pipeline.DecodeError as e:
                assert e.kind == "InputBufferFull"
                full += 1
    assert full > 0 and sent <= 128 + 16 + 8 + 8
    time.sleep(0.2)
    assert 0 < h.queued_input_bytes() <= 8 * 1024 * 1024
    got = 0
    while h.try_recv() is not None:
        got += 1
    assert got == 16
    # draining un-blocks it
    t0 = time.time()
    while got < sent and time.time() - t0 < 30:
        if h.try_recv() is not None:
            got += 1
    assert got == sent
    h.finish()
    t0 = time.time()
    while not h.ended() and time.time() - t0 < 10:
        time.sleep(0.001)
    assert h.ended()
    with pytest.raises(pipeline.DecodeError) as exc:
        h.send(b"abc")
    assert exc.value.kind == "PipelineClosed"
    h.cancel()


def test_error_ends_only_that_stream(engine, sched):
    data = open(os.path.join(GOLD, "aac-stereo-48k.adts"), "rb").read()
    frames = aac_lc.split_adts(data)
    cut = sum(len(a) + 7 for _, a in frames[:10])
    bad = bytearray(data)
    # corrupt the payload of frame 10 beyond repair: max_sfb / section data of the first channel
    for k in range(cut + 7, cut + 7 + 24):
        bad[k] = 0xFF
    good_h, bad_h = sched.spawn(), sched.spawn()
    for h, d in ((good_h, data), (bad_h, bytes(bad))):
        h.send(d)
        h.finish()
    good, broken = drain([good_h, bad_h])
    assert len(good) == 48 and not any(isinstance(g, Exception) for g in good)
    assert len(broken) == 11 and isinstance(broken[-1], pipeline.DecodeError) and broken[-1].kind == "DecodingFailed"
    assert as_tuples(broken[:10]) == as_tuples(good[:10])
    assert "Decoding failed" in str(broken[-1])
    good_h.cancel(), bad_h.cancel()


def test_cancel_frees_handles(engine):
    s = pipeline.BatchScheduler(engine, entropy_threads=2, max_streams=4)
    data = open(os.path.join(GOLD, "aac-stereo-48k.adts"), "rb").read()
    for round_ in range(5):
        hs = [s.spawn() for _ in range(4)]
        with pytest.raises(Exception):
            s.spawn()
        for h in hs:
            h.send(data)
        for h in hs:
            h.cancel()  # some of them with frames in flight
        time.sleep(0.05)
    st = s.stats()
    assert st["errors"] == 0
    s.close()


def test_wait_outputs_serves_many_handles_from_one_thread(engine, sched):
    """sk_pipeline_wait_outputs: block until some handle has news instead of polling every handle."""
    data = open(os.path.join(GOLD, "aac-stereo-48k.adts"), "rb").read()
    handles = {h.id: h for h in (sched.spawn(pipeline.DecodeOptions(16, 16000, 1)) for _ in range(10))}
    for h in handles.values():
        h.send(data)
        h.finish()
    got = {i: [] for i in handles}
    live = set(handles)
    t0 = time.time()
    while live:
        assert time.time() - t0 < 60
        for i in sched.wait_outputs(timeout_ms=200):
            if i not in handles:
                continue
            while True:
                a = handles[i].try_recv()
                if a is None:
                    break
                got[i].append(a.data.tobytes())
            if handles[i].ended():
                live.discard(i)
    first = got[next(iter(got))]
    assert len(first) == 13 and all(v == first for v in got.values())  # 12 full chunks + the flushed tail
    for h in handles.values():
        h.cancel()
