"""BASELINE config 5's stream count under the driver: 8192 concurrent ADTS AAC-LC streams through the batch scheduler
(sk_pipeline_*), 96 access units each (the reference's 48 kHz stereo TS sample, looped as one continuous stream),
48 kHz stereo -> 16 kHz mono s16 -- once with the entropy front-end on host threads, once on the GPU, once split (host
Huffman decode, the rest of the front-end on the GPU: the quantised hand-over).

Every stream gets the same bytes, so every stream must deliver the same AudioData sequence: the harness
(csrc/load_gen.cpp, sk_loadgen_run_checked) keeps per stream an order-sensitive FNV-1a over (frames, channels, bits,
rate, bytes) of each output, the output count, the byte count and the error count.  Checked on all 8192 streams:
no error, the same count / bytes / hash everywhere.  Three streams (first, middle, last) are captured whole and compared
with the CPU chain: oracle front-end -> oracle synthesis -> float_sample_to_i16 -> / 32768 -> oracle streaming
resampler -> mono downmix -> s16 (the reference worker's order, soundkit-decoder lib.rs:1793-1813, 3324-3456)."""
import ctypes as C
import os

import numpy as np
import pytest

from soundkit_amd import pipeline
from soundkit_amd._lib import DecodeOptionsC

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CLIP = os.path.join(ROOT, "tests", "golden", "aac", "aac-stereo-48k.adts")
STREAMS, LOOPS = 8192, 2


class Result(C.Structure):
    _fields_ = [("seconds", C.c_double)] + [(n, C.c_uint64) for n in ("access_units", "outputs", "pcm_frames", "pcm_bytes", "errors", "input_full")]


class Check(C.Structure):
    _fields_ = [("hash", C.c_void_p), ("outputs", C.c_void_p), ("bytes", C.c_void_p), ("errors", C.c_void_p), ("capture", C.c_void_p),
                ("n_capture", C.c_uint32), ("capture_buf", C.c_void_p), ("capture_cap", C.c_size_t), ("capture_len", C.c_void_p)]


@pytest.fixture(scope="module")
def expected(oracle):
    """the CPU chain on LOOPS passes of the clip: list of s16 mono arrays, one per AudioData the worker would send"""
    from oracle import aac_frontend as OF
    frames = OF.split_adts(open(CLIP, "rb").read())
    dec = OF.Decoder(frames[0][0])
    chans = [oracle.Channel() for _ in range(dec.channels)]
    rs = oracle.StreamingResampler(dec.sample_rate, 16000, dec.channels)
    want = []
    for _ in range(LOOPS):
        for _, au in frames:
            coeffs, seqs, shapes = dec.decode_access_unit(au)
            pcm, _ = oracle.synthesize_stream(coeffs[None], [seqs], [shapes], chans)
            q = oracle.planar_f32_to_s16_interleaved(pcm[0]).reshape(1024, dec.channels).T.astype(np.float32) / np.float32(32768.0)
            res = rs.process(q)
            if res.shape[1]:
                want.append(oracle.planar_f32_to_s16_interleaved(oracle.downmix_mono(res)[None]))
    tail = rs.flush()
    if tail.shape[1]:
        want.append(oracle.planar_f32_to_s16_interleaved(oracle.downmix_mono(tail)[None]))
    return len(frames), want


@pytest.mark.parametrize("gpu_entropy", [0, 1, 2], ids=["host_front_end", "gpu_front_end", "host_huffman_gpu_rest"])
def test_8192_streams_through_the_scheduler(engine, expected, gpu_entropy):
    units, want = expected
    from soundkit_amd import aac_lc
    clip = open(CLIP, "rb").read()
    clip = clip[:sum(len(au) + 7 for _, au in aac_lc.split_adts(clip))]   # whole ADTS frames only: the clip is looped
    lg = C.CDLL(os.path.join(ROOT, "soundkit_amd", "libsk_loadgen.so"))
    lg.sk_loadgen_run_checked.restype = C.c_int
    lg.sk_loadgen_run_checked.argtypes = [C.c_void_p, C.c_char_p, C.c_size_t, C.c_uint32, C.c_uint32, C.c_uint32, C.c_void_p,
                                          C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p]
    hashes, outputs = np.zeros(STREAMS, np.uint64), np.zeros(STREAMS, np.uint32)
    nbytes, errors = np.zeros(STREAMS, np.uint64), np.zeros(STREAMS, np.uint32)
    capture = np.array([0, STREAMS // 2 + 1, STREAMS - 1], np.uint32)
    cap = 1 << 20
    buf, lens = np.zeros((capture.size, cap), np.uint8), np.zeros(capture.size, np.uint64)
    chk = Check(hashes.ctypes.data, outputs.ctypes.data, nbytes.ctypes.data, errors.ctypes.data, capture.ctypes.data, capture.size,
                buf.ctypes.data, cap, lens.ctypes.data)
    sched = pipeline.BatchScheduler(engine, max_streams=STREAMS, gpu_entropy=gpu_entropy, lanes=1)
    try:
        res = Result()
        opt = DecodeOptionsC(16000, 16, 1, 0)
        rc = lg.sk_loadgen_run_checked(sched._h, clip, len(clip), units, STREAMS, LOOPS, C.byref(opt), 6, 0, C.byref(res), C.byref(chk))
        assert rc == 0, "load generator: %d (-8 = its progress deadline: the scheduler's state is on stderr)" % rc
    finally:
        sched.close()
    # every stream: no error, and exactly what the CPU chain sends -- count, bytes, order (the hash is order-sensitive)
    assert res.errors == 0 and not errors.any()
    assert res.access_units == STREAMS * LOOPS * units
    total = sum(w.size for w in want)
    assert (outputs == len(want)).all(), (np.unique(outputs), len(want))
    assert (nbytes == 2 * total).all(), (np.unique(nbytes), 2 * total)
    assert (hashes == hashes[0]).all(), "%d streams delivered something else than stream 0" % int((hashes != hashes[0]).sum())
    assert res.outputs == STREAMS * len(want) and res.pcm_bytes == STREAMS * 2 * total
    # three streams against the oracle chain, sample by sample (the FIR sums in another order: +-1 LSB on < 1 %)
    exp = np.concatenate(want).astype(np.int32)
    for k in range(capture.size):
        assert int(lens[k]) == 2 * total
        mine = np.frombuffer(buf[k, :int(lens[k])].tobytes(), "<i2").astype(np.int32)
        d = np.abs(mine - exp)
        assert d.max() <= 1 and (d > 0).mean() < 0.01, (k, int(d.max()), float((d > 0).mean()))


@pytest.mark.parametrize("lanes, quota, front_end, loops", [(2, 32, 1, 60), (1, 32, 1, 60), (2, 8, 0, 30), (2, 8, 2, 30)],
                         ids=["two_lanes_gpu_front_end", "one_lane_gpu_front_end", "two_lanes_host_front_end", "two_lanes_split_front_end"])
def test_every_stream_delivers_the_same_bytes_over_many_ticks(engine, lanes, quota, front_end, loops):
    """2048 streams x 2880 (1440) access units each, 48 kHz stereo -> 16 kHz mono s16: every stream is fed the same
    bytes, so every stream's order-sensitive hash over its AudioData must be the same -- over thousands of chunks per stream, with
    the ticks' contents changing from tick to tick.  With two lanes this is the regression test of round 4's finding: two engines
    whose ticks ran on the device at the same time delivered short bursts of slightly wrong samples in a third of the streams
    (tools/debug/stream_hashes.py, profiles/r04_lanes_corruption.md); the engines of a device take turns since."""
    from soundkit_amd import aac_lc
    streams = 2048
    clip = open(CLIP, "rb").read()
    frames = aac_lc.split_adts(clip)
    clip = clip[:sum(len(au) + 7 for _, au in frames)]
    lg = C.CDLL(os.path.join(ROOT, "soundkit_amd", "libsk_loadgen.so"))
    lg.sk_loadgen_run_checked.restype = C.c_int
    lg.sk_loadgen_run_checked.argtypes = [C.c_void_p, C.c_char_p, C.c_size_t, C.c_uint32, C.c_uint32, C.c_uint32, C.c_void_p,
                                          C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p]
    hashes, outputs = np.zeros(streams, np.uint64), np.zeros(streams, np.uint32)
    nbytes, errors = np.zeros(streams, np.uint64), np.zeros(streams, np.uint32)
    capture = np.array([0], np.uint32)
    buf, lens = np.zeros((1, 1 << 22), np.uint8), np.zeros(1, np.uint64)
    chk = Check(hashes.ctypes.data, outputs.ctypes.data, nbytes.ctypes.data, errors.ctypes.data, capture.ctypes.data, 1, buf.ctypes.data,
                buf.shape[1], lens.ctypes.data)
    sched = pipeline.BatchScheduler(engine, max_streams=streams, gpu_entropy=front_end, lanes=lanes, max_stream_frames_per_tick=quota)
    try:
        res = Result()
        opt = DecodeOptionsC(16000, 16, 1, 0)
        rc = lg.sk_loadgen_run_checked(sched._h, clip, len(clip), len(frames), streams, loops, C.byref(opt), 4, 0, C.byref(res), C.byref(chk))
        assert rc == 0
    finally:
        sched.close()
    assert res.errors == 0 and not errors.any()
    assert res.access_units == streams * loops * len(frames)
    assert np.unique(outputs).size == 1 and np.unique(nbytes).size == 1
    odd = np.flatnonzero(hashes != hashes[0])
    assert odd.size == 0, "%d of %d streams delivered other bytes than stream 0 (first: %s)" % (odd.size, streams, odd[:8].tolist())


def test_two_engines_called_directly_from_two_threads_take_turns():
    """Two engines of one process on one device, driven through the plain entry points (no scheduler): one thread keeps the 48 -> 16 kHz
    FIR on f32 rows in flight (matrix instructions), the other runs a synthesis plan on a fixed input again and again.  On this platform
    a synthesis launch that shares CUs with those matrix instructions comes out wrong in 29 of 30 launches
    (profiles/r04_lanes_corruption.md); while a process has several engines on a device, the compute entry points take the device's turn
    and wait for their work before they give it back, so every synthesis result must be the first one's, bit for bit."""
    import threading

    import torch

    import soundkit_amd
    dev = torch.device("cuda:0")
    streams, frames = 2048, 16
    eng_a, eng_b = soundkit_amd.Engine(0, 64), soundkit_amd.Engine(0, streams + 8)
    try:
        g = torch.Generator(device="cpu").manual_seed(21)
        rows = 4096
        x = (torch.rand((rows, 48000), generator=g) * 2 - 1).to(dev)
        n16 = eng_a.downsample_out_frames(48000)
        y = torch.empty((rows, n16), device=dev)
        coeffs = ((torch.rand((streams * frames, 2, 1024), generator=g) * 24) - 12).to(dev)
        sids = np.array([eng_b.open_stream(48000, 2) for _ in range(streams)], np.uint32)
        shapes = np.tile((np.arange(frames) & 1).astype(np.uint8), streams)[:, None].repeat(2, 1)
        descs, n = soundkit_amd.descs_from_arrays(np.repeat(sids, frames), 2, np.zeros((streams * frames, 2), np.uint8), shapes)
        plan = eng_b.plan(descs, n)
        torch.cuda.synchronize()
        stop = threading.Event()
        fir_calls = [0]

        def aggressor():
            while not stop.is_set():
                eng_a.downsample_48k_16k_dev(x, 48000, rows, 48000, y, n16)
                fir_calls[0] += 1
            eng_a.synchronize()

        th = threading.Thread(target=aggressor)
        th.start()
        try:
            pcm = torch.empty_like(coeffs)
            first, deviating = None, 0
            for it in range(40):
                for sid in sids:
                    eng_b.reset_stream(int(sid))
                plan.run_f32(coeffs, pcm)
                eng_b.synchronize()
                got = pcm.view(torch.int32).clone()
                if first is None:
                    first = got
                elif not torch.equal(got, first):
                    deviating += 1
        finally:
            stop.set()
            th.join()
        assert fir_calls[0] > 40  # the other engine was at work throughout
        assert deviating == 0, deviating
        plan.destroy()
    finally:
        eng_a.close()
        eng_b.close()
