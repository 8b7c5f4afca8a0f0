"""Two engines on one device, each running the same tick (sk_tick_run: host spectra in, synthesis, streaming 48 -> 16 kHz, mono s16
out) from its own thread on fresh streams: does a tick ever hand back other bytes than the first one did?
   python tools/debug/two_engines_tick.py [threads] [iterations] [streams] [units per stream] [resample 0/1]"""
import hashlib
import sys
import threading

import numpy as np

sys.path.insert(0, ".")
import soundkit_amd  # noqa: E402

n_threads = int(sys.argv[1]) if len(sys.argv) > 1 else 2
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 100
streams = int(sys.argv[3]) if len(sys.argv) > 3 else 256
units = int(sys.argv[4]) if len(sys.argv) > 4 else 16
resample = int(sys.argv[5]) if len(sys.argv) > 5 else 1
bad = [0] * n_threads
rng = np.random.default_rng(3)
coeffs = (rng.random((streams * units, 2, 1024), dtype=np.float32) * 2000 - 1000).astype(np.float32)


def work(t):
    eng = soundkit_amd.Engine(0, streams + 8)
    first = None
    for it in range(iters):
        sids = [eng.open_stream(48000, 2) for _ in range(streams)]
        if resample:
            for s in sids:
                eng.resampler_open(s, 48000, 16000)
        ids = np.repeat(np.array(sids, np.uint32), units)
        seqs = np.zeros((streams * units, 2), np.uint8)
        shapes = np.tile((np.arange(units) & 1).astype(np.uint8), streams)[:, None].repeat(2, 1)
        descs, n = soundkit_amd.descs_from_arrays(ids, 2, seqs, shapes)
        ts = [dict(stream=s, n_frames=units, out_bits=16, out_channels=1, resample=resample, flush=1) for s in sids]
        res = eng.tick_run(ts, descs, n, coeffs)
        h = hashlib.sha1()
        for r in res:
            h.update(bytes([r[0] & 255, r[1] & 255]))
            h.update(r[5])
        dig = h.hexdigest()
        if first is None:
            first = (dig, res)
        elif dig != first[0]:
            bad[t] += 1
            if bad[t] <= 2:
                for a, b in zip(res, first[1]):
                    if a[5] != b[5]:
                        x, y = np.frombuffer(a[5], "<i2").astype(np.int32), np.frombuffer(b[5], "<i2").astype(np.int32)
                        d = np.flatnonzero(x != y) if x.size == y.size else np.array([-1])
                        print("thread", t, "iteration", it, "stream index", a[0], "frames", a[2], "differs at", d.size, "samples", d[:6].tolist(),
                              "max", int(np.abs(x - y).max()) if x.size == y.size else None, flush=True)
                        break
        for s in sids:
            eng.close_stream(s)


threads = [threading.Thread(target=work, args=(t,)) for t in range(n_threads)]
for th in threads:
    th.start()
for th in threads:
    th.join()
print("threads", n_threads, "iterations", iters, "streams", streams, "units", units, "resample", resample, "deviating ticks per thread:", bad)
