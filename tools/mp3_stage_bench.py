"""The two GPU stages of the MP3 path at scale, for rocprofv3 --kernel-trace --stats (profiles/r03_mp3_kernel_stats.csv):
STREAMS stereo streams x GRANULES granules of random integers through sk_mp3_decode_granules_s16 (requantisation +
mid/side + reorder, then the hybrid synthesis), band tables and window synthetic.  Host-buffer entry point: the wall time
printed includes the PCIe copies; the kernel times are what the profiler reports.
    python3 tools/mp3_stage_bench.py [streams] [granules_per_stream] [repeats]"""
import ctypes as C
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))

import soundkit_amd  # noqa: E402
from soundkit_amd import mp3  # noqa: E402
from soundkit_amd._lib import Mp3GranuleDesc, check, lib  # noqa: E402
from soundkit_amd.engine import _ptr  # noqa: E402


def main():
    streams = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
    per = int(sys.argv[2]) if len(sys.argv) > 2 else 32
    repeats = int(sys.argv[3]) if len(sys.argv) > 3 else 5
    rng = np.random.default_rng(1)
    engine = soundkit_amd.Engine(0, streams)
    long_o = np.array([0, 4, 8, 12, 16, 20, 24, 30, 36, 44, 52, 62, 74, 90, 110, 134, 162, 196, 238, 288, 342, 418, 576], np.uint16)
    short_o = np.array([0, 4, 8, 12, 16, 22, 30, 40, 52, 66, 84, 106, 136, 192], np.uint16)
    assert mp3.set_band_tables(44100, long_o, short_o, np.zeros(22, np.uint8), engine) == 0
    from oracle import mp3_hybrid
    mp3.set_synthesis_window(mp3_hybrid.synthetic_window(3), engine)
    sids = [engine.open_stream(44100, 2) for _ in range(streams)]
    n = streams * per
    granules = []
    for g in range(per):          # granule-major: the streams advance together
        for s in range(streams):
            bt = int(rng.integers(0, 4)) if (s % 8 == 0) else 0
            ch = {"global_gain": 130, "scalefac_scale": 0, "preflag": 0, "block_type": bt, "mixed_block_flag": 0, "subblock_gain": [0, 0, 0],
                  "scalefac_l": [1] * 21 + [0], "scalefac_s": [[1, 1, 1]] * 12 + [[0, 0, 0]]}
            granules.append((sids[s], {"sample_rate": 44100, "channels": 2, "ms_stereo": 1, "ch": [ch, ch]}, bt))
    req = mp3.make_requant_granules([g for _, g, _ in granules])
    descs = (Mp3GranuleDesc * n)()
    for i, (sid, _, bt) in enumerate(granules):
        descs[i].stream, descs[i].channels = sid, 2
        descs[i].block_type[0] = descs[i].block_type[1] = bt
    quant = rng.integers(-8, 9, (2 * n, 576)).astype(np.int16)
    pcm = np.zeros((n, 576, 2), np.int16)
    status = np.zeros(n, np.int32)
    times = []
    for _ in range(repeats):
        t0 = time.perf_counter()
        check(lib.sk_mp3_decode_granules_s16(engine._h, req, descs, _ptr(quant), _ptr(pcm), n, _ptr(status)), "sk_mp3_decode_granules_s16", engine._h)
        times.append(time.perf_counter() - t0)
    assert not status.any() and pcm.any()
    best = min(times)
    print(json.dumps({"workload": "mp3 stages: %d stereo streams x %d granules (44.1 kHz), integers in host memory -> s16 PCM in host memory" % (streams, per),
                      "granule_channels": 2 * n, "best_call_ms": best * 1e3, "granule_channels_per_s": 2 * n / best,
                      "x_realtime_stereo": n * 576 / 44100 / best, "note": "wall time of the host-buffer call incl. PCIe; kernel times: rocprofv3 stats"}))
    engine.close()


if __name__ == "__main__":
    main()
