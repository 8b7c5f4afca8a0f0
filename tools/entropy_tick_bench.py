#!/usr/bin/env python3
"""One big sk_tick_run_au call repeated: S streams x 16 access units of the 48 kHz stereo fixture (each stream at its own
offset into the clip), 16 kHz mono s16 out.  Meant to run under `rocprofv3 --kernel-trace --stats` with
SK_ENTROPY_LANE_SHIFT set, to time the front-end kernels at a chosen tick size:  entropy_tick_bench.py <streams> <ticks>"""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import soundkit_amd
from soundkit_amd import aac_lc

n_streams, n_ticks = int(sys.argv[1]), int(sys.argv[2])
gold = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden", "aac")
# SK_TICK_CLIPS=all: streams take turns over the three stereo fixtures (48 + 131 + 46 distinct access units, three sampling
# rates) instead of the 48 kHz clip alone -- fewer identical units side by side in a wave
names = ["aac-stereo-48k.adts", "stereo-music-44100-192k.aac", "A_Tusk_is_used_to_make_costly_gifts_encoded.aac"] \
    if os.environ.get("SK_TICK_CLIPS") == "all" else ["aac-stereo-48k.adts"]
rates = {"aac-stereo-48k.adts": 48000, "stereo-music-44100-192k.aac": 44100, "A_Tusk_is_used_to_make_costly_gifts_encoded.aac": 16000}
clips = [(rates[n], [au for _, au in aac_lc.split_adts(open(os.path.join(gold, n), "rb").read())]) for n in names]
eng = soundkit_amd.Engine(0, max(n_streams, 16))
sids, clip_of = [], []
for i in range(n_streams):
    rate, _ = clips[i % len(clips)]
    sid = eng.open_stream(rate, 2)
    eng.resampler_open(sid, rate, 16000)
    sids.append(sid)
    clip_of.append(i % len(clips))
pos = [(7 * i) % len(clips[clip_of[i]][1]) for i in range(n_streams)]
for t in range(n_ticks):
    table, units = [], []
    for i, sid in enumerate(sids):
        aus = clips[clip_of[i]][1]
        take = [aus[(pos[i] + k) % len(aus)] for k in range(16)]
        pos[i] = (pos[i] + 16) % len(aus)
        table.append({"stream": sid, "n_frames": 16, "out_bits": 16, "out_channels": 1, "resample": True, "flush": False})
        units += take
    res = eng.tick_run_au(table, units)
    bad = [r for r in res if r[1] != 0]
    assert not bad, bad[:3]
print("ok", n_streams * 16, "units per tick,", len(res), "records in the last tick")
eng.close()
