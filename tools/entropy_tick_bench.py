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
root = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden", "aac", "aac-stereo-48k.adts")
frames = aac_lc.split_adts(open(root, "rb").read())
aus = [au for _, au in frames]
eng = soundkit_amd.Engine(0, max(n_streams, 16))
sids = []
for i in range(n_streams):
    sid = eng.open_stream(48000, 2)
    eng.resampler_open(sid, 48000, 16000)
    sids.append(sid)
pos = [(7 * i) % len(aus) for i in range(n_streams)]
for t in range(n_ticks):
    table, units = [], []
    for i, sid in enumerate(sids):
        take = [aus[(pos[i] + k) % len(aus)] for k in range(16)]
        pos[i] = (pos[i] + 16) % len(aus)
        table.append({"stream": sid, "n_frames": 16, "out_bits": 16, "out_channels": 1, "resample": True, "flush": False})
        units += take
    res = eng.tick_run_au(table, units)
    bad = [r for r in res if r[1] != 0]
    assert not bad, bad[:3]
print("ok", n_streams * 16, "units per tick,", len(res), "records in the last tick")
eng.close()
