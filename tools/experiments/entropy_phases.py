#!/usr/bin/env python3
"""Per-phase clock time of k_aac_entropy_parse from a timing build (tools/build_ab.sh aac_entropy SK_EC_PROFILE):
  SOUNDKIT_AMD_LIB=soundkit_amd/ab/lib_SK_EC_PROFILE.so SK_ENTROPY_LANE_SHIFT=1 python3 tools/entropy_phases.py 3400"""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import soundkit_amd
from soundkit_amd import aac_lc, _lib

n_streams = int(sys.argv[1])
root = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden", "aac", "aac-stereo-48k.adts")
aus = [au for _, au in aac_lc.split_adts(open(root, "rb").read())]
eng = soundkit_amd.Engine(0, max(n_streams, 16))
sids = []
for i in range(n_streams):
    sid = eng.open_stream(48000, 2)
    eng.resampler_open(sid, 48000, 16000)
    sids.append(sid)
lib = _lib.lib
stamps = np.zeros((8192, 24), np.uint64)
for t in range(2):
    table, units = [], []
    for i, sid in enumerate(sids):
        take = [aus[(7 * i + 16 * t + k) % len(aus)] for k in range(16)]
        table.append({"stream": sid, "n_frames": 16, "out_bits": 16, "out_channels": 1, "resample": True, "flush": False})
        units += take
    lib.sk_debug_ec_stamps(None, 1)
    res = eng.tick_run_au(table, units)
    bad = [r for r in res if r[1] != 0]
    if bad: print("records with errors:", len(bad), bad[:2])
lib.sk_debug_ec_stamps(stamps.ctypes.data_as(C.c_void_p), 0)
used = stamps[stamps[:, 7] != 0].astype(np.int64)
order = [0, 1, 12, 15, 13, 14, 8, 9, 10, 11, 2, 3, 4, 5, 6, 7]
names = ["tables + zero fill", "unit + task records", "stream record, bit reader set up, first scratch store", "first 3 bits of the unit", "tag, common_window, common ics", "ms mask", "L: global gain (+ ics)", "L: sections", "L: scale factors",
         "L: pulse, tns flags/data", "spectrum L", "header R", "spectrum R", "after spectrum", "side record store"]
sel = used[:, order]
d = np.diff(sel, axis=1)
print("waves", len(used), "mean wave time (clock ticks)", int((used[:, 7] - used[:, 0]).mean()))
for i, n in enumerate(names):
    print("  %-48s mean %9.0f  max %9.0f ticks  (%4.1f %% of a wave's time)" % (n, d[:, i].mean(), d[:, i].max(), 100.0 * d[:, i].sum() / d.sum()))

fin = used[:, [16, 17, 18, 19, 20, 21, 22]]
fd = np.diff(fin, axis=1)
print("finish kernel, mean wave time %d ticks" % int((fin[:, -1] - fin[:, 0]).mean()))
for i, n in enumerate(["tables, records, status", "noise fill (both channels)", "stereo tools", "TNS, first channel", "TNS, second channel", "rest of the unit, status"]):
    print("  %-48s mean %9.0f  max %9.0f ticks  (%4.1f %% of a wave's time)" % (n, fd[:, i].mean(), fd[:, i].max(), 100.0 * fd[:, i].sum() / fd.sum()))
counts = np.zeros((8192, 10, 64), np.uint32)
lib.sk_debug_ec_counts(counts.ctypes.data_as(C.c_void_p))
cw = counts[:len(used)]
for ch, name in ((0, "first"), (2, "second")):
    passes, words = cw[:, ch, :32].astype(np.int64), cw[:, ch + 1, :32].astype(np.int64)
    i = order.index(3 if ch == 0 else 5) - 1
    print("  spectrum of the %s channel: passes per lane mean %.0f, per wave (max over lanes) mean %.0f; codewords per lane %.0f; "
          "ticks per wave-pass %.0f" % (name, passes.mean(), passes.max(axis=1).mean(), words.mean(), d[:, i].mean() / passes.max(axis=1).mean()))
for base, name in ((4, "first"), (7, "second")):
    t = cw[:, base:base + 3, 0].astype(np.float64) * 16.0   # wave-level clocks (lane 0's copy)
    tot = t.sum()
    print("  spectral loop of the %s channel, share of its time: open group/band %.0f %%, codeword (look, tables, signs, escapes) %.0f %%, "
          "pulses + dequantise + store + advance %.0f %%" % (name, 100 * t[:, 0].sum() / tot, 100 * t[:, 1].sum() / tot, 100 * t[:, 2].sum() / tot))
if os.environ.get("SK_PHASE_DUMP"):
    col = order.index(13) - 1
    print("first-read phase per wave (first 48 waves):", d[:48, col].tolist())
    print("spectrum L per wave:", d[:48, order.index(3) - 1].tolist())
eng.close()
