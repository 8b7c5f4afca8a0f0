#!/usr/bin/env python3
"""Streaming rate of a few soundkit::audio_bytes conversions on device-resident data (in + out bytes per second)."""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
import soundkit_amd
from soundkit_amd.engine import PCM_OP

eng = soundkit_amd.Engine(0, 16)
ext = torch.cuda.ExternalStream(eng.hip_stream)
n = 1 << 29
cases = [("S32LE_TO_I32", 4, 4), ("S32LE_TO_I16", 4, 2), ("I16LE_TO_F32", 2, 4), ("FLOAT_TO_I16_ROUND", 4, 2)]
for name, ib, ob in cases:
    if name not in PCM_OP:
        continue
    x = torch.randint(0, 255, (n * ib,), dtype=torch.uint8, device="cuda")
    y = torch.empty(n * ob, dtype=torch.uint8, device="cuda")
    torch.cuda.synchronize()
    for _ in range(3):
        eng.pcm_convert_dev(name, x, y, n)
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(ext)
    for _ in range(10):
        eng.pcm_convert_dev(name, x, y, n)
    b.record(ext)
    eng.synchronize()
    ms = a.elapsed_time(b) / 10
    print("%-16s %.3f ms  %.2f TB/s in+out" % (name, ms, n * (ib + ob) / ms / 1e9))
eng.close()
