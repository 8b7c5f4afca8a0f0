"""Diagnostic for tests/test_scale_gpu.py: which streams deliver different bytes, where and by how much."""
import ctypes as C
import os
import sys
import collections

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import soundkit_amd  # noqa: E402
from soundkit_amd import aac_lc, pipeline  # noqa: E402
from soundkit_amd._lib import DecodeOptionsC  # noqa: E402
from test_scale_gpu import Check, Result  # noqa: E402

streams = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
loops = int(sys.argv[2]) if len(sys.argv) > 2 else 2
gpu_entropy = int(sys.argv[3]) if len(sys.argv) > 3 else 1
out_rate = int(sys.argv[4]) if len(sys.argv) > 4 else 16000
out_ch = int(sys.argv[5]) if len(sys.argv) > 5 else 1
ncap = min(streams, 256)
clip = open(os.path.join(ROOT, "tests", "golden", "aac", "aac-stereo-48k.adts"), "rb").read()
units = len(aac_lc.split_adts(clip))
clip = clip[:sum(len(au) + 7 for _, au in aac_lc.split_adts(clip))]
lg = C.CDLL(os.path.join(ROOT, "soundkit_amd", "libsk_loadgen.so"))
lg.sk_loadgen_run_checked.restype = C.c_int
lg.sk_loadgen_run_checked.argtypes = [C.c_void_p, C.c_char_p, C.c_size_t, C.c_uint32, C.c_uint32, C.c_uint32, C.c_void_p,
                                      C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p]
hashes, outputs = np.zeros(streams, np.uint64), np.zeros(streams, np.uint32)
nbytes, errors = np.zeros(streams, np.uint64), np.zeros(streams, np.uint32)
capture = np.arange(ncap, dtype=np.uint32)
cap = 1 << 20
buf, lens = np.zeros((ncap, cap), np.uint8), np.zeros(ncap, np.uint64)
chk = Check(hashes.ctypes.data, outputs.ctypes.data, nbytes.ctypes.data, errors.ctypes.data, capture.ctypes.data, ncap,
            buf.ctypes.data, cap, lens.ctypes.data)
eng = soundkit_amd.Engine(0, streams)
extra = {k[3:].lower(): int(v) for k, v in os.environ.items() if k.startswith("SD_")}
print("scheduler options", extra)
sched = pipeline.BatchScheduler(eng, max_streams=streams, gpu_entropy=gpu_entropy, lanes=1, **extra)
res = Result()
opt = DecodeOptionsC(out_rate, 16, out_ch, 0)
rc = lg.sk_loadgen_run_checked(sched._h, clip, len(clip), units, streams, loops, C.byref(opt), 6, 0, C.byref(res), C.byref(chk))
sched.close()
cnt = collections.Counter(hashes.tolist())
print("rc", rc, "errors", res.errors, "distinct hashes", len(cnt), "top", cnt.most_common(3), "outputs", np.unique(outputs), "bytes", np.unique(nbytes))
major = cnt.most_common(1)[0][0]
ref = next(k for k in range(ncap) if hashes[k] == major)
a = np.frombuffer(buf[ref, :int(lens[ref])].tobytes(), "<i2").astype(np.int32)
shown = 0
for k in range(ncap):
    if hashes[k] != major and shown < 6:
        b = np.frombuffer(buf[k, :int(lens[k])].tobytes(), "<i2").astype(np.int32)
        d = np.nonzero(a != b)[0]
        if out_rate == 0 and out_ch == 0:
            per = 4096 // 2
            aus = sorted(set((d // per).tolist()))
            print("   differing access units:", aus)
        print("stream", k, "differs at", d.size, "samples; first", d[:5], "last", d[-3:], "max abs", np.abs(a - b).max(), "of", a.size)
        shown += 1
eng.close()
