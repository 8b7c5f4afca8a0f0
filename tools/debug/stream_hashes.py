"""Do all streams of a whole-decode run deliver the same bytes?  Every stream is fed the same clip, so every stream's
order-sensitive hash over its AudioData (csrc/load_gen.cpp, sk_loadgen_run_checked) must be the same.
   python tools/debug/stream_hashes.py STREAMS LOOPS LANES QUOTA [FRONT_END]   (on the GPU box)"""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import soundkit_amd  # noqa: E402
from soundkit_amd import aac_lc, pipeline  # noqa: E402
from soundkit_amd._lib import DecodeOptionsC  # noqa: E402
from test_scale_gpu import Check, Result  # noqa: E402

streams, loops, lanes, quota = (int(v) for v in sys.argv[1:5])
front = int(sys.argv[5]) if len(sys.argv) > 5 else 1
clip = open(os.path.join(ROOT, "tests", "golden", "aac", "aac-stereo-48k.adts"), "rb").read()
frames = aac_lc.split_adts(clip)
clip = clip[:sum(len(au) + 7 for _, au in frames)]
lg = C.CDLL(os.path.join(ROOT, "soundkit_amd", "libsk_loadgen.so"))
lg.sk_loadgen_run_checked.restype = C.c_int
lg.sk_loadgen_run_checked.argtypes = [C.c_void_p, C.c_char_p, C.c_size_t, C.c_uint32, C.c_uint32, C.c_uint32, C.c_void_p, C.c_uint32, C.c_uint32,
                                      C.c_void_p, C.c_void_p]
hashes, outputs = np.zeros(streams, np.uint64), np.zeros(streams, np.uint32)
nbytes, errors = np.zeros(streams, np.uint64), np.zeros(streams, np.uint32)
capture = np.arange(4, dtype=np.uint32)
buf, lens = np.zeros((capture.size, 1 << 23), np.uint8), np.zeros(capture.size, np.uint64)
chk = Check(hashes.ctypes.data, outputs.ctypes.data, nbytes.ctypes.data, errors.ctypes.data, capture.ctypes.data, capture.size, buf.ctypes.data,
            buf.shape[1], lens.ctypes.data)
eng = soundkit_amd.Engine(0, streams + 16)
sched = pipeline.BatchScheduler(eng, max_streams=streams, gpu_entropy=front, lanes=lanes, max_stream_frames_per_tick=quota)
res = Result()
opt = DecodeOptionsC(int(os.environ.get("DBG_RATE", "16000")), 16, int(os.environ.get("DBG_CH", "1")), 0)
rc = lg.sk_loadgen_run_checked(sched._h, clip, len(clip), len(frames), streams, loops, C.byref(opt), 4, 0, C.byref(res), C.byref(chk))
sched.close()
vals, counts = np.unique(hashes, return_counts=True)
major = vals[np.argmax(counts)]
odd = np.flatnonzero(hashes != major)
print("rc", rc, "streams", streams, "loops", loops, "lanes", lanes, "quota", quota, "front", front, "| units/s %.2f M" % (res.access_units / res.seconds / 1e6),
      "| distinct hashes", vals.size, "| streams off the majority:", odd.size, odd[:16].tolist(), "| errors", int(res.errors),
      "| outputs", np.unique(outputs).tolist()[:4], "bytes", np.unique(nbytes).tolist()[:4])

pcm = [np.frombuffer(buf[k, :int(lens[k])].tobytes(), "<i2").astype(np.int32) for k in range(capture.size)]
ref = None
for k in range(capture.size):
    if hashes[capture[k]] == major:
        ref = pcm[k]
for k in range(capture.size):
    if ref is None or pcm[k].size != ref.size:
        print("stream", capture[k], "no reference / other length", pcm[k].size)
        continue
    d = np.abs(pcm[k] - ref)
    bad = np.flatnonzero(d > 0)
    if bad.size == 0:
        print("stream", capture[k], "equals the majority")
        continue
    # group into bursts
    gaps = np.flatnonzero(np.diff(bad) > 200)
    starts = np.concatenate([[bad[0]], bad[gaps + 1]])
    ends = np.concatenate([bad[gaps], [bad[-1]]])
    print("stream", capture[k], "differs at", bad.size, "samples in", starts.size, "bursts; max", int(d.max()))
    for a, b in list(zip(starts, ends))[:12]:
        print("    burst [%d..%d] len %d  | chunk %.3f  unit %.3f | max %d" % (a, b, b - a + 1, a * 3 / 4096, a * 3 / 1024, int(d[a:b + 1].max())))
