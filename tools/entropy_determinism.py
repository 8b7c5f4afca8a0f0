"""Diagnostic: N identical streams through sk_aac_entropy_decode in one launch; every stream must produce stream 0's
spectra (and stream 0 the oracle's).  SK_ENTROPY_LANE_SHIFT selects the units-per-wave variant."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import soundkit_amd  # noqa: E402
from oracle import aac_frontend as OF  # noqa: E402

n_streams = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
per = int(sys.argv[2]) if len(sys.argv) > 2 else 24
clip = open(os.path.join(ROOT, "tests", "golden", "aac", sys.argv[3] if len(sys.argv) > 3 else "aac-stereo-48k.adts"), "rb").read()
frames = OF.split_adts(clip)
dec = OF.Decoder(frames[0][0])
units = [au for _, au in frames][:per]
want = [dec.decode_access_unit(au) for au in units]
eng = soundkit_amd.Engine(0, n_streams)
sids = [eng.open_stream(dec.sample_rate, dec.channels) for _ in range(n_streams)]
got = eng.entropy_decode([(s, per) for s in sids], units * n_streams)
bad_streams, first = 0, None
for s in range(n_streams):
    ok = True
    for i in range(per):
        st, c, seq, shape = got[s * per + i]
        if st != 0 or not np.array_equal(c.view(np.uint32), want[i][0].view(np.uint32)) or (seq, shape) != (want[i][1], want[i][2]):
            ok = False
            if first is None:
                d = np.nonzero(c.view(np.uint32) != want[i][0].view(np.uint32))
                first = (s, i, st, [x[:8].tolist() for x in d], c[d][:8].tolist(), want[i][0][d][:8].tolist())
            break
    bad_streams += not ok
print("lane_shift=%s streams=%d per=%d bad_streams=%d first=%s" % (os.environ.get("SK_ENTROPY_LANE_SHIFT"), n_streams, per, bad_streams, first))
eng.close()
