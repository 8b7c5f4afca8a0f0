"""oracle/mp3_iso.py -- the Layer III data tables for the CPU checker, in the dict shape oracle/mp3_bitstream.py takes.

TEST INFRASTRUCTURE ONLY; nothing under soundkit_amd/ imports it.

The numbers are oracle/mp3_iso_tables.json, written by tools/transcribe_iso_mp3_tables.py together with the product's
csrc/mp3_iso_tables.h (normative constants of ISO/IEC 11172-3 Tables B.3 / B.6 / B.7 / B.8 and 13818-3 2.4.3.2; the tool's
header says where they were read and what they were checked against).  tests/test_mp3_iso_tables.py compares this copy
with what the library hands out through sk_mp3_iso_tables."""
import json
import os

import numpy as np

_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "mp3_iso_tables.json")
_cache = None


def tables():
    global _cache
    if _cache is None:
        with open(_PATH) as fh:
            doc = json.load(fh)
        _cache = {"big_values": doc["big_values"], "count1": doc["count1"], "slen": doc["slen"], "lsf_partitions": doc["lsf_partitions"],
                  "bands": {int(rate): (rows[0], rows[1]) for rate, rows in doc["bands"].items()}, "pretab": doc["pretab"],
                  "window": np.asarray(doc["window_q16"], np.float64) / 65536.0}
    return _cache
