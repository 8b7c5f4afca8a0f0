#!/usr/bin/env python3
"""Extract the ADTS AAC elementary stream of an MPEG-TS file (188-byte packets): PAT -> PMT -> the first
stream of type 0x0F, PES payloads concatenated.  Used once to turn the reference's only 48 kHz stereo
AAC-LC fixture (testdata/mpeg-ts/aac-stereo-48k.ts, 48 access units, testdata/mpeg-ts/README.md) into
tests/golden/aac/aac-stereo-48k.adts; the scheduler bench loops those 48 packets.

    python tools/ts_to_adts.py IN.ts OUT.adts
"""
import sys


def packets(data):
    for off in range(0, len(data) - 187, 188):
        p = data[off:off + 188]
        if p[0] != 0x47:
            raise ValueError("lost TS sync at %d" % off)
        pid = ((p[1] & 0x1F) << 8) | p[2]
        start = bool(p[1] & 0x40)
        afc = (p[3] >> 4) & 3
        body = 4
        if afc & 2:
            body += 1 + p[4]
        if afc & 1 and body < 188:
            yield pid, start, p[body:]


def section(payload):  # PSI section carried at the start of a payload_unit_start packet
    ptr = payload[0]
    return payload[1 + ptr:]


def extract(data):
    pmt_pid = audio_pid = None
    out = bytearray()
    for pid, start, payload in packets(data):
        if pid == 0 and start and pmt_pid is None:
            s = section(payload)
            n = ((s[1] & 0x0F) << 8) | s[2]
            for i in range(8, 3 + n - 4, 4):
                if (s[i] << 8) | s[i + 1]:
                    pmt_pid = ((s[i + 2] & 0x1F) << 8) | s[i + 3]
                    break
        elif pid == pmt_pid and start and audio_pid is None:
            s = section(payload)
            n = ((s[1] & 0x0F) << 8) | s[2]
            i = 12 + (((s[10] & 0x0F) << 8) | s[11])
            while i < 3 + n - 4:
                stype, epid = s[i], ((s[i + 1] & 0x1F) << 8) | s[i + 2]
                if stype == 0x0F:
                    audio_pid = epid
                    break
                i += 5 + (((s[i + 3] & 0x0F) << 8) | s[i + 4])
        elif pid == audio_pid and audio_pid is not None:
            if start:
                if payload[:3] != b"\x00\x00\x01":
                    raise ValueError("bad PES start code")
                payload = payload[9 + payload[8]:]
            out += payload
    return bytes(out)


if __name__ == "__main__":
    es = extract(open(sys.argv[1], "rb").read())
    open(sys.argv[2], "wb").write(es)
    print("%d bytes of ADTS" % len(es))
