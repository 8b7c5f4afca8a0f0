"""The AAC-LC front-end on the GPU (csrc/aac_entropy.hip over csrc/aac_entropy_core.h): first against the ORACLE
(oracle/aac_frontend.py) through sk_aac_entropy_decode -- spectra bit for bit, window fields and error kinds on every
fixture access unit and on damaged ones -- then the whole worker path (sk_tick_run_au: raw access units to output
bytes without the host touching the bitstream) against the same tick fed by the host front-end."""
import os

import numpy as np
import pytest

from soundkit_amd import aac_lc
from soundkit_amd.engine import make_descs

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "aac")
FILES = ["aac-stereo-48k.adts", "mono16k_A_Tusk.aac", "stereo-music-44100-192k.aac", "A_Tusk_is_used_to_make_costly_gifts_encoded.aac"]


def load(name):
    frames = aac_lc.split_adts(open(os.path.join(GOLD, name), "rb").read())
    fe = aac_lc.AacLcFrontEnd(frames[0][0])
    return fe, [au for _, au in frames]


def run(engine, mode, specs, per_tick):
    """specs: [(front-end, access units, out_bits, out_rate, out_channels)]; mode 'host' parses on the CPU and calls
    sk_tick_run, mode 'gpu' hands the access units to sk_tick_run_au.  -> per stream list of (status, bits, ch, bytes)"""
    sids, pos, outs = [], [0] * len(specs), [[] for _ in specs]
    fes = []
    for fe, aus, bits, rate, ch in specs:
        sid = engine.open_stream(fe.sample_rate, fe.channels)
        if rate and rate != fe.sample_rate:
            engine.resampler_open(sid, fe.sample_rate, rate)
        sids.append(sid)
        fes.append(aac_lc.AacLcFrontEnd(bytes([((2 << 3) | (sf_index(fe.sample_rate) >> 1)) & 0xFF,
                                                ((sf_index(fe.sample_rate) & 1) << 7) | (fe.channels << 3)])))
    alive = [True] * len(specs)
    while any(alive[i] and pos[i] < len(specs[i][1]) for i in range(len(specs))):
        table, units, descs_in, coeffs, index, quants, sides = [], [], [], [], [], [], []
        for i, (fe, aus, bits, rate, ch) in enumerate(specs):
            if not alive[i] or pos[i] >= len(aus):
                continue
            take = aus[pos[i]:pos[i] + per_tick[i]]
            pos[i] += len(take)
            last = pos[i] >= len(aus)
            resample = bool(rate and rate != fe.sample_rate)
            n_ok = len(take)
            if mode == "quant":
                parsed_q = []
                for au in take:
                    try:
                        parsed_q.append(fes[i].parse_q(au))
                    except aac_lc.AacLcError as e:
                        outs[i].append(("error", e.status))
                        alive[i] = False
                        break
                n_ok = len(parsed_q)
                for q, side, seqs, shapes in parsed_q:
                    descs_in.append((sids[i], fe.channels, list(seqs) + [0] * (2 - fe.channels), list(shapes) + [0] * (2 - fe.channels)))
                    quants.append(q.ravel())
                    sides.append(side)
            elif mode == "host":
                parsed = []
                for au in take:
                    try:
                        parsed.append(fes[i].parse(au))
                    except aac_lc.AacLcError as e:
                        outs[i].append(("error", e.status))
                        alive[i] = False
                        break
                n_ok = len(parsed)
                for c, seqs, shapes in parsed:
                    descs_in.append((sids[i], fe.channels, list(seqs) + [0] * (2 - fe.channels), list(shapes) + [0] * (2 - fe.channels)))
                    coeffs.append(c.ravel())
            else:
                units += take
            table.append({"stream": sids[i], "n_frames": n_ok, "out_bits": bits or 16, "out_channels": ch or fe.channels,
                          "resample": resample, "flush": last and resample and alive[i]})
            index.append(i)
        if mode == "quant":
            descs, n = make_descs(descs_in)
            res = engine.tick_run_q(table, descs, n, np.stack(sides) if sides else np.zeros((0, 8), np.uint8),
                                    np.concatenate(quants) if quants else np.zeros(0, np.int16))
        elif mode == "host":
            descs, n = make_descs(descs_in)
            res = engine.tick_run(table, descs, n, np.concatenate(coeffs) if coeffs else np.zeros(0, np.float32))
        else:
            res = engine.tick_run_au(table, units)
        pending_error = {}
        for idx, status, nframes, ch_o, bits_o, data in res:
            i = index[idx]
            if status != 0:
                outs[i].append(("error", status))
                alive[i] = False
            else:
                outs[i].append((bits_o, ch_o, data))
    for sid in sids:
        engine.close_stream(sid)
    return outs


def test_gpu_front_end_equals_oracle_on_every_fixture_unit(engine):
    """All 273 access units of the four ADTS fixtures, every file its own stream, decoded by the gfx950 front-end in
    two calls (the PNS generator is carried in the engine between them): bit-identical to the oracle's spectra."""
    from oracle import aac_frontend as OF
    loaded = []
    for name in FILES:
        frames = OF.split_adts(open(os.path.join(GOLD, name), "rb").read())
        dec = OF.Decoder(frames[0][0])
        want = [dec.decode_access_unit(au) for _, au in frames]
        loaded.append((engine.open_stream(dec.sample_rate, dec.channels), [au for _, au in frames], want))
    try:
        total = 0
        done = [0] * len(loaded)
        for part in (0, 1):
            spans = [(0, len(u) // 3) if part == 0 else (len(u) // 3, len(u)) for _, u, _ in loaded]
            got = engine.entropy_decode([(sid, b - a) for (sid, _, _), (a, b) in zip(loaded, spans)],
                                        [au for (_, u, _), (a, b) in zip(loaded, spans) for au in u[a:b]])
            pos = 0
            for (sid, units, want), (a, b) in zip(loaded, spans):
                for i in range(a, b):
                    status, coeffs, seq, shape = got[pos]
                    pos += 1
                    assert status == 0, (sid, i, status)
                    assert (seq, shape) == (want[i][1], want[i][2]), (sid, i)
                    assert np.array_equal(coeffs.view(np.uint32), want[i][0].view(np.uint32)), (sid, i, np.abs(coeffs - want[i][0]).max())
                    total += 1
        assert total == 273
    finally:
        for sid, _, _ in loaded:
            engine.close_stream(sid)


def test_gpu_front_end_rejects_what_the_oracle_rejects(engine):
    """1200 damaged access units (the mutations of tests/test_oracle_frontend.py), each on a fresh stream so that the
    PNS generator starts from the reference's seed on both sides, all in one launch: the same verdict as the oracle --
    identical spectra when it accepts, the same AacLcError kind when it rejects."""
    from oracle import aac_frontend as OF
    from soundkit_amd._lib import ERR_NAMES
    state = 0x2545F4914F6CDD1D
    cases = []
    for name in ("aac-stereo-48k.adts", "mono16k_A_Tusk.aac", "stereo-music-44100-192k.aac"):
        frames = OF.split_adts(open(os.path.join(GOLD, name), "rb").read())
        for trial in range(400):
            state ^= (state << 13) & 0xFFFFFFFFFFFFFFFF
            state ^= state >> 7
            state ^= (state << 17) & 0xFFFFFFFFFFFFFFFF
            au = bytearray(frames[(state >> 8) % len(frames)][1])
            for k in range(1 + (state >> 20) % 3):
                r = (state >> (24 + 9 * k)) & 0xFFFFFF
                if r % 3 == 0:
                    au[(r >> 4) % len(au)] ^= 1 << (r & 7)
                elif r % 3 == 1:
                    au[(r >> 4) % len(au)] = (r >> 12) & 0xFF
                else:
                    del au[(r >> 4) % len(au):]
                    if not au:
                        au = bytearray(b"\0")
            oracle = OF.Decoder(frames[0][0])
            try:
                want = oracle.decode_access_unit(bytes(au))
            except OF.AacError as e:
                want = e
            cases.append((engine.open_stream(oracle.sample_rate, oracle.channels), bytes(au), want))
    try:
        got = engine.entropy_decode([(sid, 1) for sid, _, _ in cases], [au for _, au, _ in cases])
        accepted = rejected = 0
        for k, ((sid, au, want), (status, coeffs, seq, shape)) in enumerate(zip(cases, got)):
            if isinstance(want, OF.AacError):
                assert status != 0 and ERR_NAMES[status] == want.kind, (k, status, want.kind, str(want))
                assert not coeffs.any()
                rejected += 1
            else:
                assert status == 0, (k, status)
                assert (seq, shape) == (want[1], want[2]), k
                assert np.array_equal(coeffs.view(np.uint32), want[0].view(np.uint32)), k
                accepted += 1
        assert accepted > 50 and rejected > 300
    finally:
        for sid, _, _ in cases:
            engine.close_stream(sid)


def test_quantised_hand_over_equals_oracle_on_every_fixture_unit(engine):
    """SURVEY 8f rank 1 against the ORACLE: every fixture access unit through sk_aac_decoder_parse_q (host: Huffman decode to
    i16 + side record) and sk_aac_expand_q_decode (device: dequantisation dsp.rs:397-405, noise, intensity / mid-side
    stereo.rs:114-448, TNS tns.rs:237-276) -- spectra bit for bit and the window fields of oracle/aac_frontend.py, in two
    calls per stream (the PNS generator is carried in the engine between them)."""
    from oracle import aac_frontend as OF
    loaded = []
    for name in FILES:
        frames = OF.split_adts(open(os.path.join(GOLD, name), "rb").read())
        dec = OF.Decoder(frames[0][0])
        want = [dec.decode_access_unit(au) for _, au in frames]
        fe = aac_lc.AacLcFrontEnd(frames[0][0])
        loaded.append((engine.open_stream(dec.sample_rate, dec.channels), [fe.parse_q(au) for _, au in frames], want))
    try:
        total = 0
        for part in (0, 1):
            spans = [(0, len(u) // 3) if part == 0 else (len(u) // 3, len(u)) for _, u, _ in loaded]
            got = engine.expand_q_decode([(sid, b - a) for (sid, _, _), (a, b) in zip(loaded, spans)],
                                         [q for (_, u, _), (a, b) in zip(loaded, spans) for q in u[a:b]])
            pos = 0
            for (sid, units, want), (a, b) in zip(loaded, spans):
                for i in range(a, b):
                    status, coeffs, seq, shape = got[pos]
                    pos += 1
                    assert status == 0, (sid, i, status)
                    assert (seq, shape) == (want[i][1], want[i][2]), (sid, i)
                    assert np.array_equal(coeffs.view(np.uint32), want[i][0].view(np.uint32)), (sid, i, np.abs(coeffs - want[i][0]).max())
                    total += 1
        assert total == 273
    finally:
        for sid, _, _ in loaded:
            engine.close_stream(sid)


def test_quantised_hand_over_rejects_what_the_oracle_rejects(engine):
    """The damaged access units of the test above this one's model (1200 mutations), each on a fresh stream: the two
    halves of the quantised hand-over together give the oracle's verdict -- the host half (parse_q) rejects what fails
    up to the spectral data, the device half what fails in the stereo tools, TNS or the rest of the unit -- with the
    reference's error kind, identical spectra when both accept."""
    from oracle import aac_frontend as OF
    from soundkit_amd._lib import ERR_NAMES
    state = 0x2545F4914F6CDD1D
    cases = []
    host_rejected = 0
    for name in ("aac-stereo-48k.adts", "mono16k_A_Tusk.aac", "stereo-music-44100-192k.aac"):
        frames = OF.split_adts(open(os.path.join(GOLD, name), "rb").read())
        for trial in range(400):
            state ^= (state << 13) & 0xFFFFFFFFFFFFFFFF
            state ^= state >> 7
            state ^= (state << 17) & 0xFFFFFFFFFFFFFFFF
            au = bytearray(frames[(state >> 8) % len(frames)][1])
            for k in range(1 + (state >> 20) % 3):
                r = (state >> (24 + 9 * k)) & 0xFFFFFF
                if r % 3 == 0:
                    au[(r >> 4) % len(au)] ^= 1 << (r & 7)
                elif r % 3 == 1:
                    au[(r >> 4) % len(au)] = (r >> 12) & 0xFF
                else:
                    del au[(r >> 4) % len(au):]
                    if not au:
                        au = bytearray(b"\0")
            oracle = OF.Decoder(frames[0][0])
            try:
                want = oracle.decode_access_unit(bytes(au))
            except OF.AacError as e:
                want = e
            fe = aac_lc.AacLcFrontEnd(frames[0][0])
            try:
                parsed = fe.parse_q(bytes(au))
            except aac_lc.AacLcError as e:
                assert isinstance(want, OF.AacError) and e.kind == want.kind and str(want) in str(e), (name, trial, str(e), str(want))
                host_rejected += 1
                continue
            cases.append((engine.open_stream(oracle.sample_rate, oracle.channels), parsed, want))
    try:
        got = engine.expand_q_decode([(sid, 1) for sid, _, _ in cases], [q for _, q, _ in cases])
        accepted = rejected = 0
        for k, ((sid, q, want), (status, coeffs, seq, shape)) in enumerate(zip(cases, got)):
            if isinstance(want, OF.AacError):
                assert status != 0 and ERR_NAMES[status] == want.kind, (k, status, want.kind, str(want))
                assert not coeffs.any()
                rejected += 1
            else:
                assert status == 0, (k, status)
                assert (seq, shape) == (want[1], want[2]), k
                assert np.array_equal(coeffs.view(np.uint32), want[0].view(np.uint32)), k
                accepted += 1
        assert accepted > 50 and host_rejected > 200 and rejected > 20, (accepted, host_rejected, rejected)
    finally:
        for sid, _, _ in cases:
            engine.close_stream(sid)


def sf_index(rate):
    return [96000, 88200, 64000, 48000, 44100, 32000, 24000, 22050, 16000, 12000, 11025, 8000, 7350].index(rate)


@pytest.mark.parametrize("bits,rate,ch", [(None, None, None), (16, 16000, 1), (24, None, 1), (32, 8000, None)])
def test_gpu_front_end_equals_host_front_end(engine, bits, rate, ch):
    specs = []
    for name in FILES:
        fe, aus = load(name)
        specs.append((fe, aus, bits, rate, ch))
    per_tick = [5, 3, 8, 1]
    want = run(engine, "host", specs, per_tick)
    got = run(engine, "gpu", specs, per_tick)
    for name, w, g in zip(FILES, want, got):
        assert len(w) == len(g) and len(w) > 0, name
        assert w == g, name


def test_damaged_access_units_fail_with_the_host_codes(engine):
    """Mutated units: the GPU front-end rejects exactly the units the host front-end rejects, with the same status
    code, and delivers identical bytes for everything before them; the neighbouring stream is untouched."""
    fe, aus = load("aac-stereo-48k.adts")
    state = 0x853C49E6748FEA9B
    for trial in range(24):
        damaged = list(aus)
        state = (state * 6364136223846793005 + 1442695040888963407) & 0xFFFFFFFFFFFFFFFF
        k = 3 + (state >> 33) % 30
        au = bytearray(damaged[k])
        for j in range(1 + (state >> 20) % 4):
            r = (state >> (7 * j + 3)) & 0xFFFFF
            au[(r >> 3) % len(au)] ^= 1 << (r & 7)
        damaged[k] = bytes(au)
        specs = [(fe, damaged, None, 16000, 1), (fe, aus, None, None, None)]
        want = run(engine, "host", specs, [4, 6])
        got = run(engine, "gpu", specs, [4, 6])
        assert want[1] == got[1] and len(got[1]) == 48
        assert want[0] == got[0], (trial, k, [x for x in want[0] if x[0] == "error"], [x for x in got[0] if x[0] == "error"])


@pytest.mark.parametrize("bits,rate,ch", [(None, None, None), (16, 16000, 1), (24, None, 1)])
def test_quantised_hand_over_equals_the_f32_path(engine, bits, rate, ch):
    """the host front-end stops after the Huffman decode (sk_aac_decoder_parse_q: i16 quantised values +
    a 1.3 KB side record per unit) and dequantisation, PNS, intensity / mid-side and TNS run on the device
    (sk_tick_run_q) -- byte-identical AudioData to the tick fed with the host's finished f32 spectra, on every fixture
    (PNS, IS, MS, TNS, short windows all exercised), uneven units per tick."""
    specs = []
    for name in FILES:
        fe, aus = load(name)
        specs.append((fe, aus, bits, rate, ch))
    per_tick = [5, 3, 8, 1]
    want = run(engine, "host", specs, per_tick)
    got = run(engine, "quant", specs, per_tick)
    for name, w, g in zip(FILES, want, got):
        assert len(w) == len(g) and len(w) > 0, name
        assert w == g, name


def test_quantised_hand_over_on_generated_units_and_failures(engine):
    """generated units (pulse data, escapes, every codebook) through both paths; a unit that fails on the device (TNS /
    stereo tools / the host's verdict on its tail) ends its stream with the host path's status"""
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from au_builder import asc_for, random_access_unit
    specs = []
    for k, (sf_index, channels) in enumerate([(3, 2), (4, 1), (8, 2), (0, 2)]):
        rng = np.random.default_rng(400 + k)
        units = [random_access_unit(rng, sf_index, channels) for _ in range(30)]
        if k == 0:
            units[17] = units[17] + b"\x00\x80"     # non-zero trailing bits: found by the host, reported after the device's checks
        specs.append((aac_lc.AacLcFrontEnd(asc_for(sf_index, channels)), units, None, None, None))
    want = run(engine, "host", specs, [7, 4, 9, 2])
    got = run(engine, "quant", specs, [7, 4, 9, 2])
    assert want[1:] == got[1:] and all(len(w) == 30 for w in want[1:])
    # stream 0: the 17 units before the damaged one, then its error (this harness notes a host-side rejection before the
    # tick that carries the units in front of it, the device-side one after: compare without the position)
    good = lambda outs: [o for o in outs if o[0] != "error"]
    errs = lambda outs: [o for o in outs if o[0] == "error"]
    assert good(want[0]) == good(got[0]) and len(good(got[0])) == 17
    assert errs(want[0]) == errs(got[0]) == [("error", -108)]
