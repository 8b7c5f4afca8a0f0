"""Random, syntactically valid access units (tests/au_builder.py: every tool, pulse data and escapes included) through
the oracle and through the product's front-ends: the host C++ (`sk_aac_decoder_parse`) here on the CPU, the gfx950
kernels (`sk_aac_entropy_decode`) under -m gpu.  Spectra bit for bit, window fields, and the PNS generator carried
from unit to unit of a stream."""
import numpy as np
import pytest

from au_builder import asc_for, extreme_scalefactor_units, random_access_unit
from oracle import aac_frontend as OF
from soundkit_amd import aac_lc

CONFIGS = [(3, 2), (4, 2), (8, 1), (4, 1), (11, 2), (0, 2), (6, 1)]  # (sampling frequency index, channels)


def make_stream(seed, sf_index, channels, n):
    rng = np.random.default_rng(seed)
    return [random_access_unit(rng, sf_index, channels) for _ in range(n)]


def oracle_decode(asc, units):
    """-> [(coeffs, seq, shape)] up to the first rejected unit, then the AacError (or None)"""
    dec, out = OF.Decoder(asc), []
    for au in units:
        try:
            out.append(dec.decode_access_unit(au))
        except OF.AacError as e:
            return out, e
    return out, None


def test_builder_covers_every_tool():
    """the generator reaches what it claims to reach (counted on the oracle's parse of its output)"""
    seen = {"pulse": 0, "tns": 0, "noise": 0, "intensity": 0, "short": 0, "escape_book": 0, "ms": 0}
    for k, (sf_index, channels) in enumerate(CONFIGS[:3]):
        for au in make_stream(100 + k, sf_index, channels, 25):
            r = OF.Bits(au)
            spy = OF.Decoder(asc_for(sf_index, channels))
            real = OF.Channel.__init__

            def init(self, rr, common, read_delta=None):
                real(self, rr, common, read_delta)
                seen["pulse"] += self.pulse is not None
                seen["tns"] += self.tns is not None
                flat = [b for row in self.books for b in row]
                seen["noise"] += 13 in flat
                seen["intensity"] += 14 in flat or 15 in flat
                seen["escape_book"] += 11 in flat
                seen["short"] += self.ics.sequence == 2
            OF.Channel.__init__ = init
            try:
                spy.decode_access_unit(au)
            except OF.AacError:
                pass
            finally:
                OF.Channel.__init__ = real
            del r
    assert all(v >= 3 for k, v in seen.items() if k != "ms"), seen


@pytest.mark.parametrize("k", range(len(CONFIGS)))
def test_host_front_end_equals_oracle_on_random_units(k):
    sf_index, channels = CONFIGS[k]
    asc = asc_for(sf_index, channels)
    accepted = 0
    for run in range(6):
        units = make_stream(1000 * k + run, sf_index, channels, 20)
        want, err = oracle_decode(asc, units)
        fe = aac_lc.AacLcFrontEnd(asc)
        for i, (coeffs, seq, shape) in enumerate(want):
            got, gseq, gshape = fe.parse(units[i])
            assert (gseq, gshape) == (seq, shape), (k, run, i)
            assert np.array_equal(got.view(np.uint32), coeffs.view(np.uint32)), (k, run, i, np.abs(got - coeffs).max())
        accepted += len(want)
        if err is not None:
            with pytest.raises(aac_lc.AacLcError) as exc:
                fe.parse(units[len(want)])
            assert exc.value.kind == err.kind and str(err) in str(exc.value)
    assert accepted >= 100


def test_values_beyond_the_reference_tables_host():
    """scale factors past -256..511, intensity positions past +-256: the reference leaves its tables for powf there"""
    for sf_index, channels, au in extreme_scalefactor_units():
        want, seq, shape = OF.Decoder(asc_for(sf_index, channels)).decode_access_unit(au)
        got, gseq, gshape = aac_lc.AacLcFrontEnd(asc_for(sf_index, channels)).parse(au)
        assert np.isfinite(want).all() and np.count_nonzero(want[-1]) >= 16
        assert (gseq, gshape) == (seq, shape) and np.array_equal(got.view(np.uint32), want.view(np.uint32))


@pytest.mark.gpu
def test_values_beyond_the_reference_tables_gpu(engine):
    """the same on the device, which evaluates no powf: every reachable value is tabulated from the host's libm.
    Also an escape sequence of the largest size the reference accepts (16 extra bits, magnitude >= 65536)."""
    cases = list(extreme_scalefactor_units())
    from au_builder import Writer
    w = Writer()
    w.put(0, 3), w.put(0, 4), w.put(60, 8)
    w.put(0, 1), w.put(0, 2), w.put(0, 1), w.put(1, 6), w.put(0, 1)
    w.put(11, 4), w.put(1, 5), w.code("sf", 60), w.put(0, 1), w.put(0, 1), w.put(0, 1)
    w.code(11, 16 * 17 + 16), w.put(1, 1), w.put(0, 1)                 # pair (16, 16), signs -, +
    for n, tail in ((12, 0x1234), (9, 0x0ABC)):                         # escapes of 16 and 13 bits
        for _ in range(n):
            w.put(1, 1)
        w.put(0, 1), w.put(tail, n + 4)
    w.code(11, 0)
    w.put(7, 3)
    cases.append((4, 1, w.bytes()))
    sids = [engine.open_stream(OF.RATES[sf], ch) for sf, ch, _ in cases]
    try:
        got = engine.entropy_decode([(sid, 1) for sid in sids], [au for _, _, au in cases])
        for (sf_index, channels, au), (status, coeffs, seq, shape) in zip(cases, got):
            want, wseq, wshape = OF.Decoder(asc_for(sf_index, channels)).decode_access_unit(au)
            assert status == 0 and (seq, shape) == (wseq, wshape)
            assert np.array_equal(coeffs.view(np.uint32), want.view(np.uint32)), np.abs(coeffs - want).max()
        assert abs(got[-1][1][0, 0]) > 2000   # the 16-bit escape really is there: (2^16 + 0x1234)^(4/3) * 2^-10
    finally:
        for sid in sids:
            engine.close_stream(sid)


@pytest.mark.gpu
def test_gpu_front_end_equals_oracle_on_random_units(engine):
    """every configuration as its own stream in ONE sk_aac_entropy_decode call, then a second call on the same
    streams (the PNS generator state lives in the engine between calls)"""
    from soundkit_amd._lib import ERR_NAMES
    rates = OF.RATES
    streams, units, wants = [], [], []
    for k, (sf_index, channels) in enumerate(CONFIGS):
        for run in range(4):
            u = make_stream(5000 + 10 * k + run, sf_index, channels, 24)
            want, err = oracle_decode(asc_for(sf_index, channels), u)
            streams.append((engine.open_stream(rates[sf_index], channels), u, want, err))
    try:
        for first, last in ((0, 9), (9, 24)):
            got = engine.entropy_decode([(sid, last - first) for sid, _, _, _ in streams],
                                        [au for _, u, _, _ in streams for au in u[first:last]])
            pos = 0
            for sid, u, want, err in streams:
                for i in range(first, last):
                    status, coeffs, seq, shape = got[pos]
                    pos += 1
                    if i < len(want):
                        assert status == 0, (sid, i, status)
                        assert (seq, shape) == (want[i][1], want[i][2]), (sid, i)
                        assert np.array_equal(coeffs.view(np.uint32), want[i][0].view(np.uint32)), (sid, i)
                    elif i == len(want):
                        assert ERR_NAMES[status] == err.kind, (sid, i, status, err.kind)
                        assert not coeffs.any()
        assert sum(len(w) for _, _, w, _ in streams) > 400
    finally:
        for sid, _, _, _ in streams:
            engine.close_stream(sid)


@pytest.mark.gpu
@pytest.mark.parametrize("damage", ["trailing_bits", "truncated"])
def test_a_failed_unit_leaves_the_stream_where_the_sequential_decoder_does(engine, damage):
    """A unit that fails late (non-zero trailing bits: found after everything else has been decoded) or while parsing
    (cut short), in the middle of a launch: the units behind it are not decoded (status -199, zero spectra) and the
    stream's PNS generator stands where the oracle's stands when it raises -- the next launch on the stream continues
    bit-identically with the oracle decoder that saw the same error."""
    from soundkit_amd._lib import ERR_NAMES
    sf_index, channels = 3, 2
    u = make_stream(7100, sf_index, channels, 14)
    asc = asc_for(sf_index, channels)

    def attempt(bad):
        """oracle on u[:5] + bad -> (decoder, results, error, generator before / after the bad unit)"""
        dec = OF.Decoder(asc)
        res = [dec.decode_access_unit(au) for au in u[:5]]
        before = dec.pns_state
        try:
            dec.decode_access_unit(bad)
        except OF.AacError as e:
            return dec, res, e, before, dec.pns_state
        return dec, res, None, before, dec.pns_state
    # a damaged unit 5 whose failure comes AFTER it has drawn noise, so that the generator's position is part of the test
    bad = None
    for donor in u[5:]:
        full = attempt(donor)
        if full[2] is not None or full[3] == full[4]:
            continue                                     # this unit draws no noise
        if damage == "trailing_bits":
            cands = [donor + b"\x00\x80"]
        else:
            cands = [donor[:cut] for cut in range(len(donor) - 1, len(donor) // 2, -1)]
        for cand in cands:
            dec, want, err, before, after = attempt(cand)
            if err is not None and after != before and (damage == "trailing_bits" or after != full[4]):
                bad = cand
                break
        if bad is not None:
            break
    assert bad is not None, "the generator never produced a suitable unit"
    if damage == "trailing_bits":
        assert "trailing bits" in str(err)
    later = [dec.decode_access_unit(au) for au in u[9:14]]        # the oracle decoder carries on from where it stopped
    exc = type("E", (), {"value": err})
    sid = engine.open_stream(OF.RATES[sf_index], channels)
    try:
        got = engine.entropy_decode([(sid, 9)], u[:5] + [bad] + u[6:9])
        for i in range(5):
            assert got[i][0] == 0 and np.array_equal(got[i][1].view(np.uint32), want[i][0].view(np.uint32)), i
        assert ERR_NAMES[got[5][0]] == exc.value.kind and not got[5][1].any()
        for i in range(6, 9):
            assert got[i][0] == -199 and not got[i][1].any() and got[i][2] == [0, 0], (i, got[i][0])
        got = engine.entropy_decode([(sid, 5)], u[9:14])
        for i in range(5):
            assert got[i][0] == 0 and (got[i][2], got[i][3]) == (later[i][1], later[i][2])
            assert np.array_equal(got[i][1].view(np.uint32), later[i][0].view(np.uint32)), (damage, i)
    finally:
        engine.close_stream(sid)


@pytest.mark.gpu
def test_quantised_hand_over_equals_oracle_on_random_units(engine):
    """The generated units of every configuration (pulse data, every codebook, multi-filter TNS, grouping) through the
    quantised hand-over -- sk_aac_decoder_parse_q on the host, sk_aac_expand_q_decode on the device -- against the oracle,
    in two calls per stream; then the same with escape sequences of up to 16 extra bits, magnitudes an i16 cannot hold:
    they travel in the side record's list of wide values and come out as the oracle's."""
    import au_builder
    from soundkit_amd._lib import ERR_NAMES
    rates = OF.RATES
    for sizes, seed0 in ((None, 5000), ([0, 4, 9, 11, 12], 6000)):
        saved = au_builder.ESCAPE_SIZES
        if sizes:
            au_builder.ESCAPE_SIZES = sizes
        try:
            streams = []
            for k, (sf_index, channels) in enumerate(CONFIGS):
                for run in range(4):
                    u = make_stream(seed0 + 10 * k + run, sf_index, channels, 24)
                    want, err = oracle_decode(asc_for(sf_index, channels), u)
                    streams.append((engine.open_stream(rates[sf_index], channels), asc_for(sf_index, channels), u, want, err))
        finally:
            au_builder.ESCAPE_SIZES = saved
        wide_units = 0
        try:
            fes = [aac_lc.AacLcFrontEnd(asc) for _, asc, _, _, _ in streams]
            alive = [True] * len(streams)
            for first, last in ((0, 9), (9, 24)):
                table, parsed, owners = [], [], []
                for si, (sid, asc, u, want, err) in enumerate(streams):
                    mine = []
                    for i in range(first, last):
                        if not alive[si]:
                            break
                        try:
                            mine.append(fes[si].parse_q(u[i]))
                        except aac_lc.AacLcError as e:   # the host half rejects what fails up to the spectral data
                            assert i == len(want) and e.kind == err.kind and str(err) in str(e), (si, i, str(e), str(err))
                            alive[si] = False
                    if mine:
                        table.append((sid, len(mine)))
                        parsed += mine
                        owners += [(si, first + j) for j in range(len(mine))]
                got = engine.expand_q_decode(table, parsed)
                for (si, i), (q, side, _, _), (status, coeffs, seq, shape) in zip(owners, parsed, got):
                    sid, asc, u, want, err = streams[si]
                    wide_units += bool((q == -32768).any())
                    if i < len(want):
                        assert status == 0, (si, i, status)
                        assert (seq, shape) == (want[i][1], want[i][2]), (si, i)
                        assert np.array_equal(coeffs.view(np.uint32), want[i][0].view(np.uint32)), (si, i, np.abs(coeffs - want[i][0]).max())
                    elif i == len(want):
                        assert ERR_NAMES[status] == err.kind, (si, i, status, err.kind)
                        assert not coeffs.any()
                        alive[si] = False
            assert sum(len(w) for _, _, _, w, _ in streams) > 300
            if sizes:
                assert wide_units >= 10, wide_units
        finally:
            for sid, _, _, _, _ in streams:
                engine.close_stream(sid)
