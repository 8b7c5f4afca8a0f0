"""Random, syntactically valid access units (tests/au_builder.py: every tool, pulse data and escapes included) through
the oracle and through the product's front-ends: the host C++ (`sk_aac_decoder_parse`) here on the CPU, the gfx950
kernels (`sk_aac_entropy_decode`) under -m gpu.  Spectra bit for bit, window fields, and the PNS generator carried
from unit to unit of a stream."""
import numpy as np
import pytest

from au_builder import asc_for, random_access_unit
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
