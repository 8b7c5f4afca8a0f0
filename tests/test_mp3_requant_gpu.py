"""csrc/mp3_requant.hip (Layer III requantisation, mid/side, MPEG-1 and 13818-3 intensity stereo, short-block reorder on gfx950,
behind sk_mp3_requantize) against oracle/mp3_bitstream.py requantize_granule (ISO/IEC 11172-3 2.4.3.4.7-9 in f64).
Tolerance: 1e-6 relative per line (north_star's float tolerance; for a joint-stereo pair relative to the larger of the
two channels at that line, since mid/side is a sum and a difference).  The scale-factor band tables are synthetic
partitions -- Table B.8 is not in this tree -- so these tests pin the arithmetic, not the tables ("parity unpinned")."""
import numpy as np
import pytest

from oracle import mp3_bitstream as ref
from soundkit_amd import mp3
from soundkit_amd._lib import SK_OK, SoundkitError

pytestmark = pytest.mark.gpu
UNSUPPORTED, INVALID = -303, -304


def partition(rng, parts, total, must_have):
    """a strictly increasing partition 0 .. total with `parts` intervals, containing `must_have`"""
    while True:
        inner = set(int(v) for v in rng.choice(np.arange(1, total), parts - 1, replace=False))
        if must_have is None or must_have in inner:
            return np.array([0] + sorted(inner) + [total], np.uint16)
        inner = sorted(inner)
        inner[int(rng.integers(0, len(inner)))] = must_have
        if len(set(inner)) == parts - 1:
            return np.array([0] + sorted(inner) + [total], np.uint16)


def tables(seed, mixed_boundary=True):
    rng = np.random.default_rng(seed)
    return (partition(rng, 22, 576, 36 if mixed_boundary else None), partition(rng, 13, 192, 12 if mixed_boundary else None),
            rng.integers(0, 5, 22).astype(np.uint8))


def random_channel(rng, block_type, mixed):
    return {"global_gain": int(rng.integers(0, 256)), "scalefac_scale": int(rng.integers(0, 2)), "preflag": int(rng.integers(0, 2)),
            "block_type": block_type, "mixed_block_flag": mixed, "subblock_gain": [int(v) for v in rng.integers(0, 8, 3)],
            "scalefac_l": [int(v) for v in rng.integers(0, 16, 21)] + [0], "scalefac_s": [[int(v) for v in rng.integers(0, 8, 3)] for _ in range(12)] + [[0, 0, 0]]}


def random_quant(rng, zero_from=576):
    q = rng.integers(-15, 16, 576)
    big = rng.random(576) < 0.05
    q[big] = rng.integers(-8206, 8207, int(big.sum()))
    q[rng.random(576) < 0.3] = 0
    q[zero_from:] = 0
    return q.astype(np.int16)


def check(granules, quant, xr, status, long_o, short_o, pretab):
    at = 0
    for i, g in enumerate(granules):
        ch = g["channels"]
        assert status[i] == SK_OK
        want = ref.requantize_granule(g, quant[at:at + ch], long_o, short_o, pretab)
        got = xr[at:at + ch].astype(np.float64)
        scale = np.abs(want)
        if ch == 2 and (g.get("ms_stereo") or g.get("intensity_stereo")):
            scale = np.broadcast_to(scale.max(axis=0), scale.shape)
        assert np.all(np.abs(got - want) <= 1e-6 * scale), (i, np.max(np.abs(got - want) / np.maximum(scale, 1e-300)))
        assert not got[want == 0].any()
        at += ch


def test_all_block_types_gains_and_scale_factors(engine):
    rng = np.random.default_rng(1)
    long_o, short_o, pretab = tables(2)
    assert mp3.set_band_tables(44100, long_o, short_o, pretab, engine) == SK_OK
    granules, quant = [], []
    for k in range(403):  # not a multiple of the four granules a block takes
        ch = 1 + (k % 2)
        g = {"sample_rate": 44100, "channels": ch, "ch": []}
        for _c in range(ch):
            bt = int(rng.integers(0, 4))
            g["ch"].append(random_channel(rng, bt, int(rng.integers(0, 2)) if bt == 2 else 0))
            quant.append(random_quant(rng))
        granules.append(g)
    quant = np.stack(quant)
    xr, status = mp3.requantize(granules, quant, engine)
    check(granules, quant, xr, status, long_o, short_o, pretab)


def test_known_values_and_the_reorder_permutation(engine):
    long_o, short_o, pretab = tables(3)
    assert mp3.set_band_tables(32000, long_o, short_o, pretab, engine) == SK_OK
    flat = {"global_gain": 210, "scalefac_scale": 0, "preflag": 0, "block_type": 0, "mixed_block_flag": 0, "subblock_gain": [0, 0, 0],
            "scalefac_l": [0] * 22, "scalefac_s": [[0, 0, 0]] * 13}
    q = np.zeros((3, 576), np.int16)
    q[0, :5] = [8, -27, 1, 0, 64]
    q[1] = q[2] = np.arange(576)
    granules = [{"sample_rate": 32000, "channels": 1, "ch": [dict(flat)]}, {"sample_rate": 32000, "channels": 1, "ch": [dict(flat, block_type=2)]},
                {"sample_rate": 32000, "channels": 1, "ch": [dict(flat, block_type=2, mixed_block_flag=1)]}]
    xr, status = mp3.requantize(granules, q, engine)
    assert not status.any()
    assert xr[0, :5].tolist() == [16.0, -81.0, 1.0, 0.0, 256.0]
    # |is|^(4/3) is monotone: undoing it shows where each bitstream-order line went
    back = np.rint(xr[1:].astype(np.float64) ** 0.75).astype(int)
    for row, mixed in ((back[0], 0), (back[1], 1)):
        assert sorted(row.tolist()) == list(range(576))
        for dest, src in enumerate(row):
            if mixed and src < 36:
                assert dest == src
                continue
            band = max(s for s in range(13) if 3 * short_o[s] <= src)
            width = int(short_o[band + 1] - short_o[band])
            w, j = divmod(src - 3 * int(short_o[band]), width)
            assert dest == 3 * (int(short_o[band]) + j) + w


@pytest.mark.parametrize("block_type", [0, 1, 2, 3])
def test_mid_side_and_intensity_stereo(engine, block_type):
    rng = np.random.default_rng(10 + block_type)
    long_o, short_o, pretab = tables(4)
    assert mp3.set_band_tables(48000, long_o, short_o, pretab, engine) == SK_OK
    granules, quant = [], []
    for k in range(120):
        ms, intensity = [(1, 0), (0, 1), (1, 1), (0, 0)][k % 4]
        mixed = int(block_type == 2 and k % 8 >= 4)  # mixed granules with and without intensity stereo
        left, right = random_channel(rng, block_type, mixed), random_channel(rng, block_type, mixed)
        # intensity positions live in the right channel's scale factors: 0..6 are positions, 7 says "not intensity coded"
        right["scalefac_l"] = [int(v) for v in rng.integers(0, 8, 21)] + [0]
        right["scalefac_s"] = [[int(v) for v in rng.integers(0, 8, 3)] for _ in range(12)] + [[0, 0, 0]]
        granules.append({"sample_rate": 48000, "channels": 2, "ms_stereo": ms, "intensity_stereo": intensity, "ch": [left, right]})
        quant.append(random_quant(rng))
        zero_from = [576, 0, int(rng.integers(0, 577)), int(rng.integers(0, 577))][k % 4 if k % 16 else 1]
        if k % 16 in (5, 6):
            # the right channel ends with the band below the last one: the last band is intensity coded and takes the default position
            # (the band below holds a scale factor, not a position), k % 16 == 6: ends one band lower -- the last takes that band's position
            back = 1 if k % 16 == 5 else 2
            zero_from = 3 * int(short_o[13 - back]) if block_type == 2 else int(long_o[22 - back])
        quant.append(random_quant(rng, zero_from))
    quant = np.stack(quant)
    xr, status = mp3.requantize(granules, quant, engine)
    check(granules, quant, xr, status, long_o, short_o, pretab)
    # intensity with an all-zero right channel and position 6 everywhere: everything goes left
    g = granules[1]
    g["ch"][1]["scalefac_l"] = [6] * 21 + [0]
    g["ch"][1]["scalefac_s"] = [[6, 6, 6]] * 12 + [[0, 0, 0]]
    q = np.stack([random_quant(rng), np.zeros(576, np.int16)])
    xr, status = mp3.requantize([g], q, engine)
    assert not status.any() and not xr[1].any()
    plain, _ = mp3.requantize([dict(g, intensity_stereo=0, ms_stereo=0)], q, engine)
    assert np.array_equal(xr[0], plain[0])


@pytest.mark.parametrize("block_type", [0, 2, 3])
def test_lsf_intensity_stereo(engine, block_type):
    """ISO/IEC 13818-3 2.4.3.2: positions up to 31 in the right channel's scale factors, ratios 1 : i0^n / i0^n : 1 with i0 = 2^-1/4 or
    2^-1/2 (intensity_stereo bit 1), bit 7 = "not intensity coded" (the field's largest value, marked by the host), with and
    without mid/side in the bands below the bound"""
    rng = np.random.default_rng(70 + block_type)
    long_o, short_o, pretab = tables(4)
    assert mp3.set_band_tables(24000, long_o, short_o, pretab, engine) == SK_OK
    granules, quant = [], []
    for k in range(96):
        ms, scale = k & 1, (k >> 1) & 1
        mixed = int(block_type == 2 and k % 8 >= 4)
        left, right = random_channel(rng, block_type, mixed), random_channel(rng, block_type, mixed)
        left["preflag"] = right["preflag"] = 0
        mark = lambda v: int(v) | (0x80 if rng.random() < 0.2 else 0)
        right["scalefac_l"] = [mark(v) for v in rng.integers(0, 32, 21)] + [0]
        right["scalefac_s"] = [[mark(v) for v in rng.integers(0, 32, 3)] for _ in range(12)] + [[0, 0, 0]]
        granules.append({"sample_rate": 24000, "channels": 2, "ms_stereo": ms, "intensity_stereo": 1 | (scale << 1), "lsf": 1, "ch": [left, right]})
        quant.append(random_quant(rng))
        quant.append(random_quant(rng, [0, int(rng.integers(0, 577)), int(rng.integers(0, 300)), 576][k % 4]))
    quant = np.stack(quant)
    xr, status = mp3.requantize(granules, quant, engine)
    check(granules, quant, xr, status, long_o, short_o, pretab)
    # position 0 everywhere, right channel empty: both channels carry the left channel's lines
    g = granules[0]
    g["ch"][1]["scalefac_l"], g["ch"][1]["scalefac_s"] = [0] * 22, [[0, 0, 0]] * 13
    g["ms_stereo"] = 0
    q = np.stack([random_quant(rng), np.zeros(576, np.int16)])
    xr, status = mp3.requantize([g], q, engine)
    assert not status.any() and np.array_equal(xr[0], xr[1]) and xr[0].any()


def test_rejected_granules_are_silent_and_do_not_disturb_the_others():
    import soundkit_amd
    engine = soundkit_amd.Engine(0, 4)  # its own: which rates have band tables is engine state
    try:
        rejected_granules(engine)
    finally:
        engine.close()


def rejected_granules(engine):
    rng = np.random.default_rng(30)
    long_o, short_o, pretab = tables(5)
    plain_long, plain_short, _ = tables(6, mixed_boundary=False)
    assert 36 not in plain_long.tolist()
    assert mp3.set_band_tables(22050, long_o, short_o, pretab, engine) == SK_OK
    assert mp3.set_band_tables(24000, plain_long, plain_short, pretab, engine) == SK_OK
    assert mp3.set_band_tables(12345, long_o, short_o, pretab, engine) == -6       # not an MPEG audio rate
    bad = long_o.copy()
    bad[5] = bad[4]
    assert mp3.set_band_tables(22050, bad, short_o, pretab, engine) == -1          # not a partition

    def granule(rate, bt=(0, 0), mixed=(0, 0), **kw):
        return dict({"sample_rate": rate, "channels": 2, "ch": [random_channel(rng, bt[0], mixed[0]), random_channel(rng, bt[1], mixed[1])]}, **kw)
    good = granule(22050)
    cases = [(good, SK_OK), (granule(8000), UNSUPPORTED),                                   # no table for that rate
             (granule(22050, bt=(2, 0), ms_stereo=1), INVALID),                             # a pair cut up differently
             (granule(22050, bt=(2, 0)), SK_OK),                                            # ... which is fine without joint stereo
             (granule(22050, intensity_stereo=1, lsf=1), SK_OK),                            # 13818-3 intensity, i0 = 2^-1/4
             (granule(22050, intensity_stereo=3, lsf=1, ms_stereo=1), SK_OK),               # ... i0 = 2^-1/2, mid/side below the bound
             (granule(22050, bt=(2, 2), mixed=(1, 1), intensity_stereo=1), SK_OK),          # intensity in a mixed granule
             (granule(24000, bt=(2, 2), mixed=(1, 1)), UNSUPPORTED),                        # tables without a boundary at line 36
             (granule(22050, bt=(1, 1), mixed=(1, 1)), INVALID), (good, SK_OK)]
    granules = [g for g, _ in cases]
    quant = np.stack([random_quant(rng) for _ in range(2 * len(cases))])
    quant[-2:] = quant[:2]
    xr, status = mp3.requantize(granules, quant, engine)
    assert status.tolist() == [s for _, s in cases]
    for i, (g, s) in enumerate(cases):
        if s != SK_OK:
            assert not xr[2 * i:2 * i + 2].any()
    ok = [i for i, (_, s) in enumerate(cases) if s == SK_OK]
    check([granules[i] for i in ok], np.concatenate([quant[2 * i:2 * i + 2] for i in ok]),
          np.concatenate([xr[2 * i:2 * i + 2] for i in ok]), np.zeros(len(ok), np.int32), long_o, short_o, pretab)
    assert np.array_equal(xr[:2], xr[-2:])
    three = dict(good, channels=3)
    with pytest.raises(SoundkitError) as exc:
        mp3.requantize([good, three], quant[:4], engine)
    assert exc.value.status == -1
    xr, status = mp3.requantize([], np.zeros((0, 576), np.int16), engine)
    assert xr.shape == (0, 576)


def test_requantised_lines_feed_the_hybrid_synthesis(engine):
    """the two GPU stages of the MP3 path in a row: what sk_mp3_requantize writes is what sk_mp3_hybrid_synthesize reads"""
    from oracle import mp3_hybrid as M
    rng = np.random.default_rng(40)
    window = M.synthetic_window(7)
    mp3.set_synthesis_window(window, engine)
    long_o, short_o, pretab = tables(7)
    assert mp3.set_band_tables(16000, long_o, short_o, pretab, engine) == SK_OK
    sid = engine.open_stream(16000, 2)
    try:
        seq = [(0, 0), (1, 0), (2, 0), (2, 1), (3, 0), (0, 0)]
        granules, quant = [], []
        for bt, mixed in seq:
            left, right = random_channel(rng, bt, mixed), random_channel(rng, bt, mixed)
            for ch in (left, right):
                ch["global_gain"] = int(rng.integers(120, 150))  # samples well inside +-1
            granules.append({"sample_rate": 16000, "channels": 2, "ms_stereo": 1, "lsf": 1, "ch": [left, right]})
            quant += [random_quant(rng), random_quant(rng)]
        quant = np.stack(quant)
        xr, status = mp3.requantize(granules, quant, engine)
        assert not status.any()
        pcm, status = mp3.hybrid_synthesize([(sid, 2, [bt, bt], [mx, mx]) for bt, mx in seq], xr.reshape(len(seq), 2, 576), engine)
        assert not status.any()
        check(granules, quant, xr, status, long_o, short_o, pretab)
        for c in range(2):
            chan = M.Channel()
            want = np.concatenate([chan.granule(xr[2 * i + c].astype(np.float64), seq[i][0], seq[i][1], window.astype(np.float32).astype(np.float64))
                                   for i in range(len(seq))])
            got = pcm[:, :, c].reshape(-1).astype(np.float64)
            assert np.sqrt(np.mean((got - want) ** 2)) <= 1e-6 * np.sqrt(np.mean(want ** 2))
            assert 1e-4 < np.abs(want).max() < 1.0
    finally:
        engine.close_stream(sid)
