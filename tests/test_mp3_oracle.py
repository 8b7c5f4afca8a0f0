"""oracle/mp3_hybrid.py (ISO/IEC 11172-3 2.4.3.4 restated in f64) checked against properties the standard's filterbank
must have -- there is no reference-held vector for this path (nanomp3's source is absent; "parity unpinned"):
the IMDCT / window / overlap stage is the inverse of the forward MDCT with the same windows (time-domain alias
cancellation), through a whole normal -> start -> short -> stop -> normal sequence; the alias butterflies are rotations;
the polyphase stage equals its definition written out sample by sample."""
import numpy as np

from oracle import mp3_hybrid as M


def forward(block36, block_type):
    """the encoder's side (ISO 11172-3 C.1.5.3.3): windowed MDCT of a 36-sample block -> 18 lines (short: 3 x 6,
    interleaved), normalised by n / 4 as the standard's encoder does, so that the decoder's IMDCT needs no factor"""
    if block_type != 2:
        z = block36 * M.block_window(block_type)
        i = np.arange(36)[None, :]
        k = np.arange(18)[:, None]
        return (np.cos(np.pi / 72 * (2 * i + 1 + 18) * (2 * k + 1)) * z[None, :]).sum(axis=1) / 9.0
    out = np.zeros(18)
    win = M.block_window(2)
    for w in range(3):
        z = block36[6 * w + 6:6 * w + 18] * win
        i = np.arange(12)[None, :]
        k = np.arange(6)[:, None]
        out[w::3] = (np.cos(np.pi / 24 * (2 * i + 1 + 6) * (2 * k + 1)) * z[None, :]).sum(axis=1) / 3.0
    return out


def test_imdct_window_overlap_inverts_the_forward_transform():
    rng = np.random.default_rng(3)
    types = [0, 0, 1, 2, 3, 0, 0, 1, 2, 2, 3, 0]
    x = rng.standard_normal(18 * (len(types) + 1))
    overlap = np.zeros(18)
    got = []
    for g, bt in enumerate(types):
        lines = forward(x[18 * g:18 * g + 36], bt)
        raw = M.subband_block(lines, bt)
        got.append(raw[:18] + overlap)
        overlap = raw[18:]
    got = np.concatenate(got[1:])          # the first block has no predecessor to cancel its aliasing
    want = x[18:18 * len(types)]
    assert np.abs(got - want).max() < 1e-12   # perfect reconstruction: the aliasing of every block type cancels


def test_alias_butterflies_are_rotations():
    assert np.allclose(M.CS ** 2 + M.CA ** 2, 1.0, atol=1e-15)
    x = np.random.default_rng(4).standard_normal(576)
    for bt, mixed in ((0, 0), (1, 0), (3, 0), (2, 1)):
        y = M.alias_reduce(x, bt, mixed)
        assert abs(np.linalg.norm(y) - np.linalg.norm(x)) < 1e-12
    assert np.array_equal(M.alias_reduce(x, 2, 0), x)
    y = M.alias_reduce(x, 2, 1)
    assert np.array_equal(y[36:], x[36:]) and not np.array_equal(y[:36], x[:36])   # mixed: the first boundary only


def test_polyphase_equals_its_definition_written_out():
    rng = np.random.default_rng(5)
    d = M.synthetic_window(1)
    ch = M.Channel()
    slots = [rng.standard_normal(32) for _ in range(40)]
    history = []
    for t, s in enumerate(slots):
        got = ch.polyphase(s, d)
        history.insert(0, M.MATRIX @ s)   # V vectors, newest first
        want = np.zeros(32)
        for a in range(min(16, len(history))):
            v = history[a]
            part = v[:32] if a % 2 == 0 else v[32:]
            want += part * d[32 * a:32 * a + 32]
        assert np.abs(got - want).max() < 1e-12, t


def test_mixed_blocks_use_the_long_transform_in_the_two_lowest_subbands():
    rng = np.random.default_rng(6)
    xr = rng.standard_normal(576)
    a, b = M.Channel(), M.Channel()
    mixed = a.hybrid(xr, 2, 1)
    plain = b.hybrid(xr, 2, 0)
    assert not np.allclose(mixed[:2], plain[:2]) and np.allclose(mixed[3:], plain[3:])
