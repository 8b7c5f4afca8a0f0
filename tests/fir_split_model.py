#!/usr/bin/env python3
"""Host model of fir_bf16.hip's arithmetic: three-way bf16 split of samples and taps by truncation, the products kept
per 32-sample window, f32 accumulation after every window (the matrix instruction's internal sum taken as exact).
Prints the error against an f64 evaluation of the same filter for several product sets -- the budget behind kProducts.
No GPU needed:  python tests/fir_split_model.py  (test infrastructure: it uses the oracle's taps)"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))  # the repository root: oracle/
from oracle import oracle  # noqa: E402


def trunc_bf16(v):
    return (v.view(np.uint32) & np.uint32(0xFFFF0000)).view(np.float32)


def split3(v):
    a = trunc_bf16(v)
    r = (v - a).astype(np.float32)
    b = trunc_bf16(r)
    return a, b, (r - b).astype(np.float32)


SETS = {
    "60 MFMAs per tile (all six products everywhere)": [6] * 10,
    "41 (kProducts)": [1, 3, 6, 6, 6, 6, 6, 3, 3, 1],
    "39": [1, 3, 4, 6, 6, 6, 6, 3, 3, 1],
    "35": [1, 3, 4, 6, 6, 6, 4, 3, 1, 1],
}


def errors(sets=SETS, n=40000, verbose=False):
    """{label: (relative RMS error, max abs error)} of the split arithmetic against an f64 evaluation."""
    h = oracle.resampler_taps(16000 / 48000).astype(np.float32)
    peak = np.abs(h).max()
    if verbose:
        for s in range(10):
            lo, hi = max(0, 32 * s - 48), min(255, 32 * s + 28)
            print("window %d meets taps %3d..%3d, largest |h| / peak = 2^%.1f" % (s, lo, hi, np.log2(np.abs(h[lo:hi + 1]).max() / peak)))
    p6 = [(0, 0), (1, 0), (0, 1), (1, 1), (2, 0), (0, 2)]  # (tap piece, sample piece): x1h1 | x1h2 x2h1 | x2h2 x1h3 x3h1
    rng = np.random.default_rng(1)
    result = {}
    x = rng.uniform(-1, 1, n).astype(np.float32)
    hs, xs = split3(h), split3(x)
    tiles = (n - 600) // 48
    exact = None
    for label, count in sets.items():
        out = np.zeros((tiles, 16), np.float32)
        ex = np.zeros((tiles, 16))
        for i in range(16):
            acc = np.zeros(tiles, np.float32)
            for s in range(10):
                k = np.arange(32)
                p = 32 * s + k - 3 * i - 3
                ok = (p >= 0) & (p < 256)
                base = 48 * np.arange(tiles)[:, None] + 72 + 32 * s + k[None, :]
                for hp, xp in p6[:count[s]]:
                    hv = np.where(ok, hs[hp][np.clip(p, 0, 255)], 0).astype(np.float64)
                    acc = (acc + (xs[xp][base].astype(np.float64) * hv[None, :]).sum(1).astype(np.float32)).astype(np.float32)
                if exact is None:
                    hv = np.where(ok, h[np.clip(p, 0, 255)], 0).astype(np.float64)
                    ex[:, i] += (x[base].astype(np.float64) * hv[None, :]).sum(1)
            out[:, i] = acc
        if exact is None:
            exact = ex
        result[label] = (float(np.sqrt(np.mean((out - exact) ** 2) / np.mean(exact ** 2))), float(np.abs(out - exact).max()))
        if verbose:
            print("%-50s relative RMS error %.3e, max abs %.3e" % ((label,) + result[label]))
    return result


F16_SETS = {
    "f16, 30 MFMAs per tile (x1h1, x1h2, x2h1 everywhere)": [3] * 10,
    "f16, 24 (kProductsF16)": [1, 3, 3, 3, 3, 3, 3, 3, 1, 1],
    "f16, 40 (all four products)": [4] * 10,
}


def errors_f16(sets=F16_SETS, n=40000, amplitudes=(32767.0, 300.0, 3.0), verbose=False):
    """The s16-rows form on the f16 matrix instruction (fir_bf16.hip, F16): samples are 16-bit integers, taps times 2^16, both
    as two f16 values (samples toward zero, taps to nearest); products x1h1 | x1h2, x2h1 | x2h2; f32 accumulation per window; result * 2^-31.
    {(label, amplitude): (relative RMS error, max abs error)} against an f64 evaluation on s / 32768."""
    h = oracle.resampler_taps(16000 / 48000).astype(np.float32)
    hs1 = (h * np.float32(65536.0)).astype(np.float16)
    hs2 = ((h * np.float32(65536.0)) - hs1.astype(np.float32)).astype(np.float16)
    p4 = [(0, 0), (1, 0), (0, 1), (1, 1)]
    rng = np.random.default_rng(2)
    result = {}
    for amp in amplitudes:
        x = np.round(rng.uniform(-1, 1, n) * amp).astype(np.int32)
        x1 = x.astype(np.float32).astype(np.float16)                      # to nearest ...
        away = np.abs(x1.astype(np.int64)) > np.abs(x)
        x1 = np.where(away, np.nextafter(x1, np.float16(0)), x1).astype(np.float16)   # ... then toward zero, as v_cvt_pkrtz_f16_f32
        x2 = (x - x1.astype(np.int32)).astype(np.float16)
        assert np.array_equal(x1.astype(np.int64) + x2.astype(np.int64), x)
        hs, xs = (hs1, hs2), (x1, x2)
        tiles = (n - 600) // 48
        exact = None
        for label, count in sets.items():
            out = np.zeros((tiles, 16), np.float32)
            ex = np.zeros((tiles, 16))
            for i in range(16):
                acc = np.zeros(tiles, np.float32)
                for s in range(10):
                    k = np.arange(32)
                    p = 32 * s + k - 3 * i - 3
                    ok = (p >= 0) & (p < 256)
                    base = 48 * np.arange(tiles)[:, None] + 72 + 32 * s + k[None, :]
                    for hp, xp in p4[:count[s]]:
                        hv = np.where(ok, hs[hp][np.clip(p, 0, 255)].astype(np.float64), 0)
                        acc = (acc + (xs[xp][base].astype(np.float64) * hv[None, :]).sum(1).astype(np.float32)).astype(np.float32)
                    if exact is None:
                        hv = np.where(ok, h[np.clip(p, 0, 255)], 0).astype(np.float64)
                        ex[:, i] += ((x[base] / 32768.0) * hv[None, :]).sum(1)
                out[:, i] = acc * np.float32(2.0 ** -31)
            if exact is None:
                exact = ex
            result[(label, amp)] = (float(np.sqrt(np.mean((out - exact) ** 2) / np.mean(exact ** 2))), float(np.abs(out - exact).max()))
            if verbose:
                print("%-56s amplitude %7.0f: relative RMS error %.3e, max abs %.3e" % ((label, amp) + result[(label, amp)]))
    return result


if __name__ == "__main__":
    errors(verbose=True)
    errors_f16(verbose=True)
