#!/usr/bin/env python3
"""Generates the committed golden vectors under tests/golden/ from the CPU oracle.

The reference (Rust) cannot run in the build container, so these vectors are outputs of the
oracle restatement (oracle/sk_oracle.c), which tests/test_oracle_pins.py pins against the
reference's own in-file known answers.  Inputs are the reference's test LCG
(soundkit-aac-lc/src/dsp.rs:725-738) and seeded numpy generators; everything needed to
regenerate is in this file:  python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from oracle import oracle as O  # noqa: E402

SEQS = [[0, 0], [1, 0], [2, 1], [2, 2], [3, 2], [0, 3], [0, 0], [1, 1]]     # per frame: [left, right]
SHAPES = [[0, 1], [0, 1], [1, 1], [1, 0], [0, 0], [1, 0], [1, 1], [0, 0]]


def aac():
    frames, ch = len(SEQS), 2
    coeffs = np.stack([[O.seeded_spectrum(1024, (0x12345678 + 0x9E3779B9 * (2 * f + c)) & 0xFFFFFFFF) * np.float32(900.0)
                        for c in range(ch)] for f in range(frames)])
    pcm, chans = O.synthesize_stream(coeffs, SEQS, SHAPES)
    s16 = np.stack([O.planar_f32_to_s16_interleaved(pcm[f]).reshape(1024, ch) for f in range(frames)])
    np.savez_compressed(os.path.join(HERE, "aac_synth.npz"), coeffs=coeffs, seqs=np.array(SEQS, np.uint8),
                        shapes=np.array(SHAPES, np.uint8), pcm=pcm, s16=s16,
                        delay=np.stack([c.delay for c in chans]),
                        prev_shape=np.array([c.prev_shape for c in chans], np.uint8))


def fir():
    rng = np.random.default_rng(20240601)
    x = rng.uniform(-1, 1, (3, 6000)).astype(np.float32)
    x[2] = (0.5 * np.sin(2 * np.pi * 440.0 * np.arange(6000) / 48000.0)).astype(np.float32)
    y = O.downsample_planar(x, 48000, 16000)
    np.savez_compressed(os.path.join(HERE, "fir_48k_16k.npz"), x=x, y=y, taps=O.resampler_taps(16000 / 48000))


def pcm():
    rng = np.random.default_rng(7)
    out = {}
    edge = np.array([0.0, -0.0, 1.0, -1.0, 0.5, -0.5, 0.99999994, -0.99999994, 1.0000001, 2.0, -2.0, np.nan, np.inf,
                     -np.inf, 1e-8, 3.0517578e-05, 0.25, -0.25, 1e30], np.float32)
    for op, name in enumerate(O.OPS):
        ib = O.lib().sko_op_in_bytes(op)
        n = 257
        if name.startswith(("F32", "VEC_F32", "FLOAT_", "MP3_")):
            x = rng.uniform(-1.3, 1.3, n).astype(np.float32)
            x[:edge.size] = edge
            raw = (x.byteswap() if "BE" in name else x).view(np.uint8)
        else:
            raw = rng.integers(0, 256, n * ib, dtype=np.uint8)
        out["in_" + name] = raw
        out["out_" + name] = O.pcm_convert(op, raw, n)
    np.savez_compressed(os.path.join(HERE, "pcm_ops.npz"), **out)


if __name__ == "__main__":
    aac()
    fir()
    pcm()
    print("wrote", sorted(f for f in os.listdir(HERE) if f.endswith(".npz")))
