"""The kernels' lane-level index arithmetic, re-enacted in numpy, against the oracle (no GPU)."""
import numpy as np

import kernel_model as K


def test_long_imdct_slot_mapping(oracle):
    x = oracle.seeded_spectrum(1024, 0xDEADBEEF)
    assert np.abs(K.long_imdct(x) - oracle.imdct_direct_f64(x)).max() < 1e-15
    pos = K.slot_positions()
    assert sorted(pos.ravel().tolist()) == list(range(1024))  # every sample owned exactly once


def test_eight_short_buffer(oracle):
    x = oracle.seeded_spectrum(1024, 0x12345678)
    ps, cs = oracle.sine_window(256).astype(float), oracle.kbd_window(256, 6.0).astype(float)
    buf = K.short_buffer(x, ps, cs)
    ch = oracle.Channel()
    out = ch.synthesize(x, oracle.EIGHT_SHORT, oracle.KBD)
    assert np.abs(out - buf[:1024]).max() < 1e-10 and np.abs(ch.delay - buf[1024:]).max() < 1e-10


def test_fir_mfma_schedule(oracle):
    taps = oracle.resampler_taps(1 / 3).astype(float)
    rng = np.random.default_rng(3)
    x = rng.uniform(-1, 1, (2, 700)).astype(np.float32)
    y = oracle.downsample_planar(x, 48000, 16000)
    ym = K.fir_model(x, taps, 0, y.shape[1])
    assert np.abs(ym - y).max() < 1e-6
