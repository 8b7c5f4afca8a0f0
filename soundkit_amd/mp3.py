"""Host-side mirror of the transform half of soundkit-mp3's decoder (soundkit-mp3/src/lib.rs:147-374): the Layer III
hybrid synthesis filterbank that `nanomp3::Decoder::decode` ends with, batched on the GPU (csrc/mp3_hybrid.hip), and the
reference's `f32_to_i16` tail.  The bitstream side and the standard's synthesis window D are not part of this tree
(include/soundkit_amd.h); the window is supplied by the caller."""
import ctypes as C

import numpy as np

from ._lib import Mp3GranuleDesc, check, lib
from .engine import _ptr, default_engine


def set_synthesis_window(d512, engine=None):
    """ISO/IEC 11172-3 Table B.3, 512 coefficients, once per engine"""
    engine = engine or default_engine()
    d = np.ascontiguousarray(d512, np.float32)
    assert d.shape == (512,)
    check(lib.sk_mp3_set_synthesis_window(engine._h, _ptr(d)), "sk_mp3_set_synthesis_window", engine._h)


def make_descs(granules):
    """granules: iterable of (stream, channels, (block_type...), (mixed_block_flag...))"""
    granules = list(granules)
    arr = (Mp3GranuleDesc * max(len(granules), 1))()
    for i, (stream, ch, block_types, mixed) in enumerate(granules):
        arr[i].stream, arr[i].channels = stream, ch
        for c in range(min(ch, 2)):  # a count outside 1..2 is the engine's to reject
            arr[i].block_type[c] = block_types[c]
            arr[i].mixed_block_flag[c] = mixed[c]
    return arr, len(granules)


def hybrid_synthesize(granules, xr, engine=None, s16=False):
    """xr [n][channels][576] f32 -> (pcm [n][576][channels] f32 in +-1.0, or s16 through f32_to_i16; status [n])"""
    engine = engine or default_engine()
    descs, n = make_descs(granules)
    xr = np.ascontiguousarray(xr, np.float32)
    ch = xr.shape[1] if xr.ndim == 3 else 1
    out = np.zeros((n, 576, ch), np.int16 if s16 else np.float32)
    status = np.zeros(max(n, 1), np.int32)
    fn = lib.sk_mp3_hybrid_synthesize_s16 if s16 else lib.sk_mp3_hybrid_synthesize_f32
    check(fn(engine._h, descs, _ptr(xr), _ptr(out), n, _ptr(status)), "sk_mp3_hybrid_synthesize", engine._h)
    return out, status[:n]
