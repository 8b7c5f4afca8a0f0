"""Mirror of soundkit-aac's `AacDecoder` (soundkit-aac/src/lib.rs:108-266): the `soundkit::audio_packet::Decoder`
surface (`decode_i16 / decode_i32 / decode_f32`, soundkit/src/audio_packet.rs:22-26) for an ADTS AAC-LC stream.
Each call appends its input, decodes every whole frame that is buffered and fits in the output, and returns the
number of interleaved samples written; 0 = needs more input / drained.  Entropy decode on the host, synthesis and
the i16 conversion in one batched GPU call per decode_i16 (csrc/adts_decoder.cpp)."""
import ctypes as C

import numpy as np

from ._lib import lib
from .engine import default_engine


class AacDecoder:
    def __init__(self, engine=None):
        self.engine = engine or default_engine()
        h = C.c_void_p()
        rc = lib.sk_adts_decoder_create(self.engine._h, C.byref(h))
        if rc != 0:
            raise RuntimeError(lib.sk_strerror(rc).decode())
        self._h = h

    @classmethod
    def new(cls, engine=None):
        return cls(engine)

    def init(self):  # lib.rs:129-131
        return None

    def close(self):
        if getattr(self, "_h", None):
            lib.sk_adts_decoder_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _info(self):
        rate, ch = C.c_uint32(), C.c_uint8()
        lib.sk_adts_decoder_info(self._h, C.byref(rate), C.byref(ch))
        return rate.value, ch.value

    def sample_rate(self):
        return self._info()[0] or None

    def channels(self):
        return self._info()[1] or None

    def _decode(self, fn, data, output):
        data = bytes(data)
        written = C.c_size_t()
        rc = fn(self._h, data if data else None, len(data), output.ctypes.data, output.size, C.byref(written))
        if rc != 0:
            raise ValueError(lib.sk_adts_decoder_last_error(self._h).decode() or lib.sk_strerror(rc).decode())  # Err(String)
        return written.value

    def decode_i16(self, data, output, fec=False):
        assert output.dtype == np.int16 and output.flags.c_contiguous
        return self._decode(lib.sk_adts_decoder_decode_i16, data, output)

    def decode_f32(self, data, output, fec=False):
        assert output.dtype == np.float32 and output.flags.c_contiguous
        return self._decode(lib.sk_adts_decoder_decode_f32, data, output)

    def decode_i32(self, data, output, fec=False):
        raise ValueError("Not implemented.")  # lib.rs:246-253


def decode_i16_with_drain(decoder, chunk, output):
    """soundkit-decoder lib.rs:2150-2181: one call with the data, then empty calls until 0; -> list of sample arrays"""
    results = []
    n = decoder.decode_i16(chunk, output)
    if n:
        results.append(output[:n].copy())
    while True:
        n = decoder.decode_i16(b"", output)
        if n == 0:
            break
        results.append(output[:n].copy())
    return results
