"""Access-unit-shaped handle onto the batched synthesis engine.

Mirrors the synthesis tail of soundkit-aac-lc's AacLcDecoder (decoder.rs:46-391): one object
per stream, created from the channel count, fed one frame of dequantised spectra at a time
(what decoder.rs:336-374 synthesize_channel consumes) and returning PlanarF32
(decoder.rs:22-36).  Many handles share one Engine; `synthesize_batch` is the same call for
a whole batch of streams.
"""
import numpy as np

from .engine import Engine, default_engine, descs_from_arrays, make_descs


class AacLcFrame:  # decoder.rs:38-43
    def __init__(self, sample_rate, channels, frames=1024):
        self.sample_rate, self.channels, self.frames = sample_rate, channels, frames


class PlanarF32:  # decoder.rs:22-36
    def __init__(self, channels):
        self._channels = channels

    def channels(self):
        return self._channels

    def frames(self):
        return self._channels.shape[1]


class AacLcSynth:
    def __init__(self, sample_rate=48000, channels=2, engine=None):
        self.engine = engine or default_engine()
        self._info = AacLcFrame(sample_rate, channels)
        self.stream = self.engine.open_stream(sample_rate, channels)

    def frame_info(self):  # decoder.rs:88-94
        return self._info

    def close(self):
        if self.stream is not None:
            self.engine.close_stream(self.stream)
            self.stream = None

    def reset(self):
        self.engine.reset_stream(self.stream)

    def synthesize(self, coeffs, window_sequence, window_shape):
        """coeffs [channels][1024]; window_sequence / window_shape per channel -> PlanarF32."""
        ch = self._info.channels
        coeffs = np.ascontiguousarray(coeffs, np.float32).reshape(ch, 1024)
        seqs = list(window_sequence) + [0] * (2 - ch)
        shapes = list(window_shape) + [0] * (2 - ch)
        descs, n = make_descs([(self.stream, ch, seqs, shapes)])
        pcm, status = self.engine.aac_synthesize(descs, n, coeffs)
        if status[0] != 0:
            raise ValueError("invalid AAC config: frame rejected with status %d" % status[0])
        return PlanarF32(pcm.reshape(ch, 1024))

    def synthesize_s16(self, coeffs, window_sequence, window_shape):
        """decode_aac_access_unit's output (soundkit-decoder lib.rs:1793-1813): interleaved i16."""
        ch = self._info.channels
        coeffs = np.ascontiguousarray(coeffs, np.float32).reshape(ch, 1024)
        seqs = list(window_sequence) + [0] * (2 - ch)
        shapes = list(window_shape) + [0] * (2 - ch)
        descs, n = make_descs([(self.stream, ch, seqs, shapes)])
        pcm, status = self.engine.aac_synthesize(descs, n, coeffs, out="s16")
        if status[0] != 0:
            raise ValueError("invalid AAC config: frame rejected with status %d" % status[0])
        return pcm


def synthesize_batch(engine: Engine, streams, channels, coeffs, seqs, shapes, out="f32"):
    """streams [n] ids; coeffs [n][channels][1024]; seqs/shapes [n][2] -> pcm, status."""
    descs, n = descs_from_arrays(streams, channels, seqs, shapes)
    pcm, status = engine.aac_synthesize(descs, n, coeffs, out=out)
    if out == "f32":
        return pcm.reshape(n, channels, 1024), status
    return pcm.reshape(n, 1024, channels), status


# ---- access-unit decoder: host entropy front-end + GPU synthesis ------------------------------------

class AacLcError(Exception):
    """Mirror of soundkit-aac-lc's AacLcError (error.rs:5-18): .kind is the variant name."""

    def __init__(self, status, message):
        from ._lib import ERR_NAMES
        self.status = status
        self.kind = ERR_NAMES.get(status, str(status))
        super().__init__("%s: %s" % (self.kind, message))


def parse_adts_access_unit(data):
    """soundkit-decoder lib.rs:1007-1027: (asc bytes, raw access unit, frame length) of one ADTS frame."""
    import ctypes as C
    from ._lib import lib
    buf = np.frombuffer(bytes(data), np.uint8)
    frame_len, off, plen = C.c_size_t(), C.c_size_t(), C.c_size_t()
    asc = (C.c_uint8 * 2)()
    rc = lib.sk_adts_parse(buf.ctypes.data, buf.size, C.byref(frame_len), C.byref(off), C.byref(plen), asc)
    if rc != 0:
        raise ValueError("invalid ADTS access unit")
    if frame_len.value > buf.size:
        raise ValueError("truncated ADTS header")
    return bytes(asc), bytes(buf[off.value:frame_len.value]), frame_len.value


def split_adts(data):
    """All (asc, access unit) pairs of an ADTS byte stream."""
    data = bytes(data)
    pos, out = 0, []
    while pos + 7 <= len(data):
        asc, au, flen = parse_adts_access_unit(data[pos:])
        out.append((asc, au))
        pos += flen
    return out


class AacLcFrontEnd:
    """The entropy half of AacLcDecoder (decoder.rs:104-334) on host cores: raw access unit ->
    dequantised spectra + window fields.  No GPU involved."""

    def __init__(self, asc):
        import ctypes as C
        from ._lib import lib
        self._lib = lib
        buf = np.frombuffer(bytes(asc), np.uint8)
        h = C.c_void_p()
        rc = lib.sk_aac_decoder_create(buf.ctypes.data, buf.size, C.byref(h))
        if rc != 0:
            raise AacLcError(rc, lib.sk_strerror(rc).decode())
        self._h = h
        rate, ch = C.c_uint32(), C.c_uint8()
        lib.sk_aac_decoder_info(h, C.byref(rate), C.byref(ch))
        self.sample_rate, self.channels = rate.value, ch.value

    def close(self):
        if getattr(self, "_h", None):
            self._lib.sk_aac_decoder_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def tool_usage(self):
        out = np.zeros(8, np.uint32)
        self._lib.sk_aac_decoder_tool_usage(self._h, out.ctypes.data)
        keys = ("frames", "short", "transition", "tns", "pns_bands", "is_bands", "ms_bands", "pulse")
        return dict(zip(keys, out.tolist()))

    def parse(self, access_unit):
        """-> (coeffs [channels][1024] f32, window_sequence [channels], window_shape [channels])"""
        import ctypes as C
        from ._lib import FrameDesc
        buf = np.frombuffer(bytes(access_unit), np.uint8)
        coeffs = np.zeros((self.channels, 1024), np.float32)
        desc = FrameDesc()
        rc = self._lib.sk_aac_decoder_parse(self._h, buf.ctypes.data if buf.size else None, buf.size, coeffs.ctypes.data,
                                            C.byref(desc))
        if rc != 0:
            raise AacLcError(rc, self._lib.sk_aac_decoder_last_error(self._h).decode())
        return coeffs, list(desc.window_sequence)[:self.channels], list(desc.window_shape)[:self.channels]


    def parse_q(self, access_unit):
        """sk_aac_decoder_parse_q: the Huffman half alone -> (quant [channels][1024] i16, side record bytes,
        window_sequence [channels], window_shape [channels]); the rest runs on the device (Engine.tick_run_q)"""
        import ctypes as C
        from ._lib import AAC_UNIT_SIDE_BYTES, FrameDesc
        buf = np.frombuffer(bytes(access_unit), np.uint8)
        quant = np.zeros((self.channels, 1024), np.int16)
        side = np.zeros(AAC_UNIT_SIDE_BYTES, np.uint8)
        desc = FrameDesc()
        rc = self._lib.sk_aac_decoder_parse_q(self._h, buf.ctypes.data if buf.size else None, buf.size, quant.ctypes.data,
                                              side.ctypes.data, C.byref(desc))
        if rc != 0:
            raise AacLcError(rc, self._lib.sk_aac_decoder_last_error(self._h).decode())
        return quant, side, list(desc.window_sequence)[:self.channels], list(desc.window_shape)[:self.channels]


class AacLcDecoder:
    """Drop-in shape of soundkit_aac_lc::AacLcDecoder: from_audio_specific_config / decode_access_unit /
    frame_info (decoder.rs:46-164).  Entropy decode on the host, IMDCT + window + overlap-add on the GPU."""

    def __init__(self, asc, engine=None):
        self.front = AacLcFrontEnd(asc)
        self.synth = AacLcSynth(self.front.sample_rate, self.front.channels, engine)

    @classmethod
    def from_audio_specific_config(cls, asc, engine=None):
        return cls(asc, engine)

    def frame_info(self):
        return self.synth.frame_info()

    def decode_access_unit(self, data):
        coeffs, seqs, shapes = self.front.parse(data)
        return self.synth.synthesize(coeffs, seqs, shapes)

    def decode_access_unit_s16(self, data):
        """decode_aac_access_unit (soundkit-decoder lib.rs:1793-1813): interleaved i16."""
        coeffs, seqs, shapes = self.front.parse(data)
        return self.synth.synthesize_s16(coeffs, seqs, shapes)

    def close(self):
        self.front.close()
        self.synth.close()
