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
