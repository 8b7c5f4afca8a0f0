"""Mirror of the conversion/resample end of soundkit-decoder's worker (soundkit-decoder/src/lib.rs).

float_sample_to_i16 :1815, StreamingResampler :1917-2060, exact_signed_pcm_to_i16 :3458,
downmix_channels :3492 (mono), audio_data_to_f32_channels :3563, f32_channels_to_bytes :3619,
and the parts of apply_output_options :3324 that route between them.
"""
import numpy as np

from ._lib import SoundkitError
from .audio_types import AudioData, EncodingFlag, Endianness
from .engine import (FMT_F32BE, FMT_F32LE, FMT_S16BE, FMT_S16LE, FMT_S24BE, FMT_S24LE, FMT_S32BE, FMT_S32LE,
                     default_engine)

RESAMPLE_CHUNK_SIZE = 4096  # lib.rs:79


def _fmt_of(audio):
    le = audio.endianness == Endianness.LittleEndian
    if audio.audio_format == EncodingFlag.PCMFloat:
        if audio.bits_per_sample != 32:
            raise ValueError("floating-point PCM must contain 32-bit samples")
        return FMT_F32LE if le else FMT_F32BE
    table = {16: (FMT_S16LE, FMT_S16BE), 24: (FMT_S24LE, FMT_S24BE), 32: (FMT_S32LE, FMT_S32BE)}
    if audio.bits_per_sample not in table:
        raise ValueError("PCM data is unsupported or contains a partial frame")
    return table[audio.bits_per_sample][0 if le else 1]


def float_sample_to_i16(samples):
    return default_engine().pcm_convert("FLOAT_TO_I16_ROUND", np.ascontiguousarray(samples, np.float32))


def audio_data_to_f32_channels(audio):
    ch = audio.channel_count
    if ch == 0:
        raise ValueError("Channel count must be > 0")
    fmt = _fmt_of(audio)
    bps = (audio.bits_per_sample + 7) // 8
    if audio.data.size % bps or (audio.data.size // bps) % ch:
        raise ValueError("PCM data is unsupported or contains a partial frame")
    return default_engine().bytes_to_f32_planar(0, fmt, audio.data, ch)


def f32_channels_to_bytes(channels, bits_per_sample, output_format):
    channels = np.ascontiguousarray(channels, np.float32)
    if channels.size == 0:
        return np.zeros(0, np.uint8)
    if output_format == EncodingFlag.PCMFloat:
        if bits_per_sample != 32:
            raise ValueError("PCMFloat output requires 32-bit samples")
        fmt = FMT_F32LE
    else:
        fmt = {16: FMT_S16LE, 24: FMT_S24LE, 32: FMT_S32LE}.get(bits_per_sample)
        if fmt is None:
            raise ValueError("Unsupported output bits per sample: %d" % bits_per_sample)
    return default_engine().f32_planar_to_bytes(fmt, channels)


def downmix_channels(channels, target_channels):
    channels = np.ascontiguousarray(channels, np.float32)
    if channels.size == 0 or target_channels == 0:
        return np.zeros((0, 0), np.float32)
    if target_channels == 1:
        return default_engine().downmix_mono(channels)[None, :]
    if target_channels == 2 and channels.shape[0] > 2:
        raise SoundkitError(-6, "surround downmix")  # lib.rs:3512-3556: out of the hot-path scope
    return channels[:target_channels].copy()


def exact_signed_pcm_to_i16(audio):
    fmt = _fmt_of(audio)
    out = default_engine().exact_to_i16(fmt, audio.data)
    return AudioData(16, audio.channel_count, audio.sampling_rate, out, EncodingFlag.PCMSigned, Endianness.LittleEndian)


class StreamingResampler:
    """lib.rs:1917-2060: fixed 4096-frame chunks, resampler history kept on the GPU (48000 -> 16000 on
    the MFMA FIR, other pairs of common rates on the generic sinc kernel)."""

    def __init__(self, input_sample_rate, output_sample_rate, channels, engine=None):
        self.engine = engine or default_engine()
        self.channels = channels
        self.input_sample_rate, self.output_sample_rate = input_sample_rate, output_sample_rate
        self.stream = self.engine.open_stream(input_sample_rate, channels)
        self._fill = 0
        try:
            self.engine.resampler_open(self.stream, input_sample_rate, output_sample_rate)
        except SoundkitError:
            self.engine.close_stream(self.stream)
            raise

    def process(self, channels):
        channels = np.ascontiguousarray(channels, np.float32)
        if channels.shape[0] != self.channels:
            raise ValueError("Channel count changed mid-stream: expected %d, got %d" % (self.channels, channels.shape[0]))
        self._fill = (self._fill + channels.shape[1]) % RESAMPLE_CHUNK_SIZE
        return self.engine.resampler_process([self.stream], channels[None], self.channels)[0]

    def process_chunks(self, channels):
        """The reference's return shape (lib.rs:1970-2003): one entry per completed 4096-frame chunk."""
        channels = np.ascontiguousarray(channels, np.float32)
        out, pos = [], 0
        while pos < channels.shape[1]:
            n = min(channels.shape[1] - pos, RESAMPLE_CHUNK_SIZE - self._fill)
            got = self.process(channels[:, pos:pos + n])
            if got.shape[1]:
                out.append(got)
            pos += n
        return out

    def flush_chunks(self):
        got = self.flush()
        return [got] if got.shape[1] else []

    def flush(self):
        return self.engine.resampler_flush([self.stream], self.channels)[0]

    def close(self):
        if self.stream is not None:
            self.engine.close_stream(self.stream)
            self.stream = None


def apply_output_options(audio, output_bits_per_sample=None, output_sample_rate=None, output_channels=None,
                         resampler=None):
    """lib.rs:3324-3456.  Returns (list of AudioData, resampler)."""
    rate = output_sample_rate or audio.sampling_rate
    bits = output_bits_per_sample or audio.bits_per_sample
    chans = output_channels or audio.channel_count
    if rate == audio.sampling_rate and bits == audio.bits_per_sample and chans == audio.channel_count:
        return [audio], resampler
    if (rate == audio.sampling_rate and chans == audio.channel_count and bits == 16
            and audio.audio_format == EncodingFlag.PCMSigned and audio.bits_per_sample in (24, 32)):
        return [exact_signed_pcm_to_i16(audio)], resampler
    if bits not in (16, 24, 32):
        raise ValueError("Unsupported output bits per sample: %d" % bits)
    out_format = EncodingFlag.PCMFloat if (bits == 32 and audio.audio_format == EncodingFlag.PCMFloat) else EncodingFlag.PCMSigned
    channels = audio_data_to_f32_channels(audio)
    if rate != audio.sampling_rate:
        if resampler is None:
            resampler = StreamingResampler(audio.sampling_rate, rate, channels.shape[0])
        elif (resampler.input_sample_rate, resampler.channels, resampler.output_sample_rate) != (
                audio.sampling_rate, channels.shape[0], rate):
            raise ValueError("Resampler configuration changed mid-stream")
        return emit_resampled_chunks(resampler.process_chunks(channels), bits, chans, rate, out_format), resampler
    return emit_resampled_chunks([channels], bits, chans, rate, out_format), resampler


def emit_resampled_chunks(chunks, bits, chans, rate, out_format):
    """lib.rs:3261-3290: downmix + byte conversion, one AudioData per chunk."""
    out = []
    for channels in chunks:
        out_ch = channels.shape[0]
        if chans < out_ch:
            channels = downmix_channels(channels, chans)
            out_ch = chans
        data = f32_channels_to_bytes(channels, bits, out_format)
        out.append(AudioData(bits, out_ch, rate, data, out_format, Endianness.LittleEndian))
    return out


def flush_resampler_frames(resampler, bits, chans, out_format=EncodingFlag.PCMSigned):
    """lib.rs:3292-3305"""
    return emit_resampled_chunks(resampler.flush_chunks(), bits, chans, resampler.output_sample_rate, out_format)
