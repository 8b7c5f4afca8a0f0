"""Mirror of the hot-path parts of soundkit::audio_pipeline (soundkit/src/audio_pipeline.rs).

vec_f32_to_i16 :17, vec_i16_to_f32 :29, vec_i32_to_f32 :40, audio_to_f32_channels :74,
downsample_audio :438 -- GPU-backed through the C ABI.
"""
import numpy as np

from .audio_types import AudioData, EncodingFlag
from .engine import FMT_F32LE, FMT_S16LE, FMT_S24LE, FMT_S32LE, default_engine

COMMON_SAMPLE_RATES = [8000, 16000, 22050, 24000, 32000, 44100, 48000, 88200, 96000]  # :12-13
COMMON_BITS_PER_SAMPLE = [16, 24, 32]  # :15


def vec_f32_to_i16(values):
    return default_engine().pcm_convert("VEC_F32_TO_I16", np.ascontiguousarray(values, np.float32))


def vec_i16_to_f32(values):
    return default_engine().pcm_convert("VEC_I16_TO_F32", np.ascontiguousarray(values, np.int16))


def vec_i32_to_f32(values):
    return default_engine().pcm_convert("VEC_I32_TO_F32", np.ascontiguousarray(values, np.int32))


def audio_to_f32_channels(audio):
    """audio_pipeline.rs:74-98 -> [channels][frames] f32."""
    ch = audio.channel_count
    if ch == 0:
        raise ValueError("Channel count must be > 0")
    bits = audio.bits_per_sample
    if bits == 32 and audio.audio_format != EncodingFlag.PCMFloat:
        fmt = FMT_S32LE
    elif bits == 16:
        fmt = FMT_S16LE
    elif bits == 24:
        fmt = FMT_S24LE
    elif bits == 32:
        fmt = FMT_F32LE
    else:
        raise ValueError("deserialize_audio failed: unsuporrted type")
    return default_engine().bytes_to_f32_planar(1, fmt, audio.data, ch)


def downsample_audio(audio, sampling_rate):
    """audio_pipeline.rs:438-493: 48000 -> 16000 runs on the MFMA FIR, every other pair of common
    rates on the generic sinc kernel."""
    if audio.channel_count == 0:
        raise ValueError("Channel count must be > 0")
    if audio.bits_per_sample not in COMMON_BITS_PER_SAMPLE:
        raise ValueError("Unsupported bits_per_sample: %d" % audio.bits_per_sample)
    if audio.sampling_rate == 0 or sampling_rate == 0:
        raise ValueError("sampling_rate must be > 0")
    if audio.sampling_rate not in COMMON_SAMPLE_RATES:
        raise ValueError("Unsupported input sample_rate: %d" % audio.sampling_rate)
    if sampling_rate not in COMMON_SAMPLE_RATES:
        raise ValueError("Unsupported output sample_rate: %d" % sampling_rate)
    data = audio_to_f32_channels(audio)
    if data.size == 0:
        return np.zeros((audio.channel_count, 0), np.float32)
    return default_engine().downsample(data, audio.sampling_rate, sampling_rate)
