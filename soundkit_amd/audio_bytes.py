"""Mirror of soundkit::audio_bytes (soundkit/src/audio_bytes.rs:3-373), GPU-backed.

Same function names and argument meaning as the reference; byte inputs are anything that
views as uint8 (bytes, bytearray, numpy), sample inputs are numpy arrays.  Every call runs
the corresponding kernel of pcm.hip through the C ABI (sk_pcm_convert & friends).
"""
import numpy as np

from .engine import default_engine


def _bytes(data):
    if isinstance(data, (bytes, bytearray, memoryview)):
        return np.frombuffer(bytes(data), np.uint8)
    return np.ascontiguousarray(data).view(np.uint8).ravel()


def _require_multiple(data, k, what):
    # the reference asserts on these two (audio_bytes.rs:4-7, 26-29); the rest use chunks_exact
    if _bytes(data).size % k:
        raise AssertionError(what)


def _op(name, data):
    return default_engine().pcm_convert(name, _bytes(data))


def i16le_to_f32(data):
    _require_multiple(data, 2, "Bytes length must be a multiple of 2")
    return _op("I16LE_TO_F32", data)


def i16_to_i16le(samples):
    return _op("I16_TO_I16LE", np.ascontiguousarray(samples, np.int16)).view(np.uint8)


def i16le_to_i16(data):
    _require_multiple(data, 2, "Bytes length must be a multiple of 2")
    return _op("I16LE_TO_I16", data)


def s24le_to_i32(data):
    return _op("S24LE_TO_I32", data)


def s24le_to_i16(data):
    return _op("S24LE_TO_I16", data)


def s24be_to_i16(data):
    return _op("S24BE_TO_I16", data)


def s32le_to_i32(data):
    return _op("S32LE_TO_I32", data)


def s32be_to_i32(data):
    return _op("S32BE_TO_I32", data)


def s32le_to_s24(data):
    return _op("S32LE_TO_S24", data)


def s32be_to_s24(data):
    return _op("S32BE_TO_S24", data)


def s32le_to_f32(data):
    return _op("S32LE_TO_F32", data)


def s32be_to_f32(data):
    return _op("S32BE_TO_F32", data)


def s32le_to_i16(data):
    return _op("S32LE_TO_I16", data)


def s32be_to_i16(data):
    return _op("S32BE_TO_I16", data)


def f32le_to_i16(data):
    return _op("F32LE_TO_I16", data)


def f32be_to_i16(data):
    return _op("F32BE_TO_I16", data)


def f32le_to_i32(data):
    return _op("F32LE_TO_I32", data)


def f32le_to_s24(data):
    return _op("F32LE_TO_S24", data)


def s16be_to_i16(data):
    return _op("S16BE_TO_I16", data)


def s16le_to_i16(data):
    return _op("S16LE_TO_I16", data)


def s16le_to_i32(data):
    return _op("S16LE_TO_I32", data)


def interleave_vecs_i16(channels):
    return default_engine().interleave_i16(np.ascontiguousarray(channels, np.int16))


def deinterleave_vecs_i16(data, channel_count):
    return default_engine().deinterleave("i16", _bytes(data), channel_count)


def deinterleave_vecs_s24(data, channel_count):
    return default_engine().deinterleave("s24", _bytes(data), channel_count)


def deinterleave_vecs_f32(data, channel_count):
    return default_engine().deinterleave("f32", _bytes(data), channel_count)


def stereo_to_mono_take_left(samples):
    samples = np.ascontiguousarray(samples, np.int16)
    if samples.size % 2:
        raise AssertionError("Stereo buffer must contain an even number of samples")
    return _op("STEREO_TO_MONO_TAKE_LEFT", samples)


def stereo_to_mono_avg(samples):
    samples = np.ascontiguousarray(samples, np.int16)
    if samples.size % 2:
        raise AssertionError("Stereo buffer must contain an even number of samples")
    return _op("STEREO_TO_MONO_AVG", samples)


# the *_inplace_* forms (audio_bytes.rs:331, 360) return the leading `frames` samples
stereo_to_mono_inplace_take_left = stereo_to_mono_take_left
stereo_to_mono_inplace_avg = stereo_to_mono_avg
