"""ctypes binding of the CPU oracle (oracle/libsk_oracle.so).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg.  The product package (soundkit_amd) never imports this module.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# SK_ORACLE_LIB: another build of the same sk_oracle.c (bench.py's cpu_baseline uses one compiled for the host it runs on)
_LIB_PATH = os.environ.get("SK_ORACLE_LIB") or os.path.join(_HERE, "libsk_oracle.so")

ONLY_LONG, LONG_START, EIGHT_SHORT, LONG_STOP = 0, 1, 2, 3
SINE, KBD = 0, 1

FMT_S16LE, FMT_S16BE, FMT_S24LE, FMT_S24BE, FMT_S32LE, FMT_S32BE, FMT_F32LE, FMT_F32BE = range(8)

OPS = [
    "I16LE_TO_F32", "I16_TO_I16LE", "I16LE_TO_I16", "S24LE_TO_I32", "S24LE_TO_I16", "S24BE_TO_I16",
    "S32LE_TO_I32", "S32BE_TO_I32", "S32LE_TO_S24", "S32BE_TO_S24", "S32LE_TO_F32", "S32BE_TO_F32",
    "S32LE_TO_I16", "S32BE_TO_I16", "F32LE_TO_I16", "F32BE_TO_I16", "F32LE_TO_I32", "F32LE_TO_S24",
    "S16BE_TO_I16", "S16LE_TO_I16", "S16LE_TO_I32", "STEREO_TO_MONO_TAKE_LEFT", "STEREO_TO_MONO_AVG",
    "VEC_F32_TO_I16", "VEC_I16_TO_F32", "VEC_I32_TO_F32", "FLOAT_TO_I16_ROUND", "MP3_F32_TO_I16",
    "MP3_F32_TO_I32",
]
OP = {name: i for i, name in enumerate(OPS)}


def build_native(out_dir):
    """sk_oracle.c compiled for the cores of THIS host (-O3 -march=native; -ffp-contract=off stays, so every f32 result
    is the portable build's) into out_dir; returns the path, or None when no compiler is available."""
    out = os.path.join(out_dir, "libsk_oracle_native.so")
    cmd = [os.environ.get("CC", "gcc"), "-O3", "-march=native", "-fPIC", "-std=c11", "-ffp-contract=off", "-fno-fast-math",
           "-D_GNU_SOURCE", "-shared", "-o", out, os.path.join(_HERE, "sk_oracle.c"), "-lm"]
    try:
        subprocess.check_call(cmd, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    except (OSError, subprocess.CalledProcessError):
        return None
    return out


def build(force=False):
    if os.environ.get("SK_ORACLE_LIB"):
        return _LIB_PATH
    if force or not os.path.exists(_LIB_PATH) or os.path.getmtime(_LIB_PATH) < os.path.getmtime(
            os.path.join(_HERE, "sk_oracle.c")):
        subprocess.check_call(["make", "-C", _HERE, "-s"] + (["-B"] if force else []))
    return _LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_LIB_PATH)
        fp, dp, vp = C.POINTER(C.c_float), C.POINTER(C.c_double), C.c_void_p
        L.sko_imdct_direct_f32.argtypes = [fp, fp, C.c_int]
        L.sko_imdct_direct_f64.argtypes = [fp, dp, C.c_int]
        L.sko_imdct_fast.argtypes = [fp, fp, C.c_int]
        L.sko_imdct_fast.restype = C.c_int
        L.sko_sine_window.argtypes = [C.c_int, fp]
        L.sko_kbd_window.argtypes = [C.c_int, C.c_float, fp]
        L.sko_channel_init.argtypes = [vp]
        L.sko_synthesize_channel.argtypes = [vp, fp, C.c_int, C.c_int, fp]
        L.sko_synthesize_channel.restype = C.c_int
        L.sko_pow43.argtypes = [C.c_uint32]
        L.sko_pow43.restype = C.c_float
        L.sko_scalefactor_multiplier.argtypes = [C.c_int]
        L.sko_scalefactor_multiplier.restype = C.c_float
        L.sko_dequantize_signed.argtypes = [C.c_int32, C.c_int]
        L.sko_dequantize_signed.restype = C.c_float
        L.sko_seeded_spectrum.argtypes = [C.c_int, C.c_uint32, fp]
        L.sko_pcm_stats_from.argtypes = [fp, C.c_size_t, vp]
        L.sko_float_sample_to_i16.argtypes = [C.c_float]
        L.sko_float_sample_to_i16.restype = C.c_int16
        L.sko_mp3_f32_to_i16.argtypes = [C.c_float]
        L.sko_mp3_f32_to_i16.restype = C.c_int16
        L.sko_mp3_f32_to_i32.argtypes = [C.c_float]
        L.sko_mp3_f32_to_i32.restype = C.c_int32
        L.sko_op_in_bytes.argtypes = [C.c_int]
        L.sko_op_out_bytes.argtypes = [C.c_int]
        L.sko_pcm_convert.argtypes = [C.c_int, vp, vp, C.c_size_t]
        L.sko_pcm_convert.restype = C.c_int
        for name in ("sko_interleave_i16", "sko_deinterleave_i16", "sko_deinterleave_s24", "sko_deinterleave_f32",
                     "sko_interleave_f32"):
            getattr(L, name).argtypes = [vp, C.c_size_t, C.c_int, vp]
        L.sko_fmt_bytes.argtypes = [C.c_int]
        L.sko_decoder_bytes_to_f32_planar.argtypes = [C.c_int, vp, C.c_size_t, C.c_int, vp]
        L.sko_core_bytes_to_f32_planar.argtypes = [C.c_int, vp, C.c_size_t, C.c_int, vp]
        L.sko_f32_planar_to_bytes.argtypes = [C.c_int, vp, C.c_size_t, C.c_int, vp]
        L.sko_downmix_mono.argtypes = [vp, C.c_size_t, C.c_int, vp]
        L.sko_exact_signed_pcm_to_i16.argtypes = [C.c_int, vp, C.c_size_t, vp]
        L.sko_planar_f32_to_s16_interleaved.argtypes = [vp, C.c_size_t, C.c_int, vp]
        L.sko_resampler_new.argtypes = [C.c_double, C.c_size_t, C.c_int]
        L.sko_resampler_new.restype = vp
        L.sko_resampler_free.argtypes = [vp]
        L.sko_resampler_output_frames_max.argtypes = [vp]
        L.sko_resampler_output_frames_max.restype = C.c_size_t
        L.sko_resampler_process.argtypes = [vp, vp, C.c_size_t, vp, C.c_size_t]
        L.sko_resampler_process.restype = C.c_size_t
        L.sko_resampler_process_partial.argtypes = [vp, vp, C.c_size_t, C.c_size_t, vp, C.c_size_t]
        L.sko_resampler_process_partial.restype = C.c_size_t
        L.sko_resampler_taps_phase0.argtypes = [vp, vp]
        L.sko_downsample_out_max.argtypes = [C.c_size_t, C.c_uint32, C.c_uint32]
        L.sko_downsample_out_max.restype = C.c_size_t
        L.sko_downsample_planar.argtypes = [vp, C.c_size_t, C.c_int, C.c_uint32, C.c_uint32, vp, C.c_size_t]
        L.sko_downsample_planar.restype = C.c_size_t
        L.sko_streaming_new.argtypes = [C.c_uint32, C.c_uint32, C.c_int]
        L.sko_streaming_new.restype = vp
        L.sko_streaming_free.argtypes = [vp]
        L.sko_streaming_process.argtypes = [vp, vp, C.c_size_t, C.c_size_t, vp, C.c_size_t, C.c_size_t]
        L.sko_streaming_process.restype = C.c_size_t
        L.sko_streaming_flush.argtypes = [vp, vp, C.c_size_t, C.c_size_t]
        L.sko_streaming_flush.restype = C.c_size_t
        _lib = L
    return _lib


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _fp(a):
    return a.ctypes.data_as(C.POINTER(C.c_float))


def _vp(a):
    return a.ctypes.data_as(C.c_void_p)


def imdct_direct_f32(x):
    x = _f32(x)
    out = np.empty(2 * x.size, np.float32)
    lib().sko_imdct_direct_f32(_fp(x), _fp(out), x.size)
    return out


def imdct_direct_f64(x):
    x = _f32(x)
    out = np.empty(2 * x.size, np.float64)
    lib().sko_imdct_direct_f64(_fp(x), out.ctypes.data_as(C.POINTER(C.c_double)), x.size)
    return out


def imdct_fast(x):
    x = _f32(x)
    out = np.empty(2 * x.size, np.float32)
    assert lib().sko_imdct_fast(_fp(x), _fp(out), x.size) == 0
    return out


def sine_window(n):
    out = np.empty(n, np.float32)
    lib().sko_sine_window(n, _fp(out))
    return out


def kbd_window(n, alpha):
    out = np.empty(n, np.float32)
    lib().sko_kbd_window(n, C.c_float(alpha), _fp(out))
    return out


def seeded_spectrum(n, seed):
    out = np.empty(n, np.float32)
    lib().sko_seeded_spectrum(n, seed & 0xFFFFFFFF, _fp(out))
    return out


class Channel:
    """DspChannel state (delay[1024] + previous window shape), dsp.rs:143-171."""

    def __init__(self):
        self._buf = np.zeros(1025, np.float32)  # 1024 floats + int32 shape
        lib().sko_channel_init(_vp(self._buf))

    @property
    def delay(self):
        return self._buf[:1024].copy()

    @property
    def prev_shape(self):
        return int(self._buf[1024:].view(np.int32)[0])

    def set_state(self, delay, prev_shape):
        self._buf[:1024] = delay
        self._buf[1024:].view(np.int32)[0] = prev_shape

    def synthesize(self, coeffs, seq, shape):
        coeffs = _f32(coeffs)
        assert coeffs.size == 1024
        out = np.empty(1024, np.float32)
        rc = lib().sko_synthesize_channel(_vp(self._buf), _fp(coeffs), int(seq), int(shape), _fp(out))
        if rc != 0:
            raise ValueError("invalid window sequence/shape")
        return out


def synthesize_stream(coeffs, seqs, shapes, channels=None):
    """coeffs [frames][ch][1024], seqs/shapes [frames][ch] -> pcm [frames][ch][1024] (+ final channels)."""
    coeffs = _f32(coeffs)
    frames, ch, n = coeffs.shape
    assert n == 1024
    chans = channels if channels is not None else [Channel() for _ in range(ch)]
    out = np.empty_like(coeffs)
    for f in range(frames):
        for c in range(ch):
            out[f, c] = chans[c].synthesize(coeffs[f, c], seqs[f][c], shapes[f][c])
    return out, chans


def pcm_stats(pcm):
    pcm = _f32(pcm).ravel()

    class S(C.Structure):
        _fields_ = [("n", C.c_uint64), ("rms", C.c_double), ("peak", C.c_double), ("checksum", C.c_uint64)]

    s = S()
    lib().sko_pcm_stats_from(_fp(pcm), pcm.size, C.byref(s))
    return {"sample_count": s.n, "rms": s.rms, "peak_abs": s.peak, "checksum": s.checksum}


_OUT_DTYPE = {2: np.int16, 4: None}


def pcm_convert(op, data, n=None):
    """Run elementwise op on raw input bytes/array; returns numpy array of the op's output type."""
    if isinstance(op, str):
        op = OP[op]
    raw = np.ascontiguousarray(data).view(np.uint8).ravel()
    ib, ob = lib().sko_op_in_bytes(op), lib().sko_op_out_bytes(op)
    if n is None:
        n = raw.size // ib
    name = OPS[op]
    if ob == 2:
        dt = np.int16
    elif name.endswith("_F32"):
        dt = np.float32
    else:
        dt = np.int32
    out = np.empty(n, dt)
    assert lib().sko_pcm_convert(op, _vp(raw), _vp(out), n) == 0
    return out


def float_sample_to_i16(x):
    return int(lib().sko_float_sample_to_i16(C.c_float(x)))


def interleave_i16(planar):
    planar = np.ascontiguousarray(planar, np.int16)
    ch, frames = planar.shape
    out = np.empty(ch * frames * 2, np.uint8)
    lib().sko_interleave_i16(_vp(planar), frames, ch, _vp(out))
    return out


def deinterleave(kind, data, ch):
    raw = np.ascontiguousarray(data).view(np.uint8).ravel()
    bps = {"i16": 2, "s24": 3, "f32": 4}[kind]
    frames = raw.size // (bps * ch)
    dt = {"i16": np.int16, "s24": np.int32, "f32": np.float32}[kind]
    out = np.empty((ch, frames), dt)
    getattr(lib(), "sko_deinterleave_" + kind)(_vp(raw), frames, ch, _vp(out))
    return out


def interleave_f32(planar):
    planar = _f32(planar)
    ch, frames = planar.shape
    out = np.empty(ch * frames * 4, np.uint8)
    lib().sko_interleave_f32(_vp(planar), frames, ch, _vp(out))
    return out


def fmt_bytes(fmt):
    return lib().sko_fmt_bytes(fmt)


def decoder_bytes_to_f32_planar(fmt, data, ch):
    raw = np.ascontiguousarray(data).view(np.uint8).ravel()
    frames = raw.size // (fmt_bytes(fmt) * ch)
    out = np.empty((ch, frames), np.float32)
    assert lib().sko_decoder_bytes_to_f32_planar(fmt, _vp(raw), frames, ch, _vp(out)) == 0
    return out


def core_bytes_to_f32_planar(fmt, data, ch):
    raw = np.ascontiguousarray(data).view(np.uint8).ravel()
    frames = raw.size // (fmt_bytes(fmt) * ch)
    out = np.empty((ch, frames), np.float32)
    assert lib().sko_core_bytes_to_f32_planar(fmt, _vp(raw), frames, ch, _vp(out)) == 0
    return out


def f32_planar_to_bytes(fmt, planar):
    planar = _f32(planar)
    ch, frames = planar.shape
    out = np.empty(ch * frames * fmt_bytes(fmt), np.uint8)
    assert lib().sko_f32_planar_to_bytes(fmt, _vp(planar), frames, ch, _vp(out)) == 0
    return out


def downmix_mono(planar):
    planar = _f32(planar)
    ch, frames = planar.shape
    out = np.empty(frames, np.float32)
    lib().sko_downmix_mono(_vp(planar), frames, ch, _vp(out))
    return out


def exact_signed_pcm_to_i16(fmt, data):
    raw = np.ascontiguousarray(data).view(np.uint8).ravel()
    n = raw.size // fmt_bytes(fmt)
    out = np.empty(n * 2, np.uint8)
    assert lib().sko_exact_signed_pcm_to_i16(fmt, _vp(raw), n, _vp(out)) == 0
    return out


def planar_f32_to_s16_interleaved(planar):
    planar = _f32(planar)
    ch, frames = planar.shape
    out = np.empty(frames * ch, np.int16)
    lib().sko_planar_f32_to_s16_interleaved(_vp(planar), frames, ch, _vp(out))
    return out


def resampler_taps(ratio):
    r = lib().sko_resampler_new(ratio, 1024, 1)
    taps = np.empty(256, np.float32)
    lib().sko_resampler_taps_phase0(r, _vp(taps))
    lib().sko_resampler_free(r)
    return taps


def downsample_planar(planar, in_hz, out_hz):
    """soundkit::downsample_audio on planar f32 [ch][frames] -> [ch][out_frames]."""
    planar = _f32(planar)
    ch, frames = planar.shape
    cap = lib().sko_downsample_out_max(frames, in_hz, out_hz)
    out = np.zeros((ch, cap), np.float32)
    n = lib().sko_downsample_planar(_vp(planar), frames, ch, in_hz, out_hz, _vp(out), cap)
    return out[:, :n].copy()


class StreamingResampler:
    """soundkit-decoder StreamingResampler (lib.rs:1917-2060) restated."""

    def __init__(self, in_hz, out_hz, channels):
        self._h = lib().sko_streaming_new(in_hz, out_hz, channels)
        self.channels = channels
        self.in_hz, self.out_hz = in_hz, out_hz

    def __del__(self):
        if getattr(self, "_h", None):
            lib().sko_streaming_free(self._h)
            self._h = None

    def process(self, planar):
        planar = _f32(planar)
        ch, n = planar.shape
        cap = int((n + 4096) * self.out_hz / self.in_hz * 2 + 64)
        out = np.zeros((ch, cap), np.float32)
        got = lib().sko_streaming_process(self._h, _vp(planar), n, n, _vp(out), cap, 0)
        return out[:, :got].copy()

    def flush(self):
        cap = int(4096 * self.out_hz / self.in_hz * 2 + 64)
        out = np.zeros((self.channels, cap), np.float32)
        got = lib().sko_streaming_flush(self._h, _vp(out), cap, 0)
        return out[:, :got].copy()
