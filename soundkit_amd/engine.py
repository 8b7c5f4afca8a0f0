"""Engine: thin object wrapper over the C ABI (include/soundkit_amd.h).

Host entry points take/return numpy arrays; the *_dev entry points take device addresses
(ints, or anything with .data_ptr() such as a torch tensor) and enqueue on the engine's
HIP stream.  All compute happens in libsoundkit_amd.so on the GPU.
"""
import ctypes as C
import os

import numpy as np

from . import _lib
from ._lib import FrameDesc, check, lib

ONLY_LONG, LONG_START, EIGHT_SHORT, LONG_STOP = 0, 1, 2, 3
SINE, KBD = 0, 1

FMT_S16LE, FMT_S16BE, FMT_S24LE, FMT_S24BE, FMT_S32LE, FMT_S32BE, FMT_F32LE, FMT_F32BE = range(8)

PCM_OPS = [
    "I16LE_TO_F32", "I16_TO_I16LE", "I16LE_TO_I16", "S24LE_TO_I32", "S24LE_TO_I16", "S24BE_TO_I16",
    "S32LE_TO_I32", "S32BE_TO_I32", "S32LE_TO_S24", "S32BE_TO_S24", "S32LE_TO_F32", "S32BE_TO_F32",
    "S32LE_TO_I16", "S32BE_TO_I16", "F32LE_TO_I16", "F32BE_TO_I16", "F32LE_TO_I32", "F32LE_TO_S24",
    "S16BE_TO_I16", "S16LE_TO_I16", "S16LE_TO_I32", "STEREO_TO_MONO_TAKE_LEFT", "STEREO_TO_MONO_AVG",
    "VEC_F32_TO_I16", "VEC_I16_TO_F32", "VEC_I32_TO_F32", "FLOAT_TO_I16_ROUND", "MP3_F32_TO_I16",
    "MP3_F32_TO_I32",
]
PCM_OP = {name: i for i, name in enumerate(PCM_OPS)}


def _ptr(x):
    """Device or host address of x (torch tensor, numpy array, int or None)."""
    if x is None:
        return None
    if isinstance(x, int):
        return C.c_void_p(x)
    if hasattr(x, "data_ptr"):
        return C.c_void_p(x.data_ptr())
    if isinstance(x, np.ndarray):
        return C.c_void_p(x.ctypes.data)
    raise TypeError("cannot take the address of %r" % type(x))


def make_descs(frames):
    """frames: iterable of (stream, channels, (seq...), (shape...)) -> ctypes array of sk_aac_frame_desc."""
    frames = list(frames)
    arr = (FrameDesc * max(len(frames), 1))()
    for i, (stream, ch, seqs, shapes) in enumerate(frames):
        d = arr[i]
        d.stream = stream
        d.channels = ch
        for c in range(min(ch, 2)):
            d.window_sequence[c] = seqs[c]
            d.window_shape[c] = shapes[c]
    return arr, len(frames)


def descs_from_arrays(streams, channels, seqs, shapes):
    """Vectorised builder: streams [n], channels scalar or [n], seqs/shapes [n][2] -> (ctypes array, n)."""
    streams = np.asarray(streams, np.uint32)
    n = streams.size
    raw = np.zeros(n, dtype=np.dtype([("stream", "<u4"), ("channels", "u1"), ("seq", "u1", (2,)),
                                      ("shape", "u1", (2,)), ("reserved", "u1", (3,))]))
    assert raw.dtype.itemsize == C.sizeof(FrameDesc)
    raw["stream"] = streams
    raw["channels"] = channels
    raw["seq"] = np.asarray(seqs, np.uint8).reshape(n, 2)
    raw["shape"] = np.asarray(shapes, np.uint8).reshape(n, 2)
    arr = (FrameDesc * max(n, 1)).from_buffer_copy(raw.tobytes() if n else bytes(C.sizeof(FrameDesc)))
    return arr, n


class Plan:
    """A validated, device-resident schedule of AAC frames (sk_aac_plan)."""

    def __init__(self, engine, descs, n):
        self.engine = engine
        self.n = n
        self.status = np.zeros(max(n, 1), np.int32)
        h = C.c_void_p()
        check(lib.sk_aac_plan_create(engine._h, descs, n, _ptr(self.status), C.byref(h)), "sk_aac_plan_create", engine._h)
        self._h = h
        self.status = self.status[:n]
        self.elements = int(lib.sk_aac_plan_elements(h))
        self.frames_ok = int(lib.sk_aac_plan_frames_ok(h))

    def run_f32(self, d_coeffs, d_pcm):
        check(lib.sk_aac_plan_run_f32_dev(self.engine._h, self._h, _ptr(d_coeffs), _ptr(d_pcm)),
              "sk_aac_plan_run_f32_dev", self.engine._h)

    def run_s16_planar(self, d_coeffs, d_pcm16):
        """planar s16 PCM (float_sample_to_i16 of every sample), same packing as run_f32's output"""
        check(lib.sk_aac_plan_run_s16_planar_dev(self.engine._h, self._h, _ptr(d_coeffs), _ptr(d_pcm16)),
              "sk_aac_plan_run_s16_planar_dev", self.engine._h)

    def run_tail_s16(self, d_coeffs, stream_stride, channels, frames_per_stream, d_out, out_stride):
        """sk_aac_plan_run_tail_s16_dev: run_s16_planar + the one-shot 48k->16k FIR to interleaved s16 as one launch.  WITHDRAWN:
        raises SoundkitError -6 (unsupported) unless SK_AAC_TAIL_ONE_LAUNCH=1 is set -- the kernel is computed wrongly by the platform
        once several workgroups share a CU (include/soundkit_amd.h); also -6 for plans the fused kernel does not cover"""
        got = C.c_uint32()
        check(lib.sk_aac_plan_run_tail_s16_dev(self.engine._h, self._h, _ptr(d_coeffs), stream_stride, channels, frames_per_stream,
                                               _ptr(d_out), out_stride, C.byref(got)), "sk_aac_plan_run_tail_s16_dev", self.engine._h)
        return got.value

    def run_s16(self, d_coeffs, d_pcm):
        check(lib.sk_aac_plan_run_s16_dev(self.engine._h, self._h, _ptr(d_coeffs), _ptr(d_pcm)),
              "sk_aac_plan_run_s16_dev", self.engine._h)

    def destroy(self):
        if self._h:
            lib.sk_aac_plan_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.destroy()
        except Exception:
            pass


class Engine:
    def __init__(self, device=0, max_streams=4096):
        h = C.c_void_p()
        check(lib.sk_engine_create(device, max_streams, C.byref(h)), "sk_engine_create")
        self._h = h
        self.device = device
        self.max_streams = max_streams
        self._channels = {}

    # ---- lifetime ---------------------------------------------------------------------
    def close(self):
        if getattr(self, "_h", None):
            lib.sk_engine_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    @property
    def hip_stream(self):
        return lib.sk_engine_hip_stream(self._h)

    def synchronize(self):
        check(lib.sk_engine_synchronize(self._h), "sk_engine_synchronize", self._h)

    # ---- streams ----------------------------------------------------------------------
    def open_stream(self, sample_rate=48000, channels=2):
        sid = C.c_uint32()
        check(lib.sk_stream_open(self._h, sample_rate, channels, C.byref(sid)), "sk_stream_open", self._h)
        self._channels[sid.value] = channels
        return sid.value

    def close_stream(self, sid):
        check(lib.sk_stream_close(self._h, sid), "sk_stream_close", self._h)

    def reset_stream(self, sid):
        check(lib.sk_stream_reset(self._h, sid), "sk_stream_reset", self._h)

    def get_state(self, sid, channels):
        delay = np.zeros((channels, 1024), np.float32)
        shape = np.zeros(channels, np.uint8)
        check(lib.sk_stream_get_state(self._h, sid, _ptr(delay), _ptr(shape)), "sk_stream_get_state", self._h)
        return delay, shape

    def set_state(self, sid, delay, shape):
        delay = np.ascontiguousarray(delay, np.float32)
        shape = np.ascontiguousarray(shape, np.uint8)
        check(lib.sk_stream_set_state(self._h, sid, _ptr(delay), _ptr(shape)), "sk_stream_set_state", self._h)

    # ---- AAC synthesis ----------------------------------------------------------------
    def plan(self, descs, n):
        return Plan(self, descs, n)

    def aac_synthesize(self, descs, n, coeffs, out="f32", pcm=None):
        """Host-buffer batch synthesis.  coeffs: packed f32 (sum(ch) * 1024).  Returns (pcm, status)."""
        coeffs = np.ascontiguousarray(coeffs, np.float32).ravel()
        status = np.zeros(max(n, 1), np.int32)
        if out == "f32":
            if pcm is None:
                pcm = np.zeros(coeffs.size, np.float32)
            rc = lib.sk_aac_synthesize_f32(self._h, descs, _ptr(coeffs), _ptr(pcm), n, _ptr(status))
        else:
            if pcm is None:
                pcm = np.zeros(coeffs.size, np.int16)
            rc = lib.sk_aac_synthesize_s16(self._h, descs, _ptr(coeffs), _ptr(pcm), n, _ptr(status))
        check(rc, "sk_aac_synthesize_" + out, self._h)
        return pcm, status[:n]

    def dequantize(self, quant, scalefactor):
        quant = np.ascontiguousarray(quant, np.int16).ravel()
        scalefactor = np.ascontiguousarray(scalefactor, np.int16).ravel()
        assert quant.size == scalefactor.size
        out = np.zeros(quant.size, np.float32)
        check(lib.sk_aac_dequantize(self._h, _ptr(quant), _ptr(scalefactor), _ptr(out), quant.size),
              "sk_aac_dequantize", self._h)
        return out

    # ---- PCM ---------------------------------------------------------------------------
    @staticmethod
    def _op_out_dtype(op):
        name = PCM_OPS[op]
        ob = lib.sk_pcm_op_out_bytes(op)
        if ob == 2:
            return np.int16
        return np.float32 if name.endswith("_F32") else np.int32

    def pcm_convert(self, op, data, n=None):
        if isinstance(op, str):
            op = PCM_OP[op]
        raw = np.ascontiguousarray(data).view(np.uint8).ravel()
        if n is None:
            n = raw.size // lib.sk_pcm_op_in_bytes(op)
        out = np.zeros(n, self._op_out_dtype(op))
        check(lib.sk_pcm_convert(self._h, op, _ptr(raw), _ptr(out), n), "sk_pcm_convert", self._h)
        return out

    def pcm_convert_dev(self, op, d_in, d_out, n):
        if isinstance(op, str):
            op = PCM_OP[op]
        check(lib.sk_pcm_convert_dev(self._h, op, _ptr(d_in), _ptr(d_out), n), "sk_pcm_convert_dev", self._h)

    def interleave_i16(self, planar):
        planar = np.ascontiguousarray(planar, np.int16)
        ch, frames = planar.shape
        out = np.zeros(ch * frames * 2, np.uint8)
        check(lib.sk_pcm_interleave_i16(self._h, _ptr(planar), frames, ch, _ptr(out)), "sk_pcm_interleave_i16", self._h)
        return out

    def deinterleave(self, kind, data, ch):
        raw = np.ascontiguousarray(data).view(np.uint8).ravel()
        bps = {"i16": 2, "s24": 3, "f32": 4}[kind]
        dt = {"i16": np.int16, "s24": np.int32, "f32": np.float32}[kind]
        frames = raw.size // (bps * ch)
        out = np.zeros((ch, frames), dt)
        fn = getattr(lib, "sk_pcm_deinterleave_" + kind)
        check(fn(self._h, _ptr(raw), frames, ch, _ptr(out)), "sk_pcm_deinterleave_" + kind, self._h)
        return out

    def interleave_f32(self, planar):
        planar = np.ascontiguousarray(planar, np.float32)
        ch, frames = planar.shape
        out = np.zeros(ch * frames * 4, np.uint8)
        check(lib.sk_pcm_interleave_f32(self._h, _ptr(planar), frames, ch, _ptr(out)), "sk_pcm_interleave_f32", self._h)
        return out

    def bytes_to_f32_planar(self, variant, fmt, data, ch):
        raw = np.ascontiguousarray(data).view(np.uint8).ravel()
        frames = raw.size // (lib.sk_pcm_fmt_bytes(fmt) * ch)
        out = np.zeros((ch, frames), np.float32)
        check(lib.sk_pcm_bytes_to_f32_planar(self._h, variant, fmt, _ptr(raw), frames, ch, _ptr(out)),
              "sk_pcm_bytes_to_f32_planar", self._h)
        return out

    def f32_planar_to_bytes(self, fmt, planar):
        planar = np.ascontiguousarray(planar, np.float32)
        ch, frames = planar.shape
        out = np.zeros(ch * frames * lib.sk_pcm_fmt_bytes(fmt), np.uint8)
        check(lib.sk_pcm_f32_planar_to_bytes(self._h, fmt, _ptr(planar), frames, ch, _ptr(out)),
              "sk_pcm_f32_planar_to_bytes", self._h)
        return out

    def downmix_mono(self, planar):
        planar = np.ascontiguousarray(planar, np.float32)
        ch, frames = planar.shape
        out = np.zeros(frames, np.float32)
        check(lib.sk_pcm_downmix_mono(self._h, _ptr(planar), frames, ch, _ptr(out)), "sk_pcm_downmix_mono", self._h)
        return out

    def exact_to_i16(self, fmt, data):
        raw = np.ascontiguousarray(data).view(np.uint8).ravel()
        n = raw.size // lib.sk_pcm_fmt_bytes(fmt)
        out = np.zeros(n * 2, np.uint8)
        check(lib.sk_pcm_exact_to_i16(self._h, fmt, _ptr(raw), n, _ptr(out)), "sk_pcm_exact_to_i16", self._h)
        return out

    # ---- resampling ---------------------------------------------------------------------
    @staticmethod
    def downsample_out_frames(frames, in_hz=48000, out_hz=16000):
        if (in_hz, out_hz) == (48000, 16000):
            return int(lib.sk_downsample_48k_16k_out_frames(frames))
        return int(lib.sk_downsample_out_frames(frames, in_hz, out_hz))

    def taps(self):
        t = np.zeros(256, np.float32)
        check(lib.sk_downsample_48k_16k_taps(self._h, _ptr(t)), "sk_downsample_48k_16k_taps", self._h)
        return t

    def downsample_48k_16k(self, rows):
        """rows: [n_rows][frames] f32 -> [n_rows][out_frames]"""
        rows = np.ascontiguousarray(rows, np.float32)
        n_rows, frames = rows.shape
        n_out = self.downsample_out_frames(frames)
        out = np.zeros((n_rows, n_out), np.float32)
        got = C.c_uint32()
        check(lib.sk_downsample_48k_16k_f32(self._h, _ptr(rows), n_rows, frames, _ptr(out), C.byref(got)),
              "sk_downsample_48k_16k_f32", self._h)
        assert got.value == n_out
        return out

    def downsample_48k_16k_dev(self, d_in, in_stride, rows, frames, d_out, out_stride):
        got = C.c_uint32()
        check(lib.sk_downsample_48k_16k_f32_dev(self._h, _ptr(d_in), in_stride, rows, frames, _ptr(d_out), out_stride,
                                                C.byref(got)), "sk_downsample_48k_16k_f32_dev", self._h)
        return got.value

    def downsample_dev(self, d_in, in_stride, rows, frames, in_hz, out_hz, d_out, out_stride):
        """sk_downsample_f32_dev: one-shot downsample_audio of device rows, any pair of the reference's common rates"""
        got = C.c_uint32()
        check(lib.sk_downsample_f32_dev(self._h, _ptr(d_in), in_stride, rows, frames, in_hz, out_hz, _ptr(d_out), out_stride, C.byref(got)),
              "sk_downsample_f32_dev", self._h)
        return got.value

    def downsample(self, rows, in_hz, out_hz):
        """rows: [n_rows][frames] f32 -> [n_rows][out_frames], any pair of the reference's common rates."""
        rows = np.ascontiguousarray(rows, np.float32)
        n_rows, frames = rows.shape
        n_out = int(lib.sk_downsample_out_frames(frames, in_hz, out_hz))
        out = np.zeros((n_rows, n_out), np.float32)
        got = C.c_uint32()
        check(lib.sk_downsample_f32(self._h, _ptr(rows), n_rows, frames, in_hz, out_hz, _ptr(out), n_out, C.byref(got)),
              "sk_downsample_f32", self._h)
        assert got.value == n_out
        return out

    def downsample_48k_16k_frames_s16_dev(self, d_pcm, stream_stride, frame_stride, channels, n_streams, frames_per_stream,
                                          d_out, out_stride):
        """FIR with the s16 output stage fused: d_out [n_streams][out_stride][channels] int16."""
        got = C.c_uint32()
        check(lib.sk_downsample_48k_16k_frames_s16_dev(self._h, _ptr(d_pcm), stream_stride, frame_stride, channels, n_streams,
                                                       frames_per_stream, _ptr(d_out), out_stride, C.byref(got)),
              "sk_downsample_48k_16k_frames_s16_dev", self._h)
        return got.value

    def downsample_48k_16k_frames_s16_to_s16_dev(self, d_pcm16, stream_stride, frame_stride, channels, n_streams, frames_per_stream,
                                                 d_out, out_stride):
        """the worker's resample step on planar s16 PCM (Plan.run_s16_planar): d_out [n_streams][out_stride][channels] int16"""
        got = C.c_uint32()
        check(lib.sk_downsample_48k_16k_frames_s16_to_s16_dev(self._h, _ptr(d_pcm16), stream_stride, frame_stride, channels,
                                                              n_streams, frames_per_stream, _ptr(d_out), out_stride, C.byref(got)),
              "sk_downsample_48k_16k_frames_s16_to_s16_dev", self._h)
        return got.value

    def downsample_48k_16k_frames_s16_to_f32_dev(self, d_pcm16, stream_stride, frame_stride, channels, n_streams, frames_per_stream,
                                                 d_out, out_stride):
        got = C.c_uint32()
        check(lib.sk_downsample_48k_16k_frames_s16_to_f32_dev(self._h, _ptr(d_pcm16), stream_stride, frame_stride, channels,
                                                              n_streams, frames_per_stream, _ptr(d_out), out_stride, C.byref(got)),
              "sk_downsample_48k_16k_frames_s16_to_f32_dev", self._h)
        return got.value

    def downsample_48k_16k_frames_dev(self, d_pcm, stream_stride, frame_stride, channels, n_streams, frames_per_stream,
                                      d_out, out_stride):
        got = C.c_uint32()
        check(lib.sk_downsample_48k_16k_frames_dev(self._h, _ptr(d_pcm), stream_stride, frame_stride, channels, n_streams,
                                                   frames_per_stream, _ptr(d_out), out_stride, C.byref(got)),
              "sk_downsample_48k_16k_frames_dev", self._h)
        return got.value

    def f32_planar_to_bytes_batch_dev(self, fmt, d_planar, batch, plane_stride, frames, ch, d_out):
        check(lib.sk_pcm_f32_planar_to_bytes_batch_dev(self._h, fmt, _ptr(d_planar), batch, plane_stride, frames, ch,
                                                       _ptr(d_out)), "sk_pcm_f32_planar_to_bytes_batch_dev", self._h)

    _rs_ratio_max = 96000 / 8000  # largest ratio among the common rates: sizes host output buffers

    def resampler_open(self, sid, in_hz=48000, out_hz=16000):
        check(lib.sk_resampler_open(self._h, sid, in_hz, out_hz), "sk_resampler_open", self._h)

    def resampler_close(self, sid):
        check(lib.sk_resampler_close(self._h, sid), "sk_resampler_close", self._h)

    def resampler_process(self, sids, data, channels):
        """data: [n_streams][channels][frames] f32 -> list of [channels][out_frames] arrays."""
        sids = np.ascontiguousarray(sids, np.uint32)
        data = np.ascontiguousarray(data, np.float32)
        n, ch, frames = data.shape
        assert ch == channels
        cap = int((frames + 4096) * self._rs_ratio_max + 64)
        out = np.zeros((n, ch, cap), np.float32)
        got = np.zeros(n, np.uint32)
        check(lib.sk_resampler_process_f32(self._h, _ptr(sids), n, _ptr(data), frames, _ptr(out), cap, _ptr(got)),
              "sk_resampler_process_f32", self._h)
        return [out[i, :, :got[i]].copy() for i in range(n)]

    def resampler_flush(self, sids, channels):
        sids = np.ascontiguousarray(sids, np.uint32)
        n = sids.size
        cap = int(4096 * self._rs_ratio_max + 64)
        out = np.zeros((n, channels, cap), np.float32)
        got = np.zeros(n, np.uint32)
        check(lib.sk_resampler_flush_f32(self._h, _ptr(sids), n, _ptr(out), cap, _ptr(got)),
              "sk_resampler_flush_f32", self._h)
        return [out[i, :, :got[i]].copy() for i in range(n)]


    def tick_run(self, streams, descs, n_frames, coeffs):
        """One scheduler tick (sk_tick_run).  streams: list of dicts {stream, n_frames, out_bits, out_channels,
        resample, flush}; descs / coeffs as aac_synthesize.  -> list of (stream_index, status, frames, channels,
        bits, bytes) in the order the reference's worker would have sent them."""
        import ctypes as C
        from ._lib import TickStream, TickOutput
        ts = (TickStream * max(len(streams), 1))()
        for i, s in enumerate(streams):
            ts[i].stream, ts[i].n_frames = int(s["stream"]), int(s["n_frames"])
            ts[i].out_bits, ts[i].out_channels = int(s.get("out_bits", 16)), int(s["out_channels"])
            ts[i].resample, ts[i].flush = int(bool(s.get("resample", 0))), int(bool(s.get("flush", 0)))
        max_out = C.c_uint32()
        cap = lib.sk_tick_out_bound_on(self._h, ts, len(streams), C.byref(max_out))
        out = np.zeros(max(cap, 16), np.uint8)
        recs = (TickOutput * max(max_out.value, 1))()
        n_out, used = C.c_uint32(), C.c_size_t()
        coeffs = np.ascontiguousarray(coeffs, np.float32)
        check(lib.sk_tick_run(self._h, ts, len(streams), descs, _ptr(coeffs) if n_frames else None, n_frames, _ptr(out),
                              out.size, recs, max_out.value, C.byref(n_out), C.byref(used)), "sk_tick_run", self._h)
        res = []
        for r in recs[:n_out.value]:
            res.append((r.stream_index, r.status, r.frames, r.channels, r.bits,
                        out[r.byte_offset:r.byte_offset + r.bytes].tobytes()))
        return res


    def tick_run_q(self, streams, descs, n_units, sides, quant):
        """sk_tick_run_q: as tick_run, fed by AacLcFrontEnd.parse_q -- sides [n_units][SK_AAC_UNIT_SIDE_BYTES] u8, quant the
        units' i16 values packed like tick_run's coeffs"""
        import ctypes as C
        from ._lib import TickStream, TickOutput
        ts = (TickStream * max(len(streams), 1))()
        for i, s in enumerate(streams):
            ts[i].stream, ts[i].n_frames = int(s["stream"]), int(s["n_frames"])
            ts[i].out_bits, ts[i].out_channels = int(s.get("out_bits", 16)), int(s["out_channels"])
            ts[i].resample, ts[i].flush = int(bool(s.get("resample", 0))), int(bool(s.get("flush", 0)))
        max_out = C.c_uint32()
        cap = lib.sk_tick_out_bound_on(self._h, ts, len(streams), C.byref(max_out))
        out = np.zeros(max(cap, 16), np.uint8)
        recs = (TickOutput * max(max_out.value, 1))()
        n_out, used = C.c_uint32(), C.c_size_t()
        sides = np.ascontiguousarray(sides, np.uint8)
        quant = np.ascontiguousarray(quant, np.int16)
        check(lib.sk_tick_run_q(self._h, ts, len(streams), descs, _ptr(sides) if n_units else None, _ptr(quant) if n_units else None,
                                n_units, _ptr(out), out.size, recs, max_out.value, C.byref(n_out), C.byref(used)), "sk_tick_run_q", self._h)
        return [(r.stream_index, r.status, r.frames, r.channels, r.bits, out[r.byte_offset:r.byte_offset + r.bytes].tobytes())
                for r in recs[:n_out.value]]

    def tick_run_au(self, streams, access_units):
        """sk_tick_run_au: as tick_run, but the entropy front-end runs on the GPU.  access_units: the raw access
        units (bytes) of all streams, stream by stream in the order of `streams`."""
        import ctypes as C
        from ._lib import TickStream, TickOutput
        ts = (TickStream * max(len(streams), 1))()
        for i, s in enumerate(streams):
            ts[i].stream, ts[i].n_frames = int(s["stream"]), int(s["n_frames"])
            ts[i].out_bits, ts[i].out_channels = int(s.get("out_bits", 16)), int(s["out_channels"])
            ts[i].resample, ts[i].flush = int(bool(s.get("resample", 0))), int(bool(s.get("flush", 0)))
        n = len(access_units)
        items = np.zeros((max(n, 1), 2), np.uint32)
        blob = bytearray()
        for k, au in enumerate(access_units):
            items[k] = (len(blob), len(au))
            blob += bytes(au) + b"\0" * (8 + (-len(au)) % 4)   # >= 8 zero bytes, next unit 4-byte aligned
        blob = np.frombuffer(bytes(blob) + b"\0" * 8, np.uint8)
        max_out = C.c_uint32()
        cap = lib.sk_tick_out_bound_on(self._h, ts, len(streams), C.byref(max_out))
        out = np.zeros(max(cap, 16), np.uint8)
        recs = (TickOutput * max(max_out.value, 1))()
        n_out, used = C.c_uint32(), C.c_size_t()
        check(lib.sk_tick_run_au(self._h, ts, len(streams), _ptr(items), n, _ptr(blob), blob.size, _ptr(out), out.size, recs,
                                 max_out.value, C.byref(n_out), C.byref(used)), "sk_tick_run_au", self._h)
        return [(r.stream_index, r.status, r.frames, r.channels, r.bits, out[r.byte_offset:r.byte_offset + r.bytes].tobytes())
                for r in recs[:n_out.value]]


    def entropy_decode(self, streams, access_units):
        """sk_aac_entropy_decode: the AAC-LC front-end alone, on the GPU.  streams: [(stream id, unit count)];
        access_units: the raw units of all streams, stream by stream.  -> [(status, coeffs [ch][1024] f32,
        window_sequence[ch], window_shape[ch])] per unit."""
        import ctypes as C
        n = len(access_units)
        ids = np.array([s for s, _ in streams], np.uint32)
        counts = np.array([c for _, c in streams], np.uint32)
        assert int(counts.sum()) == n
        items = np.zeros((max(n, 1), 2), np.uint32)
        blob = bytearray()
        for k, au in enumerate(access_units):
            items[k] = (len(blob), len(au))
            blob += bytes(au) + b"\0" * (8 + (-len(au)) % 4)
        blob = np.frombuffer(bytes(blob) + b"\0" * 8, np.uint8)
        ch = np.repeat(np.array([self._channels[int(s)] for s in ids], np.uint32), counts) if n else np.zeros(0, np.uint32)
        coeffs = np.zeros(max(int(ch.sum()) * 1024, 1), np.float32)
        descs = (FrameDesc * max(n, 1))()
        status = np.zeros(max(n, 1), np.int32)
        check(lib.sk_aac_entropy_decode(self._h, _ptr(ids), _ptr(counts), len(streams), _ptr(items), n, _ptr(blob), blob.size,
                                        _ptr(coeffs), descs, _ptr(status)), "sk_aac_entropy_decode", self._h)
        out, off = [], 0
        for k in range(n):
            c = int(ch[k])
            out.append((int(status[k]), coeffs[off:off + c * 1024].reshape(c, 1024).copy(),
                        [int(descs[k].window_sequence[i]) for i in range(c)], [int(descs[k].window_shape[i]) for i in range(c)]))
            off += c * 1024
        return out


    def expand_q_decode(self, streams, parsed):
        """sk_aac_expand_q_decode: the device half of the quantised hand-over alone.  streams: [(stream id, unit count)];
        parsed: AacLcFrontEnd.parse_q's results (quant, side, sequences, shapes) of all units, stream by stream.
        -> [(status, coeffs [ch][1024] f32, window_sequence[ch], window_shape[ch])] per unit."""
        from ._lib import AAC_UNIT_SIDE_BYTES
        n = len(parsed)
        ids = np.array([s for s, _ in streams], np.uint32)
        counts = np.array([c for _, c in streams], np.uint32)
        assert int(counts.sum()) == n
        ch = np.repeat(np.array([self._channels[int(s)] for s in ids], np.uint32), counts) if n else np.zeros(0, np.uint32)
        owner = np.repeat(ids, counts) if n else np.zeros(0, np.uint32)
        descs_in = (FrameDesc * max(n, 1))()
        for k, (q, side, seqs, shapes) in enumerate(parsed):
            descs_in[k].stream, descs_in[k].channels = int(owner[k]), int(ch[k])
            for c in range(int(ch[k])):
                descs_in[k].window_sequence[c], descs_in[k].window_shape[c] = int(seqs[c]), int(shapes[c])
        sides = np.ascontiguousarray(np.stack([p[1] for p in parsed]) if n else np.zeros((1, AAC_UNIT_SIDE_BYTES)), np.uint8)
        quant = np.ascontiguousarray(np.concatenate([p[0].ravel() for p in parsed]) if n else np.zeros(1), np.int16)
        coeffs = np.zeros(max(int(ch.sum()) * 1024, 1), np.float32)
        descs = (FrameDesc * max(n, 1))()
        status = np.zeros(max(n, 1), np.int32)
        check(lib.sk_aac_expand_q_decode(self._h, _ptr(ids), _ptr(counts), len(streams), descs_in, _ptr(sides), _ptr(quant), n,
                                         _ptr(coeffs), descs, _ptr(status)), "sk_aac_expand_q_decode", self._h)
        out, off = [], 0
        for k in range(n):
            c = int(ch[k])
            out.append((int(status[k]), coeffs[off:off + c * 1024].reshape(c, 1024).copy(),
                        [int(descs[k].window_sequence[i]) for i in range(c)], [int(descs[k].window_shape[i]) for i in range(c)]))
            off += c * 1024
        return out

    def where(self):
        """sk_engine_where: the stage of the engine's current tick ("idle" outside one)"""
        return lib.sk_engine_where(self._h).decode()

    def set_resampler_exact(self, exact=True):
        """generic-ratio resampling in rubato's own order of operations (bit-identical to the restated reference) instead of
        the matrix-core form"""
        check(lib.sk_engine_set_resampler_exact(self._h, 1 if exact else 0), "sk_engine_set_resampler_exact", self._h)

    def set_wait_bound(self, seconds):
        check(lib.sk_engine_set_wait_bound(self._h, float(seconds)), "sk_engine_set_wait_bound")

    def debug_fail_after(self, n_hip_calls):
        """error-path tests: the n-th HIP call from now fails as a launch failure"""
        check(lib.sk_engine_debug_fail_after(self._h, int(n_hip_calls)), "sk_engine_debug_fail_after")


_default = None


def default_engine():
    """Process-wide engine on HIP device 0 (LOCAL_RANK if set), created on first use."""
    global _default
    if _default is None:
        import os
        _default = Engine(int(os.environ.get("LOCAL_RANK", "0")), 8192)
    return _default
