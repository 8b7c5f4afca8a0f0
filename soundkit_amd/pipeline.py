"""Host-side mirror of soundkit-decoder's streaming pipeline handle (soundkit-decoder/src/lib.rs:2590-2889) on
top of the batch scheduler (csrc/pipeline.cpp): `DecodePipeline.spawn_with_options(..)` returns a handle with
`send / finish / try_recv / recv / cancel / queued_input_bytes`, but instead of one worker thread per stream every
handle feeds one shared scheduler per GPU (entropy decode on host threads, everything after it batched on the GPU).

Input: ADTS AAC-LC.  The reference's other formats stay with their CPU decoders (out of scope)."""
import ctypes as C
from dataclasses import dataclass
from typing import Optional

import numpy as np

from ._lib import AudioInfo, DecodeOptionsC, PipelineConfig, PipelineStats, SoundkitError, check, lib
from .audio_types import AudioData, EncodingFlag, Endianness
from .engine import default_engine

SK_PIPE_INPUT_FULL, SK_PIPE_CLOSED, SK_PIPE_CHUNK_TOO_LARGE, SK_ERR_CAPACITY = -201, -202, -203, -7


@dataclass
class DecodeOptions:  # lib.rs:147-151
    output_bits_per_sample: Optional[int] = None
    output_sample_rate: Optional[int] = None
    output_channels: Optional[int] = None


class DecodeError(Exception):  # lib.rs:108-141
    def __init__(self, kind, message="", status=0):
        self.kind, self.status = kind, status
        super().__init__(message or {"InputBufferFull": "Input buffer full", "PipelineClosed": "Decode pipeline is closed"}.get(kind, kind))


class BatchScheduler:
    """One per GPU.  Keyword arguments are the fields of sk_pipeline_config (0 / missing = default)."""

    def __init__(self, engine=None, **config):
        self.engine = engine or default_engine()
        cfg = PipelineConfig()
        for k, v in config.items():
            if not hasattr(cfg, k):
                raise TypeError("unknown scheduler option %r" % k)
            setattr(cfg, k, int(v))
        h = C.c_void_p()
        check(lib.sk_pipeline_create(self.engine._h, C.byref(cfg), C.byref(h)), "sk_pipeline_create", self.engine._h)
        self._h = h

    def close(self):
        if getattr(self, "_h", None):
            lib.sk_pipeline_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def spawn(self, options=None):
        options = options or DecodeOptions()
        o = DecodeOptionsC(options.output_sample_rate or 0, options.output_bits_per_sample or 0, options.output_channels or 0, 0)
        handle = C.c_uint32()
        rc = lib.sk_pipeline_spawn(self._h, C.byref(o), C.byref(handle))
        if rc != 0:
            raise SoundkitError(rc, "sk_pipeline_spawn")
        return DecodePipelineHandle(self, handle.value)

    def wait_outputs(self, timeout_ms=100, cap=256):
        """Handles (as spawned: `handle.id`) that have outputs or have ended; blocks up to timeout_ms.  For callers that
        serve many streams from one thread."""
        arr = (C.c_uint32 * cap)()
        n = lib.sk_pipeline_wait_outputs(self._h, arr, cap, timeout_ms)
        if n < 0:
            raise SoundkitError(n, "sk_pipeline_wait_outputs")
        return list(arr[:n])

    def stats(self):
        st = PipelineStats()
        check(lib.sk_pipeline_get_stats(self._h, C.byref(st)), "sk_pipeline_get_stats")
        return {name: getattr(st, name) for name, _ in st._fields_ if name != "reserved"}


class DecodePipelineHandle:
    """lib.rs:2788-2889"""

    def __init__(self, scheduler, handle):
        self._s, self._id = scheduler, handle
        self.id = handle
        self._buf = np.empty(1 << 16, np.uint8)
        self._info = AudioInfo()

    def send(self, data):
        if self._id is None:
            raise DecodeError("PipelineClosed")
        data = bytes(data)
        rc = lib.sk_pipeline_send(self._s._h, self._id, data if data else None, len(data))
        if rc == SK_PIPE_INPUT_FULL:
            raise DecodeError("InputBufferFull")
        if rc == SK_PIPE_CLOSED:
            raise DecodeError("PipelineClosed")
        if rc == SK_PIPE_CHUNK_TOO_LARGE:
            raise DecodeError("InputChunkTooLarge", "Input chunk of %d bytes exceeds the 4 MiB limit" % len(data))
        if rc != 0:
            raise SoundkitError(rc, "sk_pipeline_send")

    def finish(self):
        self.send(b"")

    def _take(self, call, on_empty=None):
        if self._id is None:
            raise DecodeError("PipelineClosed")
        for _ in range(2):
            rc = call()
            if rc == SK_ERR_CAPACITY:
                self._buf = np.empty(int(self._info.bytes) * 2, np.uint8)
                continue
            break
        if rc == SK_PIPE_CLOSED:
            return None
        if rc == 0:
            if on_empty is not None:
                raise on_empty
            return None
        if rc != 1:
            raise SoundkitError(rc, "sk_pipeline_recv")
        i = self._info
        payload = self._buf[:i.bytes].tobytes()
        if i.is_error:
            return DecodeError("DecodingFailed", payload.decode("utf-8", "replace"), i.status)
        return AudioData(i.bits_per_sample, i.channel_count, i.sampling_rate, payload, EncodingFlag.PCMSigned, Endianness.LittleEndian)

    def try_recv(self):
        """None when nothing is ready (or the stream has ended and is drained); else AudioData or a DecodeError value."""
        return self._take(lambda: lib.sk_pipeline_try_recv(self._s._h, self._id, self._buf.ctypes.data, self._buf.size,
                                                           C.byref(self._info)))

    def recv(self, timeout_ms=10000):
        """Blocks for the next output: AudioData or a DecodeError value; None only when the stream has ended and is
        drained.  A wait that runs out raises TimeoutError -- it never looks like the end of the stream."""
        return self._take(lambda: lib.sk_pipeline_recv(self._s._h, self._id, self._buf.ctypes.data, self._buf.size,
                                                       C.byref(self._info), timeout_ms),
                          TimeoutError("no output within %d ms" % timeout_ms))

    def ended(self):
        """True once the worker side has ended and every output has been taken."""
        if self._id is None:
            return True
        rc = lib.sk_pipeline_try_recv(self._s._h, self._id, self._buf.ctypes.data, 0, C.byref(self._info))
        return rc == SK_PIPE_CLOSED

    def cancel(self):
        if self._id is not None and self._s._h:
            lib.sk_pipeline_cancel(self._s._h, self._id)
        self._id = None

    def queued_input_bytes(self):
        if self._id is None:
            raise DecodeError("PipelineClosed")
        return lib.sk_pipeline_queued_input_bytes(self._s._h, self._id)

    def __del__(self):  # Drop, lib.rs:2884-2888
        try:
            self.cancel()
        except Exception:
            pass


_scheduler = None


def default_scheduler():
    global _scheduler
    if _scheduler is None:
        _scheduler = BatchScheduler()
    return _scheduler


class DecodePipeline:
    """lib.rs:2590-2786, for ADTS AAC-LC"""

    @staticmethod
    def spawn():
        return default_scheduler().spawn(DecodeOptions())

    @staticmethod
    def spawn_with_options(options):
        return default_scheduler().spawn(options)
