"""Host-side mirror of soundkit-mp3's decoder (soundkit-mp3/src/lib.rs:147-374): frame sync, header, side information,
the bit reservoir, scale factors and the Huffman stage on the host (csrc/mp3_bitstream.cpp, csrc/mp3_decoder.cpp);
requantisation, joint stereo and the short-block reorder (csrc/mp3_requant.hip) and the hybrid synthesis filterbank
(csrc/mp3_hybrid.hip) on the GPU; the reference's `f32_to_i16` tail.  The standard's data tables (ISO/IEC 11172-3 B.3 /
B.6 / B.7 / B.8, 13818-3 2.4.3.2) are csrc/mp3_iso_tables.h; `Mp3Decoder()` uses them unless given another Codebook."""
import ctypes as C

import numpy as np

from ._lib import (Mp3FrameInfo, Mp3GranuleData, Mp3GranuleDesc, Mp3RequantGranule, Mp3SideInfo, SoundkitError, check, lib)
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


# ---- the fixed-syntax front (host) ----------------------------------------------------------------------------------------

def _as_dict(struct):
    out = {}
    for name, _ in struct._fields_:
        v = getattr(struct, name)
        out[name] = v if isinstance(v, int) else np.ctypeslib.as_array(v).tolist()
    return out


def _bytes(data):
    return np.frombuffer(bytes(data), np.uint8) if len(data) else np.zeros(1, np.uint8)


def parse_header(data):
    """4 bytes -> (status, Mp3FrameInfo)"""
    info = Mp3FrameInfo()
    b = _bytes(data)
    return lib.sk_mp3_parse_header(_ptr(b), len(data), C.byref(info)), info


def parse_side_info(frame, info):
    side = Mp3SideInfo()
    b = _bytes(frame)
    return lib.sk_mp3_parse_side_info(_ptr(b), len(frame), C.byref(info), C.byref(side)), side


def scan(data, cap=1 << 16):
    """every frame of a byte string -> ([Mp3FrameInfo], bytes consumed)"""
    frames = (Mp3FrameInfo * cap)()
    n, used = C.c_uint32(0), C.c_size_t(0)
    b = _bytes(data)
    check(lib.sk_mp3_scan(_ptr(b), len(data), frames, cap, C.byref(n), C.byref(used)), "sk_mp3_scan")
    return [frames[i] for i in range(min(n.value, cap))], used.value


def scan_free(data, free_format_bytes=0, cap=1 << 16):
    """sk_mp3_scan_free -> ([Mp3FrameInfo], bytes consumed, free_format_bytes to hand to the next call)"""
    frames = (Mp3FrameInfo * cap)()
    n, used, ffb = C.c_uint32(0), C.c_size_t(0), C.c_uint32(free_format_bytes)
    b = _bytes(data)
    check(lib.sk_mp3_scan_free(_ptr(b), len(data), frames, cap, C.byref(n), C.byref(used), C.byref(ffb)), "sk_mp3_scan_free")
    return [frames[i] for i in range(min(n.value, cap))], used.value, ffb.value


def main_data(frame, info, side, reservoir):
    """(status, the bytes parts 2 + 3 of this frame are read from)"""
    out = np.zeros(len(reservoir) + len(frame) + 16, np.uint8)
    n = C.c_size_t(0)
    f, r = _bytes(frame), _bytes(reservoir)
    rc = lib.sk_mp3_main_data(_ptr(f), len(frame), C.byref(info), C.byref(side), _ptr(r), len(reservoir), _ptr(out), out.size, C.byref(n))
    return rc, out[:n.value].tobytes()


# ---- requantisation / joint stereo / reorder (GPU) -------------------------------------------------------------------------

def set_band_tables(sample_rate, long_offsets, short_offsets, pretab, engine=None):
    """ISO/IEC 11172-3 Table B.8 (23 long, 14 short offsets) and the pre-emphasis table (22), once per engine and rate"""
    engine = engine or default_engine()
    lo, so, pt = (np.ascontiguousarray(long_offsets, np.uint16), np.ascontiguousarray(short_offsets, np.uint16),
                  np.ascontiguousarray(pretab, np.uint8))
    assert lo.shape == (23,) and so.shape == (14,) and pt.shape == (22,)
    return lib.sk_mp3_set_band_tables(engine._h, sample_rate, _ptr(lo), _ptr(so), _ptr(pt))


def make_requant_granules(granules):
    """granules: dicts as oracle/mp3_bitstream.py requantize_granule takes them, plus sample_rate and lsf"""
    arr = (Mp3RequantGranule * max(len(granules), 1))()
    for i, g in enumerate(granules):
        a = arr[i]
        a.sample_rate, a.channels = g["sample_rate"], g["channels"]
        a.ms_stereo, a.intensity_stereo, a.lsf = g.get("ms_stereo", 0), g.get("intensity_stereo", 0), g.get("lsf", 0)
        for c in range(min(g["channels"], 2)):
            src, dst = g["ch"][c], a.ch[c]
            for name in ("global_gain", "scalefac_scale", "preflag", "block_type", "mixed_block_flag"):
                setattr(dst, name, src[name])
            for w in range(3):
                dst.subblock_gain[w] = src["subblock_gain"][w]
            for b in range(22):
                dst.scalefac_l[b] = src["scalefac_l"][b]
            for b in range(13):
                for w in range(3):
                    dst.scalefac_s[b][w] = src["scalefac_s"][b][w]
    return arr


def requantize(granules, quant, engine=None):
    """quant: int16, the granules' channels one after another, 576 lines each, bitstream order -> (xr f32 same shape, status)"""
    engine = engine or default_engine()
    arr = make_requant_granules(granules)
    n = len(granules)
    q = np.ascontiguousarray(quant, np.int16)
    xr = np.zeros(q.shape, np.float32)
    status = np.zeros(max(n, 1), np.int32)
    check(lib.sk_mp3_requantize(engine._h, arr, _ptr(q), _ptr(xr), n, _ptr(status)), "sk_mp3_requantize", engine._h)
    return xr, status[:n]


# ---- parts 2 + 3 over caller-supplied tables, and the decoder handle -----------------------------------------------------------

def iso_tables():
    """the standard's tables as the library holds them (sk_mp3_iso_tables) -> Mp3Tables"""
    from ._lib import Mp3Tables
    t = Mp3Tables()
    check(lib.sk_mp3_iso_tables(C.byref(t)), "sk_mp3_iso_tables")
    return t


class Codebook:
    """sk_mp3_codebook: host-side decoding structures built from an Mp3Tables; tables=None: the standard's (Table B.7 and
    friends, csrc/mp3_iso_tables.h)"""

    def __init__(self, tables=None):
        self._h = C.c_void_p()
        if tables is None:
            check(lib.sk_mp3_codebook_create_iso(C.byref(self._h)), "sk_mp3_codebook_create_iso")
        else:
            check(lib.sk_mp3_codebook_create(C.byref(tables), C.byref(self._h)), "sk_mp3_codebook_create")

    def close(self):
        if self._h:
            lib.sk_mp3_codebook_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        self.close()


def decode_main_data(codebook, info, side, main):
    """(status, [granule][channel] Mp3GranuleData) -- scale factors and the 576 integers of every granule of one frame"""
    out = ((Mp3GranuleData * 2) * 2)()
    b = _bytes(main)
    rc = lib.sk_mp3_decode_main_data(codebook._h, C.byref(info), C.byref(side), _ptr(b), len(main), out)
    return rc, out


MAX_SAMPLES_PER_FRAME = 2304


class Mp3Decoder:
    """soundkit-mp3's Mp3Decoder (soundkit-mp3/src/lib.rs:147-374): new / sample_rate / channels / buffer_len / reset /
    decode_i16 / decode_i32 / decode_f32, bytes in at any chunking, interleaved samples out; errors raise SoundkitError"""

    def __init__(self, codebook=None, engine=None):
        self._engine = engine or default_engine()
        self._codebook = codebook  # None: Mp3Decoder::new() -- the standard's tables
        self._h = C.c_void_p()
        check(lib.sk_mp3_decoder_create(self._engine._h, codebook._h if codebook is not None else None, C.byref(self._h)),
              "sk_mp3_decoder_create", self._engine._h)

    def close(self):
        if self._h:
            lib.sk_mp3_decoder_destroy(self._h)
            self._h = C.c_void_p()

    def _info(self):
        rate, ch, n, frames = C.c_uint32(0), C.c_uint8(0), C.c_size_t(0), C.c_uint64(0)
        check(lib.sk_mp3_decoder_info(self._h, C.byref(rate), C.byref(ch), C.byref(n), C.byref(frames)), "sk_mp3_decoder_info")
        return rate.value, ch.value, n.value, frames.value

    def sample_rate(self):
        return self._info()[0] or None

    def channels(self):
        return self._info()[1] or None

    def buffer_len(self):
        return self._info()[2]

    def frames_decoded(self):
        return self._info()[3]

    def reset(self):
        check(lib.sk_mp3_decoder_reset(self._h), "sk_mp3_decoder_reset")

    def _decode(self, fn, data, out):
        b = _bytes(data)
        n = C.c_size_t(0)
        rc = fn(self._h, _ptr(b), len(data), _ptr(out), out.size, C.byref(n))
        if rc != 0:
            raise SoundkitError(rc, "sk_mp3_decoder_decode", lib.sk_engine_last_hip_error(self._engine._h).decode() if rc in (-3, -4, -8) else "")
        return n.value

    def decode_i16(self, data, out, fec=False):
        assert out.dtype == np.int16
        return self._decode(lib.sk_mp3_decoder_decode_i16, data, out)

    def decode_i32(self, data, out, fec=False):
        assert out.dtype == np.int32
        return self._decode(lib.sk_mp3_decoder_decode_i32, data, out)

    def decode_f32(self, data, out, fec=False):
        assert out.dtype == np.float32
        return self._decode(lib.sk_mp3_decoder_decode_f32, data, out)
