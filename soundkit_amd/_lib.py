"""Loader for libsoundkit_amd.so (the C ABI of include/soundkit_amd.h).

The HIP library is the product; there is no Python or CPU compute fallback.  If the
shared object is missing or cannot be loaded this module raises ImportError loudly.
"""
import ctypes as C
import os
import re

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("SOUNDKIT_AMD_LIB") or os.path.join(_HERE, "libsoundkit_amd.so")  # override: A/B builds
HEADER_PATH = os.path.join(os.path.dirname(_HERE), "include", "soundkit_amd.h")

SK_OK = 0
AAC_UNIT_SIDE_BYTES = 1600  # SK_AAC_UNIT_SIDE_BYTES
ERR_NAMES = {0: "SK_OK", -1: "SK_ERR_INVALID_ARG", -2: "SK_ERR_NO_DEVICE", -3: "SK_ERR_HIP", -4: "SK_ERR_OOM",
             -5: "SK_ERR_BAD_STREAM", -6: "SK_ERR_UNSUPPORTED", -7: "SK_ERR_CAPACITY", -8: "SK_ERR_TIMEOUT", -9: "SK_ERR_INTERNAL",
             -101: "UnexpectedEof", -102: "InvalidAudioObjectType", -103: "UnsupportedAudioObjectType",
             -104: "UnsupportedSamplingFrequencyIndex", -105: "UnsupportedChannelConfig", -106: "UnsupportedFeature",
             -107: "InvalidConfig", -108: "InvalidBitstream",
             -301: "Mp3NeedMore", -302: "Mp3NoSync", -303: "Mp3Unsupported", -304: "Mp3Invalid",
             -201: "InputBufferFull", -202: "PipelineClosed", -203: "InputChunkTooLarge"}


class FrameDesc(C.Structure):
    """sk_aac_frame_desc"""
    _fields_ = [("stream", C.c_uint32), ("channels", C.c_uint8), ("window_sequence", C.c_uint8 * 2),
                ("window_shape", C.c_uint8 * 2), ("reserved", C.c_uint8 * 3)]


class Mp3GranuleDesc(C.Structure):
    """sk_mp3_granule_desc"""
    _fields_ = [("stream", C.c_uint32), ("channels", C.c_uint8), ("block_type", C.c_uint8 * 2), ("mixed_block_flag", C.c_uint8 * 2),
                ("reserved", C.c_uint8 * 3)]


class Mp3FrameInfo(C.Structure):
    """sk_mp3_frame_info"""
    _fields_ = [("offset", C.c_uint32), ("frame_bytes", C.c_uint32), ("sample_rate", C.c_uint32), ("bitrate_kbps", C.c_uint16),
                ("samples_per_channel", C.c_uint16)] + [(n, C.c_uint8) for n in ("version", "channels", "mode", "mode_ext", "has_crc",
                                                                                  "padding", "granules", "side_info_bytes")]


class Mp3GranuleSide(C.Structure):
    """sk_mp3_granule_side"""
    _fields_ = [("part2_3_length", C.c_uint16), ("big_values", C.c_uint16), ("scalefac_compress", C.c_uint16), ("global_gain", C.c_uint8),
                ("window_switching", C.c_uint8), ("block_type", C.c_uint8), ("mixed_block_flag", C.c_uint8), ("table_select", C.c_uint8 * 3),
                ("subblock_gain", C.c_uint8 * 3), ("region0_count", C.c_uint8), ("region1_count", C.c_uint8), ("preflag", C.c_uint8),
                ("scalefac_scale", C.c_uint8), ("count1table_select", C.c_uint8), ("reserved", C.c_uint8)]


class Mp3SideInfo(C.Structure):
    """sk_mp3_side_info"""
    _fields_ = [("main_data_begin", C.c_uint16), ("granules", C.c_uint8), ("channels", C.c_uint8), ("scfsi", (C.c_uint8 * 4) * 2),
                ("gr", (Mp3GranuleSide * 2) * 2)]


class Mp3RequantChannel(C.Structure):
    """sk_mp3_requant_channel"""
    _fields_ = [("global_gain", C.c_uint8), ("scalefac_scale", C.c_uint8), ("preflag", C.c_uint8), ("block_type", C.c_uint8),
                ("mixed_block_flag", C.c_uint8), ("subblock_gain", C.c_uint8 * 3), ("scalefac_l", C.c_uint8 * 22),
                ("scalefac_s", (C.c_uint8 * 3) * 13), ("reserved", C.c_uint8)]


class Mp3RequantGranule(C.Structure):
    """sk_mp3_requant_granule"""
    _fields_ = [("sample_rate", C.c_uint32), ("channels", C.c_uint8), ("ms_stereo", C.c_uint8), ("intensity_stereo", C.c_uint8),
                ("lsf", C.c_uint8), ("ch", Mp3RequantChannel * 2)]


class Mp3CodeTable(C.Structure):
    """sk_mp3_code_table"""
    _fields_ = [("xlen", C.c_uint8), ("linbits", C.c_uint8), ("hlen", C.POINTER(C.c_uint8)), ("hcod", C.POINTER(C.c_uint32))]


class Mp3Tables(C.Structure):
    """sk_mp3_tables"""
    _fields_ = [("big_values", Mp3CodeTable * 32), ("count1_hlen", (C.c_uint8 * 16) * 2), ("count1_hcod", (C.c_uint8 * 16) * 2),
                ("slen", (C.c_uint8 * 2) * 16), ("lsf_partitions", ((C.c_uint8 * 4) * 3) * 6), ("long_offsets", (C.c_uint16 * 23) * 9),
                ("short_offsets", (C.c_uint16 * 14) * 9), ("rates_present", C.c_uint8 * 9), ("pretab", C.c_uint8 * 22),
                ("window", C.c_float * 512)]


class Mp3GranuleData(C.Structure):
    """sk_mp3_granule_data"""
    _fields_ = [("is_", C.c_int16 * 576), ("scalefac_l", C.c_uint8 * 22), ("scalefac_s", (C.c_uint8 * 3) * 13), ("preflag", C.c_uint8),
                ("intensity_scale", C.c_uint8), ("part2_bits", C.c_uint16), ("nonzero_lines", C.c_uint16), ("part3_bits", C.c_uint16), ("status", C.c_int32)]


class TickStream(C.Structure):
    """sk_tick_stream"""
    _fields_ = [("stream", C.c_uint32), ("n_frames", C.c_uint32), ("out_bits", C.c_uint8), ("out_channels", C.c_uint8),
                ("resample", C.c_uint8), ("flush", C.c_uint8), ("codec", C.c_uint8), ("reserved", C.c_uint8 * 3)]


class TickInput(C.Structure):
    """sk_tick_input"""
    _fields_ = [("descs", C.c_void_p), ("coeffs", C.c_void_p), ("units", C.c_void_p), ("au_bytes", C.c_void_p), ("au_bytes_len", C.c_size_t),
                ("q_sides", C.c_void_p), ("q_quant", C.c_void_p), ("n_aac_units", C.c_uint32), ("n_mp3_granules", C.c_uint32),
                ("mp3_granules", C.c_void_p), ("mp3_descs", C.c_void_p), ("mp3_is", C.c_void_p)]


class TickOutput(C.Structure):
    """sk_tick_output"""
    _fields_ = [("stream_index", C.c_uint32), ("frames", C.c_uint32), ("byte_offset", C.c_uint64), ("bytes", C.c_uint32),
                ("status", C.c_int32), ("channels", C.c_uint8), ("bits", C.c_uint8), ("reserved", C.c_uint16)]


class PipelineConfig(C.Structure):
    """sk_pipeline_config"""
    _fields_ = [(n, C.c_uint32) for n in ("entropy_threads", "max_streams", "max_frames_per_tick",
                                          "max_stream_frames_per_tick", "input_buffer", "output_buffer", "tick_wait_us",
                                          "gpu_entropy", "lanes")]


class DecodeOptionsC(C.Structure):
    """sk_decode_options"""
    _fields_ = [("output_sample_rate", C.c_uint32), ("output_bits_per_sample", C.c_uint8), ("output_channels", C.c_uint8),
                ("reserved", C.c_uint16)]


class AudioInfo(C.Structure):
    """sk_audio_info"""
    _fields_ = [("sampling_rate", C.c_uint32), ("frames", C.c_uint32), ("bytes", C.c_uint32), ("status", C.c_int32),
                ("bits_per_sample", C.c_uint8), ("channel_count", C.c_uint8), ("is_error", C.c_uint8), ("reserved", C.c_uint8)]


class PipelineStats(C.Structure):
    """sk_pipeline_stats"""
    _fields_ = [(n, C.c_uint64) for n in ("ticks", "frames", "outputs", "errors", "parse_ns", "tick_ns", "idle_ns", "deliver_ns")] + [
        ("entropy_threads", C.c_uint32), ("lanes", C.c_uint32)]


def declared_symbols():
    """Every function name include/soundkit_amd.h declares."""
    text = open(HEADER_PATH).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(sk_[a-z0-9_]+)\s*\(", text)))


def _preload_hip_runtime():
    """One process must use ONE HIP/HSA runtime.  PyTorch-ROCm wheels bundle their own
    libamdhip64.so (same SONAME libamdhip64.so.7 as /opt/rocm's): if libsoundkit_amd.so pulled in
    the system copy first, a later `import torch` would start a second runtime that finds no device.
    So when torch is installed, bind to its copy (set SOUNDKIT_AMD_HIP_RUNTIME=system to opt out;
    a non-Python host simply links the system ROCm)."""
    if os.environ.get("SOUNDKIT_AMD_HIP_RUNTIME", "torch") != "torch":
        return None
    try:
        import importlib.util
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        spec = None
    if spec is None or not spec.origin:
        return None
    cand = os.path.join(os.path.dirname(spec.origin), "lib", "libamdhip64.so")
    if not os.path.exists(cand):
        return None
    return C.CDLL(cand, mode=C.RTLD_GLOBAL)


def _load():
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            "soundkit_amd: %s is missing -- build it with `make -C soundkit_amd/csrc` "
            "(or python -c 'import __graft_entry__ as g; g.build()'); there is no CPU fallback" % LIB_PATH)
    try:
        _preload_hip_runtime()
        return C.CDLL(LIB_PATH)
    except OSError as exc:  # pragma: no cover - depends on the host
        raise ImportError("soundkit_amd: cannot load %s: %s (no CPU fallback exists)" % (LIB_PATH, exc))


lib = _load()

_vp, _u32, _sz, _i = C.c_void_p, C.c_uint32, C.c_size_t, C.c_int
_sig = {
    "sk_engine_create": (_i, [_i, _u32, C.POINTER(_vp)]),
    "sk_engine_destroy": (None, [_vp]),
    "sk_engine_device": (_i, [_vp]),
    "sk_engine_hip_stream": (_vp, [_vp]),
    "sk_engine_synchronize": (_i, [_vp]),
    "sk_engine_last_hip_error": (C.c_char_p, [_vp]),
    "sk_kernels_use_packed_f32": (C.c_int, []),
    "sk_strerror": (C.c_char_p, [_i]),
    "sk_version": (C.c_char_p, []),
    "sk_stream_open": (_i, [_vp, _u32, C.c_uint8, C.POINTER(_u32)]),
    "sk_stream_close": (_i, [_vp, _u32]),
    "sk_stream_reset": (_i, [_vp, _u32]),
    "sk_stream_get_state": (_i, [_vp, _u32, _vp, _vp]),
    "sk_stream_set_state": (_i, [_vp, _u32, _vp, _vp]),
    "sk_aac_synthesize_f32": (_i, [_vp, _vp, _vp, _vp, _u32, _vp]),
    "sk_aac_synthesize_s16": (_i, [_vp, _vp, _vp, _vp, _u32, _vp]),
    "sk_aac_plan_create": (_i, [_vp, _vp, _u32, _vp, C.POINTER(_vp)]),
    "sk_aac_plan_destroy": (None, [_vp]),
    "sk_aac_plan_elements": (C.c_uint64, [_vp]),
    "sk_aac_plan_frames_ok": (_u32, [_vp]),
    "sk_aac_plan_run_f32_dev": (_i, [_vp, _vp, _vp, _vp]),
    "sk_aac_plan_run_s16_dev": (_i, [_vp, _vp, _vp, _vp]),
    "sk_aac_plan_run_s16_planar_dev": (_i, [_vp, _vp, _vp, _vp]),
    "sk_downsample_48k_16k_frames_s16_to_f32_dev": (_i, [_vp, _vp, _sz, _sz, _u32, _u32, _u32, _vp, _sz, C.POINTER(_u32)]),
    "sk_downsample_48k_16k_frames_s16_to_s16_dev": (_i, [_vp, _vp, _sz, _sz, _u32, _u32, _u32, _vp, _sz, C.POINTER(_u32)]),
    "sk_aac_decoder_create": (_i, [_vp, _sz, C.POINTER(_vp)]),
    "sk_aac_decoder_destroy": (None, [_vp]),
    "sk_aac_decoder_info": (_i, [_vp, C.POINTER(_u32), C.POINTER(C.c_uint8)]),
    "sk_aac_decoder_last_error": (C.c_char_p, [_vp]),
    "sk_aac_decoder_tool_usage": (_i, [_vp, _vp]),
    "sk_aac_decoder_parse": (_i, [_vp, _vp, _sz, _vp, _vp]),
    "sk_aac_decoder_parse_q": (_i, [_vp, _vp, _sz, _vp, _vp, _vp]),
    "sk_tick_run_q": (_i, [_vp, _vp, _u32, _vp, _vp, _vp, _u32, _vp, _sz, _vp, _u32, C.POINTER(_u32), C.POINTER(_sz)]),
    "sk_adts_parse": (_i, [_vp, _sz, C.POINTER(_sz), C.POINTER(_sz), C.POINTER(_sz), _vp]),
    "sk_aac_dequantize_dev": (_i, [_vp, _vp, _vp, _vp, _sz]),
    "sk_aac_dequantize": (_i, [_vp, _vp, _vp, _vp, _sz]),
    "sk_pcm_op_in_bytes": (_i, [_i]),
    "sk_pcm_op_out_bytes": (_i, [_i]),
    "sk_pcm_convert": (_i, [_vp, _i, _vp, _vp, _sz]),
    "sk_pcm_convert_dev": (_i, [_vp, _i, _vp, _vp, _sz]),
    "sk_pcm_fmt_bytes": (_i, [_i]),
    "sk_pcm_bytes_to_f32_planar": (_i, [_vp, _i, _i, _vp, _sz, _u32, _vp]),
    "sk_pcm_bytes_to_f32_planar_dev": (_i, [_vp, _i, _i, _vp, _sz, _u32, _vp]),
    "sk_pcm_f32_planar_to_bytes": (_i, [_vp, _i, _vp, _sz, _u32, _vp]),
    "sk_pcm_f32_planar_to_bytes_dev": (_i, [_vp, _i, _vp, _sz, _u32, _vp]),
    "sk_pcm_downmix_mono": (_i, [_vp, _vp, _sz, _u32, _vp]),
    "sk_pcm_downmix_mono_dev": (_i, [_vp, _vp, _sz, _u32, _vp]),
    "sk_pcm_exact_to_i16": (_i, [_vp, _i, _vp, _sz, _vp]),
    "sk_pcm_exact_to_i16_dev": (_i, [_vp, _i, _vp, _sz, _vp]),
    "sk_downsample_48k_16k_out_frames": (_u32, [_u32]),
    "sk_downsample_48k_16k_taps": (_i, [_vp, _vp]),
    "sk_downsample_48k_16k_f32": (_i, [_vp, _vp, _u32, _u32, _vp, C.POINTER(_u32)]),
    "sk_downsample_48k_16k_f32_dev": (_i, [_vp, _vp, _sz, _u32, _u32, _vp, _sz, C.POINTER(_u32)]),
    "sk_downsample_48k_16k_frames_dev": (_i, [_vp, _vp, _sz, _sz, _u32, _u32, _u32, _vp, _sz, C.POINTER(_u32)]),
    "sk_pcm_f32_planar_to_bytes_batch_dev": (_i, [_vp, _i, _vp, _sz, _sz, _sz, _u32, _vp]),
    "sk_downsample_out_frames": (_u32, [_u32, _u32, _u32]),
    "sk_downsample_f32": (_i, [_vp, _vp, _u32, _u32, _u32, _u32, _vp, _u32, C.POINTER(_u32)]),
    "sk_downsample_f32_dev": (_i, [_vp, _vp, _sz, _u32, _u32, _u32, _u32, _vp, _sz, C.POINTER(_u32)]),
    "sk_downsample_48k_16k_frames_s16_dev": (_i, [_vp, _vp, _sz, _sz, _u32, _u32, _u32, _vp, _sz, C.POINTER(_u32)]),
    "sk_resampler_open": (_i, [_vp, _u32, _u32, _u32]),
    "sk_resampler_close": (_i, [_vp, _u32]),
    "sk_resampler_process_f32": (_i, [_vp, _vp, _u32, _vp, _u32, _vp, _u32, _vp]),
    "sk_resampler_flush_f32": (_i, [_vp, _vp, _u32, _vp, _u32, _vp]),
    "sk_adts_decoder_create": (_i, [_vp, C.POINTER(_vp)]),
    "sk_adts_decoder_destroy": (None, [_vp]),
    "sk_adts_decoder_decode_i16": (_i, [_vp, _vp, _sz, _vp, _sz, C.POINTER(_sz)]),
    "sk_adts_decoder_decode_f32": (_i, [_vp, _vp, _sz, _vp, _sz, C.POINTER(_sz)]),
    "sk_adts_decoder_info": (_i, [_vp, C.POINTER(_u32), C.POINTER(C.c_uint8)]),
    "sk_adts_decoder_last_error": (C.c_char_p, [_vp]),
    "sk_pipeline_create": (_i, [_vp, _vp, C.POINTER(_vp)]),
    "sk_pipeline_destroy": (None, [_vp]),
    "sk_pipeline_spawn": (_i, [_vp, _vp, C.POINTER(_u32)]),
    "sk_pipeline_send": (_i, [_vp, _u32, _vp, _sz]),
    "sk_pipeline_finish": (_i, [_vp, _u32]),
    "sk_pipeline_try_recv": (_i, [_vp, _u32, _vp, _sz, _vp]),
    "sk_pipeline_recv": (_i, [_vp, _u32, _vp, _sz, _vp, _u32]),
    "sk_pipeline_cancel": (_i, [_vp, _u32]),
    "sk_pipeline_wait_outputs": (_i, [_vp, _vp, _u32, _u32]),
    "sk_pipeline_queued_input_bytes": (_sz, [_vp, _u32]),
    "sk_pipeline_get_stats": (_i, [_vp, _vp]),
    "sk_tick_run_au": (_i, [_vp, _vp, _u32, _vp, _u32, _vp, _sz, _vp, _sz, _vp, _u32, C.POINTER(_u32), C.POINTER(_sz)]),
    "sk_mp3_set_synthesis_window": (_i, [_vp, _vp]),
    "sk_mp3_hybrid_synthesize_f32": (_i, [_vp, _vp, _vp, _vp, _u32, _vp]),
    "sk_mp3_hybrid_synthesize_s16": (_i, [_vp, _vp, _vp, _vp, _u32, _vp]),
    "sk_mp3_hybrid_synthesize_f32_dev": (_i, [_vp, _vp, _vp, _vp, _u32, _vp]),
    "sk_mp3_parse_header": (_i, [_vp, _sz, _vp]),
    "sk_mp3_parse_side_info": (_i, [_vp, _sz, _vp, _vp]),
    "sk_mp3_scan": (_i, [_vp, _sz, _vp, _u32, C.POINTER(_u32), C.POINTER(_sz)]),
    "sk_mp3_scan_free": (_i, [_vp, _sz, _vp, _u32, C.POINTER(_u32), C.POINTER(_sz), C.POINTER(_u32)]),
    "sk_mp3_parse_header_free": (_i, [_vp, _sz, _u32, _vp]),
    "sk_mp3_main_data": (_i, [_vp, _sz, _vp, _vp, _vp, _sz, _vp, _sz, C.POINTER(_sz)]),
    "sk_mp3_decode_granules_f32": (_i, [_vp, _vp, _vp, _vp, _vp, _u32, _vp]),
    "sk_mp3_decode_granules_s16": (_i, [_vp, _vp, _vp, _vp, _vp, _u32, _vp]),
    "sk_mp3_codebook_create": (_i, [_vp, C.POINTER(_vp)]),
    "sk_mp3_codebook_destroy": (None, [_vp]),
    "sk_mp3_iso_tables": (_i, [C.POINTER(Mp3Tables)]),
    "sk_mp3_codebook_create_iso": (_i, [C.POINTER(_vp)]),
    "sk_mp3_decode_main_data": (_i, [_vp, _vp, _vp, _vp, _sz, _vp]),
    "sk_mp3_decoder_create": (_i, [_vp, _vp, C.POINTER(_vp)]),
    "sk_mp3_decoder_destroy": (None, [_vp]),
    "sk_mp3_decoder_reset": (_i, [_vp]),
    "sk_mp3_decoder_info": (_i, [_vp, C.POINTER(_u32), C.POINTER(C.c_uint8), C.POINTER(_sz), C.POINTER(C.c_uint64)]),
    "sk_mp3_decoder_decode_i16": (_i, [_vp, _vp, _sz, _vp, _sz, C.POINTER(_sz)]),
    "sk_mp3_decoder_decode_i32": (_i, [_vp, _vp, _sz, _vp, _sz, C.POINTER(_sz)]),
    "sk_mp3_decoder_decode_f32": (_i, [_vp, _vp, _sz, _vp, _sz, C.POINTER(_sz)]),
    "sk_mp3_set_band_tables": (_i, [_vp, _u32, _vp, _vp, _vp]),
    "sk_mp3_requantize": (_i, [_vp, _vp, _vp, _vp, _u32, _vp]),
    "sk_aac_entropy_decode": (_i, [_vp, _vp, _vp, _u32, _vp, _u32, _vp, _sz, _vp, _vp, _vp]),
    "sk_aac_plan_run_tail_s16_dev": (_i, [_vp, _vp, _vp, _sz, _u32, _u32, _vp, _sz, C.POINTER(_u32)]),
    "sk_aac_expand_q_decode": (_i, [_vp, _vp, _vp, _u32, _vp, _vp, _vp, _u32, _vp, _vp, _vp]),
    "sk_engine_where": (C.c_char_p, [_vp]),
    "sk_last_exception": (C.c_char_p, []),
    "sk_debug_throw_after": (_i, [_i, _i]),
    "sk_debug_throw_in_thread": (_i, [_i, _i]),
    "sk_engine_set_wait_bound": (_i, [_vp, C.c_double]),
    "sk_engine_set_resampler_exact": (_i, [_vp, _i]),
    "sk_engine_debug_fail_after": (_i, [_vp, _i]),
    "sk_pipeline_debug_dump": (_sz, [_vp, _vp, _sz]),
    "sk_tick_out_bound": (_sz, [_vp, _u32, C.POINTER(_u32)]),
    "sk_tick_out_bound_on": (_sz, [_vp, _vp, _u32, C.POINTER(_u32)]),
    "sk_tick_run": (_i, [_vp, _vp, _u32, _vp, _vp, _u32, _vp, _sz, _vp, _u32, C.POINTER(_u32), C.POINTER(_sz)]),
    "sk_tick_run_mixed": (_i, [_vp, _vp, _u32, _vp, _vp, _sz, _vp, _u32, C.POINTER(_u32), C.POINTER(_sz)]),
}
for _name in ("sk_pcm_interleave_i16", "sk_pcm_deinterleave_i16", "sk_pcm_deinterleave_s24", "sk_pcm_deinterleave_f32",
              "sk_pcm_interleave_f32"):
    _sig[_name] = (_i, [_vp, _vp, _sz, _u32, _vp])
    _sig[_name + "_dev"] = (_i, [_vp, _vp, _sz, _u32, _vp])

for _name, (_res, _args) in _sig.items():
    _fn = getattr(lib, _name)  # AttributeError here = the library lacks a declared symbol
    _fn.restype = _res
    _fn.argtypes = _args


class SoundkitError(RuntimeError):
    """A negative sk_status; the reference's errors on this path are Strings / DecodeError
    (soundkit-decoder/src/lib.rs:108-118), so the message is what callers see."""

    def __init__(self, status, what="", detail=""):
        self.status = status
        name = ERR_NAMES.get(status, str(status))
        msg = lib.sk_strerror(status).decode()
        super().__init__("%s failed: %s (%s)%s" % (what, name, msg, (": " + detail) if detail else ""))


def check(status, what, engine_handle=None):
    if status != SK_OK:
        detail = ""
        if engine_handle is not None and status in (-3, -4, -8):
            detail = lib.sk_engine_last_hip_error(engine_handle).decode()
        if status in (-4, -9) and not detail:
            detail = lib.sk_last_exception().decode()
        raise SoundkitError(status, what, detail)
