"""The C-ABI library loads and exports every symbol include/soundkit_amd.h declares (no GPU)."""
import ctypes as C
import subprocess

import numpy as np
import pytest

import soundkit_amd
from soundkit_amd import _lib


def test_library_exports_every_declared_symbol():
    declared = soundkit_amd.declared_symbols()
    assert len(declared) >= 50
    out = subprocess.check_output(["nm", "-D", "--defined-only", soundkit_amd.LIB_PATH], text=True)
    exported = {line.split()[-1] for line in out.splitlines() if " T " in line}
    missing = [s for s in declared if s not in exported]
    assert not missing, missing
    # and nothing is exported under the sk_ prefix that the header does not declare
    extra = [s for s in exported if s.startswith("sk_") and s not in declared]
    assert not extra, extra


def test_no_torch_types_in_signatures():
    text = open(_lib.HEADER_PATH).read()
    assert "torch" not in text and "at::" not in text and "Tensor" not in text
    assert 'extern "C"' in text


def test_frame_desc_layout():
    assert C.sizeof(_lib.FrameDesc) == 12
    descs, n = soundkit_amd.descs_from_arrays([3, 7], 2, [[0, 1], [2, 3]], [[1, 0], [0, 1]])
    assert n == 2 and descs[1].stream == 7 and descs[1].channels == 2
    assert list(descs[0].window_sequence) == [0, 1] and list(descs[1].window_shape) == [0, 1]
    d2, n2 = soundkit_amd.make_descs([(3, 2, (0, 1), (1, 0)), (7, 2, (2, 3), (0, 1))])
    assert bytes(d2)[:24] == bytes(descs)[:24]


def test_static_tables_match_oracle(oracle):
    for op in range(29):
        assert _lib.lib.sk_pcm_op_in_bytes(op) == oracle.lib().sko_op_in_bytes(op)
        assert _lib.lib.sk_pcm_op_out_bytes(op) == oracle.lib().sko_op_out_bytes(op)
    assert _lib.lib.sk_pcm_op_in_bytes(29) == -1
    for fmt in range(8):
        assert _lib.lib.sk_pcm_fmt_bytes(fmt) == oracle.fmt_bytes(fmt)
    assert soundkit_amd.engine.PCM_OPS == oracle.OPS


@pytest.mark.parametrize("frames", [0, 1, 131, 132, 133, 134, 135, 136, 1000, 4096, 48000, 48001, 48002])
def test_out_frames_formula_matches_oracle(oracle, frames):
    want = 0
    if frames:
        x = np.zeros((1, frames), np.float32)
        want = oracle.downsample_planar(x, 48000, 16000).shape[1]
    assert soundkit_amd.Engine.downsample_out_frames(frames) == want


def test_strerror_and_version():
    assert _lib.lib.sk_strerror(0) == b"ok"
    assert b"stream" in _lib.lib.sk_strerror(-5)
    assert b"gfx950" in _lib.lib.sk_version()


def test_engine_create_fails_loudly_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    with pytest.raises(soundkit_amd.SoundkitError) as exc:
        soundkit_amd.Engine(0, 16)
    assert exc.value.status == -2  # SK_ERR_NO_DEVICE: there is no CPU path to fall back to


def test_product_never_imports_the_oracle():
    import os
    root = os.path.dirname(os.path.abspath(soundkit_amd.__file__))
    for dirpath, _, files in os.walk(root):
        for f in files:
            if f.endswith((".py", ".cpp", ".hip", ".h")):
                text = open(os.path.join(dirpath, f), errors="ignore").read()
                assert "sk_oracle" not in text and "from oracle" not in text and "import oracle" not in text, f


# ---- the exception barrier (csrc/sk_abi.h) -----------------------------------------------------------------------------

def _entry_points_of(path):
    """(name, text between the signature's ')' and the body's '{') of every sk_* function defined at the top level of an
    extern "C" block"""
    import re
    src = open(path).read()
    src = re.sub(r"//[^\n]*", "", src)
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    out = []
    for block in re.finditer(r'extern "C" \{', src):
        depth, i = 1, block.end()
        start = i
        while depth and i < len(src):
            c = src[i]
            if c == '"':
                i += 1
                while src[i] != '"':
                    i += 2 if src[i] == "\\" else 1
            elif c == "'":
                i += 1
                while src[i] != "'":
                    i += 2 if src[i] == "\\" else 1
            elif c == "{":
                if depth == 1:
                    head = src[start:i]
                    m = re.search(r"\b(sk_\w+)\s*\([^;{}]*\)\s*(\w*)\s*$", head, flags=re.S)
                    if m and not re.search(r"\bstatic\b", head[max(0, m.start() - 40):m.start()]):
                        out.append((m.group(1), m.group(2)))
                depth += 1
            elif c == "}":
                depth -= 1
                if depth == 1:
                    start = i + 1
            elif c == ";" and depth == 1:
                start = i + 1
            i += 1
    return out


def test_every_entry_point_is_a_function_try_block():
    """no C++ exception may cross the C ABI: each extern "C" definition in the host sources catches everything"""
    import os
    csrc = os.path.join(os.path.dirname(os.path.abspath(soundkit_amd.__file__)), "csrc")
    exempt = {"sk_last_exception", "sk_debug_throw_after", "sk_debug_throw_in_thread"}  # touch atomics / a thread_local buffer only
    seen = set()
    for name in ("engine.cpp", "pipeline.cpp", "adts_decoder.cpp", "mp3_decoder.cpp", "mp3_bitstream.cpp", "aac_frontend.cpp"):
        for fn, after in _entry_points_of(os.path.join(csrc, name)):
            seen.add(fn)
            assert after == "try" or fn in exempt, "%s in %s has no exception barrier" % (fn, name)
    declared = set(soundkit_amd.declared_symbols())
    assert declared <= seen, sorted(declared - seen)


@pytest.mark.parametrize("kind, status", [(0, -4), (1, -9), (2, -9)])
def test_a_throw_inside_the_library_comes_back_as_a_status(kind, status):
    """sk_debug_throw_after: the next entry throws std::bad_alloc / std::length_error / a foreign type inside the library;
    the caller sees SK_ERR_OOM / SK_ERR_INTERNAL and the text, and the process is alive for the next call"""
    lib = _lib.lib
    data = np.frombuffer(b"\xff\xfb\x90\x00" * 64, np.uint8)
    frames = (_lib.Mp3FrameInfo * 8)()
    n, used = C.c_uint32(0), C.c_size_t(0)
    calls = [
        ("sk_mp3_scan", lambda: lib.sk_mp3_scan(data.ctypes.data_as(C.c_void_p), data.size, frames, 8, C.byref(n), C.byref(used))),
        ("sk_mp3_codebook_create_iso", lambda: lib.sk_mp3_codebook_create_iso(C.byref(C.c_void_p()))),
        ("sk_aac_decoder_create", lambda: lib.sk_aac_decoder_create(np.array([0x11, 0x90], np.uint8).ctypes.data_as(C.c_void_p), 2, C.byref(C.c_void_p()))),
        ("sk_engine_create", lambda: lib.sk_engine_create(0, 16, C.byref(C.c_void_p()))),
    ]
    for name, call in calls:
        assert lib.sk_debug_throw_after(0, kind) == -1
        try:
            rc = call()
        finally:
            lib.sk_debug_throw_after(-1, 0)
        assert rc == status, (name, rc)
        assert lib.sk_last_exception().decode().startswith(name + ":")
    assert lib.sk_mp3_scan(data.ctypes.data_as(C.c_void_p), data.size, frames, 8, C.byref(n), C.byref(used)) == 0
    # entry points that return no status swallow it
    lib.sk_debug_throw_after(0, kind)
    lib.sk_mp3_codebook_destroy(None)
    assert lib.sk_debug_throw_after(-1, 0) == -1  # the countdown was consumed by that call
