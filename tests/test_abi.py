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
