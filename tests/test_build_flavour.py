"""The library's kernel flavour is what it says it is.  The default build must not contain a single packed-f32 vector instruction
(v_pk_mul/add/fma_f32): on MI355X they deliver a wrong low half now and then while another wave of the CU executes a 16x16x32 matrix
instruction (profiles/r04_lanes_corruption.md), and the immunity of the default build rests on their absence from EVERY kernel --
a builtin or a line of inline assembly that brings one back would not fail any parity test."""
import os
import re
import shutil
import subprocess

import pytest

from soundkit_amd import _lib

OBJDUMP = "/opt/rocm/lib/llvm/bin/llvm-objdump"


@pytest.mark.skipif(not os.path.exists(OBJDUMP), reason="llvm-objdump of the ROCm toolchain not found")
def test_default_build_has_no_packed_f32_instructions(tmp_path):
    so = str(tmp_path / "libsoundkit_amd.so")
    shutil.copy(_lib.LIB_PATH, so)
    subprocess.check_call([OBJDUMP, "--offloading", so], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)  # writes the bundles beside its input
    objects = [f for f in os.listdir(tmp_path) if "gfx950" in f]
    assert len(objects) >= 7, objects  # one code object per .hip file
    packed, total = 0, 0
    for f in objects:
        text = subprocess.run([OBJDUMP, "-d", str(tmp_path / f)], capture_output=True, text=True, check=True).stdout
        total += len(re.findall(r"^\s+v_", text, re.M))
        packed += len(re.findall(r"\bv_pk_(?:mul|add|fma)_f32\b", text))
    assert total > 50000, total  # the disassembly is really there
    if _lib.lib.sk_kernels_use_packed_f32() == 0:
        assert packed == 0, packed
    else:
        assert packed > 0  # make PACKED_F32=1: the flavour that needs its GPU to itself
