"""The batch scheduler's host logic without a GPU: csrc/pipeline.cpp and the real AAC front-end compiled together
with a stand-in engine (tests/sched_stub.cpp) and run under ThreadSanitizer, then AddressSanitizer + UBSan.
Scenarios: 32 streams fed in ragged chunks from two threads (every access unit delivered once, in order, to its own
stream), input/output backpressure and the 4 MiB chunk limit, a corrupted access unit ending only its own stream
after the outputs that precede it, garbage input, and cancel/respawn churn with work in flight (no handle or
engine stream leaked), a failing tick (its batch's streams end with one error each, nothing stalls), streams that never
frame mixed with real ones over several delivery threads, and the slow-tick regime under the bench's load generator.
The GPU suite (test_scheduler_gpu.py) checks the same scheduler for sample-exact output."""
import os
import subprocess

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
CLIP = os.path.join(HERE, "golden", "aac", "aac-stereo-48k.adts")


@pytest.mark.parametrize("sanitizer", ["thread", "address,undefined"])
def test_scheduler_scenarios_under_sanitizers(tmp_path, sanitizer):
    exe = str(tmp_path / "sched_stub")
    subprocess.check_call(["g++", "-O1", "-g", "-std=c++17", "-fsanitize=" + sanitizer, "-Wno-subobject-linkage",
                           "-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include", "-o", exe, os.path.join(HERE, "sched_stub.cpp"),
                           "-lpthread"], cwd=HERE)
    out = subprocess.run([exe, CLIP], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0 and "scheduler scenarios ok" in out.stdout, (out.stdout[-1000:], out.stderr[-3000:])
    assert "ThreadSanitizer" not in out.stderr and "AddressSanitizer" not in out.stderr and "runtime error" not in out.stderr
    # the slow-tick regime (the tick as the slowest stage, all three batches in rotation, outputs at their bound, feeders
    # bouncing off InputBufferFull), driven by the bench's own load generator with its progress deadline armed: drawn
    # shapes, every third one the quantised hand-over; `tools/sched_soak.sh` runs the same binary mode for thousands
    out = subprocess.run([exe, CLIP, "slow", "6", "7"], capture_output=True, text=True, timeout=900)
    assert out.returncode == 0 and "slow-tick scenarios ok" in out.stdout, (out.stdout[-1000:], out.stderr[-3000:])
    assert "STALL" not in out.stderr
    assert "ThreadSanitizer" not in out.stderr and "AddressSanitizer" not in out.stderr and "runtime error" not in out.stderr
