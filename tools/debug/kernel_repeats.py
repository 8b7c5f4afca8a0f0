"""Does every matrix-instruction kernel of the library repeat itself bit for bit at its bench size?  (A kernel whose own waves do FFT-
shaped work beside its matrix instructions would not, on this platform: profiles/r04_lanes_corruption.md.)  Fixed inputs, N runs each.
python tools/debug/kernel_repeats.py [runs] [streams]"""
import sys

import torch

sys.path.insert(0, ".")
import soundkit_amd  # noqa: E402

runs = int(sys.argv[1]) if len(sys.argv) > 1 else 10
streams = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
dev = torch.device("cuda:0")
eng = soundkit_amd.Engine(0, 64)
ext = torch.cuda.ExternalStream(eng.hip_stream, device=dev)
rows = streams * 2
g = torch.Generator(device="cpu").manual_seed(3)


def repeat(name, fn, out):
    fn()
    eng.synchronize()
    first = out.clone()
    bad = 0
    for _ in range(runs):
        out.zero_()
        fn()
        eng.synchronize()
        bad += int(not torch.equal(out.view(torch.int16 if out.dtype == torch.int16 else torch.int32), first.view(torch.int16 if out.dtype == torch.int16 else torch.int32)))
    print(f"{name:70s} {bad} of {runs} runs differ from the first", flush=True)


with torch.cuda.stream(ext):
    x48 = (torch.rand((rows, 48000), generator=g) * 2 - 1).to(dev)
    n16 = eng.downsample_out_frames(48000)
    y = torch.empty((rows, n16), device=dev)
    torch.cuda.synchronize()
    repeat("48k->16k FIR, f32 rows (k_fir_*: v_mfma_f32_16x16x32_bf16)", lambda: eng.downsample_48k_16k_dev(x48, 48000, rows, 48000, y, n16), y)
    x44 = x48[:, :44100].contiguous()
    n44 = eng.downsample_out_frames(44100, 44100, 16000)
    y2 = torch.empty((rows, n44), device=dev)
    torch.cuda.synchronize()
    repeat("44.1k->16k generic resampler (k_sinc_mfma: v_mfma_f32_16x16x32_bf16)", lambda: eng.downsample_dev(x44, 44100, rows, 44100, 44100, 16000, y2, n44), y2)
    n48 = eng.downsample_out_frames(44100, 44100, 48000)
    y3 = torch.empty((rows, n48), device=dev)
    torch.cuda.synchronize()
    repeat("44.1k->48k generic resampler", lambda: eng.downsample_dev(x44, 44100, rows, 44100, 44100, 48000, y3, n48), y3)
