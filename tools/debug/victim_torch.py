"""Library-free victims for the co-residency experiments: stock PyTorch/rocFFT/rocBLAS kernels on fixed inputs, each compared bit for
bit with its own first result.  Run it alone (every line must say 0) and beside tools/probe/mfma_aggressor.hip in another process.
python tools/debug/victim_torch.py [runs] [only the victims whose name starts with this]"""
import sys

import torch

runs = int(sys.argv[1]) if len(sys.argv) > 1 else 30
dev = torch.device("cuda:0")
g = torch.Generator(device="cpu").manual_seed(11)
x = (torch.rand((65536, 1024), generator=g) * 24 - 12).to(dev)
xc = torch.complex(x[:, :512].contiguous(), x[:, 512:].contiguous())
m = (torch.rand((4096, 4096), generator=g) - 0.5).to(dev)
mb = m.to(torch.bfloat16)

victims = {
    "rocFFT c64 512-point x 65536": lambda: torch.view_as_real(torch.fft.fft(xc, dim=1)),
    "rocFFT r2c 1024-point x 65536": lambda: torch.view_as_real(torch.fft.rfft(x, dim=1)),
    "elementwise x*1.0001+0.5 (x8)": lambda: ((((((((x * 1.0001 + 0.5) * 0.999 - 0.25) * 1.0002 + 0.125) * 0.9995) + 1.5) * 1.0003) - 0.75) * 0.9998),
    "softmax rows of 1024": lambda: torch.softmax(x, dim=1),
    "cumsum rows of 1024": lambda: torch.cumsum(x, dim=1),
    "sort rows of 1024": lambda: torch.sort(x, dim=1).values,
    "matmul f32 4096^3": lambda: m @ m,
    "matmul bf16 4096^3": lambda: (mb @ mb).float(),
}
only = sys.argv[2] if len(sys.argv) > 2 else ""
for name, fn in victims.items():
    if not name.startswith(only):
        continue
    first = fn().clone()
    torch.cuda.synchronize()
    bad, words, worst = 0, 0, 0.0
    for _ in range(runs):
        y = fn()
        torch.cuda.synchronize()
        ne = y.view(torch.int32) != first.view(torch.int32)
        k = int(ne.sum().item())
        if k:
            bad += 1
            words += k
            worst = max(worst, float((y.double() - first.double()).abs().max().item()))
    print(f"{name:34s} {bad} of {runs} runs differ, {words} words, max abs {worst:.3g} (rms {float(first.double().pow(2).mean().sqrt()):.3g})", flush=True)
