#!/usr/bin/env python3
"""What this box streams: a plain copy, a read-only pass and a write-only pass over the synthesis kernel's byte count
(2.18 GB each way), timed with events.  The ceiling the HBM-bound kernels are compared with."""
import torch

n = 4096 * 64 * 2 * 1024
x = torch.rand(n, device="cuda")
y = torch.empty_like(x)


def timed(fn, reps=20):
    for _ in range(3):
        fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps


ms = timed(lambda: y.copy_(x))
print("copy      %.3f ms  %.2f TB/s (read + write)" % (ms, 2 * n * 4 / ms / 1e9))
ms = timed(lambda: x.sum())
print("read-only %.3f ms  %.2f TB/s" % (ms, n * 4 / ms / 1e9))
ms = timed(lambda: y.fill_(1.0))
print("write-only %.3f ms  %.2f TB/s" % (ms, n * 4 / ms / 1e9))
ms = timed(lambda: torch.add(x, 1.0, out=y))
print("add       %.3f ms  %.2f TB/s (read + write)" % (ms, 2 * n * 4 / ms / 1e9))
