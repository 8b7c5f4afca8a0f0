"""A foreign load for the co-residency experiments: plain PyTorch work on the same GPU for N seconds (copy = HBM traffic,
matmul = matrix cores, lds = a softmax over short rows)."""
import sys
import time

import torch

kind, secs = sys.argv[1], float(sys.argv[2])
dev = torch.device("cuda:0")
a = torch.empty(256 << 20, dtype=torch.uint8, device=dev)
b = torch.empty_like(a)
m = torch.randn(4096, 4096, device=dev, dtype=torch.bfloat16)
x = torch.randn(1 << 20, 64, device=dev)
t0, n = time.time(), 0
while time.time() - t0 < secs:
    for _ in range(20):
        if kind == "copy":
            b.copy_(a)
        elif kind == "matmul":
            m2 = m @ m
        else:
            y = torch.softmax(x, dim=1)
    torch.cuda.synchronize()
    n += 20
print("aggressor", kind, "iterations", n)
