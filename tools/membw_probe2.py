#!/usr/bin/env python3
"""Probe: in-place read-modify-write vs out-of-place (read A, write B) streaming at the same
1:1 read:write byte mix, f64, 4 GiB per buffer."""
import time
import torch

dev = torch.device("cuda:0")
n = (4 << 30) // 8
x = torch.ones(n, dtype=torch.float64, device=dev)
y = torch.empty_like(x)


def bench(fn, bytes_moved, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return bytes_moved * reps / (time.perf_counter() - t0) / 1e12


print(f"in-place  x.mul_(c)              : {bench(lambda: x.mul_(1.0000001), 2 * n * 8):.2f} TB/s (r+w)")
print(f"out-of-pl torch.mul(x,c,out=y)   : {bench(lambda: torch.mul(x, 1.0000001, out=y), 2 * n * 8):.2f} TB/s (r+w)")
print(f"copy      y.copy_(x)             : {bench(lambda: y.copy_(x), 2 * n * 8):.2f} TB/s (r+w)")
print(f"read-only x.sum()                : {bench(lambda: x.sum(), n * 8):.2f} TB/s (r)")
print(f"write-only y.fill_(1)            : {bench(lambda: y.fill_(1.0), n * 8):.2f} TB/s (w)")
