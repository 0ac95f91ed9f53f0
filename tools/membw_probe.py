#!/usr/bin/env python3
"""Probe: read+write bandwidth of an in-place sweep as a function of the working set
(does the 256 MiB Infinity Cache serve a re-swept working set faster than HBM?).
torch is used only as a convenient streaming-kernel launcher."""
import time
import torch

dev = torch.device("cuda:0")
for mib in [16, 32, 64, 128, 192, 256, 512, 2048, 8192]:
    n = mib * (1 << 20) // 8
    x = torch.ones(n, dtype=torch.float64, device=dev)
    reps = max(4, min(400, (32 << 30) // (mib << 20) // 2))
    for _ in range(3):
        x.mul_(1.0000001)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        x.mul_(1.0000001)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(f"working set {mib:6d} MiB  in-place f64 mul: {2 * mib * (1 << 20) * reps / dt / 1e12:6.2f} TB/s (r+w)  "
          f"{dt / reps * 1e6:8.1f} us/sweep", flush=True)
    del x
