#!/usr/bin/env python3
"""Throughput of the HOST-buffer entry (what a Julia batchdecode! call hits: host arrays in,
host arrays out, PCIe included) next to the HBM-resident entry, for the small-code workloads."""
import os, sys, time
import numpy as np
import scipy.sparse as sp
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import ldpcdecoders_jl_amd as ldpc

cases = [("BB-72 per=0.005", sp.csc_matrix(ldpc.codes.bivariate_bicycle_72_12_6()[0]), 0.005, 1 << 20),
         ("(3,6) n=1008 per=0.01", ldpc.codes.parity_check_csc(1008, 6, 3), 0.01, 1 << 18),
         ("(4,8) n=16384 per=0.02", ldpc.codes.parity_check_csc(16384, 8, 4), 0.02, 1 << 14)]
for name, H, per, B in cases:
    H.sort_indices()
    n = H.shape[1]
    E = ldpc.codes.random_errors(n, min(B, 1 << 16), per, seed=1)
    syn = ldpc.codes.syndromes_of(H, E)
    syn = np.ascontiguousarray(np.tile(syn, (B // syn.shape[0], 1)))
    dec = ldpc.BeliefPropagationDecoder(H, per, 50)
    out = (np.empty((B, n), dtype=np.uint8), np.empty(B, dtype=np.uint8))   # caller-owned, reused
    for _ in range(2):
        dec.decode_batch_host(syn, out=out)
    t0 = time.perf_counter(); reps = 3
    for _ in range(reps):
        dec.decode_batch_host(syn, out=out)
    th = (time.perf_counter() - t0) / reps
    d_syn = torch.from_numpy(syn).cuda()
    d_err = torch.empty((B, n), dtype=torch.uint8, device="cuda"); d_conv = torch.empty(B, dtype=torch.uint8, device="cuda")
    for _ in range(2):
        dec.decode_batch_device(d_syn, d_err, d_conv)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps):
        dec.decode_batch_device(d_syn, d_err, d_conv)
    torch.cuda.synchronize(); td = (time.perf_counter() - t0) / reps
    mb = (syn.nbytes + B * n + B) / 1e6
    print(f"{name:26s} B={B:8d}  host entry {th*1e3:8.2f} ms ({B/th/1e6:7.2f} M/s, {mb/th/1e3:5.1f} GB/s PCIe-equivalent)   "
          f"device entry {td*1e3:8.2f} ms ({B/td/1e6:7.2f} M/s)")
