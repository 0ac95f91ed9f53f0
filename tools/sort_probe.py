#!/usr/bin/env python3
"""Would grouping syndromes of similar difficulty into the same 64-syndrome tile pay?  A tile runs until its
slowest lane has converged; the syndrome weight predicts the iteration count.  Decodes the same batch in
its given order and sorted by syndrome weight (sorting done here with torch, outside the timed region)."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ldpcdecoders_jl_amd as ldpc
import bench

n, B = 16384, 65536
H = ldpc.codes.parity_check_csc(n, 8, 4)
for per in (0.02, 0.04, 0.06):
    dec = ldpc.BeliefPropagationDecoder(H, per, 50)
    syn = bench.make_syndromes(torch, H.tocsr(), n, B, per, seed=5, device=torch.device("cuda:0"))
    w = syn.sum(dim=1, dtype=torch.int32)
    order = torch.argsort(w)
    syn_sorted = syn[order].contiguous()
    err = torch.empty((B, n), dtype=torch.uint8, device="cuda"); conv = torch.empty(B, dtype=torch.uint8, device="cuda")
    its = torch.empty(B, dtype=torch.int32, device="cuda")
    res = {}
    for name, s_in in (("given order", syn), ("sorted by weight", syn_sorted)):
        for _ in range(2):
            dec.decode_batch_device(s_in, err, conv, None, its)
        torch.cuda.synchronize()
        ts = []
        for _ in range(4):
            t0 = time.perf_counter(); dec.decode_batch_device(s_in, err, conv, None, its); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
        res[name] = (np.median(ts) * 1e3, its.float().mean().item(), its.clone(), err.clone())
        print(f"per {per}: {name:18s} {res[name][0]:8.2f} ms   mean iterations {res[name][1]:.2f}", flush=True)
    assert torch.equal(res["given order"][2][order], res["sorted by weight"][2]) and torch.equal(res["given order"][3][order], res["sorted by weight"][3])
    dec.close()
