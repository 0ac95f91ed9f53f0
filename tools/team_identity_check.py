#!/usr/bin/env python3
"""C3 code (n 16384): tile kernel, team kernel and auto-dispatch must agree bit for bit -- hard decisions,
flags, iteration counts and LLR bit patterns -- for batches from one syndrome (64 workgroups on one tile,
dealt over all XCDs) to 9000 (141 tiles, teams of 3).  A manual companion of tests/test_gpu_parity.py's
n = 4096 identity test at the size the oracle cannot check in seconds."""
import os, sys, torch, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ldpcdecoders_jl_amd as ldpc
n=16384
H = ldpc.codes.parity_check_csc(n, 8, 4)
for B, per in [(9000, 0.06), (700, 0.06), (5000, 0.045), (130, 0.07), (1, 0.06), (63, 0.05)]:
    syn = torch.from_numpy(ldpc.codes.syndromes_of(H, ldpc.codes.random_errors(n, B, per, seed=B))).cuda()
    res = {}
    for variant in (1, 4, 0):
        dec = ldpc.BeliefPropagationDecoder(H, per, 50, kernel_variant=variant)
        err = torch.empty((B, n), dtype=torch.uint8, device="cuda"); conv = torch.empty(B, dtype=torch.uint8, device="cuda")
        llr = torch.full((B, n), float("nan"), dtype=torch.float64, device="cuda"); its = torch.empty(B, dtype=torch.int32, device="cuda")
        dec.decode_batch_device(syn, err, conv, llr, its); torch.cuda.synchronize()
        res[variant] = (err, conv, its, llr.view(torch.int64)); info = dec.info(); k = (info.last_kernel, info.last_team_size)
        print(B, per, "variant", variant, "kernel", k, "conv %.3f" % conv.float().mean().item(), "iters", int(its.min()), int(its.max()), flush=True)
        dec.close()
    for v in (4, 0):
        assert all(torch.equal(a, b) for a, b in zip(res[1], res[v])), (B, per, v)
print("all identical")
