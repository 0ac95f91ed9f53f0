#!/usr/bin/env python3
"""Small batches on codes too large for the LDS kernel: the tile kernel (kernel_variant 1, one syndrome
per lane) against the node-parallel kernel (kernel_variant 3, one workgroup per syndrome), ms per
call through the device-resident entry, next to the edge-list CPU oracle (one core).  Used to place
the auto-dispatch crossover."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import ldpcdecoders_jl_amd as ldpc
from oracle import BPOracle

cases = [(4096, 0.02), (16384, 0.02), (16384, 0.10), (65536, 0.02)]
if os.environ.get("CASES"):
    cases = [(int(c.split(":")[0]), float(c.split(":")[1])) for c in os.environ["CASES"].split(",")]
AUTO = os.environ.get("AUTO") == "1"   # AUTO=1: time kernel_variant 0 (what a caller gets) instead of the forced node kernel
ALL = os.environ.get("ALL") == "1"     # ALL=1: time tile, node, team AND auto; print auto / best of the forced three
batches = [int(x) for x in os.environ.get("BATCHES", "1,64,256,512,1024,2048,4096").split(",")]
for n, per in cases:
    H = ldpc.codes.parity_check_csc(n, 8, 4)
    oc = BPOracle(csc=(H.indptr, H.indices), shape=H.shape, per=per, max_iters=50, dense=False)
    decs = {v: ldpc.BeliefPropagationDecoder(H, per, 50, kernel_variant=v) for v in ((1, 3, 4, 0) if ALL else (1, (0 if AUTO else 3), 4))}
    for batch in batches:
        E = ldpc.codes.random_errors(n, batch, per, seed=3)
        S = ldpc.codes.syndromes_of(H, E)
        Sd = torch.from_numpy(np.ascontiguousarray(S)).cuda()
        res = {}
        outs = {}
        for v, dec in decs.items():
            err = torch.empty((batch, n), dtype=torch.uint8, device="cuda")
            conv = torch.empty(batch, dtype=torch.uint8, device="cuda")
            for _ in range(2):
                dec.decode_batch_device(Sd, err, conv)
            torch.cuda.synchronize()
            ts = []
            for _ in range(5):
                t0 = time.perf_counter(); dec.decode_batch_device(Sd, err, conv); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
            res[v] = np.median(ts) * 1e3
            outs[v] = (err.cpu().numpy(), conv.cpu().numpy())
        same = all(np.array_equal(outs[1][0], outs[v][0]) and np.array_equal(outs[1][1], outs[v][1]) for v in list(decs)[1:])
        t0 = time.perf_counter()
        for b in range(min(batch, 4)):
            oc.decode(S[b])
        cpu = (time.perf_counter() - t0) / min(batch, 4)
        if ALL:
            best = min(res[1], res[3], res[4])
            flag = "  <-- dispatch off by >15 %" if res[0] > 1.15 * best else ""
            print(f"n {n:6d} per {per:.2f} batch {batch:5d}: tile {res[1]:9.3f}  node {res[3]:9.3f}  team {res[4]:9.3f}  auto {res[0]:9.3f} ms  "
                  f"auto/best {res[0] / best:5.2f}  identical {same}{flag}", flush=True)
            continue
        print(f"n {n:6d} per {per:.2f} batch {batch:5d}: tile {res[1]:9.3f} ms  {'auto' if AUTO else 'node'} {res[list(decs)[1]]:9.3f} ms  team {res[4]:9.3f} ms  identical {same}  "
              f"CPU oracle {cpu*1e3:7.3f} ms/syndrome", flush=True)
    for d in decs.values():
        d.close()
