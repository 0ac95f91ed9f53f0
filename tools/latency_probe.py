#!/usr/bin/env python3
"""Latency of a single decode! (BASELINE configs[0]: (3,6)-regular n=1008, per 0.01, batch 1)
through the host-buffer entry, next to the CPU oracle in both storage modes."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ldpcdecoders_jl_amd as ldpc
from oracle import BPOracle

H = ldpc.codes.parity_check_csc(1008, 6, 3)
E = ldpc.codes.random_errors(1008, 300, 0.01, seed=1)
S = ldpc.codes.syndromes_of(H, E)
dec = ldpc.BeliefPropagationDecoder(H, 0.01, 50)
for name, fn in [("decode_ (err + LLR, python mirror)", lambda b: ldpc.decode_(dec, S[b])),
                 ("decode_batch_host(1) no LLR", lambda b: dec.decode_batch_host(S[b:b + 1]))]:
    for b in range(20):
        fn(b)
    ts = []
    for b in range(300):
        t0 = time.perf_counter(); fn(b); ts.append(time.perf_counter() - t0)
    print(f"GPU  {name:40s} median {np.median(ts)*1e6:8.1f} us   p90 {np.percentile(ts,90)*1e6:8.1f} us")
for dense in (False, True):
    oc = BPOracle(csc=(H.indptr, H.indices), shape=H.shape, per=0.01, max_iters=50, dense=dense)
    ts = []
    for b in range(300):
        t0 = time.perf_counter(); oc.decode(S[b]); ts.append(time.perf_counter() - t0)
    print(f"CPU  oracle {'dense (reference-faithful)' if dense else 'edge list':33s} median {np.median(ts)*1e6:8.1f} us")

# the C3 code (n 16384): one workgroup per syndrome (bp_node_kernels.hpp) through the same entry
H = ldpc.codes.parity_check_csc(16384, 8, 4)
for per in (0.02, 0.10):
    E = ldpc.codes.random_errors(16384, 40, per, seed=2)
    S = ldpc.codes.syndromes_of(H, E)
    dec = ldpc.BeliefPropagationDecoder(H, per, 50)
    oc = BPOracle(csc=(H.indptr, H.indices), shape=H.shape, per=per, max_iters=50, dense=False)
    for name, fn in [("decode_ (err + LLR)", lambda b: ldpc.decode_(dec, S[b])), ("decode_batch_host(1) no LLR", lambda b: dec.decode_batch_host(S[b:b + 1]))]:
        for b in range(5):
            fn(b)
        ts = []
        for b in range(40):
            t0 = time.perf_counter(); fn(b); ts.append(time.perf_counter() - t0)
        print(f"GPU  n=16384 per {per:.2f} {name:28s} median {np.median(ts)*1e6:9.1f} us")
    ts = []
    for b in range(8):
        t0 = time.perf_counter(); oc.decode(S[b]); ts.append(time.perf_counter() - t0)
    print(f"CPU  n=16384 per {per:.2f} oracle edge list                median {np.median(ts)*1e6:9.1f} us")
