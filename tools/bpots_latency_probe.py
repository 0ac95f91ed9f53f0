#!/usr/bin/env python3
"""Latency of a single BP-OTS decode! through the host-buffer entry (ldpc_bpots_decode_batch), next
to the C oracle on one core: BB-72 (n 72) and a (3,6)-regular n = 504 code, 100 iterations max."""
import os, sys, time
import numpy as np
import scipy.sparse as sp
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ldpcdecoders_jl_amd as ldpc
from oracle.bpots import BPOTSOracle

HX, _ = ldpc.codes.bivariate_bicycle_72_12_6()
for name, H, per in [("BB-72", HX, 0.01), ("(3,6) n=504", ldpc.codes.parity_check_csc(504, 6, 3), 0.03)]:
    H = sp.csc_matrix(H); H.sort_indices()
    n = H.shape[1]
    E = ldpc.codes.random_errors(n, 300, per, seed=5)
    S = ldpc.codes.syndromes_of(H, E)
    dec = ldpc.BPOTSDecoder(H, per, 100, T=9, C=3.0)
    oc = BPOTSOracle((H.indptr, H.indices), H.shape, per, 100, 9, 3.0)
    for b in range(20):
        dec.decode_batch_host(S[b:b + 1])
    ts, its = [], []
    for b in range(300):
        t0 = time.perf_counter(); r = dec.decode_batch_host(S[b:b + 1]); ts.append(time.perf_counter() - t0); its.append(int(r[2][0]))
    to = []
    for b in range(300):
        t0 = time.perf_counter(); oc.batchdecode(S[b:b + 1]); to.append(time.perf_counter() - t0)
    print(f"{name:12s} per {per}: GPU median {np.median(ts)*1e6:7.1f} us  p90 {np.percentile(ts,90)*1e6:7.1f} us   "
          f"CPU oracle median {np.median(to)*1e6:7.1f} us   (median iterations {np.median(its):.0f})")
