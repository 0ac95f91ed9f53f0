#!/usr/bin/env python3
"""Does a host-side wait expire?  (host_wait.hpp)  Decodes 4,096 syndromes of the C3 code (50 iterations) through the host entry under
wait limits of 1000, 20 and 1 ms and prints what came back and after how long."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ldpcdecoders_jl_amd as ldpc
L = ldpc._capi.lib()
H = ldpc.codes.parity_check_csc(16384, 8, 4)
syn = ldpc.codes.syndromes_of(H, ldpc.codes.random_errors(16384, 4096, 0.10, seed=1))
dec = ldpc.BeliefPropagationDecoder(H, 0.10, 50)
dec.decode_batch_host(syn[:64])
for lim in (1000, 1000, 20, 1):
    L.ldpc_set_wait_limit_ms(lim)
    t0 = time.time()
    try:
        dec.decode_batch_host(syn)
        print(f"limit {lim} ms: ok after {1e3 * (time.time() - t0):.1f} ms, kernel {dec.last_timing()[0]:.1f} ms", flush=True)
    except ldpc.LdpcError as e:
        print(f"limit {lim} ms: {e.status} after {1e3 * (time.time() - t0):.1f} ms: {e.message[:160]}", flush=True)
        break
dec.close()
