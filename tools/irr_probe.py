#!/usr/bin/env python3
"""Irregular graphs of the C3 size: whole checks in LDS (the IRR instantiation of the team kernel, round 4) against every row in
the slot (LDPC_TEAM_ROWS=0: the round-3 path) and the tile kernel; 16,384 syndromes x 50 iterations at per 0.10; results must be
identical.  Also a regular (7,4) graph -- wait, (4,7): check degree 7 has an instantiation now -- as a cross-check of the plan."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import scipy.sparse as sp
import torch
import ldpcdecoders_jl_amd as ldpc


def irregular(n, s, seed, dmin=2, dmax=6):
    rng = np.random.default_rng(seed)
    rows, cols = [], []
    for j in range(n):
        for i in rng.choice(s, int(rng.integers(dmin, dmax)), replace=False):
            rows.append(int(i)); cols.append(j)
    H = sp.csc_matrix((np.ones(len(rows), dtype=np.uint8), (rows, cols)), shape=(s, n))
    H.sum_duplicates(); H.data[:] = 1; H.sort_indices()
    return H


batch, iters, per = int(os.environ.get("BATCH", "16384")), 50, 0.10
SIZES = [int(x) for x in os.environ.get("SIZES", "16384,32768").split(",")]
GRAPHS = [(f"irregular n {n} s {n // 2} (bits 2..5)", irregular(n, n // 2, n)) for n in SIZES]
if 16384 in SIZES:
    GRAPHS.insert(1, ("irregular n 16384 s 8192 (bits 3..4)", irregular(16384, 8192, 2, 3, 5)))
for name, H in GRAPHS:
    n = H.shape[1]
    S = torch.from_numpy(ldpc.codes.syndromes_of(H, ldpc.codes.random_errors(n, batch, per, seed=3))).cuda()
    ref = None
    for cfg, variant, env in (("default", 0, {}), ("every row in the slot", 0, {"LDPC_TEAM_ROWS": "0"}), ("teams whatever the slots", 4, {"LDPC_TEAM_CACHE_MIB": "100000"}),
                              ("tile kernel", 1, {})):
        for k in ("LDPC_TEAM_ROWS", "LDPC_TEAM_CACHE_MIB"):
            os.environ.pop(k, None)
        os.environ.update(env)
        dec = ldpc.BeliefPropagationDecoder(H, per, iters, kernel_variant=variant, experiments=True)
        err = torch.empty((batch, n), dtype=torch.uint8, device="cuda"); conv = torch.empty(batch, dtype=torch.uint8, device="cuda")
        its = torch.empty(batch, dtype=torch.int32, device="cuda")
        dec.decode_batch_device(S, err, conv, None, its); dec.last_status()
        ts = []
        for _ in range(2):
            dec.decode_batch_device(S, err, conv, None, its); dec.last_status(); ts.append(dec.last_timing())
        k_ms, _, sum_iters = min(ts)
        inf = dec.info()
        same = "reference" if ref is None else ("identical" if all(torch.equal(a, b) for a, b in zip(ref, (err, conv, its))) else "DIFFERENT")
        if ref is None:
            ref = (err.clone(), conv.clone(), its.clone())
        print(f"{name}, nnz {H.nnz}, batch {batch}: {cfg:22s} kernel {k_ms:8.1f} ms  {sum_iters * 32.0 * H.nnz / (k_ms * 1e-3) / 1e12:5.2f} TB/s algorithmic  "
              f"k{inf.last_kernel} G{inf.last_team_size} slots {inf.resident_tiles // max(inf.last_team_size, 1)} rows on chip {inf.last_rows_on_chip} mean iters {sum_iters / batch:.1f}  {same}", flush=True)
        dec.close()
